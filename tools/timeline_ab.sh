#!/bin/bash
# rocprofv3 kernel timeline of the registrations of bench.py's timed loop, staged vs resident entry, averaged over the timed
# registrations: start offset, duration and gap of every launch of a registration (speculated no-op launches included).
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
for entry in staged resident; do
rm -rf /tmp/tl_$entry
rocprofv3 --kernel-trace --output-format csv -d /tmp/tl_$entry -- python $R/bench.py --steps 24 --warmup 4 --entry $entry --no-cpu-baseline --no-profile-pass --no-secondary --no-kernel-events > /tmp/tl_$entry.log 2>&1
python - $entry <<'PY'
import csv, glob, sys
import numpy as np
entry = sys.argv[1]
f = glob.glob(f"/tmp/tl_{entry}/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "scan_keys" in r["Kernel_Name"]]
regs = [(idx[k], idx[k + 1]) for k in range(len(idx) - 25, len(idx) - 1)]  # the timed registrations (the last 24 complete ones)
names = None
acc = []
for a, b in regs:
    t0 = int(rows[a]["Start_Timestamp"]); prev = t0
    rec = []
    for r in rows[a:b]:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        rec.append(((s - t0) / 1e3, (e - s) / 1e3, (s - prev) / 1e3, r["Kernel_Name"].split("(")[0].replace("void soicp::", "")[:34]))
        prev = e
    rec.append(((int(rows[b]["Start_Timestamp"]) - t0) / 1e3, 0.0, (int(rows[b]["Start_Timestamp"]) - prev) / 1e3, "next scan_keys"))
    acc.append(rec)
n = min(len(r) for r in acc)
same = [r for r in acc if len(r) == len(acc[0])]
print(f"== {entry}: {len(same)} of {len(acc)} registrations with {len(acc[0])} launches")
for i in range(len(acc[0])):
    st = np.mean([r[i][0] for r in same]); du = np.mean([r[i][1] for r in same]); ga = np.mean([r[i][2] for r in same])
    print(f"  {st:8.1f} us  dur {du:7.2f}  gap {ga:6.2f}  {same[0][i][3]}")
PY
done
