#!/bin/bash
# kernel timeline of one registration (start offsets, durations, gaps) from a rocprofv3 kernel trace of bench.py
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf /tmp/tl
rocprofv3 --kernel-trace --output-format csv -d /tmp/tl -- python $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-profile-pass > /tmp/tl.log 2>&1
python - <<'PY'
import csv, glob
f = glob.glob("/tmp/tl/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
# last registration: find the last scan_keys
idx = [i for i, r in enumerate(rows) if "scan_keys" in r["Kernel_Name"]]
a = idx[-2]; b = idx[-1]
t0 = int(rows[a]["Start_Timestamp"]); prev_end = t0
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f'{(s - t0) / 1e3:8.1f} us  dur {(e - s) / 1e3:7.1f}  gap {(s - prev_end) / 1e3:6.1f}  {r["Kernel_Name"].split("(")[0][:50]}')
    prev_end = e
print("  next scan_keys starts %.1f us after the last launch above ended" % ((int(rows[b]["Start_Timestamp"]) - prev_end) / 1e3))
print("registration period (scan_keys to scan_keys): %.1f us" % ((int(rows[b]["Start_Timestamp"]) - t0) / 1e3))
PY
