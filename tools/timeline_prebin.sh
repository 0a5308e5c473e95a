#!/bin/bash
# kernel timeline of the staged protocol with scans binned ahead: every dispatch of two steady-state registrations, both queues
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf /tmp/tlp
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/tlp -- python $R/bench.py --steps 24 --warmup 4 --no-cpu-baseline --no-profile-pass --no-secondary --no-kernel-events > /tmp/tlp.log 2>&1
tail -2 /tmp/tlp.log | cut -c1-300
python - <<'PY'
import csv, glob
f = glob.glob("/tmp/tlp/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
m = glob.glob("/tmp/tlp/**/*memory_copy_trace.csv", recursive=True)
copies = sorted(csv.DictReader(open(m[0])), key=lambda r: int(r["Start_Timestamp"])) if m else []
anchor = [i for i, r in enumerate(rows) if "reg_begin_prebinned" in r["Kernel_Name"]]
if len(anchor) < 8:
    anchor = [i for i, r in enumerate(rows) if "scan_keys" in r["Kernel_Name"]]
print("columns:", list(rows[0].keys()))
for k in (-7, -4):
    a, b = anchor[k], anchor[k + 1]
    t0, t1 = int(rows[a]["Start_Timestamp"]), int(rows[b]["Start_Timestamp"])
    print("---- registration", k, "period %.1f us" % ((t1 - t0) / 1e3))
    ev = []
    for r in rows:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        if t0 <= s < t1:
            ev.append((s, "K q%-3s %-34s dur %6.2f end %7.2f" % (r.get("Queue_Id", "?"), r["Kernel_Name"].split("(")[0].replace("void soicp::", "")[:34], (e - s) / 1e3, (e - t0) / 1e3)))
    for r in copies:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        if t0 <= s < t1:
            ev.append((s, "C      %-34s dur %6.2f end %7.2f" % (r.get("Direction", r.get("Name", "copy"))[:34], (e - s) / 1e3, (e - t0) / 1e3)))
    for s, txt in sorted(ev):
        print("%9.2f  %s" % ((s - t0) / 1e3, txt))
PY
