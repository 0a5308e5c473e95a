#!/bin/bash
# kernel timeline of the registrations of one so_icp_register_sequence call (tools/seq_rate.py under rocprofv3 --kernel-trace).  usage: bash tools/seq_timeline.sh [--stock] 
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf /tmp/sql
rocprofv3 --kernel-trace --output-format csv -d /tmp/sql -- python $R/tools/seq_rate.py --count 16 --reps 2 --no-single $1 > /tmp/sql.log 2>&1
grep "sequence" /tmp/sql.log | tail -1
python - <<'PY'
import csv, glob
f = glob.glob("/tmp/sql/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
main = [r for r in rows if "knn_" in r["Kernel_Name"] or "solve_kernel" in r["Kernel_Name"]]
rows = rows[-44:]
t0 = int(rows[0]["Start_Timestamp"]); prev_end = {}
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    q = r.get("Queue_Id", "?")
    print(f'{(s - t0) / 1e3:8.1f} us  dur {(e - s) / 1e3:7.1f}  gap(same queue) {(s - prev_end.get(q, s)) / 1e3:6.1f}  q {q:>3}  {r["Kernel_Name"].split("(")[0][:56]}')
    prev_end[q] = e
PY
