#!/bin/bash
# kernel timeline of registrations at the stock operating point (tools/stock_rate.py under rocprofv3 --kernel-trace).  usage: bash tools/stock_timeline.sh [case] [n registrations shown]
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
CASE=${1:-os1_128}
rm -rf /tmp/stl
rocprofv3 --kernel-trace --output-format csv -d /tmp/stl -- python $R/tools/stock_rate.py --case $CASE --calls 24 --reps 1 > /tmp/stl.log 2>&1
tail -1 /tmp/stl.log
python - ${2:-2} <<'PY'
import csv, glob, sys
f = glob.glob("/tmp/stl/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-int(sys.argv[1]) * 8:]
t0 = int(rows[0]["Start_Timestamp"]); prev_end = t0
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f'{(s - t0) / 1e3:8.1f} us  dur {(e - s) / 1e3:7.1f}  gap {(s - prev_end) / 1e3:6.1f}  grid {r.get("Grid_Size_X", "?"):>7}  {r["Kernel_Name"].split("(")[0][:60]}')
    prev_end = e
PY
