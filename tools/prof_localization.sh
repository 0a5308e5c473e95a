#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
python $R/tools/localization_rate.py
rm -rf /tmp/prof_loc
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_loc -- python $R/tools/localization_rate.py --calls 10 > /tmp/prof_loc.log 2>&1
f=$(find /tmp/prof_loc -name "*kernel_stats.csv" | head -1)
mkdir -p $R/gpurun_out/prof_loc; cp $f $R/gpurun_out/prof_loc/kernel_stats.csv
python - $f <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:24]:
    print(f'{r["Name"][:64]:64s} {int(r["Calls"]):6d} {float(r["TotalDurationNs"])/1e3:10.1f} {float(r["AverageNs"])/1e3:8.2f} {float(r["Percentage"]):6.2f}')
print("sum of kernel time per call (us), 12 calls:", tot / 1e3 / 12)
PY
