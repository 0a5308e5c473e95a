#!/bin/bash
# rocprofv3 kernel statistics of Localization() (registration + device-side map insert) at BASELINE sizes.
# usage (GPU box): bash tools/prof_localization.sh <tag>
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-run}
rm -rf /tmp/prof_loc_$TAG
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_loc_$TAG -- python $R/tools/localization_rate.py --calls 30 > /tmp/prof_loc_$TAG.log 2>&1
mkdir -p $R/gpurun_out/prof_$TAG
f=$(find /tmp/prof_loc_$TAG -name "*kernel_stats.csv" | head -1)
cp $f $R/gpurun_out/prof_$TAG/localization_kernel_stats.csv
tail -1 /tmp/prof_loc_$TAG.log
python - $f <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
print("kernel, calls, total_us, avg_us, pct   (32 Localization() calls + one map load)")
for r in rows[:40]:
    print(f'{r["Name"][:90]:90s} {int(r["Calls"]):6d} {float(r["TotalDurationNs"])/1e3:10.1f} {float(r["AverageNs"])/1e3:8.2f} {float(r["Percentage"]):6.2f}')
PY
