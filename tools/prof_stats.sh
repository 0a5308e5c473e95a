#!/bin/bash
# rocprofv3 kernel-trace summary of the default bench (GPU box).  usage: bash tools/prof_stats.sh <tag>
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-run}
rm -rf /tmp/prof_$TAG
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -- python $R/bench.py --steps 12 --warmup 2 --no-cpu-baseline --no-profile-pass > /tmp/prof_$TAG.log 2>&1
mkdir -p $R/gpurun_out/prof_$TAG
f=$(find /tmp/prof_$TAG -name "*kernel_stats.csv" | head -1)
cp $f $R/gpurun_out/prof_$TAG/kernel_stats.csv
tail -1 /tmp/prof_$TAG.log > $R/gpurun_out/prof_$TAG/bench_line.json
python - $f <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("kernel, calls, total_us, avg_us, pct   (14 registrations incl. warm-up)")
for r in rows[:16]:
    print(f'{r["Name"][:70]:70s} {int(r["Calls"]):6d} {float(r["TotalDurationNs"])/1e3:10.1f} {float(r["AverageNs"])/1e3:8.2f} {float(r["Percentage"]):6.2f}')
print("sum of kernel time per registration (us):", tot / 1e3 / 14)
PY
