#!/bin/bash
# rocprofv3 kernel-trace summary of the default bench (GPU box).  usage: bash tools/prof_stats.sh <tag>
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-run}
rm -rf /tmp/prof_$TAG
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -- python $R/bench.py --steps 12 --warmup 2 --no-cpu-baseline --no-profile-pass --no-secondary > /tmp/prof_$TAG.log 2>&1
mkdir -p $R/gpurun_out/prof_$TAG
f=$(find /tmp/prof_$TAG -name "*kernel_stats.csv" | head -1)
cp $f $R/gpurun_out/prof_$TAG/kernel_stats.csv
grep "^{\"metric" /tmp/prof_$TAG.log | tail -1 > $R/gpurun_out/prof_$TAG/bench_line.json
# second pass without speculative enqueue: every knn_plane_kernel / solve_kernel launch in it is a real one, so the
# rocprofv3 averages can be compared directly with the HIP-event averages bench.py reports (which exclude no-op launches)
rm -rf /tmp/prof_${TAG}_ns
SOICP_SPECULATE=0 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_${TAG}_ns -- python $R/bench.py --steps 12 --warmup 2 --no-cpu-baseline --no-profile-pass --no-secondary > /tmp/prof_${TAG}_ns.log 2>&1
g=$(find /tmp/prof_${TAG}_ns -name "*kernel_stats.csv" | head -1)
cp $g $R/gpurun_out/prof_$TAG/kernel_stats_no_speculation.csv
grep "^{\"metric" /tmp/prof_${TAG}_ns.log | tail -1 > $R/gpurun_out/prof_$TAG/bench_line_no_speculation.json
python - $g $R/gpurun_out/prof_$TAG/bench_line_no_speculation.json <<'PY'
import csv, json, sys
rows = list(csv.DictReader(open(sys.argv[1])))
d = json.load(open(sys.argv[2]))
for r in rows:
    if "knn_plane_kernel" in r["Name"] or "solve_kernel" in r["Name"]:
        print("no speculation:", r["Name"].split("(")[0], "calls", r["Calls"], "avg us %.2f" % (float(r["AverageNs"]) / 1e3))
print("no speculation: (compare with roofline.avg_launch_ms of the UNPROFILED bench line; under rocprofv3 the HIP events of this run read %.2f us)" % (1e3 * d["roofline"]["avg_launch_ms"]))
PY
python - $f <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("kernel, calls, total_us, avg_us, pct   (14 registrations incl. warm-up)")
for r in rows[:16]:
    print(f'{r["Name"][:70]:70s} {int(r["Calls"]):6d} {float(r["TotalDurationNs"])/1e3:10.1f} {float(r["AverageNs"])/1e3:8.2f} {float(r["Percentage"]):6.2f}')
print("sum of kernel time per registration (us):", tot / 1e3 / 14)
PY
