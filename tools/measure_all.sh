#!/bin/bash
# Round measurement on the GPU box: full bench line (with CPU baseline), rocprofv3 kernel stats, PMC traffic.
# usage: bash tools/measure_all.sh <tag>   (outputs under gpurun_out/; copy what should be judged into profiles/)
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-run}
cd $R
python bench.py 2> gpurun_out/bench_$TAG.err | tail -1 > gpurun_out/bench_$TAG.json
cat gpurun_out/bench_$TAG.json
bash tools/prof_stats.sh $TAG
bash tools/pmc_traffic.sh $TAG
