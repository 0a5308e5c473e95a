#!/bin/bash
# Round measurement on the GPU box: full bench line (with the CPU baselines), rocprofv3 kernel statistics (with and without
# speculation), PMC traffic + SQ counters (separate passes), Localization() and Seam B rates.
# usage: bash tools/measure_all.sh <tag>   (outputs under gpurun_out/; copy what should be judged into profiles/)
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-run}
cd $R
mkdir -p gpurun_out
python bench.py 2> gpurun_out/bench_$TAG.err | tail -1 > gpurun_out/bench_$TAG.json
cat gpurun_out/bench_$TAG.json
bash tools/prof_stats.sh $TAG 2>&1 | tail -40
bash tools/pmc_traffic.sh $TAG 2>&1 | tail -3
bash tools/pmc_knn.sh 0 2>&1 | tail -20
python tools/localization_rate.py 2>&1 | tail -2 | tee gpurun_out/localization_$TAG.txt
python tools/seam_b_rate.py 2>&1 | tail -4 | tee gpurun_out/seam_b_$TAG.txt
python tools/f4_rates.py 2>&1 | grep "^deskew\|^node shell" | tee gpurun_out/f4_rates_$TAG.txt
