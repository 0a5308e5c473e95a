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
# rocprofv3 kernel statistics of the de-skew kernel (f4) and of Localization() (registration + device-side map insert, f1)
( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_dsk_$TAG && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_dsk_$TAG -- python $R/tools/f4_rates.py --deskew-only > /tmp/prof_dsk_$TAG.log 2>&1
  f=$(find /tmp/prof_dsk_$TAG -name "*kernel_stats.csv" | head -1); grep "Name\|deskew_kernel" $f > $R/gpurun_out/prof_$TAG/deskew_kernel_stats.csv; cat $R/gpurun_out/prof_$TAG/deskew_kernel_stats.csv )
bash tools/prof_localization.sh $TAG 2>&1 | tail -30
