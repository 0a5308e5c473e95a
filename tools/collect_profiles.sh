#!/bin/bash
# copy the outputs of tools/measure_all.sh <tag> (+ optional two-rank run) from gpurun_out/ into profiles/r02/ (tracked)
# usage (development container): bash tools/collect_profiles.sh <tag>
R=/root/repo; TAG=${1:-r02}; D=$R/profiles/r02
mkdir -p $D/pmc
cp $R/gpurun_out/bench_$TAG.json $D/bench_line.json
cp $R/gpurun_out/prof_$TAG/kernel_stats.csv $D/kernel_stats.csv
cp $R/gpurun_out/prof_$TAG/kernel_stats_no_speculation.csv $D/kernel_stats_no_speculation.csv
cp $R/gpurun_out/prof_$TAG/bench_line.json $D/bench_line_under_rocprofv3.json
cp $R/gpurun_out/prof_$TAG/bench_line_no_speculation.json $D/bench_line_under_rocprofv3_no_speculation.json
cp $R/gpurun_out/pmc_$TAG/p_FETCH_SIZE.csv $R/gpurun_out/pmc_$TAG/p_WRITE_SIZE.csv $D/pmc/
cp $R/gpurun_out/pmc_$TAG/knn_traffic.json $D/pmc/knn_traffic.json
cp $R/gpurun_out/pmc_knn/ablate_0.txt $D/pmc/sq_counters_knn_solve.txt
cp $R/gpurun_out/pmc_knn/knn_counters.json $D/pmc/knn_counters.json
cp $R/gpurun_out/localization_$TAG.txt $D/localization_rate.txt
cp $R/gpurun_out/seam_b_$TAG.txt $D/seam_b_rate.txt
[ -f $R/gpurun_out/f4_rates_$TAG.txt ] && cp $R/gpurun_out/f4_rates_$TAG.txt $D/f4_rates.txt
for f in deskew_kernel_stats.csv localization_kernel_stats.csv; do [ -f $R/gpurun_out/prof_$TAG/$f ] && cp $R/gpurun_out/prof_$TAG/$f $D/$f; done
for n in 2 4 8; do [ -f $R/gpurun_out/ranks$n.json ] && cp $R/gpurun_out/ranks$n.json $D/ranks${n}_on_one_gpu_bench_line.json; done
# the two files bench.py reads (current round)
cp $D/pmc/knn_traffic.json $R/profiles/knn_traffic.json
cp $D/pmc/knn_counters.json $R/profiles/knn_counters.json
ls -la $D $D/pmc
