#!/bin/bash
# Round-4 numbers of record, one gpurun call: the bench line (CPU baselines included), rocprofv3 kernel statistics of the bench
# and of a 64-hypothesis batch, PMC traffic + SQ counters (separate passes, tied to kernels.hip by sha256), in-kernel phase
# stamps of the instrumented k-NN instantiation, Localization() and Seam B rates.   usage: bash tools/measure_r04.sh <tag>
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r04}
cd $R; O=gpurun_out/$TAG; mkdir -p $O
timeout 1800 python -m pytest tests -q -m gpu 2>&1 | grep -E "passed|failed|error" | tail -2 | tee $O/pytest_gpu.txt
python bench.py 2> $O/bench.err | tail -1 > $O/bench_line.json
for r in 1 2 3; do python bench.py --steps 20 --warmup 5 2>> $O/bench.err | tail -1 > $O/bench_line_steps20_$r.json; done
python - $O <<'PY'
import json, sys
for r in (1, 2, 3):
    d = json.load(open(f"{sys.argv[1]}/bench_line_steps20_{r}.json"))
    print("driver protocol (--steps 20 --warmup 5) run", r, "value", round(d["value"], 1), "resident", round(d["entry_points"]["resident"], 1), "c_abi", round(d["host"]["c_abi_ms_per_step"], 4),
          "overhead", round(d["host"]["fixed_overhead_ms_per_step"], 4), "batch64", round(d["batch64"]["value"]))
PY
python - $O/bench_line.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print("value", round(d["value"], 1), "entry_points", {k: round(v, 1) for k, v in d["entry_points"].items() if k != "note"}, "knn us", round(1e3 * d["roofline"]["avg_launch_ms"], 2),
      "frac", round(d["roofline"]["frac"], 4), "traffic", d["roofline"]["traffic"], "valu_issue_frac", (d["roofline"]["valu_issue"] or {}).get("valu_issue_frac"))
print("kernels", {k: round(v, 4) for k, v in d["kernels"].items() if isinstance(v, float)}, "batch64", round(d["batch64"]["value"]), d["batch64"]["ms_per_batch"])
print("cpu", d["cpu_baseline"]["value"], d["cpu_baseline_oracle_b"].get("value"), d["cpu_baseline_all_cores"].get("value"), d["parity_vs_oracle_m_rad"], d["parity_iteration_counts_and_histograms_equal"])
PY
bash tools/prof_stats.sh $TAG 2>&1 | tail -24 | tee $O/prof_stats.txt
cp gpurun_out/prof_$TAG/*.csv $O/ 2>/dev/null; cp gpurun_out/prof_$TAG/bench_line.json $O/bench_line_under_rocprofv3.json; cp gpurun_out/prof_$TAG/bench_line_no_speculation.json $O/bench_line_under_rocprofv3_no_speculation.json
( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_batch_$TAG && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_batch_$TAG -- python $R/tools/batch_rate.py --scans 3 > /tmp/prof_batch_$TAG.log 2>&1
  f=$(find /tmp/prof_batch_$TAG -name "*kernel_stats.csv" | head -1); cp $f $R/$O/batch_kernel_stats.csv; grep "batch mode" /tmp/prof_batch_$TAG.log ) | tee $O/batch_under_rocprofv3.txt
python tools/batch_rate.py 2>&1 | grep "batch mode" | tee $O/batch_rate.txt
bash tools/pmc_traffic.sh $TAG 2>&1 | tail -1 > $O/pmc_traffic.log; cp gpurun_out/pmc_$TAG/* $O/ 2>/dev/null
bash tools/pmc_knn.sh 0 2>&1 | tail -18 > $O/sq_counters_knn_solve.txt; cp gpurun_out/pmc_knn/knn_counters.json $O/
SOICP_ABLATE=128 python tools/eval_stamps.py 2>&1 | tail -28 | tee $O/phase_stamps_instrumented_build.txt
python tools/knn_pack_stats.py 2>&1 | tail -4 | tee $O/knn_pack_stats.txt
python tools/batch1_rate.py 2>&1 | tail -6 | tee $O/batch1_rate.txt
python tools/localization_rate.py 2>&1 | tail -1 | tee $O/localization_rate.txt
python tools/seam_b_rate.py 2>&1 | tail -4 | tee $O/seam_b_rate.txt
bash tools/prof_localization.sh $TAG 2>&1 | tail -3; cp gpurun_out/prof_$TAG/localization_kernel_stats.csv $O/ 2>/dev/null
