#!/bin/bash
# Round 3, one gpurun call: batched-hypothesis tests first (fast failure), the whole -m gpu suite, the bench line, the batch A/B,
# rocprofv3 kernel statistics.  usage: bash tools/round3_gpu.sh <tag> [stages]   stages: any of t s b a p c (default: all)
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r03a}
ST=${2:-tsbapc}
cd $R
O=gpurun_out/$TAG
mkdir -p $O
if [[ $ST == *t* ]]; then
  timeout 900 python -m pytest tests/test_gpu_configs.py -x -q -m gpu -k "batched_hypotheses or batch64" 2>&1 | tail -40 > $O/pytest_batch.log; tail -5 $O/pytest_batch.log
fi
if [[ $ST == *s* ]]; then
  timeout 1800 python -m pytest tests -q -m gpu --durations=8 2>&1 | tail -80 > $O/pytest_gpu.log; tail -15 $O/pytest_gpu.log
fi
if [[ $ST == *b* ]]; then
  timeout 900 python bench.py 2> $O/bench.err | tail -1 > $O/bench.json; python - $O/bench.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print("value", round(d["value"], 1), "ms/step", round(d["ms_per_step"], 4), "entry_points", {k: (round(v, 1) if isinstance(v, float) else v) for k, v in d["entry_points"].items() if k != "note"})
print("knn us", round(1e3 * d["roofline"]["avg_launch_ms"], 2), "frac", round(d["roofline"]["frac"], 4), "kernels", {k: (round(v, 4) if isinstance(v, float) else v) for k, v in (d.get("kernels") or {}).items() if k != "note"})
print("batch64", {k: (round(v, 2) if isinstance(v, float) else v) for k, v in (d.get("batch64") or {}).items() if k not in ("parallelism",)})
print("cpu", d.get("cpu_baseline", {}).get("value"), d.get("cpu_baseline_oracle_b", {}).get("value"), d.get("cpu_baseline_all_cores", {}).get("value"), "parity", d.get("parity_vs_oracle_m_rad"), d.get("parity_iteration_counts_and_histograms_equal"))
PY
fi
if [[ $ST == *a* ]]; then
  ( python tools/batch_rate.py; SOICP_BATCH_WG_PER_CU=1 python tools/batch_rate.py; SOICP_BATCH_MODE=lanes python tools/batch_rate.py ) 2>&1 | grep "batch mode" | tee $O/batch_ab.txt
fi
if [[ $ST == *p* ]]; then
  bash tools/prof_stats.sh $TAG 2>&1 | tail -32 | tee $O/prof_stats.txt
  # kernel statistics of the batched path (one 64-hypothesis batch per scan)
  ( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_batch_$TAG && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_batch_$TAG -- python $R/tools/batch_rate.py --scans 2 > /tmp/prof_batch_$TAG.log 2>&1
    f=$(find /tmp/prof_batch_$TAG -name "*kernel_stats.csv" | head -1); cp $f $R/$O/batch_kernel_stats.csv; grep "batch mode" /tmp/prof_batch_$TAG.log
    python - $f <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:10]:
    print(f'{r["Name"][:60]:60s} calls {int(r["Calls"]):5d} total_us {float(r["TotalDurationNs"])/1e3:10.1f} avg_us {float(r["AverageNs"])/1e3:9.2f} pct {float(r["Percentage"]):6.2f}')
PY
  ) 2>&1 | tee $O/batch_prof.txt
fi
if [[ $ST == *c* ]]; then
  bash tools/pmc_traffic.sh $TAG 2>&1 | tail -2
  bash tools/pmc_knn.sh 0 2>&1 | tail -24
fi
