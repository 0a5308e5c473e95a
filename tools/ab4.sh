#!/bin/bash
# A/B on one box: bench (single) + batch rate per library variant, then the GPU suite on the last variant
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
OUT=gpurun_out/$1; shift; mkdir -p $OUT
for round in 1 2; do
for v in "$@"; do
  cp superodom_amd/lib/libsoicp_$v.so superodom_amd/lib/libsoicp.so
  timeout 300 python bench.py --no-cpu-baseline --no-secondary --steps 240 --warmup 8 2>$OUT/err_$v.txt | tail -1 > $OUT/bench_${v}_$round.json
  python - $OUT/bench_${v}_$round.json $v $round <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1])); k = d["kernels"]
    print(sys.argv[2], "round", sys.argv[3], "value %.1f" % d["value"], "knn_launch_us %.2f" % (1e3 * d["roofline"]["avg_launch_ms"]),
          "knn %.1f solve %.1f bin %.1f rest %.1f" % (1e3 * k["knn_ms_per_registration"], 1e3 * k["solve_ms_per_registration"], 1e3 * k["binning_ms_per_registration"], 1e3 * k["rest_ms_per_registration"]))
except Exception as e:
    print(sys.argv[2], "round", sys.argv[3], "FAILED", e)
PY
  echo -n "$v round $round: "; timeout 300 python tools/batch_rate.py 2>&1 | grep "batch mode" | sed 's/converged.*histogram/hist/'
done
done 2>&1 | tee $OUT/ab.txt
timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | tail -6
