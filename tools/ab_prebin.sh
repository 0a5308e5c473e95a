#!/bin/bash
# A/B on one box: scans binned ahead of their registration (default) vs SOICP_PREBIN=0.  usage: bash tools/ab_prebin.sh <tag> [t]
R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=${1:-abp}; cd $R; O=gpurun_out/$TAG; mkdir -p $O
if [[ "$2" == *t* ]]; then
  timeout 900 python -m pytest tests/test_gpu_binned_ahead.py tests/test_gpu_configs.py tests/test_gpu_adapter.py tests/test_gpu_node.py -q -m gpu -x 2>&1 | tail -25 | tee $O/pytest_subset.log
fi
one() { python - "$1" "$2" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
k = d.get("kernels") or {}
print("%-10s value %.1f ms/step %.4f | resident %.1f | knn us %.2f | binning %.4f knn %.4f solve %.4f rest %.4f | parity %s %s" % (sys.argv[2], d["value"], d["ms_per_step"], d["entry_points"]["resident"],
      1e3 * d["roofline"]["avg_launch_ms"], k.get("binning_ms_per_registration", -1), k.get("knn_ms_per_registration", -1), k.get("solve_ms_per_registration", -1), k.get("rest_ms_per_registration", -1),
      d.get("parity_vs_oracle_m_rad"), d.get("parity_iteration_counts_and_histograms_equal")))
PY
}
for r in 1 2 3; do
  SOICP_PREBIN=0 timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary 2>>$O/err | tail -1 > $O/off_$r.json; one $O/off_$r.json "off s20"
  timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary 2>>$O/err | tail -1 > $O/on_$r.json; one $O/on_$r.json "on  s20"
done
SOICP_PREBIN=0 timeout 600 python bench.py --no-cpu-baseline --no-secondary 2>>$O/err | tail -1 > $O/off_240.json; one $O/off_240.json "off 240"
timeout 600 python bench.py --no-cpu-baseline --no-secondary 2>>$O/err | tail -1 > $O/on_240.json; one $O/on_240.json "on  240"
tail -5 $O/err
