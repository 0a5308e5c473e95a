#!/bin/bash
# Round 4, one gpurun call.  usage: bash tools/round4_gpu.sh <tag> [stages] [variants...]
#   t  the whole -m gpu suite on superodom_amd/lib/libsoicp.so
#   a  A/B of prebuilt library variants (superodom_amd/lib/libsoicp_<v>.so): in-kernel solve stamps, driver-protocol bench lines
#   g  registered vs pageable host scan buffers (staging path) under the driver's protocol
#   h  so_icp_register_batch with ONE hypothesis against so_icp_register_dev
#   b  one full bench line (default arguments) -> $O/bench.json
#   p  rocprofv3 kernel statistics (tools/prof_stats.sh)
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r04a}
ST=${2:-tagh}
shift; shift
VARS="$@"
cd $R
O=gpurun_out/$TAG
mkdir -p $O
line() { python - "$1" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
except Exception as e:
    print("no bench line:", e); sys.exit(0)
h = d.get("host", {})
print("value %.1f  ms/step %.4f  c_abi %.4f  overhead %.4f  stage_wait %.4f  dma/copied/declined %s/%s/%s | entry_points %s" % (
    d["value"], d["ms_per_step"], h.get("c_abi_ms_per_step", 0), h.get("fixed_overhead_ms_per_step", 0), h.get("stage_wait_ms_per_step", 0),
    h.get("staged_by_dma_from_registered_memory"), h.get("staged_through_copy_thread"), h.get("stage_declined"),
    {k: round(v, 1) for k, v in d["entry_points"].items() if k != "note"}))
k = d.get("kernels") or {}
print("   knn us %.2f frac %.4f | per registration: knn %.4f solve %.4f binning %.4f rest %.4f | batch64 %s | parity %s %s" % (
    1e3 * d["roofline"]["avg_launch_ms"], d["roofline"]["frac"], k.get("knn_ms_per_registration", 0), k.get("solve_ms_per_registration", 0),
    k.get("binning_ms_per_registration", 0), k.get("rest_ms_per_registration", 0), round((d.get("batch64") or {}).get("value", 0), 1),
    d.get("parity_vs_oracle_m_rad"), d.get("parity_iteration_counts_and_histograms_equal")))
PY
}
if [[ $ST == *t* ]]; then
  timeout 1800 python -m pytest tests -q -m gpu --durations=6 2>&1 | tail -120 > $O/pytest_gpu.log; tail -12 $O/pytest_gpu.log
fi
if [[ $ST == *a* ]]; then
  cp superodom_amd/lib/libsoicp.so /tmp/libsoicp_default.so
  for round in 1 2; do
  for v in $VARS; do
    cp superodom_amd/lib/libsoicp_$v.so superodom_amd/lib/libsoicp.so
    if [ $round = 1 ]; then
      echo "== $v stamps"; SOICP_ABLATE=128 timeout 200 python tools/eval_stamps.py --reps 8 2>&1 | grep -E "^(fit|eval) " | tee -a $O/ab_stamps_$v.txt
    fi
    echo "== $v bench --steps 20 --warmup 5 (round $round)"
    timeout 300 python bench.py --steps 20 --warmup 5 --cpu-sample 1 2>$O/ab_$v.err | tail -1 > $O/ab_${v}_$round.json; line $O/ab_${v}_$round.json
  done
  done
  cp /tmp/libsoicp_default.so superodom_amd/lib/libsoicp.so
fi
if [[ $ST == *g* ]]; then
  # staging matrix: where the host scan buffers live x who waits for a DMA-staged copy, under the driver's protocol and a long run
  for m in pinned; do
    for at in 0 1 2; do w=device
      [ $m = pageable ] && [ $w = host ] && continue
      echo "== staging: scan buffers $m, copy enqueued at $at"
      for k in 20 240; do
        SOICP_STAGE_AT=$at SOICP_STAGE_WAIT=$w timeout 300 python bench.py --steps $k --warmup 5 --no-cpu-baseline --no-profile-pass --scan-buffers $m 2>>$O/stage.err | tail -1 > $O/stage_${m}_at${at}_$k.json; line $O/stage_${m}_at${at}_$k.json
      done
    done
  done
fi
if [[ $ST == *h* ]]; then
  timeout 300 python tools/batch1_rate.py 2>&1 | tail -4 | tee $O/batch1.txt
fi
if [[ $ST == *b* ]]; then
  timeout 900 python bench.py 2> $O/bench.err | tail -1 > $O/bench.json; line $O/bench.json
  timeout 300 python bench.py --steps 20 --warmup 5 2> $O/bench20.err | tail -1 > $O/bench20.json; line $O/bench20.json
fi
if [[ $ST == *p* ]]; then
  bash tools/prof_stats.sh $TAG 2>&1 | tail -32 | tee $O/prof_stats.txt
fi
