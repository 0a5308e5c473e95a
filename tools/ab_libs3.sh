#!/bin/bash
# A/B of prebuilt library variants on ONE box: in-kernel phase stamps of the solve passes + driver-protocol bench lines, interleaved.  tools/ab_libs3.sh V0 V1 ...
cd "$(dirname "$0")/.."
cp superodom_amd/lib/libsoicp.so /tmp/libsoicp_keep.so
for round in 1 2 3; do
for v in "$@"; do
  cp superodom_amd/lib/libsoicp_$v.so superodom_amd/lib/libsoicp.so
  if [ $round = 1 ]; then echo "== $v stamps"; SOICP_ABLATE=128 timeout 200 python tools/eval_stamps.py --reps 8 2>&1 | grep -E "^(fit|eval) "; fi
  timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels']
print('%-8s' % '$v', 'value %.0f ms %.4f | knn us %.2f | profile pass: knn %.1f solve %.1f' % (d['value'], d['ms_per_step'], 1e3*d['roofline']['avg_launch_ms'], 1e3*k['knn_ms_per_registration'], 1e3*k['solve_ms_per_registration']))"
done
done
cp /tmp/libsoicp_keep.so superodom_amd/lib/libsoicp.so
