#!/bin/bash
# A/B of prebuilt library variants on ONE box: tools/ab_libs2.sh V0 V1 ...  (superodom_amd/lib/libsoicp_<tag>.so): driver-protocol bench lines, interleaved
cd "$(dirname "$0")/.."
cp superodom_amd/lib/libsoicp.so /tmp/libsoicp_keep.so
for round in 1 2 3; do
for v in "$@"; do
  cp superodom_amd/lib/libsoicp_$v.so superodom_amd/lib/libsoicp.so
  timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels']
print('%-8s' % '$v', 'value %.0f ms %.4f | knn us (timed loop events) %.2f | profile pass: knn %.1f solve %.1f bin %.1f' % (d['value'], d['ms_per_step'], 1e3*d['roofline']['avg_launch_ms'], 1e3*k['knn_ms_per_registration'], 1e3*k['solve_ms_per_registration'], 1e3*k['binning_ms_per_registration']))"
done
done
cp /tmp/libsoicp_keep.so superodom_amd/lib/libsoicp.so
