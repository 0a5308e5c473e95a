#!/bin/bash
# A/B of library variants with the k-NN stamps: tools/ab_knn.sh V10 V11 ...
cd "$(dirname "$0")/.."
for round in 1 2; do
for v in "$@"; do
  cp superodom_amd/lib/libsoicp_$v.so superodom_amd/lib/libsoicp.so
  if [ $round = 1 ]; then
    echo "== $v stamps"; SOICP_ABLATE=128 timeout 200 python tools/eval_stamps.py --reps 6 2>&1 | grep -E "^(knn sweep|   per chunk|   workgroup|   slowest)"
  fi
  echo "== $v bench (round $round)"
  timeout 300 python bench.py --steps 48 --warmup 8 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), 'reg/s', d['ms_per_step'], 'ms', 'knn/launch', d['roofline']['avg_launch_ms'])"
done
done
