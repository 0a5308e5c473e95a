#!/bin/bash
# A/B of prebuilt library variants (superodom_amd/lib/libsoicp_<tag>.so) on ONE box: bench value + k-NN launch time, three rounds
cd ${GRAFT_REPO_ROOT:-/root/repo}
for round in 1 2 3; do
for v in "$@"; do
  cp superodom_amd/lib/libsoicp_$v.so superodom_amd/lib/libsoicp.so
  echo -n "$v round $round: "
  timeout 300 python bench.py --steps 96 --warmup 8 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), 'reg/s | knn us', round(1e3*d['roofline']['avg_launch_ms'],2), '| kernels', {k: round(1e3*v,1) for k,v in d['kernels'].items() if k.endswith('ms_per_registration')})"
done
done
