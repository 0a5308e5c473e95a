#!/bin/bash
# experiment: rocPRIM merge-sort tunings for the per-registration scan sort (SOICP_SORT_CFG)
for c in ${1:-0 1 2 3 4 5}; do
  SOICP_SORT_CFG=$c python bench.py --no-cpu-baseline --steps 24 --time-all-kernels 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('cfg $c prep+sort ms/step', round(d['kernels']['prep_sort_ms_per_step'],4), 'value', round(d['value'],1))"
done
