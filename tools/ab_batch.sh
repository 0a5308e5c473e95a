#!/bin/bash
# A/B of the batched-hypothesis path on ONE box: tools/batch_rate.py per (library variant, environment) pair.
# usage: bash tools/ab_batch.sh <outtag> "<variant>[:ENV=V[,ENV=V]]" ...
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
OUT=gpurun_out/$1; shift; mkdir -p $OUT
for round in 1 2; do
for spec in "$@"; do
  v=${spec%%:*}; envs=""; [[ $spec == *:* ]] && envs=$(echo ${spec#*:} | tr ',' ' ')
  cp superodom_amd/lib/libsoicp_$v.so superodom_amd/lib/libsoicp.so
  echo -n "$spec round $round: "; env $envs timeout 300 python tools/batch_rate.py 2>&1 | grep "batch mode" | sed 's/converged.*histogram/hist/'
done
done | tee $OUT/ab_batch.txt
