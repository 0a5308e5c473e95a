#!/bin/bash
# A/B of prebuilt library variants on ONE box, batched hypotheses by batch size: tools/ab_batch_sizes.sh V0 V1 ...
cd "$(dirname "$0")/.."
cp superodom_amd/lib/libsoicp.so /tmp/libsoicp_keep.so
for round in 1 2; do
for v in "$@"; do
  cp superodom_amd/lib/libsoicp_$v.so superodom_amd/lib/libsoicp.so
  for h in 8 16 64; do timeout 300 python tools/batch_rate.py --hyp $h --scans 3 2>/dev/null | grep "batch mode" | sed "s/^/$v  /" | cut -c1-150; done
done
done
cp /tmp/libsoicp_keep.so superodom_amd/lib/libsoicp.so
