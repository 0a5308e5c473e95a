#!/bin/bash
# A/B of prebuilt library variants on ONE box: Localization() rates (tools/localization_rate.py), interleaved.  tools/ab_libs_loc.sh V0 V1 ...
cd "$(dirname "$0")/.."
cp superodom_amd/lib/libsoicp.so /tmp/libsoicp_keep.so
for round in 1 2 3; do
for v in "$@"; do
  cp superodom_amd/lib/libsoicp_$v.so superodom_amd/lib/libsoicp.so
  echo "== $v"; timeout 300 python tools/localization_rate.py --calls 64 --modes default,node 2>&1 | grep "^\[" | cut -c1-230
done
done
cp /tmp/libsoicp_keep.so superodom_amd/lib/libsoicp.so
