#!/bin/bash
# A/B of compiler flags on the GPU box: rebuild libsoicp.so with extra flags, bench, restore the shipped library.
# usage: bash tools/ab_flag.sh <tag> "<flags>" ["<flags2>" ...]
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
TAG=$1; shift
cp superodom_amd/lib/libsoicp.so /tmp/libsoicp_base.so
bash tools/bench_only.sh ${TAG}_base 2>&1 | sed 's/^/base: /'
i=0
for F in "$@"; do
  i=$((i+1))
  if SOICP_EXTRA_CXXFLAGS="$F" python -m superodom_amd.build --force > /tmp/build_$i.log 2>&1; then
    bash tools/bench_only.sh ${TAG}_$i 2>&1 | sed "s/^/[$F]: /"
  else echo "[$F]: build failed"; grep -i "error\|unknown" /tmp/build_$i.log | head -3; fi
done
cp /tmp/libsoicp_base.so superodom_amd/lib/libsoicp.so
