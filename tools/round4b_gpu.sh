#!/bin/bash
# Round 4, second session: the device-built map insert.  usage: bash tools/round4b_gpu.sh <tag> [stages]
#   m  tests/test_gpu_map_update.py (default switches; on failure again with SOICP_MAP_FAST=0 to tell the two paths apart)
#   t  the whole -m gpu suite
#   l  Localization() rate: device-built + deferred / device-built, waiting / host-built rounds
#   p  rocprofv3: timeline of one Localization() call and kernel statistics
#   s  soaks: map insert against the oracle, Localization sequences
#   b  one bench line (driver protocol)
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r04b}
ST=${2:-mtlpsb}
cd $R
O=gpurun_out/$TAG
mkdir -p $O
if [[ $ST == *m* ]]; then
  timeout 600 python -m pytest tests/test_gpu_map_update.py -x -q -m gpu -s 2>&1 | tail -40 > $O/pytest_map.log; tail -15 $O/pytest_map.log
  if ! grep -q " passed" $O/pytest_map.log || grep -q "failed" $O/pytest_map.log; then
    echo "== again with SOICP_MAP_FAST=0"
    SOICP_MAP_FAST=0 timeout 600 python -m pytest tests/test_gpu_map_update.py -q -m gpu 2>&1 | tail -40 > $O/pytest_map_hostbuilt.log; tail -15 $O/pytest_map_hostbuilt.log
  fi
fi
if [[ $ST == *t* ]]; then
  timeout 1500 python -m pytest tests -q -m gpu --durations=6 2>&1 | tail -120 > $O/pytest_gpu.log; tail -14 $O/pytest_gpu.log
fi
if [[ $ST == *l* ]]; then
  timeout 400 python tools/localization_rate.py --calls 64 --modes default,hostbuilt,staged,staged_hostbuilt,node,node_pageable,node_samequeue,node_hostbuilt,node_r3,node 2>&1 | tail -8 | tee $O/localization_rate.txt
fi
if [[ $ST == *p* ]]; then
  timeout 400 bash tools/localization_timeline.sh $TAG 2>&1 | tail -60
  timeout 400 bash tools/prof_localization.sh $TAG 2>&1 | tail -45
fi
if [[ $ST == *n* ]]; then
  timeout 400 bash tools/localization_timeline.sh $TAG node 2>&1 | tail -70
fi
if [[ $ST == *f* ]]; then
  timeout 300 python tools/soak_prefilter.py --seconds 30 2>&1 | tail -4 | tee $O/soak_prefilter.txt
  timeout 300 python -m pytest tests/test_gpu_node.py -x -q -m gpu 2>&1 | tail -4
fi
if [[ $ST == *s* ]]; then
  timeout 500 python tools/soak_map_insert.py --oracle --seconds 40 2>&1 | tail -6 | tee $O/soak_map_insert.txt
  timeout 500 python tools/soak_localization.py --seconds 40 2>&1 | tail -6 | tee $O/soak_localization.txt
fi
if [[ $ST == *e* ]]; then
  # A/B on one box: the watchdog event behind the reporting launch (1) against hipStreamQuery (0, default)
  for rep in 1 2; do for ev in 1 0; do
    SOICP_OUTER_EVENTS=$ev timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-profile-pass 2>>$O/ab_events.err | tail -1 > $O/ab_events_${ev}_$rep.json
    python - $O/ab_events_${ev}_$rep.json $ev <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("outer_events=%s value %.1f ms/step %.4f | entry_points %s | batch64 %s | parity %s" % (sys.argv[2], d["value"], d["ms_per_step"],
      {k: round(v, 1) for k, v in d["entry_points"].items() if k != "note"}, round((d.get("batch64") or {}).get("value", 0), 1), d.get("parity_vs_oracle_m_rad")))
PY
  done; done
fi
if [[ $ST == *b* ]]; then
  timeout 600 python bench.py --steps 20 --warmup 5 2>$O/bench.err | tail -1 > $O/bench_steps20.json
  python - $O/bench_steps20.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("value %.1f ms/step %.4f | entry_points %s | knn us %.2f frac %.4f | batch64 %s | parity %s" % (
    d["value"], d["ms_per_step"], {k: round(v, 1) for k, v in d["entry_points"].items() if k != "note"}, 1e3 * d["roofline"]["avg_launch_ms"],
    d["roofline"]["frac"], round((d.get("batch64") or {}).get("value", 0), 1), d.get("parity_vs_oracle_m_rad")))
print("localization", {k: (round(v, 4) if isinstance(v, float) else v) for k, v in (d.get("localization") or {}).items() if k != "note"})
PY
fi
