#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
for round in 1 2 3; do
for v in base prio; do
  cp superodom_amd/lib/libsoicp_$v.so superodom_amd/lib/libsoicp.so
  python bench.py --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 > /tmp/bl.json
  python - /tmp/bl.json $v <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print(sys.argv[2], "value", round(d["value"], 1), "knn us", round(1e3 * d["roofline"]["avg_launch_ms"], 2), "kernels", {k: round(v, 4) for k, v in d["kernels"].items() if isinstance(v, float) and 'launches' not in k})
PY
done
done
