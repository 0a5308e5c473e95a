#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/g29; mkdir -p $O
cd $R
for a in 32768 16384 32768 16384; do
SOICP_ABLATE=$a python bench.py --no-cpu-baseline --no-secondary --entry resident --steps 100 2>/dev/null | tail -1 > $O/bench_line.json
python - $O/bench_line.json $a <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print("ablate", sys.argv[2], "value", round(d["value"], 1), "knn us", round(1e3 * d["roofline"]["avg_launch_ms"], 2), "kernels", {k: round(v, 4) for k, v in d["kernels"].items() if isinstance(v, float)}, d["executed"])
PY
done
