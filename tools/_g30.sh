#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/g30; mkdir -p $O
cd $R
timeout 900 python -m pytest tests -q -m gpu -x -k "parity or bit or knn or configs" 2>&1 | grep -E "passed|failed|error|Error|assert" | tail -5 | tee $O/pytest.txt
for i in 1 2 3; do
python bench.py --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 > $O/bench_line.json
python - $O/bench_line.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print("value", round(d["value"], 1), "resident", round(d["entry_points"].get("resident", 0), 1), "knn us", round(1e3 * d["roofline"]["avg_launch_ms"], 2), "kernels", {k: round(v, 4) for k, v in d["kernels"].items() if isinstance(v, float)})
PY
done
SOICP_ABLATE=128 python tools/eval_stamps.py 2>&1 | tail -28 > $O/stamps.txt
grep -E "knn sweep|life us|slowest chunk|      [0-9]" $O/stamps.txt
python tools/batch_rate.py --scans 4 2>&1 | grep "batch mode"
