#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/g24; mkdir -p $O
cd $R
timeout 900 python -m pytest tests -q -m gpu -x -k "batch or parity or bit" 2>&1 | tail -3 | tee $O/pytest.txt
python tools/batch_rate.py 2>&1 | grep "batch mode" | tee $O/batch_rate.txt
python tools/batch_stamps.py --hyp 64 --scans 2 2>&1 | tail -6 | tee $O/stamps.txt
python bench.py --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 > $O/bench_line.json
python - $O/bench_line.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print("value", round(d["value"], 1), "entry_points", {k: round(v, 1) for k, v in d["entry_points"].items() if k != "note"}, "knn us", round(1e3 * d["roofline"]["avg_launch_ms"], 2))
print("kernels", {k: round(v, 4) for k, v in d["kernels"].items() if isinstance(v, float)})
PY
