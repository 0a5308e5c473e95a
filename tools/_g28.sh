#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/g28; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | grep -E "passed|failed|error|Error|assert" | tail -5 | tee $O/pytest.txt
python tools/batch_rate.py --scans 6 2>&1 | grep "batch mode" | tee $O/batch_rate.txt
for i in 1 2; do
python bench.py --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 > $O/bench_line.json
python - $O/bench_line.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print("value", round(d["value"], 1), "knn us", round(1e3 * d["roofline"]["avg_launch_ms"], 2), "kernels", {k: round(v, 4) for k, v in d["kernels"].items() if isinstance(v, float)})
PY
done
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/bt
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/bt -- python $R/tools/batch_rate.py --scans 3 > $O/bt.log 2>&1
f=$(find /tmp/bt -name "*kernel_stats.csv" | head -1); head -8 $f | cut -c1-150
