#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/g31; mkdir -p $O
cd $R
timeout 900 python -m pytest tests -q -m gpu -x -k "parity or bit or knn or configs" 2>&1 | grep -E "passed|failed|error|Error|assert" | tail -5 | tee $O/pytest.txt
for e in 2 0 2 0 1 4; do
SOICP_KNN_DEFER=$e python bench.py --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 > $O/bench_line.json
python - $O/bench_line.json $e <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print("defer", sys.argv[2], "value", round(d["value"], 1), "knn us", round(1e3 * d["roofline"]["avg_launch_ms"], 2), "kernels", {k: round(v, 4) for k, v in d["kernels"].items() if isinstance(v, float) and 'launches' not in k})
PY
done
python - <<'PY'
import sys, numpy as np
sys.path.insert(0, '.')
from superodom_amd import binding, synth
sc = synth.Scene("os1_128_2m")
slam = binding.LidarSlamGpu(device_id=0, plane_res=sc.plane_res, line_res=sc.plane_res / 2, max_iterations=5, lm_max_iterations=4, max_surface_features=-1)
slam.add_surf_point_cloud(sc.map_points)
for i in range(4):
    d = slam.upload_scan(sc.scan(i)); st = binding.Stats()
    t0 = slam.timing().knn_deferred_queries
    slam.register_dev(d[0], d[1], sc.guess(i), st)
    print("scan", i, "deferred queries", slam.timing().knn_deferred_queries - t0, "outer", st.n_iterations)
PY
SOICP_ABLATE=128 python tools/eval_stamps.py 2>&1 | tail -28 > $O/stamps.txt
grep -E "knn sweep|life us|slowest chunk|      [0-9]" $O/stamps.txt
