#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
SOICP_DEBUG_DEFER=1 timeout 120 python - <<'PY'
import sys, time, numpy as np
sys.path.insert(0, '.')
from superodom_amd import binding, synth
sc = synth.Scene("os1_128_2m")
slam = binding.LidarSlamGpu(device_id=0, plane_res=sc.plane_res, line_res=sc.plane_res / 2, max_iterations=5, lm_max_iterations=4, max_surface_features=-1)
slam.add_surf_point_cloud(sc.map_points)
for i in range(6):
    d = slam.upload_scan(sc.scan(i % 4)); st = binding.Stats()
    t = time.perf_counter()
    rc = slam.register_dev(d[0], d[1], sc.guess(i % 4), st)
    tm = slam.timing()
    print("scan", i, "rc", rc if not isinstance(rc, tuple) else rc[0], "ms", round(1e3 * (time.perf_counter() - t), 3), "elapsed", round(st.time_elapsed_ms, 3), flush=True)
PY
for e in 2 0 2 0; do
SOICP_KNN_DEFER=$e timeout 300 python bench.py --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 > /tmp/bench_line.json
python - /tmp/bench_line.json $e <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
