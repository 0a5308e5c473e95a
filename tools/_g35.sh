#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/kt
cat > /tmp/snip.py <<'PY'
import sys, time, numpy as np
sys.path.insert(0, '/root/repo')
from superodom_amd import binding, synth
sc = synth.Scene("os1_128_2m")
slam = binding.LidarSlamGpu(device_id=0, plane_res=sc.plane_res, line_res=sc.plane_res / 2, max_iterations=5, lm_max_iterations=4, max_surface_features=-1)
slam.add_surf_point_cloud(sc.map_points)
for i in range(6):
    d = slam.upload_scan(sc.scan(i % 4)); st = binding.Stats()
    t = time.perf_counter()
    rc = slam.register_dev(d[0], d[1], sc.guess(i % 4), st)
    print("scan", i, "rc", rc if not isinstance(rc, tuple) else rc[0], "ms", round(1e3 * (time.perf_counter() - t), 3), "elapsed", round(st.time_elapsed_ms, 3), flush=True)
PY
rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -- python /tmp/snip.py 2>&1 | grep "^scan"
f=$(find /tmp/kt -name "*kernel_trace.csv" | head -1)
python - $f <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
prev = None
for r in rows:
    n = r["Kernel_Name"]
    if "knn_plane" in n or "solve_kernel" in n or "scan_keys" in n:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        print(f"gap {((s - prev) / 1e3) if prev else 0:10.1f} dur {(e - s) / 1e3:10.1f} us {n[13:40]}")
        prev = e
PY
