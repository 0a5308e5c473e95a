#!/usr/bin/env python3
"""Registration rate when the boundary hands over HOST buffers (so_icp_register: one H2D copy of the scan per call) against
the resident-scan rate of bench.py (so_icp_register_dev)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from superodom_amd import binding, synth  # noqa: E402
sc = synth.Scene("os1_128_2m")
slam = binding.LidarSlamGpu(device_id=0, plane_res=sc.plane_res, line_res=sc.plane_res / 2, max_iterations=5, lm_max_iterations=4,
                            max_surface_features=-1)
slam.add_surf_point_cloud(sc.map_points)
scans = [np.ascontiguousarray(sc.scan(i)) for i in range(4)]; guesses = [sc.guess(i) for i in range(4)]
d = [slam.upload_scan(s) for s in scans]
for mode in ("host", "resident"):
    for k in range(4):
        (slam.register(scans[k], guesses[k]) if mode == "host" else slam.register_dev(d[k][0], d[k][1], guesses[k]))
    t = time.perf_counter()
    for k in range(48):
        i = k % 4
        (slam.register(scans[i], guesses[i]) if mode == "host" else slam.register_dev(d[i][0], d[i][1], guesses[i]))
    dt = (time.perf_counter() - t) / 48
    print(f"{mode:9s}: {1e3 * dt:.3f} ms per registration, {1 / dt:.0f} registrations/s")
