#!/usr/bin/env python3
"""Statistics of the packed light chunks of the k-NN sweep (PROF instantiation): SOICP_ABLATE=65536 python tools/knn_pack_stats.py"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("SOICP_ABLATE", "65536")
from superodom_amd import binding, synth  # noqa: E402
sc = synth.Scene("os1_128_2m")
slam = binding.LidarSlamGpu(device_id=0, plane_res=sc.plane_res, line_res=sc.plane_res / 2, max_iterations=5, lm_max_iterations=4,
                            max_surface_features=-1, time_kernels=2)
slam.add_surf_point_cloud(sc.map_points)
st = binding.Stats()
for i in range(4):
    d = slam.upload_scan(sc.scan(i))
    slam.reset_timing()
    slam.register_dev(d[0], d[1], sc.guess(i), st)
    slam.synchronize()
    t = slam.timing()
    print("scan %d: knn launches %d, %.2f us each | group passes %d, fallback lanes %d, candidates %d | packed rows %d, too many x-runs %d, tile full %d, kept per row %.1f" % (
        i, t.knn_launches, 1e3 * t.knn_ms_total / max(t.knn_launches, 1), t.knn_group_passes, t.knn_fallback_lanes, t.knn_candidates_scanned,
        t.knn_packed_rows, t.knn_packed_rows_too_many_runs, t.knn_packed_rows_tile_full, t.knn_packed_kept / max(t.knn_packed_rows, 1)))
