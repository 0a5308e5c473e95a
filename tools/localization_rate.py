#!/usr/bin/env python3
"""Localization() end to end at BASELINE sizes (registration + device-side map insert): wall time per call.
usage (GPU box): python tools/localization_rate.py [--calls 64]"""
import argparse, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from superodom_amd import binding, synth  # noqa: E402

ap = argparse.ArgumentParser(); ap.add_argument("--calls", type=int, default=64); a = ap.parse_args()
sc = synth.Scene("os1_128_2m")
slam = binding.LidarSlamGpu(device_id=0, plane_res=sc.plane_res, line_res=sc.plane_res / 2, max_iterations=5, lm_max_iterations=4,
                            max_surface_features=-1)
slam.add_surf_point_cloud(sc.map_points)
slam.shift_map(sc.gt_pose(0)[:3])
scans = [sc.scan(i % 4) for i in range(4)]; guesses = [sc.guess(i % 4) for i in range(4)]
times = []
for k in range(a.calls + 2):
    i = k % 4
    t = time.perf_counter()
    rc, pose, st = slam.localization(True, guesses[i], scans[i], 0.1 * k)
    times.append(time.perf_counter() - t)
    assert rc == 0
t = np.array(times[2:]) * 1e3
print("Localization() ms per call: mean %.3f median %.3f min %.3f max %.3f | registration part (time_elapsed_ms of the last call) %.3f | map size %d" % (
    t.mean(), float(np.median(t)), t.min(), t.max(), st.time_elapsed_ms, slam.map_size()))
