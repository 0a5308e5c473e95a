#!/usr/bin/env python3
"""one registration through the BATCH instantiation of the solve (so_icp_register_batch with ONE
hypothesis: 256 VGPRs, two workgroups per compute unit, host-synchronised rounds) against so_icp_register_dev (the
persistent single-registration solve).  usage (GPU box): python tools/batch1_rate.py [--reps 40]"""
import argparse, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from superodom_amd import binding, synth  # noqa: E402

ap = argparse.ArgumentParser(); ap.add_argument("--reps", type=int, default=40)
a = ap.parse_args()
sc = synth.Scene("os1_128_2m")
slam = binding.LidarSlamGpu(device_id=0, plane_res=sc.plane_res, line_res=sc.plane_res / 2, max_iterations=5, lm_max_iterations=4,
                            max_surface_features=-1)
slam.add_surf_point_cloud(sc.map_points)
scans = [slam.upload_scan(sc.scan(i)) for i in range(4)]
guesses = [np.ascontiguousarray(sc.guess(i), dtype=np.float64) for i in range(4)]
st = binding.Stats()
for mode in ("register_dev", "register_batch(1)"):
    poses = []
    for r in range(a.reps + 8):
        if r == 8:
            slam.synchronize(); t0 = time.perf_counter()
        i = r % 4
        if mode == "register_dev":
            rc, pose, _ = slam.register_dev(scans[i][0], scans[i][1], guesses[i], st)
        else:
            ok, rcs, out, sts = slam.register_batch(None, guesses[i][None, :], d_scan=scans[i][0], n=scans[i][1])
            pose = out[0]
        if r < 4:
            poses.append(np.array(pose))
    slam.synchronize()
    dt = (time.perf_counter() - t0) / a.reps
    print("%-18s %.4f ms per registration (%.0f /s); pose[0] %s" % (mode, 1e3 * dt, 1.0 / dt, np.array2string(poses[0], precision=12)))
