#!/usr/bin/env python3
"""BASELINE.json configs[4] on one GPU: 64 pose hypotheses per scan (dt ~ U(-0.5, 0.5) m, dtheta ~ U(-5, 5) deg, SURVEY 8d
seeds), one so_icp_register_batch call per scan, on the batched kernels (SOICP_BATCH_MODE=one_per_cu limits the resident solve
workgroups per compute unit).
usage (GPU box): python tools/batch_rate.py [--hyp 64] [--scans 3]"""
import argparse, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from superodom_amd import binding, synth  # noqa: E402

ap = argparse.ArgumentParser(); ap.add_argument("--hyp", type=int, default=64); ap.add_argument("--scans", type=int, default=3)
a = ap.parse_args()
sc = synth.Scene("os1_128_2m")
slam = binding.LidarSlamGpu(device_id=0, plane_res=sc.plane_res, line_res=sc.plane_res / 2, max_iterations=5, lm_max_iterations=4,
                            max_surface_features=-1)
slam.add_surf_point_cloud(sc.map_points)
tot_t = 0.0; tot_ok = 0; errs = []; outer = []
for i in range(a.scans + 1):
    d, n = slam.upload_scan(sc.scan(i))
    poses = np.stack([synth.perturb_pose(sc.gt_pose(i), 5000 + 64 * i + h, 0.5, 5.0) for h in range(a.hyp)])
    t = time.perf_counter()
    ok, rcs, out, sts = slam.register_batch(None, poses, d_scan=d, n=n)
    dt = time.perf_counter() - t
    if i > 0:  # the first call allocates the per-hypothesis buffers
        tot_t += dt; tot_ok += ok
        errs += [synth.pose_error(out[h], sc.gt_pose(i))[0] for h in range(a.hyp)]
        outer += [s.n_iterations for s in sts]
    slam.free_scan(d)
errs = np.array(errs)
print("batch mode %s, wg/cu %s: %d hypotheses/scan, %.2f ms per batch, %.0f registrations/s; returned ok %d; converged to < 1 cm: %d / %d; "
      "outer iterations per hypothesis: mean %.2f, histogram %s" % (
          os.environ.get("SOICP_BATCH_MODE", "batched"), "-", a.hyp, 1e3 * tot_t / a.scans,
          a.hyp * a.scans / tot_t, tot_ok, int(np.sum(errs < 0.01)), len(errs), np.mean(outer), np.bincount(outer).tolist()))
