#!/usr/bin/env python3
"""Phase stamps of the batched solve (SOICP_ABLATE=128, the instrumented instantiation): the controller workgroup of
hypothesis 0, its fit pass and its first plain evaluation pass of the LAST round that hypothesis took part in.
usage (GPU box): python tools/batch_stamps.py [--hyp 64] [--scans 3]"""
import argparse, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("SOICP_ABLATE", "128")
from superodom_amd import binding, synth  # noqa: E402

NAMES = ["loops(all virtual wgs)", "last lds_reduce+store", "poll records", "-", "combine", "controller+publish", "total", "controller"]
ap = argparse.ArgumentParser(); ap.add_argument("--hyp", type=int, default=64); ap.add_argument("--scans", type=int, default=3)
a = ap.parse_args()
sc = synth.Scene("os1_128_2m")
slam = binding.LidarSlamGpu(device_id=0, plane_res=sc.plane_res, line_res=sc.plane_res / 2, max_iterations=5, lm_max_iterations=4,
                            max_surface_features=-1)
slam.add_surf_point_cloud(sc.map_points)
for i in range(a.scans + 1):
    d, n = slam.upload_scan(sc.scan(i))
    poses = np.stack([synth.perturb_pose(sc.gt_pose(i), 5000 + 64 * i + h, 0.5, 5.0) for h in range(a.hyp)])
    ok, rcs, out, sts = slam.register_batch(None, poses, d_scan=d, n=n)
    s = slam.debug_stamps().astype(np.float64) * 0.01
    if i > 0:
        print("hyp", a.hyp, "scan", i, "outer iterations of hypothesis 0:", sts[0].n_iterations)
        print("  fit pass :", {k: round(v, 1) for k, v in zip(NAMES, s[0:8])})
        print("  eval pass:", {k: round(v, 1) for k, v in zip(NAMES, s[8:16])})
    slam.free_scan(d)
