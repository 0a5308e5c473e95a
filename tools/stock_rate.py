#!/usr/bin/env python3
"""The stock operating point on its own (config/os1_128.yaml:26-28: max_surface_features 2000, 5 iterations; node order: the raw sweep is
voxel-filtered by so_icp_prefilter_scan, the registration's sampling rule keeps ~2000 of the ~13 k points): back-to-back
so_icp_register_dev calls on the resident filtered clouds.  usage (GPU box): python tools/stock_rate.py [--calls 96] [--case os1_128|livox]
Under rocprofv3 (tools/stock_timeline.sh) the kernel trace of the last registrations is the timeline of the path."""
import argparse, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from superodom_amd import binding, synth  # noqa: E402

ap = argparse.ArgumentParser(); ap.add_argument("--calls", type=int, default=96); ap.add_argument("--case", default="os1_128")
ap.add_argument("--reps", type=int, default=3)
a = ap.parse_args()
if a.case == "os1_128":
    sc = synth.Scene("os1_128_2m"); max_feat = 2000
else:
    sc = synth.Scene("mid360_like"); max_feat = 4000
scans = [np.ascontiguousarray(sc.scan(i), dtype=np.float32) for i in range(4)]
gs = [np.ascontiguousarray(sc.guess(i), dtype=np.float64) for i in range(4)]
cx = binding.LidarSlamGpu(rank=0, world_size=1, time_kernels=0, device_id=0, plane_res=sc.plane_res, line_res=sc.plane_res / 2,
                          max_iterations=5, lm_max_iterations=4, max_surface_features=max_feat)
cx.add_surf_point_cloud(sc.map_points)
cx.shift_map(sc.gt_pose(0)[:3])
filt = []
for s_ in scans:
    d_f, n_f, _ = cx.prefilter_scan(s_, False, sc.plane_res / 2, sc.plane_res)
    filt.append(cx.download_scan(d_f, n_f))
d_filt = [cx.upload_scan(f) for f in filt]
K = a.calls
st_k = [binding.Stats() for _ in range(K)]
po_k = [np.zeros(7) for _ in range(K)]
calls = [cx.prepare_register_dev(d_filt[k % 4][0], d_filt[k % 4][1], gs[k % 4], st_k[k], po_k[k]) for k in range(K)]
for rep in range(a.reps):
    for k in range(8):
        calls[k]()
    cx.synchronize()
    t0 = time.perf_counter()
    for k in range(K):
        assert calls[k]() == 0
    cx.synchronize()
    t = (time.perf_counter() - t0) / K
    print("[%s] %d filtered points, %d sampled queries, flags %#x: %.4f ms per registration (%.0f /s), outer %.2f lm %.2f" % (
        a.case, len(filt[0]), int(sum(st_k[0].iterations[0].reject_hist)), st_k[0].flags, 1e3 * t, 1 / t,
        sum(s_.n_iterations for s_ in st_k) / K, sum(s_.iterations[i].lm_iterations for s_ in st_k for i in range(s_.n_iterations)) / K))
cx.close()
