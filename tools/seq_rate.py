#!/usr/bin/env python3
"""so_icp_register_sequence on the configuration of record (131 072-point scans vs the 2 M-point map), or on the stock operating point
(--stock: pre-filtered resident clouds, max_surface_features 2000): registrations per second of one call over --count scans, beside the loop of
single calls.  usage (GPU box): python tools/seq_rate.py [--count 48] [--stock] [--reps 3]; under rocprofv3: tools/seq_timeline.sh"""
import argparse, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from superodom_amd import binding, synth  # noqa: E402

ap = argparse.ArgumentParser(); ap.add_argument("--count", type=int, default=48); ap.add_argument("--stock", action="store_true")
ap.add_argument("--reps", type=int, default=3); ap.add_argument("--no-single", action="store_true", help="skip the loop of single calls (kernel timelines of the sequence alone)")
a = ap.parse_args()
sc = synth.Scene("os1_128_2m")
S = 4
max_feat = 2000 if a.stock else -1
cx = binding.LidarSlamGpu(rank=0, world_size=1, time_kernels=0, device_id=0, plane_res=sc.plane_res, line_res=sc.plane_res / 2,
                          max_iterations=5, lm_max_iterations=4, max_surface_features=max_feat)
cx.add_surf_point_cloud(sc.map_points)
cx.shift_map(sc.gt_pose(0)[:3])
raw = [np.ascontiguousarray(sc.scan(i), dtype=np.float32) for i in range(S)]
guesses = [np.ascontiguousarray(sc.guess(i), dtype=np.float64) for i in range(S)]
if a.stock:
    filt = []
    for s_ in raw:
        d_f, n_f, _ = cx.prefilter_scan(s_, False, sc.plane_res / 2, sc.plane_res)
        filt.append(cx.download_scan(d_f, n_f))
    scans = [cx.upload_scan(f) for f in filt]  # (device pointer, n)
else:
    scans = [cx.host_alloc_like(s_) for s_ in raw]
K = a.count
seq = [scans[k % S] for k in range(K)]
d = np.zeros((K, 7)); d[:, 6] = 1.0
for k in range(1, K):
    d[k] = synth.pose_between(sc.gt_pose((k - 1) % S), guesses[k % S])
for rep in range(a.reps):
    call, out, g, st, n_done, keep = cx.prepare_register_sequence(seq, guesses[0], d, on_device=a.stock)
    cx.synchronize()
    t0 = time.perf_counter()
    rc = call()
    cx.synchronize()
    t = (time.perf_counter() - t0) / K
    assert rc == 0 and n_done.value == K, (rc, n_done.value, cx.last_error())
    tm = cx.timing()
    print("[sequence%s] %d registrations in one call: %.4f ms each (%.0f /s), chained %d, outer %.2f, chain breaks so far %d" % (
        " stock" if a.stock else "", K, 1e3 * t, 1 / t, sum(1 for s_ in st if s_.flags & binding.FLAG_CHAINED), sum(s_.n_iterations for s_ in st) / K, tm.seq_chain_breaks))
# the loop of single calls on the same scans (resident / staged entry of the bench)
stk = [binding.Stats() for _ in range(K)]; pk = [np.zeros(7) for _ in range(K)]
if a.stock and not a.no_single:
    calls = [cx.prepare_register_dev(seq[k][0], seq[k][1], guesses[k % S], stk[k], pk[k]) for k in range(K)]
    for rep in range(a.reps):
        cx.synchronize(); t0 = time.perf_counter()
        for k in range(K):
            assert calls[k]() == 0
        cx.synchronize(); t = (time.perf_counter() - t0) / K
        print("[single calls stock] %.4f ms each (%.0f /s)" % (1e3 * t, 1 / t))
cx.close()
