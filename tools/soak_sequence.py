#!/usr/bin/env python3
"""Soak of so_icp_register_sequence / so_icp_sequence_announce_next (GPU box): a stream of scans worked off in calls of random length, the scan
behind every call announced correctly, wrongly or not at all, kernel timing events on or off, host or resident scans, guesses sometimes far off
(chain breaks); every registration must equal so_icp_register from the guess the run reports, bit for bit.
usage: python tools/soak_sequence.py [--calls 120] [--scene small|tiny|os1_128_2m] [--seed 1]"""
import argparse, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from superodom_amd import binding, synth  # noqa: E402

ap = argparse.ArgumentParser(); ap.add_argument("--calls", type=int, default=120); ap.add_argument("--scene", default="small"); ap.add_argument("--seed", type=int, default=1)
a = ap.parse_args()
rng = np.random.default_rng(a.seed)
sc = synth.Scene(a.scene)
S = 8
for time_kernels in (0, 1):
    mk = dict(plane_res=sc.plane_res, line_res=sc.plane_res / 2, max_surface_features=-1, max_iterations=5, time_kernels=time_kernels, device_id=0)
    slam, plain = binding.LidarSlamGpu(**mk), binding.LidarSlamGpu(**{**mk, "time_kernels": 0})
    for s in (slam, plain):
        s.add_surf_point_cloud(sc.map_points)
        s.shift_map(sc.gt_pose(0)[:3])
    scans = [slam.host_alloc_like(np.ascontiguousarray(sc.scan(i), dtype=np.float32)) for i in range(S)]
    d_scans = [slam.upload_scan(s_) for s_ in scans]
    pos = 0  # index of the next scan of the stream (cycling through the S scans)
    n_reg = n_chained = n_checked = 0
    for call in range(a.calls):
        count = int(rng.integers(1, 8))
        ids = [(pos + k) % S for k in range(count)]
        on_dev = bool(rng.integers(0, 4) == 0)
        deltas = np.zeros((count, 7)); deltas[:, 6] = 1
        for k in range(1, count):
            far = rng.integers(0, 9) == 0
            g = synth.perturb_pose(sc.gt_pose(ids[k]), int(rng.integers(1 << 30)), 0.45 if far else 0.1, 4.0 if far else 1.0)
            deltas[k] = synth.pose_between(sc.gt_pose(ids[k - 1]), g)
        pose0 = synth.perturb_pose(sc.gt_pose(ids[0]), int(rng.integers(1 << 30)), 0.1, 1.0)
        ann = int(rng.integers(0, 4))  # 0 none, 1 the right scan, 2 a wrong scan, 3 withdrawn
        nxt = (pos + count) % S
        if ann == 1:
            slam.sequence_announce_next(scans[nxt], synth.pose_between(sc.gt_pose(ids[-1]), sc.guess(nxt)))
        elif ann == 2:
            slam.sequence_announce_next(scans[(nxt + 3) % S], synth.pose_between(sc.gt_pose(ids[-1]), sc.guess((nxt + 3) % S)))
        elif ann == 3:
            slam.sequence_announce_next(None, None)
        rc, poses, guesses, stats, n_done = slam.register_sequence([d_scans[i] for i in ids] if on_dev else [scans[i] for i in ids], pose0, deltas, on_device=on_dev)
        assert rc == 0 and n_done == count, (call, rc, n_done, slam.last_error())
        for k in range(count):
            n_reg += 1; n_chained += bool(stats[k].flags & binding.FLAG_CHAINED)
            if rng.integers(0, 3) == 0 or k == 0:
                prc, ppose, pst = plain.register(scans[ids[k]], guesses[k])
                assert prc == 0 and np.array_equal(ppose, poses[k]) and pst.n_iterations == stats[k].n_iterations, (call, k, ids, on_dev, ann)
                n_checked += 1
        pos = (pos + count) % S
        if rng.integers(0, 10) == 0:  # something else in between: a plain registration, a map query
            slam.register(scans[pos], sc.guess(pos)); slam.map_size()
    t = slam.timing()
    print("time_kernels %d: %d calls, %d registrations (%d chained, %d chain breaks), %d compared with so_icp_register bit for bit: ok" % (
        time_kernels, a.calls, n_reg, n_chained, t.seq_chain_breaks, n_checked))
    slam.close(); plain.close()
