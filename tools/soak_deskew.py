#!/usr/bin/env python3
"""Differential soak of so_icp_deskew_scan (GPU box): random sweeps, pose buffers (rates, spans, sign flips, exact hits of
point times on pose times, buffers ending inside the sweep), strides and extrinsics, device vs CPU restatement:
bit-identical for all but a sliver of the coordinates, the rest within one float32 spacing; clamped counts and the
sweep-start frame identical.  usage: python tools/soak_deskew.py [--seconds 60] [--seed 0]"""
import argparse, os, sys, time
import numpy as np
from scipy.spatial.transform import Rotation as R
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import deskew_data as dd  # noqa: E402
import oracle_py as oracle  # noqa: E402
from superodom_amd import binding  # noqa: E402

ap = argparse.ArgumentParser(); ap.add_argument("--seconds", type=float, default=60.0); ap.add_argument("--seed", type=int, default=0)
a = ap.parse_args()
rng = np.random.default_rng(a.seed)
slam = binding.LidarSlamGpu()
t_end, runs, worst_frac, worst_ulp = time.time() + a.seconds, 0, 1.0, 0.0
while time.time() < t_end:
    t0 = float(rng.choice([0.0, 12.5, 1.7e9 + rng.uniform(0, 1)]))
    stride, toff = [(32, 20), (16, 12), (48, 40), (32, 12)][int(rng.integers(0, 4))]
    n = int(10 ** rng.uniform(0, 5.2))
    rec = dd.sweep(n, stride, toff, sweep_s=float(rng.uniform(0.01, 0.3)), seed=int(rng.integers(1 << 30)), nan_every=int(rng.choice([0, 0, 97])))
    imu = bool(rng.integers(0, 2))
    poses = dd.pose_buffer(t0, rate_hz=float(rng.choice([10.0, 50.0, 200.0, 1000.0, 6000.0])), before_s=float(rng.uniform(0.0, 0.05)),
                           after_s=float(rng.uniform(0.005, 0.4)), seed=int(rng.integers(1 << 30)), translate=not imu, flip_signs=bool(rng.integers(0, 2)))
    if rng.random() < 0.3 and len(poses) > 3:  # some point times hit pose times exactly
        k = rng.integers(0, len(poses), min(len(rec), 50))
        tt = (poses[k, 0] - t0).astype(np.float32)
        raw = rec.view(np.float32).reshape(len(rec), stride // 4); raw[:len(k), toff // 4] = tt; rec = raw.view(np.uint8).reshape(len(rec), stride)
    til = None if rng.random() < 0.4 else np.concatenate([rng.normal(0, 0.1, 3), R.from_rotvec(rng.normal(0, 0.5, 3)).as_quat()])
    want, wstart, wbeyond = oracle.deskew(rec, toff, t0, poses, imu, til)
    got, info = slam.deskew_scan(rec, toff, t0, poses, imu, til)
    assert info.n_clamped == wbeyond, (runs, info.n_clamped, wbeyond)
    assert list(info.t_w_original_l) + list(info.q_w_original_l) == list(wstart), runs
    keep = np.ones(stride, bool); keep[:12] = False
    assert np.array_equal(got[:, keep], rec[:, keep]), runs
    x, y = dd.xyz_of(got).reshape(-1), dd.xyz_of(want).reshape(-1)
    same = (x.view(np.uint32) == y.view(np.uint32)) | (np.isnan(x) & np.isnan(y))
    d = np.abs(x.astype(np.float64) - y.astype(np.float64))[~same]
    ulp = np.spacing(np.maximum(np.abs(x), np.abs(y)).astype(np.float32)).astype(np.float64)[~same]
    frac = same.mean() if len(same) else 1.0
    wu = float((d / ulp).max()) if len(d) else 0.0
    assert wu <= 1.0 and (frac > 0.995 or len(same) < 3000), (runs, frac, wu, n)
    worst_frac, worst_ulp = min(worst_frac, frac if len(same) >= 3000 else 1.0), max(worst_ulp, wu)
    runs += 1
print(f"soak ok: {runs} sweeps, smallest bit-identical fraction {worst_frac:.5f}, largest difference {worst_ulp:.1f} float32 spacings (seed {a.seed})")
