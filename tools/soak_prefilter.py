#!/usr/bin/env python3
"""Differential soak of so_icp_prefilter_scan = laserMapping::adjustVoxelSize (laserMapping.cpp:598-651) on the GPU box:
random clouds (boxes, dense clusters, noisy planes, points on leaf boundaries, sensor sweeps, tiny and huge clouds, clouds far
from the origin) at random resolutions, with and without auto_voxel_size: the resolution choice and the far-point count
against their float restatement, the VoxelGrid centroids point for point (order included) against the oracle's
pcl::VoxelGrid restatement.   usage: python tools/soak_prefilter.py [--seconds 60] [--seed 0]"""
import argparse, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]
import oracle_py as oracle  # noqa: E402
from superodom_amd import binding  # noqa: E402

ap = argparse.ArgumentParser(); ap.add_argument("--seconds", type=float, default=60.0); ap.add_argument("--seed", type=int, default=0)
a = ap.parse_args()
rng = np.random.default_rng(a.seed)


def cloud():
    kind = rng.integers(0, 5); n = int(10 ** rng.uniform(0.5, 5.3))
    if kind == 0:
        p = rng.uniform(-1, 1, (n, 3)) * rng.uniform(1, 150, 3)
    elif kind == 1:
        k = rng.integers(1, 6)
        p = np.concatenate([rng.normal(0, rng.uniform(0.01, 0.5), (n // k + 1, 3)) + rng.uniform(-30, 30, 3) for _ in range(k)])
    elif kind == 2:
        p = rng.uniform(-60, 60, (n, 3)); p[:, 2] = rng.normal(0, 0.02, n) + rng.integers(-2, 3) * 3.0
    elif kind == 3:
        p = np.round(rng.uniform(-80, 80, (n, 3)) / 0.2) * 0.2 + rng.choice([0.0, 1e-7, -1e-7], (n, 3))
    else:
        m = max(n // 64, 1)
        az = np.tile(np.linspace(0, 2 * np.pi, m, endpoint=False), 64)[:n]; el = np.repeat(np.linspace(-0.6, 0.2, 64), m)[:n]
        r = np.minimum(1.5 / np.maximum(-np.sin(el), 1e-3), rng.uniform(20, 120))
        p = np.stack([r * np.cos(el) * np.cos(az), r * np.cos(el) * np.sin(az), r * np.sin(el)], 1) + rng.normal(0, 0.01, (len(az), 3))
    if rng.random() < 0.2:
        p = p + rng.uniform(-400, 400, 3)
    return p.astype(np.float32)


slam = binding.LidarSlamGpu(plane_res=0.4, line_res=0.2)
t_end, n_clouds, n_pts, n_bad = time.time() + a.seconds, 0, 0, 0
n_announced = 0
while time.time() < t_end:
    raw = cloud()
    auto = bool(rng.random() < 0.5)
    line, plane = [(0.1, 0.2), (0.2, 0.4), (0.05, 0.1), (0.4, 0.8)][int(rng.integers(0, 4))]
    # so_icp_prefilter_announce at random: for this cloud (the staged copy must be taken), for another buffer (ignored), announced and
    # withdrawn, or not at all -- the filtered cloud must not depend on it
    how = int(rng.integers(0, 4))
    if how == 0:
        slam.prefilter_announce(raw)
    elif how == 1:
        other = cloud(); slam.prefilter_announce(other)
    elif how == 2:
        slam.prefilter_announce(raw); slam.prefilter_announce(None)
    d, n, info = slam.prefilter_scan(raw, auto, line, plane)
    n_announced += int(info.reserved == 1)
    if (how == 0) != (info.reserved == 1):
        n_bad += 1; print("announcement taken / not taken against expectation:", how, info.reserved)
    got = slam.download_scan(d, n) if n else np.zeros((0, 3), np.float32)
    ab = np.abs(raw).astype(np.float32)
    avg = [np.add.accumulate(ab[:, k], dtype=np.float32)[-1] / np.float32(len(raw)) for k in range(3)]
    avg_dist = float(np.float32(avg[0]) * np.float32(avg[1]) * np.float32(avg[2]))
    far = int(np.sum((raw[:, 0] * raw[:, 0] + raw[:, 1] * raw[:, 1] + raw[:, 2] * raw[:, 2]) > np.float32(9)))
    edge = min(abs(avg_dist - 25), abs(avg_dist - 65)) < 2e-2 * max(avg_dist, 1.0)   # the statistic sits on a threshold: either side is right
    if auto:   # laserMapping.cpp:603-635: statistic, far-point count, resolution choice (between the thresholds the resolution stays)
        ok = info.count_far_points == far and abs(info.average_distance - avg_dist) <= 1e-2 * max(avg_dist, 1e-6)  # (fp64 tree sums here, in-order float sums upstream: DESIGN section 8)
        want = 0.2 if avg_dist < 25 else (0.8 if avg_dist > 65 else plane)
        ok = ok and (edge or abs(info.plane_res - want) < 1e-6)
    else:      # no statistic without auto_voxel_size
        ok = info.count_far_points == 0 and info.average_distance == 0 and abs(info.plane_res - plane) < 1e-6
    ref = oracle.voxel_grid(raw, float(info.plane_res))
    ok = ok and got.shape == ref.shape and np.array_equal(got.view(np.uint32), ref.view(np.uint32))
    n_clouds += 1; n_pts += len(raw)
    if not ok:
        n_bad += 1
        print(f"MISMATCH cloud {n_clouds} n {len(raw)} auto {auto} res {line}/{plane} -> {info.plane_res} far {info.count_far_points}/{far} avg {info.average_distance}/{avg_dist} out {got.shape}/{ref.shape}", flush=True)
print(f"soak: {n_clouds} clouds, {n_pts} points, {n_announced} filtered from an announced copy, {n_bad} mismatches (seed {a.seed})")
sys.exit(1 if n_bad else 0)
