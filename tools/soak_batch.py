#!/usr/bin/env python3
"""Differential soak of so_icp_register_batch (GPU box): random scenes, hypothesis counts (1..150: one to three groups of the
batched kernels), perturbation sizes, sampling limits and iteration caps; every hypothesis of every batch against its single
registration -- status, pose, final normal equations, every per-iteration statistic, bit for bit.
usage: python tools/soak_batch.py [--seconds 120] [--seed 0]"""
import argparse, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
from superodom_amd import binding, synth  # noqa: E402

ap = argparse.ArgumentParser(); ap.add_argument("--seconds", type=float, default=120.0); ap.add_argument("--seed", type=int, default=0)
a = ap.parse_args()
rng = np.random.default_rng(a.seed)
scenes = {name: synth.Scene(name) for name in ("tiny", "small")}


def same(x, y):
    if x.n_iterations != y.n_iterations or not np.array_equal(np.array(x.JtJ), np.array(y.JtJ)) or not np.array_equal(np.array(x.Jtr), np.array(y.Jtr)):
        return False
    for it in range(x.n_iterations):
        p, q = x.iterations[it], y.iterations[it]
        if (p.lm_iterations, p.num_successful_steps, p.termination, p.num_surf_from_scan, p.initial_cost, p.final_cost) != \
           (q.lm_iterations, q.num_successful_steps, q.termination, q.num_surf_from_scan, q.initial_cost, q.final_cost):
            return False
        if list(p.reject_hist) != list(q.reject_hist) or list(p.obs_hist) != list(q.obs_hist) or not np.array_equal(np.array(p.pose_after), np.array(q.pose_after)):
            return False
    return True


t_end, n_batches, n_hyp_total, n_bad, outer_hist = time.time() + a.seconds, 0, 0, 0, np.zeros(8, np.int64)
while time.time() < t_end:
    name = str(rng.choice(list(scenes)))
    sc = scenes[name]
    max_it = int(rng.integers(1, 6))
    msf = int(rng.choice([-1, -1, 500, 2000, 4000]))
    slam = binding.LidarSlamGpu(plane_res=sc.plane_res, line_res=sc.plane_res / 2, max_surface_features=msf, max_iterations=max_it)
    slam.add_surf_point_cloud(sc.map_points)
    for _ in range(4):
        i = int(rng.integers(0, 32))
        scan = sc.scan(i)
        if rng.random() < 0.3:
            scan = scan[rng.permutation(len(scan))[: int(rng.integers(200, len(scan)))]]
        B = int(rng.choice([1, 2, 3, 7, 16, 31, 64, 65, 100, 150]))
        dt, dth = float(rng.uniform(0.0, 0.6)), float(rng.uniform(0.0, 6.0))
        poses = np.stack([synth.perturb_pose(sc.gt_pose(i), int(rng.integers(1 << 30)), dt * rng.random(), dth * rng.random()) for _ in range(B)])
        d, n = slam.upload_scan(scan)
        ok, rcs, out, sts = slam.register_batch(None, poses, d_scan=d, n=n)
        n_batches += 1
        for h in range(B):
            rc, ph, sh = slam.register_dev(d, n, poses[h])
            n_hyp_total += 1
            outer_hist[min(7, sts[h].n_iterations)] += 1
            if rc != int(rcs[h]) or not np.array_equal(ph, out[h]) or not same(sts[h], sh):
                n_bad += 1
                print(f"MISMATCH scene {name} scan {i} B {B} h {h} max_it {max_it} msf {msf} rc {rc}/{int(rcs[h])}", flush=True)
        slam.free_scan(d)
    slam.close()
print(f"soak: {n_batches} batches, {n_hyp_total} hypotheses against their single registrations, {n_bad} mismatches; outer iterations histogram {outer_hist.tolist()} (seed {a.seed})")
sys.exit(1 if n_bad else 0)
