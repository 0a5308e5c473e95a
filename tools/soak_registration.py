#!/usr/bin/env python3
"""Differential soak of the registration (GPU box): random scans, guesses, resolutions and sampling limits on the small
scenes, the HIP path against the CPU oracle -- iteration counts, LM iterations / termination codes, both histograms, the
per-query MatchingResult of the last iteration, poses to 1e-8.  usage: python tools/soak_registration.py [--seconds 120] [--seed 0] [--scenes tiny,small]"""
import argparse, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]
import oracle_py as oracle  # noqa: E402
from superodom_amd import binding, synth  # noqa: E402

ap = argparse.ArgumentParser(); ap.add_argument("--seconds", type=float, default=120.0); ap.add_argument("--seed", type=int, default=0)
ap.add_argument("--scenes", default="tiny,small", help="comma-separated synth scenes (os1_128_2m = the configuration of record: ~1 s of oracle per registration)")
a = ap.parse_args()
rng = np.random.default_rng(a.seed)
oracle.set_num_threads(max(1, (os.cpu_count() or 2) // 2))
scenes = {name: synth.Scene(name) for name in a.scenes.split(",")}
t_end, n_reg, n_bad, worst = time.time() + a.seconds, 0, 0, (0.0, 0.0)
while time.time() < t_end:
    name = str(rng.choice(list(scenes)))
    sc = scenes[name]
    max_it = int(rng.integers(1, 6))
    msf = int(rng.choice([-1, -1, 500, 2000, 4000]))
    slam = binding.LidarSlamGpu(plane_res=sc.plane_res, line_res=sc.plane_res / 2, max_surface_features=msf, max_iterations=max_it)
    slam.add_surf_point_cloud(sc.map_points)
    om = oracle.OracleMap(plane_res=sc.plane_res); om.add_surf(slam.export_map(), raw=True)
    cfg = oracle.default_config(max_iterations=max_it, max_surface_features=msf)
    for _ in range(6):
        i = int(rng.integers(0, 32))
        scan = sc.scan(i)
        if rng.random() < 0.3:
            scan = scan[rng.permutation(len(scan))[: int(rng.integers(200, len(scan)))]]
        guess = synth.perturb_pose(sc.gt_pose(i), int(rng.integers(1 << 30)), float(rng.uniform(0.0, 0.6)), float(rng.uniform(0.0, 6.0)))
        rc, pose, st = slam.register(scan, guess)
        orc, opose, ost, corrs = om.register(scan, guess, cfg, want_corrs=True)
        n_reg += 1
        ok = rc == orc and st.n_iterations == ost.n_iterations
        if ok and rc == 0:
            for it in range(st.n_iterations):
                x, y = st.iterations[it], ost.iters[it]
                ok = ok and (x.lm_iterations, x.num_successful_steps, x.termination, x.num_surf_from_scan) == (y.lm_iterations, y.num_successful_steps, y.termination, y.num_surf)
                ok = ok and list(x.reject_hist) == list(y.reject_hist) and list(x.obs_hist) == list(y.obs_hist)
            if msf < 0 or len(scan) <= msf:  # (with the sampling rule active the skipped queries carry different placeholder codes on the two sides)
                ok = ok and np.array_equal(slam.match_status(len(scan)), corrs["status"].astype(np.uint8))
            dt, dr = synth.pose_error(pose, opose)
            worst = (max(worst[0], dt), max(worst[1], dr))
            ok = ok and dt < 1e-8 and dr < 1e-8
        if not ok:
            n_bad += 1
            print("MISMATCH", dict(scene=name, scan=i, n=len(scan), max_it=max_it, msf=msf, rc=(rc, orc), outer=(st.n_iterations, ost.n_iterations),
                                   err=synth.pose_error(pose, opose)), flush=True)
    slam.close()
print(f"soak: {n_reg} registrations, {n_bad} mismatches, worst pose difference {worst[0]:.2e} m {worst[1]:.2e} rad (seed {a.seed})")
sys.exit(1 if n_bad else 0)
