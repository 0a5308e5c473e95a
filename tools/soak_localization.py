#!/usr/bin/env python3
"""Differential soak of LidarSLAM::Localization as a SEQUENCE (GPU box): frames of a synthetic trajectory through
so_icp_localization (window roll + registration + device-side map insert) against the CPU oracle doing the same steps
(LocalMap::shiftMap, the registration, transformAndAddToMap) -- with random planeRes switches between frames (what
auto_voxel_size does, laserMapping.cpp:604-649), sampling limits and iteration caps.  After EVERY frame: status, iteration
counts, LM iterations / termination codes, both histograms and the pose (1e-8) of the registration, and the whole map
bit for bit.  The oracle inserts the frame with the PRODUCT's pose (the two poses differ in their last bits, which would
otherwise let the maps drift apart by float spacings and end the comparison).
usage: python tools/soak_localization.py [--seconds 120] [--seed 0]"""
import argparse, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]
import oracle_py as oracle  # noqa: E402
from superodom_amd import binding, synth  # noqa: E402

ap = argparse.ArgumentParser(); ap.add_argument("--seconds", type=float, default=120.0); ap.add_argument("--seed", type=int, default=0)
a = ap.parse_args()
rng = np.random.default_rng(a.seed)
oracle.set_num_threads(max(1, (os.cpu_count() or 2) // 2))
scenes = {name: synth.Scene(name) for name in ("tiny", "small")}
t_end, n_seq, n_frames, n_bad, worst, n_res, n_ties = time.time() + a.seconds, 0, 0, 0, [0.0, 0.0], 0, 0


def same_map(slam, om):
    x, y = slam.export_map(), om.export()
    return x.shape == y.shape and np.array_equal(x[np.lexsort(x.T)].view(np.uint32), y[np.lexsort(y.T)].view(np.uint32))


while time.time() < t_end:
    name = str(rng.choice(list(scenes)))
    sc = scenes[name]
    max_it = int(rng.integers(1, 6)); msf = int(rng.choice([-1, -1, 1000, 4000]))
    res = float(sc.plane_res)
    slam = binding.LidarSlamGpu(plane_res=res, line_res=res / 2, max_surface_features=msf, max_iterations=max_it)
    om = oracle.OracleMap(plane_res=res, line_res=res / 2)
    cfg = oracle.default_config(max_iterations=max_it, max_surface_features=msf)
    i0 = int(rng.integers(0, 20))
    T = sc.gt_pose(i0)
    if rng.random() < 0.5:  # start from the scene's map (loaded around the first pose, like localization_mode: laserMapping.cpp:161-171),
        slam.set_origin(T[:3]); om.set_origin(T[:3])  # else from the first scan alone (Localization(initialization = false))
        slam.add_surf_point_cloud(sc.map_points); om.add_surf(sc.map_points)
    rc, _, _ = slam.localization(False, T, sc.scan(i0), 0.0)
    om.set_origin(T[:3])                                                      # LidarSlam.cpp:87 (also over a loaded map)
    om.transform_and_add(sc.scan(i0), T)
    ok = rc == 2 and same_map(slam, om)
    if not ok:
        n_bad += 1
        print(f"MISMATCH at the seeding frame: scene {name} start {i0} rc {rc} map {slam.map_size()}/{om.size()}", flush=True)
    prev_hist = None
    n_seq += 1
    for k in range(1, int(rng.integers(3, 9))):
        if not ok:
            break
        i = i0 + k
        if rng.random() < 0.25:  # planeRes switch between two frames
            res = float(rng.choice([0.1, 0.2, 0.4])) if name == "tiny" else float(rng.choice([0.2, 0.4]))
            slam.set_resolution(res / 2, res); om.set_resolution(res / 2, res); n_res += 1
        scan = sc.scan(i % 32)
        if rng.random() < 0.3:
            scan = scan[rng.permutation(len(scan))[: int(rng.integers(500, len(scan)))]]
        guess = synth.perturb_pose(sc.gt_pose(i % 32), int(rng.integers(1 << 30)), float(rng.uniform(0.0, 0.3)), float(rng.uniform(0.0, 3.0)))
        map_before = slam.export_map()
        rc, pose, st = slam.localization(True, guess, scan, 0.1 * k)
        om.shift(guess[:3])                                                   # LidarSlam.cpp:363
        orc, opose, ost, _ = om.register(scan, guess, cfg, prev_obs_hist=prev_hist)
        n_frames += 1

        def same_registration(ost, opose):
            good = st.n_iterations == ost.n_iterations
            for it in range(st.n_iterations if good else 0):
                x, y = st.iterations[it], ost.iters[it]
                good = good and (x.lm_iterations, x.num_successful_steps, x.termination, x.num_surf_from_scan) == (y.lm_iterations, y.num_successful_steps, y.termination, y.num_surf)
                good = good and list(x.reject_hist) == list(y.reject_hist) and list(x.obs_hist) == list(y.obs_hist)
            d = synth.pose_error(pose, opose)
            return good and d[0] < 1e-8 and d[1] < 1e-8 and np.allclose(list(st.uncertainty), list(ost.uncertainty), atol=1e-12), d

        ok = rc == orc
        if ok and rc == 0:
            ok, d = same_registration(ost, opose)
            if not ok:
                # Two map points at exactly the same distance from a query: which one is the 5th neighbour depends on the
                # storage order (nanoflann.hpp:124 keeps the earlier one), and this oracle map -- built by its own inserts --
                # stores in VoxelGrid order, the product in (cell, leaf) order.  Repeat with an oracle holding the product's
                # map in the product's order (what the parity tests do): equal then => a tie, not a difference.
                om2 = oracle.OracleMap(plane_res=res, line_res=res / 2); om2.set_origin(T[:3]); om2.shift(guess[:3])
                om2.add_surf(map_before, raw=True)
                orc2, opose2, ost2, _ = om2.register(scan, guess, cfg, prev_obs_hist=prev_hist)
                ok, d = same_registration(ost2, opose2) if orc2 == rc else (False, d)
                if ok:
                    n_ties += 1; ost, opose = ost2, opose2
            worst = [max(worst[0], d[0]), max(worst[1], d[1])] if ok else worst
            if ok:
                prev_hist = np.array(ost.iters[ost.n_iterations - 1].obs_hist, np.int32)
                om.transform_and_add(scan, pose)                              # LidarSlam.cpp:60-80 with the product's pose
        ok = ok and same_map(slam, om)
        if not ok:
            n_bad += 1
            print(f"MISMATCH scene {name} start {i0} frame {k} res {res} max_it {max_it} msf {msf} rc {rc}/{orc} map {slam.map_size()}/{om.size()}", flush=True)
            if rc == orc == 0:
                print("   outer", st.n_iterations, ost.n_iterations, "pose delta", synth.pose_error(pose, opose),
                      [(st.iterations[q].lm_iterations, st.iterations[q].termination, st.iterations[q].num_surf_from_scan) for q in range(st.n_iterations)],
                      [(ost.iters[q].lm_iterations, ost.iters[q].termination, ost.iters[q].num_surf) for q in range(ost.n_iterations)], flush=True)
            os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
            np.savez_compressed(os.path.join(ROOT, "gpurun_out", f"soak_loc_fail_{a.seed}_{n_seq}_{k}.npz"), map_before=map_before, scan=scan, guess=guess,
                                res=res, max_it=max_it, msf=msf, prev_hist=np.zeros(0) if prev_hist is None else prev_hist, pose=pose, opose=opose)
    slam.close()
print(f"soak: {n_seq} sequences, {n_frames} Localization frames ({n_res} planeRes switches), {n_bad} mismatches ({n_ties} frames needed the oracle in the product's storage order: equidistant neighbours), worst pose difference {worst[0]:.2e} m {worst[1]:.2e} rad (seed {a.seed})")
sys.exit(1 if n_bad else 0)
