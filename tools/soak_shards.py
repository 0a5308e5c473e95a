#!/usr/bin/env python3
"""Differential soak of the SHARDED path on one GPU (GPU box): N = 2..4 shard contexts joined by an in-process group (the
sums take the place of the RCCL all-reduce / peer exchange), each driven from its own thread, against ONE context holding
the whole map: Localization frame after frame (sharded device-side insert, per-cube counts by collective) with random
guess errors up to 0.6 m / 6 degrees (queries change owner between outer iterations), planeRes switches (the shards are
re-cut: a collective step) and sub-sampled scans.
Every frame: all ranks return the same bits; status, iteration counts, termination codes, both histograms equal the single
context's; poses to 1e-9; every resident centroid is a centroid of the unsharded map, the shards' union is the whole map.
usage: python tools/soak_shards.py [--seconds 120] [--seed 0]"""
import argparse, os, sys, threading, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
from superodom_amd import binding, synth  # noqa: E402

ap = argparse.ArgumentParser(); ap.add_argument("--seconds", type=float, default=120.0); ap.add_argument("--seed", type=int, default=0)
a = ap.parse_args()
rng = np.random.default_rng(a.seed)
scenes = {name: synth.Scene(name) for name in ("tiny", "small")}


def in_threads(fns):
    res = [None] * len(fns)

    def run(k):
        res[k] = fns[k]()
    th = [threading.Thread(target=run, args=(k,)) for k in range(len(fns))]
    for t in th: t.start()
    for t in th: t.join(120)
    return res


def key(arr):
    return {tuple(v) for v in arr.view(np.uint32).reshape(-1, 3).tolist()}


t_end, n_seq, n_frames, n_bad, worst = time.time() + a.seconds, 0, 0, 0, 0.0
while time.time() < t_end:
    name = str(rng.choice(list(scenes))); sc = scenes[name]
    world = int(rng.integers(2, 5)); max_it = int(rng.integers(1, 6)); msf = int(rng.choice([-1, -1, 2000]))
    res = float(sc.plane_res)
    mk = dict(plane_res=res, line_res=res / 2, max_surface_features=msf, max_iterations=max_it)
    one = binding.LidarSlamGpu(**mk)
    shards = [binding.LidarSlamGpu(rank=r, world_size=world, **mk) for r in range(world)]
    gkey = 0x100000 + n_seq
    for sh in shards: sh.comm_init_inprocess(gkey)
    i0 = int(rng.integers(0, 20)); T = sc.gt_pose(i0)
    if rng.random() < 0.5:
        one.set_origin(T[:3]); one.add_surf_point_cloud(sc.map_points)
        in_threads([lambda sh=sh: (sh.set_origin(T[:3]), sh.add_surf_point_cloud(sc.map_points)) for sh in shards])
    rc0 = one.localization(False, T, sc.scan(i0), 0.0)[0]
    rcs = in_threads([lambda sh=sh: sh.localization(False, T, sc.scan(i0), 0.0) for sh in shards])
    ok = all(r is not None and r[0] == rc0 for r in rcs)
    n_seq += 1
    for k in range(1, int(rng.integers(3, 7))):
        if not ok: break
        if rng.random() < 0.2:
            res = float(rng.choice([0.2, 0.4]))
            one.set_resolution(res / 2, res)
            in_threads([lambda sh=sh: (sh.set_resolution(res / 2, res), True)[1] for sh in shards])
        i = (i0 + k) % 32
        scan = sc.scan(i)
        if rng.random() < 0.3: scan = scan[rng.permutation(len(scan))[: int(rng.integers(500, len(scan)))]]
        guess = synth.perturb_pose(sc.gt_pose(i), int(rng.integers(1 << 30)), float(rng.uniform(0.0, 0.6)), float(rng.uniform(0.0, 6.0)))
        rc, pose, st = one.localization(True, guess, scan, 0.1 * k)
        out = in_threads([lambda sh=sh: sh.localization(True, guess, scan, 0.1 * k) for sh in shards])
        n_frames += 1
        ok = all(r is not None and r[0] == rc for r in out) and all(np.array_equal(out[0][1], r[1]) for r in out)
        if ok and rc == 0:
            for r in out:
                s2 = r[2]
                ok = ok and s2.n_iterations == st.n_iterations and s2.laser_cloud_surf_from_map_num == st.laser_cloud_surf_from_map_num
                for it in range(st.n_iterations if ok else 0):
                    x, y = s2.iterations[it], st.iterations[it]
                    ok = ok and (x.lm_iterations, x.num_successful_steps, x.termination, x.num_surf_from_scan) == (y.lm_iterations, y.num_successful_steps, y.termination, y.num_surf_from_scan)
                    ok = ok and list(x.reject_hist) == list(y.reject_hist) and list(x.obs_hist) == list(y.obs_hist)
                d = synth.pose_error(r[1], pose)
                ok = ok and d[0] < 1e-9 and d[1] < 1e-9
                worst = max(worst, d[0])
        if ok:
            full = one.export_map(); kfull = key(full); union = set()
            for sh in shards:
                total, mine = sh.map_size(this_rank=True)
                part = sh.export_map(); kp = key(part)
                ok = ok and total == len(full) and len(part) == mine and kp <= kfull
                union |= kp
            ok = ok and union == kfull
        if not ok:
            n_bad += 1
            print(f"MISMATCH scene {name} world {world} start {i0} frame {k} res {res} max_it {max_it} msf {msf} rc {rc} / {[None if r is None else r[0] for r in out]}", flush=True)
    for sh in shards: sh.close()
    one.close()
print(f"soak: {n_seq} sequences, {n_frames} sharded Localization frames (2..4 ranks), {n_bad} mismatches, worst pose difference to the single context {worst:.2e} m (seed {a.seed})")
sys.exit(1 if n_bad else 0)
