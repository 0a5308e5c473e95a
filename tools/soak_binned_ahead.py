#!/usr/bin/env python3
"""Differential soak of the binning ahead (GPU box): streams of staged scans -- random scans of the trajectory, random subsets, guesses up
to 0.6 m / 6 degrees off, random sampling limits and iteration bounds, consecutive frames and arbitrary jumps (so that a scan is
binned under a pose up to the whole trajectory away from its own guess) -- registered through so_icp_stage_scan + so_icp_register
with the scans binned ahead, against the plain so_icp_register of the same scan and guess on the same context: return codes, poses,
J^T J, every per-iteration statistic and the per-query MatchingResult must agree BIT FOR BIT.  Every 4th stream also runs the
oracle on two of its scans.  usage: python tools/soak_binned_ahead.py [--seconds 60] [--seed 0]"""
import argparse, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]
import oracle_py as oracle  # noqa: E402
from superodom_amd import binding, synth  # noqa: E402

ap = argparse.ArgumentParser(); ap.add_argument("--seconds", type=float, default=60.0); ap.add_argument("--seed", type=int, default=0)
ap.add_argument("--scenes", default="tiny,small", help="comma-separated synth scenes (os1_128_2m = the configuration of record; the oracle is skipped there)")
a = ap.parse_args()
rng = np.random.default_rng(a.seed)
scenes = {name: synth.Scene(name) for name in a.scenes.split(",")}


def key(st, status):
    out = [st.n_iterations, bytes(np.array(st.JtJ)), bytes(np.array(st.Jtr))]
    for it in range(st.n_iterations):
        x = st.iterations[it]
        out += [x.lm_iterations, x.num_successful_steps, x.termination, x.num_surf_from_scan, tuple(x.reject_hist), tuple(x.obs_hist),
                np.float64(x.final_cost).tobytes(), np.float64(x.initial_cost).tobytes()]
    return out, status.tobytes()


t_end, n_reg, n_bad, n_ahead, n_streams = time.time() + a.seconds, 0, 0, 0, 0
while time.time() < t_end:
    name = str(rng.choice(list(scenes)))
    sc = scenes[name]
    max_it = int(rng.integers(1, 6))
    msf = int(rng.choice([-1, -1, 500, 2000, 4000]))
    slam = binding.LidarSlamGpu(plane_res=sc.plane_res, line_res=sc.plane_res / 2, max_surface_features=msf, max_iterations=max_it)
    slam.add_surf_point_cloud(sc.map_points)
    n_streams += 1
    K = int(rng.integers(4, 10))
    ids = [int(rng.integers(0, 32)) for _ in range(K)] if rng.random() < 0.5 else list(range(int(rng.integers(0, 20)), 32))[:K]
    scans, guesses = [], []
    for i in ids:
        s = sc.scan(i)
        if rng.random() < 0.3:
            s = s[np.sort(rng.permutation(len(s))[: int(rng.integers(300, len(s)))])]
        scans.append(slam.host_alloc_like(np.ascontiguousarray(s, dtype=np.float32)))
        guesses.append(synth.perturb_pose(sc.gt_pose(i), int(rng.integers(1 << 30)), float(rng.uniform(0.0, 0.6)), float(rng.uniform(0.0, 6.0))))
    ref = []
    for s, g in zip(scans, guesses):
        rc, pose, st = slam.register(s, g)
        ref.append((rc, pose, key(st, slam.match_status(len(s)))))
    slam.stage_scan(scans[0])
    for k in range(len(scans)):
        if k + 1 < len(scans):
            slam.stage_scan(scans[k + 1])
        rc, pose, st = slam.register(scans[k], guesses[k])
        n_reg += 1
        n_ahead += 1 if (st.flags & binding.FLAG_BINNED_AHEAD) else 0
        ok = rc == ref[k][0] and np.array_equal(pose, ref[k][1]) and key(st, slam.match_status(len(scans[k]))) == ref[k][2]
        # (a scan is binned ahead unless the registration before it needed more than 300 us to reach the point where it enqueues the
        #  copy: then the copy thread did, without binning -- counted below, not an error)
        ok = ok and bool(st.flags & binding.FLAG_STAGED_SCAN) and not (k == 0 and (st.flags & binding.FLAG_BINNED_AHEAD))
        if not ok:
            n_bad += 1
            print("MISMATCH", dict(scene=name, stream=ids, k=k, n=len(scans[k]), max_it=max_it, msf=msf, rc=(rc, ref[k][0]), flags=hex(st.flags),
                                   err=synth.pose_error(pose, ref[k][1])), flush=True)
    if n_streams % 4 == 0 and len(sc.map_points) <= 500_000:
        om = oracle.OracleMap(plane_res=sc.plane_res); om.add_surf(slam.export_map(), raw=True)
        cfg = oracle.default_config(max_iterations=max_it, max_surface_features=msf)
        for k in (0, len(scans) - 1):
            orc, opose, ost, _ = om.register(np.asarray(scans[k]), guesses[k], cfg)
            dt, dr = synth.pose_error(ref[k][1], opose)
            if not (orc == ref[k][0] and (orc != 0 or (dt < 1e-8 and dr < 1e-8))):
                n_bad += 1
                print("ORACLE MISMATCH", dict(scene=name, scan=ids[k], rc=(ref[k][0], orc), err=(dt, dr)), flush=True)
    slam.close()
print(f"soak_binned_ahead: {n_streams} streams, {n_reg} staged registrations ({n_ahead} binned ahead), {n_bad} mismatches (seed {a.seed})")
sys.exit(1 if n_bad else 0)
