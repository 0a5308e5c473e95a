#!/usr/bin/env python3
"""bench.py -- ICP registrations/sec on the BASELINE.json workload.

    python bench.py --gpus 1 --steps K --warmup W            (default: N=1)
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A "step" is ONE scan-to-map registration (so_icp_register_dev: spatial sort, then per outer iteration the
k-NN + plane-fit kernel and the fused LM evaluations, <=5 outer x <=4 LM iterations) of a synthetic
OS1-128 scan (131 072 points, already resident in HBM) against the 2M-point local map
(BASELINE.json configs[2]).  N > 1: one process per GPU; the map is sharded by brick-hash of the voxel
grid, every rank registers the SAME scan over its shard and the 45 fp64 normal-equation scalars are
all-reduced over RCCL once per evaluation (configs[3]) -> total work is fixed: "scaling": "strong".
Rank 0 prints ONE JSON line with the roofline of the dominant (k-NN) kernel, measured with HIP events on
the library's stream inside the timed region, and the CPU baseline (the oracle restating the reference
path, timed on this host's cores on a bounded sample)."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=24)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--workload", default="os1_128_2m")
    ap.add_argument("--scans", type=int, default=4, help="distinct synthetic scans cycled through the steps")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--max-outer", type=int, default=5, help="LocalizationICPMaxIter (5 = config of record)")
    ap.add_argument("--time-all-kernels", action="store_true", help="HIP events around every kernel (adds bubbles)")
    ap.add_argument("--cpu-sample", type=int, default=1, help="registrations timed for the CPU baseline")
    ap.add_argument("--shuffle-scan", action="store_true", help="experiment: random point order inside every scan (worst case for the binning atomics)")
    ap.add_argument("--no-kernel-events", action="store_true", help="experiment: no HIP events around the k-NN launches (no roofline)")
    ap.add_argument("--no-profile-pass", action="store_true", help="skip the kernel-split pass after the timed region (runs under rocprofv3 use it: one registration = one set of launches)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N > 1")
        args.gpus = world

    from superodom_amd import binding, synth

    dist = None
    if world > 1:  # control plane only: rendezvous, barriers, max-over-ranks; the data plane is RCCL inside libsoicp
        import torch.distributed as dist_mod
        dist = dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="gloo", rank=rank, world_size=world)

    # ---------------- synthetic workload (seeded) ----------------
    sc = synth.Scene(args.workload)
    max_outer, lm_iters = args.max_outer, 4
    slam = binding.LidarSlamGpu(device_id=int(os.environ.get("SOICP_BENCH_DEVICE", local_rank)), rank=rank, world_size=world, plane_res=sc.plane_res,
                                line_res=sc.plane_res / 2, max_iterations=max_outer, lm_max_iterations=lm_iters,
                                max_surface_features=-1, time_kernels=2 if args.time_all_kernels else (0 if args.no_kernel_events else 1))
    if world > 1:
        uid = [binding.comm_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(uid, src=0)
        slam.comm_init(uid[0])
    n_map = slam.add_surf_point_cloud(sc.map_points)
    scans = [sc.scan(i) for i in range(args.scans)]
    if args.shuffle_scan:
        scans = [s_[np.random.default_rng(77 + i).permutation(len(s_))] for i, s_ in enumerate(scans)]
    guesses = [sc.guess(i) for i in range(args.scans)]
    d_scans = [slam.upload_scan(s) for s in scans]  # inputs resident in HBM before the timed region
    Q = len(scans[0])
    map_total, map_rank = slam.map_size(this_rank=True)

    def barrier():
        slam.synchronize()
        if dist is not None:
            dist.barrier()

    st = binding.Stats()
    poses = []
    for w in range(args.warmup):
        i = w % args.scans
        slam.register_dev(d_scans[i][0], d_scans[i][1], guesses[i], st)
    # the timed loop only calls the C ABI: results land in preallocated structures and are examined afterwards; the
    # ctypes arguments of every call are built before the clock starts
    step_stats = [binding.Stats() for _ in range(args.steps)]
    step_pose = [np.zeros(7) for _ in range(args.steps)]
    g64 = [np.ascontiguousarray(g, dtype=np.float64) for g in guesses]
    rcs = [0] * args.steps
    calls = [slam.prepare_register_dev(d_scans[k % args.scans][0], d_scans[k % args.scans][1], g64[k % args.scans], step_stats[k], step_pose[k])
             for k in range(args.steps)]
    slam.reset_timing()
    barrier()
    t0 = time.perf_counter()
    for k in range(args.steps):
        rcs[k] = calls[k]()
    slam.synchronize()
    t_local = time.perf_counter() - t0
    iters_outer = iters_lm = accepted = 0
    for k in range(args.steps):
        assert rcs[k] == 0, rcs[k]
        st = step_stats[k]
        iters_outer += st.n_iterations
        for it in range(st.n_iterations):
            iters_lm += st.iterations[it].lm_iterations
        accepted += st.iterations[max(st.n_iterations - 1, 0)].num_surf_from_scan
        if k < args.scans:
            poses.append(step_pose[k])
    if dist is not None:
        dist.barrier()
        import torch
        tt = torch.tensor([t_local], dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        t_max = float(tt.item())
    else:
        t_max = t_local
    tm = slam.timing()
    # ---- kernel split of a registration: a profiling pass AFTER the timed region (every launch bracketed by events,
    #      which would cost ~25 us per registration inside it); every rank runs it (the collectives need all of them)
    prof = None
    if not args.no_kernel_events and not args.no_profile_pass:
        slam.set_time_kernels(2)
        slam.reset_timing()
        n_prof = 2 * args.scans
        for k in range(n_prof):
            slam.register_dev(d_scans[k % args.scans][0], d_scans[k % args.scans][1], g64[k % args.scans], st)
        slam.synchronize()
        tp = slam.timing()
        prof = {"registrations": n_prof, "knn_ms": tp.knn_ms_total / n_prof, "solve_ms": tp.eval_ms_total / n_prof,
                "binning_ms": tp.prep_ms_total / n_prof, "knn_launches": tp.knn_launches / n_prof, "solve_launches": tp.eval_launches / n_prof,
                "host_ms": tp.host_ms_total / n_prof}

    if rank != 0:
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return

    value = args.steps / t_max
    # ---- roofline of the dominant kernel (k-NN + plane fit): algorithmic bytes per launch
    #      B_knn = 12*Q (query xyz in) + 12*M_t (map xyz in) + 24*Q (n,d,w,status record out)
    #      M_t = map points in the 50 m cubes the scan touches, NOT the whole map
    knn_ms = tm.knn_ms_total / max(tm.knn_launches, 1)
    q_per_launch = tm.knn_queries / max(tm.knn_launches, 1)
    map_xyz = slam.export_map() if world == 1 else sc.map_points
    map_cube = np.floor((map_xyz.astype(np.float64) + 25.0) / 50.0).astype(np.int64)
    cube_key = lambda c: (c[:, 0] + 64) * 16384 + (c[:, 1] + 64) * 128 + (c[:, 2] + 64)
    mkeys = cube_key(map_cube)
    m_t = []
    for i in range(args.scans):
        R = synth.quat_to_R(guesses[i][3:])
        w = scans[i].astype(np.float64) @ R.T + guesses[i][:3]
        touched = np.unique(cube_key(np.floor((w + 25.0) / 50.0).astype(np.int64)))
        m_t.append(int(np.isin(mkeys, touched).sum()))
    m_per_launch = float(np.mean(m_t)) / world
    b_knn = 12.0 * q_per_launch + 12.0 * m_per_launch + 24.0 * q_per_launch
    b_knn_whole_map = 36.0 * q_per_launch + 12.0 * tm.knn_map_points / max(tm.knn_launches, 1)
    achieved = b_knn / (knn_ms * 1e-3) / 1e9 if knn_ms > 0 else 0.0
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "knn_traffic.json")  # PMC pass result (bytes per launch), see profiles/README.md
    if os.path.exists(tpath):
        try:
            traffic = json.load(open(tpath)).get("hbm_bytes_per_launch")
        except Exception:
            traffic = None
    ms_per_step_for_split = 1e3 * t_max / args.steps

    errs = [synth.pose_error(poses[i], sc.gt_pose(i)) for i in range(len(poses))]
    out = {
        "metric": "icp_registrations_per_sec", "value": value, "unit": "registrations/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * t_max / args.steps,
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"{args.workload}: OS1-128 synthetic scan ({Q} pts, resident in HBM) vs {n_map}-pt local map, "
                               f"full ICP loop (kNN + plane fit + Jacobian + 6x6 reduce) in HIP",
                   "queries": Q, "map_points": int(map_total), "map_points_this_rank": int(map_rank),
                   "max_iterations": max_outer, "lm_iterations": lm_iters, "plane_res": sc.plane_res, "k": 5,
                   "parallelism": ("single GPU" if world == 1 else f"map sharded by brick-hash x{world}, 45-fp64 RCCL all-reduce per evaluation"),
                   "distinct_scans": args.scans},
        "executed": {"outer_iterations_per_step": iters_outer / args.steps, "lm_iterations_per_step": iters_lm / args.steps,
                     "accepted_correspondences": accepted / args.steps,
                     "pose_error_vs_ground_truth_m_rad": [max(e[0] for e in errs), max(e[1] for e in errs)]},
        "roofline": {"bound": "hbm", "kernel": "knn_plane_kernel", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                     "algorithmic_bytes_per_launch": b_knn, "avg_launch_ms": knn_ms, "launches": int(tm.knn_launches),
                     "timing": "HIP events attached to the kernel's dispatch (hipExtLaunchKernelGGL) on the context's stream, inside the timed region, "
                               "on every 3rd registration (every launch with --time-all-kernels); no-op launches after convergence excluded",
                     "queries_per_launch": q_per_launch, "map_points_in_touched_cubes": m_per_launch,
                     "note": "B = 36*Q + 12*M_t (SURVEY 8d); with M_t = whole map (BASELINE.md table) B would be %.0f and frac %.4f"
                             % (b_knn_whole_map, (b_knn_whole_map / (knn_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if knn_ms > 0 else 0.0)},
        "host": {"c_abi_ms_per_step": tm.host_ms_total / max(tm.registrations, 1),
                 "note": "wall time inside so_icp_register_dev (enqueue + wait + post-processing); ms_per_step - this = Python/ctypes overhead of the bench loop"},
        # ms of one registration by kernel family, from the profiling pass after the timed region (real launches only:
        # no-op launches after convergence excluded).  solve = plane fit + every LM evaluation + controller (one persistent
        # launch per outer iteration on one GPU; eval + all-reduce + controller launches when the map is sharded).
        "kernels": ({"note": "profiling pass after the timed region: every launch bracketed by HIP events",
                     "registrations_profiled": prof["registrations"],
                     "knn_ms_per_registration": prof["knn_ms"], "solve_ms_per_registration": prof["solve_ms"],
                     "binning_ms_per_registration": prof["binning_ms"],
                     "knn_launches_per_registration": prof["knn_launches"], "solve_launches_per_registration": prof["solve_launches"],
                     "rest_ms_per_registration": max(ms_per_step_for_split - prof["knn_ms"] - prof["solve_ms"] - prof["binning_ms"], 0.0)}
                    if prof else None),
    }

    # ---- CPU baseline: the oracle (restatement of the reference CPU path), same scans, bounded sample
    if not args.no_cpu_baseline:
        import oracle_py
        om = oracle_py.OracleMap(plane_res=sc.plane_res)
        om.add_surf(slam.export_map() if world == 1 else sc.map_points, raw=(world == 1))
        cfg = oracle_py.default_config(max_iterations=max_outer, lm_max_iterations=lm_iters, use_grid_knn=1)
        om.ensure_grids()  # index build is map maintenance, not registration (the reference builds octrees at insert time)
        oracle_py.set_num_threads(1)  # faithful: the reference's correspondence loop is serial, Ceres num_threads = 1
        t0 = time.perf_counter()
        worst = (0.0, 0.0)
        for i in range(args.cpu_sample):
            orc, opose, ost, _ = om.register(scans[i % args.scans], guesses[i % args.scans], cfg)
            if i < len(poses):
                e = synth.pose_error(poses[i], opose)
                worst = (max(worst[0], e[0]), max(worst[1], e[1]))
        t_cpu = time.perf_counter() - t0
        out["cpu_baseline"] = {"value": args.cpu_sample / t_cpu, "unit": "registrations/s", "cores": 1, "kind": "port",
                               "sample": f"{args.cpu_sample} registration(s) of the same {Q}-pt scans vs the same map, oracle/liboracle.so "
                                         f"(exact grid k-NN), 1 thread, {t_cpu:.1f} s",
                               "host_cpu": _cpu_model(), "host_cores": os.cpu_count()}
        # the "fair" CPU number: the same oracle with OpenMP over the queries on every host core
        try:
            ncore = max(1, (os.cpu_count() or 2) // 2)  # physical cores (SMT siblings do not help the fp64 loops)
            oracle_py.set_num_threads(ncore)
            om.register(scans[0], guesses[0], cfg)  # warm the threads
            t0 = time.perf_counter()
            reps = max(2, args.cpu_sample)
            for i in range(reps):
                om.register(scans[i % args.scans], guesses[i % args.scans], cfg)
            t_all = (time.perf_counter() - t0) / reps
            out["cpu_baseline_all_cores"] = {"value": 1.0 / t_all, "unit": "registrations/s", "cores": ncore, "kind": "port",
                                             "sample": f"{reps} registrations, oracle with OpenMP over the queries, {ncore} threads"}
            oracle_py.set_num_threads(1)
        except Exception as e:  # the number of record is the 1-thread baseline above
            out["cpu_baseline_all_cores"] = {"error": str(e)}
        out["parity_vs_oracle_m_rad"] = [worst[0], worst[1]]
        out["speedup_vs_cpu_1thread"] = value * t_cpu / args.cpu_sample
    print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


if __name__ == "__main__":
    main()
