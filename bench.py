#!/usr/bin/env python3
"""bench.py -- ICP registrations/sec on the BASELINE.json workload.

    python bench.py --gpus 1 --steps K --warmup W            (default: N=1)
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A "step" is ONE scan-to-map registration through so_icp_register -- HOST scan buffer in, pose + statistics out -- of a
synthetic OS1-128 scan (131 072 points) against the 2M-point local map (BASELINE.json configs[2]): spatial binning, then
per outer iteration the k-NN kernel and the persistent solve launch (plane fit + fused LM evaluations + controller),
<= 5 outer x <= 4 LM iterations.  The scan's H2D copy is INSIDE the timed region (SURVEY 8d "scan H2D copy included"):
the next scan is announced with so_icp_stage_scan (copy thread + copy stream) while the current one registers, like the
node's feature callback would (laserMapping.cpp:21-25 receives the cloud long before process() reaches it).  The same
line carries the resident-scan rate (so_icp_register_dev, no copy) and the serial rate (so_icp_register, copy then register).

N > 1 (configs[3]): one process per GPU; the map is sharded by brick-hash of the voxel grid (device-resident shards, query
ownership re-derived every outer iteration), every rank registers the SAME scan over its shard, and the 45 fp64
normal-equation scalars of every evaluation are summed over the ranks -- by the persistent solve launches themselves
through hipIpc-mapped inboxes (peer exchange; gloo carries the handles and the agreement on its self-test), or, when that
is unavailable, by an RCCL all-reduce per evaluation -> total work is fixed: "scaling": "strong".  Every line also carries
`batch64` (configs[4]): 64 hypotheses per scan, the map replicated, the hypotheses split over the ranks, no collective.

Rank 0 prints ONE JSON line with the roofline of the dominant (k-NN) kernel, measured with HIP events attached to the
kernel's dispatch on the library's stream inside the timed region, and the CPU baselines (Oracle-A = restatement with an
exact grid k-NN, Oracle-B = the same with the k-NN through the reference's own octree.h compiled into oracle/_ref),
timed on this host's cores on a bounded sample."""
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
N_SIMD = 1024          # 256 CUs x 4 SIMDs


def _kernels_sha():
    import hashlib
    try:
        return hashlib.sha256(open(os.path.join(ROOT, "superodom_amd", "csrc", "kernels.hip"), "rb").read()).hexdigest()
    except OSError:
        return None


def _load_counters(name):
    """A PMC result file under profiles/ (written by tools/pmc_*.sh on a GPU box) -- quoted ONLY when it was collected on the
    kernel source that is being timed now (kernels_hip_sha256); returns (json or None, reason when None)."""
    path = os.path.join(ROOT, "profiles", name)
    if not os.path.exists(path):
        return None, f"profiles/{name} absent"
    try:
        j = json.load(open(path))
    except Exception as e:  # noqa: BLE001
        return None, f"profiles/{name} unreadable: {e}"
    if j.get("kernels_hip_sha256") != _kernels_sha():
        return None, f"profiles/{name} was collected on another kernels.hip (sha256 {str(j.get('kernels_hip_sha256'))[:12]} != {str(_kernels_sha())[:12]}): stale, not quoted"
    return j, None


class _stdout_to_stderr:
    """gloo announces its connections on the process's stdout (fd 1); rank 0's JSON line must be the only thing there."""

    def __enter__(self):
        sys.stdout.flush()
        self.saved = os.dup(1)
        os.dup2(2, 1)

    def __exit__(self, *exc):
        sys.stdout.flush()
        os.dup2(self.saved, 1)
        os.close(self.saved)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=240)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--workload", default="os1_128_2m")
    ap.add_argument("--scans", type=int, default=4, help="distinct synthetic scans cycled through the steps")
    ap.add_argument("--entry", default="auto", choices=["auto", "chained", "staged", "host", "resident"],
                    help="entry point of the TIMED loop: staged = so_icp_register with the next scan announced by so_icp_stage_scan "
                         "(default, PCIe inside the clock, overlapped); host = so_icp_register alone (copy, then register); "
                         "resident = so_icp_register_dev on scans uploaded before the clock (profiling runs)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the resident / serial / batch64 measurements after the timed region")
    ap.add_argument("--max-outer", type=int, default=5, help="LocalizationICPMaxIter (5 = config of record)")
    ap.add_argument("--time-all-kernels", action="store_true", help="HIP events around every kernel (adds bubbles)")
    ap.add_argument("--cpu-sample", type=int, default=4, help="registrations timed for each CPU baseline")
    ap.add_argument("--shuffle-scan", action="store_true", help="experiment: random point order inside every scan (worst case for the binning atomics)")
    ap.add_argument("--no-kernel-events", action="store_true", help="experiment: no HIP events around the k-NN launches (no roofline)")
    ap.add_argument("--no-profile-pass", action="store_true", help="skip the kernel-split pass after the timed region (runs under rocprofv3 use it: one registration = one set of launches)")
    ap.add_argument("--shard-mode", default="map", choices=["map", "queries"],
                    help="N > 1: how `value` splits the registration over the ranks -- map = brick-hash shards of the voxel map, queries follow "
                         "their cell's owner (BASELINE configs[3], north_star); queries = map replicated, the scan's 64-point segments dealt "
                         "round-robin (equal shares, no halo, no re-binning).  The other mode is measured too and reported under `other_shard_mode`")
    ap.add_argument("--xgmi-exchange-us", type=float, default=3.0, help="assumed cost of one 45-double exchange between the ranks' solve launches "
                                                                        "over xGMI, for `predicted_scaling` (unmeasured on a one-GPU box)")
    ap.add_argument("--scan-buffers", default="pinned", choices=["pinned", "registered", "pageable"],
                    help="where the HOST scan buffers of the timed loop live: pinned = so_icp_host_alloc (a node that keeps its feature clouds in a pinned "
                         "pool: DMA straight from them), registered = numpy memory pinned with so_icp_host_register, pageable = plain numpy memory (the "
                         "staged copies then go through the context's copy thread, which packs them into a pinned buffer first)")
    ap.add_argument("--stage-protocol", default="steady", choices=["steady", "cold"],
                    help="entry 'staged': steady = the clock starts with scan 0's copy done (announced behind the last warm-up registration) and covers "
                         "K registrations + K copies (scans 1 .. K), the state of a node in the middle of a stream; cold = nothing announced when the "
                         "clock starts (K copies, the first hidden by nothing)")
    ap.add_argument("--no-concurrent", action="store_true", help="skip the concurrent-contexts measurement")
    ap.add_argument("--no-stock", action="store_true", help="skip the stock-operating-point measurement (config/os1_128.yaml, livox_mid360.yaml)")
    ap.add_argument("--no-open-scene", action="store_true", help="skip the second perf scene (open hall, 0.5 m / 5 deg guesses)")
    ap.add_argument("--dry-control-plane", action="store_true",
                    help="N > 1 plumbing check without a GPU: spawn / join the ranks, rendezvous over gloo, barrier + max-over-ranks, rank 0 prints "
                         "one JSON line -- what `bench.py --gpus N` does around the registrations (tests/test_bench_launch.py)")
    args = ap.parse_args()

    if "RANK" not in os.environ and args.gpus > 1:
        # `python bench.py --gpus N` without a launcher: become the launcher -- one rank per GPU through torch.distributed.run on
        # 127.0.0.1, the same command line; rank 0's JSON line goes to our stdout unchanged.
        import socket
        import subprocess
        with socket.socket() as s_:
            s_.bind(("127.0.0.1", 0))
            port = s_.getsockname()[1]
        env = dict(os.environ)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # hipIpc handles (peer exchange) and RCCL need dmabuf IPC on this driver
        env.setdefault("OMP_NUM_THREADS", "1")
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd, env=env))

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        args.gpus = world  # (started by a launcher with another rank count: the launcher decides)

    if args.dry_control_plane:
        import datetime
        import torch
        import torch.distributed as dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world > 1:
            with _stdout_to_stderr():
                dist_mod.init_process_group(backend="gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=120))
                dist_mod.barrier()
            tt = torch.tensor([float(rank + 1)], dtype=torch.float64)
            dist_mod.all_reduce(tt, op=dist_mod.ReduceOp.MAX)
            top = float(tt.item())
            dist_mod.barrier()
            dist_mod.destroy_process_group()
        else:
            top = 1.0
        if rank == 0:
            print(json.dumps({"dry_control_plane": True, "n_gpus": world, "max_over_ranks_of_rank_plus_1": top}))
        return

    from superodom_amd import binding, synth

    dist = None
    if world > 1:  # control plane only: rendezvous, barriers, max-over-ranks; the data plane is RCCL inside libsoicp
        import torch.distributed as dist_mod
        dist = dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        import datetime
        # (every wait of the control plane is bounded: a rank that died must end the run, not hang it)
        with _stdout_to_stderr():
            dist.init_process_group(backend="gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=240))
            dist.barrier()

    # ---------------- synthetic workload (seeded) ----------------
    sc = synth.Scene(args.workload)
    max_outer, lm_iters = args.max_outer, 4
    n_dev = max(binding.device_count(), 1)
    device = int(os.environ.get("SOICP_BENCH_DEVICE", local_rank % n_dev))
    ranks_per_device = world if "SOICP_BENCH_DEVICE" in os.environ else (world + n_dev - 1) // n_dev
    if ranks_per_device > 1:
        # development / first-run-proofing on a box with fewer GPUs than ranks: the ranks share devices.  RCCL refuses two ranks per
        # device (the peer exchange carries the sums), and the persistent solve launches of the co-located ranks must fit the
        # device together (one workgroup per compute unit each)
        os.environ.setdefault("SOICP_BENCH_NO_RCCL", "1")
        os.environ.setdefault("SOICP_SOLVE_WORKGROUPS", str(max(16, 200 // ranks_per_device)))
    mk = dict(device_id=device, plane_res=sc.plane_res, line_res=sc.plane_res / 2, max_iterations=max_outer, lm_max_iterations=lm_iters,
              max_surface_features=-1)
    def make_rank_context(shard_mode, time_kernels):
        """This rank's context for one N > 1 mode: communicator, peer handshake (collective over gloo), map."""
        ctx = binding.LidarSlamGpu(rank=rank, world_size=world, time_kernels=time_kernels,
                                   shard_mode=binding.SHARD_QUERIES if shard_mode == "queries" else binding.SHARD_MAP, **mk)
        if world > 1 and not os.environ.get("SOICP_BENCH_NO_RCCL"):  # (NO_RCCL: development on a one-GPU box, where RCCL refuses two ranks per device)
            # (collective and guarded: a rank whose communicator cannot be built must not leave the others inside ncclCommInitRank's
            #  rendezvous without a word -- the peer exchange below can still carry the sums without RCCL)
            import torch
            try:
                uid = [binding.comm_unique_id() if rank == 0 else None]
                ok_uid = 1
            except Exception as e:  # noqa: BLE001
                print(f"rank {rank}: RCCL unavailable ({e})", file=sys.stderr)
                uid, ok_uid = [None], 0
            dist.broadcast_object_list(uid, src=0)
            t_ok = torch.tensor([1 if (ok_uid and uid[0] is not None) else 0])
            dist.all_reduce(t_ok, op=dist.ReduceOp.MIN)
            if bool(t_ok.item()):
                try:
                    ctx.comm_init(uid[0])
                    good = 1
                except Exception as e:  # noqa: BLE001
                    print(f"rank {rank}: ncclCommInitRank failed ({e}); the run continues only if the peer exchange works", file=sys.stderr)
                    good = 0
                t_ok = torch.tensor([good])
                dist.all_reduce(t_ok, op=dist.ReduceOp.MIN)
            if not bool(t_ok.item()):
                os.environ["SOICP_BENCH_NO_RCCL"] = "1"  # (every rank took the same branch: the all-reduce above decided)
        on_peer = False
        if world > 1 and not os.environ.get("SOICP_BENCH_NO_PEER"):
            # peer exchange: the ranks' persistent solve launches trade their records through hipIpc-mapped inboxes instead of an
            # RCCL all-reduce per evaluation.  gloo carries the handles and the agreement on the self-test (include/so_icp.h).
            import torch
            try:
                handles = [None] * world
                dist.all_gather_object(handles, ctx.peer_export())
                dist.barrier()
                ok = ctx.peer_connect(handles)
            except Exception as e:  # noqa: BLE001 -- any failure means "use the collective path"
                print(f"rank {rank}: peer exchange unavailable: {e}", file=sys.stderr)
                ok = False
            t = torch.tensor([1 if ok else 0])
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            on_peer = bool(t.item())
            ctx.peer_enable(on_peer)
        return ctx, on_peer, ctx.add_surf_point_cloud(sc.map_points)
    slam, peer, n_map = make_rank_context(args.shard_mode, 2 if args.time_all_kernels else (0 if args.no_kernel_events else 1))
    scans = [np.ascontiguousarray(sc.scan(i), dtype=np.float32) for i in range(args.scans)]
    if args.shuffle_scan:
        scans = [np.ascontiguousarray(s_[np.random.default_rng(77 + i).permutation(len(s_))]) for i, s_ in enumerate(scans)]
    guesses = [sc.guess(i) for i in range(args.scans)]
    if args.scan_buffers == "pinned":      # host scan buffers in pinned memory: so_icp_stage_scan / so_icp_register copy by DMA straight from them
        scans = [slam.host_alloc_like(s_) for s_ in scans]
    elif args.scan_buffers == "registered":
        for s_ in scans:
            slam.host_register(s_)
    d_scans = [slam.upload_scan(s) for s in scans]  # resident copies: the secondary / profiling loops and --entry resident
    Q = len(scans[0])
    map_total, map_rank = slam.map_size(this_rank=True)

    def barrier():
        slam.synchronize()
        if dist is not None:
            dist.barrier()

    def max_over_ranks(t):
        if dist is None:
            return t
        import torch
        tt = torch.tensor([t], dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        return float(tt.item())

    g64 = [np.ascontiguousarray(g, dtype=np.float64) for g in guesses]
    if args.entry == "auto":
        # N = 1: the K timed steps are ONE so_icp_register_sequence call (round 6) -- the stream entry: guesses chained on the device, no host
        # turn-around between registrations; every copy and binning launch inside the clock.  N > 1: the staged loop of single calls (r05).
        args.entry = "chained" if (world == 1 and args.scans >= 2) else "staged"
    # the chained entry's motion predictions: delta_k = gt(k-1)^-1 o guesses[k] -- an odometry source that lands where the independent
    # guesses of the other entries are, up to the millimetres by which registration k - 1 ends off its ground truth
    def seq_args(steps):
        S_ = args.scans
        d_ = np.zeros((steps, 7)); d_[:, 6] = 1.0
        for k in range(1, steps):
            d_[k] = synth.pose_between(sc.gt_pose((k - 1) % S_), guesses[k % S_])
        return [scans[k % S_] for k in range(steps)], d_
    step_guess = {}  # timed step -> the guess it started from (chained: formed on the device), for the oracle's parity check

    def timed_loop(entry, steps, rewarm=0, slam=slam, protocol=None):
        """`steps` registrations through one entry point, only C calls between the two clock reads (arguments pre-built).
        rewarm: untimed registrations run right before the clock starts, after the argument lists are built -- the W warm-up
        steps of the contract leave the device idle for the milliseconds Python needs to build them, and the first
        registrations after an idle period run on a device that is still raising its clocks."""
        if entry == "chained":
            seq_scans, seq_d = seq_args(steps)
            call, seq_out, seq_g, seq_st, seq_n, _keep = slam.prepare_register_sequence(seq_scans, g64[0], seq_d)
            steady = (protocol or args.stage_protocol) == "steady"
            if steady and rewarm >= 2:
                # Steady state of a stream worked off in calls of K scans: the scan that starts a call was copied and binned beside the last
                # registration of the call before (so_icp_sequence_announce_next) -- here: of the untimed warm-up call --, and the timed call
                # does the same for the scan behind its own last one: K copies and K binnings inside the clock, like the staged entry's steady
                # protocol.  (`--stage-protocol cold`: nothing announced; the first scan's copy and binning are inside, hidden by nothing.)
                ws, wd = seq_args(rewarm)
                slam.sequence_announce_next(seq_scans[0], synth.pose_between(sc.gt_pose((rewarm - 1) % args.scans), guesses[0]))
                assert slam.register_sequence(ws, g64[0], wd)[0] == 0, slam.last_error()
                slam.sequence_announce_next(scans[steps % args.scans], synth.pose_between(sc.gt_pose((steps - 1) % args.scans), guesses[steps % args.scans]))
            else:
                for w in range(rewarm):  # (untimed registrations right before the clock: see below)
                    i = w % args.scans
                    slam.register(scans[i], guesses[i])
            barrier()
            t0 = time.perf_counter()
            rc_ = call()
            slam.synchronize()
            t_local = time.perf_counter() - t0
            slam.sequence_announce_next(None, None)  # (withdraws the copy staged for the call after the clock)
            if dist is not None:
                dist.barrier()
            assert rc_ == 0 and seq_n.value == steps, (entry, rc_, seq_n.value, slam.last_error())
            if not step_guess and args.entry == "chained":  # (the headline's timed loop is the first one through here; N > 1: the headline is the staged loop)
                for k in range(steps):
                    step_guess[k] = np.array(seq_g[k])
            return max_over_ranks(t_local), list(seq_st), [np.array(seq_out[k]) for k in range(steps)]
        stats = [binding.Stats() for _ in range(steps)]
        pose = [np.zeros(7) for _ in range(steps)]
        if entry == "resident":
            calls = [slam.prepare_register_dev(d_scans[k % args.scans][0], d_scans[k % args.scans][1], g64[k % args.scans], stats[k], pose[k])
                     for k in range(steps)]
            stage = None
        else:
            calls = [slam.prepare_register(scans[k % args.scans], g64[k % args.scans], stats[k], pose[k]) for k in range(steps)]
            stage = [slam.prepare_stage_scan(scans[k % args.scans]) for k in range(steps)] if entry == "staged" else None
        rcs = [0] * steps
        steady = bool(stage) and (protocol or args.stage_protocol) == "steady" and args.scans >= 2
        if steady:
            # Steady state of a node that registers a stream of sweeps: while scan k registers, scan k + 1 is on its way -- through the
            # warm-up registrations and across the start of the clock alike.  The clock covers K registrations AND K copies: those of
            # scans 1 .. K, the last one announced during the K-th registration for the registration that would follow -- what every
            # window of K frames of the stream contains; the first timed scan's copy was the share of the window before (announced
            # during the last warm-up registration, which enqueues it like any other).  `--stage-protocol cold` starts the clock with
            # nothing announced: K copies inside, the first one hidden by nothing (the r01 - r04 protocol).
            extra = slam.prepare_stage_scan(scans[steps % args.scans])
            wi = [(w - rewarm) % args.scans for w in range(rewarm)]  # (ends on the scan before the first timed one: no buffer is announced twice)
            if rewarm:
                slam.stage_scan(scans[wi[0]])
            for w in range(rewarm):
                if w + 1 < rewarm:
                    slam.stage_scan(scans[wi[w + 1]])
                else:
                    stage[0]()
                slam.register(scans[wi[w]], guesses[wi[w]])
            if not rewarm:
                stage[0]()
        else:
            for w in range(rewarm):
                i = w % args.scans
                if entry == "resident":
                    slam.register_dev(d_scans[i][0], d_scans[i][1], guesses[i], st)
                else:
                    if entry == "staged":
                        slam.stage_scan(scans[i])
                    slam.register(scans[i], guesses[i])
        barrier()
        t0 = time.perf_counter()
        if steady:
            for k in range(steps):
                (stage[k + 1] if k + 1 < steps else extra)()
                rcs[k] = calls[k]()
        elif stage:
            stage[0]()
            for k in range(steps):
                if k + 1 < steps:
                    stage[k + 1]()
                rcs[k] = calls[k]()
        else:
            for k in range(steps):
                rcs[k] = calls[k]()
        slam.synchronize()
        t_local = time.perf_counter() - t0
        if steady:
            slam.stage_cancel(scans[steps % args.scans])  # (the copy announced for the registration after the clock)
        if dist is not None:
            dist.barrier()
        for k in range(steps):
            assert rcs[k] == 0, (entry, k, rcs[k], slam.last_error())
        return max_over_ranks(t_local), stats, pose

    st = binding.Stats()

    def warm():
        if args.entry == "chained":  # (the sequence entry allocates its scan slots and work lists on first use)
            ws, wd = seq_args(max(2, min(args.warmup, 8)))
            assert slam.register_sequence(ws, g64[0], wd)[0] == 0, slam.last_error()
            return
        for w in range(args.warmup):  # untimed: the entry point of the timed loop, every scan of the rotation at least once
            i = w % args.scans
            if args.entry == "resident":
                slam.register_dev(d_scans[i][0], d_scans[i][1], guesses[i], st)
            else:
                if args.entry == "staged":
                    slam.stage_scan(scans[i])
                slam.register(scans[i], guesses[i])
    if peer:  # the first registrations over the peer path decide, collectively, whether the timed region uses it
        import torch
        try:
            warm()
            good = 1
        except Exception as e:  # noqa: BLE001
            print(f"rank {rank}: peer exchange failed in warm-up, falling back to the RCCL all-reduce: {e}", file=sys.stderr)
            good = 0
        t = torch.tensor([good])
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        if not bool(t.item()):
            peer = False
            slam.peer_enable(False)
            slam.synchronize()
            dist.barrier()
            warm()
    else:
        warm()
    slam.reset_timing()
    peer_lost = False
    if peer:
        # A rank that is descheduled for longer than SOICP_PEER_TIMEOUT_MS in the middle of the run makes its peers' solve launches
        # give up: so_icp_register returns an error on EVERY rank (their states can no longer be assumed equal) and the peer path is
        # off until a new handshake.  The decision to go on with the RCCL all-reduce per evaluation is collective: every rank reports
        # whether its timed loop came through, and if one did not, all of them time the loop again on the fall-back transport.
        import torch
        try:
            t_max, step_stats, step_pose = timed_loop(args.entry, args.steps, rewarm=args.warmup)
            good = 1
        except Exception as e:  # noqa: BLE001
            print(f"rank {rank}: the peer exchange failed in the timed region ({e}); falling back to the RCCL all-reduce", file=sys.stderr)
            good = 0
        t = torch.tensor([good])
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        if not bool(t.item()):
            peer = False; peer_lost = True
            slam.peer_enable(False)
            slam.synchronize()
            dist.barrier()
            warm()
            slam.reset_timing()
            t_max, step_stats, step_pose = timed_loop(args.entry, args.steps, rewarm=args.warmup)
    else:
        entry_fallback = None
        try:
            t_max, step_stats, step_pose = timed_loop(args.entry, args.steps, rewarm=args.warmup)  # (the W warm-up steps run again, back to back with the clock)
        except Exception as e:  # noqa: BLE001 -- the line must be printed: the sequence entry is this round's code, the staged loop is r05's
            if args.entry != "chained":
                raise
            entry_fallback = repr(e)
            print(f"bench: the chained entry failed in the timed region ({e}); timing the staged loop instead", file=sys.stderr)
            step_guess.clear()
            args.entry = "staged"
            slam.synchronize(); warm(); slam.reset_timing()
            t_max, step_stats, step_pose = timed_loop(args.entry, args.steps, rewarm=args.warmup)
    tm = slam.timing()
    iters_outer = iters_lm = accepted = 0
    poses, flags = [], 0
    for k in range(args.steps):
        s_ = step_stats[k]
        flags |= s_.flags
        iters_outer += s_.n_iterations
        for it in range(s_.n_iterations):
            iters_lm += s_.iterations[it].lm_iterations
        accepted += s_.iterations[max(s_.n_iterations - 1, 0)].num_surf_from_scan
        if k < args.scans:
            poses.append(step_pose[k])

    # ---- secondary measurements, same process, after the timed region: the other entry points, the kernel split, batch64
    secondary = {}
    if not args.no_secondary:
        slam.set_time_kernels(0)
        for entry in ("resident", "host", "staged"):
            if entry == args.entry:
                continue
            t_e, _, _ = timed_loop(entry, args.steps, rewarm=args.warmup)
            secondary[entry] = args.steps / t_e
        if args.entry == "chained":
            other_protocol = "cold" if args.stage_protocol == "steady" else "steady"
            t_e, _, _ = timed_loop("chained", args.steps, rewarm=args.warmup, protocol=other_protocol)
            secondary["chained_%s_protocol" % other_protocol] = args.steps / t_e
        if args.entry in ("staged", "chained") and args.scans >= 2:
            # the staged loop under the OTHER start-of-clock protocol: `value` of r01 - r04 was measured with nothing announced
            # when the clock starts ("cold"), r05 on with the stream crossing the clock start ("steady") -- both are in every line
            other_protocol = "cold" if args.stage_protocol == "steady" else "steady"
            t_e, _, _ = timed_loop("staged", args.steps, rewarm=args.warmup, protocol=other_protocol)
            secondary["staged_%s_protocol" % other_protocol] = args.steps / t_e
        if world == 1 and args.scans >= 2 and args.entry != "chained":
            # `steps` registrations as ONE so_icp_register_sequence call (round 6): the guesses chain on the device, guess_k = T_(k-1) o delta_k
            # with delta_k = gt(k-1)^-1 o guesses[k] (an odometry prediction that lands where the headline's guesses are, up to the
            # millimetres registration k - 1 ends from its ground truth), the launches of registration k + 1 are enqueued behind those of k
            # before k has reported, scan k + 1 is copied and binned beside registration k.  Same scans, every copy and binning launch
            # inside the clock.  The first chained registrations are repeated through so_icp_register from the guesses the run reports:
            # equal bits.
            try:
                S_ = args.scans
                seq_scans = [scans[k % S_] for k in range(args.steps)]
                seq_d = np.zeros((args.steps, 7)); seq_d[:, 6] = 1.0
                for k in range(1, args.steps):
                    seq_d[k] = synth.pose_between(sc.gt_pose((k - 1) % S_), guesses[k % S_])
                for rep in range(2):  # (the first run is the warm-up: buffers of the sequence path are allocated on first use)
                    call, seq_out, seq_g, seq_st, seq_n, _keep = slam.prepare_register_sequence(seq_scans, g64[0], seq_d)
                    slam.synchronize()
                    t0 = time.perf_counter()
                    rc_ = call()
                    slam.synchronize()
                    t_seq = time.perf_counter() - t0
                    assert rc_ == 0 and seq_n.value == args.steps, (rc_, seq_n.value, slam.last_error())
                n_ch = sum(1 for s_ in seq_st if s_.flags & binding.FLAG_CHAINED)
                same = True
                for k in range(min(args.steps, 6)):
                    rc_p, pose_p, st_p = slam.register(seq_scans[k], seq_g[k])
                    same = same and rc_p == 0 and bool(np.array_equal(pose_p, seq_out[k])) and st_p.n_iterations == seq_st[k].n_iterations
                secondary["chained"] = args.steps / t_seq
                secondary["chained_detail"] = {"registrations": args.steps, "chained": n_ch,
                                               "outer_iterations_per_step": sum(s_.n_iterations for s_ in seq_st) / args.steps,
                                               "equal_bits_with_so_icp_register_from_the_reported_guesses": bool(same),
                                               "guess_offset_from_the_headline_guesses_m": float(max(np.linalg.norm(seq_g[k][:3] - guesses[k % S_][:3]) for k in range(args.steps)))}
            except Exception as e:  # noqa: BLE001 -- a secondary measurement must not cost the line
                secondary["chained_detail"] = {"unavailable": repr(e)}
        if args.scan_buffers != "pageable" and world == 1:
            # the protocol of rounds 1 - 3 beside the headline's: the same staged loop on PAGEABLE numpy buffers -- the copy
            # thread packs and copies them, nothing is binned ahead
            keep = scans
            scans = [np.array(s_, dtype=np.float32, copy=True) for s_ in keep]
            try:
                timed_loop("staged", args.steps, rewarm=args.warmup)  # (untimed: the copy thread's pinned bounce buffers are allocated on first use)
                t_e, st_pg, _ = timed_loop("staged", args.steps, rewarm=args.warmup)
                secondary["staged_pageable_buffers"] = args.steps / t_e
            finally:
                scans = keep
    other_mode = None
    if world > 1 and not args.no_secondary:
        # the other way of splitting one registration over the ranks, same scans, same protocol (collective: every rank runs it)
        om_name = "queries" if args.shard_mode == "map" else "map"
        try:
            slam.peer_enable(False)
            ctx2, peer2, _ = make_rank_context(om_name, 0)
            scans2 = scans
            if args.scan_buffers == "pinned":
                scans2 = [ctx2.host_alloc_like(s_) for s_ in scans]
            elif args.scan_buffers == "registered":
                for s_ in scans:
                    ctx2.host_register(s_)
            keep = scans
            scans = scans2  # (timed_loop reads the list by name)
            try:
                t2, st2, _ = timed_loop("staged", args.steps, rewarm=args.warmup, slam=ctx2)
            finally:
                scans = keep
            other_mode = {"shard_mode": om_name, "value": args.steps / t2, "unit": "registrations/s", "ms_per_step": 1e3 * t2 / args.steps,
                          "peer_exchange": bool(peer2), "map_points_this_rank": int(ctx2.map_size(this_rank=True)[1]),
                          "outer_iterations_per_step": sum(s_.n_iterations for s_ in st2) / args.steps}
            ctx2.close()
            slam.peer_enable(peer)
        except Exception as e:  # noqa: BLE001
            other_mode = {"shard_mode": om_name, "error": str(e)}
    # N > 1: what N independent registration STREAMS deliver (one robot / one sensor per GPU: map replicated, every rank registers its own
    # scans through the staged entry, no exchange at all) -- the weak-scaling figure beside `value`, which shards ONE registration
    independent = None
    if world > 1 and not args.no_secondary:
        try:
            slam.peer_enable(False)
            solo = binding.LidarSlamGpu(rank=0, world_size=1, time_kernels=0, **mk)
            solo.add_surf_point_cloud(sc.map_points)
            keep = scans
            scans = [solo.host_alloc_like(np.asarray(s_)) for s_ in keep]
            try:
                stream_entry = "chained" if args.scans >= 2 else "staged"  # (the stream entry of round 6: one so_icp_register_sequence call per rank)
                timed_loop(stream_entry, args.steps, rewarm=args.warmup, slam=solo)  # (untimed: buffers, clocks)
                t_s, st_s, _ = timed_loop(stream_entry, args.steps, rewarm=args.warmup, slam=solo)  # (max over the ranks, barrier on both sides)
            finally:
                scans = keep
            independent = {"value": world * args.steps / t_s, "unit": "registrations/s (all ranks together)", "per_rank": args.steps / t_s,
                           "scaling": "weak", "entry": stream_entry, "binned_ahead_steps_rank0": int(sum(1 for s_ in st_s if s_.flags & binding.FLAG_BINNED_AHEAD)),
                           "note": "N independent streams, one per rank, each on the whole map: every rank times the same K registrations, the "
                                   "slowest rank's time counts"}
            solo.close()
            slam.peer_enable(peer)
        except Exception as e:  # noqa: BLE001 -- a secondary measurement must not cost the line
            independent = {"error": str(e)}
    prof = None
    if not args.no_kernel_events and not args.no_profile_pass:
        # kernel split of a registration: every launch bracketed by events (would cost ~25 us per registration inside the
        # timed region); every rank runs it (the collectives need all of them)
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
        slam.set_time_kernels(0)
    batch = None
    if not args.no_secondary:
        # BASELINE configs[4]: 64 hypotheses per scan (SURVEY 8d seeds: +-0.5 m / +-5 deg, 5000 + 64 i + h), map replicated on
        # every rank, hypotheses h = rank, rank + N, ... per rank, no collective until the gather of the poses
        full = slam if world == 1 else binding.LidarSlamGpu(rank=0, world_size=1, time_kernels=0, **mk)
        if world > 1:
            full.add_surf_point_cloud(sc.map_points)
        reps = min(3, args.scans)
        mine = list(range(rank, 64, world))
        hyp = [np.stack([synth.perturb_pose(sc.gt_pose(i), 5000 + 64 * i + h, 0.5, 5.0) for h in mine]) for i in range(reps)]
        d_full = d_scans if world == 1 else [full.upload_scan(s) for s in scans[:reps]]
        full.register_batch(None, hyp[0], d_scan=d_full[0][0], n=d_full[0][1])  # warm: worker contexts + their buffers
        full.synchronize()
        barrier()
        t0 = time.perf_counter()
        outs = []
        for i in range(reps):
            outs.append(full.register_batch(None, hyp[i], d_scan=d_full[i][0], n=d_full[i][1]))
        full.synchronize()
        t_b = max_over_ranks(time.perf_counter() - t0)
        ok = sum(o[0] for o in outs)
        good = sum(1 for i, o in enumerate(outs) for h in range(len(mine)) if synth.pose_error(o[2][h], sc.gt_pose(i))[0] < 0.02)
        outer_b = sum(s_.n_iterations for o in outs for s_ in o[3])
        if dist is not None:
            import torch
            cnt = torch.tensor([ok, good, outer_b], dtype=torch.float64)
            dist.all_reduce(cnt)
            ok, good, outer_b = (int(v) for v in cnt.tolist())
        batch = {"value": 64 * reps / t_b, "unit": "registrations/s", "hypotheses_per_scan": 64, "scans": reps,
                 "hypotheses_per_rank": len(mine), "ms_per_batch": 1e3 * t_b / reps, "returned_ok": ok, "within_2cm_of_ground_truth": good,
                 "outer_iterations_per_hypothesis": outer_b / (64.0 * reps),
                 "parallelism": f"map replicated on {world} GPU(s), hypotheses split over the ranks; on a GPU the hypotheses advance together in batched kernels "
                                "(one binning / k-NN / persistent solve launch per round over all of them, a workgroup group + LM controller per hypothesis), no collective",
                 "scaling": "strong"}
        if world == 1:
            # what a rank of an N-GPU node would run: 64 hypotheses split over N ranks are batches of 64 / N -- measured HERE, on one
            # GPU, so that predicted_scaling.batch64_replicated_map[N] = N x rate(64 / N) is a number and not an extrapolation
            by_size = {}
            for B in (8, 16, 32):
                try:
                    full.register_batch(None, hyp[0][:B], d_scan=d_full[0][0], n=d_full[0][1])
                    full.synchronize()
                    t0 = time.perf_counter()
                    n_b = 0
                    for i in range(reps):
                        for c0 in range(0, 64, B):
                            full.register_batch(None, hyp[i][c0:c0 + B], d_scan=d_full[i][0], n=d_full[i][1])
                            n_b += B
                    full.synchronize()
                    by_size[str(B)] = n_b / (time.perf_counter() - t0)
                except Exception as e:  # noqa: BLE001
                    by_size[str(B)] = {"error": repr(e)}
            by_size["64"] = batch["value"]
            batch["registrations_per_s_by_batch_size"] = by_size
        if world > 1:
            full.close()

    # ---- concurrent contexts: 1 / 2 / 4 contexts on ONE device, each registering its own scans one after the other from its own host
    #      thread and stream -- the single-GPU throughput over independent registrations (a node that tracks several sensors / maps),
    #      beside the latency-chain number `value`.  The persistent solve launches of the contexts must be resident together:
    #      each takes 200 / n workgroups (so_icp_config::solve_workgroups).
    conc = None
    if world == 1 and not args.no_secondary and not args.no_concurrent:
        import threading
        conc = {}
        if True:
            for n_ctx in (1, 2, 4):
                ctxs = []
                try:
                    for c in range(n_ctx):
                        cx = binding.LidarSlamGpu(rank=0, world_size=1, time_kernels=0, solve_workgroups=(200 // n_ctx if n_ctx > 1 else 0), **mk)
                        cx.add_surf_point_cloud(sc.map_points)
                        ctxs.append(cx)
                    steps_c = max(args.steps, 16)
                    plans = []
                    for c, cx in enumerate(ctxs):
                        dsc = [cx.upload_scan(s_) for s_ in scans]
                        st_c = [binding.Stats() for _ in range(steps_c)]
                        po_c = [np.zeros(7) for _ in range(steps_c)]
                        calls = [cx.prepare_register_dev(dsc[(k + c) % args.scans][0], dsc[(k + c) % args.scans][1], g64[(k + c) % args.scans], st_c[k], po_c[k])
                                 for k in range(steps_c)]
                        plans.append((calls, st_c, po_c))
                    gate = threading.Barrier(n_ctx + 1)
                    rcs_c = [[0] * steps_c for _ in range(n_ctx)]

                    def work(c):
                        calls = plans[c][0]
                        for k in range(min(8, steps_c)):
                            calls[k]()
                        ctxs[c].synchronize()
                        gate.wait()
                        for k in range(steps_c):
                            rcs_c[c][k] = calls[k]()
                        ctxs[c].synchronize()
                        gate.wait()
                    th = [threading.Thread(target=work, args=(c,)) for c in range(n_ctx)]
                    for t_ in th:
                        t_.start()
                    gate.wait()
                    t0 = time.perf_counter()
                    gate.wait()
                    dt_c = time.perf_counter() - t0
                    for t_ in th:
                        t_.join()
                    flags_c = 0
                    for c in range(n_ctx):
                        assert all(r == 0 for r in rcs_c[c]), rcs_c[c]
                        for s_ in plans[c][1]:
                            flags_c |= s_.flags
                    # same scans, same guesses as the timed loop: the poses must be the timed loop's poses (bit for bit when the
                    # summation grid is the same; the reduced solve grid of n > 1 changes the order of the sums)
                    dmax = max(max(synth.pose_error(plans[c][2][k], step_pose[(k + c) % args.scans])) for c in range(n_ctx) for k in range(min(steps_c, args.scans)))
                    conc[str(n_ctx)] = {"registrations_per_s": n_ctx * steps_c / dt_c, "solve_workgroups_per_context": (200 // n_ctx if n_ctx > 1 else "one per compute unit"),
                                        "stats_flags": int(flags_c), "max_pose_delta_vs_timed_loop_m_or_rad": dmax}
                except Exception as e:  # noqa: BLE001
                    conc[str(n_ctx)] = {"error": repr(e)}
                finally:
                    for cx in ctxs:
                        cx.close()
        conc["note"] = ("n contexts on one device, one host thread + stream each, resident scans, different scans in flight at the same time; "
                        "stats_flags 0x1/0x2 = a persistent solve launch was abandoned (the contexts' launches did not become resident together)")

    # ---- Localization() per frame (rows f1 / f2 next to the path): registration + device-side map insert, on a context of its own
    #      (the insert changes the map).  "raw_sweep": the 131 072-point scans themselves, the next one announced ahead
    #      (so_icp_stage_scan); "node_order": what laserMapping does per frame -- so_icp_prefilter_scan (VoxelGrid at planeRes,
    #      lmap.cpp:600-651) and Localization() on the filtered cloud (:250-263).
    loc = None
    if world == 1 and not args.no_secondary:
        try:
            ls = binding.LidarSlamGpu(rank=0, world_size=1, time_kernels=0, **mk)
            ls.add_surf_point_cloud(sc.map_points)
            ls.shift_map(sc.gt_pose(0)[:3])
            bufs = [ls.host_alloc_like(x) for x in scans]
            frames = 48

            def run(node_order):
                ts = []
                for k in range(frames + 4):
                    if k == 4:
                        ls.map_size()  # (settles the insert in flight: the clock starts on an idle device)
                        t_all = time.perf_counter()
                    i = k % args.scans
                    t = time.perf_counter()
                    if node_order:
                        d_f, n_f, _ = ls.prefilter_scan(bufs[i], False, sc.plane_res / 2, sc.plane_res)
                        ls.prefilter_announce(bufs[(k + 1) % args.scans])  # (the feature callback has the next raw cloud before process() reaches it, lmap.cpp:250-263)
                        rc_l, _, _ = ls.localization_dev(True, guesses[i], d_f, n_f, 0.1 * k)
                    else:
                        if k == 0:
                            ls.stage_scan(bufs[0])
                        ls.stage_scan(bufs[(k + 1) % args.scans])
                        rc_l, _, _ = ls.localization(True, guesses[i], bufs[i], 0.1 * k)
                    ts.append(time.perf_counter() - t)
                    assert rc_l == 0
                ls.map_size()  # (the last insert included)
                t_done = time.perf_counter()
                ls.prefilter_announce(None)  # (the cloud announced behind the last frame is not going to be filtered: withdrawn)
                return 1e3 * (t_done - t_all) / frames, 1e3 * float(np.median(ts[4:]))

            raw_ms, raw_call = run(False)
            node_ms, node_call = run(True)
            built, handed_back = ls.map_insert_stats()
            loc = {"raw_sweep_ms_per_frame": raw_ms, "raw_sweep_call_ms_median": raw_call, "node_order_ms_per_frame": node_ms,
                   "node_order_call_ms_median": node_call, "frames": frames, "inserts_laid_out_by_the_device": built, "handed_back_to_the_host": handed_back,
                   "note": "ms per frame = wall time of `frames` back-to-back frames incl. every map insert; call = prefilter (node order) + Localization() "
                           "as the caller sees them (the insert completes behind the call)"}
            ls.close()
        except Exception as e:  # a secondary measurement must not cost the line
            loc = {"unavailable": repr(e)}

    # ---- the stock operating point (what a drop-in user of the node runs): config/os1_128.yaml:26-28 -- max_iterations 5,
    #      max_surface_features 2000, planeRes 0.2 -- in node order: so_icp_prefilter_scan (VoxelGrid at planeRes, lmap.cpp:600-651),
    #      then the sampling rule of LS.cpp:346-359 inside the registration; and config/livox_mid360.yaml:26-28 (planeRes 0.1, 4000
    #      features) on a 20 000-point stand-in sweep.  GPU times here; Oracle-A on one core + parity in the cpu_baseline leg.
    stock = None
    stock_cpu_jobs = []
    if world == 1 and not args.no_secondary and not args.no_stock:
        stock = {}

        def stock_case(name, scene, host_scans, gs, plane_res, line_res, max_feat, cite):
            cx = binding.LidarSlamGpu(rank=0, world_size=1, time_kernels=0, device_id=device, plane_res=plane_res, line_res=line_res,
                                      max_iterations=5, lm_max_iterations=4, max_surface_features=max_feat)
            try:
                cx.add_surf_point_cloud(scene.map_points)
                cx.shift_map(scene.gt_pose(0)[:3])
                map_before = cx.export_map()
                filt = []
                for s_ in host_scans:
                    d_f, n_f, _ = cx.prefilter_scan(s_, False, line_res, plane_res)
                    filt.append(cx.download_scan(d_f, n_f))
                d_filt = [cx.upload_scan(f) for f in filt]
                ns = len(host_scans)
                K = 96
                st_k = [binding.Stats() for _ in range(K)]
                po_k = [np.zeros(7) for _ in range(K)]
                gk = [np.ascontiguousarray(g, dtype=np.float64) for g in gs]
                calls = [cx.prepare_register_dev(d_filt[k % ns][0], d_filt[k % ns][1], gk[k % ns], st_k[k], po_k[k]) for k in range(K)]
                for k in range(8):
                    calls[k]()
                cx.synchronize()
                t0 = time.perf_counter()
                for k in range(K):
                    rc_ = calls[k]()
                    assert rc_ == 0, (name, k, rc_)
                cx.synchronize()
                t_reg = (time.perf_counter() - t0) / K
                # the same K registrations as ONE so_icp_register_sequence call on the resident clouds (a backlog in localization mode: guesses
                # chained on the device, delta_k = gt(k-1)^-1 o guess_k)
                t_seq = None
                try:
                    dk = np.zeros((K, 7)); dk[:, 6] = 1.0
                    for k in range(1, K):
                        dk[k] = synth.pose_between(scene.gt_pose((k - 1) % ns), gs[k % ns])
                    seq_list = [d_filt[k % ns] for k in range(K)]
                    for rep_ in range(2):  # (the first call allocates the sequence's buffers)
                        call_s, _o, _g, st_s, n_s, _keep_s = cx.prepare_register_sequence(seq_list, gk[0], dk, on_device=True)
                        cx.synchronize()
                        t0 = time.perf_counter()
                        rc_s = call_s()
                        cx.synchronize()
                        t_seq = (time.perf_counter() - t0) / K
                        assert rc_s == 0 and n_s.value == K, (rc_s, n_s.value)
                except Exception as e_seq:  # noqa: BLE001
                    t_seq = None
                # node order per frame: pre-filter + Localization() (registration + map insert), host buffers in pinned memory
                bufs = [cx.host_alloc_like(x) for x in host_scans]
                frames = 32
                for k in range(frames + 4):
                    if k == 4:
                        cx.map_size()
                        t0 = time.perf_counter()
                    d_f, n_f, _ = cx.prefilter_scan(bufs[k % ns], False, line_res, plane_res)
                    cx.prefilter_announce(bufs[(k + 1) % ns])  # (the next raw cloud's copy runs beside this frame's registration)
                    rc_l, _, _ = cx.localization_dev(True, gs[k % ns], d_f, n_f, 0.1 * k)
                    assert rc_l == 0
                cx.map_size()
                t_frame = (time.perf_counter() - t0) / frames
                cx.prefilter_announce(None)  # (withdrawn: nothing is filtered behind the last frame)
                sampled = int(sum(st_k[0].iterations[0].reject_hist))
                stock[name] = {"config": cite, "plane_res": plane_res, "max_surface_features": max_feat, "max_iterations": 5,
                               "raw_points": int(len(host_scans[0])), "filtered_points": int(len(filt[0])), "sampled_queries": sampled,
                               "registration_ms": 1e3 * t_reg, "registrations_per_s": 1.0 / t_reg, "node_frame_ms": 1e3 * t_frame,
                               "registration_ms_in_a_sequence": (1e3 * t_seq if t_seq else None),
                               "outer_iterations": sum(s_.n_iterations for s_ in st_k) / K,
                               "lm_iterations": sum(s_.iterations[i].lm_iterations for s_ in st_k for i in range(s_.n_iterations)) / K,
                               "note": "registration_ms = so_icp_register_dev on the pre-filtered resident cloud (sampling rule inside); registration_ms_in_a_sequence = "
                                       "the same registrations as one so_icp_register_sequence call (guesses chained on the device); node_frame_ms = "
                                       "so_icp_prefilter_scan + so_icp_localization_dev incl. the map insert, back-to-back frames"}
                stock_cpu_jobs.append((name, map_before, filt, gs, plane_res, max_feat, st_k[:ns], po_k[:ns]))
            finally:
                cx.close()
        try:
            stock_case("os1_128", sc, scans, guesses, sc.plane_res, sc.plane_res / 2, 2000, "config/os1_128.yaml:26-28")
        except Exception as e:  # noqa: BLE001 -- a secondary measurement must not cost the line
            stock["os1_128"] = {"unavailable": repr(e)}
        try:
            scl = synth.Scene("mid360_like")
            stock_case("livox_mid360_like", scl, [np.ascontiguousarray(scl.scan(i), dtype=np.float32) for i in range(args.scans)],
                       [scl.guess(i) for i in range(args.scans)], scl.plane_res, scl.plane_res / 2, 4000, "config/livox_mid360.yaml:26-28 (planeRes 0.1, 4000 features); "
                       "synthetic 40 x 500 sweep standing in for the Mid-360 pattern")
        except Exception as e:  # noqa: BLE001
            stock["livox_mid360_like"] = {"unavailable": repr(e)}

    # ---- second perf scene: an OPEN hall (1 m interior walls: the sweep reaches its 100 m range and touches nearly every occupied
    #      cube: M_t ~ 1.9 M of the 2 M map points) registered from 0.5 m / 5 deg guesses (3 - 4 outer iterations) -- the other end of
    #      the workload space from the headline's occluded rooms (M_t 0.42 M, 2 outer iterations)
    open_scene = None
    open_cpu_job = None
    if world == 1 and not args.no_secondary and not args.no_open_scene:
        try:
            sco = synth.Scene("open_2m")
            cx = binding.LidarSlamGpu(rank=0, world_size=1, time_kernels=1, device_id=device, plane_res=sco.plane_res, line_res=sco.plane_res / 2,
                                      max_iterations=max_outer, lm_max_iterations=lm_iters, max_surface_features=-1)
            n_map_o = cx.add_surf_point_cloud(sco.map_points)
            ns = args.scans
            o_scans = [cx.host_alloc_like(np.ascontiguousarray(sco.scan(i), dtype=np.float32)) for i in range(ns)]
            o_guess = [np.ascontiguousarray(synth.perturb_pose(sco.gt_pose(i), 5000 + 64 * i, 0.5, 5.0), dtype=np.float64) for i in range(ns)]
            K = max(args.steps, 24)
            st_k = [binding.Stats() for _ in range(K)]
            po_k = [np.zeros(7) for _ in range(K)]
            calls = [cx.prepare_register(o_scans[k % ns], o_guess[k % ns], st_k[k], po_k[k]) for k in range(K)]
            stg = [cx.prepare_stage_scan(o_scans[k % ns]) for k in range(K)]
            for k in range(8):
                cx.stage_scan(o_scans[k % ns]); cx.register(o_scans[k % ns], o_guess[k % ns])
            cx.synchronize()
            cx.reset_timing()
            t0 = time.perf_counter()
            stg[0]()
            for k in range(K):
                if k + 1 < K:
                    stg[k + 1]()
                rc_ = calls[k]()
                assert rc_ == 0, (k, rc_, cx.last_error())
            cx.synchronize()
            t_o = (time.perf_counter() - t0) / K
            tmo = cx.timing()
            mko = np.floor((sco.map_points.astype(np.float64) + 25.0) / 50.0).astype(np.int64)
            ck = lambda c: (c[:, 0] + 64) * 16384 + (c[:, 1] + 64) * 128 + (c[:, 2] + 64)
            mt_o = []
            for i in range(ns):
                w_ = np.asarray(o_scans[i], np.float64) @ synth.quat_to_R(o_guess[i][3:]).T + o_guess[i][:3]
                mt_o.append(int(np.isin(ck(mko), np.unique(ck(np.floor((w_ + 25.0) / 50.0).astype(np.int64)))).sum()))
            knn_ms_o = tmo.knn_ms_total / max(tmo.knn_launches, 1)
            b_o = 36.0 * len(o_scans[0]) + 12.0 * float(np.mean(mt_o))
            errs_o = [synth.pose_error(po_k[i], sco.gt_pose(i)) for i in range(ns)]
            open_scene = {"workload": f"open_2m: OS1-128 synthetic scan ({len(o_scans[0])} pts) vs {n_map_o}-pt map of an open hall, guesses +-0.5 m / +-5 deg per axis, staged entry",
                          "value": 1.0 / t_o, "unit": "registrations/s", "ms_per_step": 1e3 * t_o, "steps": K,
                          "outer_iterations_per_step": sum(s_.n_iterations for s_ in st_k) / K,
                          "lm_iterations_per_step": sum(s_.iterations[i].lm_iterations for s_ in st_k for i in range(s_.n_iterations)) / K,
                          "map_points_in_touched_cubes": float(np.mean(mt_o)), "knn_avg_launch_ms": knn_ms_o, "knn_launches_timed": int(tmo.knn_launches),
                          "knn_algorithmic_bytes_per_launch": b_o, "knn_hbm_frac": (b_o / (knn_ms_o * 1e-3) / 1e9 / HBM_PEAK_GBS) if knn_ms_o > 0 else None,
                          "pack_light": {"registrations_with_packed_light_chunks": int(tmo.knn_pack_registrations), "registrations": int(tmo.registrations),
                                         "switched_off_by_leftover_rule": int(tmo.knn_pack_holds)},
                          "stats_flags": int(np.bitwise_or.reduce([s_.flags for s_ in st_k])),
                          "pose_error_vs_ground_truth_m_rad": [max(e[0] for e in errs_o), max(e[1] for e in errs_o)]}
            open_cpu_job = (cx.export_map(), [np.array(x) for x in o_scans], o_guess, sco.plane_res, st_k[:ns], po_k[:ns])
            cx.close()
        except Exception as e:  # noqa: BLE001
            open_scene = {"unavailable": repr(e)}

    if rank != 0:
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return

    value = args.steps / t_max
    # ---- roofline of the dominant kernel (k-NN): algorithmic bytes per launch
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
    tr, tr_why = _load_counters("knn_traffic.json")     # PMC pass results (per launch), see profiles/README.md
    traffic = tr.get("hbm_bytes_per_launch") if tr else None
    ctr, ctr_why = _load_counters("knn_counters.json")
    valu = {"unavailable": ctr_why} if ctr is None else None
    if ctr and ctr.get("valu_wave_insts_per_launch") and knn_ms > 0:
        clk = float(ctr.get("shader_clock_ghz", 2.07))
        insts = float(ctr["valu_wave_insts_per_launch"])
        # Issue floor of the sweep: the SQ counts the quad-cycles its VALUs were busy (SQ_ACTIVE_INST_VALU) -- 1.01 per VALU wave-
        # instruction in this kernel, i.e. 4 cycles each -- so floor = busy quad-cycles x 4 / 1024 SIMDs / shader clock: the time the
        # sweep would take if every SIMD issued VALU work back to back and all SIMDs carried the same load.
        busy = float(ctr.get("active_inst_valu_x4") or insts)
        t_busy = busy * 4 / N_SIMD / (clk * 1e9)
        valu = {"valu_wave_insts_per_launch": insts, "valu_busy_quad_cycles_per_launch": busy, "salu_wave_insts_per_launch": ctr.get("salu_wave_insts_per_launch"),
                "shader_clock_ghz": clk, "issue_bound_ms": t_busy * 1e3, "valu_issue_frac": t_busy / (knn_ms * 1e-3),
                "kernels_hip_sha256": ctr.get("kernels_hip_sha256"), "source": ctr.get("source"), "solve_kernel": ctr.get("solve_kernel")}
    ms_per_step = 1e3 * t_max / args.steps
    shard_info = None
    if world > 1 and args.shard_mode == "queries":
        shard_info = {"queries_owned_per_rank_mean": [Q / world] * world, "imbalance_max_over_mean_per_scan": [1.0] * args.scans,
                      "note": "64-point segments of the scan dealt round-robin: equal shares by construction (+-64 points)"}
    elif world > 1:  # how the queries of the bench scans fall to the ranks under their initial poses (ownership rule of the sharded map)
        hists = np.stack([binding.shard_histogram(scans[i], guesses[i], slam.origin(), sc.plane_res, world) for i in range(args.scans)])
        mean_owned = hists.mean(axis=0)
        shard_info = {"queries_owned_per_rank_mean": [float(v) for v in mean_owned],
                      "imbalance_max_over_mean_per_scan": [float(h.max() / h.mean()) for h in hists],
                      "note": "brick-hash ownership (4 x 4 x 4 cells): the dense near-field floor under the sensor is one or two bricks, so the "
                              "busiest rank sets the sweep and fit times; what sharding can gain is bounded by the step's serial chains (DESIGN section 5)"}

    # ---- predicted_scaling: what splitting ONE registration over N ranks can deliver, from this run's own kernel split -- a claim
    #      the first multi-GPU run tests.  Only the k-NN sweeps and the fit loops shrink with N (by the busiest rank's share of
    #      the queries); binning, the pass chains of the solve and the launches do not, and every pass gains one exchange.
    predicted = None
    if prof and world == 1:
        o_per = iters_outer / args.steps
        passes = iters_lm / args.steps + o_per
        fit_loop_ms, knn_floor_ms = 0.0091, 0.005  # fit loop of a fit pass (in-kernel stamps, profiles/r04); a sweep never beats its launch + slowest chunk
        rest_ms = max(ms_per_step - prof["knn_ms"] - prof["solve_ms"] - prof["binning_ms"], 0.0)

        def step_ms(n, mode):
            if n == 1:
                return ms_per_step
            if mode == "map":
                h = np.stack([binding.shard_histogram(scans[i], guesses[i], slam.origin(), sc.plane_res, n) for i in range(args.scans)]) if world == 1 \
                    else None
                share = float(np.mean(h.max(axis=1) / h.sum(axis=1))) if h is not None else 1.0 / n
                binning = prof["binning_ms"] * o_per          # re-binned under the current pose every outer iteration
            else:
                share = 1.0 / n
                binning = max(prof["binning_ms"] * share, 0.012)  # three launches
            knn = max(prof["knn_ms"] / max(prof["knn_launches"], 1) * share, knn_floor_ms) * o_per
            solve = prof["solve_ms"] - o_per * fit_loop_ms * (1.0 - share) + passes * args.xgmi_exchange_us * 1e-3
            return binning + knn + solve + rest_ms
        predicted = {"unit": "registrations/s", "model": "step(N) = binning + outer x max(knn x share, 5 us) + [solve - outer x fit_loop x (1 - share) + passes x exchange] + rest; "
                                                         "share = the busiest rank's part of the queries (brick-hash histogram of the bench scans / 1/N)",
                     "assumptions": {"xgmi_exchange_us_per_pass": args.xgmi_exchange_us, "fit_loop_us": 1e3 * fit_loop_ms, "passes_per_step": passes,
                                     "measured_on": "this run's profiling pass (N = %d)" % world},
                     "map_shards": {str(n): 1.0 / step_ms(n, "map") * 1e3 for n in (1, 2, 4, 8)} if world == 1 else None,
                     "query_split": {str(n): 1.0 / step_ms(n, "queries") * 1e3 for n in (1, 2, 4, 8)},
                     # N ranks x the MEASURED one-GPU rate of a batch of 64 / N hypotheses (batch64.registrations_per_s_by_batch_size)
                     "batch64_replicated_map": ({str(n): (n * batch["registrations_per_s_by_batch_size"][str(64 // n)]
                                                          if isinstance(batch["registrations_per_s_by_batch_size"].get(str(64 // n)), float) else None)
                                                 for n in (1, 2, 4, 8)} if (batch and world == 1 and "registrations_per_s_by_batch_size" in batch) else None),
                     "batch64_8gpu_over_1gpu": ((8 * batch["registrations_per_s_by_batch_size"]["8"] / batch["value"])
                                                if (batch and world == 1 and isinstance(batch.get("registrations_per_s_by_batch_size", {}).get("8"), float)) else None),
                     "note": "one registration is a latency chain of ~7 passes: sharding it cannot scale; throughput over independent registrations "
                             "(batch64: hypotheses split over the ranks, no collective) is what scales with N"}

    errs = [synth.pose_error(poses[i], sc.gt_pose(i)) for i in range(len(poses))]
    entry_text = {"staged": "so_icp_register on HOST scan buffers, the next scan announced with so_icp_stage_scan (DMA on the copy stream straight "
                            "from pinned caller memory, enqueued by the registration in flight; copy thread for pageable buffers): "
                            "every scan's H2D copy AND its spatial binning (scan_keys -> bin_offsets -> bin_place, enqueued behind the copy on the copy queue) "
                            "are inside the timed region, overlapped with the previous registration",
                  "host": "so_icp_register on HOST scan buffers, copy then register (nothing overlapped)",
                  "resident": "so_icp_register_dev on scans uploaded BEFORE the timed region",
                  "chained": "the K steps as ONE so_icp_register_sequence call on HOST scan buffers (the stream entry, round 6): guess_k = T_(k-1) o delta_k formed "
                             "on the device (laserMapping.cpp:345-372), the launches of registration k + 1 enqueued behind those of k before k has reported, "
                             "scan k + 1 copied (DMA from pinned caller memory) and binned beside registration k: every H2D copy AND every binning launch inside "
                             "the timed region; each registration is bit for bit the one so_icp_register runs from the same guess (entry_points.staged = that "
                             "loop of single calls, the r05 entry)"}[args.entry]
    out = {
        "metric": "icp_registrations_per_sec", "value": value, "unit": "registrations/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"{args.workload}: OS1-128 synthetic scan ({Q} pts) vs {n_map}-pt local map, "
                               f"full ICP loop (kNN + plane fit + Jacobian + 6x6 reduce) in HIP; " + entry_text,
                   "entry": args.entry, "entry_fallback": (entry_fallback if world == 1 else None), "queries": Q, "map_points": int(map_total), "map_points_this_rank": int(map_rank),
                   "max_iterations": max_outer, "lm_iterations": lm_iters, "plane_res": sc.plane_res, "k": 5,
                   "parallelism": ("single GPU" if world == 1 else
                                   (f"map replicated on {world} ranks, the scan's 64-point segments dealt round-robin, " +
                                    ("persistent solve launches trading their 45-double records through hipIpc-mapped inboxes over xGMI (peer exchange)" if peer
                                     else "45-fp64 RCCL all-reduce per evaluation")) if args.shard_mode == "queries" else
                                   (f"map sharded by brick-hash x{world}, ownership re-derived every outer iteration, persistent solve launches trading their "
                                    f"45-double records through hipIpc-mapped inboxes over xGMI (peer exchange)" if peer else
                                    f"map sharded by brick-hash x{world}, ownership re-derived every outer iteration, 45-fp64 RCCL all-reduce per evaluation")),
                   "shard_mode": (None if world == 1 else args.shard_mode),
                   "peer_exchange": bool(peer), "peer_exchange_lost_mid_run": bool(peer_lost),
                   "transport": (None if world == 1 else ("peer exchange: tagged 16-byte chunks pushed into hipIpc-mapped inboxes by the persistent solve launches"
                                                          if peer else "RCCL all-reduce of 45 fp64 per evaluation + controller launch")),
                   "shards": shard_info,
                   "distinct_scans": args.scans},
        "executed": {"outer_iterations_per_step": iters_outer / args.steps, "lm_iterations_per_step": iters_lm / args.steps,
                     "accepted_correspondences": accepted / args.steps, "stats_flags": int(flags),
                     "pose_error_vs_ground_truth_m_rad": [max(e[0] for e in errs), max(e[1] for e in errs)]},
        # the same registrations through the other entry points, `steps` each, after the timed region
        "entry_points": {"note": "registrations/s; 'staged' and 'host' include the scan's H2D copy (1.5 MB), 'resident' does not; "
                                 "'staged_pageable_buffers' = the staged loop on pageable numpy buffers (copy thread, nothing binned ahead: the r01 - r03 protocol); "
                                 "'chained' = the same registrations as ONE so_icp_register_sequence call (guesses chained on the device, the next registration enqueued behind "
                                 "the current one: no host turn-around between registrations; chained_detail); 'staged_cold_protocol' / 'staged_steady_protocol' = the staged loop under the start-of-clock protocol `value` was NOT measured with "
                                 "(host.stage_protocol names the one it was: 'cold' = nothing announced when the clock starts, r01 - r04; 'steady' = the stream crosses the clock start, r05 on)",
                         args.entry: value, **secondary},
        "roofline": {"bound": "hbm", "kernel": "knn_plane_kernel", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_note": tr_why or tr.get("source"),
                     "algorithmic_bytes_per_launch": b_knn, "avg_launch_ms": knn_ms, "launches": int(tm.knn_launches),
                     "timing": "HIP events attached to the kernel's dispatch (hipExtLaunchKernelGGL) on the context's stream, inside the timed region, "
                               "on every 3rd registration (every launch with --time-all-kernels); no-op launches after convergence excluded",
                     "queries_per_launch": q_per_launch, "map_points_in_touched_cubes": m_per_launch,
                     "valu_issue": valu,
                     "note": "B = 36*Q + 12*M_t (SURVEY 8d); with M_t = whole map (BASELINE.md table) B would be %.0f and frac %.4f; "
                             "measured traffic = algorithmic bytes (no re-read waste): the sweep is bounded by instruction issue and by its slowest "
                             "wavefronts, not by HBM -- see valu_issue"
                             % (b_knn_whole_map, (b_knn_whole_map / (knn_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if knn_ms > 0 else 0.0)},
        "host": {"c_abi_ms_per_step": tm.host_ms_total / max(tm.registrations, 1),
                 # what the timed region holds beyond the registration cores: the first scan's copy (nothing hides it), the scan
                 # hand-overs, the ctypes calls, the final synchronisation -- per step
                 "fixed_overhead_ms_per_step": ms_per_step - tm.host_ms_total / max(tm.registrations, 1),
                 "stage_wait_ms_per_step": tm.stage_wait_ms_total / max(tm.registrations, 1),
                 "staged_by_dma_from_registered_memory": int(tm.staged_direct), "staged_through_copy_thread": int(tm.staged_copied),
                 "stage_declined": int(tm.stage_declined), "stage_protocol": (args.stage_protocol if args.entry in ("staged", "chained") else None),
                 # timed registrations whose scan had been spatially binned behind its DMA, on the copy queue, while the registration before
                 # it ran (so_icp_stats::flags & SO_ICP_FLAG_BINNED_AHEAD; SOICP_PREBIN=0 disables): they start with their k-NN sweep
                 "binned_ahead_timed_steps": int(sum(1 for s_ in step_stats if s_.flags & binding.FLAG_BINNED_AHEAD)),
                 # timed registrations whose launches were enqueued behind the registration before them, before that one had reported
                 # (so_icp_register_sequence, SO_ICP_FLAG_CHAINED; SOICP_SEQ_CHAIN=0 disables), and how often a chain broke
                 "chained_timed_steps": int(sum(1 for s_ in step_stats if s_.flags & binding.FLAG_CHAINED)), "chain_breaks": int(tm.seq_chain_breaks),
                 "scan_buffers": {"pinned": "pinned host memory (so_icp_host_alloc)", "registered": "registered host memory (so_icp_host_register)",
                                  "pageable": "pageable"}[args.scan_buffers],
                 "note": "c_abi = wall time inside the registration core (enqueue + wait + post-processing); stage_wait = host time the registrations "
                         "waited for a staged scan still on its way through the copy thread (0 with registered buffers: the registration's first kernel "
                         "waits for the DMA on the device)"},
        # ms of one registration by kernel family, from the profiling pass after the timed region (real launches only:
        # no-op launches after convergence excluded).  solve = plane fit + every LM evaluation + controller (one persistent
        # launch per outer iteration on one GPU; eval + all-reduce + controller launches when the map is sharded).
        "kernels": ({"note": "profiling pass after the timed region (resident scans): every launch bracketed by HIP events",
                     "registrations_profiled": prof["registrations"],
                     "knn_ms_per_registration": prof["knn_ms"], "solve_ms_per_registration": prof["solve_ms"],
                     "binning_ms_per_registration": prof["binning_ms"],
                     "knn_launches_per_registration": prof["knn_launches"], "solve_launches_per_registration": prof["solve_launches"],
                     "rest_ms_per_registration": max(ms_per_step - prof["knn_ms"] - prof["solve_ms"] - prof["binning_ms"], 0.0)}
                    if prof else None),
        "batch64": batch,
        "localization": loc,
        "concurrent_contexts": conc,
        "stock": stock,
        "open_scene": open_scene,
        "other_shard_mode": other_mode,
        "independent_streams": independent,
        "predicted_scaling": predicted,
    }

    # ---- CPU baselines: the oracle (restatement of the reference CPU path), same scans, bounded sample, N = 1 only
    if not args.no_cpu_baseline and world == 1:
        import oracle_py
        om = oracle_py.OracleMap(plane_res=sc.plane_res)
        om.add_surf(slam.export_map(), raw=True)
        cfg_a = oracle_py.default_config(max_iterations=max_outer, lm_max_iterations=lm_iters, use_grid_knn=1)
        om.ensure_grids()  # index build is map maintenance, not registration (the reference builds octrees at insert time)
        oracle_py.set_num_threads(1)  # faithful: the reference's correspondence loop is serial, Ceres num_threads = 1
        n_cpu = max(1, args.cpu_sample)
        t0 = time.perf_counter()
        worst = (0.0, 0.0)
        oposes = []
        stats_equal = True  # executed iteration counts, termination codes, 7 + 9 bin histograms of every outer iteration (SURVEY 8d "Parity check")
        for i in range(n_cpu):
            orc, opose, ost, _ = om.register(scans[i % args.scans], step_guess.get(i, guesses[i % args.scans]), cfg_a)  # (chained entry: the guess the device formed)
            oposes.append(opose)
            if i < len(poses):
                e = synth.pose_error(poses[i], opose)
                worst = (max(worst[0], e[0]), max(worst[1], e[1]))
                g = step_stats[i]
                same = orc == 0 and g.n_iterations == ost.n_iterations
                for it in range(min(g.n_iterations, ost.n_iterations)):
                    a_, b_ = g.iterations[it], ost.iters[it]
                    same = same and (a_.lm_iterations, a_.num_successful_steps, a_.termination, a_.num_surf_from_scan) == \
                        (b_.lm_iterations, b_.num_successful_steps, b_.termination, b_.num_surf)
                    same = same and list(a_.reject_hist) == list(b_.reject_hist) and list(a_.obs_hist) == list(b_.obs_hist)
                stats_equal = stats_equal and bool(same)
        t_cpu = time.perf_counter() - t0
        out["cpu_baseline"] = {"value": n_cpu / t_cpu, "unit": "registrations/s", "cores": 1, "kind": "port",
                               "sample": f"{n_cpu} registration(s) of the same {Q}-pt scans vs the same map, oracle/liboracle.so = Oracle-A "
                                         f"(restatement of the reference path, exact grid k-NN), 1 thread, {t_cpu:.1f} s",
                               "host_cpu": _cpu_model(), "host_cores": os.cpu_count()}
        # Oracle-B: the same registration with the k-NN through the reference's OWN octree
        # (flann/octree.h + nanoflann.h compiled into oracle/_ref, Octree::knnNeighbors oct.h:509-519, 1004-1055): the closest
        # thing to the stock binary's CPU path that can run here.  Trees are built before the clock (insert time upstream).
        try:
            if oracle_py.enable_oracle_b():
                cfg_b = oracle_py.default_config(max_iterations=max_outer, lm_max_iterations=lm_iters, use_grid_knn=2)
                om.register(scans[0], guesses[0], cfg_b)  # builds one octree per map block (LocalMap.h:638), untimed
                t0 = time.perf_counter()
                dab = (0.0, 0.0)
                for i in range(n_cpu):
                    orc, bpose, bst, _ = om.register(scans[i % args.scans], guesses[i % args.scans], cfg_b)
                    e = synth.pose_error(oposes[i], bpose)
                    dab = (max(dab[0], e[0]), max(dab[1], e[1]))
                t_b = time.perf_counter() - t0
                out["cpu_baseline_oracle_b"] = {"value": n_cpu / t_b, "unit": "registrations/s", "cores": 1, "kind": "reference",
                                                "sample": f"{n_cpu} registration(s), same scans; k-NN = the reference's flann/octree.h (oracle/_ref/libref_octree.so, "
                                                          f"one tree per 50 m block), everything else = oracle/liboracle.so, 1 thread, {t_b:.1f} s",
                                                "delta_pose_vs_oracle_a_m_rad": [dab[0], dab[1]],
                                                "note": "the stock octree prunes wrongly (oct.h:384-385, 988-990): its neighbours differ for a share of the queries, hence the pose delta"}
                oracle_py.reset_oracle_b()
            else:
                out["cpu_baseline_oracle_b"] = {"error": "oracle/_ref/libref_octree.so not built"}
        except Exception as e:
            out["cpu_baseline_oracle_b"] = {"error": str(e)}
        # the "fair" CPU number: Oracle-A with OpenMP over the queries on every physical host core
        try:
            ncore = max(1, (os.cpu_count() or 2) // 2)  # physical cores (SMT siblings do not help the fp64 loops)
            oracle_py.set_num_threads(ncore)
            om.register(scans[0], guesses[0], cfg_a)  # warm the threads
            t0 = time.perf_counter()
            for i in range(n_cpu):
                om.register(scans[i % args.scans], guesses[i % args.scans], cfg_a)
            t_all = (time.perf_counter() - t0) / n_cpu
            out["cpu_baseline_all_cores"] = {"value": 1.0 / t_all, "unit": "registrations/s", "cores": ncore, "kind": "port",
                                             "sample": f"{n_cpu} registrations, Oracle-A with OpenMP over the queries, {ncore} threads"}
            oracle_py.set_num_threads(1)
        except Exception as e:  # the number of record is the 1-thread baseline above
            out["cpu_baseline_all_cores"] = {"error": str(e)}
        def oracle_check(map_xyz, h_scans, gs, plane_res, max_feat, g_stats, g_poses, n_check):
            """Oracle-A on one core at the same configuration: (registrations/s, worst pose delta, counts / codes / histograms equal)"""
            om2 = oracle_py.OracleMap(plane_res=plane_res)
            om2.add_surf(map_xyz, raw=True)
            om2.ensure_grids()
            cfg2 = oracle_py.default_config(max_iterations=max_outer, lm_max_iterations=lm_iters, use_grid_knn=1, max_surface_features=max_feat)
            w2, same2, t2 = (0.0, 0.0), True, 0.0
            for i in range(n_check):
                t0_ = time.perf_counter()
                orc, opose, ost, _ = om2.register(h_scans[i], gs[i], cfg2)
                t2 += time.perf_counter() - t0_
                e = synth.pose_error(g_poses[i], opose)
                w2 = (max(w2[0], e[0]), max(w2[1], e[1]))
                g = g_stats[i]
                ok2 = orc == 0 and g.n_iterations == ost.n_iterations
                for it in range(min(g.n_iterations, ost.n_iterations)):
                    a_, b_ = g.iterations[it], ost.iters[it]
                    ok2 = ok2 and (a_.lm_iterations, a_.num_successful_steps, a_.termination, a_.num_surf_from_scan) == \
                        (b_.lm_iterations, b_.num_successful_steps, b_.termination, b_.num_surf)
                    ok2 = ok2 and list(a_.reject_hist) == list(b_.reject_hist) and list(a_.obs_hist) == list(b_.obs_hist)
                same2 = same2 and bool(ok2)
            return n_check / t2, [w2[0], w2[1]], same2
        for (name, map_b, filt, gs_, pres, mfeat, g_st, g_po) in stock_cpu_jobs:
            try:
                rate, wpar, same2 = oracle_check(map_b, filt, gs_, pres, mfeat, g_st, g_po, len(filt))
                stock[name].update({"cpu_oracle_a_1thread_registrations_per_s": rate, "cpu_oracle_a_1thread_ms": 1e3 / rate,
                                    "parity_vs_oracle_m_rad": wpar, "parity_iteration_counts_and_histograms_equal": same2,
                                    "speedup_vs_cpu_1thread": stock[name]["registrations_per_s"] / rate})
            except Exception as e:  # noqa: BLE001
                stock[name]["cpu_oracle_a"] = {"error": repr(e)}
        if open_cpu_job is not None:
            try:
                map_b, h_sc, gs_, pres, g_st, g_po = open_cpu_job
                rate, wpar, same2 = oracle_check(map_b, h_sc, gs_, pres, -1, g_st, g_po, min(2, len(h_sc)))
                open_scene.update({"cpu_oracle_a_1thread_registrations_per_s": rate, "parity_vs_oracle_m_rad": wpar,
                                   "parity_iteration_counts_and_histograms_equal": same2, "parity_scans_checked": min(2, len(h_sc))})
            except Exception as e:  # noqa: BLE001
                open_scene["cpu_oracle_a"] = {"error": repr(e)}
        out["parity_vs_oracle_m_rad"] = [worst[0], worst[1]]
        out["parity_iteration_counts_and_histograms_equal"] = stats_equal
        out["parity_scans_checked"] = min(n_cpu, len(poses))
        out["speedup_vs_cpu_1thread"] = value * t_cpu / n_cpu
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
