"""-m gpu: N > 1 beyond the brick-hash shards of test_gpu_configs.py / test_gpu_peer.py.

  * SO_ICP_SHARD_QUERIES (map replicated on every rank, the scan's 64-point segments dealt round-robin, the same 45-double
    exchange per evaluation): ranks on ONE device, joined by an in-process group or by the peer exchange -- results equal to the
    single context (iteration counts, termination codes, histograms, accepted counts; poses to 1e-9) for even / odd world sizes,
    scan lengths that are not a multiple of 64, and with the max_surface_features sampling rule active;
  * the tests of the data plane ACROSS devices, which a one-GPU box cannot run: they skip unless so_icp_device_count() >= 2 and
    run the day a multi-GPU node is visible -- one process per device, the parent as control plane over pipes:
      - RCCL: ncclAllReduce of the 45 sums per evaluation (+ the map-count all-reduce of the insert) with both shard modes;
      - the peer exchange with hipIpcOpenMemHandle to ANOTHER device (tagged 16-byte chunks over xGMI), both shard modes;
      - the RCCL all-gather re-cut of the shards at a planeRes change (so_icp_set_resolution under a communicator).
"""
import os
import sys
import threading

import numpy as np
import pytest

from helpers import pose_close
from superodom_amd import synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CASES = ((3, 0.4, 3.0), (9, 0.1, 1.0))  # (scan, guess dt, guess dtheta): the first needs several outer iterations


def _summary(res):
    rc, pose, st = res
    return (rc, np.asarray(pose).tolist(), st.n_iterations, st.flags,
            [(st.iterations[it].lm_iterations, st.iterations[it].num_successful_steps, st.iterations[it].termination,
              st.iterations[it].num_surf_from_scan, list(st.iterations[it].reject_hist), list(st.iterations[it].obs_hist))
             for it in range(st.n_iterations)], st.laser_cloud_surf_stack_num, st.laser_cloud_surf_from_map_num)


def _assert_same(a, ref, tag):
    assert a[0] == ref[0] == 0, (tag, a[0], ref[0])
    assert a[2] == ref[2], (tag, "outer iterations", a[2], ref[2])
    for it in range(ref[2]):
        assert a[4][it] == ref[4][it], (tag, it, a[4][it], ref[4][it])
    assert a[5] == ref[5] and a[6] == ref[6], (tag, a[5:], ref[5:])
    ok, dt, dr = pose_close(np.array(a[1]), np.array(ref[1]), 1e-9, 1e-9)
    assert ok, (tag, dt, dr)


def _in_threads(n, fn):
    out = [None] * n

    def run(r):
        out[r] = fn(r)
    th = [threading.Thread(target=run, args=(r,)) for r in range(n)]
    [t.start() for t in th]; [t.join(240) for t in th]
    assert all(o is not None for o in out), "a rank did not return"
    return out


@pytest.mark.parametrize("world,n_keep,max_sf", [(2, None, -1), (3, 7001, -1), (4, 6000, 2500)])
def test_query_split_ranks_equal_the_single_context(soicp, world, n_keep, max_sf):
    """SO_ICP_SHARD_QUERIES with `world` contexts on this one GPU joined by an in-process group (per-evaluation launches, the sums
    through host memory): every rank registers its 64-point segments of the scan against the whole map."""
    sc = synth.Scene("small")
    mk = dict(plane_res=sc.plane_res, line_res=sc.plane_res / 2, max_surface_features=max_sf, max_iterations=5)
    one = soicp.LidarSlamGpu(**mk)
    one.add_surf_point_cloud(sc.map_points)
    ranks = [soicp.LidarSlamGpu(rank=r, world_size=world, shard_mode=soicp.SHARD_QUERIES, **mk) for r in range(world)]
    for sh in ranks:
        sh.comm_init_inprocess(0xA000 + world)
        assert sh.add_surf_point_cloud(sc.map_points) == len(sc.map_points)
        assert sh.map_size(this_rank=True) == (len(sc.map_points), len(sc.map_points)), "the map is replicated, not sharded"
    for i, dt, dth in CASES:
        scan = sc.scan(i)
        if n_keep:
            scan = np.ascontiguousarray(scan[:n_keep])
        guess = sc.guess(i, dt=dt, dth_deg=dth)
        ref = _summary(one.register(scan, guess))
        if i == 3 and max_sf < 0:
            assert ref[2] >= 3, "the test needs several outer iterations"
        res = _in_threads(world, lambda r: _summary(ranks[r].register(scan, guess)))
        for r in range(world):
            assert res[r][1] == res[0][1], "all ranks hold the same sums: identical decisions, identical bits"
            assert res[r][3] & soicp.FLAG_QUERY_SPLIT and res[r][3] & soicp.FLAG_SHARDED
            _assert_same(res[r], ref, ("query split", world, i, r))
    for sh in ranks + [one]:
        sh.close()


def test_query_split_over_the_peer_exchange_in_one_process(soicp, monkeypatch):
    """The same with the ranks' persistent solve launches trading their records through the inboxes (two contexts of one process,
    100 workgroups each so that both launches are co-resident on this one device)."""
    sc = synth.Scene("small")
    mk = dict(plane_res=sc.plane_res, line_res=sc.plane_res / 2, max_surface_features=-1, max_iterations=5)
    one = soicp.LidarSlamGpu(**mk)
    one.add_surf_point_cloud(sc.map_points)
    monkeypatch.setenv("SOICP_SOLVE_WORKGROUPS", "100")
    ranks = [soicp.LidarSlamGpu(rank=r, world_size=2, shard_mode=soicp.SHARD_QUERIES, **mk) for r in range(2)]
    for sh in ranks:
        sh.add_surf_point_cloud(sc.map_points)
    handles = [sh.peer_export() for sh in ranks]
    assert _in_threads(2, lambda r: ranks[r].peer_connect(handles)) == [True, True], [sh.last_error() for sh in ranks]
    for sh in ranks:
        sh.peer_enable(True)
    for i, dt, dth in CASES:
        scan, guess = sc.scan(i), sc.guess(i, dt=dt, dth_deg=dth)
        ref = _summary(one.register(scan, guess))
        res = _in_threads(2, lambda r: _summary(ranks[r].register(scan, guess)))
        for r in range(2):
            assert res[r][1] == res[0][1]
            assert not (res[r][3] & soicp.FLAG_PER_EVAL_LAUNCHES), "the persistent solve launch must survive N > 1"
            _assert_same(res[r], ref, ("query split, peer exchange", i, r))
    for sh in ranks + [one]:
        sh.close()


# ------------------------------------------------------------------------------------------------------------------------
# across devices: one process per GPU (skipped on a one-GPU box)
# ------------------------------------------------------------------------------------------------------------------------
def _rank_process(rank, world, shard_mode, transport, recut, conn):
    try:
        sys.path.insert(0, ROOT)
        from superodom_amd import binding as soicp, synth as sy
        sc = sy.Scene("small")
        sh = soicp.LidarSlamGpu(device_id=rank, plane_res=sc.plane_res, line_res=sc.plane_res / 2, max_surface_features=-1, max_iterations=5,
                                rank=rank, world_size=world, shard_mode=shard_mode)
        if transport == "rccl" or recut:
            if rank == 0:
                conn.send(soicp.comm_unique_id())
            uid = conn.recv()
            sh.comm_init(uid)
        sh.add_surf_point_cloud(sc.map_points)  # (collective under a communicator: the per-block counts of the full map)
        if transport == "peer":
            conn.send(sh.peer_export())
            handles = conn.recv()
            ok = sh.peer_connect(handles)
            conn.send((ok, sh.last_error()))
            sh.peer_enable(conn.recv())
        out = []
        for i, dt, dth in CASES:
            conn.recv()  # barrier
            out.append(_summary(sh.register(sc.scan(i), sc.guess(i, dt=dt, dth_deg=dth))))
            conn.send(True)
        if recut:  # planeRes change over resident shards: the RCCL all-gather of every rank's owned points, then the same scans again
            conn.recv()
            sh.set_resolution(0.2, 0.4)
            sizes = sh.map_size(this_rank=True)
            out.append(("sizes", sizes))
            for i, dt, dth in CASES:
                conn.recv()
                out.append(_summary(sh.register(sc.scan(i), sc.guess(i, dt=dt, dth_deg=dth))))
                conn.send(True)
        conn.send(out)
        conn.recv()
        sh.close()
    except BaseException as e:  # noqa: BLE001
        import traceback
        conn.send(("error", f"rank {rank}: {e!r}\n{traceback.format_exc()}"))
        raise


def _run_ranks(soicp, world, shard_mode, transport, recut=False):
    import multiprocessing as mp
    ctx = mp.get_context("spawn")
    pipes = [ctx.Pipe() for _ in range(world)]
    procs = [ctx.Process(target=_rank_process, args=(r, world, shard_mode, transport, recut, pipes[r][1])) for r in range(world)]
    for p in procs:
        p.start()
    conns = [pp[0] for pp in pipes]

    def recv_all(which=None):
        out = []
        for c in (conns if which is None else [conns[k] for k in which]):
            assert c.poll(300), "a rank process did not answer"
            out.append(c.recv())
        errors = [o[1] for o in out if isinstance(o, tuple) and len(o) == 2 and o[0] == "error"]
        if errors:
            for p in procs:
                p.kill()
            pytest.fail("a rank process failed:\n" + "\n".join(errors))
        return out

    def send_all(v):
        for c in conns:
            c.send(v)
    if transport == "rccl" or recut:
        send_all(recv_all([0])[0])  # rank 0's unique id to everybody
    if transport == "peer":
        send_all(recv_all())        # all handles, rank order
        oks = recv_all()
        agreed = all(o[0] for o in oks)
        send_all(agreed)
        assert agreed, f"hipIpc mapping across devices / self-test failed: {oks}"
    n_rounds = len(CASES) * (2 if recut else 1)
    for k in range(n_rounds):
        if recut and k == len(CASES):
            send_all("recut")
        send_all("go")
        recv_all()
    res = recv_all()
    send_all("bye")
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    return res


def _single_context_reference(soicp, plane_res=None):
    sc = synth.Scene("small")
    one = soicp.LidarSlamGpu(plane_res=sc.plane_res, line_res=sc.plane_res / 2, max_surface_features=-1, max_iterations=5)
    one.add_surf_point_cloud(sc.map_points)
    if plane_res:
        one.set_resolution(plane_res / 2, plane_res)
    out = [_summary(one.register(sc.scan(i), sc.guess(i, dt=dt, dth_deg=dth))) for i, dt, dth in CASES]
    one.close()
    return out


def _need_devices(soicp, n):
    have = soicp.device_count()
    if have < n:
        pytest.skip(f"needs {n} HIP devices, this box has {have}: the cross-device data plane (RCCL / hipIpc over xGMI) cannot run here")


@pytest.mark.parametrize("shard_mode", [0, 1])
def test_rccl_two_ranks_on_two_devices_equal_the_single_context(soicp, shard_mode):
    _need_devices(soicp, 2)
    ref = _single_context_reference(soicp)
    res = _run_ranks(soicp, 2, shard_mode, "rccl")
    for k in range(len(CASES)):
        assert res[0][k][1] == res[1][k][1], "both ranks all-reduce the same sums: identical poses"
        for r in range(2):
            assert res[r][k][3] & soicp.FLAG_PER_EVAL_LAUNCHES, "RCCL path: one launch per evaluation"
            _assert_same(res[r][k], ref[k], ("rccl", shard_mode, k, r))


@pytest.mark.parametrize("shard_mode", [0, 1])
def test_peer_exchange_across_two_devices_equals_the_single_context(soicp, shard_mode):
    _need_devices(soicp, 2)
    ref = _single_context_reference(soicp)
    res = _run_ranks(soicp, 2, shard_mode, "peer")
    for k in range(len(CASES)):
        assert res[0][k][1] == res[1][k][1]
        for r in range(2):
            assert not (res[r][k][3] & soicp.FLAG_PER_EVAL_LAUNCHES), "the persistent solve launch must survive N > 1"
            _assert_same(res[r][k], ref[k], ("peer over xGMI", shard_mode, k, r))


def test_rccl_all_gather_recuts_the_shards_at_a_plane_res_change(soicp):
    _need_devices(soicp, 2)
    ref_a = _single_context_reference(soicp)
    ref_b = _single_context_reference(soicp, plane_res=0.4)
    res = _run_ranks(soicp, 2, 0, "rccl", recut=True)
    n = len(CASES)
    for r in range(2):
        for k in range(n):
            _assert_same(res[r][k], ref_a[k], ("before the re-cut", k, r))
        tag, sizes = res[r][n]
        assert tag == "sizes" and sizes[1] < sizes[0], "a shard, not the whole map, after the re-cut"
        for k in range(n):
            _assert_same(res[r][n + 1 + k], ref_b[k], ("after the re-cut", k, r))


def test_eight_ranks_on_eight_devices(soicp):
    """The bench's N = 8 configuration in miniature: peer exchange, brick-hash shards."""
    _need_devices(soicp, 8)
    ref = _single_context_reference(soicp)
    res = _run_ranks(soicp, 8, 0, "peer")
    for k in range(len(CASES)):
        for r in range(8):
            assert res[r][k][1] == res[0][k][1]
            _assert_same(res[r][k], ref[k], ("8 ranks", k, r))
