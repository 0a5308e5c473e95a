"""-m gpu: scans binned AHEAD of their registration (round 5; icp_context.cpp stage_prebin, kernels.hip scan_keys_kernel with
prebin_ctr / reg_begin_prebinned_kernel).  A scan announced with so_icp_stage_scan from DMA-able memory is hash-binned on the copy
queue behind its copy, under the guess of the registration in flight; its own registration starts with the k-NN sweep
(so_icp_stats::flags & SO_ICP_FLAG_BINNED_AHEAD).  The binning only decides which queries share a wavefront: every result must be
bit-identical to the plain registration of the same scan, whatever pose the scan was binned under -- and equal to the oracle."""
import os

import numpy as np
import pytest

from superodom_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _chunked_sweep_for_small_scans(monkeypatch):
    """The scans of these scenes keep <= 4 096 queries, which round 6 sweeps with one wavefront per query and does not bin at all
    (SO_ICP_FLAG_QUERY_WAVES, tests/test_gpu_query_waves.py).  This file is about the binned path: switch that off (read at so_icp_create)."""
    monkeypatch.setenv("SOICP_QUERY_WAVES", "0")


def _stats_tuple(st):
    out = [st.n_iterations]
    for it in range(st.n_iterations):
        a = st.iterations[it]
        out += [a.lm_iterations, a.num_successful_steps, a.termination, a.num_surf_from_scan, tuple(a.reject_hist), tuple(a.obs_hist),
                np.float64(a.final_cost).tobytes(), np.float64(a.initial_cost).tobytes()]
    return out


def _run_stream(slam, scans, guesses, order):
    """the bench / node pattern: announce the next scan, register the current one"""
    res = []
    slam.stage_scan(scans[order[0]])
    for k, i in enumerate(order):
        if k + 1 < len(order):
            slam.stage_scan(scans[order[k + 1]])
        res.append(slam.register(scans[i], guesses[i]))
    return res


@pytest.mark.parametrize("max_surface_features", [-1, 3000])
def test_binned_ahead_bit_identical_to_plain_registration(oracle, soicp, gpu_slam_factory, max_surface_features):
    sc = synth.Scene("small")
    slam = gpu_slam_factory(plane_res=sc.plane_res, line_res=sc.plane_res / 2, max_surface_features=max_surface_features, max_iterations=5)
    slam.add_surf_point_cloud(sc.map_points)
    n = 8
    scans = [slam.host_alloc_like(np.ascontiguousarray(sc.scan(i), dtype=np.float32)) for i in range(n)]
    guesses = [sc.guess(i) for i in range(n)]
    ref = [slam.register(scans[i], guesses[i]) for i in range(n)]
    assert all(r[0] == 0 and not (r[2].flags & (soicp.FLAG_STAGED_SCAN | soicp.FLAG_BINNED_AHEAD)) for r in ref)
    # consecutive scans (a frame of motion between the binning pose and the scan's own guess), then an order with jumps of up to
    # seven frames (1.2 m / 12 degrees: every far chunk straddles cells and cubes it was not binned for)
    n_ahead_total = 0
    for order in (list(range(n)), [0, 7, 1, 6, 2, 5, 3, 4]):
        got = _run_stream(slam, scans, guesses, order)
        ahead = [bool(r[2].flags & soicp.FLAG_BINNED_AHEAD) for r in got]
        # nobody was in flight to bin the first one; the others are binned by the registration before them -- unless that one took
        # more than 300 us to reach the point where it enqueues the copy (then the copy thread did, without binning: a loaded host)
        assert not ahead[0], (order, ahead)
        n_ahead_total += sum(ahead)
        for k, i in enumerate(order):
            rc, pose, st = got[k]
            assert rc == 0 and (st.flags & soicp.FLAG_STAGED_SCAN)
            assert np.array_equal(pose, ref[i][1]), (order, k)
            assert np.array_equal(np.array(st.JtJ), np.array(ref[i][2].JtJ)) and np.array_equal(np.array(st.Jtr), np.array(ref[i][2].Jtr))
            assert _stats_tuple(st) == _stats_tuple(ref[i][2]), (order, k)
    # (how MANY scans were binned ahead depends on the registration thread reaching the point where it enqueues the next copy within
    #  300 us of the announcement: on a loaded host the copy thread wins and nothing is binned -- no functional bug, but then this
    #  test has not exercised the path it is about; the bit-identity asserts above are strict either way)
    if n_ahead_total < 10:
        pytest.xfail(f"only {n_ahead_total} of 14 scans were binned ahead (loaded host?): the binned path was not exercised enough")
    # the same slots again, now with scans from PAGEABLE memory (copy thread, nothing binned ahead): a slot's work list must not
    # outlive the scan it was built for (the slots still hold the lists of the pinned scans above, same sizes)
    pageable = [np.array(s_, dtype=np.float32, copy=True) for s_ in scans]
    order = [3, 1, 4, 0, 2, 6]
    got = _run_stream(slam, pageable, guesses, order)
    for k, i in enumerate(order):
        rc, pose, st = got[k]
        assert rc == 0 and (st.flags & soicp.FLAG_STAGED_SCAN) and not (st.flags & soicp.FLAG_BINNED_AHEAD), (k, hex(st.flags))
        assert np.array_equal(pose, ref[i][1]) and _stats_tuple(st) == _stats_tuple(ref[i][2]), ("pageable after pinned", k)
    # ... and the oracle agrees with what both paths produced
    om = oracle.OracleMap(plane_res=sc.plane_res)
    om.add_surf(slam.export_map(), raw=True)
    cfg = oracle.default_config(max_iterations=5, max_surface_features=max_surface_features)
    for i in (1, 6):
        orc, opose, ost, _ = om.register(np.asarray(scans[i]), guesses[i], cfg)
        assert orc == 0 and ost.n_iterations == ref[i][2].n_iterations
        dt, dr = synth.pose_error(ref[i][1], opose)
        assert dt < 1e-8 and dr < 1e-8


def test_binned_ahead_switch_and_pageable_buffers(soicp, gpu_slam_factory):
    """SOICP_PREBIN=0 (read when the context is created) gives the round-4 path: same bits, flag clear.  Pageable scans go through
    the copy thread and are not binned ahead."""
    sc = synth.Scene("tiny")
    scans_np = [np.ascontiguousarray(sc.scan(i), dtype=np.float32) for i in range(4)]
    guesses = [sc.guess(i) for i in range(4)]
    out = {}
    few_ahead = False
    for mode in ("on", "off", "pageable"):
        if mode == "off":
            os.environ["SOICP_PREBIN"] = "0"
        try:
            slam = gpu_slam_factory(plane_res=sc.plane_res, max_surface_features=-1, max_iterations=5)
        finally:
            os.environ.pop("SOICP_PREBIN", None)
        slam.add_surf_point_cloud(sc.map_points)
        scans = scans_np if mode == "pageable" else [slam.host_alloc_like(s) for s in scans_np]
        out[mode] = _run_stream(slam, scans, guesses, [0, 1, 2, 3])
        flags = [bool(r[2].flags & soicp.FLAG_BINNED_AHEAD) for r in out[mode]]
        assert (not flags[0]) if mode == "on" else flags == [False] * 4, (mode, flags)
        few_ahead = few_ahead or (mode == "on" and sum(flags) < 2)
        slam.close()
    for k in range(4):
        for mode in ("off", "pageable"):
            assert out[mode][k][0] == out["on"][k][0] == 0
            assert np.array_equal(out[mode][k][1], out["on"][k][1]) and _stats_tuple(out[mode][k][2]) == _stats_tuple(out["on"][k][2])
    if few_ahead:
        pytest.xfail("fewer than 2 of 4 scans were binned ahead (loaded host?): the binned path was not exercised enough")


def test_binned_ahead_through_localization_with_map_inserts(soicp, gpu_slam_factory):
    """Localization() frames (registration + device-side map insert + window bookkeeping) with the next scan announced ahead: the map the
    scan was binned against changes before its registration runs (the insert of the frame in between adds points and cells).  Poses,
    statistics and the final map equal those of a context with the binning switched off."""
    sc = synth.Scene("small")
    out = {}
    few_on = False
    for mode in ("on", "off"):
        if mode == "off":
            os.environ["SOICP_PREBIN"] = "0"
        try:
            slam = gpu_slam_factory(plane_res=sc.plane_res, line_res=sc.plane_res / 2, max_surface_features=-1, max_iterations=5)
        finally:
            os.environ.pop("SOICP_PREBIN", None)
        slam.add_surf_point_cloud(sc.map_points[::2])  # half the map: the frames' inserts add the rest of what they see
        scans = [slam.host_alloc_like(np.ascontiguousarray(sc.scan(i), dtype=np.float32)) for i in range(6)]
        res = []
        slam.stage_scan(scans[0])
        for i in range(6):
            if i + 1 < 6:
                slam.stage_scan(scans[i + 1])
            res.append(slam.localization(True, sc.guess(i), scans[i], 0.1 * (i + 1)))
        out[mode] = (res, slam.export_map())
        ahead = [bool(r[2].flags & soicp.FLAG_BINNED_AHEAD) for r in res]
        assert (not ahead[0]) if mode == "on" else ahead == [False] * 6, (mode, ahead)
        few_on = few_on or (mode == "on" and sum(ahead) < 3)
        slam.close()
    for a, b in zip(out["on"][0], out["off"][0]):
        assert a[0] == b[0] and np.array_equal(a[1], b[1]) and _stats_tuple(a[2]) == _stats_tuple(b[2])
    assert np.array_equal(out["on"][1], out["off"][1])
    if few_on:
        pytest.xfail("fewer than 3 of 6 frames were binned ahead (loaded host?): the binned path was not exercised enough")
