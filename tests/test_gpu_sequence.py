"""-m gpu: so_icp_register_sequence (round 6) -- a run of scans whose guesses chain on the device: guess_k = T_(k-1) o delta_k
(laserMapping.cpp:345-372, T_w_lidar = T_w_lidar * prediction), the launches of registration k + 1 enqueued behind those of k before k
has reported.  Required: every registration of the run is THE registration so_icp_register performs from guesses_out[k] -- poses, normal
equations, statistics identical bit for bit --, guesses_out[k] is pose_compose(pose_after of k - 1, delta_k) as the host evaluates it,
the unchained path (SOICP_SEQ_CHAIN=0) returns the same bits, a registration that needs more outer iterations than were enqueued ahead
breaks the chain without changing a result, and the oracle agrees from the same guesses."""
import os

import numpy as np
import pytest

from superodom_amd import synth

pytestmark = pytest.mark.gpu


def _stats_tuple(st):
    out = [st.n_iterations]
    for it in range(st.n_iterations):
        a = st.iterations[it]
        out += [a.lm_iterations, a.num_successful_steps, a.termination, a.num_surf_from_scan, tuple(a.reject_hist), tuple(a.obs_hist),
                np.float64(a.final_cost).tobytes(), np.float64(a.initial_cost).tobytes(), np.array(a.pose_after).tobytes()]
    out += [np.array(st.JtJ).tobytes(), np.array(st.Jtr).tobytes(), tuple(st.pos_in_localmap), st.laser_cloud_surf_from_map_num,
            st.laser_cloud_surf_stack_num, np.array(st.uncertainty).tobytes()]
    return out


def _deltas(sc, ids, off=None):
    """motion predictions that put guess k near sc.guess(ids[k]): gt(k-1)^-1 o guess(k) (the registration of k - 1 ends within millimetres of gt)"""
    d = np.zeros((len(ids), 7)); d[:, 6] = 1.0
    for k in range(1, len(ids)):
        d[k] = synth.pose_between(sc.gt_pose(ids[k - 1]), sc.guess(ids[k]) if off is None or k not in off else synth.perturb_pose(sc.gt_pose(ids[k]), 77 + k, *off[k]))
    return d


def _check_run(slam, plain, scans, pose0, deltas, res, oracle_map=None, oracle=None, cfg=None, on_device_scans=None):
    rc, poses, guesses, stats, n_done = res
    assert rc == 0 and n_done == len(scans), (rc, n_done, slam.last_error())
    assert np.array_equal(guesses[0], np.asarray(pose0, float))
    for k in range(len(scans)):
        if k:
            last = stats[k - 1].iterations[stats[k - 1].n_iterations - 1]
            want = synth.pose_compose(np.array(last.pose_after), deltas[k])
            assert np.array_equal(guesses[k], want), (k, guesses[k] - want)  # the device's composition == the host's, bit for bit
        # the same registration through the ordinary entry point, from the guess the run reports
        prc, ppose, pst = plain.register(scans[k], guesses[k])
        assert prc == 0
        assert np.array_equal(ppose, poses[k]), (k, ppose - poses[k])
        assert _stats_tuple(pst) == _stats_tuple(stats[k]), k
        if oracle_map is not None:
            orc, opose, ost, _ = oracle_map.register(scans[k], guesses[k], cfg)
            assert orc == 0 and ost.n_iterations == stats[k].n_iterations
            for it in range(ost.n_iterations):
                assert list(stats[k].iterations[it].reject_hist) == list(ost.iters[it].reject_hist)
                assert stats[k].iterations[it].lm_iterations == ost.iters[it].lm_iterations
            dt, dr = synth.pose_error(poses[k], opose)
            assert dt < 1e-8 and dr < 1e-8, (k, dt, dr)


@pytest.mark.parametrize("scene,max_feat,ids", [("small", -1, [0, 1, 2, 3, 4, 5, 6]), ("small", 3000, [2, 3, 4, 5]), ("tiny", -1, [0, 1, 2, 3, 4, 5])])
def test_sequence_equals_single_registrations_and_the_oracle(oracle, soicp, gpu_slam_factory, scene, max_feat, ids):
    sc = synth.Scene(scene)
    mk = dict(plane_res=sc.plane_res, line_res=sc.plane_res / 2, max_surface_features=max_feat, max_iterations=5)
    slam, plain = gpu_slam_factory(**mk), gpu_slam_factory(**mk)
    for s in (slam, plain):
        s.add_surf_point_cloud(sc.map_points)
    om = oracle.OracleMap(plane_res=sc.plane_res)
    om.add_surf(slam.export_map(), raw=True)
    scans = [slam.host_alloc_like(np.ascontiguousarray(sc.scan(i), dtype=np.float32)) for i in ids]
    pose0 = sc.guess(ids[0]); deltas = _deltas(sc, ids)
    res = slam.register_sequence(scans, pose0, deltas)
    _check_run(slam, plain, scans, pose0, deltas, res, om, oracle, oracle.default_config(max_iterations=5, max_surface_features=max_feat))
    flags = [st.flags for st in res[3]]
    iters = [st.n_iterations for st in res[3]]
    print(scene, max_feat, "outer iterations", iters, "flags", [hex(f) for f in flags], "chain breaks", slam.timing().seq_chain_breaks)
    # (a registration that needs more outer iterations than the one before it breaks the chain for the scan behind it: how many are chained
    #  depends on the scene; that every result is the single registration's was checked above)
    assert not (flags[0] & soicp.FLAG_CHAINED) and sum(bool(f & soicp.FLAG_CHAINED) for f in flags) >= 2, [hex(f) for f in flags]
    qw = scene == "tiny" or max_feat == 3000
    assert all(bool(f & soicp.FLAG_QUERY_WAVES) == qw for f in flags), [hex(f) for f in flags]
    assert slam.timing().seq_chained >= 2
    # the same run again (state left by the first), from resident scans, and with the chaining switched off: the same bits
    d_scans = [slam.upload_scan(s_) for s_ in scans]
    res_dev = slam.register_sequence(d_scans, pose0, deltas, on_device=True)
    assert res_dev[0] == 0 and np.array_equal(res_dev[1], res[1]) and np.array_equal(res_dev[2], res[2])
    assert [_stats_tuple(a)[:-1] for a in res_dev[3]] == [_stats_tuple(a)[:-1] for a in res[3]]  # (all but the uncertainty, which carries over from the call before)
    os.environ["SOICP_SEQ_CHAIN"] = "0"
    try:
        unchained = gpu_slam_factory(**mk)
    finally:
        del os.environ["SOICP_SEQ_CHAIN"]
    unchained.add_surf_point_cloud(sc.map_points)
    res_u = unchained.register_sequence(scans, pose0, deltas)
    assert res_u[0] == 0 and np.array_equal(res_u[1], res[1]) and np.array_equal(res_u[2], res[2])
    assert [_stats_tuple(a) for a in res_u[3]] == [_stats_tuple(a) for a in res[3]]
    assert not any(st.flags & soicp.FLAG_CHAINED for st in res_u[3])
    for s in (slam, plain, unchained):
        s.close()


def test_a_registration_that_needs_more_iterations_breaks_the_chain_not_the_results(soicp, gpu_slam_factory):
    sc = synth.Scene("small")
    mk = dict(plane_res=sc.plane_res, line_res=sc.plane_res / 2, max_surface_features=-1, max_iterations=5)
    slam, plain = gpu_slam_factory(**mk), gpu_slam_factory(**mk)
    for s in (slam, plain):
        s.add_surf_point_cloud(sc.map_points)
    ids = [0, 1, 2, 3, 4, 5, 6, 7]
    scans = [slam.host_alloc_like(np.ascontiguousarray(sc.scan(i), dtype=np.float32)) for i in ids]
    pose0 = sc.guess(0)
    # scans 3 and 6 start 0.45 m / 4 degrees off: they need more outer iterations than the two their neighbours take
    deltas = _deltas(sc, ids, off={3: (0.45, 4.0), 6: (0.45, 4.0)})
    res = slam.register_sequence(scans, pose0, deltas)
    _check_run(slam, plain, scans, pose0, deltas, res)
    iters = [st.n_iterations for st in res[3]]
    t = slam.timing()
    print("outer iterations per registration", iters, "| chained", t.seq_chained, "| chain breaks", t.seq_chain_breaks,
          "| flags", [hex(st.flags) for st in res[3]])
    assert max(iters) > min(iters), iters
    assert t.seq_chain_breaks >= 1, (iters, t.seq_chain_breaks)
    slam.close(); plain.close()


def test_sequence_edge_cases(soicp, gpu_slam_factory):
    sc = synth.Scene("tiny")
    mk = dict(plane_res=sc.plane_res, line_res=sc.plane_res / 2, max_surface_features=-1, max_iterations=4)
    slam, plain = gpu_slam_factory(**mk), gpu_slam_factory(**mk)
    for s in (slam, plain):
        s.add_surf_point_cloud(sc.map_points)
    # one scan; pageable buffers; an empty scan in the middle (NOT an error of so_icp_register: zero iterations, the chain goes on from the guess)
    one = slam.register_sequence([np.ascontiguousarray(sc.scan(2), dtype=np.float32)], sc.guess(2), np.zeros((1, 7)))
    prc, ppose, pst = plain.register(sc.scan(2), sc.guess(2))
    assert one[0] == 0 and one[4] == 1 and np.array_equal(one[1][0], ppose)
    ids = [0, 1, 2, 3]
    scans = [np.ascontiguousarray(sc.scan(i), dtype=np.float32) for i in ids]
    deltas = _deltas(sc, ids)
    res = slam.register_sequence(scans, sc.guess(0), deltas)
    _check_run(slam, plain, scans, sc.guess(0), deltas, res)
    # count == 0
    z = slam.register_sequence([], sc.guess(0), np.zeros((0, 7)))
    assert z[0] == 0 and z[4] == 0
    slam.close(); plain.close()


def test_a_run_worked_off_in_two_calls_with_the_next_scan_announced(soicp, gpu_slam_factory):
    """so_icp_sequence_announce_next: the scan that starts the second call is copied and binned beside the last registration of the first;
    the two calls give what one call over all eight scans gives, bit for bit (pose0 of the second call = the guess the chain arithmetic forms)."""
    sc = synth.Scene("small")
    mk = dict(plane_res=sc.plane_res, line_res=sc.plane_res / 2, max_surface_features=-1, max_iterations=5)
    slam, plain = gpu_slam_factory(**mk), gpu_slam_factory(**mk)
    for s in (slam, plain):
        s.add_surf_point_cloud(sc.map_points)
    ids = list(range(8))
    scans = [slam.host_alloc_like(np.ascontiguousarray(sc.scan(i), dtype=np.float32)) for i in ids]
    pose0 = sc.guess(0); deltas = _deltas(sc, ids)
    whole = plain.register_sequence(scans, pose0, deltas)
    assert whole[0] == 0
    slam.sequence_announce_next(scans[4], deltas[4])
    a = slam.register_sequence(scans[:4], pose0, deltas[:4])
    assert a[0] == 0 and np.array_equal(a[1], whole[1][:4])
    last = a[3][3].iterations[a[3][3].n_iterations - 1]
    g4 = synth.pose_compose(np.array(last.pose_after), deltas[4])
    d2 = deltas[4:].copy(); d2[0] = [0, 0, 0, 0, 0, 0, 1]
    b = slam.register_sequence(scans[4:], g4, d2)
    assert b[0] == 0 and np.array_equal(b[1], whole[1][4:]) and np.array_equal(b[2], whole[2][4:])
    assert [_stats_tuple(x) for x in list(a[3]) + list(b[3])] == [_stats_tuple(x) for x in whole[3]]
    assert b[3][0].flags & soicp.FLAG_BINNED_AHEAD and b[3][0].flags & soicp.FLAG_STAGED_SCAN
    # an announcement that is not taken up (another scan starts the next call) and a withdrawn one change nothing
    slam.sequence_announce_next(scans[1], deltas[1])
    c1 = slam.register_sequence(scans[:3], pose0, deltas[:3])
    c2 = slam.register_sequence(scans[5:], whole[2][5], np.vstack([[0, 0, 0, 0, 0, 0, 1], deltas[6:]]))
    assert np.array_equal(c1[1], whole[1][:3]) and np.array_equal(c2[1], whole[1][5:])
    slam.sequence_announce_next(None, None)
    slam.close(); plain.close()
