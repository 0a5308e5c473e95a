"""-m gpu parity at the BASELINE.json configurations the round-1 suite did not reach ("next round" item 1):

  configs[1]  16 x 1800 scans vs a 200k-point map: Seam B bit-exact for all 28 800 queries, full registrations;
  SURVEY 8(d) the 32 seeded scans of the trajectory on `small` and on `vlp16_200k`: iteration counts, 7 + 9 bin histograms,
              termination codes, poses (tolerance of record 1e-4 m / 1e-4 rad; asserted at 1e-8);
  configs[4]  64 hypotheses per scan with the SURVEY 8(d) seeds (+-0.5 m / +-5 deg, seed 5000 + 64 i + h) at full size,
              a sample of them against the oracle;
plus the host-side additions of this round (staged scans, stats flags, resolution change without re-voxelising)."""
import os

import numpy as np
import pytest

from helpers import pose_close
from superodom_amd import synth

pytestmark = pytest.mark.gpu
TOL_T, TOL_R = 1e-4, 1e-4  # north_star


def _setup(scene_name, oracle, make, **cfg):
    sc = synth.Scene(scene_name)
    slam = make(plane_res=sc.plane_res, line_res=sc.plane_res / 2, max_surface_features=-1, **cfg)
    assert slam.add_surf_point_cloud(sc.map_points) == len(sc.map_points) == slam.map_size()
    om = oracle.OracleMap(plane_res=sc.plane_res)
    assert om.add_surf(slam.export_map(), raw=True) == slam.map_size()
    return sc, slam, om


def _assert_registration_equal(st, ost, pose, opose, tag):
    assert st.n_iterations == ost.n_iterations, (tag, "outer iteration counts must agree before poses are compared")
    for it in range(st.n_iterations):
        a, b = st.iterations[it], ost.iters[it]
        assert (a.lm_iterations, a.num_successful_steps, a.termination) == (b.lm_iterations, b.num_successful_steps, b.termination), (tag, it)
        assert a.num_surf_from_scan == b.num_surf, (tag, it)
        assert list(a.reject_hist) == list(b.reject_hist), (tag, it)
        assert list(a.obs_hist) == list(b.obs_hist), (tag, it)
        assert abs(a.final_cost - b.final_cost) <= 1e-9 * max(1.0, abs(b.final_cost)), (tag, it)
    ok, dt, dr = pose_close(pose, opose, TOL_T, TOL_R)
    assert ok, f"{tag}: pose parity violated: dt={dt:.3e} m dr={dr:.3e} rad"
    assert dt < 1e-8 and dr < 1e-8, f"{tag}: expected near machine agreement, got {dt:.3e} {dr:.3e}"


def test_config1_seam_b_every_query_bit_exact(oracle, gpu_slam_factory):
    """BASELINE configs[1]: the 16 x 1800 = 28 800 world-frame queries of a scan against the 200 000-point map through
    so_icp_knn_surf -- found flags, neighbour coordinates and d2 bit-identical to the CPU restatement for EVERY query."""
    sc, slam, om = _setup("vlp16_200k", oracle, gpu_slam_factory)
    assert slam.map_size() == 200_000
    for i in (0, 13):
        gt = sc.gt_pose(i)
        q = (sc.scan(i).astype(np.float64) @ synth.quat_to_R(gt[3:]).T + gt[:3]).astype(np.float32)
        assert len(q) == 28_800
        found, nbr, d2, idx = slam.nearest_k_search_surf(q, 5)
        ofound, onbr, od2, oidx, _ = om.knn(q, 5, use_grid=1)
        assert np.array_equal(found, ofound)
        f = found.astype(bool)
        assert f.sum() > 28_000
        assert np.array_equal(d2[f].view(np.uint32), od2[f].view(np.uint32)), "d2 must be bit-identical"
        assert np.array_equal(nbr[f], onbr[f]), "same neighbours, same order, same ties"


@pytest.mark.parametrize("scene", ["small", "vlp16_200k"])
def test_all_32_seeded_scans_of_the_trajectory(oracle, gpu_slam_factory, scene):
    """seeds world=1, map-noise=2, scan-noise=3+i, guess=1000+i for the 32 scans of the trajectory."""
    sc, slam, om = _setup(scene, oracle, gpu_slam_factory, max_iterations=5)
    cfg = oracle.default_config(max_iterations=5)
    outer = []
    for i in range(32):
        scan, guess, gt = sc.scan(i), sc.guess(i), sc.gt_pose(i)
        rc, pose, st = slam.register(scan, guess)
        orc, opose, ost, corrs = om.register(scan, guess, cfg, want_corrs=True)
        assert rc == orc == 0, (scene, i)
        _assert_registration_equal(st, ost, pose, opose, (scene, i))
        # per-query MatchingResult of the last outer iteration: the accepted sets are the SAME set (Jaccard 1), not just equally large
        assert np.array_equal(slam.match_status(len(scan)), corrs["status"].astype(np.uint8)), (scene, i)
        e = synth.pose_error(pose, gt)
        assert e[0] < 0.05 and e[1] < 0.02, (scene, i, e)
        outer.append(st.n_iterations)
    assert min(outer) >= 1 and max(outer) <= 5


def test_batch64_full_size_with_the_survey_seeds(oracle, gpu_slam_factory):
    """BASELINE configs[4] / SURVEY 8(d) "batched": 64 guesses per scan, dt ~ U(-0.5, 0.5) m, dtheta ~ U(-5, 5) deg per axis,
    seeds 5000 + 64 i + h, at the full size (131 072-point scan, 2M-point map).  Every hypothesis returns; a sample of them
    (the oracle needs ~1 s per full-size hypothesis on all host cores) is compared with the oracle in full."""
    sc, slam, om = _setup("os1_128_2m", oracle, gpu_slam_factory, max_iterations=5)
    i = 1
    scan = sc.scan(i)
    poses = np.stack([synth.perturb_pose(sc.gt_pose(i), 5000 + 64 * i + h, 0.5, 5.0) for h in range(64)])
    d, n = slam.upload_scan(scan)
    ok, rcs, out, sts = slam.register_batch(None, poses, d_scan=d, n=n)
    assert ok == 64 and (rcs == 0).all()
    near = sum(1 for h in range(64) if synth.pose_error(out[h], sc.gt_pose(i))[0] < 0.02)
    assert near >= 48, f"only {near} of 64 hypotheses reached the ground truth"
    oracle.set_num_threads(max(1, (os.cpu_count() or 2) // 2))
    try:
        for h in (0, 21, 42, 63):
            orc, opose, ost, _ = om.register(scan, poses[h], oracle.default_config(max_iterations=5))
            assert orc == 0
            _assert_registration_equal(sts[h], ost, out[h], opose, ("batch64", h))
    finally:
        oracle.set_num_threads(1)
    # the batch is B independent registrations: EVERY hypothesis alone gives the same bits (pose, final normal equations,
    # every per-iteration statistic) -- the batched kernels reproduce the single registration's summation tree
    for h in range(64):
        rc, ph, sh = slam.register_dev(d, n, poses[h])
        assert rc == 0 and np.array_equal(ph, out[h]), h
        _assert_same_bits(sts[h], sh, ("batch64 vs single", h))


def _assert_same_bits(a, b, tag):
    assert a.n_iterations == b.n_iterations, tag
    assert np.array_equal(np.array(a.JtJ), np.array(b.JtJ)) and np.array_equal(np.array(a.Jtr), np.array(b.Jtr)), tag
    for it in range(a.n_iterations):
        x, y = a.iterations[it], b.iterations[it]
        assert (x.lm_iterations, x.num_successful_steps, x.termination, x.num_surf_from_scan) == \
               (y.lm_iterations, y.num_successful_steps, y.termination, y.num_surf_from_scan), (tag, it)
        assert list(x.reject_hist) == list(y.reject_hist) and list(x.obs_hist) == list(y.obs_hist), (tag, it)
        assert (x.initial_cost, x.final_cost) == (y.initial_cost, y.final_cost), (tag, it)
        assert np.array_equal(np.array(x.pose_after), np.array(y.pose_after)), (tag, it)


@pytest.mark.parametrize("scene,n_hyp,sub", [("tiny", 5, None), ("tiny", 70, None), ("small", 33, None), ("small", 9, 1500)])
def test_batched_hypotheses_equal_single_registrations_bit_for_bit(oracle, gpu_slam_factory, scene, n_hyp, sub):
    """so_icp_register_batch on the batched kernels: odd hypothesis counts (workgroups per hypothesis that do not divide the
    grid they stand in for), more hypotheses than one group holds (70 > 64), scans whose single-registration grid is small
    (tiny: 16 workgroups), a sub-sampled scan (max_surface_features), hypotheses that converge in different rounds and
    hypotheses that start far away.  Every hypothesis must equal its single registration bit for bit; a sample is compared
    with the oracle in full."""
    sc, slam, om = _setup(scene, oracle, gpu_slam_factory, max_iterations=5)
    if sub:
        slam.set_max_surface_features(sub)
    i = 2
    scan = sc.scan(i)
    poses = np.stack([synth.perturb_pose(sc.gt_pose(i), 7000 + 64 * i + h, 0.02 + 0.5 * (h % 7) / 6.0, 0.2 + 5.0 * (h % 5) / 4.0) for h in range(n_hyp)])
    d, n = slam.upload_scan(scan)
    ok, rcs, out, sts = slam.register_batch(None, poses, d_scan=d, n=n)
    assert ok == n_hyp and (rcs == 0).all()
    outer = set()
    for h in range(n_hyp):
        rc, ph, sh = slam.register_dev(d, n, poses[h])
        assert rc == 0 and np.array_equal(ph, out[h]), (scene, h)
        _assert_same_bits(sts[h], sh, (scene, h))
        outer.add(sh.n_iterations)
    if n_hyp >= 33:
        assert len(outer) >= 2, "the hypotheses should leave the rounds at different times"
    cfg = oracle.default_config(max_iterations=5, max_surface_features=sub if sub else -1)
    for h in (0, n_hyp // 2, n_hyp - 1):
        orc, opose, ost, _ = om.register(scan, poses[h], cfg)
        assert orc == 0
        _assert_registration_equal(sts[h], ost, out[h], opose, (scene, "batch", h))
    # a host scan buffer instead of a resident one; and a batch of one
    ok2, _, out2, _ = slam.register_batch(scan, poses[:3])
    assert ok2 == 3 and np.array_equal(out2, out[:3])
    ok1, _, out1, _ = slam.register_batch(None, poses[4:5], d_scan=d, n=n)
    assert ok1 == 1 and np.array_equal(out1[0], out[4])


def test_batch_with_hypotheses_that_fail_and_degenerate_inputs(oracle, gpu_slam_factory, soicp):
    """Hypotheses that leave the rounds for other reasons than convergence: a guess lifted 8 m above the scene (same map
    window, a fraction of the matches, no convergence to the truth), a guess turned by 90 degrees (matches, but few),
    duplicates of one guess, next to ordinary ones -- each must carry the status, the pose and the statistics of its single
    registration.  Then an EMPTY map (every hypothesis SO_ICP_NOT_ENOUGH_MAP_FEATURES, pose = guess, LS.cpp:113-116), a
    batch of zero hypotheses and an empty scan."""
    sc, slam, _ = _setup("small", oracle, gpu_slam_factory, max_iterations=5)
    i = 4
    scan = sc.scan(i)
    gt = sc.gt_pose(i)
    far = gt.copy(); far[2] += 8.0
    turned = gt.copy()
    turned[3:] = synth.quat_mul(gt[3:], synth.quat_from_rotvec(np.array([0.0, 0.0, np.pi / 2])))
    near = [synth.perturb_pose(gt, 8100 + h, 0.1, 1.0) for h in range(3)]
    poses = np.stack([near[0], far, near[1], turned, near[0], far, near[2]])
    d, n = slam.upload_scan(scan)
    ok, rcs, out, sts = slam.register_batch(None, poses, d_scan=d, n=n)
    singles = [slam.register_dev(d, n, p) for p in poses]
    assert [int(r) for r in rcs] == [r[0] for r in singles]
    assert sts[1].iterations[0].num_surf_from_scan < sts[0].iterations[0].num_surf_from_scan / 2, "far fewer matches from 8 m above"
    assert ok == sum(1 for r in singles if r[0] == 0) >= 4
    for h, (rc, ph, sh) in enumerate(singles):
        assert np.array_equal(out[h], ph), h
        _assert_same_bits(sts[h], sh, ("mixed batch", h))
    assert np.array_equal(out[0], out[4])
    # empty map
    empty = gpu_slam_factory(plane_res=sc.plane_res, line_res=sc.plane_res / 2, max_surface_features=-1, max_iterations=5)
    ok, rcs, out, _ = empty.register_batch(scan, poses)
    assert ok == 0 and (rcs == soicp.NOT_ENOUGH_MAP_FEATURES).all() and np.array_equal(out, poses)
    empty.close()
    # no hypotheses / no points
    ok, rcs, out, _ = slam.register_batch(None, np.zeros((0, 7)), d_scan=d, n=n)
    assert ok == 0 and len(rcs) == 0
    none = np.zeros((0, 3), np.float32)
    ok, rcs, out, _ = slam.register_batch(none, poses[:3])
    assert [int(r) for r in rcs] == [slam.register(none, p)[0] for p in poses[:3]] and ok == int((rcs == 0).sum())


@pytest.mark.parametrize("env", [{"SOICP_BATCH_MODE": "lanes"}, {"SOICP_BATCH_MODE": "one_per_cu"}, {"SOICP_BATCH_CHAIN": "0"}])
def test_batch_fallback_paths_give_the_same_bits(oracle, gpu_slam_factory, monkeypatch, env):
    """The degraded forms of so_icp_register_batch -- one solve workgroup per compute unit, and concurrent sequential
    registrations on worker contexts (what a device that cannot keep the batched solve resident falls back to) -- return the
    poses of the batched kernels bit for bit; so do the rounds with a report + synchronisation after every one of them
    (SOICP_BATCH_CHAIN=0) against the chained rounds, whose lists go stale (second batch of a context: chained by the
    survivor counts of the first)."""
    sc, slam, _ = _setup("small", oracle, gpu_slam_factory, max_iterations=5)
    scan = sc.scan(6)
    poses = np.stack([synth.perturb_pose(sc.gt_pose(6), 9000 + h, 0.05 + 0.4 * (h % 5) / 4.0, 0.5 + 4.0 * (h % 3) / 2.0) for h in range(19)])
    ok, rcs, out, sts = slam.register_batch(scan, poses)
    assert ok == 19
    ok1, _, out1, sts1 = slam.register_batch(scan, poses)  # (chained further, by what the first batch saw)
    assert ok1 == 19 and np.array_equal(out1, out)
    for h in range(19):
        _assert_same_bits(sts[h], sts1[h], ("second batch", h))
    for k, v in env.items():
        monkeypatch.setenv(k, v)  # read by so_icp_create
    _, alt, _ = _setup("small", oracle, gpu_slam_factory, max_iterations=5)
    ok2, rcs2, out2, sts2 = alt.register_batch(scan, poses)
    assert ok2 == 19 and np.array_equal(out2, out)
    for h in range(19):
        _assert_same_bits(sts[h], sts2[h], (env, h))


@pytest.mark.parametrize("buffers", ["pageable", "registered", "pinned"])
def test_staged_scan_is_the_same_registration(oracle, gpu_slam_factory, soicp, buffers):
    """so_icp_stage_scan + so_icp_register: the staged upload -- through the copy thread (pageable host memory), or by DMA straight
    from the caller's buffer (so_icp_host_register / so_icp_host_alloc), enqueued by the registration in flight -- feeds the same
    kernels: bit-identical results, the stats say which path ran; a staged scan that is never consumed and a re-staged one are
    handled; so_icp_stage_cancel withdraws a copy."""
    sc, slam, om = _setup("small", oracle, gpu_slam_factory, max_iterations=5)
    scans = [np.ascontiguousarray(sc.scan(i), dtype=np.float32) for i in range(4)]
    if buffers == "registered":
        for s_ in scans:
            slam.host_register(s_)
    elif buffers == "pinned":
        scans = [slam.host_alloc_like(s_) for s_ in scans]
    ref = [slam.register(scans[i], sc.guess(i)) for i in range(4)]
    assert all(not (r[2].flags & soicp.FLAG_STAGED_SCAN) for r in ref)
    slam.stage_scan(scans[0])
    for i in range(4):  # the pattern of the bench loop: announce i + 1, register i
        if i + 1 < 4:
            slam.stage_scan(scans[i + 1])
        rc, pose, st = slam.register(scans[i], sc.guess(i))
        assert rc == 0 and (st.flags & soicp.FLAG_STAGED_SCAN)
        assert np.array_equal(pose, ref[i][1]) and np.array_equal(np.array(st.JtJ), np.array(ref[i][2].JtJ))
    slam.stage_scan(scans[2])            # staged, then a DIFFERENT scan is registered: plain upload, staged copy ignored
    rc, pose, st = slam.register(scans[3], sc.guess(3))
    assert rc == 0 and not (st.flags & soicp.FLAG_STAGED_SCAN) and np.array_equal(pose, ref[3][1])
    rc, pose, st = slam.register(scans[2], sc.guess(2))  # still there
    assert rc == 0 and (st.flags & soicp.FLAG_STAGED_SCAN) and np.array_equal(pose, ref[2][1])
    tm = slam.timing()
    assert (tm.staged_direct > 0) == (buffers != "pageable") and (tm.staged_copied > 0) == (buffers == "pageable"), (tm.staged_direct, tm.staged_copied)
    slam.stage_scan(scans[1])            # announced, then withdrawn: the buffer is the caller's again, a later register uploads it
    slam.stage_cancel(scans[1])
    keep = scans[1].copy()
    scans[1][:] = scans[0]
    rc, pose, st = slam.register(scans[1], sc.guess(0))
    assert rc == 0 and not (st.flags & soicp.FLAG_STAGED_SCAN) and np.array_equal(pose, ref[0][1])
    scans[1][:] = keep
    # a skipped frame: 1 is announced, then 2 is announced and registered; a later register(1) is not served the old copy
    slam.stage_scan(scans[1]); slam.stage_scan(scans[2])
    rc, pose, st = slam.register(scans[2], sc.guess(2))
    assert rc == 0 and (st.flags & soicp.FLAG_STAGED_SCAN) and np.array_equal(pose, ref[2][1])
    rc, pose, st = slam.register(scans[1], sc.guess(1))  # (its copy was announced before a scan consumed since: dropped, uploaded afresh)
    assert rc == 0 and not (st.flags & soicp.FLAG_STAGED_SCAN) and np.array_equal(pose, ref[1][1])
    # a feeder THREAD (the node's feature callback) announces scans while this thread registers: it may run ahead by more than
    # one scan (older announcements are dropped, the registration uploads those itself) but never touches the slot in use
    import threading
    import time
    order = [k % 4 for k in range(24)]

    def feeder():
        for k in order:
            slam.stage_scan(scans[k])
            time.sleep(0.0002)
    th = threading.Thread(target=feeder)
    th.start()
    staged = 0
    for k in order:
        rc, pose, st = slam.register(scans[k], sc.guess(k))
        assert rc == 0 and np.array_equal(pose, ref[k][1]) and np.array_equal(np.array(st.JtJ), np.array(ref[k][2].JtJ)), k
        staged += 1 if (st.flags & soicp.FLAG_STAGED_SCAN) else 0
    th.join(30)
    assert staged >= 1
    # Localization() consumes a staged scan too and inserts it from the staged copy
    twin = gpu_slam_factory(plane_res=sc.plane_res, line_res=sc.plane_res / 2, max_surface_features=-1, max_iterations=5)
    twin.add_surf_point_cloud(sc.map_points)
    slam2 = gpu_slam_factory(plane_res=sc.plane_res, line_res=sc.plane_res / 2, max_surface_features=-1, max_iterations=5)
    slam2.add_surf_point_cloud(sc.map_points)
    slam2.stage_scan(scans[1])
    rc_a, pa, sa = slam2.localization(True, sc.guess(1), scans[1], 0.1)
    rc_b, pb, sb = twin.localization(True, sc.guess(1), scans[1], 0.1)
    assert rc_a == rc_b == 0 and np.array_equal(pa, pb) and (sa.flags & soicp.FLAG_STAGED_SCAN) and not (sb.flags & soicp.FLAG_STAGED_SCAN)
    assert slam2.map_size() == twin.map_size() and np.array_equal(slam2.export_map(), twin.export_map())


def test_stats_flags_name_the_degraded_modes(oracle, gpu_slam_factory, soicp, monkeypatch):
    sc, slam, _ = _setup("tiny", oracle, gpu_slam_factory, max_iterations=3)
    scan, guess = sc.scan(0), sc.guess(0)
    rc, pose0, st = slam.register(scan, guess)
    QW = soicp.FLAG_QUERY_WAVES  # (a 4 096-point scan: one wavefront per query, no binning -- not a degraded mode, tests/test_gpu_query_waves.py)
    assert rc == 0 and st.flags == QW, hex(st.flags)
    for env, flag in (({"SOICP_PERSISTENT": "0"}, soicp.FLAG_PER_EVAL_LAUNCHES),
                      ({"SOICP_HOST_MAP": "1"}, soicp.FLAG_HOST_MAP), ({"SOICP_READBACK": "copy"}, soicp.FLAG_COPY_READBACK)):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        _, alt, _ = _setup("tiny", oracle, gpu_slam_factory, max_iterations=3)
        rc, pose2, s2 = alt.register(scan, guess)
        assert rc == 0 and s2.flags == (flag | QW), (env, hex(s2.flags))
        assert np.array_equal(pose2, pose0) and np.array_equal(np.array(s2.JtJ), np.array(st.JtJ)), (env, "a degraded mode is the same registration, bit for bit")
        for k in env:
            monkeypatch.delenv(k)
    # an abandoned persistent solve (test hook): the repeated registration is flagged, later ones carry the degraded mode
    monkeypatch.setenv("SOICP_ABLATE", "8192")
    _, alt, _ = _setup("tiny", oracle, gpu_slam_factory, max_iterations=3)
    rc, _, s3 = alt.register(scan, guess)
    assert rc == 0 and s3.flags == (soicp.FLAG_RETRIED | soicp.FLAG_PER_EVAL_LAUNCHES), hex(s3.flags)
    rc, _, s4 = alt.register(scan, guess)
    assert rc == 0 and s4.flags == soicp.FLAG_PER_EVAL_LAUNCHES, hex(s4.flags)
    sh = gpu_slam_factory(plane_res=sc.plane_res, max_surface_features=-1, max_iterations=1, rank=0, world_size=2)
    sh.add_surf_point_cloud(sc.map_points)
    rc, _, s5 = sh.register(scan, guess)
    assert rc == 0 and (s5.flags & soicp.FLAG_SHARDED) and not (s5.flags & soicp.FLAG_HOST_MAP)  # the shard is device-resident too


def test_resolution_change_keeps_the_points_until_their_cube_is_touched(oracle, gpu_slam_factory):
    """localMap.planeRes_ is pushed every frame (laserMapping.cpp:648-649) and auto_voxel_size can flip it: the reference
    keeps the points of every block and re-filters a block only when the next insert touches it (LocalMap.h:617-641).
    The device map does the same: a resolution change rebuilds the cell tables over the resident points."""
    sc = synth.Scene("tiny")
    slam = gpu_slam_factory(plane_res=0.2, line_res=0.1, max_surface_features=-1, max_iterations=3)
    om = oracle.OracleMap(plane_res=0.2)
    slam.add_surf_point_cloud(sc.map_points); om.add_surf(sc.map_points)
    before = slam.export_map()
    for res in (0.4, 0.2, 0.8):
        slam.set_resolution(res / 2, res); om.set_resolution(res / 2, res)
        after = slam.export_map()
        assert len(after) == len(before) and np.array_equal(after[np.lexsort(after.T)], before[np.lexsort(before.T)]), "no point may move"
        # Seam B over the rebuilt tables (new cell size) against the oracle holding the same points in the same order
        om2 = oracle.OracleMap(plane_res=res); om2.add_surf(after, raw=True)
        gt = sc.gt_pose(0)
        q = (sc.scan(0) @ synth.quat_to_R(gt[3:]).T + gt[:3]).astype(np.float32)[::5]
        found, nbr, d2, idx = slam.nearest_k_search_surf(q, 5)
        of, on, od = om2.knn(q, 5)[:3]
        assert np.array_equal(found, of)
        f = found.astype(bool)
        assert np.array_equal(d2[f].view(np.uint32), od[f].view(np.uint32)) and np.array_equal(nbr[f], on[f])
        # a registration at the new resolution
        rc, pose, st = slam.register(sc.scan(1), sc.guess(1))
        orc, opose, ost, _ = om2.register(sc.scan(1), sc.guess(1), oracle.default_config(max_iterations=3))
        assert rc == orc
        if rc == 0:
            _assert_registration_equal(st, ost, pose, opose, ("resolution", res))
    # the next insert re-filters the touched cubes at the resolution in effect (0.8): product and oracle agree point for point
    gt = sc.gt_pose(2)
    w = (sc.scan(2) @ synth.quat_to_R(gt[3:]).T + gt[:3]).astype(np.float32)
    assert slam.add_surf_point_cloud(w) == om.add_surf(w)
    a, b = slam.export_map(), om.export()
    # Several OLD points now share one 0.8 m leaf: their centroid is a float sum in the block's storage order -- the output
    # order of the cube's previous VoxelGrid (ascending leaf index of the grid it was last filtered on).  The product keeps
    # its pool in (cell, leaf) order and restores that order for the re-filter (map_kernels.hip: old_order_key_kernel), so
    # the merged centroids equal the oracle's to the bit.  (Inside one leaf PCL's std::sort leaves the order of the points
    # unspecified; the oracle -- the contract -- sums in input order)
    assert len(a) == len(b)
    assert np.array_equal(a[np.lexsort(a.T)].view(np.uint32), b[np.lexsort(b.T)].view(np.uint32))


def test_plane_res_change_recuts_the_shards(oracle, gpu_slam_factory, soicp):
    """The shards are cut along the cell grid, which follows planeRes (tools/soak_shards.py saw wrong neighbours after a
    change).  so_icp_set_resolution over a sharded, non-empty map is a collective step: every rank hands out the points it
    owns, all gather all, each re-cuts its shard on the new grid.  Afterwards: every resident point is a point of the
    unsharded map, the union is the whole map, the full-map counts are right, registrations and the next insert (which
    re-filters what it touches on the new leaf grid) follow the single context.  Without a communicator: refused, loudly."""
    sc = synth.Scene("small")
    mk = dict(plane_res=0.2, line_res=0.1, max_surface_features=-1, max_iterations=4)
    lone = gpu_slam_factory(rank=0, world_size=2, **mk)
    lone.set_resolution(0.2, 0.4); lone.set_resolution(0.1, 0.2)   # empty map: nothing to re-cut
    lone.add_surf_point_cloud(sc.map_points)
    lone.set_resolution(0.1, 0.2)                                  # unchanged (the node pushes it every frame, lmap.cpp:648-649)
    with pytest.raises(Exception, match="communicator"):
        lone.set_resolution(0.2, 0.4)
    lone.close()
    one = gpu_slam_factory(**mk)
    shards = [gpu_slam_factory(rank=r, world_size=3, **mk) for r in range(3)]
    for sh in shards:
        sh.comm_init_inprocess(0x4E5)
    one.add_surf_point_cloud(sc.map_points)
    _in_threads([lambda sh=sh: sh.add_surf_point_cloud(sc.map_points) for sh in shards])
    key = lambda a: {tuple(v) for v in a.view(np.uint32).reshape(-1, 3).tolist()}
    for res in (0.4, 0.2, 0.4):
        one.set_resolution(res / 2, res)
        _in_threads([lambda sh=sh: (sh.set_resolution(res / 2, res), True)[1] for sh in shards])
        full = one.export_map(); kfull = key(full); union = set()
        for sh in shards:
            total, mine = sh.map_size(this_rank=True)
            part = sh.export_map()
            assert total == len(full) and len(part) == mine < len(full)
            assert key(part) <= kfull
            union |= key(part)
        assert union == kfull
        for i in (2, 9):
            scan, guess = sc.scan(i), sc.guess(i, dt=0.3, dth_deg=2.0)
            rc, pose, st = one.register(scan, guess)
            out = _in_threads([lambda sh=sh: sh.register(scan, guess) for sh in shards])
            assert all(r[0] == rc == 0 for r in out) and all(np.array_equal(out[0][1], r[1]) for r in out)
            for r in out:
                assert r[2].n_iterations == st.n_iterations
                for it in range(st.n_iterations):
                    a, b = r[2].iterations[it], st.iterations[it]
                    assert (a.lm_iterations, a.termination, a.num_surf_from_scan) == (b.lm_iterations, b.termination, b.num_surf_from_scan), (res, i, it)
                    assert list(a.reject_hist) == list(b.reject_hist) and list(a.obs_hist) == list(b.obs_hist)
                ok, dt, dr = pose_close(r[1], pose, 1e-9, 1e-9)
                assert ok, (res, i, dt, dr)
        # an insert on the new grid: the touched cubes are re-filtered, shard by shard, like the whole map
        gt = sc.gt_pose(4)
        w = (sc.scan(4) @ synth.quat_to_R(gt[3:]).T + gt[:3]).astype(np.float32)
        n1 = one.add_surf_point_cloud(w)
        assert [r for r in _in_threads([lambda sh=sh: sh.add_surf_point_cloud(w) for sh in shards])] == [n1] * 3
        full = one.export_map(); kfull = key(full); union = set()
        for sh in shards:
            part = sh.export_map()
            assert key(part) <= kfull, "every resident centroid is a centroid of the unsharded map, bit for bit"
            union |= key(part)
        assert union == kfull and shards[0].map_size() == len(full)
    for sh in shards:
        sh.close()


def test_two_shard_ranks_follow_the_queries_across_outer_iterations(oracle, gpu_slam_factory):
    """N = 2 with max_iterations = 5 and a 3 degree / 0.4 m initial error: between outer iterations the pose update moves far
    queries by more than a map cell (1.5 m at 30 m), i.e. out of the one-cell halo of the shard that owned them under the
    initial pose.  Ownership is therefore re-derived under the current pose at the start of every outer iteration
    (ADVICE r01, high).  Two shard contexts on this one GPU, each driven from its own thread and joined by an in-process
    group (the sums take the place of the RCCL all-reduce): every iteration's histograms, iteration counts and termination
    codes must equal the single-context registration and the oracle; the poses agree to the last bits."""
    import threading
    sc, full, om = _setup("small", oracle, gpu_slam_factory, max_iterations=5)
    shards = []
    key = 0x5151
    for rank in (0, 1):
        sh = gpu_slam_factory(plane_res=sc.plane_res, line_res=sc.plane_res / 2, max_surface_features=-1, max_iterations=5, rank=rank, world_size=2)
        sh.add_surf_point_cloud(sc.map_points)
        sh.comm_init_inprocess(key)
        shards.append(sh)
    for i, (dt_, dth) in ((3, (0.4, 3.0)), (9, (0.1, 1.0))):
        scan, guess = sc.scan(i), sc.guess(i, dt=dt_, dth_deg=dth)
        rc, pose, st = full.register(scan, guess)
        orc, opose, ost, _ = om.register(scan, guess, oracle.default_config(max_iterations=5))
        assert rc == orc == 0
        _assert_registration_equal(st, ost, pose, opose, ("single context", i))
        if i == 3:
            assert st.n_iterations >= 3, "the test needs several outer iterations"
        res = [None, None]

        def run(r):
            res[r] = shards[r].register(scan, guess)
        th = [threading.Thread(target=run, args=(r,)) for r in (0, 1)]
        for t in th:
            t.start()
        for t in th:
            t.join(120)
        assert all(r is not None and r[0] == 0 for r in res), [None if r is None else r[0] for r in res]
        assert np.array_equal(res[0][1], res[1][1]), "both ranks must take identical decisions on identical sums"
        for r in (0, 1):
            s2 = res[r][2]
            assert s2.n_iterations == st.n_iterations
            for it in range(st.n_iterations):
                a, b = s2.iterations[it], st.iterations[it]
                assert (a.lm_iterations, a.num_successful_steps, a.termination, a.num_surf_from_scan) == \
                       (b.lm_iterations, b.num_successful_steps, b.termination, b.num_surf_from_scan), (r, it)
                assert list(a.reject_hist) == list(b.reject_hist) and list(a.obs_hist) == list(b.obs_hist), (r, it)
            ok, dt, dr = pose_close(res[r][1], pose, 1e-9, 1e-9)
            assert ok, (r, dt, dr)


def _in_threads(fns):
    import threading
    res = [None] * len(fns)

    def run(k):
        res[k] = fns[k]()
    th = [threading.Thread(target=run, args=(k,)) for k in range(len(fns))]
    for t in th:
        t.start()
    for t in th:
        t.join(180)
    assert all(r is not None for r in res), "a shard rank did not return"
    return res


def test_sharded_localization_keeps_the_shard_resident_and_exact(oracle, gpu_slam_factory):
    """Localization() over consecutive scans with the map SHARDED over two ranks (both on this GPU, in-process group): each
    rank keeps its shard in HBM and inserts every scan on the device -- whole VoxelGrid leaves, so every resident centroid
    is bit-identical to the single-context map's, the shards' union is the whole map, the per-cube counts of the full map
    come out of one small collective per insert, and the poses follow the single-context run."""
    sc = synth.Scene("tiny")
    mk = dict(plane_res=sc.plane_res, line_res=sc.plane_res / 2, max_surface_features=-1, max_iterations=4)
    one = gpu_slam_factory(**mk)
    shards = [gpu_slam_factory(rank=r, world_size=2, **mk) for r in (0, 1)]
    for sh in shards:
        sh.comm_init_inprocess(0x7A7A)
    T0 = sc.gt_pose(0)
    assert one.localization(False, T0, sc.scan(0), 0.0)[0] == 2
    assert [r[0] for r in _in_threads([lambda sh=sh: sh.localization(False, T0, sc.scan(0), 0.0) for sh in shards])] == [2, 2]
    n_ok = 0
    for i in range(1, 5):
        scan, guess = sc.scan(i), sc.guess(i)
        rc, pose, st = one.localization(True, guess, scan, 0.1 * i)
        res = _in_threads([lambda sh=sh: sh.localization(True, guess, scan, 0.1 * i) for sh in shards])
        assert [r[0] for r in res] == [rc, rc]
        assert np.array_equal(res[0][1], res[1][1])
        if rc == 0:
            n_ok += 1
            for r in res:
                assert r[2].n_iterations == st.n_iterations and r[2].laser_cloud_surf_from_map_num == st.laser_cloud_surf_from_map_num
                for it in range(st.n_iterations):
                    assert list(r[2].iterations[it].reject_hist) == list(st.iterations[it].reject_hist)
                    assert list(r[2].iterations[it].obs_hist) == list(st.iterations[it].obs_hist)
                ok, dt, dr = pose_close(r[1], pose, 1e-9, 1e-9)
                assert ok, (i, dt, dr)
        full = one.export_map()
        key = lambda a: {tuple(v) for v in a.view(np.uint32).reshape(-1, 3).tolist()}
        kfull = key(full)
        union = set()
        for sh in shards:
            total, mine = sh.map_size(this_rank=True)
            assert total == len(full), "full-map count from the per-insert collective"
            part = sh.export_map()
            assert len(part) == mine < len(full)
            kp = key(part)
            assert kp <= kfull, "every resident centroid is a centroid of the unsharded map, bit for bit"
            union |= kp
        assert union == kfull, "the shards cover the map"
    assert n_ok >= 3
