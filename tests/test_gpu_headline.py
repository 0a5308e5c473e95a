"""-m gpu parity at the CONFIGURATION OF RECORD (BASELINE.json configs[2]: 131 072-point OS1-128 scan vs the 2M-point map),
closing the gaps VERDICT r02 names under "Parity is partial" (ii)-(iv):

  (a) all 32 seeded scans of the SURVEY 8(d) trajectory against Oracle-A at full size: outer / LM iteration counts,
      termination codes, 7 + 9 bin histograms, the per-query MatchingResult of the last outer iteration (accepted-set
      Jaccard 1.0), poses (tolerance of record 1e-4 m / 1e-4 rad; asserted at 1e-8)      [LidarSlam.cpp:107-152, 323-344]
  (b) the same scene through TWO shard ranks (in-process group, both on this GPU) against the single context
  (c) Seam B with the oracle's map loaded in a SHUFFLED order: the canonical index order the product exports is then not
      shared with the checker -- d2 must still be bit-identical and the neighbour sets equal, ties at the k-th distance aside
      [LocalMap.h:481-525, octree.h:93-102]"""
import os
import threading

import numpy as np
import pytest

from helpers import pose_close
from superodom_amd import synth

pytestmark = pytest.mark.gpu
TOL_T, TOL_R = 1e-4, 1e-4  # north_star


@pytest.fixture(scope="module")
def headline(oracle, gpu_slam_factory):
    sc = synth.Scene("os1_128_2m")
    slam = gpu_slam_factory(plane_res=sc.plane_res, line_res=sc.plane_res / 2, max_surface_features=-1, max_iterations=5)
    assert slam.add_surf_point_cloud(sc.map_points) == 2_000_000 == slam.map_size()
    exported = slam.export_map()
    om = oracle.OracleMap(plane_res=sc.plane_res)
    assert om.add_surf(exported, raw=True) == len(exported)
    om.ensure_grids()
    return sc, slam, om, exported


def _equal_stats(st, ost, tag):
    assert st.n_iterations == ost.n_iterations, (tag, st.n_iterations, ost.n_iterations)
    for it in range(st.n_iterations):
        a, b = st.iterations[it], ost.iters[it]
        assert (a.lm_iterations, a.num_successful_steps, a.termination) == (b.lm_iterations, b.num_successful_steps, b.termination), (tag, it)
        assert a.num_surf_from_scan == b.num_surf, (tag, it)
        assert list(a.reject_hist) == list(b.reject_hist), (tag, it)
        assert list(a.obs_hist) == list(b.obs_hist), (tag, it)
        assert abs(a.final_cost - b.final_cost) <= 1e-9 * max(1.0, abs(b.final_cost)), (tag, it)


def test_all_32_headline_scans_against_oracle_a(oracle, headline):
    sc, slam, om, _ = headline
    cfg = oracle.default_config(max_iterations=5)
    oracle.set_num_threads(max(1, (os.cpu_count() or 2) // 2))  # Oracle-A with OpenMP over the queries: same arithmetic, fixed-order sums
    worst = [0.0, 0.0]
    outer, lm = [], []
    try:
        for i in range(32):
            scan, guess, gt = sc.scan(i), sc.guess(i), sc.gt_pose(i)
            assert len(scan) == 131072
            rc, pose, st = slam.register(scan, guess)
            orc, opose, ost, corrs = om.register(scan, guess, cfg, want_corrs=True)
            assert rc == orc == 0, i
            _equal_stats(st, ost, ("os1_128_2m", i))
            status = slam.match_status(len(scan))
            ostatus = corrs["status"].astype(np.uint8)
            acc, oacc = status == 0, ostatus == 0
            union = int((acc | oacc).sum())
            assert union > 30_000 and int((acc & oacc).sum()) == union, ("accepted-set Jaccard must be 1.0", i)
            assert np.array_equal(status, ostatus), ("per-query MatchingResult", i)
            ok, dt, dr = pose_close(pose, opose, TOL_T, TOL_R)
            assert ok and dt < 1e-8 and dr < 1e-8, (i, dt, dr)
            worst = [max(worst[0], dt), max(worst[1], dr)]
            e = synth.pose_error(pose, gt)
            assert e[0] < 0.02 and e[1] < 0.004, (i, e)
            outer.append(st.n_iterations); lm.append(sum(st.iterations[k].lm_iterations for k in range(st.n_iterations)))
    finally:
        oracle.set_num_threads(1)
    assert 1 <= min(outer) and max(outer) <= 5
    print(f"32 headline scans: worst pose delta vs Oracle-A {worst[0]:.2e} m / {worst[1]:.2e} rad, outer {min(outer)}..{max(outer)}, LM {min(lm)}..{max(lm)}")


def test_two_shard_ranks_on_the_headline_scene(oracle, gpu_slam_factory, headline):
    """configs[3] arithmetic at the size of record: rank 0 / rank 1 of world 2, each with its device-resident shard of the 2M map,
    joined by an in-process group (RCCL refuses two ranks on one device).  Both ranks return the same bits; iteration counts,
    termination codes and histograms equal the single context's; poses agree to 1e-9."""
    sc, full, om, _ = headline
    mk = dict(plane_res=sc.plane_res, line_res=sc.plane_res / 2, max_surface_features=-1, max_iterations=5)
    shards = []
    for r in (0, 1):
        sh = gpu_slam_factory(rank=r, world_size=2, **mk)
        sh.add_surf_point_cloud(sc.map_points)
        sh.comm_init_inprocess(0x0512)
        total, mine = sh.map_size(this_rank=True)
        assert 0 < mine < 2_000_000
        shards.append(sh)
    for i in (0, 17):
        scan, guess = sc.scan(i), sc.guess(i)
        rc, pose, st = full.register(scan, guess)
        assert rc == 0
        res = [None, None]

        def run(r):
            res[r] = shards[r].register(scan, guess)
        th = [threading.Thread(target=run, args=(r,)) for r in (0, 1)]
        for t in th:
            t.start()
        for t in th:
            t.join(300)
        assert all(r is not None and r[0] == 0 for r in res)
        assert np.array_equal(res[0][1], res[1][1]), "identical sums, identical decisions, identical bits on both ranks"
        for r in (0, 1):
            s2 = res[r][2]
            assert s2.n_iterations == st.n_iterations
            for it in range(st.n_iterations):
                a, b = s2.iterations[it], st.iterations[it]
                assert (a.lm_iterations, a.num_successful_steps, a.termination, a.num_surf_from_scan) == \
                       (b.lm_iterations, b.num_successful_steps, b.termination, b.num_surf_from_scan), (i, r, it)
                assert list(a.reject_hist) == list(b.reject_hist) and list(a.obs_hist) == list(b.obs_hist), (i, r, it)
            ok, dt, dr = pose_close(res[r][1], pose, 1e-9, 1e-9)
            assert ok, (i, r, dt, dr)
    for sh in shards:
        sh.close()


def test_seam_b_against_an_oracle_loaded_in_shuffled_order(oracle, headline):
    """The other GPU tests load the oracle with the product's exported map, i.e. in the product's canonical order, so that
    the tie-break "earlier index first" is shared.  Here the oracle holds the same 2M points in a RANDOM order: nothing
    about the product's index order reaches it.  d2 must be bit-identical for every query; the neighbour sets must be
    equal except where the product's k-th distance is tied with a point it left out (then the odd ones out carry exactly
    that distance)."""
    sc, slam, _, exported = headline
    rng = np.random.default_rng(424242)
    om2 = oracle.OracleMap(plane_res=sc.plane_res)
    assert om2.add_surf(exported[rng.permutation(len(exported))], raw=True) == len(exported)
    gt = sc.gt_pose(5)
    q = (sc.scan(5).astype(np.float64) @ synth.quat_to_R(gt[3:]).T + gt[:3]).astype(np.float32)[::6]
    found, nbr, d2, _ = slam.nearest_k_search_surf(q, 5)
    ofound, onbr, od2, _, _ = om2.knn(q, 5, use_grid=1)
    assert np.array_equal(found, ofound)
    f = found.astype(bool)
    assert f.sum() > 20_000
    assert np.array_equal(d2[f].view(np.uint32), od2[f].view(np.uint32)), "d2 bit-identical whatever the storage order"
    a = np.sort(nbr[f].view([("x", "f4"), ("y", "f4"), ("z", "f4")]).reshape(-1, 5), axis=1)
    b = np.sort(onbr[f].view([("x", "f4"), ("y", "f4"), ("z", "f4")]).reshape(-1, 5), axis=1)
    differ = np.nonzero((a != b).any(axis=1))[0]
    assert len(differ) <= 0.002 * f.sum(), f"{len(differ)} queries with different neighbour sets: more than ties can explain"
    qf, nf, of, df = q[f], nbr[f], onbr[f], d2[f]
    for k in differ:  # every point that only one side lists sits at the k-th distance (a tie the two orders broke differently)
        sa = {tuple(p) for p in nf[k].tolist()}; sb = {tuple(p) for p in of[k].tolist()}
        for p in sa ^ sb:
            diff = (qf[k] - np.array(p, np.float32)).astype(np.float32).astype(np.float64)
            dd = np.float32((diff[0] * diff[0] + diff[1] * diff[1]) + diff[2] * diff[2])
            assert dd == df[k, 4], (k, p, dd, df[k])
