"""-m gpu: the hand-over of HARD queries inside the k-NN sweep (round 6; MatchParams::hand_ctr, so_icp_stats::knn_handed_over).

The lanes of a work-list item that its near pass cannot certify (a 5th neighbour beyond half a cell) used to run a second, dependent group
pass in their own wavefront -- the 18 - 22 us wavefronts that ended a sweep whose median wavefront lives 11 us.  Up to SOICP_KNN_HAND (2)
of them per item now go to a ring and are searched by whichever wavefront of the launch claims them (the wave-cooperative exact scan of the
27 cells).  Every variant is an exact 5-NN search (LocalMap.h:481-525, octree.h:93-102; ties by canonical index): status bytes, histograms,
iteration counts, termination codes and the bits of the pose must be IDENTICAL to the sweep that hands nothing over (SOICP_KNN_HAND=0), and
equal to the oracle's."""
import os

import numpy as np
import pytest

from helpers import pose_close
from superodom_amd import synth

pytestmark = pytest.mark.gpu


def _ctx(make, hand, **cfg):
    old = os.environ.get("SOICP_KNN_HAND")
    os.environ["SOICP_KNN_HAND"] = str(hand)  # (read when the context is created)
    try:
        return make(**cfg)
    finally:
        if old is None:
            del os.environ["SOICP_KNN_HAND"]
        else:
            os.environ["SOICP_KNN_HAND"] = old


def _same(sa, sb):
    assert sa.n_iterations == sb.n_iterations
    for it in range(sa.n_iterations):
        x, y = sa.iterations[it], sb.iterations[it]
        assert (x.lm_iterations, x.num_successful_steps, x.termination, x.num_surf_from_scan) == \
               (y.lm_iterations, y.num_successful_steps, y.termination, y.num_surf_from_scan)
        assert list(x.reject_hist) == list(y.reject_hist) and list(x.obs_hist) == list(y.obs_hist)
        assert x.initial_cost == y.initial_cost and x.final_cost == y.final_cost
        assert np.array_equal(np.array(x.pose_after), np.array(y.pose_after))
    assert np.array_equal(np.array(sa.JtJ), np.array(sb.JtJ)) and np.array_equal(np.array(sa.Jtr), np.array(sb.Jtr))


@pytest.mark.parametrize("scene,dt,dth", [("small", 0.10, 1.0), ("small", 0.5, 5.0), ("mid360_like", 0.10, 1.0)])
def test_handed_over_queries_get_the_same_lists(oracle, soicp, gpu_slam_factory, scene, dt, dth):
    sc = synth.Scene(scene)
    cfg = dict(plane_res=sc.plane_res, line_res=sc.plane_res / 2, max_surface_features=-1, max_iterations=5)
    ctxs = {h: _ctx(gpu_slam_factory, h, **cfg) for h in (0, 1, 2, 8, 64)}
    for s in ctxs.values():
        s.add_surf_point_cloud(sc.map_points)
    om = oracle.OracleMap(plane_res=sc.plane_res)
    om.add_surf(ctxs[0].export_map(), raw=True)
    handed = {h: 0 for h in ctxs}
    for i in (0, 3, 6):
        scan = sc.scan(i)
        guess = sc.guess(i) if (dt, dth) == (0.10, 1.0) else synth.perturb_pose(sc.gt_pose(i), 900 + i, dt, dth)
        assert len(scan) > 4096  # (swept in chunks, not by query waves)
        ref = None
        for h, s in ctxs.items():
            rc, pose, st = s.register(scan, guess)
            ms = s.match_status(len(scan)).copy()
            assert rc == 0
            handed[h] += st.knn_handed_over
            if ref is None:
                ref = (pose, st, ms)
                assert st.knn_handed_over == 0
                continue
            assert np.array_equal(pose, ref[0]), (h, "the bits of the pose")
            assert np.array_equal(ms, ref[2]), (h, "MatchingResult of every query")
            _same(st, ref[1])
        orc, opose, ost, _ = om.register(scan, guess, oracle.default_config(max_iterations=5))
        assert orc == 0 and ref[1].n_iterations == ost.n_iterations
        for it in range(ost.n_iterations):
            assert list(ref[1].iterations[it].reject_hist) == list(ost.iters[it].reject_hist)
            assert list(ref[1].iterations[it].obs_hist) == list(ost.iters[it].obs_hist)
        ok, e_t, e_r = pose_close(ref[0], opose, 1e-8, 1e-8)
        assert ok, (e_t, e_r)
    print(f"{scene} guesses {dt} m / {dth} deg: queries handed over by SOICP_KNN_HAND = {handed}")
    assert handed[64] >= handed[8] >= handed[2] >= handed[1] > 0, handed  # (the path was taken, and a larger limit hands over more)
    for s in ctxs.values():
        s.close()
