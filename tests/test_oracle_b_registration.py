"""PIN past the k-NN stage: a WHOLE registration with the reference's own engine in the loop.

On the corridor scene of tests/helpers.py (built so that the stock octree's two defects, oct.h:384-385 and 988-990, are inert)
Oracle-B -- every neighbour query of the registration answered by nanoflann::Octree from /root/reference's flann/octree.h
(oracle/_ref/libref_octree.so, one tree per 50 m block like MapBlock::octree_surf_, LocalMap.h:45-53, 638) -- must reproduce
Oracle-A (the exact cube-restricted search, the contract of SURVEY 8c) in every per-query MatchingResult, every neighbour
list, the 7 + 9 bin histograms, the LM iteration counts and termination codes of every outer iteration, and the BITS of the
pose.  The GPU twin of this test (tests/test_gpu_oracle_b.py) holds the product against both.

Skipped only when oracle/_ref/libref_octree.so is absent (it is built whenever /root/reference exists and travels with the
gpurun snapshot)."""
import os

import numpy as np
import pytest

from helpers import CorridorScene

REF_SO = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref", "libref_octree.so")
pytestmark = pytest.mark.skipif(not os.path.exists(REF_SO), reason="oracle/_ref/libref_octree.so not built (needs /root/reference)")


def oracle_map_of(oracle, sc):
    """The oracle's own map of the scene: inserted through ITS VoxelGrid, stored in ITS order."""
    om = oracle.OracleMap(plane_res=sc.plane_res)
    t0 = sc.gt_pose(0)[:3]
    om.set_origin(t0)
    om.shift(t0)
    assert om.add_surf(sc.map_points) > 0.99 * len(sc.map_points)
    return om


def run_a_and_b(oracle, om, scan, guess, max_iterations=5):
    cfg_a = oracle.default_config(max_iterations=max_iterations, use_grid_knn=1)
    cfg_b = oracle.default_config(max_iterations=max_iterations, use_grid_knn=2)
    a = om.register(scan, guess, cfg_a, want_corrs=True)
    assert oracle.enable_oracle_b()
    try:
        b = om.register(scan, guess, cfg_b, want_corrs=True)
    finally:
        oracle.reset_oracle_b()
        oracle.lib().orc_set_knn_hook(None)
    return a, b


def assert_same_registration(a, b, bits=True):
    (rc_a, pose_a, st_a, co_a), (rc_b, pose_b, st_b, co_b) = a, b
    assert rc_a == rc_b == 0
    assert st_a.n_iterations == st_b.n_iterations
    for it in range(st_a.n_iterations):
        x, y = st_a.iters[it], st_b.iters[it]
        assert (x.lm_iterations, x.num_successful_steps, x.termination, x.num_surf) == (y.lm_iterations, y.num_successful_steps, y.termination, y.num_surf), it
        assert list(x.reject_hist) == list(y.reject_hist) and list(x.obs_hist) == list(y.obs_hist), it
    if co_a is not None and co_b is not None:  # correspondences of the LAST outer iteration
        assert np.array_equal(co_a["status"], co_b["status"])
        ok = co_a["status"] == 0
        assert np.array_equal(co_a["nbr"][ok], co_b["nbr"][ok]), "neighbour lists (coordinates, in list order)"
        assert np.array_equal(co_a["d2"][ok].view(np.uint32), co_b["d2"][ok].view(np.uint32))
    if bits:
        assert np.array_equal(np.asarray(pose_a), np.asarray(pose_b)), (pose_a, pose_b)


@pytest.mark.parametrize("i,dt,dth", [(0, 0.10, 1.0), (3, 0.35, 2.5), (6, 0.10, 1.0)])
def test_registration_through_the_reference_octree_equals_the_exact_oracle(oracle, i, dt, dth):
    sc = CorridorScene()
    om = oracle_map_of(oracle, sc)
    scan, guess = sc.scan(i), sc.guess(i, dt, dth)
    a, b = run_a_and_b(oracle, om, scan, guess)
    assert a[2].n_iterations >= 2 and a[2].iters[0].num_surf > 0.6 * len(scan)
    from superodom_amd import synth
    et, er = synth.pose_error(a[1], sc.gt_pose(i))
    assert et < 0.02 and er < 0.004, ("the registration must converge for the comparison to mean anything", et, er)
    assert_same_registration(a, b)


def test_the_corridor_is_what_makes_the_octree_exact(oracle):
    """Control: the same comparison on the room-and-corridor world of the other tests (around the world origin) does NOT hold --
    the stock octree returns other neighbours there, which is why the contract is Oracle-A."""
    from superodom_amd import synth
    sc = synth.Scene("tiny")
    om = oracle.OracleMap(plane_res=sc.plane_res)
    om.add_surf(sc.map_points)
    a, b = run_a_and_b(oracle, om, sc.scan(0), sc.guess(0))
    differ = (a[3]["status"] != b[3]["status"]) | (a[3]["d2"].view(np.uint32) != b[3]["d2"].view(np.uint32)).any(1)
    assert differ.mean() > 0.01
