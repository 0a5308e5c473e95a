"""-m gpu: the product against BOTH oracles on the scene where the reference's own k-NN engine is exact (tests/helpers.py:
CorridorScene; CPU twin: tests/test_oracle_b_registration.py, which shows Oracle-B == Oracle-A there bit for bit).

The HIP path registers the same scans against ITS map -- the cloud inserted through so_icp_map_add_surf (device VoxelGrid),
canonical (cube, cell, leaf) order -- while the oracle holds the cloud it voxel-filtered itself, in its own storage order, and
answers its neighbour queries through nanoflann::Octree of /root/reference (oracle/_ref): nothing is shared between the two
sides but the input cloud, the scan and the guess.  Equal: outer / LM iteration counts, termination codes, accepted counts,
the 7 + 9 bin histograms of every outer iteration, the per-query MatchingResult of the last one; poses within 1e-6 m / rad
(the corridor sits 1.1 km from the origin, where the un-centred A x = -1 plane of the reference is conditioned ~1e3 worse than
in the other scenes; north_star's bar is 1e-4)."""
import os

import numpy as np
import pytest

from helpers import CorridorScene, pose_close
from test_oracle_b_registration import REF_SO, oracle_map_of, run_a_and_b, assert_same_registration

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not os.path.exists(REF_SO), reason="oracle/_ref/libref_octree.so not built")]


@pytest.mark.parametrize("i,dt,dth", [(0, 0.10, 1.0), (3, 0.35, 2.5), (6, 0.10, 1.0)])
def test_hip_registration_equals_the_reference_octree_driven_oracle(oracle, gpu_slam_factory, i, dt, dth):
    sc = CorridorScene()
    om = oracle_map_of(oracle, sc)
    slam = gpu_slam_factory(plane_res=sc.plane_res, line_res=sc.plane_res / 2, max_surface_features=-1, max_iterations=5)
    t0 = sc.gt_pose(0)[:3]
    slam.set_origin(t0); slam.shift_map(t0)
    assert slam.add_surf_point_cloud(sc.map_points) == om.size() == slam.map_size()
    # the two maps hold the same points (each side's own VoxelGrid), in unrelated orders
    mine, theirs = slam.export_map(), om.export()
    assert np.array_equal(mine[np.lexsort(mine.T)], theirs[np.lexsort(theirs.T)]) and not np.array_equal(mine, theirs)
    scan, guess = sc.scan(i), sc.guess(i, dt, dth)
    a, b = run_a_and_b(oracle, om, scan, guess)
    assert_same_registration(a, b)
    rc, pose, st = slam.register(scan, guess)
    assert rc == 0
    for tag, (orc, opose, ost, ocorr) in (("Oracle-A", a), ("Oracle-B (reference octree.h)", b)):
        assert st.n_iterations == ost.n_iterations, tag
        for it in range(st.n_iterations):
            x, y = st.iterations[it], ost.iters[it]
            assert (x.lm_iterations, x.num_successful_steps, x.termination, x.num_surf_from_scan) == \
                   (y.lm_iterations, y.num_successful_steps, y.termination, y.num_surf), (tag, it)
            assert list(x.reject_hist) == list(y.reject_hist) and list(x.obs_hist) == list(y.obs_hist), (tag, it)
        assert np.array_equal(slam.match_status(len(scan)), ocorr["status"]), tag
        ok, et, er = pose_close(pose, opose, 1e-6, 1e-6)
        assert ok, (tag, et, er)
    slam.close()
