"""Knife-edge known-answer test of the per-point gates (LidarSlam.cpp:772 lambda0 < 1e-6, lambda1/lambda2 < 0.1;
LidarSlam.cpp:820-835 |n.p + d| > planeRes/2) on tests/golden/gate_edge.npz (generator: tests/golden/make_gate_edge.py,
expected MatchingResult computed there in 80-bit long double).

CPU part: the oracle (fp64 cyclic Jacobi + column-pivoted Householder) reproduces the expected status wherever the gated
quantity is at least 1e-12 away from its threshold.  GPU part (-m gpu): the HIP plane-fit pass -- a different eigen-solver
(Newton on the characteristic cubic) -- gives the SAME per-query MatchingResult as the oracle on every cluster, with the
direct solver and with the Jacobi solver switched back in (SOICP_ABLATE=512), and equal 7 + 9 bin histograms."""
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
FIX = np.load(os.path.join(HERE, "golden", "gate_edge.npz"))
PLANE_RES = float(FIX["plane_res"])
IDENTITY = np.array([0, 0, 0, 0, 0, 0, 1.0])
ASSERT_MARGIN = 1e-12  # the distance from the threshold down to which both fp64 implementations must agree with the long-double truth


def _batch(b):
    return FIX[f"pts{b}"], FIX[f"query{b}"], FIX[f"kind{b}"], FIX[f"margin{b}"], FIX[f"expect{b}"]


def _oracle_statuses(oracle, pts, query):
    om = oracle.OracleMap(plane_res=PLANE_RES)
    assert om.add_surf(pts.reshape(-1, 3)) == pts.size // 3   # through the VoxelGrid insert: one point per leaf survives as it is
    got = om.export()
    assert np.array_equal(got[np.lexsort(got.T)], pts.reshape(-1, 3)[np.lexsort(pts.reshape(-1, 3).T)])
    rc, pose, st, corrs = om.register(query, IDENTITY, oracle.default_config(max_iterations=1), want_corrs=True)
    assert rc == 0 and st.n_iterations == 1
    # the five neighbours of a query are its own cluster
    for i in range(len(query)):
        a = corrs["nbr"][i].reshape(5, 3); b = pts[i]
        assert np.array_equal(a[np.lexsort(a.T)], b[np.lexsort(b.T)])
    return om, st, corrs["status"].copy()


@pytest.mark.parametrize("b", range(int(FIX["n_batches"])))
def test_oracle_matches_long_double_truth_at_the_gate_edges(oracle, b):
    pts, query, kind, margin, expect = _batch(b)
    _, _, status = _oracle_statuses(oracle, pts, query)
    sure = np.abs(margin) >= ASSERT_MARGIN
    assert sure.sum() >= 10
    assert np.array_equal(status[sure], expect[sure]), (kind[sure], margin[sure], status[sure], expect[sure])
    # both outcomes of every gate are present
    assert set(np.unique(expect)) == {0, 3, 5}


@pytest.mark.gpu
@pytest.mark.parametrize("ablate", ["0", "512"])
def test_hip_fit_pass_agrees_with_the_oracle_at_the_gate_edges(oracle, soicp, ablate, monkeypatch):
    monkeypatch.setenv("SOICP_ABLATE", ablate)  # read when the context is created; 512 = cyclic Jacobi in the fit pass
    n_tight = 0
    for b in range(int(FIX["n_batches"])):
        pts, query, kind, margin, expect = _batch(b)
        om, ost, ostatus = _oracle_statuses(oracle, pts, query)
        slam = soicp.LidarSlamGpu(plane_res=PLANE_RES, line_res=PLANE_RES / 2, max_surface_features=-1, max_iterations=1)
        assert slam.add_surf_point_cloud(pts.reshape(-1, 3)) == pts.size // 3
        assert slam.map_size() == pts.size // 3
        rc, pose, st = slam.register(query, IDENTITY)
        assert rc == 0 and st.n_iterations == 1
        status = slam.match_status(len(query))
        sure = np.abs(margin) >= ASSERT_MARGIN
        assert np.array_equal(status[sure], expect[sure]), (b, kind[sure], margin[sure], status[sure], expect[sure])
        # against the oracle: every cluster, whatever its margin (down to ~1e-14 in this fixture)
        assert np.array_equal(status, ostatus), (b, kind, margin, status, ostatus)
        assert list(st.iterations[0].reject_hist) == list(ost.iters[0].reject_hist)
        assert list(st.iterations[0].obs_hist) == list(ost.iters[0].obs_hist)
        n_tight += int((~sure).sum())
        slam.close()
    assert n_tight >= 20  # the fixture does contain margins below 1e-12
