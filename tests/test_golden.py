"""Committed golden fixtures (tests/golden/*.npz, generator: tests/golden/make_golden.py).

CPU: the oracle reproduces (a) the 5-NN lists and d2 bits that the REFERENCE's own octree produced when the
fixture was generated, (b) numpy/scipy answers for the restated numerics, (c) its own earlier registrations.
GPU (-m gpu): the HIP path reproduces the same vectors through the C ABI -- no /root/reference needed at run time."""
import ctypes as C
import os

import numpy as np
import pytest

from superodom_amd import synth

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _oracle_map_for(oracle, pts):
    m = oracle.OracleMap(plane_res=0.2)
    c = pts.mean(0).astype(np.float64)
    m.set_origin(c); m.shift(c)
    assert m.add_surf(pts, raw=True) == len(pts)
    return m


def test_oracle_knn_equals_reference_octree_fixture(oracle):
    z = np.load(os.path.join(G, "knn_reference_octree.npz"))
    m = _oracle_map_for(oracle, z["points"])
    found, nbr, d2, idx, cube = m.knn(z["queries"], 5, use_grid=1)
    assert found.all() and len(set(cube.tolist())) == 1
    assert np.array_equal(d2.view(np.uint32), z["d2"].view(np.uint32))
    assert np.array_equal(nbr, z["points"][z["idx"]])


def test_oracle_numerics_equal_numpy_fixture(oracle):
    z = np.load(os.path.join(G, "numerics_kat.npz")); L = oracle.lib()
    p = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
    for S, W, P, X in zip(z["S"], z["W"], z["P"], z["X"]):
        ev = np.zeros(3); V = np.zeros(9); x = np.zeros(3)
        L.orc_eig3_sym(p(np.ascontiguousarray(S.ravel())), p(ev), p(V))
        assert np.allclose(ev, W, rtol=0, atol=1e-12 * max(1.0, abs(W).max()))
        assert L.orc_plane_ls5(p(np.ascontiguousarray(P.ravel())), p(x))
        assert np.allclose(x, X, rtol=1e-8, atol=1e-12)


def _check_registration(z, k, pose, n_it, lm, acc, rej, obs, tol):
    assert n_it == z["n_iterations"][k]
    assert lm[:n_it] == list(z["lm_iterations"][k][:n_it])
    assert acc[:n_it] == list(z["accepted"][k][:n_it])
    assert rej[:n_it] == z["reject_hist"][k][:n_it].tolist()
    assert obs[:n_it] == z["obs_hist"][k][:n_it].tolist()
    dt, dr = synth.pose_error(pose, z["poses"][k])
    assert dt <= tol and dr <= tol, (dt, dr)


def test_oracle_registration_matches_fixture(oracle):
    z = np.load(os.path.join(G, "register_tiny.npz"))
    sc = synth.Scene("tiny")
    assert float(sc.map_points.astype(np.float64).sum()) == z["map_checksum"][0], "synthetic scene generator drifted"
    assert float(sc.scan(0).astype(np.float64).sum()) == z["scan0_checksum"][0]
    om = oracle.OracleMap(plane_res=sc.plane_res); om.add_surf(sc.map_points)
    for k, i in enumerate(z["scan_ids"]):
        rc, pose, st, _ = om.register(sc.scan(int(i)), sc.guess(int(i)), oracle.default_config(max_iterations=5))
        n = st.n_iterations
        _check_registration(z, k, pose, n, [st.iters[j].lm_iterations for j in range(n)], [st.iters[j].num_surf for j in range(n)],
                            [list(st.iters[j].reject_hist) for j in range(n)], [list(st.iters[j].obs_hist) for j in range(n)], 1e-12)


@pytest.mark.gpu
def test_gpu_knn_equals_reference_octree_fixture(gpu_slam_factory):
    z = np.load(os.path.join(G, "knn_reference_octree.npz"))
    slam = gpu_slam_factory(plane_res=0.2)
    c = z["points"].mean(0).astype(np.float64)
    slam.set_origin(c); slam.shift_map(c)
    assert slam.add_surf_point_cloud(z["points"]) == len(z["points"])  # sparse random points: VoxelGrid keeps them apart
    exported = slam.export_map()
    if len(exported) != len(z["points"]):
        pytest.skip("voxel filter merged fixture points")
    found, nbr, d2, idx = slam.nearest_k_search_surf(z["queries"], 5)
    assert found.all()
    assert np.array_equal(d2.view(np.uint32), z["d2"].view(np.uint32)), "d2 bits must equal the reference octree's"
    assert np.array_equal(nbr, z["points"][z["idx"]])


@pytest.mark.gpu
def test_gpu_registration_matches_fixture(gpu_slam_factory):
    z = np.load(os.path.join(G, "register_tiny.npz"))
    sc = synth.Scene("tiny")
    slam = gpu_slam_factory(plane_res=sc.plane_res, max_surface_features=-1, max_iterations=5)
    slam.add_surf_point_cloud(sc.map_points)
    for k, i in enumerate(z["scan_ids"]):
        rc, pose, st = slam.register(sc.scan(int(i)), sc.guess(int(i)))
        n = st.n_iterations
        _check_registration(z, k, pose, n, [st.iterations[j].lm_iterations for j in range(n)],
                            [st.iterations[j].num_surf_from_scan for j in range(n)],
                            [list(st.iterations[j].reject_hist) for j in range(n)], [list(st.iterations[j].obs_hist) for j in range(n)], 1e-4)
