"""Oracle LocalMap + exact k-NN self-consistency (brute force vs grid; cube rules; negative cubes)."""
import numpy as np

from helpers import noisy_planes_cloud


def test_cube_index_rule_including_negative_side(oracle):
    m = oracle.OracleMap(plane_res=0.2)
    assert list(m.origin()) == [10, 10, 5]  # LocalMap.h:141-144
    pts = np.array([[0, 0, 0], [24.99, 0, 0], [25.0, 0, 0], [-25.0, 0, 0], [-25.01, 0, 0], [-75.0, 0, 0], [-75.01, 0, 0]], np.float32)
    m.add_surf(pts, raw=True)
    found, nbr, d2, idx, cube = m.knn(pts, 5, use_grid=0)
    # LocalMap.h:488-497: int((c+25)/50) truncates toward zero, then -- when c+25 < 0
    want_i = [0, 0, 1, 0, -1, -2, -2]  # note -75.0 -> -2 (int(-1.0) = -1, then --)
    assert [int(c) % 21 for c in cube] == [10 + w for w in want_i]
    # setOrigin puts the sensor's cube at index 0: negative-side cubes fall outside the window (LocalMap.h:146-164)
    m0 = oracle.OracleMap(plane_res=0.2)
    assert list(m0.set_origin(np.zeros(3))) == [0, 0, 0]
    assert m0.add_surf(pts, raw=True) == 4


def test_set_origin_and_shift_roll(oracle):
    m = oracle.OracleMap(plane_res=0.2)
    o = m.set_origin(np.array([130.0, -80.0, 3.0]))
    assert list(o) == [-3, 2, 0]  # -cube(t)
    m2 = oracle.OracleMap(plane_res=0.2)
    pts = noisy_planes_cloud(4000, np.random.default_rng(0))
    m2.add_surf(pts)
    n0 = m2.size()
    pos = m2.shift(np.array([10.0, 0, 0]))
    assert list(pos) == [10, 10, 5]
    # drive the sensor +x until the window rolls: blocks shift, points survive, origin moves
    o_before = m2.origin().copy()
    pos = m2.shift(np.array([450.0, 0, 0]))
    assert pos[0] == 21 - 4 and m2.origin()[0] < o_before[0]
    assert m2.size() == n0
    q = pts[:50]
    f1, nbr1, d21, _, _ = m2.knn(q, 5, use_grid=0)
    assert f1.all()
    # far enough and the old blocks fall out of the window
    m2.shift(np.array([2000.0, 0, 0]))
    assert m2.size() == 0


def test_grid_equals_bruteforce_multi_cube(oracle):
    rng = np.random.default_rng(1)
    pts = np.concatenate([noisy_planes_cloud(20000, rng, offset=(dx, dy, 0)) for dx in (-40, 10) for dy in (-35, 20)])
    m = oracle.OracleMap(plane_res=0.2)
    m.add_surf(pts)  # with VoxelGrid
    q = np.concatenate([pts[rng.integers(0, len(pts), 1500)] + rng.normal(0, 0.3, (1500, 3)),
                        np.c_[25.0 + rng.normal(0, 0.5, 300), rng.random(300) * 40 - 20, rng.random(300) * 5],
                        np.c_[-25.0 + rng.normal(0, 0.5, 300), -25.0 + rng.normal(0, 0.5, 300), rng.random(300) * 5],
                        rng.random((200, 3)) * 400 - 200]).astype(np.float32)
    a = m.knn(q, 5, use_grid=0); b = m.knn(q, 5, use_grid=1)
    assert np.array_equal(a[0], b[0])
    assert np.array_equal(a[2].view(np.uint32), b[2].view(np.uint32))
    assert np.array_equal(a[3], b[3])
    f = a[0].astype(bool)
    assert 0 < f.sum() < len(q)
    # neighbours never cross a cube face (LocalMap.h:488-520)
    nbr_cube = np.floor((a[1][f] + 25.0) / 50.0)
    q_cube = np.floor((q[f] + 25.0) / 50.0)
    full = a[2][f][:, 4] < 1e30
    assert (nbr_cube[full] == q_cube[full][:, None, :]).all()


def test_count_5x5_and_export(oracle):
    m = oracle.OracleMap(plane_res=0.2)
    pts = noisy_planes_cloud(8000, np.random.default_rng(2))
    m.add_surf(pts)
    assert m.count_5x5(np.array([10, 10, 5], np.int32)) == m.size() == len(m.export())
    assert m.count_5x5(np.array([2, 2, 5], np.int32)) == 0
