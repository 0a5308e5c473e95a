#!/usr/bin/env python3
"""Differential soak of the map insert's two first stages (GPU box): the same random insert sequences through a context
with the hash grouping (default) and one with SOICP_MAP_GROUPING=sort; after every insert the exported maps must be
bitwise equal (same points, same canonical order).  usage: python tools/soak_map_insert.py [--seconds 60] [--seed 0]"""
import argparse, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from superodom_amd import binding  # noqa: E402

ap = argparse.ArgumentParser(); ap.add_argument("--seconds", type=float, default=60.0); ap.add_argument("--seed", type=int, default=0)
ap.add_argument("--oracle", action="store_true", help="also insert into the CPU oracle's LocalMap and compare the point sets")
a = ap.parse_args()
if a.oracle:
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle_py
rng = np.random.default_rng(a.seed)


def make(mode, res):
    os.environ["SOICP_MAP_GROUPING"] = mode
    return binding.LidarSlamGpu(plane_res=res, line_res=res / 2)


def cloud(centre):
    kind = rng.integers(0, 5)
    n = int(10 ** rng.uniform(0, 5.2))
    if kind == 0:   # uniform box
        p = rng.uniform(-1, 1, (n, 3)) * rng.uniform(1, 120, 3)
    elif kind == 1:  # a few very dense clusters (giant leaves) + scatter
        k = rng.integers(1, 6)
        p = np.concatenate([rng.normal(0, rng.uniform(0.01, 0.3), (n // k + 1, 3)) + rng.uniform(-30, 30, 3) for _ in range(k)])
    elif kind == 2:  # noisy planes
        p = rng.uniform(-60, 60, (n, 3)); p[:, 2] = rng.normal(0, 0.02, n) + rng.integers(-2, 3) * 3.0
    elif kind == 3:  # points on the leaf / cell / cube boundaries
        p = np.round(rng.uniform(-80, 80, (n, 3)) / 0.2) * 0.2 + rng.choice([0.0, 1e-7, -1e-7, 25.0], (n, 3))
    else:            # ring-major sweep of a sensor over a floor: long runs of consecutive points in one leaf
        az = np.tile(np.linspace(0, 2 * np.pi, max(n // 64, 1), endpoint=False), 64)[:n]
        el = np.repeat(np.linspace(-0.6, 0.2, 64), max(n // 64, 1))[:n]
        r = np.minimum(1.5 / np.maximum(-np.sin(el), 1e-3), 60.0)
        p = np.stack([r * np.cos(el) * np.cos(az), r * np.cos(el) * np.sin(az), r * np.sin(el)], 1) + rng.normal(0, 0.01, (len(az), 3))
    return (p + centre).astype(np.float32)


t_end, rounds, inserts, points = time.time() + a.seconds, 0, 0, 0
while time.time() < t_end:
    res = float(rng.choice([0.1, 0.2, 0.4, 0.8]))
    x, y = make("hash", res), make("sort", res)
    centre = rng.uniform(-300, 300, 3) * [1, 1, 0.05]
    for m in (x, y):
        m.set_origin(centre); m.shift_map(centre)
    om = None
    if a.oracle:
        om = oracle_py.OracleMap(plane_res=res, line_res=res / 2); om.set_origin(centre); om.shift(centre)
    ops = [("origin", centre.copy(), res)]
    for step in range(int(rng.integers(2, 9))):
        if rng.random() < 0.2:
            centre = centre + rng.uniform(-120, 120, 3) * [1, 1, 0.02]
            assert list(x.shift_map(centre)) == list(y.shift_map(centre))
            if om: assert list(om.shift(centre)) == list(x.shift_map(centre))
            ops.append(("shift", centre.copy(), res))
        if rng.random() < 0.1:
            res = float(rng.choice([0.1, 0.2, 0.4, 0.8]))
            x.set_resolution(res / 2, res); y.set_resolution(res / 2, res)
            if om: om.set_resolution(res / 2, res)
            ops.append(("res", centre.copy(), res))
        pts = cloud(centre)
        assert x.add_surf_point_cloud(pts) == y.add_surf_point_cloud(pts)
        ops.append(("add", pts, res))
        if om:
            assert om.add_surf(pts) >= 0
            eo, eg = om.export(), x.export_map()
            same = eo.shape == eg.shape and np.array_equal(eo[np.lexsort(eo.T)].view(np.uint32), eg[np.lexsort(eg.T)].view(np.uint32))
            if not same:  # (also after resolution changes: a re-filtered cube sums its old points in the order of its previous leaf grid)
                print("ORACLE MISMATCH", (a.seed, rounds, step, len(pts), res), eo.shape, eg.shape); sys.exit(1)
        ex, ey = x.export_map(), y.export_map()
        if not (ex.shape == ey.shape and np.array_equal(ex.view(np.uint32), ey.view(np.uint32))):
            os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
            import pickle
            pickle.dump(ops, open(os.path.join(ROOT, "gpurun_out", "soak_fail.pkl"), "wb"))
            print("MISMATCH", (a.seed, rounds, step, len(pts), res), "shapes", ex.shape, ey.shape)
            if ex.shape == ey.shape:
                bad = np.flatnonzero((ex.view(np.uint32) != ey.view(np.uint32)).any(1))
                print("differing rows", len(bad), bad[:10]); print(ex[bad[:5]]); print(ey[bad[:5]])
                sx, sy = ex[np.lexsort(ex.T)], ey[np.lexsort(ey.T)]
                print("same multiset of points:", np.array_equal(sx.view(np.uint32), sy.view(np.uint32)))
            sys.exit(1)
        inserts += 1; points += len(pts)
    x.close(); y.close()
    rounds += 1
print(f"soak ok: {rounds} maps, {inserts} inserts, {points} points, both first stages bitwise equal (seed {a.seed})")
