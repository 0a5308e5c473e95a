#!/usr/bin/env python3
"""Differential soak of Seam B (so_icp_knn_surf = LocalMap::nearestKSearchSurf, LocalMap.h:481-525) on the GPU box: random
maps (boxes, dense clusters, noisy planes, points on leaf / cell / cube boundaries, sensor sweeps) at random planeRes,
random queries (map points jittered from float spacings to metres, points near cube faces and in negative cubes, far
points, exact map points) against the oracle's exact cube-restricted search holding the same points in the same order:
found flags, d2 bit for bit, neighbour coordinates and indices; and, every few maps, the oracle against brute force.
usage: python tools/soak_knn.py [--seconds 120] [--seed 0]"""
import argparse, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]
import oracle_py as oracle  # noqa: E402
from superodom_amd import binding  # noqa: E402

ap = argparse.ArgumentParser(); ap.add_argument("--seconds", type=float, default=120.0); ap.add_argument("--seed", type=int, default=0)
a = ap.parse_args()
rng = np.random.default_rng(a.seed)


def cloud(centre, n):
    kind = rng.integers(0, 5)
    if kind == 0:
        p = rng.uniform(-1, 1, (n, 3)) * rng.uniform(1, 90, 3)
    elif kind == 1:
        k = rng.integers(1, 6)
        p = np.concatenate([rng.normal(0, rng.uniform(0.05, 2.0), (n // k + 1, 3)) + rng.uniform(-30, 30, 3) for _ in range(k)])
    elif kind == 2:
        p = rng.uniform(-60, 60, (n, 3)); p[:, 2] = rng.normal(0, 0.02, n) + rng.integers(-2, 3) * 3.0
    elif kind == 3:
        p = np.round(rng.uniform(-80, 80, (n, 3)) / 0.2) * 0.2 + rng.choice([0.0, 1e-7, -1e-7, 25.0], (n, 3))
    else:
        m = max(n // 64, 1)
        az = np.tile(np.linspace(0, 2 * np.pi, m, endpoint=False), 64)[:n]; el = np.repeat(np.linspace(-0.6, 0.2, 64), m)[:n]
        r = np.minimum(1.5 / np.maximum(-np.sin(el), 1e-3), 60.0)
        p = np.stack([r * np.cos(el) * np.cos(az), r * np.cos(el) * np.sin(az), r * np.sin(el)], 1) + rng.normal(0, 0.01, (len(az), 3))
    return (p + centre).astype(np.float32)


t_end, n_maps, n_q, n_bad, n_found = time.time() + a.seconds, 0, 0, 0, 0
while time.time() < t_end:
    res = float(rng.choice([0.1, 0.2, 0.4, 0.8]))
    centre = rng.uniform(-300, 300, 3) * [1, 1, 0.05]
    slam = binding.LidarSlamGpu(plane_res=res, line_res=res / 2)
    slam.set_origin(centre); slam.shift_map(centre)
    for _ in range(int(rng.integers(1, 4))):
        slam.add_surf_point_cloud(cloud(centre + rng.uniform(-40, 40, 3) * [1, 1, 0.02], int(10 ** rng.uniform(2, 5.3))))
    exp = slam.export_map()
    if len(exp) < 10:
        slam.close(); continue
    om = oracle.OracleMap(plane_res=res, line_res=res / 2); om.set_origin(centre); om.shift(centre)
    om.add_surf(exp, raw=True)
    nq = int(rng.integers(200, 3000))
    base = exp[rng.integers(0, len(exp), nq)]
    kind = rng.integers(0, 5, nq)
    q = base + rng.normal(0, 1, (nq, 3)) * (10.0 ** rng.uniform(-7, 0.3, (nq, 1)))
    q[kind == 1] = base[kind == 1]                                                              # exact map points
    face = np.round((base[kind == 2] + 25.0) / 50.0) * 50.0 - 25.0                                  # next to a cube face
    q[kind == 2] = np.where(rng.random((int((kind == 2).sum()), 3)) < 0.5, face + rng.normal(0, 0.3, face.shape), base[kind == 2])
    q[kind == 3] = base[kind == 3] + rng.uniform(-30, 30, (int((kind == 3).sum()), 3))               # anywhere, also outside
    q = q.astype(np.float32)
    found, nbr, d2, idx = slam.nearest_k_search_surf(q, 5)
    of, onbr, od2, oidx, _ = om.knn(q, 5, use_grid=1)
    n_maps += 1; n_q += nq; n_found += int(found.sum())
    f = found.astype(bool)
    ok = np.array_equal(found, of) and np.array_equal(d2[f].view(np.uint32), od2[f].view(np.uint32)) and np.array_equal(nbr[f], onbr[f])
    if ok and n_maps % 8 == 0:  # the checker itself against brute force inside the query's cube (float arithmetic of octree.h:93-102)
        sub = rng.integers(0, nq, 40)
        _, bn, bd, _, _ = om.knn(q[sub], 5, use_grid=0)
        ok = np.array_equal(bd[f[sub]].view(np.uint32), od2[sub][f[sub]].view(np.uint32))
    if not ok:
        n_bad += 1
        bad = np.nonzero((found != of) | (f & ((d2.view(np.uint32) != od2.view(np.uint32)).any(axis=1))))[0]
        print(f"MISMATCH map {n_maps} res {res} points {len(exp)} queries {nq}: {len(bad)} differ, first {bad[:3]}", q[bad[:2]], d2[bad[:2]], od2[bad[:2]], flush=True)
    slam.close()
print(f"soak: {n_maps} maps, {n_q} queries ({n_found} found), {n_bad} maps with a mismatch (seed {a.seed})")
sys.exit(1 if n_bad else 0)
