#!/usr/bin/env python3
"""Regenerates the committed golden fixtures (run in the build container, where /root/reference exists):

  knn_reference_octree.npz   5-NN of the REFERENCE's own nanoflann::Octree (oracle/_ref/libref_octree.so, compiled from
                             /root/reference/.../flann/octree.h) on a scene where its two pruning bugs are inert:
                             points, queries, indices, d2 -> pins oct.h:93-102 arithmetic + nf.h:117-147 ordering.
  numerics_kat.npz           numpy/scipy answers for the restated third-party numerics (eigh, lstsq).
  register_tiny.npz          oracle registrations of the seeded 'tiny' scene (poses, counts, histograms) -> drift guard
                             for the oracle and expected values for the GPU path.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np
import scipy.linalg

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import oracle_py as O  # noqa: E402
from superodom_amd import synth  # noqa: E402


def knn_reference():
    rng = np.random.default_rng(7)
    n = 4000
    pts = np.c_[1000.0 + rng.random(n) * 20.0, rng.random(n) * 6.0 - 3.0, rng.random(n) * 6.0 - 3.0].astype(np.float32)
    # one point per 0.2 m voxel, so that LocalMap::addSurfPointCloud's VoxelGrid leaves the cloud unchanged
    _, first = np.unique(np.floor(pts * np.float32(5.0)).astype(np.int64) @ np.array([1 << 40, 1 << 20, 1]), return_index=True)
    pts = pts[np.sort(first)]; n = len(pts)
    q = (pts[rng.integers(0, n, 600)] + rng.normal(0, 0.05, (600, 3))).astype(np.float32)
    idx, d2 = O.RefOctree(pts).knn(q, 5)
    np.savez_compressed(os.path.join(HERE, "knn_reference_octree.npz"), points=pts, queries=q, idx=idx.astype(np.int32), d2=d2)


def numerics():
    rng = np.random.default_rng(11)
    S = []; W = []; P = []; X = []
    for _ in range(64):
        pts = rng.normal(0, 1, (5, 3)) * rng.choice([1e-3, 1e-2, 0.1, 1.0], 3) + rng.normal(0, 50, 3)
        pts = pts.astype(np.float32).astype(np.float64)
        c = pts - pts.mean(0); s = c.T @ c
        S.append(s); W.append(np.linalg.eigh(s)[0])
        P.append(pts); X.append(scipy.linalg.lstsq(pts, -np.ones(5))[0])
    np.savez_compressed(os.path.join(HERE, "numerics_kat.npz"), S=np.array(S), W=np.array(W), P=np.array(P), X=np.array(X))


def register_tiny():
    sc = synth.Scene("tiny", cache_dir="/tmp/soicp_cache_golden")
    om = O.OracleMap(plane_res=sc.plane_res); om.add_surf(sc.map_points)
    ids = [0, 5, 11]
    poses = []; nit = []; lm = []; acc = []; rej = []; obs = []
    for i in ids:
        rc, pose, st, _ = om.register(sc.scan(i), sc.guess(i), O.default_config(max_iterations=5))
        assert rc == 0
        poses.append(pose); nit.append(st.n_iterations)
        lm.append([st.iters[k].lm_iterations for k in range(5)]); acc.append([st.iters[k].num_surf for k in range(5)])
        rej.append([list(st.iters[k].reject_hist) for k in range(5)]); obs.append([list(st.iters[k].obs_hist) for k in range(5)])
    np.savez_compressed(os.path.join(HERE, "register_tiny.npz"), scan_ids=np.array(ids), poses=np.array(poses), n_iterations=np.array(nit),
                        lm_iterations=np.array(lm), accepted=np.array(acc), reject_hist=np.array(rej), obs_hist=np.array(obs),
                        map_checksum=np.array([float(sc.map_points.astype(np.float64).sum())]),
                        scan0_checksum=np.array([float(sc.scan(0).astype(np.float64).sum())]))


if __name__ == "__main__":
    knn_reference(); numerics(); register_tiny()
    print("golden fixtures written to", HERE)
