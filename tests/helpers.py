"""Shared helpers for the parity tests (test infrastructure)."""
import numpy as np

from superodom_amd import synth


def build_oracle_map_from(oracle, points, plane_res, origin_t, raw=True):
    m = oracle.OracleMap(plane_res=plane_res)
    m.set_origin(np.asarray(origin_t, float))
    m.add_surf(points, raw=raw)
    return m


def pose_close(a, b, tol_t=1e-4, tol_r=1e-4):
    dt, dr = synth.pose_error(a, b)
    return dt <= tol_t and dr <= tol_r, dt, dr


def noisy_planes_cloud(n, rng, offset=(0.0, 0.0, 0.0), sigma=0.01):
    """Four noisy planes (floor/ceiling/2 walls) inside one 50 m cube -- the SURVEY App. D probe scene."""
    k = n // 4
    parts = []
    xy = rng.random((k, 2)) * 40 - 20
    parts.append(np.c_[xy, -1.5 + sigma * rng.standard_normal(k)])
    xy = rng.random((k, 2)) * 40 - 20
    parts.append(np.c_[xy, 6.0 + sigma * rng.standard_normal(k)])
    xz = np.c_[rng.random(k) * 40 - 20, rng.random(k) * 7.5 - 1.5]
    parts.append(np.c_[xz[:, 0], 10.0 + sigma * rng.standard_normal(k), xz[:, 1]])
    yz = np.c_[rng.random(n - 3 * k) * 40 - 20, rng.random(n - 3 * k) * 7.5 - 1.5]
    parts.append(np.c_[-12.0 + sigma * rng.standard_normal(n - 3 * k), yz[:, 0], yz[:, 1]])
    return (np.concatenate(parts) + np.asarray(offset)).astype(np.float32)
