#!/usr/bin/env python3
"""Generator of tests/golden/gate_edge.npz -- knife-edge known-answer clusters for the three per-point gates of
ComputePlaneDistanceParameters (paths relative to /root/reference/super_odometry/):

  kind 0  lambda0 < 1e-6              -> BAD_PCA_STRUCTURE   src/LidarProcess/LidarSlam.cpp:772
  kind 1  lambda1 / lambda2 < 0.1     -> BAD_PCA_STRUCTURE   src/LidarProcess/LidarSlam.cpp:772
  kind 2  |n.p_j + d| > planeRes / 2  -> MSE_TOO_LARGE       src/LidarProcess/LidarSlam.cpp:820-835

A cluster = 5 float32 map points that are the exact 5-NN of one query (clusters sit 2.4 m apart on the z axis, far more
than the 0.775 m gate) and lie in five different 0.2 m VoxelGrid leaves (the map insert keeps them as they are).  Every
cluster comes as a PAIR of float32-adjacent configurations straddling the gate: a coarse coordinate is bisected to the
last float below the threshold, then a SECOND-ORDER knob (a coordinate whose first-order effect vanishes by symmetry and
whose absolute value is tiny, hence with sub-picometre ulps) is bisected to two adjacent floats on either side -- the
gated quantity then differs from the threshold by 1e-12 .. 1e-16.  The quantity is evaluated here in 80-bit long double
(cyclic Jacobi / normal equations), i.e. with ~1e-19 of absolute error, so `margin` is trustworthy down to ~1e-17 and
the expected MatchingResult is known independently of the oracle and of the product.

    python tests/golden/make_gate_edge.py        (writes tests/golden/gate_edge.npz; numpy only, seeded)
"""
import os

import numpy as np

LD = np.longdouble
PLANE_RES = np.float32(0.2)
N_BATCH, LEVELS = 6, 20


def f2o(x):
    """float32 -> integer that orders like the float (adjacent floats <-> adjacent integers)."""
    i = int(np.float32(x).view(np.int32))
    return i if i >= 0 else -(i & 0x7FFFFFFF)


def o2f(o):
    i = o if o >= 0 else ((-o) | 0x80000000) - (1 << 32)
    return np.array([i], dtype=np.int64).astype(np.int32).view(np.float32)[0]


def jacobi_eig_ld(S):
    a = np.array(S, dtype=LD)
    for _ in range(60):
        off = a[0, 1] ** 2 + a[0, 2] ** 2 + a[1, 2] ** 2
        if off == 0:
            break
        for p, q in ((0, 1), (0, 2), (1, 2)):
            if a[p, q] == 0:
                continue
            th = (a[q, q] - a[p, p]) / (2 * a[p, q])
            with np.errstate(over='ignore'):
                t = np.sign(th) / (abs(th) + np.sqrt(th * th + 1)) if th != 0 else LD(1)
            c = 1 / np.sqrt(t * t + 1); s = t * c
            G = np.eye(3, dtype=LD); G[p, p] = c; G[q, q] = c; G[p, q] = s; G[q, p] = -s
            a = G.T @ a @ G
    return np.sort(np.diag(a))


def scatter_ld(P):
    P = np.asarray(P, dtype=np.float32).astype(LD)
    c = P - P.sum(0) / LD(5)
    return c.T @ c


def lam(P):
    return jacobi_eig_ld(scatter_ld(P))


def max_plane_dist(P):
    """max_j |n.p_j + d| of the LS plane A x = -1 (LidarSlam.cpp:798-835), long double."""
    A = np.asarray(P, dtype=np.float32).astype(LD)
    M = A.T @ A
    b = -(A.T @ np.ones(5, dtype=LD))
    # 3x3 solve by Cramer in long double
    det = np.linalg.det(M.astype(np.float64))  # only to reject singular set-ups
    assert abs(det) > 1e-12
    x = np.zeros(3, dtype=LD)
    D = M[0, 0] * (M[1, 1] * M[2, 2] - M[1, 2] * M[2, 1]) - M[0, 1] * (M[1, 0] * M[2, 2] - M[1, 2] * M[2, 0]) + M[0, 2] * (M[1, 0] * M[2, 1] - M[1, 1] * M[2, 0])
    for k in range(3):
        Mk = M.copy(); Mk[:, k] = b
        x[k] = (Mk[0, 0] * (Mk[1, 1] * Mk[2, 2] - Mk[1, 2] * Mk[2, 1]) - Mk[0, 1] * (Mk[1, 0] * Mk[2, 2] - Mk[1, 2] * Mk[2, 0])
                + Mk[0, 2] * (Mk[1, 0] * Mk[2, 1] - Mk[1, 1] * Mk[2, 0])) / D
    # one step of iterative refinement on the normal equations
    r = b - M @ x
    dx = np.linalg.solve(M.astype(np.float64), r.astype(np.float64)).astype(LD)
    x = x + dx
    nn = np.sqrt((x * x).sum())
    return (np.abs(A @ (x / nn) + 1 / nn)).max()


def bisect_float(make, q, thr, lo, hi):
    """make(v) -> points with the knob at float32 v; q(points) increasing in v on [lo, hi]; returns adjacent floats (a, b) with
    q(a) < thr <= q(b), or None when the threshold is not crossed inside the interval."""
    ol, oh = f2o(lo), f2o(hi)
    if not (q(make(o2f(ol))) < thr <= q(make(o2f(oh)))):
        return None
    while oh - ol > 1:
        om = (ol + oh) // 2
        if q(make(o2f(om))) < thr:
            ol = om
        else:
            oh = om
    return o2f(ol), o2f(oh)


TARGETS = [0.0, 1e-13, 1e-12, 1e-11, 1e-10, 1e-9]


def knife_edge(pts, q, thr, coarse_rng, fine_rng, target, side):
    """One configuration with q - thr just beyond -target (side 0: last float below thr - target) or +target (side 1: first
    float at or above thr + target).  pts(coarse, fine) -> 5 points; q increasing in both knobs on their ranges; the coarse
    knob is bisected with the fine knob at the start of its range, then the fine (second-order) knob is bisected."""
    t = thr - LD(target) if side == 0 else thr + LD(target)
    co = bisect_float(lambda v: pts(v, fine_rng[0]), q, t, coarse_rng[0], coarse_rng[1])
    if co is None:
        return None
    fi = bisect_float(lambda v: pts(co[0], v), q, t, fine_rng[0], fine_rng[1])
    if fi is None:
        return None
    P = pts(co[0], fi[side])
    return P, q(P) - thr


def build_lambda0(rng, zl, target=0.0, side=0):
    """kind 0: plane x = x0, in-plane axes y / z.  Offsets along the normal (a', a, -a, -a, dxc): lambda0 = S_xx ~ 4 a^2 + 0.8 dxc^2.
    Coarse knob: x of the first point; second-order knob: x of the fifth point (dxc >= 0)."""
    x0 = np.float32(0.35 + 0.3 * rng.random()); s = np.float32(0.3)
    a0 = 5e-4 * (1 + 0.02 * (rng.random() - 0.5))
    a = np.float32(a0)

    def pts(p1x, p5x):
        return np.array([[p1x, s, zl], [x0 + a, -s, zl], [x0 - a, 0, zl + s], [x0 - a, 0, zl - s], [p5x, 0, zl]], np.float32)

    r = knife_edge(pts, lambda P: lam(P)[0], LD(1e-6), (np.float32(x0 + 0.8 * a0), np.float32(x0 + 1.3 * a0)),
                   (x0, np.float32(x0 + 6e-5)), target, side)
    return None if r is None else (r, np.array([x0, 0.013, zl + 0.021], np.float32))


def build_ratio(rng, zl, target=0.0, side=0):
    """kind 1: plane z = zl, five points along x with alternating y (sum y = 0): lambda1 / lambda2 through 0.1.
    Coarse knob: x of the last point (stretches lambda2: the ratio FALLS with it, so the knob is -x); second-order knob: y of
    the middle point (x = 0, sum y = dy: S_yy = 4 t^2 + 0.8 dy^2, the ratio rises with dy >= 0)."""
    s = np.float32(0.22 + 0.01 * rng.random()); qz = np.float32(2e-3 * (1 + 0.2 * rng.random()))
    t = np.float32(0.1189 * float(s) / 0.22 * (1 + 0.004 * (rng.random() - 0.5)))

    def pts(mx5, dy):
        return np.array([[-2 * s, t, zl + qz], [-s, -t, zl - qz], [0, dy, zl], [s, t, zl - qz], [-mx5, -t, zl + qz]], np.float32)

    def q(P):
        e = lam(P)
        return e[1] / e[2]

    thr = LD(np.float64(0.1))  # the double literal the gate compares with (ev[1] / ev[2] < 0.1), not 1/10
    r = knife_edge(pts, q, thr, (np.float32(-2.2 * float(s)), np.float32(-1.8 * float(s))), (np.float32(0), np.float32(4e-3)), target, side)
    return None if r is None else (r, np.array([0.012, 0.017, zl + 0.05], np.float32))


def build_inlier(rng, zl, target=0.0, side=0):
    """kind 2: four corners in the plane z = zl, the centre point lifted by h ~ 0.125 away from the origin: its distance to the
    LS plane through planeRes / 2.  Coarse knob: |z| of the lifted point; second-order knob: its in-plane x offset (the
    distance is even in it; whichever way it moves the distance, the quantity is oriented so that it rises)."""
    s = np.float32(0.3 + 0.02 * rng.random())
    sgn = np.float32(1.0 if zl > 0 else -1.0)
    thr = LD(np.float64(np.float32(0.2)) / 2.0)  # (double)planeRes / 2.0, LidarSlam.cpp:820

    def pts_abs(absz5, dx):
        return np.array([[s, s, zl], [s, -s, zl], [-s, s, zl], [-s, -s, zl], [dx, 0, sgn * absz5]], np.float32)

    zmid = np.float32(abs(zl) + 0.125)
    up = max_plane_dist(pts_abs(zmid, np.float32(1e-2))) > max_plane_dist(pts_abs(zmid, np.float32(0)))
    crng = (np.float32(abs(zl) + 0.10), np.float32(abs(zl) + 0.15))
    if up:
        r = knife_edge(pts_abs, max_plane_dist, thr, crng, (np.float32(0), np.float32(3e-2)), target, side)
    else:  # the knob lowers the distance: run the bisections on (-distance) with the coarse knob reversed, sides swapped
        r = knife_edge(lambda mc, dx: pts_abs(-mc, dx), lambda P: -max_plane_dist(P), -thr, (-crng[1], -crng[0]),
                       (np.float32(0), np.float32(3e-2)), target, 1 - side)
        if r is not None:
            r = (r[0], -r[1])
    return None if r is None else (r, np.array([0.011, -0.014, zl + sgn * 0.03], np.float32))


def expected_status(P):
    """MatchingResult of a 5-neighbour set, long double (LidarSlam.cpp:749-844 without the k-NN / distance gates)."""
    e = lam(P)
    if e[0] < LD(1e-6) or e[1] / e[2] < LD(np.float64(0.1)):
        return 3
    return 5 if max_plane_dist(P) > LD(np.float64(np.float32(0.2)) / 2.0) else 0


def main():
    rng = np.random.default_rng(20260924)
    levels = [sg * (1.2 + 2.4 * k) for k in range(LEVELS // 2) for sg in (+1, -1)]
    builders = [build_lambda0, build_ratio, build_inlier]
    batches = []
    for b in range(N_BATCH):
        pts, qs, kinds, margins, expect = [], [], [], [], []
        for idx, level in enumerate(levels):  # one knife-edge configuration per level: kinds, sides and margins cycle
            kind, side, target = (idx // 2 + b) % 3, idx % 2, TARGETS[(idx // 6 + idx // 2 + b) % len(TARGETS)]
            for attempt in range(50):
                r = builders[kind](rng, np.float32(level), target, side)
                if r is not None:
                    break
            else:
                raise RuntimeError(f"no knife edge found (kind {kind}, level {level}, target {target})")
            (P, m), query = r
            pts.append(P); qs.append(query); kinds.append(kind); margins.append(float(m)); expect.append(expected_status(P))
        batches.append((np.stack(pts), np.stack(qs), np.array(kinds, np.int32), np.array(margins), np.array(expect, np.int32)))
    out = {"n_batches": np.int32(N_BATCH), "plane_res": PLANE_RES}
    for b, (p, q, k, m, e) in enumerate(batches):
        out[f"pts{b}"] = p; out[f"query{b}"] = q; out[f"kind{b}"] = k; out[f"margin{b}"] = m; out[f"expect{b}"] = e
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "gate_edge.npz")
    np.savez_compressed(path, **out)
    allm = np.concatenate([b[3] for b in batches]); alle = np.concatenate([b[4] for b in batches]); allk = np.concatenate([b[2] for b in batches])
    for kind in range(3):
        mk = np.abs(allm[allk == kind])
        print(f"kind {kind}: {len(mk)} clusters, |margin| min {mk.min():.2e} median {np.median(mk):.2e} max {mk.max():.2e}; "
              f"expected statuses {np.bincount(alle[allk == kind], minlength=6).tolist()}")
    print("wrote", path)


if __name__ == "__main__":
    main()
