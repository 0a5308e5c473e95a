"""The product's plane fit (superodom_amd/csrc/plane_fit.h -- host + device code, the closed form of the reference's 5x3
least-squares plane, LidarSlam.cpp:798-816) compiled for the HOST and checked on the CPU:

  * against the oracle (fp64 cyclic Jacobi + column-pivoted Householder, oracle/so_oracle.c: orc_plane_match) on every
    correspondence of real synthetic registrations: MatchingResult and observability labels equal, plane / coefficient to 1e-10;
  * against the 80-bit gate-edge fixture (tests/golden/gate_edge.npz);
  * against an 80-bit evaluation of the same least-squares problem on random clusters far from the origin (accuracy claim
    in the header of plane_fit.h, with column-pivoted QR in fp64 beside it);
  * the short cut of the observability labels against the reference's arithmetic as written.

The device build of the same header runs in the fit pass of solve_kernel; its GPU parity tests are the registration tests."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SRC = os.path.join(HERE, "native", "plane_fit_host.cpp")
LIB = os.path.join(HERE, "native", "libplane_fit_host.so")
HDR = os.path.join(ROOT, "superodom_amd", "csrc")
IDENTITY = np.array([0, 0, 0, 0, 0, 0, 1.0])


@pytest.fixture(scope="module")
def pf():
    deps = [SRC, os.path.join(HDR, "plane_fit.h"), os.path.join(HDR, "so_math.h")]
    if not os.path.exists(LIB) or any(os.path.getmtime(d) > os.path.getmtime(LIB) for d in deps):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared", "-I", HDR, SRC, "-o", LIB])
    L = C.CDLL(LIB)
    f32p, f64p, i32p = C.POINTER(C.c_float), C.POINTER(C.c_double), C.POINTER(C.c_int)
    L.pf_fit.argtypes = [f32p, f64p, f64p, C.c_float, C.c_double, C.c_int, C.c_int, f64p, f64p, i32p, i32p]

    def fit(nb, pw, pose, plane_res, as_written=False):
        nb = np.ascontiguousarray(nb, np.float32).reshape(-1, 15); pw = np.ascontiguousarray(pw, np.float64).reshape(-1, 3)
        pose = np.ascontiguousarray(pose, np.float64)
        n = len(nb)
        nd = np.zeros((n, 4)); co = np.zeros(n); st = np.zeros(n, np.int32); ob = np.zeros((n, 3), np.int32)
        pr = np.float32(plane_res)
        L.pf_fit(nb.ctypes.data_as(f32p), pw.ctypes.data_as(f64p), pose.ctypes.data_as(f64p), np.float32(3) * pr, float(pr) / 2.0, n,
                 1 if as_written else 0, nd.ctypes.data_as(f64p), co.ctypes.data_as(f64p), st.ctypes.data_as(i32p), ob.ctypes.data_as(i32p))
        return st, nd, co, ob
    return fit


def _world(pose, p):
    from superodom_amd import synth
    return p @ synth.quat_to_R(pose[3:]).T + pose[:3]


@pytest.mark.parametrize("scene,scans", [("tiny", (0, 1, 2)), ("small", (0, 5))])
def test_host_fit_equals_the_oracle_on_real_correspondences(oracle, pf, scene, scans):
    from superodom_amd import synth
    sc = synth.Scene(scene)
    om = oracle.OracleMap(plane_res=sc.plane_res)
    om.add_surf(sc.map_points)
    n_fit = n_ok = 0
    for i in scans:
        pose = np.asarray(sc.guess(i), np.float64)
        rc, _, st, corrs = om.register(sc.scan(i), pose, oracle.default_config(max_iterations=1), want_corrs=True)
        assert rc == 0
        reached = corrs["status"] != 1  # (1 = NOT_ENOUGH_NEIGHBORS, 2 = TOO_FAR: never reach the fit)
        reached &= corrs["status"] != 2
        c = corrs[reached]
        pw = _world(pose, np.asarray(c["p"], np.float64))
        status, nd, co, ob = pf(c["nbr"], pw, pose, sc.plane_res)
        assert np.array_equal(status, c["status"]), np.flatnonzero(status != c["status"])[:10]
        ok = status == 0
        assert ok.sum() > 100
        assert np.array_equal(ob[ok], np.asarray(c["obs"])[ok][:, :3])
        assert np.abs(nd[ok, :3] - np.asarray(c["n"])[ok]).max() < 1e-10
        assert np.abs(nd[ok, 3] - c["d"][ok]).max() < 1e-10
        assert np.abs(co[ok] - c["coeff"][ok]).max() < 1e-10
        n_fit += len(c); n_ok += int(ok.sum())
    assert n_fit > 1000 and n_ok > 500


def test_host_fit_at_the_gate_edges(pf):
    F = np.load(os.path.join(HERE, "golden", "gate_edge.npz"))
    plane_res = float(F["plane_res"])
    n_tight = 0
    for b in range(int(F["n_batches"])):
        pts, query, margin, expect = F[f"pts{b}"], F[f"query{b}"], F[f"margin{b}"], F[f"expect{b}"]
        status, _, _, _ = pf(pts.reshape(len(pts), 15), query.astype(np.float64), IDENTITY, plane_res)
        sure = np.abs(margin) >= 1e-12
        assert np.array_equal(status[sure], expect[sure]), (b, margin[sure], status[sure], expect[sure])
        # the fixture's tighter clusters too: the closed form is at least as accurate as the factorisation it replaces
        assert np.array_equal(status, expect), (b, margin, status, expect)
        n_tight += int((~sure).sum())
    assert n_tight >= 20


def _clusters(rng, n):
    P = np.zeros((n, 5, 3), np.float32)
    for t in range(n):
        kind = rng.integers(3)
        if kind == 0:
            nrm, off = np.array([0, 0, 1.0]), -1.5           # the floor under the sensor, seen 70 m away
        elif kind == 1:
            nrm, off = np.array([1.0, 0, 0]), rng.uniform(5, 60)
        else:
            nrm = rng.normal(size=3); nrm /= np.linalg.norm(nrm); off = rng.uniform(-40, 40)
        u = np.cross(nrm, [0.3, 0.5, 0.8]); u /= np.linalg.norm(u); v = np.cross(nrm, u)
        ctr = nrm * off + u * rng.uniform(-70, 70) + v * rng.uniform(-70, 70)
        P[t] = (ctr + np.outer(rng.uniform(-0.35, 0.35, 5), u) + np.outer(rng.uniform(-0.35, 0.35, 5), v)
                + np.outer(rng.normal(0, 0.01, 5), nrm)).astype(np.float32)
    return P


def _ls_plane_longdouble(P):
    """x = argmin |A x + 1| through the centred normal equations in 80-bit arithmetic; returns n, d."""
    LD = np.longdouble
    A = P.astype(LD)
    m = A.sum(0) / LD(5)
    c = A - m
    S = c.T @ c
    # solve S z = m by Cramer
    def det3(M):
        return M[0, 0] * (M[1, 1] * M[2, 2] - M[1, 2] * M[2, 1]) - M[0, 1] * (M[1, 0] * M[2, 2] - M[1, 2] * M[2, 0]) + M[0, 2] * (M[1, 0] * M[2, 1] - M[1, 1] * M[2, 0])
    D = det3(S)
    z = np.zeros(3, LD)
    for k in range(3):
        Mk = S.copy(); Mk[:, k] = m
        z[k] = det3(Mk) / D
    x = -LD(5) * z / (LD(1) + LD(5) * (m @ z))
    nn = np.sqrt(x @ x)
    return (x / nn).astype(np.float64), float(1 / nn)


def test_closed_form_is_closer_to_the_80_bit_plane_than_pivoted_qr(pf):
    import scipy.linalg
    rng = np.random.default_rng(4)
    P = _clusters(rng, 3000)
    status, nd, _, _ = pf(P.reshape(-1, 15), P.mean(1).astype(np.float64), IDENTITY, 0.2)
    worst_c = worst_q = 0.0
    n = 0
    for t in range(len(P)):
        if status[t] in (3,):   # rejected by the PCA gate before the plane is formed
            continue
        nt, dt = _ls_plane_longdouble(P[t])
        A = P[t].astype(np.float64)
        q, r, piv = scipy.linalg.qr(A, mode="economic", pivoting=True)
        y = np.linalg.solve(r, q.T @ (-np.ones(5)))
        x = np.empty(3); x[piv] = y
        nq, dq = x / np.linalg.norm(x), 1 / np.linalg.norm(x)
        if status[t] == 0:
            worst_c = max(worst_c, np.abs(nd[t, :3] - nt).max(), abs(nd[t, 3] - dt) / dt)
            n += 1
        worst_q = max(worst_q, np.abs(nq - nt).max(), abs(dq - dt) / dt)
    assert n > 1000
    assert worst_c < 1e-12, worst_c           # measured 1.4e-13
    assert worst_c < worst_q or worst_q < 1e-13, (worst_c, worst_q)   # (pivoted QR on the un-centred A: ~1e-11)


def test_observability_short_cut_equals_the_arithmetic_as_written(pf):
    rng = np.random.default_rng(9)
    P = _clusters(rng, 20000)
    # near-ties of |n.axis|: planes whose normal sits on a diagonal of the sensor axes
    for t in range(0, 4000):
        nrm = np.array([1.0, 1.0, rng.choice([0.0, 1.0, 1e-7])]) * rng.choice([-1, 1], 3); nrm /= np.linalg.norm(nrm)
        u = np.cross(nrm, [0.3, 0.5, 0.8]); u /= np.linalg.norm(u); v = np.cross(nrm, u)
        ctr = nrm * rng.uniform(2, 30) + u * rng.uniform(-20, 20) + v * rng.uniform(-20, 20)
        P[t] = (ctr + np.outer(rng.uniform(-0.35, 0.35, 5), u) + np.outer(rng.uniform(-0.35, 0.35, 5), v)
                + np.outer(rng.normal(0, 0.004, 5), nrm)).astype(np.float32)
    pw = P.mean(1).astype(np.float64) + rng.normal(0, 0.05, (len(P), 3))
    for pose in (IDENTITY, np.array([1.0, -2.0, 0.5, 0.1, -0.2, 0.3, np.sqrt(1 - 0.14)])):
        s0, nd0, c0, o0 = pf(P.reshape(-1, 15), pw, pose, 0.2, as_written=False)
        s1, nd1, c1, o1 = pf(P.reshape(-1, 15), pw, pose, 0.2, as_written=True)
        assert np.array_equal(s0, s1) and np.array_equal(nd0, nd1) and np.array_equal(c0, c1)
        assert (s0 == 0).sum() > 5000
        assert np.array_equal(o0, o1), np.flatnonzero((o0 != o1).any(1))[:10]


def _oracle_fit(L, P, pw, plane_res):
    """what orc_plane_match (oracle/so_oracle.c, LidarSlam.cpp:749-844) does behind the neighbour search, on given clusters:
    PCA gate with the oracle's Jacobi eigen-solver, the 5x3 plane by column-pivoted Householder QR, inlier gate; -> status, n, d"""
    import ctypes as C
    f64p = C.POINTER(C.c_double)
    n = len(P)
    st = np.zeros(n, np.int32); nd = np.zeros((n, 4))
    pr = np.float32(plane_res)
    for t in range(n):
        A = P[t].astype(np.float64)
        mean = A.sum(0) / 5.0
        c = A - mean
        S = np.zeros((3, 3))
        for j in range(5):
            S += np.outer(c[j], c[j])
        ev = np.zeros(3); V = np.zeros(9)
        L.orc_eig3_sym(np.ascontiguousarray(S).ctypes.data_as(f64p), ev.ctypes.data_as(f64p), V.ctypes.data_as(f64p))
        if ev[0] < 1e-6 or ev[1] / ev[2] < 0.1:
            st[t] = 3; continue
        x = np.zeros(3)
        if not L.orc_plane_ls5(np.ascontiguousarray(A).ctypes.data_as(f64p), x.ctypes.data_as(f64p)):
            st[t] = 4; continue
        nrm = np.sqrt(x[0] * x[0] + x[1] * x[1] + x[2] * x[2])
        d = 1.0 / nrm; nn = x / nrm
        dist = np.abs(A @ nn + d)
        if (dist > float(pr) / 2.0).any():
            st[t] = 5; continue
        nd[t, :3] = nn; nd[t, 3] = d
    return st, nd


def test_planes_through_the_world_origin(oracle, pf):
    """LidarSlam.cpp:798-816 fits A x = -1: for a plane through the world origin |x| = 1 / offset diverges.  Clusters on planes with
    offsets <= 1e-3 / <= 0.05 / <= 0.5 m (and 5 - 60 m as the control), the product's closed form against the oracle's pivoted QR:
    MatchingResult equal, planes within 1e-9 (measured <= 3e-10 at the small offsets; the control stays below 1e-10)."""
    import ctypes as C
    L = oracle.lib()
    L.orc_eig3_sym.restype = None
    L.orc_plane_ls5.restype = C.c_int
    rng = np.random.default_rng(12)
    for lo, hi, tol in ((0.0, 1e-3, 1e-9), (0.0, 0.05, 1e-9), (0.0, 0.5, 1e-9), (5.0, 60.0, 1e-10)):
        n = 4000
        P = np.zeros((n, 5, 3), np.float32)
        for t in range(n):
            nrm = rng.normal(size=3); nrm /= np.linalg.norm(nrm)
            off = rng.uniform(lo, hi) * rng.choice([-1.0, 1.0])
            u = np.cross(nrm, [0.3, 0.5, 0.8]); u /= np.linalg.norm(u); v = np.cross(nrm, u)
            ctr = nrm * off + u * rng.uniform(-40, 40) + v * rng.uniform(-40, 40)
            P[t] = (ctr + np.outer(rng.uniform(-0.35, 0.35, 5), u) + np.outer(rng.uniform(-0.35, 0.35, 5), v)
                    + np.outer(rng.normal(0, 0.01, 5), nrm)).astype(np.float32)
        pw = P.mean(1).astype(np.float64) + rng.normal(0, 0.05, (n, 3))
        status, nd, _, _ = pf(P.reshape(-1, 15), pw, IDENTITY, 0.2)
        ost, ond = _oracle_fit(L, P, pw, 0.2)
        assert np.array_equal(status, ost), (lo, hi, np.flatnonzero(status != ost)[:10])
        ok = status == 0
        assert ok.sum() > 2000, (lo, hi, int(ok.sum()))
        worst = np.abs(nd[ok] - ond[ok]).max()
        print(f"plane offsets {lo} .. {hi} m: {int(ok.sum())} accepted, max |plane - oracle plane| {worst:.2e}")
        assert worst < tol, (lo, hi, worst)
