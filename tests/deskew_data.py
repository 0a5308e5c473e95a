"""Seeded inputs for the de-skew tests: a sweep of point_os::PointcloudXYZITR-like records and a pose
buffer (IMU orientations or VIO odometry) around it.  Test infrastructure only."""
import numpy as np
from scipy.spatial.transform import Rotation as R


def sweep(n, stride=32, time_off=20, sweep_s=0.1, seed=0, nan_every=0):
    """records [n, stride] uint8: float x y z at 0 4 8, float time at time_off, rising over the sweep (columns share a time)."""
    rng = np.random.default_rng(seed)
    d = rng.normal(size=(n, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
    xyz = (d * rng.uniform(1.0, 80.0, (n, 1))).astype(np.float32)
    cols = max(n // 64, 1)
    t = (np.minimum(np.arange(n) // 64, cols - 1) / cols * sweep_s).astype(np.float32)
    if nan_every:
        xyz[::nan_every, rng.integers(0, 3)] = np.nan
    rec = np.zeros((n, stride // 4), np.float32)
    rec[:, 0:3] = xyz
    if stride >= 32:
        rec[:, 3] = 1.0
        rec[:, 4] = rng.uniform(0, 255, n).astype(np.float32)
    rec[:, time_off // 4] = t
    return rec.view(np.uint8).reshape(n, stride).copy()


def pose_buffer(t0, rate_hz=200.0, before_s=0.02, after_s=0.13, seed=1, translate=True, flip_signs=False):
    """[m, 8]: time, position, quaternion (x y z w) of a smooth motion: ~1 rad/s about a wobbling axis, ~2 m/s"""
    rng = np.random.default_rng(seed)
    ts = t0 - before_s + np.arange(int((before_s + after_s) * rate_hz) + 1) / rate_hz + 1.234e-4
    w = rng.normal(0, 0.8, 3)
    rot = R.from_rotvec(np.outer(ts - t0, w) + 0.02 * np.sin(np.outer(ts - t0, [7.0, 11.0, 13.0])))
    q = rot.as_quat()
    if flip_signs:
        q[1::2] *= -1.0  # q and -q are the same rotation: slerp must not take the long way round
    v = rng.normal(0, 1.5, 3)
    pos = np.outer(ts - t0, v) + 0.05 * np.sin(np.outer(ts - t0, [5.0, 3.0, 9.0])) if translate else np.zeros((len(ts), 3))
    return np.concatenate([ts[:, None], pos, q], 1)


def field(rec, off):
    return rec[:, off:off + 4].copy().view(np.float32).reshape(-1)


def xyz_of(rec):
    return np.stack([field(rec, 0), field(rec, 4), field(rec, 8)], 1)


def scipy_deskew(rec, time_off, t0, poses, imu, T_i_l=None):
    """the same computation on scipy's Rotation / Slerp (float64 throughout): returns float64 xyz [n, 3]"""
    from scipy.spatial.transform import Slerp
    xyz = xyz_of(rec).astype(np.float64)
    ts = field(rec, time_off).astype(np.float64) + t0
    rots = R.from_quat(poses[:, 4:8])
    sl = Slerp(poses[:, 0], rots)

    def at(t):
        t = np.atleast_1d(t)
        tc = np.clip(t, poses[0, 0], poses[-1, 0])
        r = sl(tc)
        p = np.stack([np.interp(tc, poses[:, 0], poses[:, 1 + k]) for k in range(3)], 1)
        return r, (np.zeros_like(p) if imu else p)
    r0, p0 = at(t0)
    rc, pc = at(ts)
    r0 = R.from_quat(np.repeat(r0.as_quat(), len(ts), 0))
    # T_original_current = T0^-1 * Tc
    roc = r0.inv() * rc
    poc = r0.inv().apply(pc - p0)
    if imu:
        til = np.array([0, 0, 0, 0, 0, 0, 1.0]) if T_i_l is None else np.asarray(T_i_l, float)
        ril, pil = R.from_quat(til[3:]), til[:3]
        rli, pli = ril.inv(), -ril.inv().apply(pil)
        # T_l_i * Toc * T_i_l
        rf = rli * roc * ril
        pf = rli.apply(roc.apply(pil) + poc) + pli
    else:
        rf, pf = roc, poc
    out = rf.apply(xyz) + pf
    bad = ~np.isfinite(xyz).all(1)
    out[bad] = xyz[bad]
    return out
