"""CPU suite: the restatement of featureExtraction::removePointDistortion (oracle/so_oracle.c orc_deskew,
featureExtraction.cpp:223-314) against scipy's Rotation / Slerp and against closed forms.  The reference ships no
vectors for this function and Eigen is not in the image: parity unpinned upstream, as for the rest of the path."""
import numpy as np
import pytest
from scipy.spatial.transform import Rotation as R

import deskew_data as dd

T0 = 1.7e9 + 0.25  # a ROS time stamp: seconds since the epoch (float64 keeps ~2e-7 s there)


@pytest.mark.parametrize("imu,stride,time_off,flip", [(True, 32, 20, False), (False, 32, 20, False), (False, 16, 12, True), (True, 32, 20, True)])
def test_oracle_against_scipy(oracle, imu, stride, time_off, flip):
    rec = dd.sweep(20000, stride, time_off, seed=3, nan_every=997)
    poses = dd.pose_buffer(T0, seed=4, translate=not imu, flip_signs=flip)
    T_i_l = np.concatenate([[0.05, -0.02, 0.1], R.from_rotvec([0.01, -0.02, 0.5]).as_quat()]) if imu else None
    out, start, beyond = oracle.deskew(rec, time_off, T0, poses, imu, T_i_l)
    want = dd.scipy_deskew(rec, time_off, T0, poses, imu, T_i_l)
    got = dd.xyz_of(out).astype(np.float64)
    ok = np.isfinite(want).all(1)
    assert beyond == 0 and ok.sum() > 19900
    # float32 output of ~80 m coordinates: half an ulp is 4e-6; the interpolation itself agrees to ~1e-9
    assert np.abs(got[ok] - want[ok]).max() < 1e-5
    # everything except x y z is untouched, and so are the non-finite points
    keep = np.ones(stride, bool); keep[:12] = False
    assert np.array_equal(out[:, keep], rec[:, keep])
    assert np.array_equal(out[~ok].view(np.uint8), rec[~ok].view(np.uint8))
    # sweep-start sensor frame: interpolated start pose (times T_i_l for an IMU buffer)
    from scipy.spatial.transform import Slerp
    r0 = Slerp(poses[:, 0], R.from_quat(poses[:, 4:8]))([T0])[0]
    if imu:
        assert (R.from_quat(start[3:]).inv() * (r0 * R.from_quat(T_i_l[3:]))).magnitude() < 1e-12
        assert np.allclose(start[:3], r0.apply(T_i_l[:3]), atol=1e-12)
    else:
        assert (R.from_quat(start[3:]).inv() * r0).magnitude() < 1e-12
        assert np.allclose(start[:3], [np.interp(T0, poses[:, 0], poses[:, 1 + k]) for k in range(3)], atol=1e-9)


def test_closed_forms(oracle):
    n = 4096
    rec = dd.sweep(n, seed=9)
    before = dd.xyz_of(rec)
    # a buffer that does not move: T_final = T_l_i * I * T_i_l, the points stay where they are (to rounding)
    still = np.zeros((8, 8)); still[:, 0] = T0 - 0.02 + 0.03 * np.arange(8); still[:, 4:8] = R.from_rotvec([0.1, 0.2, 0.3]).as_quat()
    til = np.concatenate([[0.3, 0.1, -0.2], R.from_rotvec([0.4, 0.1, -0.3]).as_quat()])
    out, _, beyond = oracle.deskew(rec, 20, T0, still, True, til)
    assert beyond == 0 and np.allclose(dd.xyz_of(out), before, rtol=0, atol=2e-5)
    # constant rate about z, IMU buffer, identity extrinsic: a point measured tau after the start turns by w * tau
    w = 2.0
    ts = T0 - 0.01 + 0.005 * np.arange(40)
    spin = np.zeros((40, 8)); spin[:, 0] = ts; spin[:, 4:8] = R.from_rotvec(np.outer(ts - T0, [0, 0, w])).as_quat()
    out, start, _ = oracle.deskew(rec, 20, T0, spin, True, None)
    # (the reference adds the float point time to the epoch start time in float64: at 1.7e9 s that keeps 2.4e-7 s, which is
    # 4e-5 m at 80 m and 2 rad/s -- the closed form uses the time the arithmetic actually sees)
    tau = (dd.field(rec, 20).astype(np.float64) + T0) - T0
    want = R.from_rotvec(np.outer(tau, [0, 0, w])).apply(before.astype(np.float64))
    assert np.abs(dd.xyz_of(out) - want).max() < 1e-5
    assert R.from_quat(start[3:]).magnitude() < 1e-9
    # pure translation at constant velocity (VIO buffer): the point moves by v * tau in the start frame
    v = np.array([3.0, -1.0, 0.5])
    lin = np.zeros((40, 8)); lin[:, 0] = ts; lin[:, 1:4] = np.outer(ts - T0, v); lin[:, 7] = 1.0
    out, _, _ = oracle.deskew(rec, 20, T0, lin, False, None)
    assert np.abs(dd.xyz_of(out) - (before + np.outer(tau, v))).max() < 1e-5


def test_edges_of_the_buffer(oracle):
    rec = dd.sweep(2048, seed=11)
    poses = dd.pose_buffer(T0, seed=12, before_s=0.02, after_s=0.05)  # ends inside the 0.1 s sweep
    out, _, beyond = oracle.deskew(rec, 20, T0, poses, False, None)
    tau = dd.field(rec, 20).astype(np.float64) + T0
    assert beyond == int((tau >= poses[-1, 0]).sum()) > 0
    # points beyond the buffer get the last pose; points before it the first
    late = poses.copy(); late[:, 0] += 1.0   # the whole buffer lies after the sweep: every point gets the first pose = the start pose
    out, start, beyond = oracle.deskew(rec, 20, T0, late, False, None)
    assert beyond == 0 and np.allclose(dd.xyz_of(out), dd.xyz_of(rec), atol=2e-5)
    assert np.array_equal(start, np.concatenate([late[0, 1:4], late[0, 4:8]]))
    # two neighbours that are the same rotation up to the last bits: Eigen's slerp switches to a plain lerp (|dot| >= 1 - eps)
    twin = poses.copy(); twin[:, 4:8] = twin[0, 4:8]
    out, _, _ = oracle.deskew(rec, 20, T0, twin, True, None)
    assert np.allclose(dd.xyz_of(out), dd.xyz_of(rec), atol=2e-5)
    # a single-entry buffer
    out, start, beyond = oracle.deskew(rec, 20, T0, poses[:1], False, None)
    assert np.allclose(dd.xyz_of(out), dd.xyz_of(rec), atol=2e-5) and beyond == 2048
