"""-m gpu: so_icp_deskew_scan (featureExtraction::removePointDistortion on the device) against the CPU
restatement on the same records.  Both sides run the same fp64 operation sequence; the only functions whose last bit
may differ are acos / sin inside slerp (glibc on the host, the device math library on the GPU), so the float32 output
must be bit-identical for all but a sliver of the points and within one float32 spacing for those."""
import numpy as np
import pytest
from scipy.spatial.transform import Rotation as R

import deskew_data as dd

pytestmark = pytest.mark.gpu
T0 = 1.7e9 + 0.25


def close_in_ulps(a, b):
    """(fraction of bit-identical values, largest difference in units of the float32 spacing at that magnitude)"""
    a, b = a.reshape(-1), b.reshape(-1)
    same = a.view(np.uint32) == b.view(np.uint32)
    both_nan = np.isnan(a) & np.isnan(b)
    d = np.abs(a.astype(np.float64) - b.astype(np.float64))[~(same | both_nan)]
    ulp = np.spacing(np.maximum(np.abs(a), np.abs(b)).astype(np.float32)).astype(np.float64)[~(same | both_nan)]
    return (same | both_nan).mean(), (d / ulp).max() if len(d) else 0.0


@pytest.mark.parametrize("imu,stride,time_off,n_poses_rate", [(True, 32, 20, 200.0), (False, 32, 20, 200.0), (False, 16, 12, 50.0), (True, 32, 20, 8000.0)])
def test_deskew_matches_the_oracle(gpu_slam_factory, oracle, imu, stride, time_off, n_poses_rate):
    slam = gpu_slam_factory()
    n = 131072  # the BASELINE sweep: 128 x 1024
    rec = dd.sweep(n, stride, time_off, seed=21, nan_every=4099)
    poses = dd.pose_buffer(T0, rate_hz=n_poses_rate, seed=22, translate=not imu, flip_signs=True)
    if n_poses_rate > 5000:
        assert len(poses) > 512, "this case takes the kernel's global-memory table path"
    T_i_l = np.concatenate([[0.05, -0.02, 0.1], R.from_rotvec([0.01, -0.02, 0.5]).as_quat()]) if imu else None
    want, wstart, wbeyond = oracle.deskew(rec, time_off, T0, poses, imu, T_i_l)
    got, info = slam.deskew_scan(rec, time_off, T0, poses, imu, T_i_l)
    assert info.n_clamped == wbeyond == 0
    assert list(info.t_w_original_l) + list(info.q_w_original_l) == list(wstart), "the sweep-start frame is host arithmetic on both sides"
    keep = np.ones(stride, bool); keep[:12] = False
    assert np.array_equal(got[:, keep], rec[:, keep]), "only x y z are rewritten"
    frac, worst = close_in_ulps(dd.xyz_of(got), dd.xyz_of(want))
    assert frac > 0.999 and worst <= 1.0, (frac, worst)
    bad = ~np.isfinite(dd.xyz_of(rec)).all(1)
    assert bad.sum() > 10 and np.array_equal(got[bad], rec[bad]), "non-finite points are left alone (featureExtraction.cpp:293-295)"
    # and against scipy, as the oracle itself is checked
    ref = dd.scipy_deskew(rec, time_off, T0, poses, imu, T_i_l)
    assert np.abs(dd.xyz_of(got)[~bad].astype(np.float64) - ref[~bad]).max() < 1e-5


def test_deskew_edges_and_errors(gpu_slam_factory, oracle, soicp):
    slam = gpu_slam_factory()
    rec = dd.sweep(5000, seed=31)
    poses = dd.pose_buffer(T0, seed=32, after_s=0.05)  # ends inside the sweep: the tail has no successor in the buffer
    want, _, wbeyond = oracle.deskew(rec, 20, T0, poses, False, None)
    got, info = slam.deskew_scan(rec, 20, T0, poses, False, None)
    assert info.n_clamped == wbeyond > 0
    frac, worst = close_in_ulps(dd.xyz_of(got), dd.xyz_of(want))
    assert frac > 0.999 and worst <= 1.0
    # single pose, empty sweep
    got, info = slam.deskew_scan(rec, 20, T0, poses[:1], True, None)
    assert info.n_clamped == 5000 and np.allclose(dd.xyz_of(got), dd.xyz_of(rec), atol=2e-5)
    got, info = slam.deskew_scan(rec[:0], 20, T0, poses, True, None)
    assert got.shape == (0, 32) and info.n_clamped == 0
    # a motionless buffer is the identity to rounding (size-independent property)
    still = poses.copy(); still[:, 1:4] = [1.0, 2.0, 3.0]; still[:, 4:8] = still[0, 4:8]
    got, _ = slam.deskew_scan(rec, 20, T0, still, False, None)
    assert np.allclose(dd.xyz_of(got), dd.xyz_of(rec), atol=2e-5)
    # errors: unsorted buffer, time field outside the record, misaligned stride
    with pytest.raises(soicp.SoIcpError, match="increase strictly"):
        slam.deskew_scan(rec, 20, T0, poses[::-1], False, None)
    with pytest.raises(soicp.SoIcpError):
        slam.deskew_scan(rec, 30, T0, poses, False, None)
    with pytest.raises(soicp.SoIcpError):
        slam.deskew_scan(rec[:, :30], 20, T0, poses, False, None)


def test_deskew_on_records_resident_in_hbm(gpu_slam_factory, oracle):
    """so_icp_deskew_scan_dev: the caller owns the device buffer (here: hipMalloc through ctypes), the records are rewritten there"""
    import ctypes as C
    hip = C.CDLL("libamdhip64.so")
    hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    hip.hipFree.argtypes = [C.c_void_p]
    slam = gpu_slam_factory()
    rec = dd.sweep(30000, seed=41)
    poses = dd.pose_buffer(T0, seed=42)
    d = C.c_void_p()
    assert hip.hipMalloc(C.byref(d), rec.nbytes) == 0
    try:
        assert hip.hipMemcpy(d, rec.ctypes.data_as(C.c_void_p), rec.nbytes, 1) == 0  # hipMemcpyHostToDevice
        info = slam.deskew_scan_dev(d.value, rec.shape[0], rec.shape[1], 20, T0, poses, False, None)
        got = np.empty_like(rec)
        assert hip.hipMemcpy(got.ctypes.data_as(C.c_void_p), d, rec.nbytes, 2) == 0  # hipMemcpyDeviceToHost
    finally:
        hip.hipFree(d)
    host, hinfo = slam.deskew_scan(rec, 20, T0, poses, False, None)
    assert np.array_equal(got, host) and info.n_clamped == hinfo.n_clamped == 0, "the two entry points run the same kernel"
    want, _, _ = oracle.deskew(rec, 20, T0, poses, False, None)
    frac, worst = close_in_ulps(dd.xyz_of(got), dd.xyz_of(want))
    assert frac > 0.999 and worst <= 1.0


def test_transform_cloud_is_the_nodes_registered_scan_bit_for_bit(gpu_slam_factory):
    """so_icp_transform_cloud against the arithmetic laserMapping::publishTopic runs on the host (Eigen's quaternion * vector in
    fp64: uv = 2 u x v, v + w uv + u x uv; then + t, rounded to float; near-sensor points untouched; keep = farther than 0.1 m
    from the world origin) -- the same IEEE operations in the same order, so the result is bit-identical."""
    slam = gpu_slam_factory()
    rng = np.random.default_rng(5)
    n = 100000
    rec = np.zeros((n, 8), np.float32)
    rec[:, :3] = rng.normal(0, 20, (n, 3)); rec[:, 3] = 1.0; rec[:, 4] = rng.uniform(0, 255, n)
    T = np.concatenate([[3.0, -4.0, 0.5], R.from_rotvec([0.02, -0.03, 0.8]).as_quat()])
    rec[::1000, :3] = rng.uniform(-0.05, 0.05, (len(rec[::1000]), 3))                           # within 0.1 m of the sensor: untouched
    back = R.from_quat(T[3:]).inv().apply(rng.uniform(-0.05, 0.05, (len(rec[5::1000]), 3)) - T[:3])
    rec[5::1000, :3] = back.astype(np.float32)                                                 # land within 0.1 m of the world origin: dropped
    got, keep, nk = slam.transform_cloud(rec.view(np.uint8).reshape(n, 32), T)
    got = got.view(np.float32).reshape(n, 8)
    x, y, z = (rec[:, k].astype(np.float64) for k in range(3))
    qx, qy, qz, qw = T[3:]
    ux, uy, uz = qy * z - qz * y, qz * x - qx * z, qx * y - qy * x
    ux, uy, uz = ux + ux, uy + uy, uz + uz
    wx = (x + qw * ux + (qy * uz - qz * uy)) + T[0]; wy = (y + qw * uy + (qz * ux - qx * uz)) + T[1]; wz = (z + qw * uz + (qx * uy - qy * ux)) + T[2]
    near = (rec[:, 0] * rec[:, 0] + rec[:, 1] * rec[:, 1] + rec[:, 2] * rec[:, 2]).astype(np.float64) < 0.01
    want = np.where(near[:, None], rec[:, :3], np.stack([wx, wy, wz], 1).astype(np.float32))
    assert np.array_equal(got[:, :3].view(np.uint32), want.view(np.uint32)) and np.array_equal(got[:, 3:], rec[:, 3:])
    wkeep = (want[:, 0] * want[:, 0] + want[:, 1] * want[:, 1] + want[:, 2] * want[:, 2]).astype(np.float64) > 0.01
    assert np.array_equal(keep.astype(bool), wkeep) and nk == wkeep.sum() and near.sum() >= 100 and (~wkeep).sum() >= 150
