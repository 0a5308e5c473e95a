"""CPU-side tests of the product library: the C ABI loads and exports every symbol include/so_icp.h
declares, compute entry points fail loudly without a device, and the host logic (LocalMap restatement,
canonical order, shard ownership, LM controller) agrees with the oracle.  No compute kernels run here."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from helpers import noisy_planes_cloud
from superodom_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol(soicp):
    hdr = open(os.path.join(ROOT, "include", "so_icp.h")).read()
    declared = sorted(set(re.findall(r"\b(so_icp_[a-z0-9_]+)\s*\(", hdr)))
    assert len(declared) >= 30
    L = soicp.load()
    for name in declared:
        assert hasattr(L, name), f"libsoicp.so does not export {name}"
    assert sorted(declared) == sorted(soicp.EXPORTED), "binding.EXPORTED must list exactly the header's functions"
    assert L.so_icp_abi_version() == 4


def test_struct_layouts_match_the_header(soicp, tmp_path):
    src = tmp_path / "sz.c"
    src.write_text('#include <stdio.h>\n#include "so_icp.h"\nint main(){printf("%zu %zu %zu %zu %zu %zu\\n",sizeof(so_icp_stats),'
                   'sizeof(so_icp_iter_stats),sizeof(so_icp_config),sizeof(so_icp_timing),sizeof(so_icp_sums),sizeof(so_icp_lm_state));return 0;}')
    exe = tmp_path / "sz"
    import subprocess
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    got = list(map(int, subprocess.check_output([str(exe)]).split()))
    want = [C.sizeof(soicp.Stats), C.sizeof(soicp.IterStats), C.sizeof(soicp.Config), C.sizeof(soicp.Timing), C.sizeof(soicp.Sums), C.sizeof(soicp.LmState)]
    assert got == want


def test_no_cpu_fallback(soicp):
    L = soicp.load()
    if L.so_icp_device_available():
        pytest.skip("a GPU is present; this test checks the no-device behaviour")
    with pytest.raises(soicp.SoIcpError, match="no HIP device"):
        soicp.LidarSlamGpu(device_id=0)
    host = soicp.LidarSlamGpu(device_id=-1, plane_res=0.2)  # host-only: map bookkeeping works ...
    host.add_surf_point_cloud(noisy_planes_cloud(2000, np.random.default_rng(0)))
    with pytest.raises(soicp.SoIcpError, match="no CPU fallback"):  # ... compute does not
        host.register(np.zeros((10, 3), np.float32), np.array([0, 0, 0, 0, 0, 0, 1.0]))
    with pytest.raises(soicp.SoIcpError, match="no CPU fallback"):
        host.nearest_k_search_surf(np.zeros((1, 3), np.float32))


def test_local_map_matches_oracle(soicp, oracle):
    rng = np.random.default_rng(1)
    host = soicp.LidarSlamGpu(device_id=-1, plane_res=0.2)
    om = oracle.OracleMap(plane_res=0.2)
    t0 = np.array([130.0, -80.0, 3.0])
    assert list(host.set_origin(t0)) == list(om.set_origin(t0))
    assert list(host.shift_map(t0)) == list(om.shift(t0))
    assert list(host.origin()) == list(om.origin())
    for step in range(4):  # incremental inserts re-filter the touched cubes (LocalMap.h:617-641)
        pts = np.concatenate([noisy_planes_cloud(6000, rng, offset=(t0[0] + dx, t0[1] + dy, 0)) for dx in (-30, 20) for dy in (-10, 35)])
        a = host.add_surf_point_cloud(pts); b = om.add_surf(pts)
        assert a == b
        assert host.map_size() == om.size()
        A = host.export_map(); B = om.export()
        assert np.array_equal(A[np.lexsort(A.T)], B[np.lexsort(B.T)]), "VoxelGrid restatements must agree bit for bit"
    pos = host.shift_map(t0); assert list(pos) == list(om.shift(t0))
    assert host.count_5x5(pos) == om.count_5x5(pos)
    # roll the window and compare again
    t1 = t0 + np.array([400.0, -260.0, 0.0])
    assert list(host.shift_map(t1)) == list(om.shift(t1))
    assert list(host.origin()) == list(om.origin()) and host.map_size() == om.size()
    # canonical order: ascending cube index, then cell (z,y,x): exported points are grouped by cube
    A = host.export_map()
    o = host.origin()
    ci = (np.floor((A.astype(np.float64) + 25.0) / 50.0).astype(int) + o)
    lin = ci[:, 0] + 21 * ci[:, 1] + 441 * ci[:, 2]
    assert (np.diff(lin) >= 0).all()


def test_cells_per_cube_covers_the_gate_radius(soicp):
    for res in (0.1, 0.2, 0.4, 0.8, 1.5):
        nc, cell = soicp.cells_per_cube(res)
        r_max = np.sqrt(float(np.float32(3) * np.float32(res)))
        assert 1 <= nc <= 64 and abs(cell * nc - 50.0) < 1e-9
        assert cell >= r_max * 1.004, "one cell must cover the neighbour-distance gate sqrt(3*planeRes)"


def test_shard_ownership_is_a_partition_and_roughly_balanced(soicp):
    sc = synth.Scene("tiny")
    gt = sc.gt_pose(0); R = synth.quat_to_R(gt[3:])
    world_pts = (sc.scan(0) @ R.T + gt[:3]).astype(np.float32)
    origin = np.array([10, 10, 5], np.int32)
    for W in (2, 4, 8):
        owners = np.array([soicp.shard_owner_of_point(p, origin, 0.2, W) for p in world_pts[::4]])
        assert owners.min() >= 0 and owners.max() < W
        frac = np.bincount(owners, minlength=W) / len(owners)
        if W == 2:  # brick-hash ownership: query density is very non-uniform (near-field floor), so only a loose bound
            assert frac.min() > 0.2, frac
    assert soicp.shard_owner_of_point(world_pts[0], origin, 0.2, 1) == 0
    # so_icp_shard_histogram (the N > 1 bench line's per-rank query counts): the same rule applied to a sensor-frame scan under a pose
    for W in (2, 8):
        h = soicp.shard_histogram(sc.scan(0), gt, origin, 0.2, W)
        owners = np.array([soicp.shard_owner_of_point(p, origin, 0.2, W) for p in world_pts])
        assert h.sum() == len(world_pts) and np.array_equal(h, np.bincount(owners, minlength=W))
    # outside the window -> rank 0 counts it
    assert soicp.shard_owner_of_point(np.array([1e5, 0, 0], np.float32), origin, 0.2, 8) == 0


def _drive(soicp, oracle, corrs, x0, plane_res, max_it=4):
    drv = soicp.LmDriver()
    cost, JtJ, Jtr, cnt = oracle.evaluate(corrs, x0, plane_res)
    more, nxt = drv.begin(x0, soicp.LmDriver.sums(cost, cnt, Jtr, JtJ), max_it)
    evals = 1
    while more:
        cost, JtJ, Jtr, cnt = oracle.evaluate(corrs, nxt, plane_res)
        more, nxt = drv.feed(soicp.LmDriver.sums(cost, cnt, Jtr, JtJ))
        evals += 1
    return drv.result() + (evals,)


def test_lm_controller_matches_oracle_lm(soicp, oracle):
    """Product LM (6x6 normal equations + Cholesky, fed with fused sums) vs oracle LM (QR on the
    stacked A x 6 Jacobian): same iteration counts, same accept/reject decisions, poses to 1e-9."""
    sc = synth.Scene("tiny")
    om = oracle.OracleMap(plane_res=sc.plane_res); om.add_surf(sc.map_points)
    for i in (0, 3, 9):
        scan, guess = sc.scan(i), sc.guess(i)
        corrs = om.plane_match(guess, scan)
        assert (corrs["status"] == 0).sum() > 1000
        pose_o, st_o = oracle.lm_solve(corrs, guess, sc.plane_res)
        pose_p, st_p, evals = _drive(soicp, oracle, corrs, guess, sc.plane_res)
        assert st_p.lm_iterations == st_o.lm_iterations
        assert st_p.num_successful_steps == st_o.num_successful_steps
        assert st_p.termination == st_o.termination
        assert st_p.num_surf_from_scan == st_o.num_surf
        assert evals == 1 + st_o.lm_iterations, "one fused evaluation per LM iteration + the initial one"
        dt, dr = synth.pose_error(pose_p, pose_o)
        assert dt < 1e-9 and dr < 1e-9, (dt, dr)
        assert abs(st_p.final_cost - st_o.final_cost) < 1e-9 * max(1, st_o.final_cost)
        # second solve from the converged pose: the num_successful_steps == 1 rule of LidarSlam.cpp:141 hinges on this
        corrs2 = om.plane_match(pose_o, scan)
        pose_o2, st_o2 = oracle.lm_solve(corrs2, pose_o, sc.plane_res)
        pose_p2, st_p2, _ = _drive(soicp, oracle, corrs2, pose_o, sc.plane_res)
        assert (st_p2.lm_iterations, st_p2.num_successful_steps, st_p2.termination) == (st_o2.lm_iterations, st_o2.num_successful_steps, st_o2.termination)


def test_lm_controller_no_residuals(soicp):
    drv = soicp.LmDriver()
    x0 = np.array([1, 2, 3, 0, 0, 0, 1.0])
    more, nxt = drv.begin(x0, soicp.LmDriver.sums(0.0, 0.0, np.zeros(6), np.zeros((6, 6))), 4)
    pose, st = drv.result()
    assert more == 0 and st.termination == 4 and np.array_equal(pose, x0)


def test_registration_error_matches_numpy_oracle(soicp, oracle):
    """so_icp_registration_error (EstimateRegistrationError, LS.cpp:854-889) is host arithmetic: covariance = inverse of
    the loss-corrected J^T J, eigen-analysis of its 3x3 blocks -- against the numpy oracle."""
    rng = np.random.default_rng(7)
    for trial in range(20):
        J = rng.normal(size=(200, 6)) * np.array([1, 1, 0.3, 5, 5, 2.0]) * (0.05 if trial == 19 else 1.0)
        H = J.T @ J
        st = soicp.Stats()
        for i in range(36):
            st.JtJ[i] = H.flat[i]
        e = soicp.registration_error(st)
        o = oracle.registration_error(H)
        assert e is not None
        cov = np.array(e.covariance).reshape(6, 6)
        assert np.allclose(cov, o["covariance"], rtol=1e-10, atol=1e-14)
        assert np.isclose(e.position_error, o["position_error"], rtol=1e-10)
        assert np.isclose(e.orientation_error_deg, o["orientation_error_deg"], rtol=1e-10)
        assert np.isclose(e.pos_inverse_condition_num, o["pos_inverse_condition_num"], rtol=1e-9)
        assert np.isclose(e.ori_inverse_condition_num, o["ori_inverse_condition_num"], rtol=1e-9)
        # eigenvector signs are free
        assert abs(abs(np.dot(np.array(e.position_error_direction), o["position_error_direction"])) - 1) < 1e-8
        assert abs(abs(np.dot(np.array(e.orientation_error_direction), o["orientation_error_direction"])) - 1) < 1e-8
    sing = soicp.Stats()  # all-zero J^T J: singular
    assert soicp.registration_error(sing) is None


def test_cpp_adapter_builds_and_fails_loudly_without_a_device(soicp, tmp_path):
    """The compiled reference-side adapter (adapter/, built by __graft_entry__.build()): on a box without a GPU the C++
    driver must report the missing device the way the node would (an exception caught by process(), laserMapping.cpp:788-790,
    here: exit code 1 + the library's message) -- never a silent CPU path."""
    import struct
    import subprocess
    driver = os.path.join(ROOT, "adapter", "adapter_driver")
    if not os.path.exists(driver):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "adapter"),
                               os.path.join(ROOT, "adapter", "lidar_slam_soicp.cpp"), os.path.join(ROOT, "adapter", "adapter_driver.cpp"), "-o", driver,
                               "-L", os.path.join(ROOT, "superodom_amd", "lib"), "-lsoicp", "-Wl,-rpath,$ORIGIN/../superodom_amd/lib",
                               "-Wl,-rpath-link,/opt/rocm/lib"])
    if soicp.load().so_icp_device_available():
        pytest.skip("a GPU is present; tests/test_gpu_adapter.py runs the driver for real")
    fin = tmp_path / "in.bin"
    pts = np.random.default_rng(0).random((64, 3)).astype(np.float32)
    with open(fin, "wb") as f:
        f.write(struct.pack("<ifii", 1, 0.2, 4, -1))
        f.write(struct.pack("<i", len(pts))); f.write(np.array([0, 0, 0, 0, 0, 0, 1.0]).tobytes()); f.write(struct.pack("<d", 0.0)); f.write(pts.tobytes())
    r = subprocess.run([driver, str(fin), str(tmp_path / "out.bin")], capture_output=True, text=True, timeout=120)
    assert r.returncode == 1 and "no HIP device" in r.stderr, (r.returncode, r.stderr)


def test_node_shell_fails_loudly_without_a_device(soicp, tmp_path):
    """adapter/node_driver (the laser_mapping_node shell): without a GPU the first thing the node does --
    LocalMap::setOrigin in initializationParam, laserMapping.cpp:159 -- reports the missing device; nothing is published."""
    import struct
    import subprocess
    driver = os.path.join(ROOT, "adapter", "node_driver")
    if not os.path.exists(driver):
        import __graft_entry__
        __graft_entry__.build()
    if soicp.load().so_icp_device_available():
        pytest.skip("a GPU is present; tests/test_gpu_node.py runs the shell for real")
    fin, fout = tmp_path / "bag.bin", tmp_path / "out.bin"
    with open(fin, "wb") as f:
        f.write(struct.pack("<ffiiiii", 0.2, 0.1, 4, -1, 0, 0, 0))
    r = subprocess.run([driver, str(fin), str(fout)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 1 and "no HIP device" in r.stderr, (r.returncode, r.stderr)
    assert os.path.getsize(fout) == 0
