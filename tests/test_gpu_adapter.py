"""-m gpu: the compiled C++ caller of the boundary.  adapter/adapter_driver (g++, links libsoicp.so; built by
__graft_entry__.build()) drives adapter/lidar_slam_soicp.{h,cpp} -- a class with LidarSLAM's public surface
(LidarSlam.h:292-293; fields read back at laserMapping.cpp:734-741, 581-596) -- the way performSLAMOptimization does:
seed with the first frame, then Localization() per frame, 32-byte pcl::PointXYZI clouds, public fields read back.
The same sequence through the ctypes binding must give the same bits."""
import os
import struct
import subprocess

import numpy as np
import pytest

from superodom_amd import synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DRIVER = os.path.join(ROOT, "adapter", "adapter_driver")


def test_cpp_adapter_matches_the_python_driven_run_bit_for_bit(gpu_slam_factory, soicp, tmp_path):
    assert os.path.exists(DRIVER), "adapter/adapter_driver not built: run python __graft_entry__.py"
    sc = synth.Scene("tiny")
    n_frames, max_it = 5, 4
    scans = [np.ascontiguousarray(sc.scan(i), np.float32) for i in range(n_frames)]
    guesses = [sc.gt_pose(0)] + [sc.guess(i) for i in range(1, n_frames)]
    times = [0.1 * i for i in range(n_frames)]
    fin, fout = tmp_path / "in.bin", tmp_path / "out.bin"
    with open(fin, "wb") as f:
        f.write(struct.pack("<ifii", n_frames, sc.plane_res, max_it, -1))
        for i in range(n_frames):
            f.write(struct.pack("<i", len(scans[i])))
            f.write(np.asarray(guesses[i], np.float64).tobytes()); f.write(struct.pack("<d", times[i])); f.write(scans[i].tobytes())
    r = subprocess.run([DRIVER, str(fin), str(fout)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    raw = open(fout, "rb").read()
    rec = struct.Struct("<5i7d2i7di I".replace(" ", ""))
    frames = [rec.unpack_from(raw, k * rec.size) for k in range(n_frames)]
    off = n_frames * rec.size
    (n_map,) = struct.unpack_from("<Q", raw, off)
    cpp_map = np.frombuffer(raw, np.float32, 3 * n_map, off + 8).reshape(-1, 3)

    slam = gpu_slam_factory(plane_res=sc.plane_res, line_res=sc.plane_res / 2, max_surface_features=-1, max_iterations=max_it)
    last_pos = None
    for i in range(n_frames):
        rc, pose, st = slam.localization(i > 0, guesses[i], scans[i], times[i])
        fr = frames[i]
        assert fr[0] == rc, (i, fr[0], rc)
        assert np.array_equal(np.array(fr[5:12]), pose), (i, "T_w_lidar must agree bit for bit")
        if rc != 0:
            continue
        assert fr[1] == st.startup_count and list(fr[2:5]) == list(st.pos_in_localmap)
        assert fr[12] == st.n_iterations and fr[13] == st.laser_cloud_surf_from_map_num
        assert fr[14] == st.total_translation and list(fr[15:21]) == list(st.uncertainty)
        assert fr[21] == st.iterations[st.n_iterations - 1].num_surf_from_scan
        assert (fr[22] & ~soicp.FLAG_STAGED_SCAN) == (st.flags & ~soicp.FLAG_STAGED_SCAN)
        if i >= 2:
            assert fr[22] & soicp.FLAG_STAGED_SCAN, "frames announced with StageNextScan ran on the staged copy"
        last_pos = list(st.pos_in_localmap)
    assert sum(1 for fr in frames if fr[0] == 0) >= 3
    assert np.array_equal(cpp_map, slam.export_map(only_5x5=True, pos=last_pos)), "get5x5LocalMap after the sequence"
