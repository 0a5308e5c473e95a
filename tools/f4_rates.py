"""Rates of the f4 rows on the GPU box (printed, copied into profiles/ by tools/collect_profiles.sh):
  * so_icp_deskew_scan on the 131 072-point sweep: host buffers in place (PCIe both ways) and records resident in HBM,
    next to the CPU restatement on one core;
  * the laser_mapping_node shell (adapter/node_driver) replaying serialised LaserFeature messages of the `small` scene."""
import os
import struct
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import ctypes as C  # noqa: E402

import cdr_py  # noqa: E402
import deskew_data as dd  # noqa: E402
import oracle_py  # noqa: E402
from superodom_amd import binding, synth  # noqa: E402


def main():
    T0 = 1.7e9 + 0.25
    slam = binding.LidarSlamGpu()
    rec = dd.sweep(131072, seed=21)
    poses = dd.pose_buffer(T0, seed=22, translate=False)
    for _ in range(3):
        slam.deskew_scan(rec, 20, T0, poses, True, None)
    info = binding.DeskewInfo()
    work = rec.copy()
    pp = np.ascontiguousarray(poses)
    t = time.perf_counter()
    reps = 50
    for _ in range(reps):
        slam.L.so_icp_deskew_scan(slam.h, work.ctypes.data_as(C.c_void_p), len(work), 32, 20, T0, pp.ctypes.data_as(C.POINTER(C.c_double)), len(pp), 1, None, C.byref(info))
    host_ms = 1e3 * (time.perf_counter() - t) / reps
    import torch
    d = torch.from_numpy(rec.copy()).cuda()
    torch.cuda.synchronize()
    for _ in range(3):
        slam.L.so_icp_deskew_scan_dev(slam.h, C.c_void_p(d.data_ptr()), len(rec), 32, 20, T0, pp.ctypes.data_as(C.POINTER(C.c_double)), len(pp), 1, None, C.byref(info))
    t = time.perf_counter()
    for _ in range(reps):
        slam.L.so_icp_deskew_scan_dev(slam.h, C.c_void_p(d.data_ptr()), len(rec), 32, 20, T0, pp.ctypes.data_as(C.POINTER(C.c_double)), len(pp), 1, None, C.byref(info))
    dev_ms = 1e3 * (time.perf_counter() - t) / reps
    if "--deskew-only" in sys.argv:  # the pass rocprofv3 runs over (tools/measure_all.sh)
        return
    t = time.perf_counter()
    oracle_py.deskew(rec, 20, T0, poses, True, None)
    cpu_ms = 1e3 * (time.perf_counter() - t)
    print(f"deskew 131072 points, {len(poses)} IMU poses: host buffers in place {host_ms:.3f} ms, resident {dev_ms:.3f} ms "
          f"(table upload + kernel + counter read-back), CPU restatement 1 core {cpu_ms:.1f} ms")

    sc = synth.Scene("small")
    n_frames = int(os.environ.get("F4_FRAMES", "24"))
    with tempfile.TemporaryDirectory() as tmp:
        fin, fout = os.path.join(tmp, "bag.bin"), os.path.join(tmp, "out.bin")
        with open(fin, "wb") as f:
            f.write(struct.pack("<ffiiiii", sc.plane_res, sc.plane_res / 2, 4, -1, 0, 0, n_frames))
            for k in range(n_frames):
                scan = np.ascontiguousarray(sc.scan(k), np.float32)
                tt = 100.0 + 0.1 * k
                stamp = (int(tt), int(round((tt - int(tt)) * 1e9)))
                m = cdr_py.default("LaserFeature")
                q = sc.gt_pose(k)[3:]
                m["initial_quaternion_x"], m["initial_quaternion_y"], m["initial_quaternion_z"], m["initial_quaternion_w"] = [float(v) for v in q]
                m["cloud_surface"] = cdr_py.cloud_msg(scan, stamp=stamp)
                m["cloud_nodistortion"] = cdr_py.cloud_msg(scan, stamp=stamp)
                m["cloud_corner"] = cdr_py.cloud_msg(scan[:64], stamp=stamp)
                m["cloud_realsense"] = cdr_py.cloud_msg(np.zeros((0, 3)), stamp=stamp)
                raw = cdr_py.encode("LaserFeature", m)
                f.write(struct.pack("<I", len(raw))); f.write(raw)
        r = subprocess.run([os.path.join(ROOT, "adapter", os.environ.get("F4_DRIVER", "node_driver")), fin, fout], capture_output=True, text=True, timeout=600)
        print(f"node shell, scene small ({len(scan)} surf points per frame, full-resolution cloud of the same size): {r.stderr.strip()} rc={r.returncode}")


if __name__ == "__main__":
    main()
