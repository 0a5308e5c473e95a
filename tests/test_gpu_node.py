"""-m gpu: SURVEY 8(f) row f4 -- the laser_mapping_node shell (adapter/laser_mapping_soicp.{h,cpp}) replaying serialised
super_odometry_msgs/LaserFeature messages (adapter/node_driver, built by __graft_entry__.build()).  What it publishes is
checked against (a) the same frame sequence driven through the ctypes binding with the node's between-frame logic
restated on scipy (tests/node_ref.py), (b) the CPU oracle run as its own chain, (c) the ground truth of the synthetic
trajectory, and (d) the message contents the reference node fills (laserMapping.cpp:415-597)."""
import os
import struct
import subprocess

import numpy as np
import pytest
from scipy.spatial.transform import Rotation as R

import cdr_py
import node_ref
from superodom_amd import synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DRIVER = os.path.join(ROOT, "adapter", "node_driver")
P = "/super_odometry"


def make_frames(sc, n_frames, with_imu=True):
    """LaserFeature messages of a replay: surf cloud = the scan (32-byte pcl::PointXYZI records), full-resolution cloud = the
    scan + three points within 0.1 m of the sensor (the node drops those), a handful of corner points, IMU orientation =
    ground-truth orientation of the frame."""
    rng = np.random.default_rng(77)
    frames = []
    for k in range(n_frames):
        scan = np.ascontiguousarray(sc.scan(k), np.float32)
        full = np.concatenate([scan[::3], np.array([[0.01, 0.02, 0.0], [0, 0, 0.05], [-0.03, 0, 0]], np.float32)])
        corner = (scan[rng.integers(0, len(scan), 40)] + rng.normal(0, 0.02, (40, 3))).astype(np.float32)
        t = 100.0 + 0.1 * k
        stamp = (int(t), int(round((t - int(t)) * 1e9)))
        m = cdr_py.default("LaserFeature")
        m["header"] = {"stamp": {"sec": stamp[0], "nanosec": stamp[1]}, "frame_id": "sensor"}
        q = sc.gt_pose(k)[3:] if with_imu else np.zeros(4)
        m["initial_quaternion_x"], m["initial_quaternion_y"], m["initial_quaternion_z"], m["initial_quaternion_w"] = [float(v) for v in q]
        m["cloud_surface"] = cdr_py.cloud_msg(scan, stamp=stamp)
        m["cloud_nodistortion"] = cdr_py.cloud_msg(full, stamp=stamp)
        m["cloud_corner"] = cdr_py.cloud_msg(corner, stamp=stamp)
        m["cloud_realsense"] = cdr_py.cloud_msg(np.zeros((0, 3)), stamp=stamp)
        frames.append(dict(msg=m, scan=scan, full=full, corner=corner, time=stamp[0] + stamp[1] * 1e-9, imu=np.asarray(q, float)))
    return frames


def run_node(tmp_path, frames, plane_res, line_res, max_it, msf, auto_voxel=0, debug_view=0, params=None, prior=None, env=None):
    assert os.path.exists(DRIVER), "adapter/node_driver not built: run python __graft_entry__.py"
    fin, fout = tmp_path / "bag.bin", tmp_path / "out.bin"
    with open(fin, "wb") as f:
        f.write(struct.pack("<ffiiiii", plane_res, line_res, max_it, msf, auto_voxel, debug_view, len(frames)))
        for fr in frames:
            raw = cdr_py.encode("LaserFeature", fr["msg"])
            f.write(struct.pack("<I", len(raw))); f.write(raw)
    r = subprocess.run([DRIVER, str(fin), str(fout)] + ([str(params)] if params else []) + ([str(prior)] if prior else []), capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, **(env or {})))
    assert r.returncode == 0, r.stderr
    raw = open(fout, "rb").read()
    at, pubs = 0, []

    def blob():
        nonlocal at
        (k,) = struct.unpack_from("<I", raw, at)
        b = raw[at + 4:at + 4 + k]
        at += 4 + k
        return b
    while True:
        (frame,) = struct.unpack_from("<I", raw, at)
        at += 4
        if frame == 0xFFFFFFFF:
            break
        topic, typ, cdr = blob().decode(), blob().decode(), blob()
        pubs.append((frame, topic, typ, cdr))
    (failed,) = struct.unpack_from("<i", raw, at)
    at += 4
    err = blob().decode()
    return pubs, failed, err


def by_frame(pubs, n):
    out = [dict() for _ in range(n)]
    order = [[] for _ in range(n)]
    for frame, topic, typ, cdr in pubs:
        out[frame][topic] = cdr_py.decode(typ.split("/")[-1], cdr)
        order[frame].append(topic)
    return out, order


def pose_of(odom):
    p, q = odom["pose"]["pose"]["position"], odom["pose"]["pose"]["orientation"]
    return np.array([p["x"], p["y"], p["z"], q["x"], q["y"], q["z"], q["w"]])


def test_node_shell_replay_against_the_mirror_the_oracle_and_ground_truth(gpu_slam_factory, oracle, tmp_path):
    sc = synth.Scene("tiny")
    n_frames, max_it, line_res = 16, 4, sc.plane_res / 2
    frames = make_frames(sc, n_frames)
    pubs, failed, err = run_node(tmp_path, frames, sc.plane_res, line_res, max_it, -1)
    assert failed == 0, err
    msgs, order = by_frame(pubs, n_frames)
    # the registered scan through the device (so_icp_transform_cloud, the path of clouds >= 32 k points) publishes the same bytes
    pubs_dev, failed_dev, err = run_node(tmp_path, frames, sc.plane_res, line_res, max_it, -1, env={"SOICP_NODE_DEVICE_TRANSFORM_MIN": "1"})
    assert failed_dev == 0, err
    assert [(f, t, c) for f, t, _, c in pubs_dev if t.endswith("/registered_scan")] == [(f, t, c) for f, t, _, c in pubs if t.endswith("/registered_scan")]

    # (d) what is published, and in which order (laserMapping.cpp:415-597; LidarSlam.cpp:965-966 for the uncertainty topics)
    tail = [P + "/prediction_source", P + "/registered_scan", P + "/aft_mapped_to_init_incremental", P + "/laser_odometry", P + "/laser_odom_path", P + "/super_odometry_stats"]
    unc = [P + "uncertainty_" + a for a in ("X", "Y", "Z", "roll", "pitch", "yaw")]
    assert order[0] == tail
    for k in range(1, n_frames):
        assert order[k] == unc + tail, (k, order[k])

    # (a) the same sequence through the ctypes binding, between-frame logic on scipy
    slam = gpu_slam_factory(plane_res=sc.plane_res, line_res=line_res, max_surface_features=-1, max_iterations=max_it)
    slam.set_origin(np.zeros(3))
    mirror = node_ref.NodeMirror()
    # (b) the oracle as its own chain
    om = oracle.OracleMap(plane_res=sc.plane_res, line_res=line_res)
    om.set_origin(np.zeros(3))
    ocfg = oracle.default_config(max_iterations=max_it)
    omirror, prev_hist = node_ref.NodeMirror(), None
    # (c) ground truth in the node's world: sensor frame of frame 0 with the roll / pitch of its IMU orientation
    g0 = sc.gt_pose(0)
    W = R.from_quat(node_ref.extract_roll_pitch(g0[3:])) * R.from_quat(g0[3:]).inv()

    registered = 0
    for k, fr in enumerate(frames):
        odom, stats = msgs[k][P + "/laser_odometry"], msgs[k][P + "/super_odometry_stats"]
        pose = pose_of(odom)
        guess = mirror.initial_guess(fr["imu"])
        d, n, info = slam.prefilter_scan(fr["scan"], False, line_res, sc.plane_res)
        rc, mpose, st = slam.localization_dev(k > 0, guess, d, n, fr["time"])
        assert rc in (0, 2) and (rc == 2) == (k == 0)
        assert np.allclose(pose[:3], mpose[:3], atol=1e-9) and node_ref.same_rotation(pose[3:], mpose[3:], 1e-9), (k, pose, mpose)
        vel_b, ang_b = mirror.update(mpose, st.startup_count if rc == 0 else 0, fr["time"])
        tw = odom["twist"]["twist"]
        assert np.allclose([tw["linear"][a] for a in "xyz"], vel_b, atol=1e-7), (k, tw, vel_b)
        assert np.allclose([tw["angular"][a] for a in "xyz"], ang_b, atol=1e-7), (k, tw, ang_b)
        # header / frames / degeneracy flag
        assert odom["header"]["frame_id"] == "sensor_init" and odom["child_frame_id"] == "sensor"
        assert odom["header"]["stamp"] == fr["msg"]["cloud_nodistortion"]["header"]["stamp"]
        assert odom["pose"]["covariance"][0] == 0.0 and not any(odom["pose"]["covariance"][1:]) and not any(odom["twist"]["covariance"])
        assert pose_of(msgs[k][P + "/aft_mapped_to_init_incremental"]).tolist() == pose.tolist()
        path = msgs[k][P + "/laser_odom_path"]
        assert len(path["poses"]) == k + 1 and path["poses"][-1]["pose"] == odom["pose"]["pose"] and path["header"]["frame_id"] == "sensor_init"
        assert msgs[k][P + "/prediction_source"]["data"] == "IMU Only Orientation Prediction"
        # statistics message: padded to four iterations (laserMapping.cpp:590-593), counts, uncertainties
        assert stats["header"] == odom["header"] and len(stats["iterations"]) >= 4
        if rc == 0:
            registered += 1
            assert stats["n_iterations"] == st.n_iterations and 1 <= st.n_iterations <= max_it
            assert stats["laser_cloud_surf_from_map_num"] == st.laser_cloud_surf_from_map_num > 50
            assert stats["laser_cloud_surf_stack_num"] == n == len(oracle.voxel_grid(fr["scan"], sc.plane_res))
            assert stats["laser_cloud_corner_stack_num"] == len(oracle.voxel_grid(fr["corner"], line_res))
            assert stats["laser_cloud_corner_from_map_num"] == 0
            assert [stats["uncertainty_" + a] for a in ("x", "y", "z", "roll", "pitch", "yaw")] == list(st.uncertainty)
            assert [msgs[k][u]["data"] for u in unc] == [np.float32(v) for v in st.uncertainty]
            for j in range(st.n_iterations):
                assert stats["iterations"][j]["num_surf_from_scan"] == st.iterations[j].num_surf_from_scan
                assert np.isclose(stats["iterations"][j]["translation_norm"], st.iterations[j].translation_norm, atol=1e-9)  # (the mirror's guess differs in the last bits)
            assert np.isclose(stats["total_translation"], st.total_translation, atol=1e-9)
            assert stats["latency"] == 0 and stats["prediction_source"] == 0 and stats["plane_match_success"] == 0
        # registered scan: the full-resolution cloud in the world frame minus the points within 0.1 m of the sensor / origin
        reg = msgs[k][P + "/registered_scan"]
        assert reg["point_step"] == 32 and reg["header"]["frame_id"] == "sensor_init" and [f["offset"] for f in reg["fields"]] == [0, 4, 8, 16]
        far = fr["full"][(fr["full"].astype(np.float32) ** 2).sum(1) >= 0.01]
        want = (R.from_quat(pose[3:]).apply(far.astype(np.float64)) + pose[:3]).astype(np.float32)
        want = want[(want ** 2).sum(1) > 0.01]
        assert np.allclose(cdr_py.cloud_xyz(reg), want, atol=2e-6)

        # (b) oracle chain
        oguess = omirror.initial_guess(fr["imu"])
        surf = oracle.voxel_grid(fr["scan"], sc.plane_res)
        if k == 0:
            om.set_origin(oguess[:3]); om.transform_and_add(surf, oguess)
            opose, ostart = oguess, 0
        else:
            orc, opose, ost, _ = om.register(surf, oguess, ocfg, prev_obs_hist=prev_hist)
            assert orc == rc == 0
            prev_hist = np.array(ost.iters[ost.n_iterations - 1].obs_hist, np.int32)
            om.transform_and_add(surf, opose)
            ostart = 0
            assert ost.n_iterations == stats["n_iterations"]
            dt, dr = synth.pose_error(pose, opose)
            assert dt <= 1e-4 and dr <= 1e-4, (k, dt, dr)
        omirror.update(opose, ostart, fr["time"])
        assert slam.map_size() == om.size()

        # (c) ground truth
        g = sc.gt_pose(k)
        gt = np.concatenate([W.apply(g[:3] - g0[:3]), (W * R.from_quat(g[3:])).as_quat()])
        dt, dr = synth.pose_error(pose, gt)
        assert dt < 0.05 and dr < 0.01, (k, dt, dr)
    assert registered == n_frames - 1


def test_node_shell_auto_voxel_size_map_topics_and_missing_imu(gpu_slam_factory, tmp_path):
    """auto_voxel_size (laserMapping.cpp:603-636), the debug map topics (:437-462) and the constant-velocity prediction the
    node falls back to without an IMU orientation (initial_quaternion all zero: :383-413, 366-370)."""
    sc = synth.Scene("tiny")
    n_frames = 5
    frames = make_frames(sc, n_frames, with_imu=False)
    # the knobs come from a ROS 2 parameter file in the reference's layout (adapter/node_config.h); the bag header's are overridden
    params = tmp_path / "params.yaml"
    params.write_text("/**:\n  ros__parameters:\n    world_frame: \"sensor_init\"\n    laser_mapping_node:\n        mapping_line_resolution: 0.2\n"
                      "        mapping_plane_resolution: 0.4\n        max_iterations: 4\n        max_surface_features: 2000\n        debug_view: true\n"
                      "        # auto_voxel_size: declared default true (laserMapping.cpp:192)\n")
    pubs, failed, err = run_node(tmp_path, frames, 0.05, 0.05, 1, 7, auto_voxel=0, debug_view=0, params=params)
    assert failed == 0, err
    msgs, order = by_frame(pubs, n_frames)
    slam = gpu_slam_factory(plane_res=0.4, line_res=0.2, max_surface_features=2000, max_iterations=4)
    slam.set_origin(np.zeros(3))
    mirror = node_ref.NodeMirror()
    line_res, plane_res = 0.2, 0.4
    for k, fr in enumerate(frames):
        guess = mirror.initial_guess(fr["imu"])
        d, n, info = slam.prefilter_scan(fr["scan"], True, line_res, plane_res)
        line_res, plane_res = info.line_res, info.plane_res
        assert (line_res, plane_res) == (np.float32(0.1), np.float32(0.2)), "a 14 m room: mean|x| mean|y| mean|z| < 25"
        rc, mpose, st = slam.localization_dev(k > 0, guess, d, n, fr["time"])
        pose = pose_of(msgs[k][P + "/laser_odometry"])
        assert np.allclose(pose[:3], mpose[:3], atol=1e-9) and node_ref.same_rotation(pose[3:], mpose[3:], 1e-9), (k, pose, mpose)
        mirror.update(mpose, st.startup_count if rc == 0 else 0, fr["time"])
        stats = msgs[k][P + "/super_odometry_stats"]
        assert np.isclose(stats["average_distance"], info.average_distance, rtol=1e-12) and 0 < stats["average_distance"] < 25
        if k >= 1:
            assert msgs[k][P + "/prediction_source"]["data"] == "Using Constant Velocity Prediction"
            assert stats["laser_cloud_surf_stack_num"] == n  # the sub-sampler acts inside the registration, not on the stack
    # frameCount is 1-based when publishTopic runs: the surround map goes out on the 5th frame; the whole map on the 20th (not reached)
    assert [P + "/laser_cloud_surround" in o for o in order] == [False, False, False, False, True]
    assert not any(P + "/laser_cloud_map" in o for o in order)
    sur = msgs[4][P + "/laser_cloud_surround"]
    xyz = cdr_py.cloud_xyz(sur)
    ref = slam.export_map(only_5x5=True, pos=list(st.pos_in_localmap))
    assert sur["header"]["frame_id"] == "sensor_init" and len(xyz) == len(ref) > 1000
    assert np.array_equal(xyz[np.lexsort(xyz.T)], ref[np.lexsort(ref.T)])


def test_node_shell_when_the_map_has_too_few_features(gpu_slam_factory, tmp_path):
    """"Not enough features for optimization" (LidarSlam.cpp:113-116): the first frame seeds only 30 points, so every later frame
    returns early -- no registration, no insert; the node still publishes the guess as the pose and the statistics of
    prepareOptimizationState (counts, cleared iteration list padded to four, uncertainties of an empty histogram)."""
    sc = synth.Scene("tiny")
    frames = make_frames(sc, 3)
    few = frames[0]["scan"][:30]
    frames[0]["msg"]["cloud_surface"] = cdr_py.cloud_msg(few, stamp=(100, 0))
    pubs, failed, err = run_node(tmp_path, frames, sc.plane_res, sc.plane_res / 2, 4, -1)
    assert failed == 0, err
    msgs, order = by_frame(pubs, 3)
    mirror = node_ref.NodeMirror()
    for k, fr in enumerate(frames):
        guess = mirror.initial_guess(fr["imu"])
        pose = pose_of(msgs[k][P + "/laser_odometry"])
        assert np.allclose(pose[:3], guess[:3], atol=1e-12) and node_ref.same_rotation(pose[3:], guess[3:], 1e-12), k
        mirror.update(pose, 0, fr["time"])
        st = msgs[k][P + "/super_odometry_stats"]
        assert st["n_iterations"] == 0 and len(st["iterations"]) == 4 and all(it["num_surf_from_scan"] == 0 for it in st["iterations"])
        if k:
            assert 0 < st["laser_cloud_surf_from_map_num"] <= 30 and st["laser_cloud_surf_stack_num"] > 1000
            assert [st["uncertainty_" + a] for a in ("x", "y", "z", "roll", "pitch", "yaw")] == [0.0] * 6
            assert [msgs[k][P + "uncertainty_" + a]["data"] for a in ("X", "Y", "Z", "roll", "pitch", "yaw")] == [0.0] * 6


@pytest.mark.parametrize("source", ["points", "pcd_file"])
def test_node_shell_localization_mode_against_a_prior_map(tmp_path, source):
    """localization_mode (laserMapping.cpp:161-171, 305-313): the prior map is loaded before the first frame, the first pose comes
    from init_x .. init_yaw (tf2 setRPY), the map's frame is the world frame -- the poses must follow the ground truth of the
    synthetic trajectory directly -- and the prior cloud goes out on /overall_map with every 20th frame.  The map arrives as
    points, or -- like the reference -- as the .pcd file the parameter `map_dir` names (adapter/pcd_io.h)."""
    sc = synth.Scene("tiny")
    n_frames = 20
    frames = make_frames(sc, n_frames)
    g0 = sc.gt_pose(0)
    roll, pitch, yaw = R.from_quat(g0[3:]).as_euler("xyz")
    params = tmp_path / "loc.yaml"
    params.write_text(f"/**:\n  ros__parameters:\n    laser_mapping_node:\n        mapping_line_resolution: {sc.plane_res / 2}\n"
                      f"        mapping_plane_resolution: {sc.plane_res}\n        max_iterations: 4\n        max_surface_features: -1\n"
                      f"        auto_voxel_size: false\n        localization_mode: true\n        init_x: {float(g0[0])!r}\n        init_y: {float(g0[1])!r}\n"
                      f"        init_z: {float(g0[2])!r}\n        init_roll: {float(roll)!r}\n        init_pitch: {float(pitch)!r}\n        init_yaw: {float(yaw)!r}\n")
    if source == "points":
        prior = tmp_path / "prior.f32"
        np.ascontiguousarray(sc.map_points, np.float32).tofile(prior)
    else:
        prior = None
        pcd = tmp_path / "pointcloud_local.pcd"
        mp = np.ascontiguousarray(sc.map_points, np.float32)
        with open(pcd, "wb") as f:
            f.write((f"# .PCD v0.7 - Point Cloud Data file format\nVERSION 0.7\nFIELDS x y z intensity\nSIZE 4 4 4 4\nTYPE F F F F\nCOUNT 1 1 1 1\n"
                     f"WIDTH {len(mp)}\nHEIGHT 1\nVIEWPOINT 0 0 0 1 0 0 0\nPOINTS {len(mp)}\nDATA binary\n").encode())
            f.write(np.c_[mp, np.zeros(len(mp), np.float32)].astype("<f4").tobytes())
        with open(params, "a") as f:
            f.write(f'    map_dir: "{pcd}"\n')  # (declared at the node's top level, next to the laser_mapping_node block: laserMapping.cpp:183-203)
    pubs, failed, err = run_node(tmp_path, frames, 0.05, 0.05, 1, 7, params=params, prior=prior)
    assert failed == 0, err
    msgs, order = by_frame(pubs, n_frames)
    for k in range(n_frames):
        pose = pose_of(msgs[k][P + "/laser_odometry"])
        dt, dr = synth.pose_error(pose, sc.gt_pose(k))
        # frame 0 is the configured start pose (float parameters: 1e-7), the others are registered against prior map + scans
        assert dt < (1e-6 if k == 0 else 0.02) and dr < (1e-6 if k == 0 else 0.005), (k, dt, dr)
        if k:
            assert msgs[k][P + "/super_odometry_stats"]["laser_cloud_surf_from_map_num"] > 5000
    assert [P + "/overall_map" in o for o in order] == [False] * 19 + [True]
    assert [P + "/laser_cloud_map" in o for o in order] == [False] * 19 + [True]
    over = msgs[19][P + "/overall_map"]
    assert over["header"]["frame_id"] == "sensor_init" and over["header"]["stamp"] == msgs[19][P + "/laser_odometry"]["header"]["stamp"]
    assert np.array_equal(cdr_py.cloud_xyz(over), np.ascontiguousarray(sc.map_points, np.float32))
    whole = cdr_py.cloud_xyz(msgs[19][P + "/laser_cloud_map"])
    assert len(whole) >= len(sc.map_points)  # prior map + the 20 inserted scans, voxel-filtered
