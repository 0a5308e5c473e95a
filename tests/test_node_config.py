"""CPU suite: the node's parameter surface without rclcpp (adapter/node_config.h): a ROS 2 parameter file
in the layout of the reference's config/*.yaml -> NodeConfig, with the defaults laserMapping::readParameters declares
(laserMapping.cpp:182-203).  Checked against PyYAML's reading of the same file; the reference's own files are parsed
too where /root/reference is present (this container), never on the GPU box."""
import glob
import os
import subprocess

import pytest
import yaml

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOOL = os.path.join(ROOT, "adapter", "wire_selftest")
DECLARED = {"mapping_line_resolution": 0.1, "mapping_plane_resolution": 0.2, "max_iterations": 4, "debug_view": False, "enable_ouster_data": False,
            "publish_only_feature_points": False, "max_surface_features": 2000, "velocity_failure_threshold": 30.0, "auto_voxel_size": True,
            "forget_far_chunks": False, "visual_confidence_factor": 1.0, "localization_mode": False,
            "init_x": 0.0, "init_y": 0.0, "init_z": 0.0, "init_roll": 0.0, "init_pitch": 0.0, "init_yaw": 0.0}


@pytest.fixture(scope="module")
def tool():
    if not os.path.exists(TOOL):
        import __graft_entry__
        __graft_entry__.build()
    return TOOL


def node_config(tool, path):
    r = subprocess.run([tool, "params", path], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0, r.stderr
    return dict(line.split("=", 1) for line in r.stdout.strip().split("\n"))


def expected(path):
    doc = yaml.safe_load(open(path))
    params = next(iter(doc.values()))["ros__parameters"]  # the first top-level entry ("/**" in the reference's files)
    lm = {**DECLARED, **params.get("laser_mapping_node", {})}
    return params, lm


def check(tool, path):
    got = node_config(tool, path)
    params, lm = expected(path)
    f32 = lambda v: __import__("numpy").float32(float(v))  # noqa: E731  NodeConfig keeps the resolutions and the start pose in float (printed with 9 digits)
    assert f32(got["lineRes"]) == f32(lm["mapping_line_resolution"]) and f32(got["planeRes"]) == f32(lm["mapping_plane_resolution"])
    assert int(got["max_iterations"]) == lm["max_iterations"] and int(got["max_surface_features"]) == lm["max_surface_features"]
    assert float(got["velocity_failure_threshold"]) == lm["velocity_failure_threshold"] and float(got["visual_confidence_factor"]) == lm["visual_confidence_factor"]
    for key, name in (("debug_view", "debug_view"), ("auto_voxel_size", "auto_voxel_size"), ("localization_mode", "localization_mode"),
                      ("forget_far_chunks", "forget_far_chunks"), ("enable_ouster_data", "enable_ouster_data")):
        assert bool(int(got[key])) == bool(lm[name]), key
    assert [f32(v) for v in got["init"].split()] == [f32(lm[k]) for k in ("init_x", "init_y", "init_z", "init_roll", "init_pitch", "init_yaw")]
    assert bool(int(got["use_imu_roll_pitch"])) == bool(params.get("use_imu_roll_pitch", False))
    assert got["world_frame"] == params.get("world_frame", "sensor_init") and got["sensor_frame"] == params.get("sensor_frame", "sensor")
    assert got["PROJECT_NAME"] == params.get("PROJECT_NAME", "") and got["map_dir"] == params.get("map_dir", "pointcloud_local.pcd")
    return got


def test_example_file_against_pyyaml(tool):
    got = check(tool, os.path.join(ROOT, "tests", "golden", "params_example.yaml"))
    assert got["PROJECT_NAME"] == "so#1" and got["sensor_frame"] == "sensor_link" and got["map_dir"] == "/data/maps/site.pcd"
    assert got["planeRes"] == "0.400000006" and got["auto_voxel_size"] == "0" and got["debug_view"] == "1" and got["init"].split()[0] == "1.5"


def test_defaults_are_the_declared_ones(tool, tmp_path):
    p = tmp_path / "empty.yaml"
    p.write_text("/**:\n  ros__parameters:\n    sensor: livox\n")
    got = check(tool, str(p))
    assert got["auto_voxel_size"] == "1" and got["max_iterations"] == "4" and got["max_surface_features"] == "2000"


def test_bad_files_are_errors(tool, tmp_path):
    for text, what in (("/**:\n  ros__parameters:\n    laser_mapping_node:\n        max_iterations: five\n", "not a number"),
                       ("/**:\n  ros__parameters:\n    laser_mapping_node:\n        auto_voxel_size: maybe\n", "not a bool"),
                       ("/**:\n  ros__parameters:\n    topics:\n      - a\n", "not supported"),
                       ("/**:\n  ros__parameters:\n\tsensor: x\n", "tab")):
        p = tmp_path / "bad.yaml"
        p.write_text(text)
        r = subprocess.run([tool, "params", str(p)], capture_output=True, text=True, timeout=60)
        assert r.returncode == 1 and what in r.stderr, (text, r.stderr)
    r = subprocess.run([tool, "params", str(tmp_path / "missing.yaml")], capture_output=True, text=True, timeout=60)
    assert r.returncode == 1 and "cannot open" in r.stderr


@pytest.mark.skipif(not os.path.isdir("/root/reference/super_odometry/config"), reason="the reference tree is only present in the development container")
def test_the_reference_parameter_files(tool):
    files = sorted(glob.glob("/root/reference/super_odometry/config/*.yaml"))
    assert len(files) >= 3
    for path in files:
        got = check(tool, path)
        if path.endswith("livox_mid360.yaml"):  # BASELINE configs[0]: planeRes 0.1, 4000 surface features, 5 iterations
            assert (got["planeRes"], got["max_surface_features"], got["max_iterations"]) == ("0.100000001", "4000", "5")
