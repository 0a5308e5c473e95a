"""CPU suite: the wire formats of SURVEY 8(f) row f4 (adapter/wire/): every message type laser_mapping_node receives or
publishes, serialised by the C++ codec (adapter/wire_selftest, built by __graft_entry__.build()), against
 (1) hand-derived byte vectors of the CDR specification,
 (2) an independent schema-driven Python codec (tests/cdr_py.py) in both directions,
 (3) hostile input (truncated buffers, sequence lengths beyond the buffer): an error, never a crash.
No rmw implementation exists in the image, so this is parity with the published format, not with a ROS 2 binary."""
import os
import struct
import subprocess

import numpy as np
import pytest

import cdr_py

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOOL = os.path.join(ROOT, "adapter", "wire_selftest")
TYPES = ["String", "Float32", "Header", "PointCloud2", "Odometry", "Path", "OptimizationStats", "LaserFeature"]


@pytest.fixture(scope="module")
def tool():
    if not os.path.exists(TOOL):
        import __graft_entry__
        __graft_entry__.build()
    assert os.path.exists(TOOL)
    return TOOL


def emit(tool, t, tmp_path):
    out = tmp_path / f"{t}.cdr"
    subprocess.run([tool, "emit", t, str(out)], check=True, timeout=60)
    return out.read_bytes()


def roundtrip(tool, t, raw, tmp_path):
    fin, fout = tmp_path / "in.cdr", tmp_path / "out.cdr"
    fin.write_bytes(raw)
    r = subprocess.run([tool, "roundtrip", t, str(fin), str(fout)], capture_output=True, text=True, timeout=60)
    return r, (fout.read_bytes() if r.returncode == 0 else None)


def test_hand_derived_byte_vectors(tool, tmp_path):
    # std_msgs/String "hello": header, length 6 (NUL included), bytes, NUL
    assert emit(tool, "String", tmp_path) == bytes.fromhex("00010000" "06000000" "68656c6c6f00")
    assert emit(tool, "Float32", tmp_path) == bytes.fromhex("00010000") + struct.pack("<f", 0.75)
    # std_msgs/Header: int32 sec, uint32 nanosec, string
    assert emit(tool, "Header", tmp_path) == bytes.fromhex("00010000") + struct.pack("<iI", 1700000000, 123456789) + struct.pack("<I", 12) + b"sensor_init\x00"
    # nav_msgs/Odometry: after "sensor_init\0" (offset 8+4+12 = 24) the child_frame_id length is already 4-aligned; after
    # "sensor\0" (24+4+7 = 35) the first float64 needs 5 bytes of padding to offset 40
    raw = emit(tool, "Odometry", tmp_path)
    body = raw[4:]
    assert body[:8] == struct.pack("<iI", 7, 250000000)
    assert body[8:24] == struct.pack("<I", 12) + b"sensor_init\x00"
    assert body[24:35] == struct.pack("<I", 7) + b"sensor\x00" and body[35:40] == b"\x00" * 5
    assert struct.unpack_from("<7d", body, 40) == (1.5, -2.5, 3.25, 0.1, 0.2, 0.3, 0.9)
    assert struct.unpack_from("<36d", body, 96) == tuple(float(i) for i in range(36))
    assert struct.unpack_from("<6d", body, 96 + 288) == (0.5, 0.25, 0.125, -1.0, -2.0, -3.0)
    assert len(body) == 96 + 288 + 48 + 288


@pytest.mark.parametrize("t", TYPES)
def test_cpp_writer_is_read_by_the_python_codec_and_back(tool, tmp_path, t):
    raw = emit(tool, t, tmp_path)
    msg = cdr_py.decode(t, raw)                       # consumes every byte (asserted inside)
    assert cdr_py.encode(t, msg) == raw               # the Python writer produces the same bytes from the decoded fields
    r, again = roundtrip(tool, t, raw, tmp_path)      # C++ reader -> C++ writer
    assert r.returncode == 0 and again == raw


def test_decoded_field_values(tool, tmp_path):
    st = cdr_py.decode("OptimizationStats", emit(tool, "OptimizationStats", tmp_path))
    assert [st[k] for k in ("laser_cloud_surf_from_map_num", "laser_cloud_corner_from_map_num", "laser_cloud_surf_stack_num", "laser_cloud_corner_stack_num")] == [11, 12, 13, 14]
    assert (st["total_translation"], st["latency"], st["n_iterations"], st["average_distance"]) == (0.5, 2.5, 2, 30.5)
    assert [st[k] for k in ("plane_match_success", "plane_unknown", "prediction_source")] == [21, 27, 1]
    assert [it["num_surf_from_scan"] for it in st["iterations"]] == [1000.0, 1001.0] and st["iterations"][1]["header"]["frame_id"] == "q"
    lf = cdr_py.decode("LaserFeature", emit(tool, "LaserFeature", tmp_path))
    assert lf["imu_preintegration_reset_id"] == -5 and lf["initial_quaternion_w"] == 1.0 and lf["imu_quaternion_w"] == 0.99
    assert [lf[k]["width"] for k in ("cloud_nodistortion", "cloud_corner", "cloud_surface", "cloud_realsense")] == [4, 1, 3, 0]
    xyz = cdr_py.cloud_xyz(lf["cloud_surface"])
    assert np.array_equal(xyz, np.array([[0, 0, 0], [1, 0.5, -0.25], [2, 1.0, -0.5]], np.float32))
    path = cdr_py.decode("Path", emit(tool, "Path", tmp_path))
    assert [p["header"]["frame_id"] for p in path["poses"]] == ["abc", "ab"] and path["poses"][1]["pose"]["position"]["z"] == 3.0


def test_python_written_messages_pass_through_the_cpp_codec(tool, tmp_path):
    rng = np.random.default_rng(5)
    lf = cdr_py.default("LaserFeature")
    lf["header"] = {"stamp": {"sec": 100, "nanosec": 999999999}, "frame_id": "sensor"}
    lf["initial_quaternion_w"] = 1.0
    for k, (n, step) in {"cloud_nodistortion": (257, 32), "cloud_corner": (3, 32), "cloud_surface": (1001, 16), "cloud_realsense": (0, 32)}.items():
        lf[k] = cdr_py.cloud_msg(rng.normal(0, 10, (n, 3)), stamp=(100, 5), point_step=step)
    raw = cdr_py.encode("LaserFeature", lf)
    r, again = roundtrip(tool, "LaserFeature", raw, tmp_path)
    assert r.returncode == 0 and again == raw
    # a big-endian stream is read (and re-written little endian)
    st = cdr_py.decode("OptimizationStats", emit(tool, "OptimizationStats", tmp_path))
    r, again = roundtrip(tool, "OptimizationStats", cdr_py.encode("OptimizationStats", st, big_endian=True), tmp_path)
    assert r.returncode == 0 and again == cdr_py.encode("OptimizationStats", st)
    # strings of every length modulo 4 keep the alignment of what follows
    for k in range(9):
        h = {"stamp": {"sec": k, "nanosec": k}, "frame_id": "f" * k}
        it = {"header": h, "translation_norm": 1.0, "rotation_norm": 2.0, "num_surf_from_scan": 3.0, "num_corner_from_scan": 4.0}
        raw = cdr_py.encode("IterationStats", it)
        r, again = roundtrip(tool, "IterationStats", raw, tmp_path)
        assert r.returncode == 0 and again == raw, k


def test_hostile_input_is_an_error_not_a_crash(tool, tmp_path):
    raw = emit(tool, "LaserFeature", tmp_path)
    for cut in (0, 3, 4, 11, 40, len(raw) // 2, len(raw) - 1):
        r, _ = roundtrip(tool, "LaserFeature", raw[:cut], tmp_path)
        assert r.returncode == 1 and "cdr:" in r.stderr, (cut, r.returncode, r.stderr)
    # a sequence length far beyond the buffer (the PointField count of the first cloud)
    body = bytearray(raw)
    lf = cdr_py.decode("LaserFeature", raw)
    at = raw.index(struct.pack("<II", lf["cloud_nodistortion"]["height"], lf["cloud_nodistortion"]["width"])) + 8
    body[at:at + 4] = struct.pack("<I", 0x7FFFFFFF)
    r, _ = roundtrip(tool, "LaserFeature", bytes(body), tmp_path)
    assert r.returncode == 1 and "sequence longer than the buffer" in r.stderr
    r, _ = roundtrip(tool, "String", b"\x00\x07\x00\x00\x01\x00\x00\x00\x00", tmp_path)  # a parameter-list encapsulation id
    assert r.returncode == 1 and "encapsulation" in r.stderr


def _random_message(t, rng, depth=0):
    """a random value of schema type t (strings of every length, sequences of 0..3 elements, payloads of 0..200 bytes)"""
    if t in cdr_py.PRIM:
        if t == "bool":
            return bool(rng.integers(0, 2))
        if t.startswith("float"):
            v = float(rng.normal(0, 1e3))
            return float(np.float32(v)) if t == "float32" else v
        info = np.iinfo(np.dtype(t))
        return int(rng.integers(info.min, info.max, dtype=np.dtype(t), endpoint=True))
    if t == "string":
        return "".join(chr(c) for c in rng.integers(32, 127, int(rng.integers(0, 12))))
    if t == "uint8[]":
        return bytes(rng.integers(0, 256, int(rng.integers(0, 200)), dtype=np.uint8))
    if t.endswith("[]"):
        return [_random_message(t[:-2], rng, depth + 1) for _ in range(int(rng.integers(0, 4)))]
    if t.endswith("]"):
        base, k = t[:-1].split("[")
        return [_random_message(base, rng, depth + 1) for _ in range(int(k))]
    return {name: _random_message(ft, rng, depth + 1) for name, ft in cdr_py.SCHEMAS[t]}


@pytest.mark.parametrize("t", ["LaserFeature", "OptimizationStats", "Odometry", "Path", "PointCloud2"])
def test_random_messages_survive_the_cpp_codec_and_truncations_never_crash_it(tool, tmp_path, t):
    rng = np.random.default_rng(__import__("zlib").crc32(t.encode()))
    msgs = [cdr_py.encode(t, _random_message(t, rng)) for _ in range(200)]
    # every second frame is followed by a truncated copy of itself (cut at a random byte): those must be refused
    frames, want = [], []
    for k, raw in enumerate(msgs):
        frames.append(raw); want.append(raw)
        if k % 2:
            cut = int(rng.integers(0, len(raw)))
            frames.append(raw[:cut]); want.append(None)
    fin, fout = tmp_path / "many.bin", tmp_path / "many.out"
    fin.write_bytes(b"".join(struct.pack("<I", len(f)) + f for f in frames))
    r = subprocess.run([tool, "roundtrip-many", t, str(fin), str(fout)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    out, at = fout.read_bytes(), 0
    for k, w in enumerate(want):
        (ln,) = struct.unpack_from("<I", out, at)
        at += 4
        if w is None:
            # a cut can land exactly on the end of the last member only when nothing was removed: never here (cut < len)
            assert ln == 0xFFFFFFFF, (t, k, "a truncated message was accepted")
        else:
            assert ln == len(w) and out[at:at + ln] == w, (t, k)
            at += ln
    assert at == len(out)
