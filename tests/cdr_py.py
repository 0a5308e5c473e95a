"""Checker for adapter/wire/cdr.h: an independent, schema-driven CDR (PLAIN_CDR v1, little endian) codec.
The schemas below are transcribed from the .msg files (super_odometry_msgs/msg/*.msg in the reference, the ROS 2 Humble
common_interfaces for the rest); the C++ side has one hand-written put/get per message, so a slip in either shows up
as a byte difference.  Test infrastructure only."""
import struct

PRIM = {"int8": "b", "uint8": "B", "bool": "?", "int16": "h", "uint16": "H", "int32": "i", "uint32": "I", "int64": "q", "uint64": "Q",
        "float32": "f", "float64": "d"}

SCHEMAS = {
    "Time": [("sec", "int32"), ("nanosec", "uint32")],
    "Header": [("stamp", "Time"), ("frame_id", "string")],
    "String": [("data", "string")],
    "Float32": [("data", "float32")],
    "PointField": [("name", "string"), ("offset", "uint32"), ("datatype", "uint8"), ("count", "uint32")],
    "PointCloud2": [("header", "Header"), ("height", "uint32"), ("width", "uint32"), ("fields", "PointField[]"), ("is_bigendian", "bool"),
                    ("point_step", "uint32"), ("row_step", "uint32"), ("data", "uint8[]"), ("is_dense", "bool")],
    "Point": [("x", "float64"), ("y", "float64"), ("z", "float64")],
    "Vector3": [("x", "float64"), ("y", "float64"), ("z", "float64")],
    "Quaternion": [("x", "float64"), ("y", "float64"), ("z", "float64"), ("w", "float64")],
    "Pose": [("position", "Point"), ("orientation", "Quaternion")],
    "Twist": [("linear", "Vector3"), ("angular", "Vector3")],
    "PoseWithCovariance": [("pose", "Pose"), ("covariance", "float64[36]")],
    "TwistWithCovariance": [("twist", "Twist"), ("covariance", "float64[36]")],
    "PoseStamped": [("header", "Header"), ("pose", "Pose")],
    "Odometry": [("header", "Header"), ("child_frame_id", "string"), ("pose", "PoseWithCovariance"), ("twist", "TwistWithCovariance")],
    "Path": [("header", "Header"), ("poses", "PoseStamped[]")],
    "IterationStats": [("header", "Header"), ("translation_norm", "float64"), ("rotation_norm", "float64"),
                       ("num_surf_from_scan", "float64"), ("num_corner_from_scan", "float64")],
    "OptimizationStats": [("header", "Header"), ("laser_cloud_surf_from_map_num", "int32"), ("laser_cloud_corner_from_map_num", "int32"),
                          ("laser_cloud_surf_stack_num", "int32"), ("laser_cloud_corner_stack_num", "int32"),
                          ("total_translation", "float64"), ("total_rotation", "float64"), ("translation_from_last", "float64"),
                          ("rotation_from_last", "float64"), ("time_elapsed", "float64"), ("latency", "float64"), ("n_iterations", "int32"),
                          ("average_distance", "float64"), ("uncertainty_x", "float64"), ("uncertainty_y", "float64"), ("uncertainty_z", "float64"),
                          ("uncertainty_roll", "float64"), ("uncertainty_pitch", "float64"), ("uncertainty_yaw", "float64"),
                          ("plane_match_success", "int32"), ("plane_no_enough_neighbor", "int32"), ("plane_neighbor_too_far", "int32"),
                          ("plane_badpca_structure", "int32"), ("plane_invalid_numerical", "int32"), ("plane_mse_too_large", "int32"),
                          ("plane_unknown", "int32"), ("prediction_source", "int32"), ("iterations", "IterationStats[]")],
    "LaserFeature": [("header", "Header"), ("sensor", "int64"), ("imu_available", "int64"), ("odom_available", "int64"),
                     ("imu_quaternion_x", "float64"), ("imu_quaternion_y", "float64"), ("imu_quaternion_z", "float64"), ("imu_quaternion_w", "float64"),
                     ("initial_pose_x", "float64"), ("initial_pose_y", "float64"), ("initial_pose_z", "float64"),
                     ("initial_quaternion_x", "float64"), ("initial_quaternion_y", "float64"), ("initial_quaternion_z", "float64"),
                     ("initial_quaternion_w", "float64"), ("imu_preintegration_reset_id", "int64"),
                     ("cloud_nodistortion", "PointCloud2"), ("cloud_corner", "PointCloud2"), ("cloud_surface", "PointCloud2"),
                     ("cloud_realsense", "PointCloud2")],
}


class _W:
    def __init__(self, big=False):
        self.b = bytearray(b"\x00\x00\x00\x00" if big else b"\x00\x01\x00\x00")
        self.e = ">" if big else "<"

    def prim(self, t, v):
        size = struct.calcsize(PRIM[t])
        while (len(self.b) - 4) % size:
            self.b.append(0)
        self.b += struct.pack(self.e + PRIM[t], v)

    def value(self, t, v):
        if t in PRIM:
            self.prim(t, v)
        elif t == "string":
            raw = v.encode()
            self.prim("uint32", len(raw) + 1)
            self.b += raw + b"\x00"
        elif t == "uint8[]":
            self.prim("uint32", len(v))
            self.b += bytes(v)
        elif t.endswith("[]"):
            self.prim("uint32", len(v))
            for x in v:
                self.value(t[:-2], x)
        elif t.endswith("]"):
            base, k = t[:-1].split("[")
            assert len(v) == int(k)
            for x in v:
                self.value(base, x)
        else:
            for name, ft in SCHEMAS[t]:
                self.value(ft, v[name])


class _R:
    def __init__(self, raw):
        assert raw[0] == 0 and raw[1] in (0, 1), "not PLAIN_CDR"
        self.e = "<" if raw[1] == 1 else ">"
        self.raw, self.at = raw, 4

    def prim(self, t):
        size = struct.calcsize(PRIM[t])
        while (self.at - 4) % size:
            self.at += 1
        (v,) = struct.unpack_from(self.e + PRIM[t], self.raw, self.at)
        self.at += size
        return v

    def value(self, t):
        if t in PRIM:
            return self.prim(t)
        if t == "string":
            k = self.prim("uint32")
            s = bytes(self.raw[self.at:self.at + k - 1]).decode()
            self.at += k
            return s
        if t == "uint8[]":
            k = self.prim("uint32")
            d = bytes(self.raw[self.at:self.at + k])
            self.at += k
            return d
        if t.endswith("[]"):
            return [self.value(t[:-2]) for _ in range(self.prim("uint32"))]
        if t.endswith("]"):
            base, k = t[:-1].split("[")
            return [self.value(base) for _ in range(int(k))]
        return {name: self.value(ft) for name, ft in SCHEMAS[t]}


def default(t):
    if t in PRIM:
        return False if t == "bool" else (0.0 if t.startswith("float") else 0)
    if t == "string":
        return ""
    if t == "uint8[]":
        return b""
    if t.endswith("[]"):
        return []
    if t.endswith("]"):
        base, k = t[:-1].split("[")
        return [default(base) for _ in range(int(k))]
    d = {name: default(ft) for name, ft in SCHEMAS[t]}
    if t == "Quaternion":
        d["w"] = 1.0
    return d


def encode(t, msg, big_endian=False):
    w = _W(big_endian)
    w.value(t, msg)
    return bytes(w.b)


def decode(t, raw):
    r = _R(raw)
    v = r.value(t)
    assert r.at == len(raw), (t, r.at, len(raw))
    return v


def cloud_msg(xyz, frame="sensor", stamp=(0, 0), point_step=32, intensity=None):
    """pcl::toROSMsg of a pcl::PointXYZI cloud: x y z at 0 4 8, intensity at 16, 32-byte points"""
    import numpy as np
    xyz = np.asarray(xyz, np.float32).reshape(-1, 3)
    n = len(xyz)
    rec = np.zeros((n, point_step // 4), np.float32)
    rec[:, 0:3] = xyz
    if point_step >= 32:
        rec[:, 3] = 1.0
        rec[:, 4] = np.arange(n, dtype=np.float32) if intensity is None else intensity
    m = default("PointCloud2")
    m["header"] = {"stamp": {"sec": stamp[0], "nanosec": stamp[1]}, "frame_id": frame}
    m["height"], m["width"] = 1, n
    fields = [("x", 0), ("y", 4), ("z", 8)] + ([("intensity", 16)] if point_step >= 32 else [])
    m["fields"] = [{"name": a, "offset": o, "datatype": 7, "count": 1} for a, o in fields]
    m["point_step"], m["row_step"], m["data"], m["is_dense"] = point_step, point_step * n, rec.tobytes(), True
    return m


def cloud_xyz(msg):
    import numpy as np
    offs = {f["name"]: f["offset"] for f in msg["fields"]}
    n = msg["width"] * msg["height"]
    raw = np.frombuffer(msg["data"], np.uint8).reshape(n, msg["point_step"]) if n else np.zeros((0, msg["point_step"]), np.uint8)
    cols = [raw[:, offs[a]:offs[a] + 4].copy().view(np.float32).reshape(-1) for a in "xyz"]
    return np.stack(cols, 1) if n else np.zeros((0, 3), np.float32)
