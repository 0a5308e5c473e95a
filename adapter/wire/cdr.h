// cdr.h -- the serialised form of the messages in messages.h: OMG CDR (DDS-XTypes "PLAIN_CDR", version 1), little endian,
// which is what every ROS 2 Humble rmw implementation (Fast DDS, Cyclone DDS) puts on the wire and what rosbag2 stores:
//   4-byte encapsulation header {0x00, 0x01, 0x00, 0x00} (CDR_LE, no options), then the members in declaration order;
//   a primitive of size s is aligned to s bytes counted from the first byte AFTER the header; string = uint32 length
//   INCLUDING the terminating NUL, the bytes, the NUL; sequence<T> = uint32 element count, then the elements; fixed arrays
//   are the elements alone; bool and uint8 are one byte; nested structs add no padding of their own.
// A big-endian stream (header {0x00, 0x00, ..}) is accepted by the reader; the writer always produces little endian.
// With this a maintainer without rosidl-generated code (a bag converter, a generic subscription handing over
// rclcpp::SerializedMessage buffers) feeds the node shim and reads its output; with ROS 2 present the typed
// publishers / subscriptions are used and this file is not needed.
// Not pinned against an rmw implementation here (none in the image): the layout above is the published specification,
// tests/test_wire_formats.py checks it against an independent Python codec and hand-derived byte vectors.
#pragma once
#include <cstring>
#include <stdexcept>

#include "messages.h"

namespace so_wire {

class CdrWriter {
 public:
  CdrWriter() { buf_ = {0x00, 0x01, 0x00, 0x00}; }
  std::vector<uint8_t> take() { return std::move(buf_); }
  const std::vector<uint8_t>& bytes() const { return buf_; }

  template <typename T> void prim(T v) {
    align(sizeof(T));
    const size_t at = buf_.size();
    buf_.resize(at + sizeof(T));
    std::memcpy(buf_.data() + at, &v, sizeof(T));  // the hosts this runs on are little endian (asserted in cdr_selftest)
  }
  void boolean(bool v) { buf_.push_back(v ? 1 : 0); }
  void string(const std::string& s) {
    prim<uint32_t>((uint32_t)s.size() + 1);
    buf_.insert(buf_.end(), s.begin(), s.end());
    buf_.push_back(0);
  }
  void bytes_seq(const std::vector<uint8_t>& d) {
    prim<uint32_t>((uint32_t)d.size());
    buf_.insert(buf_.end(), d.begin(), d.end());
  }

 private:
  void align(size_t s) {
    const size_t pos = buf_.size() - 4;
    const size_t pad = (s - pos % s) % s;
    buf_.insert(buf_.end(), pad, 0);
  }
  std::vector<uint8_t> buf_;
};

class CdrReader {
 public:
  CdrReader(const uint8_t* p, size_t n) : p_(p), n_(n) {
    if (n < 4) throw std::runtime_error("cdr: buffer shorter than the encapsulation header");
    if (p[0] != 0x00 || (p[1] != 0x00 && p[1] != 0x01)) throw std::runtime_error("cdr: not a PLAIN_CDR (version 1) encapsulation");
    swap_ = p[1] == 0x00;  // big-endian stream on a little-endian host
    at_ = 4;
  }
  size_t consumed() const { return at_; }

  template <typename T> T prim() {
    align(sizeof(T));
    need(sizeof(T));
    uint8_t tmp[sizeof(T)];
    std::memcpy(tmp, p_ + at_, sizeof(T));
    if (swap_) for (size_t i = 0; i < sizeof(T) / 2; ++i) std::swap(tmp[i], tmp[sizeof(T) - 1 - i]);
    T v;
    std::memcpy(&v, tmp, sizeof(T));
    at_ += sizeof(T);
    return v;
  }
  bool boolean() { need(1); return p_[at_++] != 0; }
  std::string string() {
    const uint32_t len = prim<uint32_t>();
    if (len == 0) return std::string();  // some writers send an empty string as length 0
    need(len);
    std::string s(reinterpret_cast<const char*>(p_ + at_), len - 1);
    at_ += len;
    return s;
  }
  uint32_t count(size_t min_bytes_per_element) {  // sequence length, checked against what is left (hostile lengths)
    const uint32_t c = prim<uint32_t>();
    if ((size_t)c * min_bytes_per_element > n_ - at_) throw std::runtime_error("cdr: sequence longer than the buffer");
    return c;
  }
  void bytes_seq(std::vector<uint8_t>& d) {
    const uint32_t c = count(1);
    d.assign(p_ + at_, p_ + at_ + c);
    at_ += c;
  }

 private:
  void need(size_t k) const { if (k > n_ - at_) throw std::runtime_error("cdr: truncated message"); }
  void align(size_t s) {
    const size_t pos = at_ - 4;
    at_ += (s - pos % s) % s;
    if (at_ > n_) throw std::runtime_error("cdr: truncated message");
  }
  const uint8_t* p_;
  size_t n_, at_ = 0;
  bool swap_ = false;
};

// ---- field lists: one `put` / `get` per message type, members in .msg order ----
inline void put(CdrWriter& w, const Time& m) { w.prim(m.sec); w.prim(m.nanosec); }
inline void get(CdrReader& r, Time& m) { m.sec = r.prim<int32_t>(); m.nanosec = r.prim<uint32_t>(); }
inline void put(CdrWriter& w, const Header& m) { put(w, m.stamp); w.string(m.frame_id); }
inline void get(CdrReader& r, Header& m) { get(r, m.stamp); m.frame_id = r.string(); }
inline void put(CdrWriter& w, const String& m) { w.string(m.data); }
inline void get(CdrReader& r, String& m) { m.data = r.string(); }
inline void put(CdrWriter& w, const Float32& m) { w.prim(m.data); }
inline void get(CdrReader& r, Float32& m) { m.data = r.prim<float>(); }

inline void put(CdrWriter& w, const PointField& m) { w.string(m.name); w.prim(m.offset); w.prim(m.datatype); w.prim(m.count); }
inline void get(CdrReader& r, PointField& m) { m.name = r.string(); m.offset = r.prim<uint32_t>(); m.datatype = r.prim<uint8_t>(); m.count = r.prim<uint32_t>(); }
inline void put(CdrWriter& w, const PointCloud2& m) {
  put(w, m.header); w.prim(m.height); w.prim(m.width);
  w.prim<uint32_t>((uint32_t)m.fields.size());
  for (const PointField& f : m.fields) put(w, f);
  w.boolean(m.is_bigendian); w.prim(m.point_step); w.prim(m.row_step); w.bytes_seq(m.data); w.boolean(m.is_dense);
}
inline void get(CdrReader& r, PointCloud2& m) {
  get(r, m.header); m.height = r.prim<uint32_t>(); m.width = r.prim<uint32_t>();
  m.fields.resize(r.count(13));
  for (PointField& f : m.fields) get(r, f);
  m.is_bigendian = r.boolean(); m.point_step = r.prim<uint32_t>(); m.row_step = r.prim<uint32_t>(); r.bytes_seq(m.data); m.is_dense = r.boolean();
}

inline void put(CdrWriter& w, const Point& m) { w.prim(m.x); w.prim(m.y); w.prim(m.z); }
inline void get(CdrReader& r, Point& m) { m.x = r.prim<double>(); m.y = r.prim<double>(); m.z = r.prim<double>(); }
inline void put(CdrWriter& w, const Vector3& m) { w.prim(m.x); w.prim(m.y); w.prim(m.z); }
inline void get(CdrReader& r, Vector3& m) { m.x = r.prim<double>(); m.y = r.prim<double>(); m.z = r.prim<double>(); }
inline void put(CdrWriter& w, const Quaternion& m) { w.prim(m.x); w.prim(m.y); w.prim(m.z); w.prim(m.w); }
inline void get(CdrReader& r, Quaternion& m) { m.x = r.prim<double>(); m.y = r.prim<double>(); m.z = r.prim<double>(); m.w = r.prim<double>(); }
inline void put(CdrWriter& w, const Pose& m) { put(w, m.position); put(w, m.orientation); }
inline void get(CdrReader& r, Pose& m) { get(r, m.position); get(r, m.orientation); }
inline void put(CdrWriter& w, const Twist& m) { put(w, m.linear); put(w, m.angular); }
inline void get(CdrReader& r, Twist& m) { get(r, m.linear); get(r, m.angular); }
inline void put(CdrWriter& w, const PoseWithCovariance& m) { put(w, m.pose); for (double c : m.covariance) w.prim(c); }
inline void get(CdrReader& r, PoseWithCovariance& m) { get(r, m.pose); for (double& c : m.covariance) c = r.prim<double>(); }
inline void put(CdrWriter& w, const TwistWithCovariance& m) { put(w, m.twist); for (double c : m.covariance) w.prim(c); }
inline void get(CdrReader& r, TwistWithCovariance& m) { get(r, m.twist); for (double& c : m.covariance) c = r.prim<double>(); }
inline void put(CdrWriter& w, const PoseStamped& m) { put(w, m.header); put(w, m.pose); }
inline void get(CdrReader& r, PoseStamped& m) { get(r, m.header); get(r, m.pose); }
inline void put(CdrWriter& w, const Odometry& m) { put(w, m.header); w.string(m.child_frame_id); put(w, m.pose); put(w, m.twist); }
inline void get(CdrReader& r, Odometry& m) { get(r, m.header); m.child_frame_id = r.string(); get(r, m.pose); get(r, m.twist); }
inline void put(CdrWriter& w, const Path& m) {
  put(w, m.header);
  w.prim<uint32_t>((uint32_t)m.poses.size());
  for (const PoseStamped& p : m.poses) put(w, p);
}
inline void get(CdrReader& r, Path& m) {
  get(r, m.header);
  m.poses.resize(r.count(68));
  for (PoseStamped& p : m.poses) get(r, p);
}

inline void put(CdrWriter& w, const IterationStats& m) {
  put(w, m.header); w.prim(m.translation_norm); w.prim(m.rotation_norm); w.prim(m.num_surf_from_scan); w.prim(m.num_corner_from_scan);
}
inline void get(CdrReader& r, IterationStats& m) {
  get(r, m.header); m.translation_norm = r.prim<double>(); m.rotation_norm = r.prim<double>();
  m.num_surf_from_scan = r.prim<double>(); m.num_corner_from_scan = r.prim<double>();
}
inline void put(CdrWriter& w, const OptimizationStats& m) {
  put(w, m.header);
  w.prim(m.laser_cloud_surf_from_map_num); w.prim(m.laser_cloud_corner_from_map_num); w.prim(m.laser_cloud_surf_stack_num); w.prim(m.laser_cloud_corner_stack_num);
  w.prim(m.total_translation); w.prim(m.total_rotation); w.prim(m.translation_from_last); w.prim(m.rotation_from_last); w.prim(m.time_elapsed); w.prim(m.latency);
  w.prim(m.n_iterations); w.prim(m.average_distance);
  w.prim(m.uncertainty_x); w.prim(m.uncertainty_y); w.prim(m.uncertainty_z); w.prim(m.uncertainty_roll); w.prim(m.uncertainty_pitch); w.prim(m.uncertainty_yaw);
  w.prim(m.plane_match_success); w.prim(m.plane_no_enough_neighbor); w.prim(m.plane_neighbor_too_far); w.prim(m.plane_badpca_structure);
  w.prim(m.plane_invalid_numerical); w.prim(m.plane_mse_too_large); w.prim(m.plane_unknown);
  w.prim(m.prediction_source);
  w.prim<uint32_t>((uint32_t)m.iterations.size());
  for (const IterationStats& it : m.iterations) put(w, it);
}
inline void get(CdrReader& r, OptimizationStats& m) {
  get(r, m.header);
  m.laser_cloud_surf_from_map_num = r.prim<int32_t>(); m.laser_cloud_corner_from_map_num = r.prim<int32_t>();
  m.laser_cloud_surf_stack_num = r.prim<int32_t>(); m.laser_cloud_corner_stack_num = r.prim<int32_t>();
  m.total_translation = r.prim<double>(); m.total_rotation = r.prim<double>(); m.translation_from_last = r.prim<double>();
  m.rotation_from_last = r.prim<double>(); m.time_elapsed = r.prim<double>(); m.latency = r.prim<double>();
  m.n_iterations = r.prim<int32_t>(); m.average_distance = r.prim<double>();
  m.uncertainty_x = r.prim<double>(); m.uncertainty_y = r.prim<double>(); m.uncertainty_z = r.prim<double>();
  m.uncertainty_roll = r.prim<double>(); m.uncertainty_pitch = r.prim<double>(); m.uncertainty_yaw = r.prim<double>();
  m.plane_match_success = r.prim<int32_t>(); m.plane_no_enough_neighbor = r.prim<int32_t>(); m.plane_neighbor_too_far = r.prim<int32_t>();
  m.plane_badpca_structure = r.prim<int32_t>(); m.plane_invalid_numerical = r.prim<int32_t>(); m.plane_mse_too_large = r.prim<int32_t>();
  m.plane_unknown = r.prim<int32_t>(); m.prediction_source = r.prim<int32_t>();
  m.iterations.resize(r.count(44));
  for (IterationStats& it : m.iterations) get(r, it);
}
inline void put(CdrWriter& w, const LaserFeature& m) {
  put(w, m.header);
  w.prim(m.sensor); w.prim(m.imu_available); w.prim(m.odom_available);
  w.prim(m.imu_quaternion_x); w.prim(m.imu_quaternion_y); w.prim(m.imu_quaternion_z); w.prim(m.imu_quaternion_w);
  w.prim(m.initial_pose_x); w.prim(m.initial_pose_y); w.prim(m.initial_pose_z);
  w.prim(m.initial_quaternion_x); w.prim(m.initial_quaternion_y); w.prim(m.initial_quaternion_z); w.prim(m.initial_quaternion_w);
  w.prim(m.imu_preintegration_reset_id);
  put(w, m.cloud_nodistortion); put(w, m.cloud_corner); put(w, m.cloud_surface); put(w, m.cloud_realsense);
}
inline void get(CdrReader& r, LaserFeature& m) {
  get(r, m.header);
  m.sensor = r.prim<int64_t>(); m.imu_available = r.prim<int64_t>(); m.odom_available = r.prim<int64_t>();
  m.imu_quaternion_x = r.prim<double>(); m.imu_quaternion_y = r.prim<double>(); m.imu_quaternion_z = r.prim<double>(); m.imu_quaternion_w = r.prim<double>();
  m.initial_pose_x = r.prim<double>(); m.initial_pose_y = r.prim<double>(); m.initial_pose_z = r.prim<double>();
  m.initial_quaternion_x = r.prim<double>(); m.initial_quaternion_y = r.prim<double>(); m.initial_quaternion_z = r.prim<double>(); m.initial_quaternion_w = r.prim<double>();
  m.imu_preintegration_reset_id = r.prim<int64_t>();
  get(r, m.cloud_nodistortion); get(r, m.cloud_corner); get(r, m.cloud_surface); get(r, m.cloud_realsense);
}

template <typename M> std::vector<uint8_t> serialize(const M& m) { CdrWriter w; put(w, m); return w.take(); }
// A PointCloud2 whose payload is produced IN PLACE (a device copy lands in the message, no intermediate cloud): the serialised message of
// `meta` (data empty, width / row_step already those of the final message) up to and including the length word of `data`, for a payload
// of payload_bytes.  The caller writes the payload behind it and one more byte: is_dense.  (put(PointCloud2) above ends with
// [uint32 length][bytes][bool]: bytes need no alignment, the bool none either.)
inline std::vector<uint8_t> cloud_prefix(const PointCloud2& meta, size_t payload_bytes) {
  if (!meta.data.empty()) throw std::runtime_error("cloud_prefix: the meta message must not carry a payload");
  std::vector<uint8_t> b = serialize(meta);
  b.pop_back();  // is_dense
  const uint32_t len = (uint32_t)payload_bytes;
  std::memcpy(b.data() + b.size() - 4, &len, 4);
  return b;
}
template <typename M> M deserialize(const uint8_t* p, size_t n) { CdrReader r(p, n); M m; get(r, m); return m; }
template <typename M> M deserialize(const std::vector<uint8_t>& b) { return deserialize<M>(b.data(), b.size()); }

}  // namespace so_wire
