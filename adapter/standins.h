// standins.h -- stand-ins for the third-party types that appear in the signature and in the public fields of
// the reference's LidarSLAM, so that the adapter (lidar_slam_soicp.{h,cpp}) compiles and runs where Eigen, PCL and ROS 2
// are absent.  In a real build of super_odometry define SUPERODOM_HAVE_ROS: the adapter then uses the node's own
// headers (utils/Twist.h, pcl/point_types.h, super_odometry_msgs) and this file is not read.
// Citations are relative to /root/reference/super_odometry/.
#pragma once
#include <cstddef>
#include <cstdint>
#include <memory>
#include <vector>

#include "wire/messages.h"

namespace so_standins {

// Eigen::Vector3d / Eigen::Quaterniond as far as the node uses them on this path (x() y() z() w(), constructors)
struct Vector3d {
  double v[3] = {0, 0, 0};
  Vector3d() = default;
  Vector3d(double x, double y, double z) : v{x, y, z} {}
  double x() const { return v[0]; } double y() const { return v[1]; } double z() const { return v[2]; }
};
struct Vector3i {
  int v[3] = {0, 0, 0};
  Vector3i() = default;
  Vector3i(int x, int y, int z) : v{x, y, z} {}
  int x() const { return v[0]; } int y() const { return v[1]; } int z() const { return v[2]; }
};
struct Quaterniond {  // Eigen constructor order: (w, x, y, z)
  double qw = 1, qx = 0, qy = 0, qz = 0;
  Quaterniond() = default;
  Quaterniond(double w, double x, double y, double z) : qw(w), qx(x), qy(y), qz(z) {}
  double x() const { return qx; } double y() const { return qy; } double z() const { return qz; } double w() const { return qw; }
};
// Transformd = Twist<double>, include/super_odometry/utils/Twist.h:47-68: public members rot, pos
struct Transformd {
  Quaterniond rot;
  Vector3d pos;
};
// pcl::PointXYZI, LidarProcess/LocalMap.h:40: 32 bytes {x, y, z, pad, intensity, pad[3]}, 16-byte aligned
struct alignas(16) PointXYZI {
  float x = 0, y = 0, z = 0, pad0 = 1.f;
  float intensity = 0, pad1[3] = {0, 0, 0};
};
static_assert(sizeof(PointXYZI) == 32, "pcl::PointXYZI layout");
template <typename P> struct PointCloud {
  using Ptr = std::shared_ptr<PointCloud<P>>;
  std::vector<P> points;
  size_t size() const { return points.size(); }
  bool empty() const { return points.empty(); }
  void push_back(const P& p) { points.push_back(p); }
};
// super_odometry_msgs/msg/IterationStats.msg, OptimizationStats.msg: field for field the message definitions (wire/messages.h)
using IterationStats = so_wire::IterationStats;
using OptimizationStats = so_wire::OptimizationStats;

}  // namespace so_standins
