// standins.h -- 60-line stand-ins for the third-party types that appear in the signature and in the public fields of
// the reference's LidarSLAM, so that the adapter (lidar_slam_soicp.{h,cpp}) compiles and runs where Eigen, PCL and ROS 2
// are absent.  In a real build of super_odometry define SUPERODOM_HAVE_ROS: the adapter then uses the node's own
// headers (utils/Twist.h, pcl/point_types.h, super_odometry_msgs) and this file is not read.
// Citations are relative to /root/reference/super_odometry/.
#pragma once
#include <cstddef>
#include <cstdint>
#include <memory>
#include <vector>

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
// super_odometry_msgs/msg/IterationStats.msg, OptimizationStats.msg (the fields LidarSlam.cpp fills)
struct IterationStats {
  double translation_norm = 0, rotation_norm = 0;
  int32_t num_surf_from_scan = 0, num_corner_from_scan = 0;
};
struct OptimizationStats {
  int32_t laser_cloud_surf_from_map_num = 0, laser_cloud_corner_from_map_num = 0, laser_cloud_surf_stack_num = 0, laser_cloud_corner_stack_num = 0;
  double total_translation = 0, total_rotation = 0, translation_from_last = 0, rotation_from_last = 0, time_elapsed = 0, latency = 0;
  int32_t n_iterations = 0;
  double average_distance = 0;
  double uncertainty_x = 0, uncertainty_y = 0, uncertainty_z = 0, uncertainty_roll = 0, uncertainty_pitch = 0, uncertainty_yaw = 0;
  int32_t prediction_source = 0;
  std::vector<IterationStats> iterations;
};

}  // namespace so_standins
