// messages.h -- the message types laser_mapping_node exchanges, as plain C++ structs with the field names and the field
// ORDER of the .msg definitions (the order is the wire format, see cdr.h).  They stand where rosidl's generated
// `<pkg>::msg::<Type>` structs stand in the reference; with ROS 2 present the generated types are used instead and this
// file is not read.  Citations relative to /root/reference/.
//
//   super_odometry_msgs/msg/LaserFeature.msg        in : <ProjectName>/feature_info       (laserMapping.cpp:49-52, 250-263)
//   super_odometry_msgs/msg/OptimizationStats.msg   out: <ProjectName>/super_odometry_stats (laserMapping.cpp:87-88, 581-596)
//   super_odometry_msgs/msg/IterationStats.msg      element of OptimizationStats.iterations
//   nav_msgs/Odometry                               out: /laser_odometry, /aft_mapped_to_init_incremental (:73-77, 504-567)
//   nav_msgs/Path, geometry_msgs/PoseStamped        out: /laser_odom_path (:569-575)
//   sensor_msgs/PointCloud2 (+PointField)           in LaserFeature; out: /registered_scan, /laser_cloud_surround, /laser_cloud_map, /overall_map
//   std_msgs/String                                 out: /prediction_source (:417-435)
//   std_msgs/Float32                                out: <ProjectName>uncertainty_{X,Y,Z,roll,pitch,yaw} (LidarSlam.cpp:22-27, 969-974)
// The common_interfaces definitions (std_msgs, sensor_msgs, geometry_msgs, nav_msgs, builtin_interfaces) are the ROS 2
// Humble ones (the reference's Dockerfile pins ros:humble); they are not part of /root/reference.
#pragma once
#include <array>
#include <cstdint>
#include <string>
#include <vector>

namespace so_wire {

struct Time {  // builtin_interfaces/Time
  int32_t sec = 0;
  uint32_t nanosec = 0;
};
struct Header {  // std_msgs/Header
  Time stamp;
  std::string frame_id;
};
struct String { std::string data; };   // std_msgs/String
struct Float32 { float data = 0.f; };  // std_msgs/Float32

struct PointField {  // sensor_msgs/PointField
  enum : uint8_t { INT8 = 1, UINT8 = 2, INT16 = 3, UINT16 = 4, INT32 = 5, UINT32 = 6, FLOAT32 = 7, FLOAT64 = 8 };
  std::string name;
  uint32_t offset = 0;
  uint8_t datatype = 0;
  uint32_t count = 0;
};
struct PointCloud2 {  // sensor_msgs/PointCloud2
  Header header;
  uint32_t height = 0, width = 0;
  std::vector<PointField> fields;
  bool is_bigendian = false;
  uint32_t point_step = 0, row_step = 0;
  std::vector<uint8_t> data;
  bool is_dense = false;
};

struct Point { double x = 0, y = 0, z = 0; };                // geometry_msgs/Point
struct Vector3 { double x = 0, y = 0, z = 0; };              // geometry_msgs/Vector3
struct Quaternion { double x = 0, y = 0, z = 0, w = 1; };    // geometry_msgs/Quaternion
struct Pose { Point position; Quaternion orientation; };     // geometry_msgs/Pose
struct Twist { Vector3 linear, angular; };                   // geometry_msgs/Twist
struct PoseWithCovariance { Pose pose; std::array<double, 36> covariance{}; };
struct TwistWithCovariance { Twist twist; std::array<double, 36> covariance{}; };
struct PoseStamped { Header header; Pose pose; };            // geometry_msgs/PoseStamped
struct Odometry {                                            // nav_msgs/Odometry
  Header header;
  std::string child_frame_id;
  PoseWithCovariance pose;
  TwistWithCovariance twist;
};
struct Path { Header header; std::vector<PoseStamped> poses; };  // nav_msgs/Path

struct IterationStats {  // super_odometry_msgs/msg/IterationStats.msg
  Header header;
  double translation_norm = 0, rotation_norm = 0, num_surf_from_scan = 0, num_corner_from_scan = 0;
};
struct OptimizationStats {  // super_odometry_msgs/msg/OptimizationStats.msg
  Header header;
  int32_t laser_cloud_surf_from_map_num = 0, laser_cloud_corner_from_map_num = 0, laser_cloud_surf_stack_num = 0, laser_cloud_corner_stack_num = 0;
  double total_translation = 0, total_rotation = 0, translation_from_last = 0, rotation_from_last = 0, time_elapsed = 0, latency = 0;
  int32_t n_iterations = 0;
  double average_distance = 0;
  double uncertainty_x = 0, uncertainty_y = 0, uncertainty_z = 0, uncertainty_roll = 0, uncertainty_pitch = 0, uncertainty_yaw = 0;
  int32_t plane_match_success = 0, plane_no_enough_neighbor = 0, plane_neighbor_too_far = 0, plane_badpca_structure = 0,
          plane_invalid_numerical = 0, plane_mse_too_large = 0, plane_unknown = 0;
  int32_t prediction_source = 0;
  std::vector<IterationStats> iterations;
};
struct LaserFeature {  // super_odometry_msgs/msg/LaserFeature.msg
  Header header;
  int64_t sensor = 0, imu_available = 0, odom_available = 0;
  double imu_quaternion_x = 0, imu_quaternion_y = 0, imu_quaternion_z = 0, imu_quaternion_w = 0;
  double initial_pose_x = 0, initial_pose_y = 0, initial_pose_z = 0;
  double initial_quaternion_x = 0, initial_quaternion_y = 0, initial_quaternion_z = 0, initial_quaternion_w = 0;
  int64_t imu_preintegration_reset_id = 0;
  PointCloud2 cloud_nodistortion, cloud_corner, cloud_surface, cloud_realsense;
};

}  // namespace so_wire
