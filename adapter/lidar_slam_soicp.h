// lidar_slam_soicp.h -- the reference-side adapter: a class with the public surface of super_odometry::LidarSLAM
// (include/super_odometry/LidarProcess/LidarSlam.h) as far as laserMapping.cpp touches it, whose Localization() runs on
// libsoicp (include/so_icp.h) instead of the CPU path.  Paths relative to /root/reference/super_odometry/.
//
//   replaces  LidarSLAM::Localization                    include/.../LidarProcess/LidarSlam.h:292-293, src/LidarProcess/LidarSlam.cpp:30-51
//   caller    laserMapping::performSLAMOptimization      src/LaserMapping/laserMapping.cpp:703-741
//   read-backs  T_w_lidar (:734-737), startupCount (:738), isDegenerate (:387,563), pos_in_localmap (:439),
//               localMap.get5x5LocalMap / getAllLocalMap (:439,450), stats (:581-596);  written by the node: frame_count,
//               laser_imu_sync (:740-741), last_T_w_lidar (:312), localMap.lineRes_/planeRes_ (:103-104, 648-649),
//               LocalizationICPMaxIter, OptSet.*, Visual_confidence_factor, localization_mode, init_*, map_dir (:103-120)
#pragma once
#include <string>
#include <vector>

#include "so_icp.h"

#ifdef SUPERODOM_HAVE_ROS
#include <pcl/point_cloud.h>
#include <pcl/point_types.h>
#include "super_odometry/utils/Twist.h"
#include <super_odometry_msgs/msg/optimization_stats.hpp>
namespace so_adapter_types {
using Transformd = ::Transformd;
using Vector3d = Eigen::Vector3d; using Vector3i = Eigen::Vector3i; using Quaterniond = Eigen::Quaterniond;
using Point = pcl::PointXYZI;
template <typename P> using PointCloud = pcl::PointCloud<P>;
using OptimizationStats = super_odometry_msgs::msg::OptimizationStats;
using IterationStats = super_odometry_msgs::msg::IterationStats;
}
#else
#include "standins.h"
namespace so_adapter_types {
using namespace so_standins;
using Point = so_standins::PointXYZI;
}
#endif

namespace super_odometry_soicp {
using namespace so_adapter_types;

// LocalMap facade: the members of LidarProcess/LocalMap.h the node uses, forwarded to the context's HBM-resident map
class LocalMapFacade {
 public:
  float lineRes_ = 0.2f, planeRes_ = 0.4f;  // LocalMap.h:760-761; written by the node every frame
  void setOrigin(const Vector3d& t_w_cur);                              // LocalMap.h:146
  void addSurfPointCloud(const PointCloud<Point>& cloud);               // LocalMap.h:591 (localization mode: prior map)
  PointCloud<Point> get5x5LocalMap(const Vector3i& pos_in_localmap);    // LocalMap.h:646-688
  PointCloud<Point> getAllLocalMap();
  // the same clouds as pcl::PointXYZI records written where the caller says -- the payload area of the PointCloud2 message being assembled
  // (so_icp_map_export_records: one gather on the device, one copy); out == nullptr: the number of points
  size_t exportRecords(void* out, size_t cap_points, bool only_5x5, const Vector3i& pos_in_localmap);
 private:
  friend class LidarSLAM;
  PointCloud<Point> export_points(int only_5x5, const int pos[3]);
  so_icp_ctx** ctx_ = nullptr;
  class LidarSLAM* owner_ = nullptr;
};

class LidarSLAM {
 public:
  enum class PredictionSource { IMU_ORIENTATION, LIO_ODOM, VIO_ODOM };  // LidarSlam.h:61
  struct LidarOptimizationSetting {                                       // LidarSlam.h:153-161 (fields the node writes)
    bool debug_view_enabled = false, use_imu_roll_pitch = false;
    double velocity_failure_threshold = 30.0, yaw_ratio = 0.0;
    int max_surface_features = 2000;
  } OptSet;

  LidarSLAM() { localMap.ctx_ = &gpu_; localMap.owner_ = this; }
  ~LidarSLAM();
  LidarSLAM(const LidarSLAM&) = delete;
  LidarSLAM& operator=(const LidarSLAM&) = delete;

  // LidarSlam.h:292-293
  void Localization(bool initialization, PredictionSource predictodom, Transformd T_w_lidar_in, PointCloud<Point>::Ptr edge_point,
                    PointCloud<Point>::Ptr planner_point, double timeLaserOdometry);
  // optional: announce the planar cloud of the NEXT frame (the feature callback has it before process() reaches it)
  void StageNextScan(const PointCloud<Point>::Ptr& planner_point);
  // laserMapping::adjustVoxelSize (laserMapping.cpp:600-651) for the surf cloud on the device: cloud statistic (auto voxel size),
  // VoxelGrid at planeRes, localMap.lineRes_/planeRes_ updated; *d_filtered stays in HBM (valid until the next call) and
  // feeds LocalizationPrefiltered -- the filtered cloud never visits the host.  xyz may point into a PointCloud2 payload.
  void PrefilterSurf(const float* xyz, size_t n, size_t stride_bytes, bool auto_voxel_size, float line_res, float plane_res,
                     so_icp_prefilter_info* info, const void** d_filtered, size_t* n_filtered);
  // optional, any thread: the raw surf cloud the NEXT PrefilterSurf call will name starts its H2D copy now (so_icp_prefilter_announce);
  // the buffer must stay where it is, unchanged, until that call.  No-op before the context exists.
  void AnnounceSurf(const float* xyz, size_t n, size_t stride_bytes);
  // utils::pointAssociateToMap over a cloud (the registered scan of laserMapping::publishTopic, laserMapping.cpp:464-493) on the
  // device: records with float x y z at 0 4 8 rewritten in place, keep[i] = the node publishes point i
  size_t TransformCloud(void* points, size_t n, size_t stride_bytes, const Transformd& T, std::vector<uint8_t>& keep);
  // A message buffer of at least `bytes` in pinned host memory (so_icp_host_alloc): copies between it and the device are DMA transfers.
  // One buffer per `which` (0, 1, ...), grown on demand, valid until the next call with the same index or the context's end.
  uint8_t* PinnedScratch(int which, size_t bytes);
  void LocalizationPrefiltered(bool initialization, PredictionSource predictodom, Transformd T_w_lidar_in, const void* d_planner_xyz,
                               size_t n_planner, int32_t n_edge_points, double timeLaserOdometry);

  // ---- public fields laserMapping.cpp reads / writes (same names) ----
  Transformd T_w_lidar, last_T_w_lidar;
  int startupCount = 0;
  bool isDegenerate = false;           // never set by the reference either (LidarSlam.cpp:977-984 is commented out)
  Vector3i pos_in_localmap;
  OptimizationStats stats;
  LocalMapFacade localMap;
  size_t LocalizationICPMaxIter = 4;   // LidarSlam.h:273
  double Visual_confidence_factor = 0, Pos_degeneracy_threshold = 0, Ori_degeneracy_threshold = 0;
  bool localization_mode = false;
  float init_x = 0, init_y = 0, init_z = 0, init_roll = 0, init_pitch = 0, init_yaw = 0;
  std::string map_dir;
  int frame_count = 0, laser_imu_sync = 0;
  double lasttimeLaserOdometry = 0;
  // ---- libsoicp specifics ----
  int device_id = 0;                   // HIP device of this process (one process per GPU)
  uint32_t last_flags = 0;             // so_icp_stats::flags of the last call (degraded modes, for the node's log)
  int last_status = 0;                 // so_icp_localization return code of the last call
  so_icp_stats last_raw;               // everything the C ABI returned (JtJ / Jtr, per-iteration histograms)

 private:
  friend class LocalMapFacade;
  void ensure_context();
  void push_knobs();
  void read_back(int32_t n_edge_points, const double T_out[7], double timeLaserOdometry);
  so_icp_ctx* gpu_ = nullptr;
  struct Scratch { uint8_t* p = nullptr; size_t cap = 0; };
  std::vector<Scratch> scratch_;
};

}  // namespace super_odometry_soicp
