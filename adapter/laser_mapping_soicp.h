// laser_mapping_soicp.h -- SURVEY 8(f) row f4: the laser_mapping_node shell around the path, without rclcpp.
// The class below has the members and the per-frame sequence of super_odometry::laserMapping
// (include/super_odometry/LaserMapping/laserMapping.h:72-311, src/LaserMapping/laserMapping.cpp) -- subscription callback,
// process() body, setInitialGuess, adjustVoxelSize, performSLAMOptimization, updatePoseAndPublish, publishTopic -- with
//   * the subscription / publishers replaced by an Outbox that receives (topic, type, CDR bytes): what an rmw would put
//     on the wire and rosbag2 would store (wire/cdr.h); typed callbacks are available too;
//   * adjustVoxelSize's statistic + VoxelGrid of the surf cloud and LidarSLAM::Localization running on libsoicp: the
//     PointCloud2 payload goes to the device with its point_step as the stride (no pcl::fromROSMsg copy), the filtered
//     cloud never returns to the host;
//   * Eigen / tf2 arithmetic written out (node_math.h).
// Paths relative to /root/reference/super_odometry/.  In a ROS 2 workspace the same class body sits under an rclcpp::Node
// ("node shell").
#pragma once
#include <chrono>
#include <functional>
#include <mutex>
#include <queue>
#include <string>
#include <vector>

#include "lidar_slam_soicp.h"
#include "wire/cdr.h"

namespace super_odometry_soicp {

// laser_mapping_config (laserMapping.h:38-69) + the globals of src/parameter/parameter.cpp the node reads (:284-297)
struct NodeConfig {
  // defaults = the values readParameters declares (laserMapping.cpp:182-203); node_config.h fills this from a parameter file
  float lineRes = 0.1f, planeRes = 0.2f;          // mapping_line_resolution / mapping_plane_resolution
  int max_iterations = 4;
  bool debug_view_enabled = false, enable_ouster_data = false, publish_only_feature_points = false;
  bool use_imu_roll_pitch = false;
  int max_surface_features = 2000;
  double velocity_failure_threshold = 30.0;
  bool auto_voxel_size = true, forget_far_chunks = false;
  double visual_confidence_factor = 1.0, pos_degeneracy_threshold = 1.0, ori_degeneracy_threshold = 1.0;
  float yaw_ratio = 0.f;
  std::string map_dir;
  bool localization_mode = false;
  float init_x = 0, init_y = 0, init_z = 0, init_roll = 0, init_pitch = 0, init_yaw = 0;
  // globals
  std::string ProjectName, WORLD_FRAME = "sensor_init", SENSOR_FRAME = "sensor";
  double imu_laser_R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};  // row-major extrinsic rotation (parameter.cpp:176)
  int device_id = 0;                                    // libsoicp: HIP device of this process
};

// where the node's publishers go
class Outbox {
 public:
  virtual ~Outbox() = default;
  virtual void publish(const std::string& topic, const std::string& type, std::vector<uint8_t>&& cdr) = 0;
  // a message assembled in a buffer the node keeps (pinned memory the device wrote the payload into): a bridge that can send from a
  // pointer (rcl_serialized_message_t wraps one) overrides this and saves the copy
  virtual void publish_bytes(const std::string& topic, const std::string& type, const uint8_t* cdr, size_t n) {
    publish(topic, type, std::vector<uint8_t>(cdr, cdr + n));
  }
};

class laserMapping {
 public:
  enum class PredictionSource { IMU_ORIENTATION, LIO_ODOM, VIO_ODOM, NEURAL_IMU_ODOM, CONSTANT_VELOCITY };  // laserMapping.h:92

  laserMapping(const NodeConfig& cfg, Outbox* out);
  void initInterface();                                                       // laserMapping.cpp:21-127 + initializationParam :129-178
  void laserFeatureInfoHandler(const so_wire::LaserFeature& msgIn);           // :250-263 (any thread)
  void laserFeatureInfoHandler(const uint8_t* cdr, size_t n);                 // the same, from a serialised message
  bool processOnce();  // one turn of process()'s loop (:768-793): false when checkDataAvailable() says no
  // the prior map of localization mode (:161-171): from config_.map_dir (a .pcd file: pcd_io.h restates the reader; false = the
  // file could not be read and the node switched to mapping mode, like the reference), or handed over as points
  bool loadPriorMap();
  void loadPriorMap(const float* xyz, size_t n, size_t stride_bytes);

  LidarSLAM slam;
  PredictionSource prediction_source = PredictionSource::IMU_ORIENTATION;
  // accumulated wall time: extract, initial guess, adjustVoxelSize, Localization, publish; then, inside publish: the map clouds (every 5th /
  // 20th frame), the registered scan, everything that went through the outbox (serialisation + the bridge's own time)
  double phase_seconds[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  int frames_failed = 0;           // frames whose processing threw (process() logs and continues, :788-790)
  std::string last_error;

 private:
  struct SensorData {               // laserMapping.h:75-88 (the prediction sources that were never released stay false)
    Quaterniond imuPrediction;
    bool vio_prediction_status = false, lio_prediction_status = false, nio_prediction_status = false, imu_orientation_status = false;
    double timestamp = 0;
  };
  bool checkDataAvailable() const;
  SensorData extractSensorData();
  void clearSensorData();
  void setInitialGuess();
  void initializeFirstFrame();
  void initializeWithIMU();
  void selectPosePrediction();
  PredictionSource determinePredictionSource();
  bool useIMUPrediction(const Quaterniond& imuPrediction);
  void adjustVoxelSize();
  void performSLAMOptimization();
  void updatePoseAndPublish();
  void publishTopic();
  template <typename M> void publish(const std::string& topic, const char* type, const M& m) {
    if (!out_) return;
    const auto t0 = std::chrono::steady_clock::now();
    out_->publish(topic, type, so_wire::serialize(m));
    phase_seconds[7] += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  }
  // laser_cloud_map / laser_cloud_surround: the map cloud gathered by the device straight into the message (so_icp_map_export_records)
  void publishMapCloud(const std::string& topic, bool only_5x5, const so_wire::Time& stamp);

  NodeConfig config_;
  Outbox* out_;
  std::mutex mBuf;
  std::queue<so_wire::PointCloud2> cornerLastBuf, surfLastBuf, realsenseBuf, fullResBuf;
  std::queue<Quaterniond> IMUPredictionBuf;
  so_wire::PointCloud2 cornerLast_, surfLast_, fullRes_;  // the frame being processed (laserCloud*Last, laserCloudFullRes)
  SensorData sensorMeas;

  int frameCount = 0, startupCount = 10;            // laserMapping.h:196-198
  double timeLaserOdometry = 0, timeLaserOdometryPrev = 0;
  bool laser_imu_sync = false, initialization = false;
  Transformd T_w_lidar, last_T_w_lidar, laser_incremental_T;
  Quaterniond q_wodom_curr, q_wodom_pre, q_w_curr;  // q_w_curr / t_w_curr: file-scope Eigen::Map globals in the reference (:7-9)
  Vector3d t_w_curr, vel_b, ang_vel_b;
  so_wire::Path laserAfterMappedPath;
  so_wire::PointCloud2 priorCloudMsg;
  // adjustVoxelSize's products on the device
  const void* d_surf_stack_ = nullptr;
  size_t n_surf_stack_ = 0;
  int32_t corner_stack_num_ = 0;
};

// pcl::toROSMsg(pcl::PointCloud<pcl::PointXYZI>) as the node uses it for its map / scan topics: 32-byte points, fields
// x y z intensity at 0 4 8 16, height 1, is_dense
so_wire::PointCloud2 to_ros_msg(const PointCloud<Point>& cloud);
// where x y z sit in a PointCloud2 (FLOAT32 fields named "x" "y" "z", as pcl::fromROSMsg's field map requires); throws otherwise
struct XyzLayout { uint32_t off_x, off_y, off_z, off_intensity; bool has_intensity, contiguous; };
XyzLayout xyz_layout(const so_wire::PointCloud2& msg);

}  // namespace super_odometry_soicp
