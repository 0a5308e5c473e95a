// laser_mapping_soicp.cpp -- see laser_mapping_soicp.h.  Function by function the body of
// /root/reference/super_odometry/src/LaserMapping/laserMapping.cpp with rclcpp / PCL / Eigen / tf2 replaced as the header says.
#include "laser_mapping_soicp.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <stdexcept>

#include "node_math.h"
#include "pcd_io.h"

namespace super_odometry_soicp {
using namespace so_node_math;

namespace {
so_wire::Time stamp_from_seconds(double t) {  // rclcpp::Time(int64_t(t * 1e9)) -> builtin_interfaces/Time (:445, 508)
  const int64_t ns = (int64_t)(t * 1e9);
  so_wire::Time s;
  s.sec = (int32_t)(ns / 1000000000LL);
  s.nanosec = (uint32_t)(ns % 1000000000LL);
  if (ns < 0 && ns % 1000000000LL) { s.sec -= 1; s.nanosec = (uint32_t)(ns % 1000000000LL + 1000000000LL); }
  return s;
}
double secs(const so_wire::PointCloud2& m) { return m.header.stamp.sec + m.header.stamp.nanosec * 1e-9; }  // laserMapping.h:158-161

// number of points pcl::VoxelGrid (leaf x leaf x leaf) returns for this cloud = occupied leaves; the node runs the
// filter on the corner cloud only to report its size (the edge path of the registration is dead, LidarSlam.cpp:402-512).
// Leaf arithmetic of pcl::VoxelGrid::applyFilter: float inverse leaf size, floor(coordinate * inverse) per axis.
int32_t voxel_grid_count(const so_wire::PointCloud2& msg, float leaf) {
  const size_t n = (size_t)msg.width * msg.height;
  if (!n) return 0;
  const XyzLayout L = xyz_layout(msg);
  const float inv = 1.0f / leaf;
  float mn[3] = {3.0e38f, 3.0e38f, 3.0e38f}, mx[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
  std::vector<float> p(3 * n);
  size_t m = 0;
  for (size_t i = 0; i < n; ++i) {
    float v[3];
    std::memcpy(&v[0], msg.data.data() + i * msg.point_step + L.off_x, 4);
    std::memcpy(&v[1], msg.data.data() + i * msg.point_step + L.off_y, 4);
    std::memcpy(&v[2], msg.data.data() + i * msg.point_step + L.off_z, 4);
    if (!msg.is_dense && !(std::isfinite(v[0]) && std::isfinite(v[1]) && std::isfinite(v[2]))) continue;
    for (int k = 0; k < 3; ++k) { mn[k] = std::min(mn[k], v[k]); mx[k] = std::max(mx[k], v[k]); p[3 * m + k] = v[k]; }
    ++m;
  }
  if (!m) return 0;
  const int64_t dx = (int64_t)((mx[0] - mn[0]) * inv) + 1, dy = (int64_t)((mx[1] - mn[1]) * inv) + 1, dz = (int64_t)((mx[2] - mn[2]) * inv) + 1;
  if (dx * dy * dz > (int64_t)INT32_MAX) return (int32_t)m;  // "leaf size is too small": the cloud passes through
  int minb[3], divb[3];
  for (int k = 0; k < 3; ++k) { minb[k] = (int)std::floor(mn[k] * inv); divb[k] = (int)std::floor(mx[k] * inv) - minb[k] + 1; }
  std::vector<int32_t> idx(m);
  for (size_t i = 0; i < m; ++i) {
    const int ix = (int)std::floor(p[3 * i] * inv) - minb[0], iy = (int)std::floor(p[3 * i + 1] * inv) - minb[1], iz = (int)std::floor(p[3 * i + 2] * inv) - minb[2];
    idx[i] = ix + iy * divb[0] + iz * divb[0] * divb[1];
  }
  std::sort(idx.begin(), idx.end());
  return (int32_t)(std::unique(idx.begin(), idx.end()) - idx.begin());
}
}  // namespace

XyzLayout xyz_layout(const so_wire::PointCloud2& msg) {
  XyzLayout L{0, 0, 0, 0, false, false};
  bool hx = false, hy = false, hz = false;
  for (const so_wire::PointField& f : msg.fields) {
    const bool f32 = f.datatype == so_wire::PointField::FLOAT32 && f.count == 1;
    if (f.name == "x" && f32) { L.off_x = f.offset; hx = true; }
    else if (f.name == "y" && f32) { L.off_y = f.offset; hy = true; }
    else if (f.name == "z" && f32) { L.off_z = f.offset; hz = true; }
    else if (f.name == "intensity" && f32) { L.off_intensity = f.offset; L.has_intensity = true; }
  }
  if (!hx || !hy || !hz) throw std::runtime_error("PointCloud2 without FLOAT32 fields x, y, z");
  if (msg.is_bigendian) throw std::runtime_error("big-endian PointCloud2 payload");
  const size_t n = (size_t)msg.width * msg.height;
  if (msg.point_step < 12 || msg.data.size() < n * msg.point_step) throw std::runtime_error("PointCloud2 payload shorter than width * height * point_step");
  if (std::max({L.off_x, L.off_y, L.off_z}) + 4 > msg.point_step) throw std::runtime_error("PointCloud2 field offset beyond point_step");
  L.contiguous = L.off_y == L.off_x + 4 && L.off_z == L.off_x + 8 && msg.point_step % 4 == 0 && L.off_x % 4 == 0;
  return L;
}

so_wire::PointCloud2 to_ros_msg(const PointCloud<Point>& cloud) {
  so_wire::PointCloud2 m;
  m.height = 1; m.width = (uint32_t)cloud.points.size();
  const char* names[4] = {"x", "y", "z", "intensity"};
  const uint32_t offs[4] = {0, 4, 8, 16};
  for (int k = 0; k < 4; ++k) { so_wire::PointField f; f.name = names[k]; f.offset = offs[k]; f.datatype = so_wire::PointField::FLOAT32; f.count = 1; m.fields.push_back(f); }
  m.is_bigendian = false; m.point_step = sizeof(Point); m.row_step = m.point_step * m.width; m.is_dense = true;
  m.data.resize(cloud.points.size() * sizeof(Point));
  if (!cloud.points.empty()) std::memcpy(m.data.data(), cloud.points.data(), m.data.size());
  return m;
}

laserMapping::laserMapping(const NodeConfig& cfg, Outbox* out) : config_(cfg), out_(out) {}

void laserMapping::initInterface() {
  slam.device_id = config_.device_id;
  slam.localMap.lineRes_ = config_.lineRes;                       // :103-120
  slam.localMap.planeRes_ = config_.planeRes;
  slam.Visual_confidence_factor = config_.visual_confidence_factor;
  slam.Pos_degeneracy_threshold = config_.pos_degeneracy_threshold;
  slam.Ori_degeneracy_threshold = config_.ori_degeneracy_threshold;
  slam.LocalizationICPMaxIter = (size_t)config_.max_iterations;
  slam.OptSet.debug_view_enabled = config_.debug_view_enabled;
  slam.OptSet.velocity_failure_threshold = config_.velocity_failure_threshold;
  slam.OptSet.max_surface_features = config_.max_surface_features;
  slam.OptSet.yaw_ratio = config_.yaw_ratio;
  slam.map_dir = config_.map_dir;
  slam.localization_mode = config_.localization_mode;
  slam.init_x = config_.init_x; slam.init_y = config_.init_y; slam.init_z = config_.init_z;
  slam.init_roll = config_.init_roll; slam.init_pitch = config_.init_pitch; slam.init_yaw = config_.init_yaw;
  prediction_source = PredictionSource::IMU_ORIENTATION;           // :122
  // initializationParam (:129-178)
  q_wodom_curr = Quaterniond(1, 0, 0, 0); q_wodom_pre = Quaterniond(1, 0, 0, 0);
  slam.localMap.setOrigin(Vector3d(slam.init_x, slam.init_y, slam.init_z));  // :159
}

void laserMapping::loadPriorMap(const float* xyz, size_t n, size_t stride_bytes) {  // :161-171
  if (!slam.localization_mode) return;
  PointCloud<Point> prior;
  prior.points.resize(n);
  for (size_t i = 0; i < n; ++i) {
    const float* p = reinterpret_cast<const float*>(reinterpret_cast<const uint8_t*>(xyz) + i * stride_bytes);
    prior.points[i].x = p[0]; prior.points[i].y = p[1]; prior.points[i].z = p[2];
  }
  slam.localMap.addSurfPointCloud(prior);
  priorCloudMsg = to_ros_msg(prior);
  priorCloudMsg.header.frame_id = config_.WORLD_FRAME;
}

// initializationParam, laserMapping.cpp:163-173, with the file: utils::readPointCloud(config_.map_dir, laserCloudPrior)
// (superodom_utils.cpp:16-33) -> addSurfPointCloud -> overall_map message; a file that cannot be read switches to mapping mode.
bool laserMapping::loadPriorMap() {
  if (!slam.localization_mode) return false;
  std::vector<float> xyzi;
  std::string err;
  if (!so_pcd::read_xyzi(config_.map_dir, xyzi, err)) {
    last_error = "Cannot read map file, switch to mapping mode: " + err;  // (:170-171)
    slam.localization_mode = false;
    return false;
  }
  PointCloud<Point> prior;
  prior.points.resize(xyzi.size() / 4);
  for (size_t i = 0; i < prior.points.size(); ++i) {
    prior.points[i].x = xyzi[4 * i]; prior.points[i].y = xyzi[4 * i + 1]; prior.points[i].z = xyzi[4 * i + 2]; prior.points[i].intensity = xyzi[4 * i + 3];
  }
  slam.localMap.addSurfPointCloud(prior);
  priorCloudMsg = to_ros_msg(prior);
  priorCloudMsg.header.frame_id = config_.WORLD_FRAME;
  return true;
}

void laserMapping::laserFeatureInfoHandler(const so_wire::LaserFeature& msgIn) {  // :250-263
  std::lock_guard<std::mutex> lk(mBuf);
  const bool next_in_line = surfLastBuf.empty();  // (process() takes the OLDEST queued frame and drops the rest, :689-699)
  cornerLastBuf.push(msgIn.cloud_corner);
  surfLastBuf.push(msgIn.cloud_surface);
  if (next_in_line) {
    // the raw surf cloud of the frame process() will take next: its copy to the device starts here, beside the frame in flight
    // (adjustVoxelSize names the same payload -- the queue's element is MOVED into surfLast_ -- and starts with its first kernel)
    const so_wire::PointCloud2& pc = surfLastBuf.front();
    const size_t n = (size_t)pc.width * pc.height;
    if (n) {
      const XyzLayout L = xyz_layout(pc);
      if (L.contiguous && !L.off_x) slam.AnnounceSurf(reinterpret_cast<const float*>(pc.data.data()), n, pc.point_step);
    }
  }
  realsenseBuf.push(msgIn.cloud_realsense);
  fullResBuf.push(msgIn.cloud_nodistortion);
  IMUPredictionBuf.push(Quaterniond(msgIn.initial_quaternion_w, msgIn.initial_quaternion_x, msgIn.initial_quaternion_y, msgIn.initial_quaternion_z));
}
void laserMapping::laserFeatureInfoHandler(const uint8_t* cdr, size_t n) {
  laserFeatureInfoHandler(so_wire::deserialize<so_wire::LaserFeature>(cdr, n));
}

bool laserMapping::checkDataAvailable() const {  // :654-658
  return !cornerLastBuf.empty() && !surfLastBuf.empty() && !fullResBuf.empty() && !IMUPredictionBuf.empty();
}

laserMapping::SensorData laserMapping::extractSensorData() {  // :660-687
  SensorData data;
  data.timestamp = secs(fullResBuf.front());
  timeLaserOdometry = data.timestamp;
  cornerLast_ = std::move(cornerLastBuf.front()); cornerLastBuf.pop();
  surfLast_ = std::move(surfLastBuf.front()); surfLastBuf.pop();
  fullRes_ = std::move(fullResBuf.front()); fullResBuf.pop();
  data.imuPrediction = qnormalized(IMUPredictionBuf.front());
  IMUPredictionBuf.pop();
  return data;
}

void laserMapping::clearSensorData() {  // :689-699 (the node always works on the oldest frame and drops the backlog)
  auto clear = [](auto& q) { while (!q.empty()) q.pop(); };
  clear(cornerLastBuf); clear(surfLastBuf); clear(fullResBuf); clear(IMUPredictionBuf);
}

void laserMapping::setInitialGuess() {  // :265-281
  if (!initialization) { initializeFirstFrame(); return; }
  if (startupCount > 0) { initializeWithIMU(); startupCount--; return; }
  selectPosePrediction();
}

void laserMapping::initializeFirstFrame() {  // :283-315
  if (sensorMeas.imuPrediction.w() != 0) {
    q_w_curr = extract_roll_pitch(sensorMeas.imuPrediction);
    Mat3 R;
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) R.m[i][j] = config_.imu_laser_R[3 * i + j];
    const Quaterniond q_extrinsic = qnormalized(from_matrix(R));
    q_w_curr = qmul(qinverse(q_extrinsic), q_w_curr);
  } else {
    q_w_curr = Quaterniond(1, 0, 0, 0);
  }
  q_wodom_pre = q_w_curr;
  T_w_lidar.rot = q_w_curr;
  T_w_lidar.pos = Vector3d(0, 0, 0);
  if (slam.localization_mode) {
    T_w_lidar.pos = Vector3d(slam.init_x, slam.init_y, slam.init_z);
    T_w_lidar.rot = set_rpy(slam.init_roll, slam.init_pitch, slam.init_yaw);
    slam.last_T_w_lidar = T_w_lidar;
  }
}

void laserMapping::initializeWithIMU() {  // :317-343
  if (sensorMeas.imuPrediction.w() != 0) {
    t_w_curr = last_T_w_lidar.pos;
    T_w_lidar.pos = t_w_curr;
    q_w_curr = sensorMeas.imuPrediction;
    T_w_lidar.rot = q_w_curr;
  } else {
    q_w_curr = last_T_w_lidar.rot;
    t_w_curr = last_T_w_lidar.pos;
    T_w_lidar = last_T_w_lidar;
  }
}

bool laserMapping::useIMUPrediction(const Quaterniond& imuPrediction) {  // :716-727
  if (imuPrediction.w() != 0) { q_wodom_curr = qnormalized(imuPrediction); return true; }
  return false;
}

laserMapping::PredictionSource laserMapping::determinePredictionSource() {  // :383-413
  if (slam.isDegenerate) {
    if (sensorMeas.vio_prediction_status) return PredictionSource::VIO_ODOM;
    if (sensorMeas.nio_prediction_status) return PredictionSource::NEURAL_IMU_ODOM;
  } else {
    if (sensorMeas.lio_prediction_status) return PredictionSource::LIO_ODOM;
    sensorMeas.imu_orientation_status = useIMUPrediction(sensorMeas.imuPrediction);
    if (sensorMeas.imu_orientation_status) return PredictionSource::IMU_ORIENTATION;
  }
  return PredictionSource::CONSTANT_VELOCITY;
}

void laserMapping::selectPosePrediction() {  // :345-381 (LIO / VIO / neural sources were never released: extractSensorData sets them false)
  prediction_source = determinePredictionSource();
  switch (prediction_source) {
    case PredictionSource::IMU_ORIENTATION: {
      const Quaterniond q_w_predict = qnormalized(qmul(qmul(q_w_curr, qinverse(q_wodom_pre)), q_wodom_curr));
      T_w_lidar.rot = q_w_predict;
      q_wodom_pre = q_wodom_curr;
      break;
    }
    case PredictionSource::CONSTANT_VELOCITY: {
      const Transformd relative_pose = tmul(tinverse(last_T_w_lidar), T_w_lidar);
      T_w_lidar = tmul(T_w_lidar, relative_pose);
      break;
    }
    default: break;
  }
  q_w_curr = T_w_lidar.rot;
  t_w_curr = T_w_lidar.pos;
}

void laserMapping::adjustVoxelSize() {  // :600-651; statistic + VoxelGrid of the surf cloud on the device (so_icp_prefilter_scan)
  const size_t n = (size_t)surfLast_.width * surfLast_.height;
  so_icp_prefilter_info info;
  std::memset(&info, 0, sizeof(info));
  d_surf_stack_ = nullptr; n_surf_stack_ = 0;
  std::vector<float> packed;
  const float* xyz = nullptr;
  size_t stride = 12;
  if (n) {
    const XyzLayout L = xyz_layout(surfLast_);
    if (L.contiguous) {  // the message payload itself is the H2D source
      xyz = reinterpret_cast<const float*>(surfLast_.data.data() + L.off_x);
      stride = surfLast_.point_step;
    }
    if (!L.contiguous || L.off_x) {  // x y z scattered in the point, or not at its start (n * stride bytes from off_x on would
                                     // run past the payload): gather once on the host
      packed.resize(3 * n);
      for (size_t i = 0; i < n; ++i) {
        const uint8_t* p = surfLast_.data.data() + i * surfLast_.point_step;
        std::memcpy(&packed[3 * i], p + L.off_x, 4); std::memcpy(&packed[3 * i + 1], p + L.off_y, 4); std::memcpy(&packed[3 * i + 2], p + L.off_z, 4);
      }
      xyz = packed.data(); stride = 12;
    }
  }
  slam.PrefilterSurf(xyz, n, stride, config_.auto_voxel_size, config_.lineRes, config_.planeRes, &info, &d_surf_stack_, &n_surf_stack_);
  if (config_.auto_voxel_size) {
    slam.stats.average_distance = info.average_distance;  // :621
    config_.lineRes = info.line_res; config_.planeRes = info.plane_res;
  }
  corner_stack_num_ = voxel_grid_count(cornerLast_, config_.lineRes);  // downSizeFilterCorner (:639-641)
  slam.localMap.lineRes_ = config_.lineRes;   // :648-649
  slam.localMap.planeRes_ = config_.planeRes;
}

void laserMapping::performSLAMOptimization() {  // :703-714
  slam.OptSet.use_imu_roll_pitch = config_.use_imu_roll_pitch;  // (the roll / pitch quaternion itself is consumed by nothing on this path)
  slam.LocalizationPrefiltered(initialization, (LidarSLAM::PredictionSource)std::min((int)prediction_source, 2), T_w_lidar,
                               d_surf_stack_, n_surf_stack_, corner_stack_num_, timeLaserOdometry);
}

void laserMapping::updatePoseAndPublish() {  // :729-765
  q_w_curr = slam.T_w_lidar.rot;
  t_w_curr = slam.T_w_lidar.pos;
  T_w_lidar.rot = slam.T_w_lidar.rot;
  T_w_lidar.pos = slam.T_w_lidar.pos;
  startupCount = slam.startupCount;
  frameCount++;
  slam.frame_count = frameCount;
  slam.laser_imu_sync = laser_imu_sync;
  initialization = true;
  const double dt = timeLaserOdometry - timeLaserOdometryPrev;
  if (dt > 1e-6) {
    const Vector3d vel_w((t_w_curr.x() - last_T_w_lidar.pos.x()) / dt, (t_w_curr.y() - last_T_w_lidar.pos.y()) / dt, (t_w_curr.z() - last_T_w_lidar.pos.z()) / dt);
    vel_b = qrot(qinverse(q_w_curr), vel_w);
    const Quaterniond dq = qmul(q_w_curr, qinverse(last_T_w_lidar.rot));
    const Vector3d aa = angle_axis_vector(dq);
    const Vector3d ang_vel_w(aa.x() / dt, aa.y() / dt, aa.z() / dt);
    ang_vel_b = qrot(qinverse(q_w_curr), ang_vel_w);
  } else {
    vel_b = Vector3d(0, 0, 0); ang_vel_b = Vector3d(0, 0, 0);
  }
  publishTopic();
  last_T_w_lidar = slam.T_w_lidar;
  timeLaserOdometryPrev = timeLaserOdometry;
}

// laser_cloud_surround / laser_cloud_map (:437-462): `pcl::toROSMsg(localMap.get...LocalMap())` -- here the device gathers the cloud as
// pcl::PointXYZI records straight into the payload area of the serialised message, which is assembled in a pinned buffer: no point cloud
// object, no to_ros_msg copy, no serialisation copy (round 5 measured 9 ms for one laser_cloud_map of the `small` scene through those).
void laserMapping::publishMapCloud(const std::string& topic, bool only_5x5, const so_wire::Time& stamp) {
  if (!out_) return;
  const size_t n = slam.localMap.exportRecords(nullptr, 0, only_5x5, slam.pos_in_localmap);
  so_wire::PointCloud2 meta = to_ros_msg(PointCloud<Point>());
  meta.width = (uint32_t)n; meta.row_step = meta.point_step * meta.width;
  meta.header.stamp = stamp; meta.header.frame_id = config_.WORLD_FRAME;
  const size_t payload = n * sizeof(Point);
  const std::vector<uint8_t> prefix = so_wire::cloud_prefix(meta, payload);
  uint8_t* buf = slam.PinnedScratch(0, prefix.size() + payload + 1);
  std::memcpy(buf, prefix.data(), prefix.size());
  const size_t got = slam.localMap.exportRecords(buf + prefix.size(), n, only_5x5, slam.pos_in_localmap);
  if (got != n) throw std::runtime_error("publishMapCloud: the map changed between the count and the export");
  buf[prefix.size() + payload] = meta.is_dense ? 1 : 0;
  const auto t0 = std::chrono::steady_clock::now();
  out_->publish_bytes(topic, "sensor_msgs/msg/PointCloud2", buf, prefix.size() + payload + 1);
  phase_seconds[7] += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}

void laserMapping::publishTopic() {  // :415-597
  const std::string& P = config_.ProjectName;
  const so_wire::Time stamp = stamp_from_seconds(timeLaserOdometry);
  so_wire::String src;
  switch (prediction_source) {
    case PredictionSource::IMU_ORIENTATION: src.data = "IMU Only Orientation Prediction"; break;
    case PredictionSource::LIO_ODOM: src.data = "Using Laser-Inertial Odometry (LIO)"; break;
    case PredictionSource::VIO_ODOM: src.data = "Using Visual-Inertial Odometry (VIO)"; break;
    case PredictionSource::NEURAL_IMU_ODOM: src.data = "Using Neural-Inertial Odometry (Neural-IMU)"; break;
    case PredictionSource::CONSTANT_VELOCITY: src.data = "Using Constant Velocity Prediction"; break;
  }
  publish(P + "/prediction_source", "std_msgs/msg/String", src);

  auto t_sub = std::chrono::steady_clock::now();
  auto lap_sub = [&](int k) { const auto now = std::chrono::steady_clock::now(); phase_seconds[k] += std::chrono::duration<double>(now - t_sub).count(); t_sub = now; };
  if (frameCount % 5 == 0 && config_.debug_view_enabled) publishMapCloud(P + "/laser_cloud_surround", true, stamp);  // :437-447
  if (frameCount % 20 == 0) {  // :449-462
    publishMapCloud(P + "/laser_cloud_map", false, stamp);
    if (slam.localization_mode) { priorCloudMsg.header.stamp = stamp; publish(P + "/overall_map", "sensor_msgs/msg/PointCloud2", priorCloudMsg); }
  }

  lap_sub(5);
  {  // registered scan (:464-493): the full-resolution cloud in the world frame, points within 0.1 m of the origin dropped.
     // Written straight into the message payload (pcl::PointXYZI records), no intermediate cloud.
    const size_t n = (size_t)fullRes_.width * fullRes_.height;
    so_wire::PointCloud2 m = to_ros_msg(PointCloud<Point>());
    size_t kept = 0;
    bool published_in_place = false;
    const XyzLayout L0 = n ? xyz_layout(fullRes_) : XyzLayout{0, 0, 0, 0, false, false};
    // (the host loop costs ~13 ns a point, the device path ~40 us flat now that both of its copies are DMA transfers: break-even near 3 000 points;
    //  SOICP_NODE_DEVICE_TRANSFORM_MIN moves the threshold)
    static const size_t device_min = std::getenv("SOICP_NODE_DEVICE_TRANSFORM_MIN") ? (size_t)std::atol(std::getenv("SOICP_NODE_DEVICE_TRANSFORM_MIN")) : 4096;
    if (n >= device_min && n && out_ && fullRes_.point_step == sizeof(Point) && L0.contiguous && L0.off_x == 0 && L0.has_intensity && L0.off_intensity == 16) {
      // the message already holds pcl::PointXYZI records: they go into the payload area of the outgoing message -- assembled in a pinned
      // buffer, so that both copies of so_icp_transform_cloud are DMA transfers --, are transformed there by the device, and the rare
      // dropped points are squeezed out in place.  (The serialised prefix has the same length whatever the point count.)
      so_wire::PointCloud2 meta = to_ros_msg(PointCloud<Point>());
      meta.header.stamp = stamp; meta.header.frame_id = config_.WORLD_FRAME;
      const size_t psize = so_wire::cloud_prefix(meta, 0).size();
      uint8_t* buf = slam.PinnedScratch(1, psize + n * sizeof(Point) + 1);
      std::memcpy(buf + psize, fullRes_.data.data(), n * sizeof(Point));
      Transformd Tw; Tw.rot = q_w_curr; Tw.pos = t_w_curr;
      std::vector<uint8_t> keep;
      kept = slam.TransformCloud(buf + psize, n, sizeof(Point), Tw, keep);
      if (kept != n) {
        size_t o = 0;
        for (size_t i = 0; i < n; ++i) if (keep[i]) { if (o != i) std::memcpy(buf + psize + o * sizeof(Point), buf + psize + i * sizeof(Point), sizeof(Point)); ++o; }
      }
      meta.width = (uint32_t)kept; meta.row_step = meta.point_step * meta.width;
      const std::vector<uint8_t> prefix = so_wire::cloud_prefix(meta, kept * sizeof(Point));
      std::memcpy(buf, prefix.data(), psize);
      buf[psize + kept * sizeof(Point)] = meta.is_dense ? 1 : 0;
      const auto t0 = std::chrono::steady_clock::now();
      out_->publish_bytes(P + "/registered_scan", "sensor_msgs/msg/PointCloud2", buf, psize + kept * sizeof(Point) + 1);
      phase_seconds[7] += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
      published_in_place = true;
    } else if (n) {
      const XyzLayout L = L0;
      m.data.resize(n * sizeof(Point));
      Point* out = reinterpret_cast<Point*>(m.data.data());
      for (size_t i = 0; i < n; ++i) {
        const uint8_t* p = fullRes_.data.data() + i * fullRes_.point_step;
        Point q;
        std::memcpy(&q.x, p + L.off_x, 4); std::memcpy(&q.y, p + L.off_y, 4); std::memcpy(&q.z, p + L.off_z, 4);
        if (L.has_intensity) std::memcpy(&q.intensity, p + L.off_intensity, 4);
        if (!(q.x * q.x + q.y * q.y + q.z * q.z < 0.01)) {  // utils::pointAssociateToMap, superodom_utils.cpp:148-158
          const Vector3d w = qrot(q_w_curr, Vector3d(q.x, q.y, q.z));
          q.x = (float)(w.x() + t_w_curr.x()); q.y = (float)(w.y() + t_w_curr.y()); q.z = (float)(w.z() + t_w_curr.z());
        }
        if (q.x * q.x + q.y * q.y + q.z * q.z > 0.01) out[kept++] = q;
      }
      m.data.resize(kept * sizeof(Point));
    }
    m.width = (uint32_t)kept; m.row_step = m.point_step * m.width;
    m.header.stamp = stamp; m.header.frame_id = config_.WORLD_FRAME;
    if (!published_in_place) publish(P + "/registered_scan", "sensor_msgs/msg/PointCloud2", m);
  }
  lap_sub(6);

  so_wire::Odometry odomAftMapped;  // :504-525
  odomAftMapped.header.frame_id = config_.WORLD_FRAME;
  odomAftMapped.child_frame_id = config_.SENSOR_FRAME;
  odomAftMapped.header.stamp = stamp;
  odomAftMapped.pose.pose.orientation = {q_w_curr.x(), q_w_curr.y(), q_w_curr.z(), q_w_curr.w()};
  odomAftMapped.pose.pose.position = {t_w_curr.x(), t_w_curr.y(), t_w_curr.z()};
  odomAftMapped.twist.twist.linear = {vel_b.x(), vel_b.y(), vel_b.z()};
  odomAftMapped.twist.twist.angular = {ang_vel_b.x(), ang_vel_b.y(), ang_vel_b.z()};

  so_wire::Odometry inc;  // :527-559 (on the first frame `initialization` is already true here: updatePoseAndPublish sets it before)
  inc.header.stamp = stamp; inc.header.frame_id = config_.WORLD_FRAME; inc.child_frame_id = config_.SENSOR_FRAME;
  if (!initialization) {
    inc.pose.pose.position = {t_w_curr.x(), t_w_curr.y(), t_w_curr.z()};
    inc.pose.pose.orientation = {q_w_curr.x(), q_w_curr.y(), q_w_curr.z(), q_w_curr.w()};
  } else {
    laser_incremental_T = T_w_lidar;  // (`rot.normalized()` at :545 discards its result)
    inc.pose.pose.position = {laser_incremental_T.pos.x(), laser_incremental_T.pos.y(), laser_incremental_T.pos.z()};
    inc.pose.pose.orientation = {laser_incremental_T.rot.x(), laser_incremental_T.rot.y(), laser_incremental_T.rot.z(), laser_incremental_T.rot.w()};
  }
  publish(P + "/aft_mapped_to_init_incremental", "nav_msgs/msg/Odometry", inc);

  odomAftMapped.pose.covariance[0] = slam.isDegenerate ? 1 : 0;  // :563-567
  publish(P + "/laser_odometry", "nav_msgs/msg/Odometry", odomAftMapped);

  so_wire::PoseStamped ps;  // :569-575
  ps.header = odomAftMapped.header; ps.pose = odomAftMapped.pose.pose;
  laserAfterMappedPath.header.stamp = odomAftMapped.header.stamp;
  laserAfterMappedPath.header.frame_id = config_.WORLD_FRAME;
  laserAfterMappedPath.poses.push_back(ps);
  publish(P + "/laser_odom_path", "nav_msgs/msg/Path", laserAfterMappedPath);

  slam.stats.header = odomAftMapped.header;  // :578-596
  slam.stats.latency = 0;  // timeLatestImuOdometry - now() of a clock this shell does not have: the node sets it to 0 on its first turn too
  slam.stats.n_iterations = (int32_t)slam.stats.iterations.size();
  while (slam.stats.iterations.size() < 4) slam.stats.iterations.push_back(so_wire::IterationStats());  // "avoid breaking rqt_multiplot"
  publish(P + "/super_odometry_stats", "super_odometry_msgs/msg/OptimizationStats", slam.stats);
  slam.stats.iterations.clear();
}

bool laserMapping::processOnce() {  // the loop body of process(), :768-793
  using clk = std::chrono::steady_clock;
  auto lap = [this](int k, clk::time_point& t) { const auto n = clk::now(); phase_seconds[k] += std::chrono::duration<double>(n - t).count(); t = n; };
  auto t = clk::now();
  {
    std::lock_guard<std::mutex> lk(mBuf);
    if (!checkDataAvailable()) return false;
    sensorMeas = extractSensorData();
    clearSensorData();
  }
  lap(0, t);
  try {
    setInitialGuess();
    lap(1, t);
    adjustVoxelSize();
    lap(2, t);
    performSLAMOptimization();
    lap(3, t);
    if (initialization) {
      // EstimateLidarUncertainty publishes the six Float32 topics from inside Localization (LidarSlam.cpp:47, 965-966) on every
      // frame after the seeding one; note the missing '/' after ProjectName (LidarSlam.cpp:22-27)
      const char* names[6] = {"uncertainty_X", "uncertainty_Y", "uncertainty_Z", "uncertainty_roll", "uncertainty_pitch", "uncertainty_yaw"};
      const double u[6] = {slam.stats.uncertainty_x, slam.stats.uncertainty_y, slam.stats.uncertainty_z, slam.stats.uncertainty_roll, slam.stats.uncertainty_pitch, slam.stats.uncertainty_yaw};
      for (int k = 0; k < 6; ++k) { so_wire::Float32 f; f.data = (float)u[k]; publish(config_.ProjectName + names[k], "std_msgs/msg/Float32", f); }
    }
    updatePoseAndPublish();
    lap(4, t);
  } catch (const std::exception& e) {  // RCLCPP_ERROR("Error in frame processing: %s") and on to the next frame
    ++frames_failed;
    last_error = e.what();
  }
  return true;
}

}  // namespace super_odometry_soicp
