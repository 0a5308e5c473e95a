// lidar_slam_soicp.cpp -- see lidar_slam_soicp.h.  What LidarSLAM::Localization (src/LidarProcess/LidarSlam.cpp:30-51)
// and its read-backs become on top of the C ABI; errors travel as exceptions because that is the reference's convention
// on this path (process() catches std::exception, logs and continues: src/LaserMapping/laserMapping.cpp:788-790).
#include "lidar_slam_soicp.h"

#include <stdexcept>

namespace super_odometry_soicp {

LidarSLAM::~LidarSLAM() { if (gpu_) so_icp_destroy(gpu_); }

void LidarSLAM::ensure_context() {
  if (gpu_) return;
  so_icp_config cfg;
  so_icp_default_config(&cfg);
  cfg.device_id = device_id;
  cfg.max_iterations = (int)LocalizationICPMaxIter;             // laserMapping.cpp:108
  cfg.max_surface_features = OptSet.max_surface_features;       // :111
  cfg.velocity_failure_threshold = OptSet.velocity_failure_threshold;  // :110
  cfg.yaw_ratio = OptSet.yaw_ratio;                             // :112
  cfg.line_res = localMap.lineRes_; cfg.plane_res = localMap.planeRes_;  // :103-104
  gpu_ = so_icp_create(&cfg);
  if (!gpu_) throw std::runtime_error(std::string("so_icp_create: ") + so_icp_last_error(nullptr));
}

void LocalMapFacade::setOrigin(const Vector3d& t) {
  owner_->ensure_context();
  const double tt[3] = {t.x(), t.y(), t.z()};
  if (so_icp_map_set_origin(*ctx_, tt, nullptr) < 0) throw std::runtime_error(so_icp_last_error(*ctx_));
}
void LocalMapFacade::addSurfPointCloud(const PointCloud<Point>& cloud) {
  owner_->ensure_context();
  if (so_icp_set_resolution(*ctx_, lineRes_, planeRes_) < 0) throw std::runtime_error(so_icp_last_error(*ctx_));
  if (cloud.points.empty()) return;
  if (so_icp_map_add_surf(*ctx_, reinterpret_cast<const float*>(cloud.points.data()), cloud.points.size(), sizeof(Point)) < 0)
    throw std::runtime_error(so_icp_last_error(*ctx_));
}
PointCloud<Point> LocalMapFacade::export_points(int only_5x5, const int pos[3]) {
  PointCloud<Point> out;
  if (!*ctx_) return out;
  size_t n = 0;
  if (so_icp_map_size(*ctx_, &n, nullptr) < 0) throw std::runtime_error(so_icp_last_error(*ctx_));
  std::vector<float> xyz(3 * (n ? n : 1));
  if (so_icp_map_export(*ctx_, xyz.data(), n, &n, only_5x5, pos) < 0) throw std::runtime_error(so_icp_last_error(*ctx_));
  out.points.resize(n);
  for (size_t i = 0; i < n; ++i) { out.points[i].x = xyz[3 * i]; out.points[i].y = xyz[3 * i + 1]; out.points[i].z = xyz[3 * i + 2]; }
  return out;
}
size_t LocalMapFacade::exportRecords(void* out, size_t cap_points, bool only_5x5, const Vector3i& p) {
  if (!*ctx_) return 0;
  const int pos[3] = {p.x(), p.y(), p.z()};
  size_t n = 0;
  if (so_icp_map_export_records(*ctx_, out, sizeof(Point), cap_points, &n, only_5x5 ? 1 : 0, pos) < 0) throw std::runtime_error(so_icp_last_error(*ctx_));
  return n;
}
PointCloud<Point> LocalMapFacade::get5x5LocalMap(const Vector3i& p) { const int pos[3] = {p.x(), p.y(), p.z()}; return export_points(1, pos); }
PointCloud<Point> LocalMapFacade::getAllLocalMap() { const int pos[3] = {0, 0, 0}; return export_points(0, pos); }

void LidarSLAM::StageNextScan(const PointCloud<Point>::Ptr& planner_point) {
  ensure_context();
  if (!planner_point || planner_point->points.empty()) return;
  if (so_icp_stage_scan(gpu_, reinterpret_cast<const float*>(planner_point->points.data()), planner_point->points.size(), sizeof(Point)) < 0)
    throw std::runtime_error(so_icp_last_error(gpu_));
}

void LidarSLAM::push_knobs() {
  // knobs the node writes into public fields before every call (laserMapping.cpp:648-649, 703-711)
  if (so_icp_set_resolution(gpu_, localMap.lineRes_, localMap.planeRes_) < 0) throw std::runtime_error(so_icp_last_error(gpu_));
  so_icp_set_max_surface_features(gpu_, OptSet.max_surface_features);
  so_icp_set_max_iterations(gpu_, (int)LocalizationICPMaxIter);
}

void LidarSLAM::read_back(int32_t n_edge_points, const double T_out[7], double timeLaserOdometry) {
  const so_icp_stats& st = last_raw;
  last_flags = st.flags;
  T_w_lidar.pos = Vector3d(T_out[0], T_out[1], T_out[2]);                 // read back at laserMapping.cpp:734-737
  T_w_lidar.rot = Quaterniond(T_out[6], T_out[3], T_out[4], T_out[5]);
  last_T_w_lidar = T_w_lidar;                                            // LidarSlam.cpp:197
  lasttimeLaserOdometry = timeLaserOdometry;
  if (last_status == SO_ICP_MAP_SEEDED) return;  // initialization == false (LidarSlam.cpp:45-46): initializeMapping only
  // from here on the call went through EstimateLidarUncertainty (:47) and prepareOptimizationState (:361-377)
  pos_in_localmap = Vector3i(st.pos_in_localmap[0], st.pos_in_localmap[1], st.pos_in_localmap[2]);  // :363, read at laserMapping.cpp:439
  stats.laser_cloud_surf_from_map_num = st.laser_cloud_surf_from_map_num;
  stats.laser_cloud_surf_stack_num = st.laser_cloud_surf_stack_num;
  stats.laser_cloud_corner_from_map_num = 0;                  // no corner map exists: nothing calls addEdgePointCloud
  stats.laser_cloud_corner_stack_num = n_edge_points;         // EdgesPoints->size(), LidarSlam.cpp:374
  stats.uncertainty_x = st.uncertainty[0]; stats.uncertainty_y = st.uncertainty[1]; stats.uncertainty_z = st.uncertainty[2];
  stats.uncertainty_roll = st.uncertainty[3]; stats.uncertainty_pitch = st.uncertainty[4]; stats.uncertainty_yaw = st.uncertainty[5];
  stats.iterations.clear();                                   // updateFeatureStats, :376
  if (last_status != SO_ICP_OK) return;  // "Not enough features for optimization" (:113-116): the rest of the statistics keeps its values
  startupCount = st.startup_count;                                       // laserMapping.cpp:738
  // OptimizationStats message (laserMapping.cpp:581-596; LidarSlam.cpp:198-210, 242-251, 969-974)
  stats.total_translation = st.total_translation; stats.total_rotation = st.total_rotation;
  stats.translation_from_last = st.translation_from_last; stats.rotation_from_last = st.rotation_from_last;
  stats.time_elapsed = st.time_elapsed_ms;
  stats.prediction_source = 0;
  for (int i = 0; i < st.n_iterations; ++i) {
    IterationStats it;
    it.translation_norm = st.iterations[i].translation_norm; it.rotation_norm = st.iterations[i].rotation_norm;
    it.num_surf_from_scan = st.iterations[i].num_surf_from_scan; it.num_corner_from_scan = 0;
    stats.iterations.push_back(it);
  }
}

void LidarSLAM::Localization(bool initialization, PredictionSource /*predictodom: only the dead VIO prior reads it, LidarSlam.cpp:281-283*/,
                             Transformd position, PointCloud<Point>::Ptr edge_point /*dead path, LidarSlam.cpp:402-512: only its size is reported*/,
                             PointCloud<Point>::Ptr planner_point, double timeLaserOdometry) {
  ensure_context();
  push_knobs();
  const double T_in[7] = {position.pos.x(), position.pos.y(), position.pos.z(),
                          position.rot.x(), position.rot.y(), position.rot.z(), position.rot.w()};
  double T_out[7];
  const float* xyz = planner_point && !planner_point->points.empty() ? reinterpret_cast<const float*>(planner_point->points.data()) : nullptr;
  const size_t n = planner_point ? planner_point->points.size() : 0;
  last_status = so_icp_localization(gpu_, initialization ? 1 : 0, T_in, xyz, n, sizeof(Point), timeLaserOdometry, T_out, &last_raw);
  if (last_status < 0) throw std::runtime_error(std::string("so_icp_localization: ") + so_icp_last_error(gpu_));
  read_back(edge_point ? (int32_t)edge_point->points.size() : 0, T_out, timeLaserOdometry);
}

void LidarSLAM::AnnounceSurf(const float* xyz, size_t n, size_t stride_bytes) {
  if (gpu_ && xyz && n) (void)so_icp_prefilter_announce(gpu_, xyz, n, stride_bytes);  // (a refused announcement only means "not staged ahead")
}

void LidarSLAM::PrefilterSurf(const float* xyz, size_t n, size_t stride_bytes, bool auto_voxel_size, float line_res, float plane_res,
                              so_icp_prefilter_info* info, const void** d_filtered, size_t* n_filtered) {
  ensure_context();
  void* d = nullptr;
  const int rc = so_icp_prefilter_scan(gpu_, xyz, n, stride_bytes, auto_voxel_size ? 1 : 0, line_res, plane_res, &d, n_filtered, info);
  if (rc < 0) throw std::runtime_error(std::string("so_icp_prefilter_scan: ") + so_icp_last_error(gpu_));
  *d_filtered = d;
  if (info) { localMap.lineRes_ = info->line_res; localMap.planeRes_ = info->plane_res; }  // laserMapping.cpp:648-649
}

size_t LidarSLAM::TransformCloud(void* points, size_t n, size_t stride_bytes, const Transformd& T, std::vector<uint8_t>& keep) {
  ensure_context();
  const double Tw[7] = {T.pos.x(), T.pos.y(), T.pos.z(), T.rot.x(), T.rot.y(), T.rot.z(), T.rot.w()};
  keep.resize(n);
  size_t kept = 0;
  if (so_icp_transform_cloud(gpu_, points, n, stride_bytes, Tw, keep.data(), &kept) < 0)
    throw std::runtime_error(std::string("so_icp_transform_cloud: ") + so_icp_last_error(gpu_));
  return kept;
}

uint8_t* LidarSLAM::PinnedScratch(int which, size_t bytes) {
  ensure_context();
  if ((size_t)which >= scratch_.size()) scratch_.resize((size_t)which + 1);
  Scratch& sc = scratch_[(size_t)which];
  if (sc.cap < bytes) {
    if (sc.p) so_icp_host_free(gpu_, sc.p);
    sc.p = nullptr; sc.cap = 0;
    void* p = nullptr;
    const size_t want = bytes + bytes / 4 + 4096;
    if (so_icp_host_alloc(gpu_, want, &p) < 0) throw std::runtime_error(std::string("so_icp_host_alloc: ") + so_icp_last_error(gpu_));
    sc.p = static_cast<uint8_t*>(p); sc.cap = want;
  }
  return sc.p;
}

void LidarSLAM::LocalizationPrefiltered(bool initialization, PredictionSource, Transformd position, const void* d_planner_xyz, size_t n_planner,
                                        int32_t n_edge_points, double timeLaserOdometry) {
  ensure_context();
  push_knobs();
  const double T_in[7] = {position.pos.x(), position.pos.y(), position.pos.z(),
                          position.rot.x(), position.rot.y(), position.rot.z(), position.rot.w()};
  double T_out[7];
  last_status = so_icp_localization_dev(gpu_, initialization ? 1 : 0, T_in, d_planner_xyz, n_planner, timeLaserOdometry, T_out, &last_raw);
  if (last_status < 0) throw std::runtime_error(std::string("so_icp_localization_dev: ") + so_icp_last_error(gpu_));
  read_back(n_edge_points, T_out, timeLaserOdometry);
}

}  // namespace super_odometry_soicp
