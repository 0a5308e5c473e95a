// adapter_driver.cpp -- C++ caller of the boundary (test binary, built by __graft_entry__.build(), run by a -m gpu test):
// drives the adapter exactly like laserMapping::performSLAMOptimization does (src/LaserMapping/laserMapping.cpp:703-741):
// first frame with initialization == false (map seeding), then Localization() per frame, reading back the public fields.
//
//   adapter_driver <in.bin> <out.bin>
// in.bin : int32 n_frames, float32 plane_res, int32 max_iterations, int32 max_surface_features; per frame: int32 n,
//          float64 guess[7], float64 time, float32 xyz[n][3]  (sensor frame, as the node passes it)
// out.bin: per frame: int32 status, int32 startupCount, int32 pos_in_localmap[3], float64 T_w_lidar[7] (tx ty tz qx qy qz qw),
//          int32 n_iterations, int32 surf_from_map, float64 total_translation, float64 uncertainty[6],
//          int32 num_surf of the last iteration, uint32 flags;  then: uint64 map size, float32 map xyz (5x5 neighbourhood)
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <vector>

#include "lidar_slam_soicp.h"

using namespace super_odometry_soicp;

template <typename T> static T rd(FILE* f) { T v; if (fread(&v, sizeof(T), 1, f) != 1) throw std::runtime_error("short input"); return v; }
template <typename T> static void wr(FILE* f, const T& v) { fwrite(&v, sizeof(T), 1, f); }

int main(int argc, char** argv) {
  if (argc < 3) { fprintf(stderr, "usage: %s in.bin out.bin\n", argv[0]); return 2; }
  try {
    FILE* in = fopen(argv[1], "rb");
    FILE* out = fopen(argv[2], "wb");
    if (!in || !out) throw std::runtime_error("cannot open files");
    const int n_frames = rd<int32_t>(in);
    LidarSLAM slam;
    slam.localMap.planeRes_ = rd<float>(in);                    // laserMapping.cpp:103-104
    slam.localMap.lineRes_ = slam.localMap.planeRes_ / 2;
    slam.LocalizationICPMaxIter = (size_t)rd<int32_t>(in);      // :108
    slam.OptSet.max_surface_features = rd<int32_t>(in);         // :111
    std::vector<PointCloud<Point>::Ptr> clouds;
    std::vector<Transformd> guesses;
    std::vector<double> times;
    for (int f = 0; f < n_frames; ++f) {
      const int n = rd<int32_t>(in);
      double g[7];
      for (double& v : g) v = rd<double>(in);
      times.push_back(rd<double>(in));
      auto cloud = std::make_shared<PointCloud<Point>>();
      cloud->points.resize(n);
      for (int i = 0; i < n; ++i) { cloud->points[i].x = rd<float>(in); cloud->points[i].y = rd<float>(in); cloud->points[i].z = rd<float>(in); cloud->points[i].intensity = (float)i; }
      clouds.push_back(cloud);
      Transformd T;
      T.pos = Vector3d(g[0], g[1], g[2]); T.rot = Quaterniond(g[6], g[3], g[4], g[5]);
      guesses.push_back(T);
    }
    auto edge = std::make_shared<PointCloud<Point>>();  // dead path: the node still passes it
    bool initialization = false;                        // laserMapping.cpp: first frame seeds the map
    for (int f = 0; f < n_frames; ++f) {
      if (f + 1 < n_frames && f >= 1) slam.StageNextScan(clouds[f + 1]);  // what the feature callback would do
      slam.Localization(initialization, LidarSLAM::PredictionSource::LIO_ODOM, guesses[f], edge, clouds[f], times[f]);
      initialization = true;
      slam.frame_count = f; slam.laser_imu_sync = 1;    // :740-741
      wr<int32_t>(out, slam.last_status); wr<int32_t>(out, slam.startupCount);
      wr<int32_t>(out, slam.pos_in_localmap.x()); wr<int32_t>(out, slam.pos_in_localmap.y()); wr<int32_t>(out, slam.pos_in_localmap.z());
      wr<double>(out, slam.T_w_lidar.pos.x()); wr<double>(out, slam.T_w_lidar.pos.y()); wr<double>(out, slam.T_w_lidar.pos.z());
      wr<double>(out, slam.T_w_lidar.rot.x()); wr<double>(out, slam.T_w_lidar.rot.y()); wr<double>(out, slam.T_w_lidar.rot.z()); wr<double>(out, slam.T_w_lidar.rot.w());
      wr<int32_t>(out, (int32_t)slam.stats.iterations.size()); wr<int32_t>(out, slam.stats.laser_cloud_surf_from_map_num);
      wr<double>(out, slam.stats.total_translation);
      wr<double>(out, slam.stats.uncertainty_x); wr<double>(out, slam.stats.uncertainty_y); wr<double>(out, slam.stats.uncertainty_z);
      wr<double>(out, slam.stats.uncertainty_roll); wr<double>(out, slam.stats.uncertainty_pitch); wr<double>(out, slam.stats.uncertainty_yaw);
      wr<int32_t>(out, slam.stats.iterations.empty() ? 0 : slam.stats.iterations.back().num_surf_from_scan);
      wr<uint32_t>(out, slam.last_flags);
      // the node pads / clears the iteration list after publishing (laserMapping.cpp:588-596)
      slam.stats.iterations.clear();
    }
    const PointCloud<Point> near = slam.localMap.get5x5LocalMap(slam.pos_in_localmap);  // laserMapping.cpp:439
    wr<uint64_t>(out, (uint64_t)near.points.size());
    for (const Point& p : near.points) { wr<float>(out, p.x); wr<float>(out, p.y); wr<float>(out, p.z); }
    fclose(in); fclose(out);
  } catch (const std::exception& e) {
    fprintf(stderr, "adapter_driver: %s\n", e.what());  // what process() would log (laserMapping.cpp:788-790)
    return 1;
  }
  return 0;
}
