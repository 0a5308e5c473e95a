// wire_selftest.cpp -- host-only check binary for wire/cdr.h (built by __graft_entry__.build(), run by
// tests/test_wire_formats.py in the CPU suite):
//   wire_selftest roundtrip <Type> <in.cdr> <out.cdr>   deserialise a message of <Type>, serialise it again
//   wire_selftest emit <Type> <out.cdr>                 serialise a message with fixed field values (the test knows them)
//   wire_selftest params <file.yaml>                    node_config.h: ROS 2 parameter file -> NodeConfig, printed
// Types: String Float32 Header PointCloud2 Odometry Path IterationStats OptimizationStats LaserFeature
#include <cstdio>
#include <cstring>
#include <string>

#include "node_config.h"
#include "pcd_io.h"
#include "wire/cdr.h"

using namespace so_wire;

static std::vector<uint8_t> slurp(const char* path) {
  FILE* f = fopen(path, "rb");
  if (!f) throw std::runtime_error(std::string("cannot open ") + path);
  std::vector<uint8_t> b;
  uint8_t tmp[65536];
  size_t k;
  while ((k = fread(tmp, 1, sizeof(tmp), f)) > 0) b.insert(b.end(), tmp, tmp + k);
  fclose(f);
  return b;
}
static void spit(const char* path, const std::vector<uint8_t>& b) {
  FILE* f = fopen(path, "wb");
  if (!f) throw std::runtime_error(std::string("cannot open ") + path);
  fwrite(b.data(), 1, b.size(), f);
  fclose(f);
}
template <typename M> static std::vector<uint8_t> again(const std::vector<uint8_t>& in) { return serialize(deserialize<M>(in)); }

static PointCloud2 sample_cloud(int n, const char* frame) {
  PointCloud2 c;
  c.header.stamp = {12, 500000000u}; c.header.frame_id = frame;
  c.height = 1; c.width = (uint32_t)n;
  const char* names[4] = {"x", "y", "z", "intensity"};
  const uint32_t offs[4] = {0, 4, 8, 16};
  for (int k = 0; k < 4; ++k) { PointField f; f.name = names[k]; f.offset = offs[k]; f.datatype = PointField::FLOAT32; f.count = 1; c.fields.push_back(f); }
  c.point_step = 32; c.row_step = 32u * n; c.is_dense = true;
  c.data.resize(32u * n);
  for (int i = 0; i < n; ++i) {
    const float v[8] = {1.0f * i, 0.5f * i, -0.25f * i, 1.0f, 100.0f + i, 0, 0, 0};
    std::memcpy(c.data.data() + 32 * i, v, 32);
  }
  return c;
}

int main(int argc, char** argv) {
  const uint16_t probe = 1;
  if (*reinterpret_cast<const uint8_t*>(&probe) != 1) { fprintf(stderr, "big-endian host: CdrWriter::prim needs a byte swap\n"); return 3; }
  try {
    if (argc == 5 && !strcmp(argv[1], "roundtrip")) {
      const std::string t = argv[2];
      const std::vector<uint8_t> in = slurp(argv[3]);
      std::vector<uint8_t> out;
      if (t == "String") out = again<String>(in);
      else if (t == "Float32") out = again<Float32>(in);
      else if (t == "Header") out = again<Header>(in);
      else if (t == "PointCloud2") out = again<PointCloud2>(in);
      else if (t == "Odometry") out = again<Odometry>(in);
      else if (t == "Path") out = again<Path>(in);
      else if (t == "IterationStats") out = again<IterationStats>(in);
      else if (t == "OptimizationStats") out = again<OptimizationStats>(in);
      else if (t == "LaserFeature") out = again<LaserFeature>(in);
      else throw std::runtime_error("unknown type " + t);
      spit(argv[4], out);
      return 0;
    }
    if (argc == 5 && !strcmp(argv[1], "roundtrip-many")) {  // in / out: concatenated [uint32 length][message]; a message that fails to parse yields length 0xFFFFFFFF
      const std::string t = argv[2];
      const std::vector<uint8_t> in = slurp(argv[3]);
      std::vector<uint8_t> out;
      size_t at = 0;
      while (at + 4 <= in.size()) {
        uint32_t len;
        std::memcpy(&len, in.data() + at, 4);
        at += 4;
        if (len > in.size() - at) throw std::runtime_error("short frame");
        const std::vector<uint8_t> one(in.begin() + at, in.begin() + at + len);
        at += len;
        std::vector<uint8_t> res;
        uint32_t rl = 0xFFFFFFFFu;
        try {
          if (t == "LaserFeature") res = again<LaserFeature>(one);
          else if (t == "OptimizationStats") res = again<OptimizationStats>(one);
          else if (t == "Odometry") res = again<Odometry>(one);
          else if (t == "Path") res = again<Path>(one);
          else if (t == "PointCloud2") res = again<PointCloud2>(one);
          else throw std::invalid_argument("unknown type " + t);
          rl = (uint32_t)res.size();
        } catch (const std::runtime_error&) { res.clear(); }
        const uint8_t* pl = reinterpret_cast<const uint8_t*>(&rl);
        out.insert(out.end(), pl, pl + 4);
        out.insert(out.end(), res.begin(), res.end());
      }
      spit(argv[4], out);
      return 0;
    }
    if (argc == 4 && !strcmp(argv[1], "emit")) {
      const std::string t = argv[2];
      std::vector<uint8_t> out;
      if (t == "String") { String m; m.data = "hello"; out = serialize(m); }
      else if (t == "Float32") { Float32 m; m.data = 0.75f; out = serialize(m); }
      else if (t == "Header") { Header m; m.stamp = {1700000000, 123456789u}; m.frame_id = "sensor_init"; out = serialize(m); }
      else if (t == "PointCloud2") out = serialize(sample_cloud(3, "sensor"));
      else if (t == "Odometry") {
        Odometry m;
        m.header.stamp = {7, 250000000u}; m.header.frame_id = "sensor_init"; m.child_frame_id = "sensor";
        m.pose.pose.position = {1.5, -2.5, 3.25}; m.pose.pose.orientation = {0.1, 0.2, 0.3, 0.9};
        for (int i = 0; i < 36; ++i) { m.pose.covariance[i] = i; m.twist.covariance[i] = -i; }
        m.twist.twist.linear = {0.5, 0.25, 0.125}; m.twist.twist.angular = {-1, -2, -3};
        out = serialize(m);
      } else if (t == "Path") {
        Path m;
        m.header.frame_id = "w";
        for (int i = 0; i < 2; ++i) { PoseStamped p; p.header.stamp = {i, 0}; p.header.frame_id = i ? "ab" : "abc"; p.pose.position = {1.0 * i, 2.0 * i, 3.0 * i}; m.poses.push_back(p); }
        out = serialize(m);
      } else if (t == "OptimizationStats") {
        OptimizationStats m;
        m.header.stamp = {3, 4}; m.header.frame_id = "sensor_init";
        m.laser_cloud_surf_from_map_num = 11; m.laser_cloud_corner_from_map_num = 12; m.laser_cloud_surf_stack_num = 13; m.laser_cloud_corner_stack_num = 14;
        m.total_translation = 0.5; m.total_rotation = 0.25; m.translation_from_last = 0.125; m.rotation_from_last = 0.0625; m.time_elapsed = 1.5; m.latency = 2.5;
        m.n_iterations = 2; m.average_distance = 30.5;
        m.uncertainty_x = 0.1; m.uncertainty_y = 0.2; m.uncertainty_z = 0.3; m.uncertainty_roll = 0.4; m.uncertainty_pitch = 0.5; m.uncertainty_yaw = 0.6;
        m.plane_match_success = 21; m.plane_no_enough_neighbor = 22; m.plane_neighbor_too_far = 23; m.plane_badpca_structure = 24;
        m.plane_invalid_numerical = 25; m.plane_mse_too_large = 26; m.plane_unknown = 27; m.prediction_source = 1;
        for (int i = 0; i < 2; ++i) { IterationStats it; it.header.frame_id = i ? "q" : ""; it.translation_norm = 0.01 * (i + 1); it.rotation_norm = 0.02 * (i + 1); it.num_surf_from_scan = 1000 + i; m.iterations.push_back(it); }
        out = serialize(m);
      } else if (t == "LaserFeature") {
        LaserFeature m;
        m.header.stamp = {12, 500000000u}; m.header.frame_id = "sensor";
        m.sensor = 1; m.imu_available = 1; m.odom_available = 0;
        m.imu_quaternion_x = 0.01; m.imu_quaternion_y = 0.02; m.imu_quaternion_z = 0.03; m.imu_quaternion_w = 0.99;
        m.initial_pose_x = 1; m.initial_pose_y = 2; m.initial_pose_z = 3;
        m.initial_quaternion_x = 0; m.initial_quaternion_y = 0; m.initial_quaternion_z = 0; m.initial_quaternion_w = 1;
        m.imu_preintegration_reset_id = -5;
        m.cloud_nodistortion = sample_cloud(4, "sensor"); m.cloud_corner = sample_cloud(1, "sensor"); m.cloud_surface = sample_cloud(3, "sensor"); m.cloud_realsense = sample_cloud(0, "");
        out = serialize(m);
      } else throw std::runtime_error("unknown type " + t);
      spit(argv[3], out);
      return 0;
    }
    if (argc == 3 && !strcmp(argv[1], "params")) {  // the node's parameter surface: file -> NodeConfig, printed as key=value
      const super_odometry_soicp::NodeConfig c = super_odometry_soicp::load_node_config(argv[2]);
      printf("lineRes=%.9g\nplaneRes=%.9g\nmax_iterations=%d\ndebug_view=%d\nenable_ouster_data=%d\npublish_only_feature_points=%d\n", c.lineRes, c.planeRes,
             c.max_iterations, (int)c.debug_view_enabled, (int)c.enable_ouster_data, (int)c.publish_only_feature_points);
      printf("max_surface_features=%d\nvelocity_failure_threshold=%.17g\nauto_voxel_size=%d\nforget_far_chunks=%d\nvisual_confidence_factor=%.17g\n",
             c.max_surface_features, c.velocity_failure_threshold, (int)c.auto_voxel_size, (int)c.forget_far_chunks, c.visual_confidence_factor);
      printf("localization_mode=%d\nmap_dir=%s\ninit=%.9g %.9g %.9g %.9g %.9g %.9g\nuse_imu_roll_pitch=%d\nworld_frame=%s\nsensor_frame=%s\nPROJECT_NAME=%s\n",
             (int)c.localization_mode, c.map_dir.c_str(), c.init_x, c.init_y, c.init_z, c.init_roll, c.init_pitch, c.init_yaw, (int)c.use_imu_roll_pitch,
             c.WORLD_FRAME.c_str(), c.SENSOR_FRAME.c_str(), c.ProjectName.c_str());
      return 0;
    }
    if (argc == 4 && !strcmp(argv[1], "pcd")) {  // the prior-map reader (pcd_io.h): file -> packed float32 {x, y, z, intensity}
      std::vector<float> xyzi;
      std::string err;
      if (!so_pcd::read_xyzi(argv[2], xyzi, err)) { fprintf(stderr, "%s\n", err.c_str()); return 3; }
      spit(argv[3], std::vector<uint8_t>(reinterpret_cast<const uint8_t*>(xyzi.data()), reinterpret_cast<const uint8_t*>(xyzi.data()) + xyzi.size() * 4));
      printf("points=%zu\n", xyzi.size() / 4);
      return 0;
    }
    fprintf(stderr, "usage: %s roundtrip <Type> in out | emit <Type> out | params <file.yaml> | pcd <file.pcd> out.f32\n", argv[0]);
    return 2;
  } catch (const std::exception& e) {
    fprintf(stderr, "wire_selftest: %s\n", e.what());
    return 1;
  }
}
