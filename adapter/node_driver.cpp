// node_driver.cpp -- runs the laser_mapping_node shell (laser_mapping_soicp.{h,cpp}) over a recorded sequence of
// serialised super_odometry_msgs/LaserFeature messages, the way a rosbag2 replay feeds the reference node, and records
// everything the node publishes (test binary: built by __graft_entry__.build(), run by tests/test_gpu_node.py).
//
//   node_driver <bag.bin> <out.bin> [params.yaml [prior_map.f32]]
// bag.bin: float32 planeRes, float32 lineRes, int32 max_iterations, int32 max_surface_features, int32 auto_voxel_size,
//          int32 debug_view, int32 n_messages; per message: uint32 length, CDR bytes
// out.bin: per published message: uint32 frame, uint32 len + topic, uint32 len + type, uint32 len + CDR bytes;
//          trailer: uint32 0xFFFFFFFF, int32 frames_failed, uint32 len + last error text
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <stdexcept>

#include "laser_mapping_soicp.h"
#include "node_config.h"

using namespace super_odometry_soicp;

template <typename T> static T rd(FILE* f) { T v; if (fread(&v, sizeof(T), 1, f) != 1) throw std::runtime_error("short input"); return v; }
template <typename T> static void wr(FILE* f, const T& v) { fwrite(&v, sizeof(T), 1, f); }
static void wr_blob(FILE* f, const void* p, size_t n) { wr<uint32_t>(f, (uint32_t)n); fwrite(p, 1, n, f); }

struct Recorder : Outbox {
  FILE* f = nullptr;
  uint32_t frame = 0;
  void publish(const std::string& topic, const std::string& type, std::vector<uint8_t>&& cdr) override {
    publish_bytes(topic, type, cdr.data(), cdr.size());
  }
  void publish_bytes(const std::string& topic, const std::string& type, const uint8_t* cdr, size_t n) override {  // (a bridge that sends from a pointer)
    wr<uint32_t>(f, frame); wr_blob(f, topic.data(), topic.size()); wr_blob(f, type.data(), type.size()); wr_blob(f, cdr, n);
  }
};

int main(int argc, char** argv) {
  if (argc < 3) { fprintf(stderr, "usage: %s bag.bin out.bin\n", argv[0]); return 2; }
  try {
    FILE* in = fopen(argv[1], "rb");
    Recorder rec;
    rec.f = fopen(argv[2], "wb");
    if (!in || !rec.f) throw std::runtime_error("cannot open files");
    NodeConfig cfg;
    cfg.planeRes = rd<float>(in); cfg.lineRes = rd<float>(in);
    cfg.max_iterations = rd<int32_t>(in); cfg.max_surface_features = rd<int32_t>(in);
    cfg.auto_voxel_size = rd<int32_t>(in) != 0; cfg.debug_view_enabled = rd<int32_t>(in) != 0;
    cfg.ProjectName = "/super_odometry";
    if (argc >= 4) {  // a ROS 2 parameter file in the reference's layout takes the place of the bag header's knobs
      const NodeConfig from_file = load_node_config(argv[3]);
      const std::string project = from_file.ProjectName.empty() ? cfg.ProjectName : from_file.ProjectName;
      cfg = from_file;
      cfg.ProjectName = project;
    }
    const int n_msgs = rd<int32_t>(in);
    laserMapping node(cfg, &rec);
    node.initInterface();
    if (argc >= 5) {  // localization mode: the prior map (laserMapping.cpp:161-171 reads map_dir through PCL; here: packed float32 xyz)
      FILE* pm = fopen(argv[4], "rb");
      if (!pm) throw std::runtime_error("cannot open the prior map");
      std::vector<float> xyz;
      float tmp[3 * 4096];
      size_t k;
      while ((k = fread(tmp, sizeof(float), 3 * 4096, pm)) > 0) xyz.insert(xyz.end(), tmp, tmp + k);
      fclose(pm);
      node.loadPriorMap(xyz.data(), xyz.size() / 3, 12);
    } else if (cfg.localization_mode) {  // ... or the .pcd file the parameter file names (map_dir), as the reference does
      if (!node.loadPriorMap()) fprintf(stderr, "%s\n", node.last_error.c_str());
    }
    std::vector<std::vector<uint8_t>> bag(n_msgs);
    for (int k = 0; k < n_msgs; ++k) {
      bag[k].resize(rd<uint32_t>(in));
      if (!bag[k].empty() && fread(bag[k].data(), 1, bag[k].size(), in) != bag[k].size()) throw std::runtime_error("short message");
    }
    double busy = 0;  // seconds inside callback + process() turn, first (seeding) frame excluded
    for (int k = 0; k < n_msgs; ++k) {
      rec.frame = (uint32_t)k;
      const auto t0 = std::chrono::steady_clock::now();
      node.laserFeatureInfoHandler(bag[k].data(), bag[k].size());  // the subscription callback ...
      while (node.processOnce()) {}                                // ... and the process() loop
      if (k) busy += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
      else for (double& v : node.phase_seconds) v = 0;  // (the first frame allocates the device buffers)
    }
    if (n_msgs > 1) {
      fprintf(stderr, "node_driver: %d frames, %.3f ms per frame (deserialise + guess + prefilter + Localization + publish)", n_msgs - 1, 1e3 * busy / (n_msgs - 1));
      const char* names[8] = {"extract", "guess", "adjustVoxelSize", "Localization", "publish", "[publish: map clouds", "registered scan", "outbox (serialise + bridge)"};
      fprintf(stderr, " | of which");
      for (int k = 0; k < 8; ++k) fprintf(stderr, " %s %.3f", names[k], 1e3 * node.phase_seconds[k] / (n_msgs - 1));
      fprintf(stderr, "]");
      fprintf(stderr, "\n");
    }
    wr<uint32_t>(rec.f, 0xFFFFFFFFu); wr<int32_t>(rec.f, node.frames_failed); wr_blob(rec.f, node.last_error.data(), node.last_error.size());
    fclose(in); fclose(rec.f);
  } catch (const std::exception& e) {
    fprintf(stderr, "node_driver: %s\n", e.what());
    return 1;
  }
  return 0;
}
