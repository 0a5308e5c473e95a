// node_config.h -- the node's parameter surface without rclcpp: reads a ROS 2 parameter file of the reference's layout
// (/root/reference/super_odometry/config/*.yaml: "/**:" or a node name, "ros__parameters:", nested maps) and fills
// NodeConfig the way laserMapping::readParameters (src/LaserMapping/laserMapping.cpp:180-247, the declared defaults of
// :182-203 included) and readGlobalparam (src/parameter/parameter.cpp:284-337) do.  The YAML subset is what those files
// use: block maps by indentation, scalars (numbers, true / false, quoted or plain strings), '#' comments.  Sequences,
// anchors and flow style are rejected, not guessed at.  The calibration file (OpenCV FileStorage: imu_laser_R,
// yaw_ratio, parameter.cpp:123-216) is not read: NodeConfig::imu_laser_R / yaw_ratio are set by the caller.
#pragma once
#include <cstdlib>
#include <fstream>
#include <map>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

#include "laser_mapping_soicp.h"

namespace super_odometry_soicp {

using ParamMap = std::map<std::string, std::string>;  // dotted name below ros__parameters -> scalar text (quotes removed)

inline ParamMap parse_ros_params_yaml(const std::string& text) {
  ParamMap out;
  struct Level { int indent; std::string prefix; bool params; };
  std::vector<Level> stack;
  std::istringstream in(text);
  std::string line;
  int lineno = 0;
  while (std::getline(in, line)) {
    ++lineno;
    // strip comments outside quotes
    bool q1 = false, q2 = false;
    size_t cut = line.size();
    for (size_t i = 0; i < line.size(); ++i) {
      if (line[i] == '\'' && !q2) q1 = !q1;
      else if (line[i] == '"' && !q1) q2 = !q2;
      else if (line[i] == '#' && !q1 && !q2 && (i == 0 || line[i - 1] == ' ' || line[i - 1] == '\t')) { cut = i; break; }
    }
    line.resize(cut);
    while (!line.empty() && (line.back() == ' ' || line.back() == '\t' || line.back() == '\r')) line.pop_back();
    size_t ind = 0;
    while (ind < line.size() && line[ind] == ' ') ++ind;
    if (ind == line.size()) continue;
    if (line[ind] == '\t') throw std::runtime_error("params yaml line " + std::to_string(lineno) + ": tab indentation");
    if (line[ind] == '-' || line[ind] == '[' || line[ind] == '{' || line[ind] == '&' || line[ind] == '*')
      throw std::runtime_error("params yaml line " + std::to_string(lineno) + ": sequences / flow style / anchors are not supported");
    // key: the text up to the first ':' that is followed by a space or ends the line (the root key "/**:" has none inside)
    size_t colon = std::string::npos;
    for (size_t i = ind; i < line.size(); ++i)
      if (line[i] == ':' && (i + 1 == line.size() || line[i + 1] == ' ')) { colon = i; break; }
    if (colon == std::string::npos) throw std::runtime_error("params yaml line " + std::to_string(lineno) + ": expected 'key: value'");
    std::string key = line.substr(ind, colon - ind), val = colon + 1 < line.size() ? line.substr(colon + 1) : "";
    while (!key.empty() && key.back() == ' ') key.pop_back();
    if (key.size() >= 2 && ((key.front() == '"' && key.back() == '"') || (key.front() == '\'' && key.back() == '\''))) key = key.substr(1, key.size() - 2);
    size_t vs = 0;
    while (vs < val.size() && val[vs] == ' ') ++vs;
    val = val.substr(vs);
    while (!stack.empty() && stack.back().indent >= (int)ind) stack.pop_back();
    const bool in_params = !stack.empty() && stack.back().params;
    const std::string prefix = stack.empty() ? "" : stack.back().prefix;
    if (val.empty()) {  // a nested map
      Level l;
      l.indent = (int)ind;
      l.params = in_params || key == "ros__parameters";
      l.prefix = in_params ? prefix + key + "." : "";
      stack.push_back(l);
      continue;
    }
    if (!in_params) continue;  // scalars outside ros__parameters are not parameters
    if (val.size() >= 2 && ((val.front() == '"' && val.back() == '"') || (val.front() == '\'' && val.back() == '\''))) val = val.substr(1, val.size() - 2);
    out[prefix + key] = val;
  }
  return out;
}

namespace detail {
inline bool has(const ParamMap& p, const std::string& k) { return p.find(k) != p.end(); }
inline double num(const ParamMap& p, const std::string& k, double dflt) {
  if (!has(p, k)) return dflt;
  char* end = nullptr;
  const std::string& s = p.at(k);
  const double v = std::strtod(s.c_str(), &end);
  if (end == s.c_str() || *end) throw std::runtime_error("parameter " + k + ": '" + s + "' is not a number");  // rclcpp: InvalidParameterTypeException
  return v;
}
inline bool flag(const ParamMap& p, const std::string& k, bool dflt) {
  if (!has(p, k)) return dflt;
  const std::string& s = p.at(k);
  if (s == "true" || s == "True" || s == "TRUE") return true;
  if (s == "false" || s == "False" || s == "FALSE") return false;
  throw std::runtime_error("parameter " + k + ": '" + s + "' is not a bool");
}
inline std::string str(const ParamMap& p, const std::string& k, const std::string& dflt) { return has(p, k) ? p.at(k) : dflt; }
}  // namespace detail

inline NodeConfig node_config_from_params(const ParamMap& p) {
  using namespace detail;
  const std::string n = "laser_mapping_node.";
  NodeConfig c;
  c.lineRes = (float)num(p, n + "mapping_line_resolution", 0.1);      // laserMapping.cpp:183-203: the declared defaults
  c.planeRes = (float)num(p, n + "mapping_plane_resolution", 0.2);
  c.max_iterations = (int)num(p, n + "max_iterations", 4);
  c.debug_view_enabled = flag(p, n + "debug_view", false);
  c.enable_ouster_data = flag(p, n + "enable_ouster_data", false);
  c.publish_only_feature_points = flag(p, n + "publish_only_feature_points", false);
  c.max_surface_features = (int)num(p, n + "max_surface_features", 2000);
  c.velocity_failure_threshold = num(p, n + "velocity_failure_threshold", 30.0);
  c.auto_voxel_size = flag(p, n + "auto_voxel_size", true);
  c.forget_far_chunks = flag(p, n + "forget_far_chunks", false);
  c.visual_confidence_factor = num(p, n + "visual_confidence_factor", 1.0);
  c.localization_mode = flag(p, n + "localization_mode", false);
  c.map_dir = str(p, "map_dir", "pointcloud_local.pcd");
  // (read_pose_file: the start pose then comes from a file under map_dir, :224-234 -- the caller sets init_* instead)
  c.init_x = (float)num(p, n + "init_x", 0.0); c.init_y = (float)num(p, n + "init_y", 0.0); c.init_z = (float)num(p, n + "init_z", 0.0);
  c.init_roll = (float)num(p, n + "init_roll", 0.0); c.init_pitch = (float)num(p, n + "init_pitch", 0.0); c.init_yaw = (float)num(p, n + "init_yaw", 0.0);
  c.use_imu_roll_pitch = flag(p, "use_imu_roll_pitch", false);       // config_.use_imu_roll_pitch = USE_IMU_ROLL_PITCH (:221; parameter.cpp:98)
  c.WORLD_FRAME = str(p, "world_frame", "sensor_init");               // parameter.cpp:289-293, 306-310
  c.SENSOR_FRAME = str(p, "sensor_frame", "sensor");
  c.ProjectName = str(p, "PROJECT_NAME", "");
  return c;
}

inline NodeConfig load_node_config(const std::string& path) {
  std::ifstream f(path);
  if (!f) throw std::runtime_error("cannot open parameter file " + path);
  std::stringstream ss;
  ss << f.rdbuf();
  return node_config_from_params(parse_ros_params_yaml(ss.str()));
}

}  // namespace super_odometry_soicp
