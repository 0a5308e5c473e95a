// local_map.h -- host side of the rolling voxel-block map (the reference's LocalMap,
// include/super_odometry/LidarProcess/LocalMap.h) and the builder of its HBM layout.
//
// Reference layout: 21x21x11 MapBlocks of 50 m (LocalMap.h:131-138), each a 32-byte-AoS
// pcl::PointXYZI cloud plus a pointer-linked octree.  MI355X layout built here:
//   * one 16-byte-per-point array {x,y,z,0} over ALL points of this rank's shard (one dwordx4 fetches a
//     candidate; a wave reads it through the scalar cache as a broadcast operand), in CANONICAL ORDER =
//     ascending (cube index, cell z, cell y, cell x), stable in the cube-local order;
//   * per occupied cube a dense table of nc^3+1 prefix offsets into that SoA (cells of 50/nc m,
//     nc = cells_per_cube(planeRes) chosen so that one cell >= sqrt(3*planeRes), the reference's
//     neighbour-distance gate, LidarSlam.cpp:526/741): the 27-cell neighbourhood of a query is
//     9 contiguous x-runs -> coalesced reads, no pointer chasing;
//   * cube_slot[4851]: table slot of a cube or -1 (= "no tree", LocalMap.h:506).
#pragma once
#include <array>
#include <cstddef>
#include <cstdint>
#include <memory>
#include <vector>

namespace soicp {

constexpr int kMapW = 21, kMapH = 21, kMapD = 11, kMapNum = kMapW * kMapH * kMapD;  // LocalMap.h:131-135
constexpr double kCube = 50.0, kHalfCube = 25.0;                                     // LocalMap.h:137-138

struct CanonicalMap {
  int nc = 1;             // cells per cube edge
  double cell = 50.0;     // cell edge (m)
  int n_slots = 0;
  std::vector<float> xyzw;           // 4 floats per point: x, y, z, 0
  size_t n_points() const { return xyzw.size() / 4; }
  std::vector<uint32_t> cell_start;  // n_slots * (nc^3 + 1), global offsets
  std::vector<int32_t> cube_slot;    // kMapNum
  std::vector<int32_t> slot_cube;    // n_slots -> cube index
  size_t total_points = 0;           // all ranks
};

int cells_per_cube(float plane_res, double* cell_size);
int shard_owner_of_cell(int wx, int wy, int wz, int cx, int cy, int cz, int world_size);

class LocalMap {
 public:
  LocalMap();
  void set_resolution(float line_res, float plane_res) { line_res_ = line_res; plane_res_ = plane_res; }
  float plane_res() const { return plane_res_; }
  float line_res() const { return line_res_; }
  const int* origin() const { return origin_; }

  void set_origin(const double t[3]);                    // LocalMap.h:146-164
  bool shift(const double t[3], int pos[3]);             // LocalMap.h:169-287; true if blocks moved
  int add_surf(const float* xyz, size_t n, size_t stride_floats);  // LocalMap.h:591-645
  int count_5x5(const int pos[3]) const;                 // LocalMap.h:292-318
  size_t size() const;
  void clear();
  int cube_index_of(const float p[3]) const;             // LocalMap.h:488-502 (-1: outside the window)
  size_t export_points(float* xyz, size_t cap, bool only_5x5, const int pos[3]) const;

  // Build the HBM layout for `rank` of `world` (world==1: everything).
  void build_canonical(int rank, int world, CanonicalMap& out) const;
  uint64_t version() const { return version_; }

  // PCL VoxelGrid restatement [UPSTREAM pcl 1.12.1 voxel_grid.hpp], exposed for tests
  static void voxel_grid(std::vector<float>& xyz /*AoS, in/out*/, float leaf);

 private:
  struct Cube { std::vector<float> xyz; };
  std::array<std::unique_ptr<Cube>, kMapNum> cubes_;
  int origin_[3];
  float line_res_ = 0.2f, plane_res_ = 0.4f;  // LocalMap.h:760-761
  uint64_t version_ = 1;
};

}  // namespace soicp
