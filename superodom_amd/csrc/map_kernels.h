// map_kernels.h -- launch interface of the device-resident LocalMap update (map_kernels.hip).
#pragma once
#include <hip/hip_runtime_api.h>

#include <cstddef>
#include <cstdint>

#include "so_math.h"

namespace soicp {

constexpr int kMaxTouched = 32;  // cubes handled by one insert round (5 key bits above 3 x 9 leaf bits; 4 cubes with 10-bit leaves)

// the cubes one insert round touches (passed by value to the kernels)
struct MapTouched {
  int32_t n;
  int32_t cube[kMaxTouched];             // block index of touched cube t, ASCENDING; unused entries INT32_MAX (touched_index)
  uint32_t slot[kMaxTouched];            // pool slot of touched cube t
  uint32_t old_prefix[kMaxTouched + 1];  // exclusive prefix of the cubes' current point counts
  int32_t leaf_lo[kMaxTouched][3];       // floor(cube_min * inv_leaf) - 1: common leaf offset of the cube
  int32_t wcube[kMaxTouched][3];         // WORLD id of the cube (shard ownership hashes it: stable under shiftMap)
  double cube_min[kMaxTouched][3];       // world coordinates of the cube's min corner
  // Invariant watch: the hash grouping of an insert lets an old point pass through when no new point shares its leaf, which
  // is only right while every cube holds at most one point per leaf.  A centroid summed in float from tens of thousands of
  // points can drift out of its own leaf (pcl::VoxelGrid has the same arithmetic); bit t of *dirty is raised when a centroid
  // of touched cube t does not lie in the leaf it was built from -- the next insert that touches the cube then re-filters
  // all of it through the sort, as the reference does with every touched block.
  float inv_leaf_watch;
  uint32_t* dirty;
  uint32_t lbits;  // bits per leaf coordinate in the keys (9: up to kMaxTouched cubes per round, 10: up to 4; map_kernels.hip leaf_key)
  // device-built rounds (insert_front_kernel): first position of cube t's range in the second stage's scratch arrays -- the
  // cubes' cell grids are scanned one by one there, each from this base (exclusive prefix of old + new points per cube)
  uint32_t region_base[kMaxTouched];
};
// (the kernels read the struct from device memory: the host-built one is stored there by a one-wavefront launch, the
//  device-built one never visits the host; old_prefix[kMaxTouched] = number of old points of the round)

// Device-built insert round (DeviceMap::insert_fast): the touched cubes, their slots and point counts are worked out on
// the device from tables it keeps itself, so the host enqueues the whole insert without a read-back; what it has to know
// afterwards arrives in pinned memory.
enum : uint32_t { kFastHaltNone = 0, kFastHaltOverflow = 1 /* a leaf too large for the grouping kernels */, kFastHaltMultiRound = 2 /* more cubes than a round holds */,
                  kFastHaltUnallocated = 3 /* a cube without a slot */, kFastHaltNeedsSort = 4 /* a cube last filtered on another grid / marked by the drift watch */ };
struct MapFastReport {
  uint32_t halt;      // kFastHalt*: != 0 -> the map is unchanged, the host repeats the insert round by round
  uint32_t n;         // touched cubes
  uint32_t n_inside;  // new points inside the 21x21x11 window
  uint32_t dirty;     // MapTouched::dirty bits
  int32_t cube[kMaxTouched];
  uint32_t count[kMaxTouched];  // the cubes' new point counts
  uint32_t n_old, pad;
  unsigned long long front_seq;  // written by the front kernel's last workgroup: nobody reads the input points any more
  unsigned long long seq;        // written last (system scope): the insert is complete
};
constexpr uint32_t kFastTicketWords = kMaxTouched + 2 + 64;
constexpr uint32_t kScanItems = 2048;                     // cell counters per workgroup of cell_scan_table_kernel
constexpr uint32_t kScanBlocksMax = (64u * 64u * 64u + 1u + kScanItems - 1u) / kScanItems;  // 129
constexpr size_t kScanStateWords = (size_t)kMaxTouched * kScanBlocksMax;  // 64-bit look-back records
struct MapFastArgs {
  const float* d_in; uint32_t n, stride_floats;  // the new points: world frame, or (transform) the scan in the sensor frame
  bool transform; Pose pose; float* d_world;     // transform: world = pose * in, packed xyz, written to d_world (and inserted)
  int32_t origin[3];
  const int32_t* d_cube_slot;                    // [kMapNum], the table the k-NN uses
  uint32_t* d_slot_count; uint32_t* d_slot_ok;   // per slot: resident points / "one point per leaf of the current grid"
  uint32_t* d_cube_cnt;                          // [kMapNum] new points per cube (zero between inserts)
  unsigned long long* d_scan_state;              // [kScanStateWords] (zero between inserts)
  uint32_t* d_tickets;                           // [kFastTicketWords]: [0, kMaxTouched) the scan's, [kMaxTouched] the front kernel's, [kMaxTouched + 1] its
                                                 // count of touched cubes (all zero between inserts), then its list of them (64 entries)
  uint32_t* d_small;                             // DeviceMap's counter block (zero between inserts)
  MapFastReport* h_report; unsigned long long seq;
  int32_t per_round;
  uint32_t small_words;
};

struct MapInsertArgs {
  MapTouched tt;             // host-built round (stored to d_tt by the launch) -- unused when the device builds the round
  MapTouched* d_tt;          // the round as the kernels read it
  const float* d_xyz;        // new world-frame points (device)
  uint32_t n_new, stride_floats, n_old;  // n_old: the round's old points (device-built round: an upper bound -- the kernels take the number from d_tt)
  uint32_t n_old_grid;       // device-built round: the number of old points the launches are sized for (an estimate; the kernels stride); 0 = n_old
  const int32_t* d_cube_of;  // per new point: cube index or -1
  float inv_leaf;
  int32_t nc; uint32_t ncell1; double inv_cell;
  float4* pool; uint32_t cap; uint32_t* cell_start;
  float4* wpts; float4* cent; float4* spts /* working set in leaf-sorted order */; uint32_t* heads /* first index of leaf o; [n_leaves] = end */;
  uint32_t *keys0, *keys1, *vals0, *vals1, *flags, *pos;
  uint32_t* d_n_cent; uint32_t* d_counts;  // [8] (zeroed by the caller: [0] centroids, [1] long leaves, [2..3] member / group cursor, [4] giant leaves, [5] halt), [kMaxTouched]
  // first stage by hash grouping (default; nullptr = sort-based first stage): table of 2^ht_log2 >= max(4096, 2 n_new) slots,
  // keys 0xFFFFFFFF and counts 0 between inserts (the offsets kernel leaves it that way)
  uint32_t *ht_key, *ht_cnt, *ht_off; uint32_t ht_log2;
  // sharded map (world > 1): this rank keeps the LEAVES that can put a centroid into a cell within one cell of a brick it
  // owns (whole leaves, so that a kept centroid is the centroid of ALL the points of its leaf), and counts the points whose
  // own cell it owns (d_owned: summed over the ranks = the cube's full point count, LocalMap.h:292-318)
  int32_t rank, world;
  uint32_t* d_owned;                       // [kMaxTouched], zeroed by the caller; nullptr when world == 1
  uint32_t *grid, *grid_scan;              // [tt.n * ncell1 + 1] each: per-cube cell grids of the second stage (+ the total behind them)
  // Sort path only: a cube that was last filtered on ANOTHER leaf grid (old_inv_leaf[t] = 1 / that planeRes, 0 = keep the pool
  // order) enters the working set in the order the reference stores it -- ascending leaf index of that grid (the output order
  // of its last VoxelGrid, LocalMap.h:621-627) -- because several of its points now share a leaf and are summed in that order.
  float old_inv_leaf[kMaxTouched];
  bool reorder_old;
  bool grid_is_clean;                      // grid[0 .. tt.n * ncell1] is all zero (left so by the previous round): no fill needed
  void* temp; size_t temp_bytes;
};

// scan pre-filter (adjustVoxelSize): statistics + VoxelGrid of one cloud
// The decisions of adjustVoxelSize + pcl::VoxelGrid's grid set-up, taken on the device (vg_decide_kernel) so that the
// whole pre-filter is ONE enqueue and one read-back: reduced statistics, the resolution the reference would choose
// (laserMapping.cpp:604-636), min_b / div_b of the leaf grid (voxel_grid.hpp applyFilter).
enum : uint32_t { kVgNeedInputOrder = 1u /* the statistic is within the rounding band of a threshold: the reference's own float
                                            accumulation has to decide (host path) */,
                  kVgLeafTooSmall = 2u /* dx dy dz overflows int32: PCL passes the cloud through (host path) */ };
struct VgDecision {
  double acc[10];  // sum |x|, |y|, |z|, points beyond 3 m, min x y z, max x y z
  float average_distance, leaf, inv_leaf, line_res, plane_res;
  int32_t choice;  // 0: statistic < 25 (0.1 / 0.2), 1: the caller's resolutions, 2: > 65 (0.4 / 0.8)
  int32_t min_b[3], div_b[3];
  uint32_t flags, n_leaves;
};
struct VgCandidates { float line_res[3], plane_res[3], inv_leaf[3]; };  // per choice; the reciprocals formed on the host
// partial sums -> VgDecision; zeroes counters[0..16)
// (also zeroes the look-back records of the filter's fused scan: d_scan_state[0..n_state))
void launch_vg_decide(const double* d_part, int blocks, uint32_t n, int auto_voxel_size, const VgCandidates& cand, VgDecision* d_out,
                      uint32_t* d_counters, unsigned long long* d_scan_state, uint32_t n_state, hipStream_t s);
struct VoxelFilterArgs {
  const float* d_xyz; uint32_t n, stride_floats;
  float inv_leaf; int min_b[3], div_b[3];
  const VgDecision* d_decision;  // non-null: inv_leaf / min_b / div_b are read from here on the device; d_n_cent[0] is copied to its n_leaves
  unsigned long long* scan_state; uint32_t n_scan_state;  // non-null (all zero, d_n_cent[8] too): flags + scan + heads in one launch
  float4 *wpts, *spts;
  uint32_t *keys0, *keys1, *vals0, *vals1, *flags, *pos, *heads;
  uint32_t* d_n_cent;  // [0] number of leaves, [1] long-leaf counter (both zeroed by the caller)
  float* d_out;        // packed xyz centroids in ascending leaf index
  void* temp; size_t temp_bytes;
};
void launch_vg_stats(const float* d_xyz, uint32_t n, uint32_t stride_floats, double* d_part /* blocks x 10 */, int blocks, hipStream_t s);
// sum |x|, sum |y|, sum |z| as the reference accumulates them: float, input order (one wavefront; see the kernel)
void launch_vg_stats_inorder(const float* d_xyz, uint32_t n, uint32_t stride_floats, float* d_out3, hipStream_t s);
void launch_voxel_filter(const VoxelFilterArgs& a, hipStream_t s);

size_t map_sort_temp_bytes(size_t n);
void launch_world_cube(const float* d_xyz, uint32_t n, uint32_t stride_floats, const int origin[3], int32_t* d_cube_of,
                       uint8_t* d_touched, uint32_t* d_n_inside, hipStream_t s);
void launch_transform_scan(const float* d_scan, uint32_t n, const Pose& pose, float* d_out, hipStream_t s);
void launch_map_insert(const MapInsertArgs& a, hipStream_t s);
// the whole insert without a host round trip: front kernel (world transform, cube of every point, the round built on the
// device), the hashed first stage, the second stage with its own scan, the report into pinned memory
void launch_map_insert_fast(const MapInsertArgs& a, const MapFastArgs& f, hipStream_t s);
void launch_map_retable(const MapInsertArgs& a, hipStream_t s);  // resolution change: new cell tables over the resident points
// re-cut of a shard: keep the candidates whose new-grid leaf this rank keeps; counters[0] = kept, [1] = of all candidates, those whose own cell it owns
void launch_shard_select(const float* d_xyz, uint32_t n, const MapTouched& tt, float inv_leaf, int nc, double inv_cell, int rank, int world,
                         float4* pool_slot, uint32_t cap, uint32_t* d_counters, hipStream_t s);
void launch_gather_export(const float4* pool, uint32_t cap, uint32_t slot, uint32_t count, float* d_out, hipStream_t s);
void launch_gather_export_records(const float4* pool, uint32_t cap, uint32_t slot, uint32_t count, uint32_t* d_out_words, uint32_t stride_words, hipStream_t s);
// pointAssociateToMap over a cloud (registered scan of the node): records rewritten in place, keep flags, number kept
void launch_transform_cloud(uint8_t* d_pts, uint32_t n, uint32_t stride, const Pose& pose, uint8_t* d_keep, uint32_t* d_n_kept /* zeroed */, hipStream_t s);
// removePointDistortion: records of `stride` bytes (float x y z at 0 4 8, float time at time_off), rewritten in place
struct DeskewFrames;
void launch_deskew(uint8_t* d_pts, uint32_t n, uint32_t stride, uint32_t time_off, double t0, const double* d_poses /* n_poses x 8 */,
                   uint32_t n_poses, const DeskewFrames& f, uint32_t* d_n_clamped /* zeroed by the caller */, hipStream_t s);

}  // namespace soicp
