// device_map.h -- the HBM-resident LocalMap (the whole map, or this rank's shard of it): the device pool is the authoritative store, the host
// keeps only the block bookkeeping of include/super_odometry/LidarProcess/LocalMap.h (origin_, which block holds data,
// per-block point counts).  Layout: every occupied 50 m cube owns a fixed region ("slot") of kCapPerSlot points in
// one pool of {x,y,z,0} records plus its nc^3+1 cell table; canonical index = slot * kCapPerSlot + position, so a map
// insert rewrites only the touched cubes and shiftMap is pure bookkeeping.
#pragma once
#include <hip/hip_runtime_api.h>

#include <cstdint>
#include <cstdlib>
#include <string>
#include <vector>

#include "kernels.h"
#include "local_map.h"
#include "map_kernels.h"

namespace soicp {

constexpr uint32_t kCapPerSlot = 1u << 20;  // points per cube region (16 MB); 4096 slots keep indices in 32 bits

class DeviceMap {
 public:
  // rank / world: this rank keeps the leaves that can reach a cell within one cell of a brick it owns (see map_kernels.hip:
  // shard_keeps_leaf); every rank inserts the SAME clouds, the per-cube point counts of the full map are summed over the
  // ranks by the caller (owned_counts / set_full_counts) after every insert
  explicit DeviceMap(hipStream_t s, int rank = 0, int world = 1) : stream_(s), rank_(rank), world_(world) {
    origin_[0] = 10; origin_[1] = 10; origin_[2] = 5; cube_slot_.assign(kMapNum, -1);
    const char* ev = std::getenv("SOICP_MAP_GROUPING");  // "sort": first stage of an insert by the stable radix sort (read per context)
    hash_grouping_ = !(ev && std::string(ev) == "sort");
    ev = std::getenv("SOICP_MAP_FAST");   // "0": every insert round by round, laid out by the host (two read-backs per insert);
    fast_enabled_ = !(ev && std::string(ev) == "0");   // "sync": laid out by the device, but Localization() waits for the insert's report
    defer_enabled_ = !(ev && std::string(ev) == "sync");
  }
  ~DeviceMap();
  // changing planeRes rebuilds the cell tables over the resident points (they are re-filtered when an insert next touches
  // their cube, like the reference); < 0: HIP error (text in err)
  int set_resolution(float line_res, float plane_res, std::string& err);
  float plane_res() const { return plane_res_; }
  const int* origin() const { return origin_; }
  void set_origin(const double t[3]);
  void shift(const double t[3], int pos[3]);
  int count_5x5(const int pos[3]) const;
  size_t size() const;        // points of the FULL map (all ranks; equals size_local() when world == 1 or before the counts were exchanged)
  size_t size_local() const;  // points resident on this rank
  bool sharded() const { return world_ > 1; }
  // sharded map: per cube index, the number of resident points whose own cell this rank owns (sum over ranks = full count)
  void owned_counts(std::vector<int32_t>& out) const;
  void set_full_counts(const std::vector<int32_t>& full);
  // sharded map, planeRes change: this rank's share of the full map / the re-cut from every rank's share (device_map.cpp)
  int export_owned(std::vector<uint8_t>& blob, std::string& err);
  int reshard(const std::vector<std::vector<uint8_t>>& blobs, float line_res, float plane_res, std::string& err);
  void clear();
  // LocalMap::addSurfPointCloud on the device.  d_xyz: device pointer, stride in floats.  Returns #points inside the window or <0.
  int add_surf_dev(const float* d_xyz, size_t n, size_t stride_floats, std::string& err);
  int add_surf_host(const float* xyz, size_t n, size_t stride_floats, std::string& err);
  // transformAndAddToMap (LidarSlam.cpp:60-80) on the device: world = T * scan (packed xyz, sensor frame) is written to d_world
  // and inserted.  defer: return as soon as the launches are in the queue and nothing reads d_scan any more; the map's
  // bookkeeping is brought up to date by the next call of any member (settle).  Returns the number of points inside the
  // window (0 when deferred) or < 0.
  int add_scan_dev(const float* d_scan, size_t n, const double T[7], float* d_world, bool defer, std::string& err);
  bool defer_enabled() const { return defer_enabled_; }
  bool insert_in_flight() const { return pending_.on; }  // a deferred insert has not been settled yet
  // completes a deferred insert: bookkeeping from the device's report, or -- when the device could not lay the round out --
  // the insert round by round.  Every member that reads or changes the bookkeeping calls it first.
  int settle(std::string& err);
  // (inserts laid out by the device / of those, the ones the host had to repeat round by round)
  void fast_stats(unsigned& inserts, unsigned& fallbacks) const { inserts = fast_inserts_; fallbacks = fast_fallbacks_; }
  size_t export_points(float* xyz, size_t cap, bool only_5x5, const int pos[3], std::string& err);
  // the same points as records of `stride` bytes (x, y, z floats first, the rest zero), every cube gathered into ONE staging buffer, ONE
  // copy to `out` (pinned memory: by DMA), ONE synchronisation; out == nullptr: the count only
  size_t export_records(void* out, size_t stride, size_t cap, bool only_5x5, const int pos[3], std::string& err);
  int view(DevMapView& v, std::string& err);  // refreshes the device cube_slot table when the bookkeeping changed; 0, -1 (a cube is full), -2 (device error)
  // leaf keys hold 9 or 10 bits per axis (50 / planeRes + 4 leaves per cube axis must fit): planeRes >= 0.05
  bool supported_resolution(float plane_res) const { return plane_res >= 0.0499f; }
  static uint32_t leaf_bits(float leaf) { return (50.0f / leaf + 6.0f < 512.0f) ? 9u : 10u; }
  static size_t max_touched(uint32_t lbits) { return lbits == 9u ? (size_t)kMaxTouched : (size_t)4; }

 private:
  int add_surf_legacy(const float* d_xyz, size_t n, size_t stride_floats, std::string& err);  // host-built rounds
  static constexpr int kNotFast = -100;
  int insert_fast(const float* d_in, size_t n, size_t stride_floats, const double* T, float* d_world, bool defer, std::string& err);
  int ensure_fast(std::string& err);
  int sync_meta(std::string& err);
  int upload_slot_table(std::string& err);
  void settle_quiet() const;
  int ensure_pool(int slots_needed, std::string& err);
  int ensure_work(size_t total, std::string& err);
  int ensure_grid(size_t gn, std::string& err, bool library_scan = true);
  int alloc_slot(int cube);
  hipStream_t stream_;
  int rank_ = 0, world_ = 1;
  std::vector<uint32_t> slot_owned_, slot_full_;  // sharded map: see owned_counts / set_full_counts
  int origin_[3];
  float line_res_ = 0.2f, plane_res_ = 0.4f, finest_res_ = 0.f;
  int nc_ = 1; double cell_ = 50.0; uint32_t ncell1_ = 2;
  std::vector<int32_t> cube_slot_;   // kMapNum: slot or -1
  std::vector<int32_t> slot_cube_;   // slot -> cube or -1 (free)
  std::vector<uint32_t> slot_count_;
  std::vector<float> slot_res_;      // planeRes the slot's cube was last filtered with (0: empty; negative: filtered with -value, but a
                                     // centroid drifted out of its leaf -- the pass-through grouping is off for the cube)
  bool slot_table_dirty_ = true;
  bool hash_grouping_ = true;
  // device
  float4* d_pool_ = nullptr; uint32_t* d_cell_start_ = nullptr; int32_t* d_cube_slot_ = nullptr; int slots_alloc_ = 0;
  // work buffers
  size_t work_cap_ = 0; size_t new_cap_ = 0;
  float4 *d_wpts_ = nullptr, *d_cent_ = nullptr, *d_spts_ = nullptr;
  uint32_t *d_k0_ = nullptr, *d_k1_ = nullptr, *d_v0_ = nullptr, *d_v1_ = nullptr, *d_flags_ = nullptr, *d_pos_ = nullptr, *d_heads_ = nullptr;
  uint32_t *d_grid_ = nullptr, *d_grid_scan_ = nullptr; size_t grid_cap_ = 0;  // dense cell grids of the touched cubes (second stage)
  uint32_t *d_ht_key_ = nullptr, *d_ht_cnt_ = nullptr, *d_ht_off_ = nullptr; uint32_t ht_log2_ = 0;  // leaf hash table of the first stage (map_kernels.hip)
  int ensure_leaf_table(size_t n_new, std::string& err);
  void* d_temp_ = nullptr; size_t temp_bytes_ = 0;
  int32_t* d_cube_of_ = nullptr; uint8_t* d_touched_ = nullptr; uint32_t* d_small_ = nullptr;
  float* d_stage_ = nullptr; size_t stage_cap_ = 0;  // host->device staging of new points / export
  uint8_t* h_touched_ = nullptr; uint32_t* h_small_ = nullptr;  // pinned; {counters[kSmallWords], touched flags[kMapNum]} in one block, like d_small_ / d_touched_
  static constexpr size_t kSmallWords = 128;
  size_t grid_zero_upto_ = 0;          // d_grid_[0 .. this) is all zero between inserts (a round cleans up after itself)
  bool block_clean_ = false;           // d_small_ / d_touched_ were cleared behind the previous insert
  // device-built inserts (insert_fast; map_kernels.hip: insert_front_kernel)
  static constexpr int kMaxSlots = 4096;
  bool fast_enabled_ = true, defer_enabled_ = true;
  MapTouched* d_tt_ = nullptr;                                   // the round as the kernels read it (host- or device-built)
  uint32_t *d_slot_count_ = nullptr, *d_slot_ok_ = nullptr;      // [kMaxSlots] each: the device's copy of slot_count_ / "slot_res_ == plane_res_"
  uint32_t* d_cube_cnt_ = nullptr; unsigned long long* d_scan_state_ = nullptr; uint32_t* d_tickets_ = nullptr;
  MapFastReport* h_report_ = nullptr;                            // pinned
  hipEvent_t ev_fast_ = nullptr;
  bool meta_dirty_ = true;    // the host changed slot counts / resolutions / slots behind the device's back: upload before the next device-built round
  bool fast_clean_ = false;   // d_cube_cnt_ / d_scan_state_ / d_tickets_ are all zero (the report kernel leaves them so)
  unsigned long long fast_seq_ = 0;
  struct Pending { bool on = false; const float* d_xyz = nullptr; size_t n = 0, stride = 0; unsigned long long seq = 0; } pending_;
  size_t est_old_ = 0;        // old points of the last device-built round (sizes the next one's launches)
  int skip_fast_ = 0;         // inserts to go round by round after the device met a scan that needs several rounds
  unsigned fast_inserts_ = 0, fast_fallbacks_ = 0;
  int deferred_rc_ = 0;
  std::string deferred_err_;  // error of a settle() inside a const member: reported (with its code) by the next call that can
};

}  // namespace soicp
