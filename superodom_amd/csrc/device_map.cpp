// device_map.cpp -- see device_map.h.  Host C++ driving map_kernels.hip; bookkeeping mirrors
// include/super_odometry/LidarProcess/LocalMap.h (paths relative to /root/reference/super_odometry/).
#include "device_map.h"

#include <algorithm>
#include <atomic>
#include <climits>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <map>

namespace soicp {

// (member functions only.  An error exit may come between a round's counting kernel and the kernel that zeroes the counters
//  again, or behind a clear that was only enqueued: the "already clean" marks are withdrawn, the next round fills everything)
#define DM_TRY(expr)                                                      \
  do {                                                                    \
    hipError_t e__ = (expr);                                              \
    if (e__ != hipSuccess) {                                              \
      err = std::string(#expr) + ": " + hipGetErrorString(e__);           \
      grid_zero_upto_ = 0; block_clean_ = false;                          \
      fast_clean_ = false; meta_dirty_ = true;                            \
      return -2;                                                          \
    }                                                                     \
  } while (0)

static inline int cidx(int i, int j, int k) { return i + kMapW * j + kMapW * kMapH * k; }

DeviceMap::~DeviceMap() {
  (void)hipStreamSynchronize(stream_);  // (a deferred insert may still be writing its report)
  for (void* p : {(void*)d_tt_, (void*)d_slot_count_, (void*)d_cube_cnt_, (void*)d_scan_state_})
    if (p) (void)hipFree(p);
  if (h_report_) (void)hipHostFree(h_report_);
  if (ev_fast_) (void)hipEventDestroy(ev_fast_);
  for (void* p : {(void*)d_pool_, (void*)d_cell_start_, (void*)d_cube_slot_, (void*)d_wpts_, (void*)d_cent_, (void*)d_k0_, (void*)d_k1_,
                  (void*)d_v0_, (void*)d_v1_, (void*)d_flags_, (void*)d_pos_, (void*)d_spts_, (void*)d_heads_, (void*)d_grid_, (void*)d_grid_scan_, d_temp_, (void*)d_cube_of_,
                  (void*)d_small_, (void*)d_stage_, (void*)d_ht_key_, (void*)d_ht_cnt_, (void*)d_ht_off_})
    if (p) (void)hipFree(p);
  if (h_small_) (void)hipHostFree(h_small_);  // (the touched flags live in the same blocks)
}

void DeviceMap::clear() {
  settle_quiet();
  meta_dirty_ = true;
  std::fill(cube_slot_.begin(), cube_slot_.end(), -1);
  std::fill(slot_cube_.begin(), slot_cube_.end(), -1);
  std::fill(slot_count_.begin(), slot_count_.end(), 0u);
  std::fill(slot_owned_.begin(), slot_owned_.end(), 0u);
  std::fill(slot_full_.begin(), slot_full_.end(), 0u);
  slot_table_dirty_ = true;
}

void DeviceMap::set_origin(const double t[3]) {  // LocalMap.h:146-164
  settle_quiet();
  for (int a = 0; a < 3; ++a) origin_[a] = -cube_coord(t[a], 0);
  slot_table_dirty_ = true;
}

int DeviceMap::alloc_slot(int cube) {
  meta_dirty_ = true;
  for (size_t s = 0; s < slot_cube_.size(); ++s)
    if (slot_cube_[s] < 0) { slot_cube_[s] = cube; slot_count_[s] = 0; slot_owned_[s] = 0; slot_full_[s] = 0; slot_res_[s] = 0.f; cube_slot_[cube] = (int)s; slot_table_dirty_ = true; return (int)s; }
  slot_cube_.push_back(cube); slot_count_.push_back(0); slot_owned_.push_back(0); slot_full_.push_back(0); slot_res_.push_back(0.f);
  cube_slot_[cube] = (int)slot_cube_.size() - 1;
  slot_table_dirty_ = true;
  return cube_slot_[cube];
}

// LocalMap::shiftMap, LocalMap.h:169-287: the block array rolls so that the sensor's block stays >= 3 blocks from the
// border; blocks leaving the window are dropped.  Here a block is just its slot id -- no point data moves.
void DeviceMap::shift(const double t[3], int pos[3]) {
  settle_quiet();
  int c[3] = {cube_coord(t[0], origin_[0]), cube_coord(t[1], origin_[1]), cube_coord(t[2], origin_[2])};
  const int dim[3] = {kMapW, kMapH, kMapD};
  auto at = [&](int i, int j, int k) -> int32_t& { return cube_slot_[cidx(i, j, k)]; };
  auto drop = [&](int32_t& s) { if (s >= 0) { meta_dirty_ = true; slot_cube_[s] = -1; slot_count_[s] = 0; slot_owned_[s] = 0; slot_full_[s] = 0; slot_res_[s] = 0.f; } s = -1; };
  bool moved = false;
  for (int axis = 0; axis < 3; ++axis) {
    while (c[axis] < 3 || c[axis] >= dim[axis] - 3) {
      const int dir = c[axis] < 3 ? +1 : -1;  // +1: contents move towards higher indices
      for (int u = 0; u < dim[(axis + 1) % 3]; ++u) for (int v = 0; v < dim[(axis + 2) % 3]; ++v) {
        auto cell = [&](int w) -> int32_t& {
          int ijk[3]; ijk[axis] = w; ijk[(axis + 1) % 3] = u; ijk[(axis + 2) % 3] = v;
          return at(ijk[0], ijk[1], ijk[2]);
        };
        if (dir > 0) { drop(cell(dim[axis] - 1)); for (int w = dim[axis] - 1; w >= 1; --w) cell(w) = cell(w - 1); cell(0) = -1; }
        else { drop(cell(0)); for (int w = 0; w < dim[axis] - 1; ++w) cell(w) = cell(w + 1); cell(dim[axis] - 1) = -1; }
      }
      c[axis] += dir; origin_[axis] += dir; moved = true;
    }
  }
  if (moved) {
    for (int cube = 0; cube < kMapNum; ++cube) if (cube_slot_[cube] >= 0) slot_cube_[cube_slot_[cube]] = cube;
    slot_table_dirty_ = true;
  }
  pos[0] = c[0]; pos[1] = c[1]; pos[2] = c[2];
}

int DeviceMap::count_5x5(const int pos[3]) const {  // LocalMap.h:292-318
  settle_quiet();
  int n = 0;
  for (int i = pos[0] - 2; i <= pos[0] + 2; ++i) for (int j = pos[1] - 2; j <= pos[1] + 2; ++j) for (int k = pos[2] - 1; k <= pos[2] + 1; ++k)
    if (i >= 0 && i < kMapW && j >= 0 && j < kMapH && k >= 0 && k < kMapD && cube_slot_[cidx(i, j, k)] >= 0)
      n += (int)(world_ > 1 ? slot_full_[cube_slot_[cidx(i, j, k)]] : slot_count_[cube_slot_[cidx(i, j, k)]]);
  return n;
}

size_t DeviceMap::size_local() const {
  settle_quiet();
  size_t n = 0;
  for (size_t s = 0; s < slot_cube_.size(); ++s) if (slot_cube_[s] >= 0) n += slot_count_[s];
  return n;
}
size_t DeviceMap::size() const {
  settle_quiet();
  if (world_ <= 1) return size_local();
  size_t n = 0;
  for (size_t s = 0; s < slot_cube_.size(); ++s) if (slot_cube_[s] >= 0) n += slot_full_[s];
  return n;
}
void DeviceMap::owned_counts(std::vector<int32_t>& out) const {
  settle_quiet();
  out.assign(kMapNum, 0);
  for (size_t s = 0; s < slot_cube_.size(); ++s) if (slot_cube_[s] >= 0) out[(size_t)slot_cube_[s]] = (int32_t)slot_owned_[s];
}
void DeviceMap::set_full_counts(const std::vector<int32_t>& full) {
  settle_quiet();
  for (size_t s = 0; s < slot_cube_.size(); ++s) if (slot_cube_[s] >= 0) slot_full_[s] = (uint32_t)std::max(0, full[(size_t)slot_cube_[s]]);
}

int DeviceMap::ensure_pool(int slots_needed, std::string& err) {
  if (slots_needed <= slots_alloc_) return 0;
  if (slots_needed > 4096) { err = "DeviceMap: more than 4096 occupied cubes"; return -1; }
  int want = std::max(16, slots_alloc_ * 2);
  while (want < slots_needed) want *= 2;
  float4* np = nullptr; uint32_t* nt = nullptr;
  DM_TRY(hipMalloc(reinterpret_cast<void**>(&np), (size_t)want * kCapPerSlot * sizeof(float4) + 64));
  DM_TRY(hipMalloc(reinterpret_cast<void**>(&nt), (size_t)want * ncell1_ * sizeof(uint32_t)));
  if (slots_alloc_) {
    DM_TRY(hipMemcpyAsync(np, d_pool_, (size_t)slots_alloc_ * kCapPerSlot * sizeof(float4), hipMemcpyDeviceToDevice, stream_));
    DM_TRY(hipMemcpyAsync(nt, d_cell_start_, (size_t)slots_alloc_ * ncell1_ * sizeof(uint32_t), hipMemcpyDeviceToDevice, stream_));
    DM_TRY(hipStreamSynchronize(stream_));
    (void)hipFree(d_pool_); (void)hipFree(d_cell_start_);
  }
  d_pool_ = np; d_cell_start_ = nt; slots_alloc_ = want;
  if (!d_cube_slot_) DM_TRY(hipMalloc(reinterpret_cast<void**>(&d_cube_slot_), kMapNum * sizeof(int32_t)));
  slot_table_dirty_ = true;
  return 0;
}

// hash table of the insert's first stage: one slot per distinct leaf of the NEW points, load factor <= 1/2, at least one
// workgroup's worth of slots for the offsets kernel; empty (keys 0xFFFFFFFF, counts 0) between inserts
int DeviceMap::ensure_leaf_table(size_t n_new, std::string& err) {
  uint32_t lg = 12;
  while (((size_t)1 << lg) < 2 * (n_new + 1)) ++lg;
  if (lg > 30) { err = "DeviceMap: too many new points for the leaf table"; return -1; }
  if (lg <= ht_log2_ && d_ht_key_) return 0;
  for (void* p : {(void*)d_ht_key_, (void*)d_ht_cnt_, (void*)d_ht_off_}) if (p) (void)hipFree(p);
  d_ht_key_ = d_ht_cnt_ = d_ht_off_ = nullptr; ht_log2_ = 0;
  const size_t slots = (size_t)1 << lg;
  DM_TRY(hipMalloc(reinterpret_cast<void**>(&d_ht_key_), slots * sizeof(uint32_t)));
  DM_TRY(hipMalloc(reinterpret_cast<void**>(&d_ht_cnt_), slots * sizeof(uint32_t)));
  DM_TRY(hipMalloc(reinterpret_cast<void**>(&d_ht_off_), slots * sizeof(uint32_t)));
  DM_TRY(hipMemsetAsync(d_ht_key_, 0xFF, slots * sizeof(uint32_t), stream_));
  DM_TRY(hipMemsetAsync(d_ht_cnt_, 0, slots * sizeof(uint32_t), stream_));
  ht_log2_ = lg;
  return 0;
}

int DeviceMap::ensure_work(size_t total, std::string& err) {
  // the counters and the touched-cube flags share one block on either side: one fill, one copy back per insert
  if (!h_small_) {
    DM_TRY(hipHostMalloc(reinterpret_cast<void**>(&h_small_), kSmallWords * sizeof(uint32_t) + kMapNum));
    h_touched_ = reinterpret_cast<uint8_t*>(h_small_ + kSmallWords);
  }
  if (!d_small_) {
    DM_TRY(hipMalloc(reinterpret_cast<void**>(&d_small_), kSmallWords * sizeof(uint32_t) + kMapNum));
    d_touched_ = reinterpret_cast<uint8_t*>(d_small_ + kSmallWords);
  }
  if (!d_tt_) DM_TRY(hipMalloc(reinterpret_cast<void**>(&d_tt_), sizeof(MapTouched)));
  if (total <= work_cap_) return 0;
  const size_t cap = total + total / 4 + 1024;
  for (void* p : {(void*)d_wpts_, (void*)d_cent_, (void*)d_spts_, (void*)d_heads_, (void*)d_k0_, (void*)d_k1_, (void*)d_v0_, (void*)d_v1_, (void*)d_flags_, (void*)d_pos_, d_temp_})
    if (p) (void)hipFree(p);
  d_wpts_ = d_cent_ = d_spts_ = nullptr; d_heads_ = nullptr; d_k0_ = d_k1_ = d_v0_ = d_v1_ = d_flags_ = d_pos_ = nullptr; d_temp_ = nullptr; work_cap_ = 0;
  DM_TRY(hipMalloc(reinterpret_cast<void**>(&d_wpts_), cap * sizeof(float4)));
  DM_TRY(hipMalloc(reinterpret_cast<void**>(&d_cent_), cap * sizeof(float4)));
  DM_TRY(hipMalloc(reinterpret_cast<void**>(&d_spts_), cap * sizeof(float4)));
  DM_TRY(hipMalloc(reinterpret_cast<void**>(&d_heads_), (cap + 1) * sizeof(uint32_t)));
  for (uint32_t** p : {&d_k0_, &d_k1_, &d_v0_, &d_v1_, &d_flags_, &d_pos_}) DM_TRY(hipMalloc(reinterpret_cast<void**>(p), cap * sizeof(uint32_t)));
  temp_bytes_ = map_sort_temp_bytes(cap) + 256;
  DM_TRY(hipMalloc(&d_temp_, temp_bytes_));
  work_cap_ = cap;
  return 0;
}

int DeviceMap::upload_slot_table(std::string& err) {
  if (!slot_table_dirty_) return 0;
  if (hipMemcpyAsync(d_cube_slot_, cube_slot_.data(), kMapNum * sizeof(int32_t), hipMemcpyHostToDevice, stream_) != hipSuccess ||
      hipStreamSynchronize(stream_) != hipSuccess) { err = "DeviceMap: cube_slot upload failed"; return -2; }
  slot_table_dirty_ = false;
  return 0;
}

int DeviceMap::view(DevMapView& v, std::string& err) {
  if (const int rs = settle(err); rs < 0) return rs;  // (-1: a cube is full -- the caller's SO_ICP_E_NOMEM --, -2: device error)
  if (ensure_pool(std::max<int>(1, (int)slot_cube_.size()), err)) return -2;
  if (upload_slot_table(err)) return -2;
  v.pts = d_pool_; v.cell_start = d_cell_start_; v.cube_slot = d_cube_slot_;
  v.nc = nc_; v.ncell1 = ncell1_; v.inv_cell = 1.0 / cell_;
  v.origin[0] = origin_[0]; v.origin[1] = origin_[1]; v.origin[2] = origin_[2];
  v.n_points = (uint32_t)size_local();
  v.n_slots = (uint32_t)std::max<size_t>(1, slot_cube_.size());
  return 0;
}

// planeRes decides the leaf of the voxel filter and the cell of the index.  The resident points keep their positions:
// the reference re-filters a block only when the next insert touches it (LocalMap.h:617-641), untouched blocks keep
// their points.  Only the cell tables are rebuilt for the new cell size (launch_map_retable), on the device.
int DeviceMap::set_resolution(float line_res, float plane_res, std::string& err) {
  line_res_ = line_res;
  if (plane_res == plane_res_ && nc_ > 1) return 0;  // (pushed every frame, lmap.cpp:648-649: nothing to wait for)
  if (const int rs = settle(err); rs < 0) return rs;
  meta_dirty_ = true;  // ("filtered on the current grid" changes its meaning with planeRes)
  const float old_res = plane_res_;
  const bool had = size_local() > 0 && nc_ > 1;
  plane_res_ = plane_res;
  finest_res_ = (had && finest_res_ > 0.f) ? std::min(finest_res_, std::min(old_res, plane_res)) : plane_res;
  double cell;
  nc_ = cells_per_cube(plane_res, &cell);
  cell_ = cell;
  const uint32_t new_ncell1 = (uint32_t)((size_t)nc_ * nc_ * nc_ + 1);
  if (new_ncell1 != ncell1_ || !d_cell_start_) {  // table geometry changed: new tables, the point pool stays
    if (d_cell_start_) (void)hipFree(d_cell_start_);
    d_cell_start_ = nullptr;
    ncell1_ = new_ncell1;
    if (slots_alloc_) {
      DM_TRY(hipMalloc(reinterpret_cast<void**>(&d_cell_start_), (size_t)slots_alloc_ * ncell1_ * sizeof(uint32_t)));
      // (a slot that is allocated but holds no point of this rank -- an empty shard after a re-cut -- must read "no points",
      //  not whatever the allocation held)
      DM_TRY(hipMemsetAsync(d_cell_start_, 0, (size_t)slots_alloc_ * ncell1_ * sizeof(uint32_t), stream_));
    }
  }
  if (!had) return 0;
  std::vector<int> occupied;
  for (size_t s = 0; s < slot_cube_.size(); ++s) if (slot_cube_[s] >= 0) occupied.push_back((int)s);
  const float inv_leaf = 1.0f / finest_res_;  // the points' leaf keys are distinct on the finest grid they were filtered on
  const uint32_t lbits = leaf_bits(finest_res_);
  const size_t per_round = max_touched(lbits);
  for (size_t r0 = 0; r0 < occupied.size(); r0 += per_round) {
    MapInsertArgs a{};
    MapTouched& tt = a.tt;
    tt.lbits = lbits;
    tt.n = (int)std::min<size_t>(per_round, occupied.size() - r0);
    uint32_t n_old = 0;
    for (int t = 0; t < tt.n; ++t) {
      const int s = occupied[r0 + t], cube = slot_cube_[s];
      tt.slot[t] = (uint32_t)s;
      tt.old_prefix[t] = n_old;
      n_old += slot_count_[s];
      const int ci = cube % kMapW, cj = (cube / kMapW) % kMapH, ck = cube / (kMapW * kMapH);
      const int w[3] = {ci - origin_[0], cj - origin_[1], ck - origin_[2]};
      for (int ax = 0; ax < 3; ++ax) {
        tt.cube_min[t][ax] = w[ax] * kCube - kHalfCube;
        tt.leaf_lo[t][ax] = (int)std::floor((float)tt.cube_min[t][ax] * inv_leaf) - 2;
      }
    }
    for (int t = tt.n; t <= kMaxTouched; ++t) tt.old_prefix[t] = n_old;
    for (int t = tt.n; t < kMaxTouched; ++t) tt.slot[t] = 0;
    if (ensure_work(n_old, err)) return -2;
    if (ensure_grid((size_t)tt.n * ncell1_ + 1024, err)) return -2;
    block_clean_ = false;
    DM_TRY(hipMemsetAsync(d_small_, 0, 48 * sizeof(uint32_t), stream_));
    a.n_old = n_old; a.inv_leaf = inv_leaf;
    a.nc = nc_; a.ncell1 = ncell1_; a.inv_cell = 1.0 / cell_;
    a.pool = d_pool_; a.cap = kCapPerSlot; a.cell_start = d_cell_start_;
    a.wpts = d_wpts_; a.cent = d_cent_; a.spts = d_spts_; a.heads = d_heads_;
    a.keys0 = d_k0_; a.keys1 = d_k1_; a.vals0 = d_v0_; a.vals1 = d_v1_; a.flags = d_flags_; a.pos = d_pos_;
    a.d_n_cent = d_small_; a.d_counts = d_small_ + 8;
    a.grid = d_grid_; a.grid_scan = d_grid_scan_;
    a.temp = d_temp_; a.temp_bytes = temp_bytes_;
    a.d_tt = d_tt_;
    {
      const size_t need = (size_t)tt.n * ncell1_ + 1;
      a.grid_is_clean = need <= grid_zero_upto_;
      grid_zero_upto_ = std::max(grid_zero_upto_, need);
    }
    launch_map_retable(a, stream_);
    DM_TRY(hipGetLastError());
    DM_TRY(hipStreamSynchronize(stream_));  // `a` travels by value, but the next round reuses the work buffers
  }
  return 0;
}

int DeviceMap::ensure_grid(size_t gn, std::string& err, bool library_scan) {
  if (gn > grid_cap_) {
    if (d_grid_) (void)hipFree(d_grid_);
    if (d_grid_scan_) (void)hipFree(d_grid_scan_);
    d_grid_ = d_grid_scan_ = nullptr; grid_cap_ = 0; grid_zero_upto_ = 0;
    DM_TRY(hipMalloc(reinterpret_cast<void**>(&d_grid_), gn * sizeof(uint32_t)));
    DM_TRY(hipMalloc(reinterpret_cast<void**>(&d_grid_scan_), gn * sizeof(uint32_t)));
    grid_cap_ = gn;
  }
  // (a device-built round scans its grids with its own kernel: no scratch; sized for a whole round it would be ~100 MB)
  if (library_scan && temp_bytes_ < map_sort_temp_bytes(gn)) {  // the scan of the grids uses the sort's scratch buffer
    if (d_temp_) (void)hipFree(d_temp_);
    d_temp_ = nullptr;
    temp_bytes_ = map_sort_temp_bytes(gn) + 256;
    DM_TRY(hipMalloc(&d_temp_, temp_bytes_));
  }
  return 0;
}

int DeviceMap::add_surf_host(const float* xyz, size_t n, size_t stride_floats, std::string& err) {
  if (const int rs = settle(err); rs < 0) return rs;  // (a deferred insert may still read the staging buffer)
  if (!n) return 0;
  if (stride_floats == 0) stride_floats = 3;
  if (n * stride_floats > stage_cap_) {
    if (d_stage_) (void)hipFree(d_stage_);
    d_stage_ = nullptr; stage_cap_ = 0;
    DM_TRY(hipMalloc(reinterpret_cast<void**>(&d_stage_), (n * stride_floats + 1024) * sizeof(float)));
    stage_cap_ = n * stride_floats + 1024;
  }
  DM_TRY(hipMemcpyAsync(d_stage_, xyz, n * stride_floats * sizeof(float), hipMemcpyHostToDevice, stream_));
  return add_surf_dev(d_stage_, n, stride_floats, err);
}

int DeviceMap::add_surf_dev(const float* d_xyz, size_t n, size_t stride_floats, std::string& err) {
  if (const int rs = settle(err); rs < 0) return rs;
  if (!n) return 0;
  if (stride_floats == 0) stride_floats = 3;
  const int r = insert_fast(d_xyz, n, stride_floats, nullptr, nullptr, false, err);
  return r != kNotFast ? r : add_surf_legacy(d_xyz, n, stride_floats, err);
}

int DeviceMap::add_scan_dev(const float* d_scan, size_t n, const double T[7], float* d_world, bool defer, std::string& err) {
  if (const int rs = settle(err); rs < 0) return rs;
  if (!n) return 0;
  const int r = insert_fast(d_scan, n, 3, T, d_world, defer, err);
  if (r != kNotFast) return r;
  launch_transform_scan(d_scan, (uint32_t)n, pose_from_array(T), d_world, stream_);
  return add_surf_legacy(d_world, n, 3, err);
}

// ---- the insert laid out by the device -----------------------------------------------------------------------------
int DeviceMap::ensure_fast(std::string& err) {
  if (d_slot_count_) return 0;
  DM_TRY(hipMalloc(reinterpret_cast<void**>(&d_slot_count_), 2 * kMaxSlots * sizeof(uint32_t)));
  d_slot_ok_ = d_slot_count_ + kMaxSlots;
  DM_TRY(hipMalloc(reinterpret_cast<void**>(&d_cube_cnt_), (kMapNum + kFastTicketWords) * sizeof(uint32_t)));
  d_tickets_ = d_cube_cnt_ + kMapNum;
  DM_TRY(hipMalloc(reinterpret_cast<void**>(&d_scan_state_), kScanStateWords * sizeof(unsigned long long)));
  DM_TRY(hipHostMalloc(reinterpret_cast<void**>(&h_report_), sizeof(MapFastReport)));
  std::memset(h_report_, 0, sizeof(MapFastReport));
  DM_TRY(hipEventCreateWithFlags(&ev_fast_, hipEventDisableTiming));
  meta_dirty_ = true; fast_clean_ = false;
  return 0;
}

// the device's copies of the per-slot bookkeeping, after the HOST changed it (a slot allocated or dropped, a round laid out
// by the host, planeRes changed, the map cleared): rare, so the upload simply waits
int DeviceMap::sync_meta(std::string& err) {
  if (!meta_dirty_) return 0;
  std::vector<uint32_t> m(2 * (size_t)kMaxSlots, 0u);
  for (size_t s = 0; s < slot_cube_.size() && s < (size_t)kMaxSlots; ++s) {
    m[s] = slot_cube_[s] >= 0 ? slot_count_[s] : 0u;
    m[kMaxSlots + s] = (m[s] == 0u || slot_res_[s] == plane_res_) ? 1u : 0u;
  }
  DM_TRY(hipMemcpyAsync(d_slot_count_, m.data(), m.size() * sizeof(uint32_t), hipMemcpyHostToDevice, stream_));
  DM_TRY(hipStreamSynchronize(stream_));
  meta_dirty_ = false;
  return 0;
}

int DeviceMap::insert_fast(const float* d_in, size_t n, size_t stride_floats, const double* T, float* d_world, bool defer, std::string& err) {
  if (!fast_enabled_ || !hash_grouping_ || world_ > 1 || slot_cube_.empty() || n >= (1u << 30)) return kNotFast;
  if (slot_cube_.size() > (size_t)kMaxSlots) return kNotFast;  // (the device's per-slot tables hold kMaxSlots entries: the host lays such rounds out)
  if (skip_fast_ > 0) { --skip_fast_; return kNotFast; }
  if (nc_ <= 1 && ncell1_ <= 2) return kNotFast;  // (no resolution yet: the map is empty)
  const uint32_t lbits = leaf_bits(plane_res_);
  const size_t per_round = max_touched(lbits);
  if (ensure_pool(std::max<int>(1, (int)slot_cube_.size()), err)) return -2;
  if (upload_slot_table(err)) return -2;
  const size_t n_old_ub = size_local();
  if (n_old_ub + n >= (1u << 31)) return kNotFast;
  if (ensure_work(n_old_ub + n, err)) return -2;
  if (ensure_fast(err)) return -2;
  if (sync_meta(err)) return -2;
  if (n > new_cap_) {
    if (d_cube_of_) (void)hipFree(d_cube_of_);
    d_cube_of_ = nullptr; new_cap_ = 0;
    DM_TRY(hipMalloc(reinterpret_cast<void**>(&d_cube_of_), (n + 1024) * sizeof(int32_t)));
    new_cap_ = n + 1024;
  }
  if (ensure_grid(per_round * ncell1_ + 1024, err, false)) return -2;
  if (ensure_leaf_table(n, err)) return -2;
  if (!block_clean_) { DM_TRY(hipMemsetAsync(d_small_, 0, kSmallWords * sizeof(uint32_t) + kMapNum, stream_)); block_clean_ = true; }
  if (!fast_clean_) {
    DM_TRY(hipMemsetAsync(d_cube_cnt_, 0, (kMapNum + kFastTicketWords) * sizeof(uint32_t), stream_));
    DM_TRY(hipMemsetAsync(d_scan_state_, 0, kScanStateWords * sizeof(unsigned long long), stream_));
    fast_clean_ = true;
  }
  {
    const size_t need = per_round * ncell1_ + 1;
    if (need > grid_zero_upto_) { DM_TRY(hipMemsetAsync(d_grid_, 0, need * sizeof(uint32_t), stream_)); grid_zero_upto_ = need; }
  }
  MapInsertArgs a{};
  a.tt.lbits = lbits; a.d_tt = d_tt_;
  a.d_xyz = T ? d_world : d_in; a.n_new = (uint32_t)n; a.stride_floats = T ? 3u : (uint32_t)stride_floats; a.n_old = (uint32_t)n_old_ub;
  // the launches are sized for what the last device-built round held (+ 25 %), not for the whole map: the kernels stride
  a.n_old_grid = (uint32_t)std::max<size_t>(1, std::min<size_t>(n_old_ub, est_old_ ? est_old_ + est_old_ / 4 + 16384 : n_old_ub));
  a.d_cube_of = d_cube_of_; a.inv_leaf = 1.0f / plane_res_;
  a.nc = nc_; a.ncell1 = ncell1_; a.inv_cell = 1.0 / cell_;
  a.pool = d_pool_; a.cap = kCapPerSlot; a.cell_start = d_cell_start_;
  a.wpts = d_wpts_; a.cent = d_cent_; a.spts = d_spts_; a.heads = d_heads_;
  a.keys0 = d_k0_; a.keys1 = d_k1_; a.vals0 = d_v0_; a.vals1 = d_v1_; a.flags = d_flags_; a.pos = d_pos_;
  a.d_n_cent = d_small_; a.d_counts = d_small_ + 8;
  a.rank = 0; a.world = 1; a.d_owned = nullptr;
  a.grid = d_grid_; a.grid_scan = d_grid_scan_; a.grid_is_clean = true;
  a.temp = d_temp_; a.temp_bytes = temp_bytes_;
  a.ht_key = d_ht_key_; a.ht_cnt = d_ht_cnt_; a.ht_off = d_ht_off_;
  a.ht_log2 = 12;
  while (((size_t)1 << a.ht_log2) < 2 * (n + 1)) ++a.ht_log2;
  MapFastArgs f{};
  f.d_in = d_in; f.n = (uint32_t)n; f.stride_floats = (uint32_t)stride_floats;
  f.transform = T != nullptr; if (T) f.pose = pose_from_array(T);
  f.d_world = d_world;
  f.origin[0] = origin_[0]; f.origin[1] = origin_[1]; f.origin[2] = origin_[2];
  f.d_cube_slot = d_cube_slot_; f.d_slot_count = d_slot_count_; f.d_slot_ok = d_slot_ok_;
  f.d_cube_cnt = d_cube_cnt_; f.d_scan_state = d_scan_state_; f.d_tickets = d_tickets_;
  f.d_small = d_small_; f.small_words = (uint32_t)kSmallWords;
  f.h_report = h_report_; f.seq = ++fast_seq_;
  f.per_round = (int32_t)per_round;
  launch_map_insert_fast(a, f, stream_);
  DM_TRY(hipGetLastError());
  DM_TRY(hipEventRecord(ev_fast_, stream_));
  pending_.on = true; pending_.d_xyz = a.d_xyz; pending_.n = n; pending_.stride = a.stride_floats; pending_.seq = f.seq;
  if (!defer) return settle(err);
  // the caller may recycle the buffer of the input points as soon as this returns: wait until the front kernel (the only one
  // that reads them; long done by the time a dozen launches have been enqueued) says so
  volatile unsigned long long* fs = &h_report_->front_seq;
  for (unsigned spin = 1; *fs != f.seq; ++spin)
    if ((spin & 0x3FFu) == 0 && hipEventQuery(ev_fast_) != hipErrorNotReady) break;  // (everything completed or failed: settle() sorts it out)
  (void)hipGetLastError();
  return 0;
}

void DeviceMap::settle_quiet() const {
  if (!pending_.on) return;
  DeviceMap* self = const_cast<DeviceMap*>(this);
  std::string e;
  if (const int rc = self->settle(e); rc < 0) { self->deferred_err_ = e; self->deferred_rc_ = rc; }
}

int DeviceMap::settle(std::string& err) {
  if (!pending_.on) {
    if (!deferred_err_.empty()) { err = deferred_err_; deferred_err_.clear(); const int rc = deferred_rc_; deferred_rc_ = 0; return rc < 0 ? rc : -2; }
    return 0;
  }
  volatile unsigned long long* seq = &h_report_->seq;
  for (unsigned spin = 1;; ++spin) {
    if (*seq == pending_.seq) break;
    if ((spin & 0x3FFu) == 0 && hipEventQuery(ev_fast_) != hipErrorNotReady) {
      (void)hipGetLastError();
      if (*seq == pending_.seq) break;
      // the launches are over and the report is not there: a kernel failed.  Nothing of the bookkeeping can be trusted to
      // match the device any more than after any other failed insert.
      pending_.on = false;
      grid_zero_upto_ = 0; block_clean_ = false; fast_clean_ = false; meta_dirty_ = true;
      err = "DeviceMap: the device-built insert did not report";
      return -2;
    }
  }
  std::atomic_thread_fence(std::memory_order_acquire);
  pending_.on = false;
  MapFastReport R;
  std::memcpy(&R, h_report_, sizeof(R));
  if (R.halt == kFastHaltNone) {
    ++fast_inserts_;
    // (the device has already moved ITS counts for every cube of the round: the host's copy follows for all of them, also when
    //  one of them is reported as an error below -- a stale host count uploaded by the next sync_meta would undo the device's)
    int bad = 0;
    for (uint32_t t = 0; t < R.n && t < (uint32_t)kMaxTouched; ++t) {
      const int cube = R.cube[t];
      const int s = (cube >= 0 && cube < kMapNum) ? cube_slot_[cube] : -1;
      if (s < 0) { if (!bad) { bad = -2; err = "DeviceMap: the device reported a cube without a slot"; } continue; }
      if (R.count[t] > kCapPerSlot && bad != -2) { bad = -1; err = "DeviceMap: a 50 m cube exceeds the per-cube capacity of 1M points"; }
      slot_count_[s] = std::min<uint32_t>(R.count[t], kCapPerSlot);
      slot_res_[s] = ((R.dirty >> t) & 1u) ? -plane_res_ : plane_res_;  // (see add_surf_legacy)
    }
    est_old_ = R.n_old;
    if (bad) { meta_dirty_ = true; return bad; }
    return (int)R.n_inside;
  }
  // the device could not lay the round out (or met a leaf the grouping kernels cannot sort): the map is unchanged, every
  // counter is back at zero -- the insert is repeated round by round
  ++fast_fallbacks_;
  if (R.halt == kFastHaltMultiRound) skip_fast_ = 16;
  if (R.halt == kFastHaltOverflow) grid_zero_upto_ = 0;  // (the centroids of the round were counted into the grids as they were produced)
  return add_surf_legacy(pending_.d_xyz, pending_.n, pending_.stride, err);
}

int DeviceMap::add_surf_legacy(const float* d_xyz, size_t n, size_t stride_floats, std::string& err) {
  if (!n) return 0;
  if (stride_floats == 0) stride_floats = 3;
  meta_dirty_ = true;  // (the host lays the rounds out and changes counts / slots: the device's copies follow before its next round)
  if (nc_ <= 1 && ncell1_ <= 2) { double cell; nc_ = cells_per_cube(plane_res_, &cell); cell_ = cell; ncell1_ = (uint32_t)((size_t)nc_ * nc_ * nc_ + 1); }
  if (ensure_work(n, err)) return -2;
  if (n > new_cap_) {
    if (d_cube_of_) (void)hipFree(d_cube_of_);
    d_cube_of_ = nullptr; new_cap_ = 0;
    DM_TRY(hipMalloc(reinterpret_cast<void**>(&d_cube_of_), (n + 1024) * sizeof(int32_t)));
    new_cap_ = n + 1024;
  }
  // 1. cube of every new point + touched flags (one small read-back)
  // (ONE fill clears the counters of the first round below, the per-rank counts and the flags -- enqueued behind the
  //  PREVIOUS insert, when the host has nothing else to do, see the end of this function)
  if (!block_clean_) DM_TRY(hipMemsetAsync(d_small_, 0, kSmallWords * sizeof(uint32_t) + kMapNum, stream_));
  block_clean_ = false;
  launch_world_cube(d_xyz, (uint32_t)n, (uint32_t)stride_floats, origin_, d_cube_of_, d_touched_, d_small_ + 48, stream_);
  DM_TRY(hipMemcpyAsync(h_small_ + 48, d_small_ + 48, (kSmallWords - 48) * sizeof(uint32_t) + kMapNum, hipMemcpyDeviceToHost, stream_));
  const float inv_leaf = 1.0f / plane_res_;
  const uint32_t lbits = leaf_bits(plane_res_);
  const size_t per_round = max_touched(lbits);  // at most kMaxTouched cubes (4 when the leaf coordinates need 10 key bits: planeRes < 0.1)
  // 2. one round: the cubes `cubes[0..count)` are re-filtered with the new points that fall into them
  auto run_round = [&](const int* cubes, int count, bool clear_counters) -> int {
    MapInsertArgs a{};
    MapTouched& tt = a.tt;
    tt.lbits = lbits;
    tt.n = count;
    uint32_t n_old = 0;
    for (int t = 0; t < tt.n; ++t) {
      const int cube = cubes[t];
      const int s = cube_slot_[cube];
      tt.cube[t] = cube;  // (ascending: the lists below are built in block order)
      tt.slot[t] = (uint32_t)s;
      tt.old_prefix[t] = n_old;
      n_old += slot_count_[s];
      const int ci = cube % kMapW, cj = (cube / kMapW) % kMapH, ck = cube / (kMapW * kMapH);
      const int w[3] = {ci - origin_[0], cj - origin_[1], ck - origin_[2]};
      for (int ax = 0; ax < 3; ++ax) {
        tt.cube_min[t][ax] = w[ax] * kCube - kHalfCube;
        tt.leaf_lo[t][ax] = (int)std::floor((float)tt.cube_min[t][ax] * inv_leaf) - 2;
        tt.wcube[t][ax] = w[ax];
      }
    }
    for (int t = tt.n; t <= kMaxTouched; ++t) tt.old_prefix[t] = n_old;
    for (int t = tt.n; t < kMaxTouched; ++t) { tt.slot[t] = 0; tt.cube[t] = INT32_MAX; }
    if (ensure_work((size_t)n_old + n, err)) return -2;
    tt.inv_leaf_watch = inv_leaf; tt.dirty = d_small_ + 7;  // (cleared with the counters)
    a.rank = rank_; a.world = world_; a.d_owned = world_ > 1 ? d_small_ + 64 : nullptr;
    if (clear_counters) {  // (the first round's counters were cleared together with the flags)
      if (world_ > 1) DM_TRY(hipMemsetAsync(d_small_ + 64, 0, kMaxTouched * sizeof(uint32_t), stream_));
      DM_TRY(hipMemsetAsync(d_small_, 0, 48 * sizeof(uint32_t), stream_));
    }
    a.d_xyz = d_xyz; a.n_new = (uint32_t)n; a.stride_floats = (uint32_t)stride_floats; a.n_old = n_old;
    a.d_cube_of = d_cube_of_; a.inv_leaf = inv_leaf;
    a.nc = nc_; a.ncell1 = ncell1_; a.inv_cell = 1.0 / cell_;
    a.pool = d_pool_; a.cap = kCapPerSlot; a.cell_start = d_cell_start_;
    a.wpts = d_wpts_; a.cent = d_cent_; a.spts = d_spts_; a.heads = d_heads_;
    a.keys0 = d_k0_; a.keys1 = d_k1_; a.vals0 = d_v0_; a.vals1 = d_v1_; a.flags = d_flags_; a.pos = d_pos_;
    a.d_n_cent = d_small_; a.d_counts = d_small_ + 8;
    a.d_tt = d_tt_;
    if (ensure_grid((size_t)tt.n * ncell1_ + 1024, err)) return -2;
    a.grid = d_grid_; a.grid_scan = d_grid_scan_;
    {  // the round leaves the counters it used at zero again (cell_table_kernel): fill only what no round has cleared yet
      const size_t need = (size_t)tt.n * ncell1_ + 1;
      a.grid_is_clean = need <= grid_zero_upto_;
      grid_zero_upto_ = std::max(grid_zero_upto_, need);
    }
    a.temp = d_temp_; a.temp_bytes = temp_bytes_;
    // first stage: leaf grouping through a hash table (default) or the stable radix sort of the whole working set
    // (the hash grouping lets an old point that shares its leaf with no NEW point pass through: valid when the cube holds one
    //  point per leaf of the CURRENT grid, i.e. it was last filtered at this planeRes; after a resolution change the first
    //  insert that touches a cube re-filters all of it, LocalMap.h:617-641 -- through the sort)
    bool one_point_per_leaf = true;
    for (int t = 0; t < tt.n; ++t) {
      const uint32_t sl = tt.slot[t];
      one_point_per_leaf = one_point_per_leaf && (slot_count_[sl] == 0 || slot_res_[sl] == plane_res_);
      // a cube last filtered on another grid (or marked by the drift watch: negative) goes in in the order of THAT grid's leaves
      // (a cube the drift watch marked at the CURRENT planeRes keeps pool order: its (cell, leaf) order is the output order of its
      //  last VoxelGrid on this very grid, whereas re-keying the drifted centroid would file it under the neighbouring leaf)
      a.old_inv_leaf[t] = (slot_count_[sl] > 0 && slot_res_[sl] != 0.f && std::fabs(slot_res_[sl]) != plane_res_) ? 1.0f / std::fabs(slot_res_[sl]) : 0.f;
      a.reorder_old = a.reorder_old || a.old_inv_leaf[t] > 0.f;
    }
    if (hash_grouping_ && one_point_per_leaf) {
      if (ensure_leaf_table(n, err)) return -2;
      a.ht_key = d_ht_key_; a.ht_cnt = d_ht_cnt_; a.ht_off = d_ht_off_;
      a.ht_log2 = 12;  // this round's share of the (all-empty) table: the kernels hash into / scan the first 2^ht_log2 slots only
      while (((size_t)1 << a.ht_log2) < 2 * (n + 1)) ++a.ht_log2;
    }
    launch_map_insert(a, stream_);
    DM_TRY(hipGetLastError());  // a refused launch must not pass for an insert
    DM_TRY(hipMemcpyAsync(h_small_, d_small_, kSmallWords * sizeof(uint32_t), hipMemcpyDeviceToHost, stream_));
    DM_TRY(hipStreamSynchronize(stream_));  // also keeps the staging buffer alive long enough
    if (a.ht_key && h_small_[5]) {
      // a leaf with more members than the grouping kernels sort in LDS: the second stage stood still (nothing of the map was
      // rewritten); the round is repeated with the sort-based first stage
      a.ht_key = a.ht_cnt = a.ht_off = nullptr;
      DM_TRY(hipMemsetAsync(d_small_, 0, 48 * sizeof(uint32_t), stream_));
      if (world_ > 1) DM_TRY(hipMemsetAsync(d_small_ + 64, 0, kMaxTouched * sizeof(uint32_t), stream_));
      launch_map_insert(a, stream_);
      DM_TRY(hipGetLastError());
      DM_TRY(hipMemcpyAsync(h_small_, d_small_, 48 * sizeof(uint32_t), hipMemcpyDeviceToHost, stream_));
      if (world_ > 1) DM_TRY(hipMemcpyAsync(h_small_ + 64, d_small_ + 64, kMaxTouched * sizeof(uint32_t), hipMemcpyDeviceToHost, stream_));
      DM_TRY(hipStreamSynchronize(stream_));
    }
    for (int t = 0; t < tt.n; ++t) {
      const uint32_t cnt = h_small_[8 + t];
      if (cnt > kCapPerSlot) { err = "DeviceMap: a 50 m cube exceeds the per-cube capacity of 1M points"; return -1; }
      slot_count_[tt.slot[t]] = cnt;
      // the whole cube has just been filtered at this leaf size -- unless one of its centroids drifted out of its leaf
      // (MapTouched::dirty): then two points may share a leaf and the next insert must not use the pass-through
      slot_res_[tt.slot[t]] = ((h_small_[7] >> t) & 1u) ? -plane_res_ : plane_res_;
      if (world_ > 1) slot_owned_[tt.slot[t]] = h_small_[64 + t];
    }
    return 0;
  };
  DM_TRY(hipStreamSynchronize(stream_));
  const int inserted_total = (int)h_small_[48];  // points inside the 21x21x11 window (LocalMap.h:605)
  std::vector<int> touched;
  for (int cube = 0; cube < kMapNum; ++cube) if (h_touched_[cube]) touched.push_back(cube);
  // the next insert's fill, now: the stream is idle and the host is about to return
  auto clear_for_next = [&]() { if (hipMemsetAsync(d_small_, 0, kSmallWords * sizeof(uint32_t) + kMapNum, stream_) == hipSuccess) block_clean_ = true; else (void)hipGetLastError(); };
  if (touched.empty()) { clear_for_next(); return 0; }
  for (int cube : touched) if (cube_slot_[cube] < 0) alloc_slot(cube);
  if (ensure_pool((int)slot_cube_.size(), err)) return -2;
  for (size_t r0 = 0; r0 < touched.size(); r0 += per_round) {  // (block order: MapTouched::cube ascends)
    const int rc = run_round(touched.data() + r0, (int)std::min<size_t>(per_round, touched.size() - r0), r0 > 0);
    if (rc) return rc;
  }
  clear_for_next();
  return inserted_total;
}

// ---- sharded map, planeRes change: re-cut of the shards -----------------------------------------------------------
// The shards are cut along the cell grid (a rank keeps the leaves within one cell of its bricks), and cells and bricks
// follow planeRes.  export_owned() hands out this rank's share of the FULL map -- the points whose own cell (current
// grid) lies in one of its bricks: every point of the map is owned by exactly one rank -- cube by cube, with the
// planeRes the cube was last filtered with (the re-filter order needs it).  The caller all-gathers the blobs;
// reshard() then keeps, of all points, those whose leaf on the NEW grid this rank would keep at an insert, and rebuilds
// the index.  Record: int32 cube, float res (negative: the drift watch had marked the cube), uint32 n, n x 3 floats.
int DeviceMap::export_owned(std::vector<uint8_t>& blob, std::string& err) {
  if (const int rs = settle(err); rs < 0) return rs;
  blob.clear();
  std::vector<float> tmp;
  const double inv_cell = 1.0 / cell_;
  for (int cube = 0; cube < kMapNum; ++cube) {
    const int s = cube_slot_[cube];
    if (s < 0 || slot_count_[s] == 0) continue;
    const uint32_t cnt = slot_count_[s];
    if ((size_t)cnt * 3 > stage_cap_) {
      if (d_stage_) (void)hipFree(d_stage_);
      d_stage_ = nullptr; stage_cap_ = 0;
      DM_TRY(hipMalloc(reinterpret_cast<void**>(&d_stage_), ((size_t)cnt * 3 + 1024) * sizeof(float)));
      stage_cap_ = (size_t)cnt * 3 + 1024;
    }
    tmp.resize((size_t)cnt * 3);
    launch_gather_export(d_pool_, kCapPerSlot, (uint32_t)s, cnt, d_stage_, stream_);
    DM_TRY(hipMemcpyAsync(tmp.data(), d_stage_, (size_t)cnt * 12, hipMemcpyDeviceToHost, stream_));
    DM_TRY(hipStreamSynchronize(stream_));
    const int ci = cube % kMapW, cj = (cube / kMapW) % kMapH, ck = cube / (kMapW * kMapH);
    const int w[3] = {ci - origin_[0], cj - origin_[1], ck - origin_[2]};
    std::vector<float> mine;
    for (uint32_t i = 0; i < cnt; ++i) {
      int g[3];
      for (int a = 0; a < 3; ++a) {  // count_owned_kernel's arithmetic
        const int v = (int)std::floor(((double)tmp[3 * (size_t)i + a] - (double)(w[a] * kCube - kHalfCube)) * inv_cell);
        g[a] = v < 0 ? 0 : (v >= nc_ ? nc_ - 1 : v);
      }
      if ((int)(brick_hash(w[0], w[1], w[2], g[0] / kBrickCells, g[1] / kBrickCells, g[2] / kBrickCells) % (uint32_t)world_) == rank_)
        mine.insert(mine.end(), tmp.begin() + 3 * (size_t)i, tmp.begin() + 3 * (size_t)i + 3);
    }
    if (mine.empty()) continue;
    const int32_t c32 = cube; const float res = slot_res_[s]; const uint32_t n = (uint32_t)(mine.size() / 3);
    const size_t at = blob.size();
    blob.resize(at + 12 + mine.size() * sizeof(float));
    std::memcpy(&blob[at], &c32, 4); std::memcpy(&blob[at + 4], &res, 4); std::memcpy(&blob[at + 8], &n, 4);
    std::memcpy(&blob[at + 12], mine.data(), mine.size() * sizeof(float));
  }
  return 0;
}

int DeviceMap::reshard(const std::vector<std::vector<uint8_t>>& blobs, float line_res, float plane_res, std::string& err) {
  if (const int rs = settle(err); rs < 0) return rs;
  meta_dirty_ = true;
  struct CubeSet { float res = 0.f; std::vector<float> xyz; };
  std::map<int, CubeSet> cubes;  // ascending block index
  for (const std::vector<uint8_t>& b : blobs) {
    size_t at = 0;
    while (at + 12 <= b.size()) {
      int32_t cube; float res; uint32_t n;
      std::memcpy(&cube, &b[at], 4); std::memcpy(&res, &b[at + 4], 4); std::memcpy(&n, &b[at + 8], 4);
      at += 12;
      if (cube < 0 || cube >= kMapNum || at + (size_t)n * 12 > b.size()) { err = "DeviceMap::reshard: malformed record"; return -2; }
      CubeSet& cs = cubes[cube];
      if (cs.res != 0.f && std::fabs(cs.res) != std::fabs(res)) { err = "DeviceMap::reshard: the ranks disagree about a cube's filter resolution"; return -2; }
      cs.res = (cs.res < 0.f || res < 0.f) ? -std::fabs(res) : res;  // (marked by the drift watch on any rank: marked)
      const size_t k = cs.xyz.size();
      cs.xyz.resize(k + (size_t)n * 3);
      std::memcpy(cs.xyz.data() + k, &b[at], (size_t)n * 12);
      at += (size_t)n * 12;
    }
    if (at != b.size()) { err = "DeviceMap::reshard: trailing bytes in a record stream"; return -2; }
  }
  double cell_new;
  const int nc_new = cells_per_cube(plane_res, &cell_new);
  const float inv_leaf_new = 1.0f / plane_res;
  if (ensure_work(1, err)) return -2;
  for (size_t s = 0; s < slot_cube_.size(); ++s) { slot_count_[s] = 0; slot_owned_[s] = 0; slot_full_[s] = 0; }
  for (const auto& kv : cubes) if (cube_slot_[kv.first] < 0) alloc_slot(kv.first);
  if (ensure_pool((int)slot_cube_.size(), err)) return -2;
  block_clean_ = false;
  for (const auto& kv : cubes) {
    const int cube = kv.first;
    const CubeSet& cs = kv.second;
    const uint32_t n = (uint32_t)(cs.xyz.size() / 3);
    if (!n) continue;
    if (n > kCapPerSlot) { err = "DeviceMap: a 50 m cube exceeds the per-cube capacity of 1M points"; return -1; }
    const int s = cube_slot_[cube];
    if ((size_t)n * 3 > stage_cap_) {
      if (d_stage_) (void)hipFree(d_stage_);
      d_stage_ = nullptr; stage_cap_ = 0;
      DM_TRY(hipMalloc(reinterpret_cast<void**>(&d_stage_), ((size_t)n * 3 + 1024) * sizeof(float)));
      stage_cap_ = (size_t)n * 3 + 1024;
    }
    MapTouched tt{};
    tt.n = 1;
    const int ci = cube % kMapW, cj = (cube / kMapW) % kMapH, ck = cube / (kMapW * kMapH);
    const int w[3] = {ci - origin_[0], cj - origin_[1], ck - origin_[2]};
    for (int ax = 0; ax < 3; ++ax) { tt.cube_min[0][ax] = w[ax] * kCube - kHalfCube; tt.wcube[0][ax] = w[ax]; }
    DM_TRY(hipMemcpyAsync(d_stage_, cs.xyz.data(), (size_t)n * 12, hipMemcpyHostToDevice, stream_));
    DM_TRY(hipMemsetAsync(d_small_, 0, 2 * sizeof(uint32_t), stream_));
    launch_shard_select(d_stage_, n, tt, inv_leaf_new, nc_new, 1.0 / cell_new, rank_, world_, d_pool_ + (size_t)s * kCapPerSlot, kCapPerSlot, d_small_, stream_);
    DM_TRY(hipGetLastError());
    DM_TRY(hipMemcpyAsync(h_small_, d_small_, 2 * sizeof(uint32_t), hipMemcpyDeviceToHost, stream_));
    DM_TRY(hipStreamSynchronize(stream_));
    slot_count_[s] = h_small_[0]; slot_owned_[s] = h_small_[1]; slot_full_[s] = n;  // (n = the cube's count in the full map)
    slot_res_[s] = cs.res;
  }
  slot_table_dirty_ = true;
  return set_resolution(line_res, plane_res, err);  // the index over the new resident set (launch_map_retable)
}

size_t DeviceMap::export_points(float* xyz, size_t cap, bool only_5x5, const int pos[3], std::string& err) {
  if (settle(err) < 0) return 0;
  size_t n = 0;
  for (int cube = 0; cube < kMapNum; ++cube) {  // ascending cube index, canonical order inside the cube
    const int s = cube_slot_[cube];
    if (s < 0 || slot_count_[s] == 0) continue;
    if (only_5x5) {
      const int ci = cube % kMapW, cj = (cube / kMapW) % kMapH, ck = cube / (kMapW * kMapH);
      if (std::abs(ci - pos[0]) > 2 || std::abs(cj - pos[1]) > 2 || std::abs(ck - pos[2]) > 1) continue;
    }
    const uint32_t cnt = slot_count_[s];
    if (xyz && n + cnt <= cap) {
      if ((size_t)cnt * 3 > stage_cap_) {
        if (d_stage_) (void)hipFree(d_stage_);
        d_stage_ = nullptr; stage_cap_ = 0;
        if (hipMalloc(reinterpret_cast<void**>(&d_stage_), ((size_t)cnt * 3 + 1024) * sizeof(float)) != hipSuccess) { err = "DeviceMap: export staging alloc failed"; return n; }
        stage_cap_ = (size_t)cnt * 3 + 1024;
      }
      launch_gather_export(d_pool_, kCapPerSlot, (uint32_t)s, cnt, d_stage_, stream_);
      if (hipMemcpyAsync(xyz + 3 * n, d_stage_, (size_t)cnt * 12, hipMemcpyDeviceToHost, stream_) != hipSuccess ||
          hipStreamSynchronize(stream_) != hipSuccess) { err = "DeviceMap: export copy failed"; return n; }
    }
    n += cnt;
  }
  return n;
}

size_t DeviceMap::export_records(void* out, size_t stride, size_t cap, bool only_5x5, const int pos[3], std::string& err) {
  if (settle(err) < 0) return 0;
  std::vector<std::pair<int, uint32_t>> todo;  // (slot, count) in ascending cube index: the order of export_points
  size_t n = 0;
  for (int cube = 0; cube < kMapNum; ++cube) {
    const int s = cube_slot_[cube];
    if (s < 0 || slot_count_[s] == 0) continue;
    if (only_5x5) {
      const int ci = cube % kMapW, cj = (cube / kMapW) % kMapH, ck = cube / (kMapW * kMapH);
      if (std::abs(ci - pos[0]) > 2 || std::abs(cj - pos[1]) > 2 || std::abs(ck - pos[2]) > 1) continue;
    }
    todo.emplace_back(s, slot_count_[s]);
    n += slot_count_[s];
  }
  if (!out || n > cap || !n) return n;
  const size_t words = n * (stride / 4);
  if (words > stage_cap_) {  // (the staging buffer of export_points, counted in 32-bit words)
    if (d_stage_) (void)hipFree(d_stage_);
    d_stage_ = nullptr; stage_cap_ = 0;
    if (hipMalloc(reinterpret_cast<void**>(&d_stage_), (words + 1024) * sizeof(float)) != hipSuccess) { err = "DeviceMap: export staging alloc failed"; return 0; }
    stage_cap_ = words + 1024;
  }
  size_t at = 0;
  for (const auto& sc : todo) {
    launch_gather_export_records(d_pool_, kCapPerSlot, (uint32_t)sc.first, sc.second, reinterpret_cast<uint32_t*>(d_stage_) + at * (stride / 4), (uint32_t)(stride / 4), stream_);
    at += sc.second;
  }
  if (hipMemcpyAsync(out, d_stage_, n * stride, hipMemcpyDeviceToHost, stream_) != hipSuccess || hipStreamSynchronize(stream_) != hipSuccess) {
    err = "DeviceMap: export copy failed"; return 0;
  }
  return n;
}

}  // namespace soicp
