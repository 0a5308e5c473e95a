// map_kernels.hip -- device-resident LocalMap update: the GPU counterpart of
//   LocalMap::addSurfPointCloud   include/super_odometry/LidarProcess/LocalMap.h:591-645
//     (bin the world-frame points into 50 m cubes, pcl::VoxelGrid(planeRes) per touched cube, rebuild the index)
// which the reference executes on the CPU after every scan (transformAndAddToMap, LidarSlam.cpp:60-80,163-167).
//
// Pipeline (all on the context's stream, two small read-backs):
//   world_cube_kernel     cube index per point (the int((c+25)/50), "--" rule) + touched-cube flags
//   gather_old_kernel     the touched cubes' current points become the head of the working set (old first: a leaf
//                         holds at most one old centroid, and PCL accumulates in input order)
//   append_new_kernel     the new points follow in input order; key = (touched-cube id, leaf z, y, x)
//   rocPRIM stable sort   by leaf key
//   leaf_heads_kernel     working set gathered into leaf-sorted order, first index of every leaf
//   leaf_centroid_kernel  one thread per leaf: float sums IN ORDER, centroid = sum / count  (VoxelGrid semantics)
//   second stage (no sort): cell_count_kernel (centroids counted into the dense cell grids of the touched cubes, atomic rank),
//                         exclusive scan of the grids = the cubes' new cell_start tables (cell_table_kernel), cell_place_kernel
//                         (into a scratch array, leaf key alongside), cell_rank_kernel (final position inside the cell =
//                         number of smaller leaf keys: canonical order = cell, then leaf); cell_table_kernel also puts the
//                         counters back to zero, so the next insert needs no fill of the grids.
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <cstring>

#include <rocprim/rocprim.hpp>

#include "deskew_math.h"
#include "local_map.h"
#include "map_kernels.h"
#include "so_math.h"

namespace soicp {

__device__ __forceinline__ int cube_coord_f(float c, int origin) {  // == int((c + 25.0) / 50.0) (+origin), "--" if negative
  const double s = (double)c + 25.0;                                 // (exact for float inputs, see kernels.hip)
  int i = (int)(s * 0.02) + origin;
  if (s < 0) i--;
  return i;
}

// LocalMap.h:596-610
__global__ __launch_bounds__(1024) void world_cube_kernel(const float* __restrict__ xyz, uint32_t n, uint32_t stride_floats,
                                                          int o0, int o1, int o2, int32_t* __restrict__ cube_of,
                                                          uint8_t* __restrict__ touched, uint32_t* __restrict__ n_inside) {
  __shared__ uint32_t cnt;
  if (threadIdx.x == 0) cnt = 0;
  __syncthreads();
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  int cube = -1;
  if (i < n) {
    const float* p = xyz + (size_t)i * stride_floats;
    const int ci = cube_coord_f(p[0], o0), cj = cube_coord_f(p[1], o1), ck = cube_coord_f(p[2], o2);
    if (ci >= 0 && ci < 21 && cj >= 0 && cj < 21 && ck >= 0 && ck < 11) cube = ci + 21 * cj + 21 * 21 * ck;
    cube_of[i] = cube;
  }
  // one flag store per DISTINCT cube of the wavefront and one counter atomic per workgroup (every lane storing the same
  // byte / every wavefront adding to the same word cost 50 us for 131 072 points)
  unsigned long long todo = __ballot(cube >= 0);
  const unsigned long long inside = todo;
  while (todo) {
    const int leader = __ffsll((long long)todo) - 1;
    const int c = __builtin_amdgcn_readlane(cube, leader);
    if ((threadIdx.x & 63) == leader) touched[c] = 1;
    todo &= ~__ballot(cube == c);
  }
  if ((threadIdx.x & 63) == 0 && inside) atomicAdd(&cnt, (uint32_t)__popcll(inside));
  __syncthreads();
  if (threadIdx.x == 0 && cnt) atomicAdd(n_inside, cnt);
}

// scan + world transform in one pass (transformAndAddToMap, LidarSlam.cpp:60-80; TransformPoint, superodom_utils.h:119-123)
__global__ __launch_bounds__(256) void transform_scan_kernel(const float* __restrict__ scan, uint32_t n, Pose pose,
                                                             float* __restrict__ out_xyz) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double wx, wy, wz;
  quat_rotate<double>(pose.q, (double)scan[3 * i], (double)scan[3 * i + 1], (double)scan[3 * i + 2], wx, wy, wz);
  out_xyz[3 * i] = (float)(wx + pose.t[0]); out_xyz[3 * i + 1] = (float)(wy + pose.t[1]); out_xyz[3 * i + 2] = (float)(wz + pose.t[2]);
}

// leaf coordinate = floor(v * inv_leaf) evaluated in FLOAT exactly like pcl::VoxelGrid; an arbitrary common offset per
// cube keeps the lexicographic (z, y, x) order, which is all VoxelGrid's linear leaf index is used for.
// Key = touched-cube id above three leaf coordinates of `lbits` bits each: 9 bits (<= 508 leaves per axis of a 50 m cube, i.e.
// planeRes >= 0.1, and 32 cubes per round) or 10 bits (planeRes down to 0.05, 4 cubes per round) -- MapTouched::lbits.
__device__ __forceinline__ uint32_t leaf_key(float x, float y, float z, float inv_leaf, int lo0, int lo1, int lo2, uint32_t tid, uint32_t lbits) {
  const int l0 = (int)floorf(x * inv_leaf) - lo0, l1 = (int)floorf(y * inv_leaf) - lo1, l2 = (int)floorf(z * inv_leaf) - lo2;
  const uint32_t m = (1u << lbits) - 1u;
  return (tid << (3u * lbits)) | (((uint32_t)l2 & m) << (2u * lbits)) | (((uint32_t)l1 & m) << lbits) | ((uint32_t)l0 & m);
}

__global__ __launch_bounds__(256) void gather_old_kernel(const MapTouched* __restrict__ ttp, const float4* __restrict__ pool, uint32_t cap, float inv_leaf,
                                                         uint32_t n_old, float4* __restrict__ wpts, uint32_t* __restrict__ keys,
                                                         uint32_t* __restrict__ vals) {
  const MapTouched& tt = *ttp;
  const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n_old) return;
  int t = 0;
#pragma unroll
  for (int step = 16; step >= 1; step >>= 1) t = (t + step < tt.n && tt.old_prefix[t + step] <= e) ? t + step : t;
  const float4 p = pool[(size_t)tt.slot[t] * cap + (e - tt.old_prefix[t])];
  wpts[e] = p;
  keys[e] = leaf_key(p.x, p.y, p.z, inv_leaf, tt.leaf_lo[t][0], tt.leaf_lo[t][1], tt.leaf_lo[t][2], (uint32_t)t, tt.lbits);
  vals[e] = e;
}

// Re-filtering a cube on a new leaf grid (first insert that touches it after a planeRes change): the reference's block
// cloud is still in the output order of its LAST VoxelGrid -- ascending leaf index of the old grid -- and the new, coarser
// leaves sum their several old points in that order.  The pool is in (cell, leaf) order, which is the same order only
// inside one cell; a leaf that straddles two cells would add its points in another order (1 ulp in the centroid).  So:
// key every old point by its leaf on the OLD grid of its cube (10 bits per axis, no cube id: the main sort groups by cube
// and is stable), sort, and gather the working set's head in that order.
struct OldGrids { float inv_leaf[kMaxTouched]; };
__global__ __launch_bounds__(256) void old_order_key_kernel(const MapTouched* __restrict__ ttp, OldGrids og, const float4* __restrict__ pool, uint32_t cap, uint32_t n_old,
                                                            uint32_t* __restrict__ keys, uint32_t* __restrict__ vals) {
  const MapTouched& tt = *ttp;
  const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n_old) return;
  int t = 0;
#pragma unroll
  for (int step = 16; step >= 1; step >>= 1) t = (t + step < tt.n && tt.old_prefix[t + step] <= e) ? t + step : t;
  const float4 p = pool[(size_t)tt.slot[t] * cap + (e - tt.old_prefix[t])];
  const float il = og.inv_leaf[t];
  uint32_t k = 0u;  // (0: a cube whose points keep the pool order -- equal keys, stable sort)
  if (il > 0.f) {
    const int lo0 = (int)floorf((float)tt.cube_min[t][0] * il) - 2, lo1 = (int)floorf((float)tt.cube_min[t][1] * il) - 2,
              lo2 = (int)floorf((float)tt.cube_min[t][2] * il) - 2;
    k = leaf_key(p.x, p.y, p.z, il, lo0, lo1, lo2, 0u, 10u);
  }
  keys[e] = k;
  vals[e] = e;
}
__global__ __launch_bounds__(256) void gather_old_ordered_kernel(const MapTouched* __restrict__ ttp, const float4* __restrict__ pool, uint32_t cap, float inv_leaf,
                                                                 uint32_t n_old, const uint32_t* __restrict__ order, float4* __restrict__ wpts,
                                                                 uint32_t* __restrict__ keys, uint32_t* __restrict__ vals) {
  const MapTouched& tt = *ttp;
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_old) return;
  const uint32_t e = order[i];
  int t = 0;
#pragma unroll
  for (int step = 16; step >= 1; step >>= 1) t = (t + step < tt.n && tt.old_prefix[t + step] <= e) ? t + step : t;
  const float4 p = pool[(size_t)tt.slot[t] * cap + (e - tt.old_prefix[t])];
  wpts[i] = p;
  keys[i] = leaf_key(p.x, p.y, p.z, inv_leaf, tt.leaf_lo[t][0], tt.leaf_lo[t][1], tt.leaf_lo[t][2], (uint32_t)t, tt.lbits);
  vals[i] = i;
}

// Sharded map: does rank `rank` keep the leaf that the point (x, y, z) of touched cube t falls into?  The decision is a
// function of the LEAF (its float index per axis, exactly as pcl::VoxelGrid computes it), so all the points of a leaf share
// it and every kept centroid is the centroid of the whole leaf -- bit-identical to the unsharded map.  A leaf is kept when
// its box (widened by the rounding of x * inv_leaf) overlaps a cell within one cell of a brick the rank owns: the shard
// then holds every centroid that can lie in the gate ball of a query it owns.
__device__ __forceinline__ bool shard_keeps_leaf(float x, float y, float z, float inv_leaf, const MapTouched& tt, int t, int nc,
                                                 double inv_cell, int rank, int world) {
  const float p[3] = {x, y, z};
  int blo[3], bhi[3];
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const double l = (double)floorf(p[a] * inv_leaf);
    const double leaf = 1.0 / (double)inv_leaf;
    const double margin = 1e-3 + fabs(l * leaf) * 1e-6;
    const double x0 = l * leaf - margin, x1 = (l + 1.0) * leaf + margin;
    int c0 = (int)floor((x0 - tt.cube_min[t][a]) * inv_cell) - 1, c1 = (int)floor((x1 - tt.cube_min[t][a]) * inv_cell) + 1;
    c0 = c0 < 0 ? 0 : (c0 >= nc ? nc - 1 : c0); c1 = c1 < 0 ? 0 : (c1 >= nc ? nc - 1 : c1);
    blo[a] = c0 / kBrickCells; bhi[a] = c1 / kBrickCells;
  }
  for (int bz = blo[2]; bz <= bhi[2]; ++bz)
    for (int by = blo[1]; by <= bhi[1]; ++by)
      for (int bx = blo[0]; bx <= bhi[0]; ++bx)
        if ((int)(brick_hash(tt.wcube[t][0], tt.wcube[t][1], tt.wcube[t][2], bx, by, bz) % (uint32_t)world) == rank) return true;
  return false;
}

// cube -> t of THIS round (binary search in the ascending list the launch carries), -1: a cube of another round
__device__ __forceinline__ int touched_index(const MapTouched& tt, int cube) {
  static_assert(kMaxTouched == 32, "five halving steps");
  int t = 0;
#pragma unroll
  for (int step = 16; step >= 1; step >>= 1) t = tt.cube[t + step] <= cube ? t + step : t;
  return tt.cube[t] == cube ? t : -1;
}

__global__ __launch_bounds__(256) void append_new_kernel(const float* __restrict__ xyz, uint32_t n, uint32_t stride_floats,
                                                         const int32_t* __restrict__ cube_of,
                                                         const MapTouched* __restrict__ ttp, float inv_leaf, uint32_t n_old, float4* __restrict__ wpts,
                                                         uint32_t* __restrict__ keys, uint32_t* __restrict__ vals, int nc, double inv_cell,
                                                         int rank, int world) {
  const MapTouched& tt = *ttp;
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float* p = xyz + (size_t)i * stride_floats;
  const int cube = cube_of[i];
  const uint32_t e = n_old + i;
  wpts[e] = make_float4(p[0], p[1], p[2], 0.f);
  vals[e] = e;
  if (cube < 0) { keys[e] = 0xFFFFFFFFu; return; }  // outside the 21x21x11 window: dropped (LocalMap.h:605)
  const int t = touched_index(tt, cube);
  if (t < 0) { keys[e] = 0xFFFFFFFFu; return; }     // a cube handled by another round of this insert
  if (world > 1 && !shard_keeps_leaf(p[0], p[1], p[2], inv_leaf, tt, t, nc, inv_cell, rank, world)) { keys[e] = 0xFFFFFFFFu; return; }  // another rank's leaf
  keys[e] = leaf_key(p[0], p[1], p[2], inv_leaf, tt.leaf_lo[t][0], tt.leaf_lo[t][1], tt.leaf_lo[t][2], (uint32_t)t, tt.lbits);
}

// sharded map: number of the cube's points whose OWN cell lies in a brick of this rank (every point of the full map is
// counted by exactly one rank: the sum over the ranks is the block's cloud size the reference reports, LocalMap.h:292-318)
__global__ __launch_bounds__(256) void count_owned_kernel(const float4* __restrict__ pool, uint32_t cap, const MapTouched* __restrict__ ttp,
                                                          const uint32_t* __restrict__ counts, int nc, double inv_cell, int rank, int world,
                                                          uint32_t* __restrict__ owned) {
  const MapTouched& tt = *ttp;
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  const int t = blockIdx.y;
  bool mine = false;
  if (i < counts[t] && i < cap) {
    const float4 p = pool[(size_t)tt.slot[t] * cap + i];
    const float c3[3] = {p.x, p.y, p.z};
    int g[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const int v = (int)floor(((double)c3[a] - tt.cube_min[t][a]) * inv_cell);
      g[a] = v < 0 ? 0 : (v >= nc ? nc - 1 : v);
    }
    mine = (int)(brick_hash(tt.wcube[t][0], tt.wcube[t][1], tt.wcube[t][2], g[0] / kBrickCells, g[1] / kBrickCells, g[2] / kBrickCells) % (uint32_t)world) == rank;
  }
  const unsigned long long m = __ballot(mine);
  if ((threadIdx.x & 63) == 0 && m) atomicAdd(&owned[t], (uint32_t)__popcll(m));
}

static inline dim3 grid_for_n(uint32_t n) { return dim3((n + 255u) / 256u); }
// Re-cut of a shard after a planeRes change (DeviceMap::reshard).  The candidates are ALL points of one cube -- every
// rank's owned points, gathered --; this rank keeps those whose leaf ON THE NEW GRID it would keep at an insert
// (shard_keeps_leaf with the new leaf size, cell size and bricks), so that the next re-filter finds every old point of
// every leaf it keeps, and nothing of the others.  Order in the pool: arbitrary (the retable that follows sorts).
__global__ __launch_bounds__(256) void shard_select_kernel(const float* __restrict__ xyz, uint32_t n, MapTouched tt, float inv_leaf, int nc,
                                                           double inv_cell, int rank, int world, float4* __restrict__ pool_slot, uint32_t cap,
                                                           uint32_t* __restrict__ counters /* [0] kept, [1] owned */) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  bool keep = false, mine = false;
  float x = 0.f, y = 0.f, z = 0.f;
  if (i < n) {
    x = xyz[3 * (size_t)i]; y = xyz[3 * (size_t)i + 1]; z = xyz[3 * (size_t)i + 2];
    keep = shard_keeps_leaf(x, y, z, inv_leaf, tt, 0, nc, inv_cell, rank, world);
    const float c3[3] = {x, y, z};
    int g[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const int v = (int)floor(((double)c3[a] - tt.cube_min[0][a]) * inv_cell);
      g[a] = v < 0 ? 0 : (v >= nc ? nc - 1 : v);
    }
    mine = (int)(brick_hash(tt.wcube[0][0], tt.wcube[0][1], tt.wcube[0][2], g[0] / kBrickCells, g[1] / kBrickCells, g[2] / kBrickCells) % (uint32_t)world) == rank;
  }
  const int lane = threadIdx.x & 63;
  const unsigned long long mk = __ballot(keep), mo = __ballot(mine);
  uint32_t base = 0;
  if (lane == 0) {
    if (mk) base = atomicAdd(&counters[0], (uint32_t)__popcll(mk));
    if (mo) atomicAdd(&counters[1], (uint32_t)__popcll(mo));
  }
  base = (uint32_t)__shfl((int)base, 0, 64);
  if (keep) {
    const uint32_t at = base + (uint32_t)__popcll(mk & ((1ull << lane) - 1ull));
    if (at < cap) pool_slot[at] = make_float4(x, y, z, 0.f);
  }
}
void launch_shard_select(const float* d_xyz, uint32_t n, const MapTouched& tt, float inv_leaf, int nc, double inv_cell, int rank, int world,
                         float4* pool_slot, uint32_t cap, uint32_t* d_counters, hipStream_t s) {
  if (!n) return;
  hipLaunchKernelGGL(shard_select_kernel, grid_for_n(n), dim3(256), 0, s, d_xyz, n, tt, inv_leaf, nc, inv_cell, rank, world, pool_slot, cap, d_counters);
}

__global__ __launch_bounds__(256) void leaf_flags_kernel(const uint32_t* __restrict__ keys, uint32_t n, uint32_t* __restrict__ flags) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  flags[i] = (keys[i] != 0xFFFFFFFFu && (i == 0 || keys[i] != keys[i - 1])) ? 1u : 0u;
}

constexpr uint32_t kLongLeaf = 64, kMaxLongLeaves = 4096;

// centroid = float sum / float count, then its cell key for the second sort
// (cc.grid != nullptr: the centroid is also counted into its cell of the dense grids right away -- the second stage's
//  cell_count_kernel folded into the kernels that produce the centroids; its rank inside the cell goes to cc.rank[o])
struct CellCount { uint32_t* grid; uint32_t* rank; uint32_t ncell1; };
__device__ __forceinline__ uint32_t emit_centroid(uint32_t o, uint32_t key, float s0, float s1, float s2, uint32_t count, const MapTouched& tt,
                                                  int nc, double inv_cell, float4* __restrict__ cent, uint32_t* __restrict__ keys2,
                                                  uint32_t* __restrict__ vals2, const CellCount cc = CellCount{nullptr, nullptr, 0u}) {
  const float cnt = (float)count;
  const float cx = s0 / cnt, cy = s1 / cnt, cz = s2 / cnt;
  cent[o] = make_float4(cx, cy, cz, 0.f);
  const int t = (int)(key >> (3u * tt.lbits));
  int g[3];
  const float c3[3] = {cx, cy, cz};
#pragma unroll
  for (int a = 0; a < 3; ++a) {  // cell of the hashed-voxel grid: floor((p - cube_min) * inv_cell), clamped (local_map.cpp: cell_of)
    const int v = (int)floor(((double)c3[a] - tt.cube_min[t][a]) * inv_cell);
    g[a] = v < 0 ? 0 : (v >= nc ? nc - 1 : v);
  }
  const uint32_t cell = (uint32_t)((g[2] * nc + g[1]) * nc + g[0]);
  const uint32_t k2 = ((uint32_t)t << 18) | cell;  // linear cell index, as in the cube's table
  keys2[o] = k2;
  if (vals2) vals2[o] = o;  // (only the sort-based second stage reads it)
  if (cc.grid) cc.rank[o] = atomicAdd(&cc.grid[(size_t)t * cc.ncell1 + cell], 1u);
  // (a lone point is its own centroid and lies in its leaf by construction; see MapTouched::dirty)
  if (count > 1u && tt.dirty && leaf_key(cx, cy, cz, tt.inv_leaf_watch, tt.leaf_lo[t][0], tt.leaf_lo[t][1], tt.leaf_lo[t][2], (uint32_t)t, tt.lbits) != key)
    atomicOr(tt.dirty, 1u << t);
  return k2;
}

// leaf-sorted working set made contiguous (spts[i] = wpts[vals[i]]) + first index of every leaf (heads[ordinal]; the
// entry behind the last leaf = number of valid elements) + number of leaves
__global__ __launch_bounds__(256) void leaf_heads_kernel(const uint32_t* __restrict__ keys, const uint32_t* __restrict__ vals,
                                                         const uint32_t* __restrict__ flags, const uint32_t* __restrict__ pos,
                                                         uint32_t n, const float4* __restrict__ wpts, float4* __restrict__ spts,
                                                         uint32_t* __restrict__ heads, uint32_t* __restrict__ n_cent) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  spts[i] = wpts[vals[i]];
  const uint32_t f = flags[i], o = pos[i];
  if (f) heads[o] = i;
  const bool valid = keys[i] != 0xFFFFFFFFu;
  if (valid && (i + 1 == n || keys[i + 1] == 0xFFFFFFFFu)) { heads[o + f] = i + 1; *n_cent = o + f; }  // last valid element
  if (i == 0 && !valid) *n_cent = 0;
}

// leaf_flags_kernel + exclusive scan + leaf_heads_kernel in ONE launch (the scan pre-filter; four launches before): a
// single-pass scan of the "first element of its leaf" flags with decoupled look-back (see cell_scan_table_kernel: ticket for
// the workgroup order, records = flag << 62 | value, all zero before the launch), 2 048 elements per workgroup.
constexpr uint32_t kHeadsItems = 2048;
__global__ __launch_bounds__(256) void leaf_heads_scan_kernel(const uint32_t* __restrict__ keys, const uint32_t* __restrict__ vals, uint32_t n,
                                                              const float4* __restrict__ wpts, float4* __restrict__ spts, uint32_t* __restrict__ heads,
                                                              uint32_t* __restrict__ n_cent, unsigned long long* __restrict__ state,
                                                              uint32_t* __restrict__ ticket) {
  __shared__ uint32_t s_bid, s_wsum[4], s_excl;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) s_bid = atomicAdd(ticket, 1u);
  __syncthreads();
  const uint32_t bid = s_bid;
  constexpr int kPer = (int)(kHeadsItems / 256u);
  const uint32_t i0 = bid * kHeadsItems + (uint32_t)tid * kPer;
  uint32_t k[kPer + 2];  // the element before the thread's first, its kPer elements, the one behind
#pragma unroll
  for (int q = 0; q < kPer + 2; ++q) {
    const uint32_t i = i0 + (uint32_t)q;  // (index of k[q] is i - 1)
    k[q] = (i >= 1u && i - 1u < n) ? keys[i - 1u] : 0xFFFFFFFFu;
  }
  uint32_t f[kPer], tsum = 0;
#pragma unroll
  for (int q = 0; q < kPer; ++q) {
    const uint32_t i = i0 + (uint32_t)q;
    f[q] = (i < n && k[q + 1] != 0xFFFFFFFFu && (i == 0u || k[q + 1] != k[q])) ? 1u : 0u;
    tsum += f[q];
  }
  uint32_t inc = tsum;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t a0 = (uint32_t)__shfl_up((int)inc, d, 64);
    if (lane >= d) inc += a0;
  }
  if (lane == 63) s_wsum[wave] = inc;
  __syncthreads();
  uint32_t wbase = 0;
  for (int w = 0; w < wave; ++w) wbase += s_wsum[w];
  const uint32_t agg = s_wsum[0] + s_wsum[1] + s_wsum[2] + s_wsum[3];
  if (wave == 0) {
    if (lane == 0) __hip_atomic_store(&state[bid], ((bid == 0u ? 2ull : 1ull) << 62) | (unsigned long long)agg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    uint32_t excl = 0;
    int base = (int)bid - 1;
    while (base >= 0) {
      const int j = base - lane;
      unsigned long long rec = 2ull << 62;
      if (j >= 0) {
        do { rec = __hip_atomic_load(&state[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); } while ((rec >> 62) == 0ull);
      }
      const unsigned long long mi = __ballot((rec >> 62) == 2ull);
      const int first = mi ? __ffsll((long long)mi) - 1 : 64;
      uint32_t contrib = lane <= first ? (uint32_t)(rec & 0xFFFFFFFFull) : 0u;
#pragma unroll
      for (int d = 32; d >= 1; d >>= 1) contrib += (uint32_t)__shfl_xor((int)contrib, d, 64);
      excl += contrib;
      if (mi) break;
      base -= 64;
    }
    if (lane == 0) {
      if (bid != 0u) __hip_atomic_store(&state[bid], (2ull << 62) | (unsigned long long)(excl + agg), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      s_excl = excl;
    }
  }
  __syncthreads();
  uint32_t o = s_excl + wbase + inc - tsum;  // exclusive prefix of the flags in front of the thread's first element
#pragma unroll
  for (int q = 0; q < kPer; ++q) {
    const uint32_t i = i0 + (uint32_t)q;
    if (i < n) {
      spts[i] = wpts[vals[i]];
      if (f[q]) heads[o] = i;
      const bool valid = k[q + 1] != 0xFFFFFFFFu;
      if (valid && (i + 1u == n || k[q + 2] == 0xFFFFFFFFu)) { heads[o + f[q]] = i + 1u; *n_cent = o + f[q]; }  // last valid element
      if (i == 0u && !valid) *n_cent = 0u;
      o += f[q];
    }
  }
}

// one thread per leaf: accumulate the leaf's points in (stable-sorted) input order in float, divide by float(count)
// (pcl::CentroidPoint / AccumulatorXYZ semantics); emits the cell key of the centroid for the second sort.
// The sums must run in order (float addition, PCL accumulates in input order), the LOADS need not: a leaf under the
// sensor collects ~1000 scan points, and a dependent key -> index -> point chain per point made this kernel 380 us.
// The points are contiguous now; sixteen are fetched per round trip, the next sixteen while these are added.
__global__ __launch_bounds__(256) void leaf_centroid_kernel(const uint32_t* __restrict__ keys, const uint32_t* __restrict__ heads,
                                                            const uint32_t* __restrict__ n_cent, const float4* __restrict__ spts,
                                                            const MapTouched* __restrict__ ttp, int nc, double inv_cell, float4* __restrict__ cent,
                                                            uint32_t* __restrict__ keys2, uint32_t* __restrict__ vals2,
                                                            uint32_t* __restrict__ long_list, uint32_t* __restrict__ long_count) {
  const MapTouched& tt = *ttp;
  const uint32_t o = blockIdx.x * blockDim.x + threadIdx.x;
  if (o >= *n_cent) return;
  const uint32_t beg = heads[o], end = heads[o + 1];
  const uint32_t key = keys[beg];
  if (long_list && end - beg > kLongLeaf) {  // a whole wavefront takes this one (leaf_centroid_long_kernel)
    const uint32_t at = atomicAdd(long_count, 1u);
    if (at < kMaxLongLeaves) { long_list[at] = o; return; }
  }
  float s0 = 0.f, s1 = 0.f, s2 = 0.f;
  constexpr int B = 16;
  float4 cur[B], nxt[B];
#pragma unroll
  for (int k = 0; k < B; ++k) cur[k] = spts[beg + k < end ? beg + k : end - 1];
  for (uint32_t j = beg; j < end; j += B) {
    const uint32_t jn = j + B;
    if (jn < end) {
#pragma unroll
      for (int k = 0; k < B; ++k) nxt[k] = spts[jn + k < end ? jn + k : end - 1];
    }
#pragma unroll
    for (int k = 0; k < B; ++k)
      if (j + k < end) { s0 += cur[k].x; s1 += cur[k].y; s2 += cur[k].z; }
#pragma unroll
    for (int k = 0; k < B; ++k) cur[k] = nxt[k];
  }
  emit_centroid(o, key, s0, s1, s2, end - beg, tt, nc, inv_cell, cent, keys2, vals2);
}

// leaves with more than kLongLeaf points (the ground right under the sensor: ~1000 scan points in one 0.2 m leaf): one
// WAVEFRONT per leaf loads 64 points per instruction (next 64 in flight) and adds them in order out of its lanes
__global__ __launch_bounds__(256) void leaf_centroid_long_kernel(const uint32_t* __restrict__ keys, const uint32_t* __restrict__ heads,
                                                                 const float4* __restrict__ spts, const MapTouched* __restrict__ ttp, int nc, double inv_cell,
                                                                 float4* __restrict__ cent, uint32_t* __restrict__ keys2,
                                                                 uint32_t* __restrict__ vals2, const uint32_t* __restrict__ long_list,
                                                                 const uint32_t* __restrict__ long_count) {
  const MapTouched& tt = *ttp;
  const uint32_t w = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  const uint32_t n_long = *long_count < kMaxLongLeaves ? *long_count : kMaxLongLeaves;
  if (w >= n_long) return;
  const uint32_t o = long_list[w];
  const uint32_t beg = heads[o], end = heads[o + 1];
  float s0 = 0.f, s1 = 0.f, s2 = 0.f;
  float4 cur = spts[beg + lane < end ? beg + lane : end - 1];
  for (uint32_t j = beg; j < end; j += 64) {
    const uint32_t jn = j + 64;
    float4 nxt = cur;
    if (jn < end) nxt = spts[jn + lane < end ? jn + lane : end - 1];
    // lanes behind the end contribute +0.0f (s + 0.0f == s exactly), so the in-order sum is a fixed, fully unrolled
    // sequence of 64 lane reads -- the same in every lane
    const bool live = j + (uint32_t)lane < end;
    const float x = live ? cur.x : 0.f, y = live ? cur.y : 0.f, z = live ? cur.z : 0.f;
#pragma unroll
    for (int k = 0; k < 64; ++k) {
      s0 += __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(x), k));
      s1 += __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(y), k));
      s2 += __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(z), k));
    }
    cur = nxt;
  }
  if (lane == 0) emit_centroid(o, keys[beg], s0, s1, s2, end - beg, tt, nc, inv_cell, cent, keys2, vals2);
}

// ------------------------------------------------------------------------------------------------
// Leaf grouping WITHOUT a sort (default first stage of an insert).  The radix sort of the whole working set (~550 k
// keys for a 131 k-point scan against the touched cubes: nine launches, ~90 us) only served to bring the points of a leaf
// together in input order.  Most leaves of the touched cubes receive no new point: their old centroid passes through
// unchanged.  So: the NEW points claim the slots of a hash table keyed by leaf (one slot per distinct leaf, counted in
// per wavefront); every OLD point probes the table read-only -- no slot: it is its leaf's only point and becomes the
// centroid directly; a slot: it joins the group.  A scan over the table hands every group a range of a member list, the
// members are placed, and each group is summed in ascending working-set index (old centroid first, then the new points in
// scan order: the order a stable sort would have produced -- float addition is not associative, pcl::VoxelGrid accumulates
// in input order).  Arrival order at a slot is not deterministic, so the members of a group are sorted by index: a
// four-element network per thread, a 64-lane bitonic network per wavefront, an LDS bitonic sort per workgroup for the
// leaves under the sensor (hundreds of points).  Centroid index space: [0, n_old) = the old points (a matched one leaves
// a hole: cell key 0xFFFFFFFF, skipped by the second stage), n_old + g = group g.
struct LeafTable { uint32_t *key, *cnt, *off; uint32_t log2_size; };
constexpr uint32_t kLeafEmpty = 0xFFFFFFFFu, kGiantLeaf = 64, kGiantCap = 4096;

__device__ __forceinline__ uint32_t leaf_hash(uint32_t key, uint32_t log2_size) { return (key * 2654435761u) >> (32 - log2_size); }

// (the new points' part of the working set -- append_new_kernel's job on the sort path -- is produced here as well: one launch less)
__global__ __launch_bounds__(256) void leafhash_insert_new_kernel(const float* __restrict__ xyz, uint32_t n_new, uint32_t stride_floats,
                                                                  const int32_t* __restrict__ cube_of,
                                                                  const MapTouched* __restrict__ ttp, float inv_leaf, int nc, double inv_cell, int rank, int world,
                                                                  float4* __restrict__ wpts, uint32_t* __restrict__ keys, LeafTable ht,
                                                                  uint32_t* __restrict__ mslot, uint32_t* __restrict__ mrank) {
  const MapTouched& tt = *ttp;
  const uint32_t n_old = tt.old_prefix[kMaxTouched];
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t e = n_old + i, total = n_old + n_new;
  uint32_t key = kLeafEmpty;
  if (i < n_new) {
    const float* p = xyz + (size_t)i * stride_floats;
    wpts[e] = make_float4(p[0], p[1], p[2], 0.f);
    const int cube = cube_of[i];
    const int t = cube < 0 ? -1 : touched_index(tt, cube);  // outside the 21x21x11 window (LocalMap.h:605) / a cube of another round: dropped
    if (t >= 0 && !(world > 1 && !shard_keeps_leaf(p[0], p[1], p[2], inv_leaf, tt, t, nc, inv_cell, rank, world)))
      key = leaf_key(p[0], p[1], p[2], inv_leaf, tt.leaf_lo[t][0], tt.leaf_lo[t][1], tt.leaf_lo[t][2], (uint32_t)t, tt.lbits);
    keys[e] = key;
  }
  const bool kept = key != kLeafEmpty;
  const int lane = threadIdx.x & 63;
  // the wavefront's points grouped by key first: one lane per distinct key touches the table, and the members a wavefront
  // adds to a group stay in lane (= scan) order
  uint32_t my_idx = 0, my_cnt = 0;
  int lead = lane;
  unsigned long long todo = __ballot(kept);
  while (todo) {
    const int L = __ffsll((long long)todo) - 1;
    const uint32_t kk = (uint32_t)__builtin_amdgcn_readlane((int)key, L);
    const unsigned long long m = __ballot(kept && key == kk);
    if (kept && key == kk) { my_idx = (uint32_t)__popcll(m & ((1ull << lane) - 1ull)); my_cnt = (uint32_t)__popcll(m); lead = L; }
    todo &= ~m;
  }
  uint32_t slot = kLeafEmpty, base = 0;
  if (kept && lead == lane) {
    const uint32_t mask = (1u << ht.log2_size) - 1u;
    uint32_t h = leaf_hash(key, ht.log2_size);
    for (;;) {
      uint32_t k = ht.key[h];
      if (k == kLeafEmpty) k = atomicCAS(&ht.key[h], kLeafEmpty, key);
      if (k == kLeafEmpty || k == key) break;
      h = (h + 1) & mask;
    }
    slot = h;
    base = atomicAdd(&ht.cnt[slot], my_cnt);
  }
  slot = (uint32_t)__shfl((int)slot, lead, 64);
  base = (uint32_t)__shfl((int)base, lead, 64);
  if (e < total) { mslot[e] = kept ? slot : kLeafEmpty; mrank[e] = base + my_idx; }
}

// (launched after leafhash_insert_new_kernel has completed: the table's keys are final, only the counts still move)
// (the old points' part of the working set -- gather_old_kernel's job on the sort path -- is produced here as well)
__global__ __launch_bounds__(256) void leafhash_match_old_kernel(const float4* __restrict__ pool, uint32_t cap, float inv_leaf, uint32_t* __restrict__ keys,
                                                                 float4* __restrict__ wpts,
                                                                 LeafTable ht, uint32_t* __restrict__ mslot, uint32_t* __restrict__ mrank,
                                                                 const MapTouched* __restrict__ ttp, int nc, double inv_cell, float4* __restrict__ cent,
                                                                 uint32_t* __restrict__ keys2, const CellCount cc) {
  const MapTouched& tt = *ttp;
  // (grid-stride: a device-built round knows the number of old points, the host that sized the launch only an estimate; the
  //  trip count is the same for every lane of a workgroup: the ballots of the counting below need whole wavefronts)
  const uint32_t n_old = tt.old_prefix[kMaxTouched];
  const int lane = threadIdx.x & 63;
  for (uint32_t first = blockIdx.x * blockDim.x; first < n_old; first += gridDim.x * blockDim.x) {
    const uint32_t e = first + threadIdx.x;
    uint32_t k2 = kLeafEmpty;  // cell key of a point that passes through (stays empty for a point that joins a group)
    if (e < n_old) {
      int t = 0;
#pragma unroll
      for (int step = 16; step >= 1; step >>= 1) t = (t + step < tt.n && tt.old_prefix[t + step] <= e) ? t + step : t;
      const float4 p = pool[(size_t)tt.slot[t] * cap + (e - tt.old_prefix[t])];
      const uint32_t key = leaf_key(p.x, p.y, p.z, inv_leaf, tt.leaf_lo[t][0], tt.leaf_lo[t][1], tt.leaf_lo[t][2], (uint32_t)t, tt.lbits);
      wpts[e] = p; keys[e] = key;
      const uint32_t mask = (1u << ht.log2_size) - 1u;
      uint32_t h = leaf_hash(key, ht.log2_size), slot = kLeafEmpty;
      for (;;) {
        const uint32_t k = ht.key[h];
        if (k == key) { slot = h; break; }
        if (k == kLeafEmpty) break;
        h = (h + 1) & mask;
      }
      mslot[e] = slot;
      if (slot != kLeafEmpty) {
        mrank[e] = atomicAdd(&ht.cnt[slot], 1u);
        keys2[e] = kLeafEmpty;  // a hole of the centroid index space: the point lives on in its group
      } else {
        // the only point of its leaf: sum = 0 + p, count = 1
        k2 = emit_centroid(e, key, 0.f + p.x, 0.f + p.y, 0.f + p.z, 1u, tt, nc, inv_cell, cent, keys2, nullptr);
      }
    }
    if (cc.grid) {
      // cell_count_kernel's counting, here: one atomic per DISTINCT cell of the wavefront (the old points come in pool
      // order, cell after cell: the lanes hit two or three counters, and 64 atomics on one word serialise).  The rank goes
      // where a matched point keeps its member rank (mrank == cc.rank: a point is one or the other)
      const bool kept = k2 != kLeafEmpty;
      uint32_t my_idx = 0, my_cnt = 0;
      int lead = lane;
      unsigned long long todo = __ballot(kept);
      while (todo) {
        const int L = __ffsll((long long)todo) - 1;
        const uint32_t kk = (uint32_t)__builtin_amdgcn_readlane((int)k2, L);
        const unsigned long long m = __ballot(kept && k2 == kk);
        if (kept && k2 == kk) { my_idx = (uint32_t)__popcll(m & ((1ull << lane) - 1ull)); my_cnt = (uint32_t)__popcll(m); lead = L; }
        todo &= ~m;
      }
      uint32_t base = 0;
      if (kept && lead == lane) base = atomicAdd(&cc.grid[(size_t)(k2 >> 18) * cc.ncell1 + (k2 & 0x3FFFFu)], my_cnt);
      base = (uint32_t)__shfl((int)base, lead, 64);
      if (kept) cc.rank[e] = base + my_idx;
    }
  }
}

// member-list ranges + group ordinals from the table counts (four slots per thread, 1024 threads per workgroup: workgroup
// scan, ONE packed atomic per workgroup: members in the low word, groups in the high word -- the ranges only have to be
// disjoint, not ordered); leaves the table empty for the next insert.  The groups of more than 16 / more than kGiantLeaf
// members are listed here (their sizes are known): the kernels that sum them need not wait for the one that sums the rest.
__global__ __launch_bounds__(1024) void leafhash_offsets_kernel(LeafTable ht, uint32_t* __restrict__ gstart, uint32_t* __restrict__ gcount,
                                                                unsigned long long* __restrict__ cursor,
                                                                uint32_t* __restrict__ medium_list, uint32_t* __restrict__ medium_count,
                                                                uint32_t* __restrict__ giant_list, uint32_t* __restrict__ giant_count) {
  __shared__ uint32_t wm[16], wg[16], wl[16], base_m, base_g, base_med, base_gia;
  const uint32_t t4 = blockIdx.x * blockDim.x + threadIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint4 c4 = reinterpret_cast<const uint4*>(ht.cnt)[t4];
  const uint32_t cnt[4] = {c4.x, c4.y, c4.z, c4.w};
  uint32_t tm = 0, tg = 0, tl = 0;  // members, groups, listed groups (medium in the low half, giant in the high half)
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    tm += cnt[k]; tg += cnt[k] ? 1u : 0u;
    tl += cnt[k] > kGiantLeaf ? 0x10000u : (cnt[k] > 16u ? 1u : 0u);
  }
  uint32_t im = tm, ig = tg, il = tl;
  if (__ballot(tm != 0)) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const uint32_t a = (uint32_t)__shfl_up((int)im, d, 64), b = (uint32_t)__shfl_up((int)ig, d, 64), c = (uint32_t)__shfl_up((int)il, d, 64);
      if (lane >= d) { im += a; ig += b; il += c; }
    }
  }
  if (lane == 63) { wm[wave] = im; wg[wave] = ig; wl[wave] = il; }
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t sm = 0, sg = 0, sl = 0;
    for (int w = 0; w < 16; ++w) { const uint32_t a = wm[w], b = wg[w], c = wl[w]; wm[w] = sm; wg[w] = sg; wl[w] = sl; sm += a; sg += b; sl += c; }
    unsigned long long old = 0ull;
    if (sm) old = atomicAdd(cursor, (unsigned long long)sm | ((unsigned long long)sg << 32));
    base_m = (uint32_t)old; base_g = (uint32_t)(old >> 32);
    // (one atomic per workgroup and list: every group appending for itself took 11 us on the two counters)
    base_med = (sl & 0xFFFFu) ? atomicAdd(medium_count, sl & 0xFFFFu) : 0u;
    base_gia = (sl >> 16) ? atomicAdd(giant_count, sl >> 16) : 0u;
  }
  __syncthreads();
  if (tm) {
    uint32_t off = base_m + wm[wave] + (im - tm), g = base_g + wg[wave] + (ig - tg);
    const uint32_t lb = wl[wave] + (il - tl);
    uint32_t pm = base_med + (lb & 0xFFFFu), pg = base_gia + (lb >> 16);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (!cnt[k]) continue;
      ht.off[4 * t4 + k] = off;
      gstart[g] = off; gcount[g] = cnt[k];
      if (cnt[k] > kGiantLeaf) giant_list[pg++] = g;
      else if (cnt[k] > 16u) medium_list[pm++] = g;
      off += cnt[k]; ++g;
      ht.key[4 * t4 + k] = kLeafEmpty;
    }
    reinterpret_cast<uint4*>(ht.cnt)[t4] = make_uint4(0, 0, 0, 0);
  }
}

__global__ __launch_bounds__(256) void leafhash_place_kernel(const uint32_t* __restrict__ mslot, const uint32_t* __restrict__ mrank, uint32_t n_new,
                                                             const MapTouched* __restrict__ ttp, const uint32_t* __restrict__ off, uint32_t* __restrict__ members) {
  const uint32_t total = ttp->old_prefix[kMaxTouched] + n_new;
  for (uint32_t e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
    const uint32_t sl = mslot[e];
    if (sl != kLeafEmpty) members[off[sl] + mrank[e]] = e;
  }
}

// Groups of up to 16 members: one thread per group (sorting network in registers): 95 % of the groups of a raw 128-beam
// sweep.  Groups of 17..64 (leafhash_offsets_kernel's medium list): one wavefront each (bitonic network over the lanes,
// then the sum in lane order).  ONE launch: workgroups [0, small_blocks) take the first kind, the others the second.
__device__ __forceinline__ void leafhash_small_groups(uint32_t g, const uint32_t* __restrict__ gstart, const uint32_t* __restrict__ gcount,
                                                      const unsigned long long* __restrict__ cursor, const uint32_t* __restrict__ members,
                                                      const float4* __restrict__ wpts, const uint32_t* __restrict__ leaf_keys,
                                                      uint32_t n_old, const MapTouched& tt, int nc, double inv_cell, float4* __restrict__ cent,
                                                      uint32_t* __restrict__ keys2, uint32_t* __restrict__ n_cent, const CellCount cc) {
  const uint32_t n_groups = (uint32_t)(*cursor >> 32);
  if (g == 0) *n_cent = n_old + n_groups;
  const bool have = g < n_groups;
  const uint32_t cnt = have ? gcount[g] : 0u, beg = have ? gstart[g] : 0u;
  if (!have || cnt > 16u) return;
  uint32_t e[16];
  if (cnt <= 4u) {
#pragma unroll
    for (int k = 0; k < 4; ++k) e[k] = (uint32_t)k < cnt ? members[beg + k] : kLeafEmpty;
    uint32_t t;
#define SO_CSWAP(a, b) { t = min(a, b); b = max(a, b); a = t; }
    SO_CSWAP(e[0], e[1]) SO_CSWAP(e[2], e[3]) SO_CSWAP(e[0], e[2]) SO_CSWAP(e[1], e[3]) SO_CSWAP(e[1], e[2])
#undef SO_CSWAP
    float s0 = 0.f, s1 = 0.f, s2 = 0.f;
    float4 p[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) p[k] = wpts[(uint32_t)k < cnt ? e[k] : e[0]];
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if ((uint32_t)k < cnt) { s0 += p[k].x; s1 += p[k].y; s2 += p[k].z; }
    emit_centroid(n_old + g, leaf_keys[e[0]], s0, s1, s2, cnt, tt, nc, inv_cell, cent, keys2, nullptr, cc);
    return;
  }
#pragma unroll
  for (int k = 0; k < 16; ++k) e[k] = (uint32_t)k < cnt ? members[beg + k] : kLeafEmpty;
  // bitonic network over 16 registers (absent members = 0xFFFFFFFF end up behind the real ones)
#pragma unroll
  for (int k = 2; k <= 16; k <<= 1) {
#pragma unroll
    for (int j = k >> 1; j > 0; j >>= 1) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int l = i ^ j;
        if (l > i) {
          const uint32_t lo = min(e[i], e[l]), hi = max(e[i], e[l]);
          const bool up = (i & k) == 0;
          e[i] = up ? lo : hi; e[l] = up ? hi : lo;
        }
      }
    }
  }
  float4 p[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) p[k] = wpts[(uint32_t)k < cnt ? e[k] : e[0]];
  float s0 = 0.f, s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int k = 0; k < 16; ++k)
    if ((uint32_t)k < cnt) { s0 += p[k].x; s1 += p[k].y; s2 += p[k].z; }
  emit_centroid(n_old + g, leaf_keys[e[0]], s0, s1, s2, cnt, tt, nc, inv_cell, cent, keys2, nullptr, cc);
}

__device__ __forceinline__ void leafhash_medium_groups(uint32_t first_wave, uint32_t n_waves, int lane, const uint32_t* __restrict__ gstart,
                                                       const uint32_t* __restrict__ gcount, const uint32_t* __restrict__ members,
                                                       const float4* __restrict__ wpts, const uint32_t* __restrict__ leaf_keys, uint32_t n_old,
                                                       const MapTouched& tt, int nc, double inv_cell, float4* __restrict__ cent,
                                                       uint32_t* __restrict__ keys2, const uint32_t* __restrict__ medium_list,
                                                       const uint32_t* __restrict__ medium_count, const CellCount cc) {
  const uint32_t n_medium = *medium_count;
  for (uint32_t w = first_wave; w < n_medium; w += n_waves) {
    const uint32_t g = medium_list[w];
    const uint32_t c = gcount[g], b = gstart[g];
    uint32_t v = (uint32_t)lane < c ? members[b + lane] : kLeafEmpty;
#pragma unroll
    for (int k = 2; k <= 64; k <<= 1) {
#pragma unroll
      for (int j = k >> 1; j > 0; j >>= 1) {
        const uint32_t o = (uint32_t)__shfl_xor((int)v, j, 64);
        const bool keep_min = ((lane & j) == 0) == ((lane & k) == 0);
        v = keep_min ? min(v, o) : max(v, o);
      }
    }
    // lane i now holds the group's i-th member in working-set order; lanes behind the end contribute +0.0f (s + 0.0f == s)
    const float4 p = wpts[(uint32_t)lane < c ? v : 0u];
    const bool live = (uint32_t)lane < c;
    const float x = live ? p.x : 0.f, y = live ? p.y : 0.f, z = live ? p.z : 0.f;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int k = 0; k < 64; ++k) {
      s0 += __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(x), k));
      s1 += __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(y), k));
      s2 += __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(z), k));
    }
    const uint32_t first = (uint32_t)__builtin_amdgcn_readlane((int)v, 0);
    if (lane == 0) emit_centroid(n_old + g, leaf_keys[first], s0, s1, s2, c, tt, nc, inv_cell, cent, keys2, nullptr, cc);
  }
}

// leaves with more than 64 points (the ground under the sensor: up to ~1 600 points of a raw 128-beam sweep in one 0.2 m
// leaf): one workgroup per leaf.  The member list is a sequence of RUNS in arrival order -- the points one wavefront of
// leafhash_insert_new_kernel added (contiguous, in scan order) -- plus the odd old point.  A new point's place in
// working-set order follows from its wavefront: members are counted per wavefront (a direct-address table in LDS, one
// entry per 64 scan points), a scan turns the counts into run positions, and inside its run a member sits where it
// sits in the list.  (A 2 048-element bitonic sort in LDS took 150 us here, ranking the runs against each other 35 us.)
// The points are gathered into LDS in final order and three wavefronts add up x, y and z in sequence, 64 values per
// round out of their lanes.  More members than kGiantCap, more than 64 x kGiantCap new points in the round, or more than
// 64 old points in one leaf: *overflow is raised, the second stage stands still and the host repeats the round with the
// sort-based first stage.
constexpr int kGiantThreads = 1024;                       // four members per thread at most: one round trip per phase
// dynamic LDS: x, y, z of up to kGiantCap members + per-wavefront counts and first list positions, one entry per wavefront of
// leafhash_insert_new_kernel rounded up to the workgroup size (64 KB for a 131 072-point sweep: two workgroups per CU)
static inline uint32_t giant_hist_entries(uint32_t n_new) {
  const uint32_t nb = (n_new + 63u) >> 6;
  const uint32_t r = (nb + kGiantThreads - 1u) / kGiantThreads * kGiantThreads;
  return r < (uint32_t)kGiantThreads ? (uint32_t)kGiantThreads : (r > kGiantCap ? kGiantCap : r);
}
static inline size_t giant_lds_bytes(uint32_t n_new) { return ((size_t)kGiantCap * 3 + (size_t)giant_hist_entries(n_new) * 2) * 4; }
__device__ __forceinline__ void leafhash_giant_groups(uint32_t first_leaf, uint32_t leaf_stride, const uint32_t* __restrict__ gstart,
                                                      const uint32_t* __restrict__ gcount, const uint32_t* __restrict__ members,
                                                      const float4* __restrict__ wpts, const uint32_t* __restrict__ leaf_keys, uint32_t n_old,
                                                      uint32_t n_new, uint32_t nhist, const MapTouched& tt, int nc, double inv_cell,
                                                      float4* __restrict__ cent, uint32_t* __restrict__ keys2, const uint32_t* __restrict__ giant_list,
                                                      const uint32_t* __restrict__ giant_count, uint32_t* __restrict__ overflow, const CellCount cc) {
  extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
  float* sx = reinterpret_cast<float*>(lds);
  float* sy = reinterpret_cast<float*>(lds + kGiantCap);
  float* sz = reinterpret_cast<float*>(lds + 2 * kGiantCap);
  uint32_t* hist = lds + 3 * kGiantCap;   // [nhist]: nhist = giant_hist_entries(n_new), a multiple of the workgroup size
  uint32_t* fpos = hist + nhist;
  __shared__ uint32_t wtot[kGiantThreads / 64], sums[3], olds[64], n_olds, first_e;
  const int tid = threadIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  constexpr int kPer = kGiantCap / kGiantThreads;  // 4
  const uint32_t hper = nhist / (uint32_t)kGiantThreads;  // 1..4 table entries per thread
  const uint32_t n_giant = *giant_count;
  const uint32_t nb = (n_new + 63u) >> 6;  // wavefronts of leafhash_insert_new_kernel
  for (uint32_t w = first_leaf; w < n_giant; w += leaf_stride) {
    const uint32_t g = giant_list[w];
    const uint32_t cnt = gcount[g], beg = gstart[g];
    if (cnt > kGiantCap || nb > nhist) { if (tid == 0) *overflow = 1u; continue; }
    __syncthreads();  // (the previous leaf of this workgroup is done with the arrays)
    uint32_t e[kPer];
    float4 p[kPer];
#pragma unroll
    for (int k = 0; k < kPer; ++k) {
      const uint32_t i = (uint32_t)tid + (uint32_t)k * kGiantThreads;
      e[k] = i < cnt ? members[beg + i] : kLeafEmpty;
    }
    for (uint32_t k = 0; k < hper; ++k) { hist[(uint32_t)tid + k * kGiantThreads] = 0u; fpos[(uint32_t)tid + k * kGiantThreads] = 0xFFFFFFFFu; }
    if (tid == 0) n_olds = 0u;
#pragma unroll
    for (int k = 0; k < kPer; ++k) p[k] = wpts[e[k] != kLeafEmpty ? e[k] : 0u];  // (on their way while the places are worked out)
    __syncthreads();
#pragma unroll
    for (int k = 0; k < kPer; ++k) {
      const uint32_t i = (uint32_t)tid + (uint32_t)k * kGiantThreads;
      if (e[k] == kLeafEmpty) continue;
      if (e[k] >= n_old) { const uint32_t b = (e[k] - n_old) >> 6; atomicAdd(&hist[b], 1u); atomicMin(&fpos[b], i); }
      else { const uint32_t at = atomicAdd(&n_olds, 1u); if (at < 64u) olds[at] = e[k]; }
    }
    __syncthreads();
    const uint32_t n_o = n_olds;
    if (n_o > 64u) { if (tid == 0) *overflow = 1u; continue; }  // (uniform: every thread read the same n_olds)
    // exclusive scan of the per-wavefront counts: hper consecutive entries per thread, a shuffle scan over the wavefront's
    // threads, the sixteen wavefront totals through LDS (two barriers; twenty with a workgroup-wide Hillis-Steele)
    uint32_t local[kPer], sum = 0;
#pragma unroll
    for (int k = 0; k < kPer; ++k) { local[k] = sum; sum += (uint32_t)k < hper ? hist[hper * (uint32_t)tid + k] : 0u; }
    uint32_t inc = sum;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const uint32_t a0 = (uint32_t)__shfl_up((int)inc, d, 64);
      if (lane >= d) inc += a0;
    }
    if (lane == 63) wtot[wave] = inc;
    __syncthreads();
    uint32_t before = inc - sum;
    for (int q = 0; q < wave; ++q) before += wtot[q];
#pragma unroll
    for (int k = 0; k < kPer; ++k)
      if ((uint32_t)k < hper) hist[hper * (uint32_t)tid + k] = before + local[k];
    __syncthreads();
    // points into their final places: the old points first (in index order), then wavefront after wavefront; inside its
    // wavefront's run a member keeps its distance from the run's first list position
#pragma unroll
    for (int k = 0; k < kPer; ++k) {
      const uint32_t i = (uint32_t)tid + (uint32_t)k * kGiantThreads;
      if (e[k] == kLeafEmpty) continue;
      uint32_t at;
      if (e[k] < n_old) {
        at = 0;
        for (uint32_t q = 0; q < n_o; ++q) at += olds[q] < e[k] ? 1u : 0u;
      } else {
        const uint32_t b = (e[k] - n_old) >> 6;
        at = n_o + hist[b] + (i - fpos[b]);
      }
      sx[at] = p[k].x; sy[at] = p[k].y; sz[at] = p[k].z;
      if (at == 0) first_e = e[k];  // the first member in working-set order (its leaf key goes with the centroid)
    }
    __syncthreads();
    if (tid < 192) {
      // wavefronts 0, 1, 2 add x, y, z: three independent chains of dependent additions, one addition per element.  Every
      // lane reads the SAME sixteen values from LDS (a broadcast, no bank conflict) and adds them in order -- the operands
      // arrive in the lane's own registers, the next sixteen are on their way meanwhile.  (Handing the elements of one
      // register around with v_readlane cost readlane + wait state + add per element: 28 us for a leaf of 1 636 points.)
      const float* src = tid < 64 ? sx : (tid < 128 ? sy : sz);
      float acc = 0.f;
      const uint32_t n16 = cnt & ~15u;
      float4 bufa[4], bufb[4];  // two register sets taking turns (no copies between them)
#pragma unroll
      for (int q = 0; q < 4; ++q) bufa[q] = n16 ? *reinterpret_cast<const float4*>(src + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
      for (uint32_t j = 0; j < n16; j += 32) {
        if (j + 16 < n16) {
#pragma unroll
          for (int q = 0; q < 4; ++q) bufb[q] = *reinterpret_cast<const float4*>(src + j + 16 + 4 * q);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) { acc += bufa[q].x; acc += bufa[q].y; acc += bufa[q].z; acc += bufa[q].w; }
        if (j + 16 < n16) {
          if (j + 32 < n16) {
#pragma unroll
            for (int q = 0; q < 4; ++q) bufa[q] = *reinterpret_cast<const float4*>(src + j + 32 + 4 * q);
          }
#pragma unroll
          for (int q = 0; q < 4; ++q) { acc += bufb[q].x; acc += bufb[q].y; acc += bufb[q].z; acc += bufb[q].w; }
        }
      }
      for (uint32_t j = n16; j < cnt; ++j) acc += src[j];
      if (lane == 0) sums[tid >> 6] = __float_as_uint(acc);
    }
    __syncthreads();
    if (tid == 0)
      emit_centroid(n_old + g, leaf_keys[first_e], __uint_as_float(sums[0]), __uint_as_float(sums[1]), __uint_as_float(sums[2]), cnt, tt, nc, inv_cell,
                    cent, keys2, nullptr, cc);
  }
}

// The centroids of every group in ONE launch of 1 024-thread workgroups: workgroups [0, giant_blocks) walk the list of
// leaves with more than 64 members (one leaf at a time each), the next small_blocks take one group of up to 16 members per
// thread, the rest one group of 17..64 members per wavefront.  The few long leaves of a raw sweep (its slowest took 23 us)
// run beside the ten thousand short ones instead of behind them; workgroups of the first kind without a leaf end at once.
__global__ __launch_bounds__(kGiantThreads) void leafhash_centroids_kernel(const uint32_t* __restrict__ gstart, const uint32_t* __restrict__ gcount,
                                                                 const unsigned long long* __restrict__ cursor, const uint32_t* __restrict__ members,
                                                                 const float4* __restrict__ wpts, const uint32_t* __restrict__ leaf_keys,
                                                                 uint32_t n_new, uint32_t nhist, const MapTouched* __restrict__ ttp, int nc,
                                                                 double inv_cell, float4* __restrict__ cent, uint32_t* __restrict__ keys2,
                                                                 uint32_t* __restrict__ n_cent, const uint32_t* __restrict__ medium_list,
                                                                 const uint32_t* __restrict__ medium_count, const uint32_t* __restrict__ giant_list,
                                                                 const uint32_t* __restrict__ giant_count, uint32_t* __restrict__ overflow,
                                                                 uint32_t giant_blocks, uint32_t small_blocks, const CellCount cc) {
  const MapTouched& tt = *ttp;
  const uint32_t n_old = tt.old_prefix[kMaxTouched];
  if (blockIdx.x < giant_blocks)
    leafhash_giant_groups(blockIdx.x, giant_blocks, gstart, gcount, members, wpts, leaf_keys, n_old, n_new, nhist, tt, nc, inv_cell, cent, keys2, giant_list,
                          giant_count, overflow, cc);
  else if (blockIdx.x < giant_blocks + small_blocks) {
    // (four of the sixteen wavefronts work, the others end at once: 256 groups per workgroup spread the gathers over all the
    //  compute units -- with 1 024 groups per workgroup half of them sat idle and this part took 19 us instead of 11)
    if (threadIdx.x < 256u)
      leafhash_small_groups((blockIdx.x - giant_blocks) * 256u + threadIdx.x, gstart, gcount, cursor, members, wpts, leaf_keys, n_old, tt, nc, inv_cell,
                            cent, keys2, n_cent, cc);
  }
  else
    leafhash_medium_groups(((blockIdx.x - giant_blocks - small_blocks) * blockDim.x + threadIdx.x) >> 6,
                           ((gridDim.x - giant_blocks - small_blocks) * blockDim.x) >> 6, threadIdx.x & 63, gstart, gcount, members, wpts, leaf_keys, n_old, tt,
                           nc, inv_cell, cent, keys2, medium_list, medium_count, cc);
}

__global__ __launch_bounds__(256) void gather_export_kernel(const float4* __restrict__ pool, uint32_t cap, uint32_t slot, uint32_t count,
                                                            float* __restrict__ out_xyz) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  const float4 p = pool[(size_t)slot * cap + i];
  out_xyz[3 * i] = p.x; out_xyz[3 * i + 1] = p.y; out_xyz[3 * i + 2] = p.z;
}

// the same gather into RECORDS of stride_words 32-bit words -- float x, y, z at words 0..2, the rest zero (pcl::PointXYZI: 8 words, intensity 0):
// the payload of a sensor_msgs/PointCloud2 as the node publishes its map clouds (laserMapping.cpp:437-462), written where the message is
__global__ __launch_bounds__(256) void gather_export_records_kernel(const float4* __restrict__ pool, uint32_t cap, uint32_t slot, uint32_t count,
                                                                    uint32_t* __restrict__ out_words, uint32_t stride_words) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;  // one thread per output WORD: coalesced stores whatever the record size
  const uint32_t i = t / stride_words, w = t - i * stride_words;
  if (i >= count) return;
  uint32_t v = 0u;
  if (w < 3u) {
    const float4 p = pool[(size_t)slot * cap + i];
    v = __float_as_uint(w == 0u ? p.x : (w == 1u ? p.y : p.z));
  }
  out_words[t] = v;
}

// ------------------------------------------------------------------------------------------------------------------
static inline dim3 grid_for(uint32_t n, int block) { return dim3((n + block - 1) / block); }

// ------------------------------------------------------------------------------------------------
// Second stage without a sort: the centroids are counted into the dense cell grids of the touched cubes (atomic rank
// inside a cell), an exclusive scan over the grids IS the cubes' new cell_start tables, the centroids are placed, and
// every cell with more than one point is put into ascending leaf order (one centroid per leaf: the order is total, so the
// result equals the stable sort by (cell, leaf) it replaces).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void cell_count_kernel(const uint32_t* __restrict__ keys2, const uint32_t* __restrict__ n_cent,
                                                         uint32_t ncell1, uint32_t* __restrict__ grid, uint32_t* __restrict__ rank,
                                                         const uint32_t* __restrict__ halt) {
  if (*halt) return;
  const uint32_t n_c = *n_cent;
  const int lane = threadIdx.x & 63;
  // (grid-stride, the trip count the same for every lane of a workgroup: the ballots below need whole wavefronts)
  for (uint32_t first = blockIdx.x * blockDim.x; first < n_c; first += gridDim.x * blockDim.x) {
    const uint32_t o = first + threadIdx.x;
    // hole of the centroid index space (hash grouping: an old point that joined a group): key 0xFFFFFFFF
    const uint32_t k = o < n_c ? keys2[o] : 0xFFFFFFFFu;
    const bool kept = k != 0xFFFFFFFFu;
    // one atomic per DISTINCT cell of the wavefront: with the old points in pool order (cell after cell) the lanes of a
    // wavefront hit two or three counters, and 64 atomics on one word serialise
    uint32_t my_idx = 0, my_cnt = 0;
    int lead = lane;
    unsigned long long todo = __ballot(kept);
    while (todo) {
      const int L = __ffsll((long long)todo) - 1;
      const uint32_t kk = (uint32_t)__builtin_amdgcn_readlane((int)k, L);
      const unsigned long long m = __ballot(kept && k == kk);
      if (kept && k == kk) { my_idx = (uint32_t)__popcll(m & ((1ull << lane) - 1ull)); my_cnt = (uint32_t)__popcll(m); lead = L; }
      todo &= ~m;
    }
    uint32_t base = 0;
    if (kept && lead == lane) base = atomicAdd(&grid[(size_t)(k >> 18) * ncell1 + (k & 0x3FFFFu)], my_cnt);
    base = (uint32_t)__shfl((int)base, lead, 64);
    if (kept) rank[o] = base + my_idx;
  }
}
// grid_scan = exclusive scan of grid over all touched cubes (one entry more than cells: the total); per cube: table entry =
// slot*cap + (scan - scan at the cube's first cell); the entry behind the last cell = the cube's new point count.
// The counters have done their work once they are scanned: this launch, one thread per cell, puts them back to zero, so
// that the next insert finds the grids clean (a 12 MB fill per insert otherwise).
__global__ __launch_bounds__(256) void cell_table_kernel(const uint32_t* __restrict__ grid_scan, const MapTouched* __restrict__ ttp, uint32_t cap, uint32_t ncell1,
                                                         uint32_t* __restrict__ cell_start, uint32_t* __restrict__ counts,
                                                         const uint32_t* __restrict__ halt, uint32_t* __restrict__ grid) {
  const MapTouched& tt = *ttp;
  const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t t = blockIdx.y;
  if (c >= ncell1 || *halt) return;  // (halted before the counting: the grids are still clean)
  grid[(size_t)t * ncell1 + c] = 0u;
  const uint32_t local = grid_scan[(size_t)t * ncell1 + c] - grid_scan[(size_t)t * ncell1];
  cell_start[(size_t)tt.slot[t] * ncell1 + c] = tt.slot[t] * cap + local;
  if (c == ncell1 - 1) counts[t] = local;
}
// pass 1: every centroid into its cell's range of a scratch array, at the atomic rank, leaf key in .w
__global__ __launch_bounds__(256) void cell_place_kernel(const uint32_t* __restrict__ keys2, const uint32_t* __restrict__ rank,
                                                         const uint32_t* __restrict__ n_cent, const uint32_t* __restrict__ grid_scan,
                                                         const float4* __restrict__ cent, const MapTouched* __restrict__ ttp, uint32_t ncell1, float inv_leaf,
                                                         float4* __restrict__ tmp, uint32_t* __restrict__ tmpk,
                                                         const uint32_t* __restrict__ halt) {
  const MapTouched& tt = *ttp;
  if (*halt) return;
  const uint32_t n_c = *n_cent;
  for (uint32_t o = blockIdx.x * blockDim.x + threadIdx.x; o < n_c; o += gridDim.x * blockDim.x) {
    const uint32_t k = keys2[o], t = k >> 18;
    if (k == 0xFFFFFFFFu) continue;
    const float4 v = cent[o];
    const uint32_t at = grid_scan[(size_t)t * ncell1 + (k & 0x3FFFFu)] + rank[o];  // position over all touched cubes
    tmp[at] = v;
    tmpk[at] = leaf_key(v.x, v.y, v.z, inv_leaf, tt.leaf_lo[t][0], tt.leaf_lo[t][1], tt.leaf_lo[t][2], t, tt.lbits);  // (the ranking pass reads 4 bytes per comparison)
  }
}
// pass 2: final position inside the cell = number of the cell's centroids with a smaller leaf key (one centroid per leaf:
// the keys are distinct), i.e. ascending leaf order -- what the stable sort by (cell, leaf) produced
// (After a resolution change -- launch_map_retable -- a cube may hold points of an older, different leaf grid, where two
//  points can share a leaf key: the order then falls back on the coordinates' bit patterns and, for bitwise equal points,
//  on the placement rank, so that every point keeps a position of its own.)
__global__ __launch_bounds__(256) void cell_rank_kernel(const uint32_t* __restrict__ keys2, const uint32_t* __restrict__ n_cent,
                                                        const uint32_t* __restrict__ grid_scan,
                                                        const float4* __restrict__ cent, const float4* __restrict__ tmp,
                                                        const uint32_t* __restrict__ tmpk, const MapTouched* __restrict__ ttp,
                                                        uint32_t cap, uint32_t ncell1, float inv_leaf, float4* __restrict__ pool,
                                                        const uint32_t* __restrict__ rank, const uint32_t* __restrict__ halt) {
  const MapTouched& tt = *ttp;
  if (*halt) return;
  const uint32_t n_c = *n_cent;
  for (uint32_t o = blockIdx.x * blockDim.x + threadIdx.x; o < n_c; o += gridDim.x * blockDim.x) {
    const uint32_t k = keys2[o], t = k >> 18;
    if (k == 0xFFFFFFFFu) continue;
    const size_t gi = (size_t)t * ncell1 + (k & 0x3FFFFu);
    const uint32_t beg = grid_scan[gi], cnt = grid_scan[gi + 1] - beg;  // (the scan has one entry behind the last cell)
    const float4 v = cent[o];
    const uint32_t kv = leaf_key(v.x, v.y, v.z, inv_leaf, tt.leaf_lo[t][0], tt.leaf_lo[t][1], tt.leaf_lo[t][2], t, tt.lbits);
    const uint32_t mine = rank[o];
    uint32_t r = 0;
    for (uint32_t j = 0; j < cnt; ++j) {
      const uint32_t ku = tmpk[beg + j];
      bool less = ku < kv;
      if (ku == kv && j != mine) {
        const float4 u = tmp[beg + j];
        const uint32_t ux = __float_as_uint(u.x), uy = __float_as_uint(u.y), uz = __float_as_uint(u.z);
        const uint32_t vx = __float_as_uint(v.x), vy = __float_as_uint(v.y), vz = __float_as_uint(v.z);
        less = uz != vz ? uz < vz : (uy != vy ? uy < vy : (ux != vx ? ux < vx : j < mine));
      }
      r += less ? 1u : 0u;
    }
    const uint32_t local = beg - grid_scan[(size_t)t * ncell1] + r;
    if (local < cap) pool[(size_t)tt.slot[t] * cap + local] = v;
  }
}


// ------------------------------------------------------------------------------------------------
// The insert without a host round trip (DeviceMap::insert_fast).  What the host used to do between world_cube_kernel and
// the first stage -- read the touched flags back, list the touched cubes, look up their slots and point counts, lay out
// the round (MapTouched) -- is done by the LAST workgroup of the front kernel from tables the device keeps itself
// (cube -> slot: the k-NN's table; points per slot: written by the scan of every round; "one point per leaf of the current
// grid" per slot: written by the report of every round).  A round the device cannot lay out (a cube without a slot, more
// cubes than a round holds, a cube that must go through the sort) raises *halt with an EMPTY round: every later kernel
// of the insert is then a no-op, the map is unchanged, and the host -- told by the report -- repeats the insert round by round.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void tt_store_kernel(MapTouched tt, MapTouched* __restrict__ out) {
  if (threadIdx.x == 0) *out = tt;
}

struct FrontBuild {
  const int32_t* cube_slot; const uint32_t* slot_count; const uint32_t* slot_ok;
  uint32_t* cube_cnt; uint32_t* n_inside; uint32_t* ticket; uint32_t* halt; uint32_t* dirty; MapTouched* tt;
  float inv_leaf; uint32_t lbits; int32_t per_round;
  MapFastReport* rep; unsigned long long seq;
  uint32_t* touched_n; int32_t* touched_list;  // the cubes that received points, in the order their counters left zero (kTouchedCap entries)
};
constexpr uint32_t kTouchedCap = 64;
// new points of a cube into its counter; whoever moves the counter off zero lists the cube.  Every access is a device-scope
// read-modify-write whose result is consumed: performed before the thread goes on (see the ticket below)
__device__ __forceinline__ void front_count(const FrontBuild& b, int c, uint32_t add) {
  const uint32_t before = __hip_atomic_fetch_add(&b.cube_cnt[c], add, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (before == 0u) {
    const uint32_t pos = __hip_atomic_fetch_add(b.touched_n, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (pos < kTouchedCap) {
      const int32_t was = __hip_atomic_exchange(&b.touched_list[pos], (int32_t)c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      asm volatile("" ::"v"(was));
    }
  }
}
static_assert(kMapW == 21 && kMapH == 21 && kMapD == 11, "cube index arithmetic below");

// transformAndAddToMap's transform (LidarSlam.cpp:60-80; TransformPoint, superodom_utils.h:119-123) + LocalMap.h:596-610 in
// one pass over the scan: world point (TRANSFORM), cube of every point, new points per cube; then the round (see above)
template <bool TRANSFORM>
__global__ __launch_bounds__(1024) void insert_front_kernel(const float* __restrict__ in, uint32_t n, uint32_t stride_floats, Pose pose,
                                                            float* __restrict__ world, int o0, int o1, int o2, int32_t* __restrict__ cube_of,
                                                            FrontBuild b) {
  __shared__ uint32_t cnt;
  __shared__ bool last;
  // the workgroup's own tally of (cube, new points) first -- a sweep touches a handful of cubes, and 2 048 wavefronts adding
  // to the same few words of memory took 60 us --, then one atomic per distinct cube and workgroup
  constexpr int kTally = 8;
  __shared__ int32_t s_tkey[kTally];
  __shared__ uint32_t s_tval[kTally];
  if (threadIdx.x == 0) cnt = 0;
  if (threadIdx.x < kTally) { s_tkey[threadIdx.x] = -1; s_tval[threadIdx.x] = 0u; }
  __syncthreads();
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  const int lane = threadIdx.x & 63;
  int cube = -1;
  if (i < n) {
    float x, y, z;
    if (TRANSFORM) {
      double wx, wy, wz;
      quat_rotate<double>(pose.q, (double)in[3 * (size_t)i], (double)in[3 * (size_t)i + 1], (double)in[3 * (size_t)i + 2], wx, wy, wz);
      x = (float)(wx + pose.t[0]); y = (float)(wy + pose.t[1]); z = (float)(wz + pose.t[2]);
      world[3 * (size_t)i] = x; world[3 * (size_t)i + 1] = y; world[3 * (size_t)i + 2] = z;
    } else {
      const float* p = in + (size_t)i * stride_floats;
      x = p[0]; y = p[1]; z = p[2];
    }
    const int ci = cube_coord_f(x, o0), cj = cube_coord_f(y, o1), ck = cube_coord_f(z, o2);
    if (ci >= 0 && ci < kMapW && cj >= 0 && cj < kMapH && ck >= 0 && ck < kMapD) cube = ci + kMapW * cj + kMapW * kMapH * ck;
    cube_of[i] = cube;
  }
  // one counter atomic per DISTINCT cube of the wavefront (a sweep touches a handful of cubes)
  unsigned long long todo = __ballot(cube >= 0);
  const unsigned long long inside = todo;
  while (todo) {
    const int leader = __ffsll((long long)todo) - 1;
    const int c = __builtin_amdgcn_readlane(cube, leader);
    const unsigned long long m = __ballot(cube == c);
    if (lane == leader) {
      const uint32_t add = (uint32_t)__popcll(m);
      bool done = false;
      for (int k = 0; k < kTally && !done; ++k) {
        const int32_t prev = atomicCAS(&s_tkey[k], -1, c);
        if (prev == -1 || prev == c) { atomicAdd(&s_tval[k], add); done = true; }
      }
      if (!done) front_count(b, c, add);  // (more than kTally cubes in one workgroup's 1 024 points)
    }
    todo &= ~m;
  }
  if (lane == 0 && inside) atomicAdd(&cnt, (uint32_t)__popcll(inside));
  __syncthreads();
  if (threadIdx.x == 0 && cnt) atomicAdd(b.n_inside, cnt);
  // The last workgroup to get here lays out the round.  All it takes from the others are the per-cube counters and the list
  // of touched cubes, which are only ever touched by device-scope atomics (read-modify-write here, atomic loads below) whose
  // results are consumed: they have been performed when their thread reaches the barrier, and the ticket is taken behind
  // the barrier -- with release semantics (below); cube_of and the world points are read by later launches only.
  if (threadIdx.x < kTally && s_tkey[threadIdx.x] >= 0) front_count(b, s_tkey[threadIdx.x], s_tval[threadIdx.x]);
  __syncthreads();
  // Round 6: the hand-off as the HIP memory model wants it -- every workgroup RELEASES with its ticket, the last one ACQUIRES.  Rounds 4 - 5
  // ran it with relaxed operations only (sound on gfx950, where a device-scope read-modify-write whose result is consumed has been
  // performed before its thread goes on, but outside the model); measured on the 131 072-point insert (128 workgroups, one release each):
  // Localization() 0.266 - 0.270 -> 0.269 - 0.278 ms per frame, node order 0.321 - 0.332 -> 0.330 - 0.337 (profiles/r06/ab_front_release_acquire.txt).
  // (A release in every WAVEFRONT -- 2 048 L2 write-backs -- was 40 us: the tally above is per workgroup for that reason too.)
  if (threadIdx.x == 0) last = __hip_atomic_fetch_add(b.ticket, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1u;
  __syncthreads();
  if (!last) return;
  __atomic_thread_fence(__ATOMIC_ACQUIRE);
  if (threadIdx.x < 64) {
    const uint32_t n_t = __hip_atomic_load(b.touched_n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    uint32_t h = n_t > (uint32_t)b.per_round ? (uint32_t)kFastHaltMultiRound : 0u;  // (per_round <= kMaxTouched < kTouchedCap)
    const int t = lane;
    // the touched cubes in ascending order: bitonic network over the lanes, absent entries (INT32_MAX) end up behind
    int cb = (h == 0u && (uint32_t)t < n_t) ? __hip_atomic_load(&b.touched_list[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : INT32_MAX;
#pragma unroll
    for (int k = 2; k <= 64; k <<= 1) {
#pragma unroll
      for (int j = k >> 1; j > 0; j >>= 1) {
        const int o = __shfl_xor(cb, j, 64);
        const bool keep_min = ((lane & j) == 0) == ((lane & k) == 0);
        cb = keep_min ? min(cb, o) : max(cb, o);
      }
    }
    const bool cand = h == 0u && (uint32_t)t < n_t;
    uint32_t newc = cand ? __hip_atomic_load(&b.cube_cnt[cb], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u, oldc = 0u, bad = 0u;
    int slot = 0;
    if (cand) {
      slot = b.cube_slot[cb];
      if (slot < 0) bad = kFastHaltUnallocated;
      else {
        oldc = b.slot_count[slot];
        if (oldc != 0u && b.slot_ok[slot] == 0u) bad = kFastHaltNeedsSort;
      }
    }
    const unsigned long long mu = __ballot(bad == (uint32_t)kFastHaltUnallocated), ms = __ballot(bad == (uint32_t)kFastHaltNeedsSort);
    if (h == 0u) h = mu ? (uint32_t)kFastHaltUnallocated : (ms ? (uint32_t)kFastHaltNeedsSort : 0u);
    const bool live = cand && h == 0u;
    if (!live) { oldc = 0u; newc = 0u; slot = 0; }
    uint32_t io = oldc, ir = oldc + newc;  // inclusive scans over the lanes
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const uint32_t a0 = (uint32_t)__shfl_up((int)io, d, 64), a1 = (uint32_t)__shfl_up((int)ir, d, 64);
      if (lane >= d) { io += a0; ir += a1; }
    }
    const uint32_t n_old = (uint32_t)__shfl((int)io, 63, 64);
    MapTouched& T = *b.tt;
    if (t < kMaxTouched) {
      T.cube[t] = live ? cb : INT32_MAX;
      T.slot[t] = (uint32_t)slot;
      T.old_prefix[t] = live ? io - oldc : n_old;
      T.region_base[t] = live ? ir - (oldc + newc) : 0u;
      int w[3] = {0, 0, 0};
      if (live) { w[0] = cb % kMapW - o0; w[1] = (cb / kMapW) % kMapH - o1; w[2] = cb / (kMapW * kMapH) - o2; }
#pragma unroll
      for (int ax = 0; ax < 3; ++ax) {  // DeviceMap::add_surf_dev's arithmetic
        const double cm = w[ax] * kCube - kHalfCube;
        T.cube_min[t][ax] = live ? cm : 0.0;
        T.leaf_lo[t][ax] = live ? (int)floorf((float)cm * b.inv_leaf) - 2 : 0;
        T.wcube[t][ax] = w[ax];
      }
    }
    if (t == kMaxTouched) T.old_prefix[kMaxTouched] = n_old;
    if (t == 0) {
      T.n = h ? 0 : (int32_t)n_t;
      T.inv_leaf_watch = b.inv_leaf; T.dirty = b.dirty; T.lbits = b.lbits;
      if (h) *b.halt = h;
      // every workgroup has passed the ticket: the input points have been read (the caller may hand their buffer on)
      __hip_atomic_store(&b.rep->front_seq, b.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

// Second stage of a device-built round: exclusive scan of the touched cubes' cell grids, cube by cube (blockIdx.y), each
// from its own base (MapTouched::region_base) -- so the cube's new cell_start table, its point count and the positions of
// the scratch arrays all come out of the SAME launch (the host-built round needs a device-wide scan whose length the host
// knows, and cell_table_kernel behind it).  Single pass with decoupled look-back: a workgroup takes a ticket (so that its
// predecessors are running), publishes its aggregate, and one wavefront sums the 64 records before it at a time until it
// meets an inclusive one.  Records: flag (1 = aggregate, 2 = inclusive) << 62 | value; all zero between inserts.
__global__ __launch_bounds__(256) void cell_scan_table_kernel(uint32_t* __restrict__ grid, uint32_t* __restrict__ grid_scan,
                                                              const MapTouched* __restrict__ ttp, uint32_t cap, uint32_t ncell1,
                                                              uint32_t* __restrict__ cell_start, uint32_t* __restrict__ counts,
                                                              uint32_t* __restrict__ slot_count, const uint32_t* __restrict__ halt,
                                                              unsigned long long* __restrict__ state, uint32_t* __restrict__ tickets, uint32_t nblk) {
  const uint32_t t = blockIdx.y;
  if ((int)t >= ttp->n || *halt) return;
  __shared__ uint32_t s_bid, s_wsum[4], s_excl;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) s_bid = atomicAdd(&tickets[t], 1u);
  __syncthreads();
  const uint32_t bid = s_bid;
  constexpr int kPer = (int)(kScanItems / 256u);
  const uint32_t c0 = bid * kScanItems + (uint32_t)tid * kPer;
  uint32_t* g = grid + (size_t)t * ncell1;
  uint32_t v[kPer], tsum = 0;
#pragma unroll
  for (int k = 0; k < kPer; ++k) { v[k] = c0 + k < ncell1 ? g[c0 + k] : 0u; tsum += v[k]; }
  uint32_t inc = tsum;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t a0 = (uint32_t)__shfl_up((int)inc, d, 64);
    if (lane >= d) inc += a0;
  }
  if (lane == 63) s_wsum[wave] = inc;
  __syncthreads();
  uint32_t wbase = 0;
  for (int w = 0; w < wave; ++w) wbase += s_wsum[w];
  const uint32_t agg = s_wsum[0] + s_wsum[1] + s_wsum[2] + s_wsum[3];
  const uint32_t texcl = wbase + inc - tsum;
  if (wave == 0) {
    unsigned long long* st = state + (size_t)t * nblk;
    if (lane == 0) __hip_atomic_store(&st[bid], ((bid == 0u ? 2ull : 1ull) << 62) | (unsigned long long)agg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    uint32_t excl = 0;
    int base = (int)bid - 1;
    while (base >= 0) {
      const int j = base - lane;
      unsigned long long rec = 2ull << 62;  // before the first workgroup: "inclusive prefix 0"
      if (j >= 0) {
        do { rec = __hip_atomic_load(&st[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); } while ((rec >> 62) == 0ull);
      }
      const unsigned long long mi = __ballot((rec >> 62) == 2ull);
      const int first = mi ? __ffsll((long long)mi) - 1 : 64;  // the nearest predecessor with an inclusive prefix
      uint32_t contrib = lane <= first ? (uint32_t)(rec & 0xFFFFFFFFull) : 0u;
#pragma unroll
      for (int d = 32; d >= 1; d >>= 1) contrib += (uint32_t)__shfl_xor((int)contrib, d, 64);
      excl += contrib;
      if (mi) break;
      base -= 64;
    }
    if (lane == 0) {
      if (bid != 0u) __hip_atomic_store(&st[bid], (2ull << 62) | (unsigned long long)(excl + agg), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      s_excl = excl;
    }
  }
  __syncthreads();
  const uint32_t slot = ttp->slot[t], rb = ttp->region_base[t];
  uint32_t run = s_excl + texcl;
#pragma unroll
  for (int k = 0; k < kPer; ++k) {
    const uint32_t c = c0 + k;
    if (c < ncell1) {
      grid_scan[(size_t)t * ncell1 + c] = rb + run;
      cell_start[(size_t)slot * ncell1 + c] = slot * cap + run;
      g[c] = 0u;  // (the counters have done their work: the next insert finds the grids clean)
      if (c == ncell1 - 1u) { counts[t] = run; slot_count[slot] = run; }
      run += v[k];
    }
  }
}

// End of a device-built insert: what the host has to know goes to pinned memory, the per-slot "one point per leaf" marks
// follow the drift watch, and every counter of the insert is put back to zero (no fill before the next one).
__global__ __launch_bounds__(256) void insert_report_kernel(const MapTouched* __restrict__ ttp, uint32_t* __restrict__ small, uint32_t small_words,
                                                            uint32_t* __restrict__ cube_cnt, unsigned long long* __restrict__ scan_state,
                                                            uint32_t* __restrict__ tickets, uint32_t* __restrict__ slot_ok,
                                                            MapFastReport* __restrict__ rep, unsigned long long seq) {
  const int tid = threadIdx.x;
  const uint32_t halt = small[5], n = (uint32_t)ttp->n, dirty = small[7];
  if (tid < kMaxTouched) {
    rep->cube[tid] = ttp->cube[tid];
    rep->count[tid] = small[8 + tid];
    if (!halt && (uint32_t)tid < n) slot_ok[ttp->slot[tid]] = ((dirty >> tid) & 1u) ? 0u : 1u;
  }
  if (tid == 0) { rep->halt = halt; rep->n = n; rep->n_inside = small[48]; rep->dirty = dirty; rep->n_old = ttp->old_prefix[kMaxTouched]; rep->pad = 0u; }
  __syncthreads();  // (every thread has read the counters)
  for (uint32_t i = (uint32_t)tid; i < small_words; i += 256u) small[i] = 0u;
  for (uint32_t i = (uint32_t)tid; i < (uint32_t)kMapNum; i += 256u) cube_cnt[i] = 0u;
  for (uint32_t i = (uint32_t)tid; i < (uint32_t)kScanStateWords; i += 256u) scan_state[i] = 0ull;
  if (tid <= kMaxTouched + 1) tickets[tid] = 0u;  // (the scan's tickets, the front kernel's ticket, its count of touched cubes)
  __threadfence_system();
  __syncthreads();
  if (tid == 0) __hip_atomic_store(&rep->seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// ------------------------------------------------------------------------------------------------
// Scan pre-filter (laserMapping::adjustVoxelSize, laserMapping.cpp:598-651): cloud statistics + pcl::VoxelGrid of the
// surf cloud at planeRes, on the device.
// ------------------------------------------------------------------------------------------------
// per-workgroup partial sums of |x|, |y|, |z| (fp64), count of points farther than 3 m, min / max of the coordinates
__global__ __launch_bounds__(256) void vg_stats_kernel(const float* __restrict__ xyz, uint32_t n, uint32_t stride_floats,
                                                       double* __restrict__ part /* [blocks][10] */) {
  __shared__ double sh[256][10];
  double a[10] = {0, 0, 0, 0, 3.0e38, 3.0e38, 3.0e38, -3.0e38, -3.0e38, -3.0e38};
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const float* p = xyz + (size_t)i * stride_floats;
    const float x = p[0], y = p[1], z = p[2];
    a[0] += (double)fabsf(x); a[1] += (double)fabsf(y); a[2] += (double)fabsf(z);
    if (x * x + y * y + z * z > 9.f) a[3] += 1.0;  // laserMapping.cpp:611 (float arithmetic)
    a[4] = fmin(a[4], (double)x); a[5] = fmin(a[5], (double)y); a[6] = fmin(a[6], (double)z);
    a[7] = fmax(a[7], (double)x); a[8] = fmax(a[8], (double)y); a[9] = fmax(a[9], (double)z);
  }
  for (int k = 0; k < 10; ++k) sh[threadIdx.x][k] = a[k];
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s)
      for (int k = 0; k < 10; ++k) {
        const double o = sh[threadIdx.x + s][k];
        double& m = sh[threadIdx.x][k];
        m = k < 4 ? m + o : (k < 7 ? fmin(m, o) : fmax(m, o));
      }
    __syncthreads();
  }
  if (threadIdx.x < 10) part[(size_t)blockIdx.x * 10 + threadIdx.x] = sh[0][threadIdx.x];
}

// pcl::VoxelGrid leaf index: ijk = floor(p * inv_leaf) - min_b in float arithmetic, idx = i0 + i1*d0 + i2*d0*d1
__global__ __launch_bounds__(256) void vg_keys_kernel(const float* __restrict__ xyz, uint32_t n, uint32_t stride_floats, float inv_leaf,
                                                      int mb0, int mb1, int mb2, int d0, int d01, float4* __restrict__ wpts,
                                                      uint32_t* __restrict__ keys, uint32_t* __restrict__ vals,
                                                      const VgDecision* __restrict__ dec) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (dec) {  // the grid was laid out on the device (vg_decide_kernel)
    inv_leaf = dec->inv_leaf; mb0 = dec->min_b[0]; mb1 = dec->min_b[1]; mb2 = dec->min_b[2];
    d0 = dec->div_b[0]; d01 = dec->div_b[0] * dec->div_b[1];
  }
  const float* p = xyz + (size_t)i * stride_floats;
  const float x = p[0], y = p[1], z = p[2];
  wpts[i] = make_float4(x, y, z, 0.f);
  vals[i] = i;
  const int i0 = (int)(floorf(x * inv_leaf) - (float)mb0), i1 = (int)(floorf(y * inv_leaf) - (float)mb1), i2 = (int)(floorf(z * inv_leaf) - (float)mb2);
  keys[i] = (uint32_t)(i0 + i1 * d0 + i2 * d01);
}

// centroid of every leaf (float sums in the stable-sorted input order), packed xyz out; long leaves go to the wavefront kernel
__global__ __launch_bounds__(256) void vg_centroid_kernel(const uint32_t* __restrict__ heads, const uint32_t* __restrict__ n_cent,
                                                          const float4* __restrict__ spts, float* __restrict__ out,
                                                          uint32_t* __restrict__ long_list, uint32_t* __restrict__ long_count,
                                                          VgDecision* __restrict__ dec) {
  const uint32_t o = blockIdx.x * blockDim.x + threadIdx.x;
  if (o == 0u && dec) dec->n_leaves = *n_cent;  // (read back with the decision)
  if (o >= *n_cent) return;
  const uint32_t beg = heads[o], end = heads[o + 1];
  if (end - beg > kLongLeaf) {
    const uint32_t at = atomicAdd(long_count, 1u);
    if (at < kMaxLongLeaves) { long_list[at] = o; return; }
  }
  float s0 = 0.f, s1 = 0.f, s2 = 0.f;
  constexpr int B = 16;
  for (uint32_t j = beg; j < end; j += B) {
    float4 cur[B];
#pragma unroll
    for (int k = 0; k < B; ++k) cur[k] = spts[j + k < end ? j + k : end - 1];
#pragma unroll
    for (int k = 0; k < B; ++k)
      if (j + k < end) { s0 += cur[k].x; s1 += cur[k].y; s2 += cur[k].z; }
  }
  const float cnt = (float)(end - beg);
  out[3 * (size_t)o] = s0 / cnt; out[3 * (size_t)o + 1] = s1 / cnt; out[3 * (size_t)o + 2] = s2 / cnt;
}
// long leaves (the ground under the sensor: up to ~1 600 points of a raw sweep in one leaf): one WAVEFRONT per leaf and
// coordinate.  The wavefront fetches 256 points at a time (coalesced, the next 256 in flight), parks its coordinate in LDS,
// and every lane adds the same values in order from its own registers: one dependent addition per element (handing the
// elements around with v_readlane cost readlane + wait state + add per element and coordinate: 20 us for 1 636 points).
// Values behind the end of the leaf are +0.0f (s + 0.0f == s).
__global__ __launch_bounds__(256) void vg_centroid_long_kernel(const uint32_t* __restrict__ heads, const float4* __restrict__ spts,
                                                               float* __restrict__ out, const uint32_t* __restrict__ long_list,
                                                               const uint32_t* __restrict__ long_count) {
  __shared__ __attribute__((aligned(16))) float buf[4][2][256];
  const uint32_t w = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const uint32_t n_long = *long_count < kMaxLongLeaves ? *long_count : kMaxLongLeaves;
  if (w >= 3u * n_long) return;
  const uint32_t o = long_list[w / 3u];
  const uint32_t axis = w % 3u;
  const uint32_t beg = heads[o], end = heads[o + 1];
  const float* src = reinterpret_cast<const float*>(spts) + axis;
  float s = 0.f;
  float nxt[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const uint32_t idx = beg + (uint32_t)k * 64u + (uint32_t)lane;
    nxt[k] = idx < end ? src[4 * (size_t)idx] : 0.f;
  }
  int ph = 0;
  for (uint32_t base = beg; base < end; base += 256u, ph ^= 1) {
    float* b = buf[wv][ph];
#pragma unroll
    for (int k = 0; k < 4; ++k) b[k * 64 + lane] = nxt[k];
    if (base + 256u < end) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const uint32_t idx = base + 256u + (uint32_t)k * 64u + (uint32_t)lane;
        nxt[k] = idx < end ? src[4 * (size_t)idx] : 0.f;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const uint32_t m = end - base < 256u ? end - base : 256u, m4 = (m + 3u) & ~3u;
    for (uint32_t q = 0; q < m4; q += 4) {
      const float4 v = *reinterpret_cast<const float4*>(b + q);
      s += v.x; s += v.y; s += v.z; s += v.w;
    }
  }
  if (lane == 0) out[3 * (size_t)o + axis] = s / (float)(end - beg);
}

// adjustVoxelSize's decision and the leaf grid of pcl::VoxelGrid, on the device (VgDecision): what so_icp_prefilter_scan used to
// work out on the host between two read-backs.  Threads 0..9 add the workgroups' partial statistics in workgroup order (the
// order the host used), thread 0 decides with the host's arithmetic (laserMapping.cpp:604-636; voxel_grid.hpp applyFilter).
constexpr int kVgStatBlocksMax = 256;
__global__ __launch_bounds__(256) void vg_decide_kernel(const double* __restrict__ part, int blocks, uint32_t n, int auto_voxel_size, VgCandidates cand,
                                                        VgDecision* __restrict__ out, uint32_t* __restrict__ counters,
                                                        unsigned long long* __restrict__ scan_state, uint32_t n_state) {
  __shared__ double acc[10];
  __shared__ double sp[kVgStatBlocksMax * 10];  // (the partials through LDS: 256 dependent round trips to memory per thread took 54 us)
  const int k = threadIdx.x;
  if (k < 16) counters[k] = 0u;
  for (uint32_t i = (uint32_t)k; i < n_state; i += blockDim.x) scan_state[i] = 0ull;
  {  // (all of a thread's loads in flight together: a loop of unknown length made ten round trips of them)
    constexpr int kLoads = kVgStatBlocksMax * 10 / 256;
    double v[kLoads];
#pragma unroll
    for (int q = 0; q < kLoads; ++q) v[q] = k + q * 256 < blocks * 10 ? part[k + q * 256] : 0.0;
#pragma unroll
    for (int q = 0; q < kLoads; ++q) sp[k + q * 256] = v[q];
  }
  __syncthreads();
  // sums, minima and maxima by three different wavefronts (one chain of dependent fp64 operations per statistic; a single loop
  // that picked the operation per lane cost 110 cycles per element: 13 us), the sums in workgroup order like the host's
  const int stat = (k & 63) + (k >> 6) * 100;  // 0..3 sums, 100..102 minima, 200..202 maxima
  if (stat < 4) {
    double a = 0.0;
    for (int b0 = 0; b0 < blocks; ++b0) a += sp[b0 * 10 + stat];
    acc[stat] = a; out->acc[stat] = a;
  } else if (stat >= 100 && stat < 103) {
    double a = 3.0e38;
    for (int b0 = 0; b0 < blocks; ++b0) a = fmin(a, sp[b0 * 10 + 4 + (stat - 100)]);
    acc[4 + stat - 100] = a; out->acc[4 + stat - 100] = a;
  } else if (stat >= 200 && stat < 203) {
    double a = -3.0e38;
    for (int b0 = 0; b0 < blocks; ++b0) a = fmax(a, sp[b0 * 10 + 7 + (stat - 200)]);
    acc[7 + stat - 200] = a; out->acc[7 + stat - 200] = a;
  }
  __syncthreads();
  if (k != 0) return;
  uint32_t flags = 0u;
  int choice = 1;
  float avg = 0.f;
  if (auto_voxel_size) {
    const double dn = (double)n;
    const float ax = (float)(acc[0] / dn), ay = (float)(acc[1] / dn), az = (float)(acc[2] / dn);
    const double stat64 = (acc[0] / dn) * (acc[1] / dn) * (acc[2] / dn);
    const double band = 3.1 * dn * 5.9604644775390625e-8 + 1e-6;
    if (fabs(stat64 - 25.0) <= 25.0 * band || fabs(stat64 - 65.0) <= 65.0 * band) flags |= kVgNeedInputOrder;
    avg = ax * ay * az;  // laserMapping.cpp:620-621 (float product)
    if ((double)avg < 25) choice = 0;
    else if ((double)avg > 65) choice = 2;
  }
  const float inv = cand.inv_leaf[choice];
  const float mn[3] = {(float)acc[4], (float)acc[5], (float)acc[6]}, mx[3] = {(float)acc[7], (float)acc[8], (float)acc[9]};
  const long long dx = (long long)((mx[0] - mn[0]) * inv) + 1, dy = (long long)((mx[1] - mn[1]) * inv) + 1, dz = (long long)((mx[2] - mn[2]) * inv) + 1;
  if (dx * dy * dz > (long long)INT32_MAX) flags |= kVgLeafTooSmall;
  for (int a = 0; a < 3; ++a) {
    out->min_b[a] = (int)floorf(mn[a] * inv);
    out->div_b[a] = (int)floorf(mx[a] * inv) - out->min_b[a] + 1;
  }
  out->average_distance = avg; out->leaf = cand.plane_res[choice]; out->inv_leaf = inv;
  out->line_res = cand.line_res[choice]; out->plane_res = cand.plane_res[choice];
  out->choice = choice; out->flags = flags; out->n_leaves = 0u;
}
void launch_vg_decide(const double* d_part, int blocks, uint32_t n, int auto_voxel_size, const VgCandidates& cand, VgDecision* d_out,
                      uint32_t* d_counters, unsigned long long* d_scan_state, uint32_t n_state, hipStream_t s) {
  hipLaunchKernelGGL(vg_decide_kernel, dim3(1), dim3(256), 0, s, d_part, blocks <= kVgStatBlocksMax ? blocks : kVgStatBlocksMax, n, auto_voxel_size, cand,
                     d_out, d_counters, d_scan_state, n_state);
}
// Stable (key, index) sort of the working set: merge sort with 2048-item block sorts and odd-even merges (measured against
// rocPRIM's own choice, its Onesweep radix sort and two other merge configurations in rounds 1-2: the fastest at these sizes).
static hipError_t map_sort(void* tmp, size_t& bytes, const uint32_t* ki, uint32_t* ko, const uint32_t* vi, uint32_t* vo, size_t n,
                           unsigned /*end_bit*/, hipStream_t s) {
  using M1 = rocprim::merge_sort_config<512, 512, 4, 128, 128, 4, (1u << 30)>;
  return rocprim::merge_sort<M1>(tmp, bytes, ki, ko, vi, vo, n, rocprim::less<uint32_t>(), s);
}

size_t map_sort_temp_bytes(size_t n) {
  size_t a = 0, b = 0;
  (void)map_sort(nullptr, a, (const uint32_t*)nullptr, (uint32_t*)nullptr, (const uint32_t*)nullptr, (uint32_t*)nullptr, n, 32, (hipStream_t)0);
  (void)rocprim::exclusive_scan(nullptr, b, (const uint32_t*)nullptr, (uint32_t*)nullptr, 0u, n, rocprim::plus<uint32_t>(), (hipStream_t)0);
  return a > b ? a : b;
}

void launch_world_cube(const float* d_xyz, uint32_t n, uint32_t stride_floats, const int origin[3], int32_t* d_cube_of, uint8_t* d_touched,
                       uint32_t* d_n_inside, hipStream_t s) {
  if (!n) return;
  hipLaunchKernelGGL(world_cube_kernel, grid_for(n, 1024), dim3(1024), 0, s, d_xyz, n, stride_floats, origin[0], origin[1], origin[2], d_cube_of,
                     d_touched, d_n_inside);
}
void launch_transform_scan(const float* d_scan, uint32_t n, const Pose& pose, float* d_out, hipStream_t s) {
  if (!n) return;
  hipLaunchKernelGGL(transform_scan_kernel, grid_for(n, 256), dim3(256), 0, s, d_scan, n, pose, d_out);
}
// first stage by hash grouping: mslot = keys1, mrank = vals1, member list = pos, group ranges = heads / flags, cell keys =
// vals0; lists of the larger groups in spts (first-stage scratch of the sort path, the second stage's scratch later): at
// most n_new / 16 groups of more than 16 members, n_new / 64 of more than 64; counters in d_n_cent[2..6]
// (a.n_old: the number of old points, or -- device-built round -- a bound on it: the kernels take the number from d_tt)
// (count_cells: the centroids are counted into the cell grids as they are produced -- no cell_count_kernel behind this stage;
//  the ranks go to vals1, where a matched old point and every new point keep their member ranks until the members are placed:
//  an old point has one or the other, and the groups' ranks are written after the placement)
static uint32_t* launch_first_stage_hashed(const MapInsertArgs& a, hipStream_t s, bool count_cells) {
  const LeafTable ht{a.ht_key, a.ht_cnt, a.ht_off, a.ht_log2};
  const CellCount cc{count_cells ? a.grid : nullptr, a.vals1, a.ncell1};
  unsigned long long* cursor = reinterpret_cast<unsigned long long*>(a.d_n_cent + 2);
  uint32_t* keys2 = a.vals0;
  uint32_t* medium_list = reinterpret_cast<uint32_t*>(a.spts);
  uint32_t* giant_list = medium_list + a.n_new / 16u + 2u;
  const uint32_t n_old_grid = a.n_old_grid ? a.n_old_grid : a.n_old, total = n_old_grid + a.n_new;  // (launch sizes)
  if (a.n_new)
    hipLaunchKernelGGL(leafhash_insert_new_kernel, grid_for(a.n_new, 256), dim3(256), 0, s, a.d_xyz, a.n_new, a.stride_floats, a.d_cube_of,
                       a.d_tt, a.inv_leaf, a.nc, a.inv_cell, a.rank, a.world, a.wpts, a.keys0, ht, a.keys1, a.vals1);
  if (a.n_old)
    hipLaunchKernelGGL(leafhash_match_old_kernel, grid_for(n_old_grid, 256), dim3(256), 0, s, a.pool, a.cap, a.inv_leaf, a.keys0, a.wpts, ht,
                       a.keys1, a.vals1, a.d_tt, a.nc, a.inv_cell, a.cent, keys2, cc);
  hipLaunchKernelGGL(leafhash_offsets_kernel, dim3((1u << a.ht_log2) / 4096u), dim3(1024), 0, s, ht, a.heads, a.flags, cursor, medium_list, a.d_n_cent + 6,
                     giant_list, a.d_n_cent + 4);
  hipLaunchKernelGGL(leafhash_place_kernel, grid_for(total, 256), dim3(256), 0, s, a.keys1, a.vals1, a.n_new, a.d_tt, a.ht_off, a.pos);
  const uint32_t max_medium = a.n_new / 16u + 1u, max_giant = a.n_new / 64u + 1u;
  const uint32_t small_blocks = (a.n_new ? a.n_new + 255u : 256u) / 256u;  // (256 groups per workgroup, see the kernel)
  const uint32_t medium_blocks = max_medium < 2048u ? (max_medium + 15u) / 16u : 128u, giant_blocks = max_giant < 512u ? max_giant : 512u;
  const size_t giant_lds = giant_lds_bytes(a.n_new);
  const uint32_t nhist = giant_hist_entries(a.n_new);
  // (per launch, not once per process: the attribute belongs to the current device, and one process may drive several)
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(leafhash_centroids_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)giant_lds);
  hipLaunchKernelGGL(leafhash_centroids_kernel, dim3(giant_blocks + small_blocks + medium_blocks), dim3(kGiantThreads), giant_lds, s, a.heads, a.flags, cursor,
                     a.pos, a.wpts, a.keys0, a.n_new, nhist, a.d_tt, a.nc, a.inv_cell, a.cent, keys2, a.d_n_cent, medium_list, a.d_n_cent + 6, giant_list,
                     a.d_n_cent + 4, a.d_n_cent + 5, giant_blocks, small_blocks, cc);
  return keys2;
}

void launch_map_insert(const MapInsertArgs& a, hipStream_t s) {
  const uint32_t total = a.n_old + a.n_new;
  if (!total) return;
  hipLaunchKernelGGL(tt_store_kernel, dim3(1), dim3(64), 0, s, a.tt, a.d_tt);
  size_t tb = a.temp_bytes;
  const bool hashed = a.ht_key != nullptr && a.grid != nullptr;
  if (!hashed) {  // (the hash grouping's first two kernels build the working set themselves)
    if (a.n_old && a.reorder_old) {
      OldGrids og;
      for (int t = 0; t < kMaxTouched; ++t) og.inv_leaf[t] = a.old_inv_leaf[t];
      hipLaunchKernelGGL(old_order_key_kernel, grid_for(a.n_old, 256), dim3(256), 0, s, a.d_tt, og, a.pool, a.cap, a.n_old, a.keys0, a.vals0);
      (void)map_sort(a.temp, tb, a.keys0, a.keys1, a.vals0, a.vals1, (size_t)a.n_old, 30, s);  // stable
      tb = a.temp_bytes;
      hipLaunchKernelGGL(gather_old_ordered_kernel, grid_for(a.n_old, 256), dim3(256), 0, s, a.d_tt, a.pool, a.cap, a.inv_leaf, a.n_old, a.vals1, a.wpts,
                         a.keys0, a.vals0);
    } else if (a.n_old)
      hipLaunchKernelGGL(gather_old_kernel, grid_for(a.n_old, 256), dim3(256), 0, s, a.d_tt, a.pool, a.cap, a.inv_leaf, a.n_old, a.wpts, a.keys0, a.vals0);
    if (a.n_new)
      hipLaunchKernelGGL(append_new_kernel, grid_for(a.n_new, 256), dim3(256), 0, s, a.d_xyz, a.n_new, a.stride_floats, a.d_cube_of,
                         a.d_tt, a.inv_leaf, a.n_old, a.wpts, a.keys0, a.vals0, a.nc, a.inv_cell, a.rank, a.world);
  }
  uint32_t* keys2 = a.keys0;  // cell key per centroid, input of the second stage
  if (hashed) {
    keys2 = launch_first_stage_hashed(a, s, false);  // first stage without a sort (see leafhash_insert_new_kernel)
  } else {
    (void)map_sort(a.temp, tb, a.keys0, a.keys1, a.vals0, a.vals1, (size_t)total, 32, s);  // stable
    hipLaunchKernelGGL(leaf_flags_kernel, grid_for(total, 256), dim3(256), 0, s, a.keys1, total, a.flags);
    tb = a.temp_bytes;
    (void)rocprim::exclusive_scan(a.temp, tb, a.flags, a.pos, 0u, (size_t)total, rocprim::plus<uint32_t>(), s);
    hipLaunchKernelGGL(leaf_heads_kernel, grid_for(total, 256), dim3(256), 0, s, a.keys1, a.vals1, a.flags, a.pos, total, a.wpts, a.spts, a.heads,
                       a.d_n_cent);
    // d_n_cent + 1 = number of long leaves (cleared with d_small_), list = the flags array (free after the scan)
    hipLaunchKernelGGL(leaf_centroid_kernel, grid_for(total, 256), dim3(256), 0, s, a.keys1, a.heads, a.d_n_cent, a.spts, a.d_tt, a.nc, a.inv_cell,
                       a.cent, a.keys0, a.vals0, a.flags, a.d_n_cent + 1);
    hipLaunchKernelGGL(leaf_centroid_long_kernel, dim3(kMaxLongLeaves / 4), dim3(256), 0, s, a.keys1, a.heads, a.spts, a.d_tt, a.nc, a.inv_cell,
                       a.cent, a.keys0, a.vals0, a.flags, a.d_n_cent + 1);
  }
  {  // second stage by counting into the cell grids (no sort)
    const size_t gn = (size_t)a.tt.n * a.ncell1;
    const uint32_t* halt = a.d_n_cent + 5;  // raised by leafhash_giant_kernel: the round is repeated with the sort-based first stage
    if (!a.grid_is_clean) (void)hipMemsetAsync(a.grid, 0, (gn + 1) * sizeof(uint32_t), s);  // (else: cleaned by the previous round's cell_table_kernel)
    hipLaunchKernelGGL(cell_count_kernel, grid_for(total, 256), dim3(256), 0, s, keys2, a.d_n_cent, a.ncell1, a.grid, a.vals1, halt);
    tb = a.temp_bytes;
    (void)rocprim::exclusive_scan(a.temp, tb, a.grid, a.grid_scan, 0u, gn + 1, rocprim::plus<uint32_t>(), s);
    hipLaunchKernelGGL(cell_table_kernel, dim3((a.ncell1 + 255) / 256, a.tt.n), dim3(256), 0, s, a.grid_scan, a.d_tt, a.cap, a.ncell1, a.cell_start,
                       a.d_counts, halt, a.grid);
    hipLaunchKernelGGL(cell_place_kernel, grid_for(total, 256), dim3(256), 0, s, keys2, a.vals1, a.d_n_cent, a.grid_scan, a.cent, a.d_tt, a.ncell1,
                       a.inv_leaf, a.spts, a.flags, halt);  // spts / flags (first-stage scratch) are free after the centroids
    hipLaunchKernelGGL(cell_rank_kernel, grid_for(total, 256), dim3(256), 0, s, keys2, a.d_n_cent, a.grid_scan, a.cent, a.spts, a.flags, a.d_tt,
                       a.cap, a.ncell1, a.inv_leaf, a.pool, a.vals1, halt);
    if (a.world > 1 && a.d_owned)
      hipLaunchKernelGGL(count_owned_kernel, dim3((total + 255) / 256, a.tt.n), dim3(256), 0, s, a.pool, a.cap, a.d_tt, a.d_counts, a.nc, a.inv_cell,
                         a.rank, a.world, a.d_owned);
    return;
  }
}

// The insert as one uninterrupted sequence of launches (DeviceMap::insert_fast): a.n_old is a BOUND on the round's old points
// (the launches are sized by it), a.tt only carries lbits; everything else of the round is laid out by the front kernel.
void launch_map_insert_fast(const MapInsertArgs& a, const MapFastArgs& f, hipStream_t s) {
  if (!f.n) return;
  const FrontBuild b{f.d_cube_slot, f.d_slot_count, f.d_slot_ok, f.d_cube_cnt, f.d_small + 48, f.d_tickets + kMaxTouched, f.d_small + 5, f.d_small + 7,
                     a.d_tt, a.inv_leaf, a.tt.lbits, f.per_round, f.h_report, f.seq,
                     f.d_tickets + kMaxTouched + 1, reinterpret_cast<int32_t*>(f.d_tickets + kMaxTouched + 2)};
  if (f.transform)
    hipLaunchKernelGGL(insert_front_kernel<true>, grid_for(f.n, 1024), dim3(1024), 0, s, f.d_in, f.n, 3u, f.pose, f.d_world, f.origin[0], f.origin[1],
                       f.origin[2], const_cast<int32_t*>(a.d_cube_of), b);
  else
    hipLaunchKernelGGL(insert_front_kernel<false>, grid_for(f.n, 1024), dim3(1024), 0, s, f.d_in, f.n, f.stride_floats, f.pose, (float*)nullptr, f.origin[0],
                       f.origin[1], f.origin[2], const_cast<int32_t*>(a.d_cube_of), b);
  uint32_t* keys2 = launch_first_stage_hashed(a, s, /*count_cells=*/true);  // (cell counting folded into the kernels that produce the centroids)
  const uint32_t total = (a.n_old_grid ? a.n_old_grid : a.n_old) + a.n_new;
  const uint32_t* halt = a.d_n_cent + 5;
  const uint32_t nblk = (a.ncell1 + kScanItems - 1u) / kScanItems;
  hipLaunchKernelGGL(cell_scan_table_kernel, dim3(nblk, (uint32_t)f.per_round), dim3(256), 0, s, a.grid, a.grid_scan, a.d_tt, a.cap, a.ncell1, a.cell_start,
                     a.d_counts, f.d_slot_count, halt, f.d_scan_state, f.d_tickets, nblk);
  hipLaunchKernelGGL(cell_place_kernel, grid_for(total, 256), dim3(256), 0, s, keys2, a.vals1, a.d_n_cent, a.grid_scan, a.cent, a.d_tt, a.ncell1,
                     a.inv_leaf, a.spts, a.flags, halt);
  hipLaunchKernelGGL(cell_rank_kernel, grid_for(total, 256), dim3(256), 0, s, keys2, a.d_n_cent, a.grid_scan, a.cent, a.spts, a.flags, a.d_tt,
                     a.cap, a.ncell1, a.inv_leaf, a.pool, a.vals1, halt);
  hipLaunchKernelGGL(insert_report_kernel, dim3(1), dim3(256), 0, s, a.d_tt, f.d_small, f.small_words, f.d_cube_cnt, f.d_scan_state, f.d_tickets, f.d_slot_ok,
                     f.h_report, f.seq);
}
// Resolution change (localMap.planeRes_ is pushed every frame, laserMapping.cpp:648-649): the points of a cube stay as they
// are -- the reference re-filters a block only when the next insert touches it (LocalMap.h:617-641) -- only the cell
// grid of the index follows the new planeRes.  The resident points take the place of the "centroids" of an insert's
// second stage: counted into the new cell grids, scanned into the new tables, placed in ascending (old) leaf order.
__global__ __launch_bounds__(256) void retable_gather_kernel(const MapTouched* __restrict__ ttp, const float4* __restrict__ pool, uint32_t cap, uint32_t n_old, int nc,
                                                             double inv_cell, float4* __restrict__ cent, uint32_t* __restrict__ keys2,
                                                             uint32_t* __restrict__ n_cent) {
  const MapTouched& tt = *ttp;
  const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e == 0) *n_cent = n_old;
  if (e >= n_old) return;
  int t = 0;
#pragma unroll
  for (int step = 16; step >= 1; step >>= 1) t = (t + step < tt.n && tt.old_prefix[t + step] <= e) ? t + step : t;
  const float4 p = pool[(size_t)tt.slot[t] * cap + (e - tt.old_prefix[t])];
  cent[e] = make_float4(p.x, p.y, p.z, 0.f);
  int g[3];
  const float c3[3] = {p.x, p.y, p.z};
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const int v = (int)floor(((double)c3[a] - tt.cube_min[t][a]) * inv_cell);
    g[a] = v < 0 ? 0 : (v >= nc ? nc - 1 : v);
  }
  keys2[e] = ((uint32_t)t << 18) | (uint32_t)((g[2] * nc + g[1]) * nc + g[0]);
}
// a.inv_leaf = 1 / the planeRes the points were last filtered with (their leaf keys are distinct there); a.grid required
void launch_map_retable(const MapInsertArgs& a, hipStream_t s) {
  const uint32_t total = a.n_old;
  if (!total) return;
  hipLaunchKernelGGL(tt_store_kernel, dim3(1), dim3(64), 0, s, a.tt, a.d_tt);
  hipLaunchKernelGGL(retable_gather_kernel, grid_for(total, 256), dim3(256), 0, s, a.d_tt, a.pool, a.cap, total, a.nc, a.inv_cell, a.cent, a.keys0,
                     a.d_n_cent);
  const size_t gn = (size_t)a.tt.n * a.ncell1;
  if (!a.grid_is_clean) (void)hipMemsetAsync(a.grid, 0, (gn + 1) * sizeof(uint32_t), s);
  hipLaunchKernelGGL(cell_count_kernel, grid_for(total, 256), dim3(256), 0, s, a.keys0, a.d_n_cent, a.ncell1, a.grid, a.vals1, a.d_n_cent + 5);
  size_t tb = a.temp_bytes;
  (void)rocprim::exclusive_scan(a.temp, tb, a.grid, a.grid_scan, 0u, gn + 1, rocprim::plus<uint32_t>(), s);
  hipLaunchKernelGGL(cell_table_kernel, dim3((a.ncell1 + 255) / 256, a.tt.n), dim3(256), 0, s, a.grid_scan, a.d_tt, a.cap, a.ncell1, a.cell_start,
                     a.d_counts, a.d_n_cent + 5, a.grid);
  hipLaunchKernelGGL(cell_place_kernel, grid_for(total, 256), dim3(256), 0, s, a.keys0, a.vals1, a.d_n_cent, a.grid_scan, a.cent, a.d_tt, a.ncell1,
                     a.inv_leaf, a.spts, a.flags, a.d_n_cent + 5);
  hipLaunchKernelGGL(cell_rank_kernel, grid_for(total, 256), dim3(256), 0, s, a.keys0, a.d_n_cent, a.grid_scan, a.cent, a.spts, a.flags, a.d_tt,
                     a.cap, a.ncell1, a.inv_leaf, a.pool, a.vals1, a.d_n_cent + 5);
}
// The reference's own accumulation of the auto-voxel statistic (laserMapping.cpp:604-611): Eigen::Vector3f average, one float
// addition per point and axis IN INPUT ORDER.  A sequential float sum cannot be re-associated without changing its roundings,
// so this is one wavefront whose lanes 0..2 carry the three running sums while all 64 lanes fetch the next 64 points into LDS
// (~3 ns per point: 0.4 ms for a 131 072-point sweep).  Launched only when the fp64 tree statistic lies so close to a
// threshold (25 / 65) that the rounding of the float sums could decide (icp_context.cpp: so_icp_prefilter_scan).
__global__ __launch_bounds__(64) void vg_stats_inorder_kernel(const float* __restrict__ xyz, uint32_t n, uint32_t stride_floats,
                                                              float* __restrict__ out3) {
  __shared__ float buf[3][64];
  const int lane = threadIdx.x;
  float sum = 0.f;
  for (uint32_t base = 0; base < n; base += 64) {
    const uint32_t i = base + (uint32_t)lane;
    float x = 0.f, y = 0.f, z = 0.f;
    if (i < n) { const float* p = xyz + (size_t)i * stride_floats; x = p[0]; y = p[1]; z = p[2]; }
    __builtin_amdgcn_wave_barrier();
    buf[0][lane] = fabsf(x); buf[1][lane] = fabsf(y); buf[2][lane] = fabsf(z);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const uint32_t m = n - base < 64u ? n - base : 64u;
    if (lane < 3)
      for (uint32_t k = 0; k < m; ++k) sum += buf[lane][k];  // (-ffp-contract=off: a plain float addition, like the reference's)
  }
  if (lane < 3) out3[lane] = sum;
}
void launch_vg_stats_inorder(const float* d_xyz, uint32_t n, uint32_t stride_floats, float* d_out3, hipStream_t s) {
  hipLaunchKernelGGL(vg_stats_inorder_kernel, dim3(1), dim3(64), 0, s, d_xyz, n, stride_floats, d_out3);
}
void launch_vg_stats(const float* d_xyz, uint32_t n, uint32_t stride_floats, double* d_part, int blocks, hipStream_t s) {
  hipLaunchKernelGGL(vg_stats_kernel, dim3(blocks), dim3(256), 0, s, d_xyz, n, stride_floats, d_part);
}
void launch_voxel_filter(const VoxelFilterArgs& a, hipStream_t s) {
  const uint32_t n = a.n;
  hipLaunchKernelGGL(vg_keys_kernel, grid_for(n, 256), dim3(256), 0, s, a.d_xyz, n, a.stride_floats, a.inv_leaf, a.min_b[0], a.min_b[1],
                     a.min_b[2], a.div_b[0], a.div_b[0] * a.div_b[1], a.wpts, a.keys0, a.vals0, a.d_decision);
  size_t tb = a.temp_bytes;
  (void)map_sort(a.temp, tb, a.keys0, a.keys1, a.vals0, a.vals1, (size_t)n, 32, s);  // stable: input order inside a leaf
  if (a.scan_state && (n + kHeadsItems - 1u) / kHeadsItems <= a.n_scan_state) {
    hipLaunchKernelGGL(leaf_heads_scan_kernel, dim3((n + kHeadsItems - 1u) / kHeadsItems), dim3(256), 0, s, a.keys1, a.vals1, n, a.wpts, a.spts, a.heads,
                       a.d_n_cent, a.scan_state, a.d_n_cent + 8);
  } else {
    hipLaunchKernelGGL(leaf_flags_kernel, grid_for(n, 256), dim3(256), 0, s, a.keys1, n, a.flags);
    tb = a.temp_bytes;
    (void)rocprim::exclusive_scan(a.temp, tb, a.flags, a.pos, 0u, (size_t)n, rocprim::plus<uint32_t>(), s);
    hipLaunchKernelGGL(leaf_heads_kernel, grid_for(n, 256), dim3(256), 0, s, a.keys1, a.vals1, a.flags, a.pos, n, a.wpts, a.spts, a.heads,
                       a.d_n_cent);
  }
  hipLaunchKernelGGL(vg_centroid_kernel, grid_for(n, 256), dim3(256), 0, s, a.heads, a.d_n_cent, a.spts, a.d_out, a.flags, a.d_n_cent + 1,
                     const_cast<VgDecision*>(a.d_decision));
  hipLaunchKernelGGL(vg_centroid_long_kernel, dim3(3 * kMaxLongLeaves / 4), dim3(256), 0, s, a.heads, a.spts, a.d_out, a.flags, a.d_n_cent + 1);
}
// ---- featureExtraction::removePointDistortion (featureExtraction.cpp:223-314): SURVEY 8(f) row f4, the step that produces
// the cloud the feature extraction (and through it this path) consumes.  One thread per point: pose of the stamped-pose
// buffer interpolated at the point's own time, T_final = [T_l_i] T_w_original^-1 T_w_current [T_i_l], x y z rewritten in
// place.  The buffer (tens of entries for a 0.1 s sweep against a 200 Hz IMU) sits in LDS; every thread runs its own
// upper_bound over it.  32 B read + 12 B written per point, ~1.5 k fp64 operations: bound by neither at these sizes.
constexpr uint32_t kDeskewLdsPoses = 512;
template <bool LDS>
__global__ __launch_bounds__(256) void deskew_kernel(uint8_t* __restrict__ pts, uint32_t n, uint32_t stride, uint32_t time_off, double t0,
                                                     const double* __restrict__ poses, uint32_t n_poses, DeskewFrames f,
                                                     uint32_t* __restrict__ n_clamped) {
  __shared__ double tab_lds[LDS ? kDeskewLdsPoses * kStampedPoseDoubles : 1];
  if (LDS) {
    for (uint32_t k = threadIdx.x; k < n_poses * kStampedPoseDoubles; k += blockDim.x) tab_lds[k] = poses[k];
    __syncthreads();
  }
  const double* tab = LDS ? tab_lds : poses;
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  bool clamped = false;
  if (i < n) {
    uint8_t* rec = pts + (size_t)i * stride;
    float* xyz = reinterpret_cast<float*>(rec);
    const float x = xyz[0], y = xyz[1], z = xyz[2];
    if (isfinite(x) && isfinite(y) && isfinite(z)) {  // :293-295
      const double ts = (double)*reinterpret_cast<const float*>(rec + time_off) + t0;  // :297
      const Rigid T = deskew_transform(tab, n_poses, ts, f, &clamped);
      double ox, oy, oz;
      quat_rotate<double>(T.q, (double)x, (double)y, (double)z, ox, oy, oz);  // Twist::operator*(Tangent3): rot * p + pos
      xyz[0] = (float)(ox + T.t[0]); xyz[1] = (float)(oy + T.t[1]); xyz[2] = (float)(oz + T.t[2]);
    }
  }
  const unsigned long long m = __ballot(clamped);
  if (m && (threadIdx.x & 63) == (uint32_t)__builtin_ctzll(m)) atomicAdd(n_clamped, (uint32_t)__popcll(m));
}

void launch_deskew(uint8_t* d_pts, uint32_t n, uint32_t stride, uint32_t time_off, double t0, const double* d_poses, uint32_t n_poses,
                   const DeskewFrames& f, uint32_t* d_n_clamped, hipStream_t s) {
  if (!n) return;
  const uint32_t blocks = (n + 255) / 256;
  if (n_poses <= kDeskewLdsPoses) deskew_kernel<true><<<blocks, 256, 0, s>>>(d_pts, n, stride, time_off, t0, d_poses, n_poses, f, d_n_clamped);
  else deskew_kernel<false><<<blocks, 256, 0, s>>>(d_pts, n, stride, time_off, t0, d_poses, n_poses, f, d_n_clamped);
}

// utils::pointAssociateToMap over a cloud as laserMapping::publishTopic applies it to the full-resolution scan
// (laserMapping.cpp:464-493, superodom_utils.cpp:148-158): a point within 0.1 m of the sensor stays as it is, every other
// point becomes q * p + t (Eigen's quaternion * vector in fp64, rounded to float); keep[i] = the result lies farther
// than 0.1 m from the world origin (the node drops the others from the published cloud).
__global__ __launch_bounds__(256) void transform_cloud_kernel(uint8_t* __restrict__ pts, uint32_t n, uint32_t stride, Pose pose, uint8_t* __restrict__ keep,
                                                              uint32_t* __restrict__ n_kept) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  bool k = false;
  if (i < n) {
    float* xyz = reinterpret_cast<float*>(pts + (size_t)i * stride);
    float x = xyz[0], y = xyz[1], z = xyz[2];
    if (!(x * x + y * y + z * z < 0.01)) {  // (float products and sums compared with the double 0.01, as the node writes it)
      double wx, wy, wz;
      quat_rotate<double>(pose.q, (double)x, (double)y, (double)z, wx, wy, wz);
      x = (float)(wx + pose.t[0]); y = (float)(wy + pose.t[1]); z = (float)(wz + pose.t[2]);
      xyz[0] = x; xyz[1] = y; xyz[2] = z;
    }
    k = x * x + y * y + z * z > 0.01;
    keep[i] = k ? 1 : 0;
  }
  const unsigned long long m = __ballot(k);
  if (m && (threadIdx.x & 63) == (uint32_t)__builtin_ctzll(m)) atomicAdd(n_kept, (uint32_t)__popcll(m));
}
void launch_transform_cloud(uint8_t* d_pts, uint32_t n, uint32_t stride, const Pose& pose, uint8_t* d_keep, uint32_t* d_n_kept, hipStream_t s) {
  if (!n) return;
  hipLaunchKernelGGL(transform_cloud_kernel, grid_for(n, 256), dim3(256), 0, s, d_pts, n, stride, pose, d_keep, d_n_kept);
}

void launch_gather_export_records(const float4* pool, uint32_t cap, uint32_t slot, uint32_t count, uint32_t* d_out_words, uint32_t stride_words, hipStream_t s) {
  if (!count) return;
  const uint64_t words = (uint64_t)count * stride_words;
  hipLaunchKernelGGL(gather_export_records_kernel, dim3((uint32_t)((words + 255u) / 256u)), dim3(256), 0, s, pool, cap, slot, count, d_out_words, stride_words);
}
void launch_gather_export(const float4* pool, uint32_t cap, uint32_t slot, uint32_t count, float* d_out, hipStream_t s) {
  if (!count) return;
  hipLaunchKernelGGL(gather_export_kernel, grid_for(count, 256), dim3(256), 0, s, pool, cap, slot, count, d_out);
}

}  // namespace soicp
