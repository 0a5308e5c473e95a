// kernels.hip -- gfx950 (MI355X / CDNA4) kernels of the scan-to-map ICP hot path.
//
//   scan_keys_kernel    registration prologue, sampling rule, world transform, spatial key (cube | half-cell octant) and
//                       the hash-binning count of every query                     (LidarSlam.cpp:346-359, 397-398)
//   bin_offsets_kernel  bucket offsets + the k-NN work lists (normal / light chunks); bin_place_kernel: binned SoA scan
//   knn_plane_kernel    one wavefront per chunk (four light chunks per wavefront, one per row of 16 lanes): cube-restricted
//                       exact 5-NN over the hashed-voxel cell grid (LDS-staged candidate tiles, selection network, exact
//                       re-rank + certification), distance gate
//                                                                                  (LocalMap.h:481-525, LidarSlam.cpp:720-747)
//   solve_kernel        ONE persistent launch per outer iteration: plane fit (PCA gate, 5x3 LS plane, inlier gate,
//                       coefficient, observability labels; LidarSlam.cpp:514-693) + every LM evaluation (residual,
//                       Tukey x coefficient weight, 6-DoF Jacobian, the 21+6+1+1 fp64 normal-equation sums;
//                       lidarOptimization.cpp:55-80) + the Ceres-equivalent LM controller (lm_solver.h), with
//                       device-side hand-offs between the passes; with a sharded map (N > 1) the ranks' launches trade their
//                       records through peer-mapped inboxes (EvalParams::peer_inbox) and every rank runs the controller
//   eval_kernel / lm_step_kernel   the same passes as one launch per evaluation (sharded map without the peer exchange: the
//                       sums pass through a collective; concurrent hypotheses of so_icp_register_batch)
//   knn_only / knn_fallback   Seam B (LocalMap::nearestKSearchSurf, LocalMap.h:481-525)
//
// Nothing here is a dense contraction, so no MFMA: these are HBM/latency-bound gather-scan-reduce
// kernels (64-wide wavefronts, fp64 VALU for the parts the reference computes in double).
// Paths in citations are relative to /root/reference/super_odometry/{src,include/super_odometry}.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <cstddef>
#include <cstdlib>
#include <cstring>
#include <type_traits>

#ifdef SO_LM_STAMPS  // profiling build: device clock at the phases of the LM controller (tools/eval_stamps.py prints them)
#define SO_LM_STAMP(dbg, i) do { if (dbg) (dbg)[i] = wall_clock64(); } while (0)
#endif
#include "kernels.h"
#include "plane_fit.h"

namespace soicp {

#define SO_MATCH_SUCCESS 0
#define SO_MATCH_NOT_ENOUGH 1
#define SO_MATCH_TOO_FAR 2
#define SO_MATCH_BAD_PCA 3
#define SO_MATCH_INVALID 4
#define SO_MATCH_MSE 5
#define SO_MATCH_DROPPED 254  // not sampled / not owned by this rank: never counted, never evaluated
#define SO_MATCH_PENDING 255  // k-NN done, plane fit not yet (internal hand-off between the two kernels)
static_assert(SO_FIT_SUCCESS == SO_MATCH_SUCCESS && SO_FIT_BAD_PCA == SO_MATCH_BAD_PCA && SO_FIT_INVALID == SO_MATCH_INVALID && SO_FIT_MSE == SO_MATCH_MSE,
              "plane_fit.h returns MatchingResult codes");

// ------------------------------------------------------------------------------------------------
// map addressing
// ------------------------------------------------------------------------------------------------
struct CellRef {
  int slot;        // -1: outside window / no tree
  int cx, cy, cz;  // cell inside the cube
};

// int((c + 25.0) / 50.0) for a FLOAT-valued c without the fp64 division: c + 25.0 is exact in double and, unless it
// is an exact multiple of 50, differs from one by at least a float ulp (>= 2^-24 relative), far more than the 2^-53
// error of multiplying by 0.02; on exact multiples the product rounds to the same integer side.  So the truncation
// is identical to the reference's division for every float input (LocalMap.h:488-497).
__device__ __forceinline__ int cube_coord_f(float c, int origin) {
  const double s = (double)c + 25.0;
  int i = (int)(s * 0.02) + origin;
  if (s < 0) i--;
  return i;
}

// cube index exactly as LocalMap::nearestKSearchSurf (LocalMap.h:488-507); then the cell of the
// hashed-voxel grid inside that cube.
__device__ __forceinline__ CellRef locate(const DevMapView& m, float qx, float qy, float qz, int* wcube = nullptr) {
  CellRef r;
  const int ci = cube_coord_f(qx, m.origin[0]);
  const int cj = cube_coord_f(qy, m.origin[1]);
  const int ck = cube_coord_f(qz, m.origin[2]);
  r.slot = -1; r.cx = r.cy = r.cz = 0;
  if (!(ci >= 0 && ci < 21 && cj >= 0 && cj < 21 && ck >= 0 && ck < 11)) return r;
  r.slot = m.cube_slot[ci + 21 * cj + 21 * 21 * ck];
  const int w0 = ci - m.origin[0], w1 = cj - m.origin[1], w2 = ck - m.origin[2];
  if (wcube) { wcube[0] = w0; wcube[1] = w1; wcube[2] = w2; }
  const double mn0 = w0 * 50.0 - 25.0, mn1 = w1 * 50.0 - 25.0, mn2 = w2 * 50.0 - 25.0;
  int cx = (int)floor(((double)qx - mn0) * m.inv_cell);
  int cy = (int)floor(((double)qy - mn1) * m.inv_cell);
  int cz = (int)floor(((double)qz - mn2) * m.inv_cell);
  const int nc1 = m.nc - 1;
  r.cx = cx < 0 ? 0 : (cx > nc1 ? nc1 : cx);
  r.cy = cy < 0 ? 0 : (cy > nc1 ? nc1 : cy);
  r.cz = cz < 0 ? 0 : (cz > nc1 ? nc1 : cz);
  return r;
}

// ------------------------------------------------------------------------------------------------
// scan preparation
// ------------------------------------------------------------------------------------------------
// BATCH instantiations (so_icp_register_batch: B hypotheses of ONE scan in the same launches): blockIdx.y picks the
// hypothesis h = active[blockIdx.y]; every per-registration array is laid out with the common element stride bv.bs
// (kernels.h: BatchView).  The single-registration instantiations compile to the code they were before.

// registration prologue: pose <- host-provided guess (kernel arguments: no H2D copy), counters and histograms cleared
// (the guess: a.pose, or for a chained registration DevState::T_chain, which no prologue writes)
__device__ __forceinline__ void reg_begin_state(DevState* st, const RegBeginArgs& a, int tid) {
  if (tid < 7) {
    double v = a.pose[tid];
    if (a.chain_expect) v = st->T_chain[tid];
    st->pose_in[tid] = v; st->T[tid] = v; st->eval_pose[tid] = v;
  }
  if (tid == 0) {
    st->max_outer = a.max_outer; st->lm_max = a.lm_max;
    st->outer_iter = 0; st->reg_done = 0; st->lm_more = 0; st->n_iterations = 0;
    st->bin_packed = 0ull;
    // (DevState::packed_leftover is NOT cleared here: when the prologue rides on the first k-NN launch, that launch's packed wavefronts add
    //  to it, and a clear ordered only by dispatch order could lose counts from run to run -- ADVICE r05.  It runs on; the host takes differences.)
  }
}
// stand-alone prologue (empty scan: scan_keys_kernel, which normally carries it, is not launched)
__global__ __launch_bounds__(512) void reg_begin_kernel(DevState* st, RegBeginArgs a, int32_t* __restrict__ hist) {
  hist[threadIdx.x] = 0;  // kHistReplicas * kHistStride ints
  reg_begin_state(st, a, threadIdx.x);
}
static_assert(kHistReplicas * kHistStride == 512, "reg_begin_kernel clears one histogram word per thread");

// calculateSamplingRate / shouldProcessPoint (LidarSlam.cpp:346-359) for point gi of a scan of ntot points (0: rate 0, every point dropped)
__device__ __forceinline__ bool sampling_keeps(uint32_t gi, uint32_t ntot, int max_surface_features) {
  if (max_surface_features >= 0 && ntot > (uint32_t)max_surface_features) {
    const double rate = 1.0 * max_surface_features / ntot;
    const double rem = fmod((double)gi * rate, 1.0);
    if (rem + 0.001 > rate) return false;
  }
  return true;
}
// Prologue of a registration whose scan was binned ahead (scan_keys_kernel with prebin_ctr, so_icp_stage_scan): the guess and the
// loop bounds, the work-list counters the binning left in *ctr, and -- when the sampling rule drops points -- their status bytes
// (the rule does not depend on the pose).  One launch instead of scan_keys -> bin_offsets -> bin_place on the registration's path.
__global__ __launch_bounds__(256) void reg_begin_prebinned_kernel(DevState* st, RegBeginArgs a, int32_t* __restrict__ hist,
                                                                  const unsigned long long* __restrict__ ctr, uint8_t* __restrict__ status,
                                                                  uint32_t n, int max_surface_features) {
  if (blockIdx.x == 0) {
    hist[threadIdx.x] = 0; hist[256 + threadIdx.x] = 0;
    reg_begin_state(st, a, threadIdx.x);
    if (threadIdx.x == 0) st->bin_packed = *ctr;  // (the same thread cleared it in reg_begin_state)
  }
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && !sampling_keeps(i, n, max_surface_features)) status[i] = SO_MATCH_DROPPED;
}

// (workgroup 0 also runs the registration prologue: one launch less per registration)
template <bool BATCH>
__global__ __launch_bounds__(256) void scan_keys_kernel(const float* __restrict__ scan, uint32_t n,
                                                        DevState* __restrict__ st, RegBeginArgs a, int32_t* __restrict__ hist,
                                                        DevMapView map,
                                                        int max_surface_features, int rank, int world,
                                                        uint32_t* __restrict__ keys, uint32_t* __restrict__ vals,
                                                        uint8_t* __restrict__ status, BinTable bt, int rebin, BatchView bv,
                                                        int qsplit, uint32_t n_total, unsigned long long* __restrict__ prebin_ctr) {
  if (BATCH) {
    const uint32_t h = bv.active[blockIdx.y];
    st += h; hist += (size_t)h * (kHistReplicas * kHistStride);
    keys += (size_t)h * bv.bs; vals += (size_t)h * bv.bs; status += (size_t)h * bv.bs;
    bt.key += (size_t)h * bv.table_stride; bt.cnt += (size_t)h * bv.table_stride;
  }
  const RegBeginArgs* ab = BATCH ? bv.begin + bv.active[blockIdx.y] : nullptr;  // (a batch reads its prologue arguments from memory)
  // rebin (sharded map, outer iteration >= 1): ownership and binning are re-derived under the CURRENT pose -- a query that
  // the pose update carried out of its owner's halo (1 degree at 50 m is more than a cell) is handed to its new owner, so
  // every rank searches only queries whose whole gate ball lies inside its shard.  No prologue; a no-op once converged.
  // prebin_ctr (so_icp_stage_scan, single device): the scan is binned AHEAD of its registration, on the copy queue, beside the
  // registration in flight -- under that registration's guess (`a.pose`; a chunk stays spatially compact under the small rigid
  // motion to the scan's own guess, exactly like the second sweep of any registration, whose pose has moved since the binning).
  // Nothing of the registration in flight is touched: no prologue, no status bytes, the work-list counters in *prebin_ctr;
  // reg_begin_prebinned_kernel adopts them when the scan's own registration starts.
  if (!BATCH && prebin_ctr) {
    if (blockIdx.x == 0 && threadIdx.x == 0) *prebin_ctr = 0ull;  // (bin_offsets adds to it)
  } else if (rebin) {
    if (st->reg_done) return;
    if (blockIdx.x == 0 && threadIdx.x == 0) st->bin_packed = 0ull;  // (bin_offsets of this round adds to it)
  } else if (blockIdx.x == 0) {
    hist[threadIdx.x] = 0; hist[256 + threadIdx.x] = 0;
    if (BATCH) reg_begin_state(st, *ab, threadIdx.x); else reg_begin_state(st, a, threadIdx.x);
  }
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const Pose pose = pose_from_array(rebin ? st->T : (BATCH ? ab->pose : a.pose));
  // key = (cube slot << 21) | Morton(half-cell: 7 bits per axis, low 3 bits = octant inside the map cell).  The two highest
  // values of the slot field are special: slot_lim = processed query whose cube is outside the window / has no tree
  // (NOT_ENOUGH_NEIGHBORS), slot_lim + 1 = query not sampled / not owned by this rank (dropped).  With more than 2046
  // occupied cubes the slot does not fit above 21 cell bits: the key falls back to whole cells.  A key only GROUPS queries
  // (every query is located again under the current pose by the k-NN sweep), so a real slot is clamped below the specials: a
  // scan binned ahead of its registration may read a slot the map gained after `map` was snapshotted.
  const int cell_bits = (map.n_slots + 2u <= 2048u) ? 21 : 18;
  const uint32_t slot_lim = (1u << (32 - cell_bits)) - 2u;
  const uint32_t kDropped = (slot_lim + 1u) << cell_bits, kNoCube = slot_lim << cell_bits;
  uint32_t key = kDropped;
  bool process = true;
  // qsplit (N > 1, map replicated, QUERIES split): `scan` is this rank's share of the scan -- its 64-point segments rank, rank +
  // world, ... gathered into one array (icp_context.cpp) --, every query of it is this rank's, and the sampling rule runs on the
  // point's index in the WHOLE scan of n_total points
  const uint32_t gi = qsplit ? ((((i >> 6) * (uint32_t)world + (uint32_t)rank) << 6) + (i & 63u)) : i;
  const uint32_t ntot = qsplit ? n_total : n;
  const bool own_all = qsplit || world <= 1;
  if (!sampling_keeps(gi, ntot, max_surface_features)) process = false;
  if (process) {
    const float px = scan[3 * i], py = scan[3 * i + 1], pz = scan[3 * i + 2];
    double wx, wy, wz;
    quat_rotate<double>(pose.q, (double)px, (double)py, (double)pz, wx, wy, wz);
    const float qx = (float)(wx + pose.t[0]), qy = (float)(wy + pose.t[1]), qz = (float)(wz + pose.t[2]);
    int w[3];
    const CellRef c = locate(map, qx, qy, qz, w);
    if (c.slot < 0) {
      key = (own_all || rank == 0) ? kNoCube : kDropped;  // counted once (NOT_ENOUGH_NEIGHBORS): by rank 0 when the MAP is sharded
    } else {
      int owner = rank;
      if (!own_all) owner = (int)(brick_hash(w[0], w[1], w[2], c.cx / kBrickCells, c.cy / kBrickCells, c.cz / kBrickCells) % (uint32_t)world);
      const uint32_t kslot = min((uint32_t)c.slot, slot_lim - 1u);
      if (owner == rank && cell_bits == 18) key = (kslot << 18) | morton3((uint32_t)c.cx, (uint32_t)c.cy, (uint32_t)c.cz);
      else if (owner == rank) {
        // Morton code of the HALF-cell: its three low bits are the octant of the cell the query sits in, so that a
        // chunk (one key) is one octant and the near pass of the k-NN kernel needs a 2x2x2 block of cells
        const double mn0 = w[0] * 50.0 - 25.0, mn1 = w[1] * 50.0 - 25.0, mn2 = w[2] * 50.0 - 25.0;
        const int hmax = 2 * map.nc - 1;
        int hx = (int)floor(((double)qx - mn0) * map.inv_cell * 2.0), hy = (int)floor(((double)qy - mn1) * map.inv_cell * 2.0);
        int hz = (int)floor(((double)qz - mn2) * map.inv_cell * 2.0);
        hx = min(max(hx, 2 * c.cx), min(2 * c.cx + 1, hmax)); hy = min(max(hy, 2 * c.cy), min(2 * c.cy + 1, hmax));
        hz = min(max(hz, 2 * c.cz), min(2 * c.cz + 1, hmax));
        key = (kslot << 21) | morton3((uint32_t)hx, (uint32_t)hy, (uint32_t)hz);
      }
    }
  }
  if (key == kDropped && status) status[i] = SO_MATCH_DROPPED;  // every other query gets its status from the k-NN sweep
  // ---- hash binning: claim / find the key's table slot, then count the query in (one atomic per distinct key of the
  //      wavefront: consecutive scan points are neighbours in space, a wavefront holds a handful of keys)
  const bool kept = key != kDropped;
  const int lane = threadIdx.x & 63;
  // group the wavefront's queries by key first: only one lane per distinct key touches the table
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
  uint32_t slot = 0xFFFFFFFFu, base = 0;
  if (kept && lead == lane) {  // claim / find the key's slot (linear probing), then count the group in
    const uint32_t mask = (1u << bt.log2_size) - 1u;
    uint32_t h = (key * 2654435761u) >> (32 - bt.log2_size);
    for (;;) {
      uint32_t k = bt.key[h];  // a stale "empty" only costs the compare-and-swap below; a key, once written, stays
      if (k == 0xFFFFFFFFu) k = atomicCAS(&bt.key[h], 0xFFFFFFFFu, key);
      if (k == 0xFFFFFFFFu || k == key) break;
      h = (h + 1) & mask;
    }
    slot = h;
    base = atomicAdd(&bt.cnt[slot], my_cnt);
  }
  slot = (uint32_t)__shfl((int)slot, lead, 64);
  base = (uint32_t)__shfl((int)base, lead, 64);
  if (!kept) slot = 0xFFFFFFFFu;
  __builtin_nontemporal_store(slot, &keys[i]);            // 0xFFFFFFFF for a dropped query
  __builtin_nontemporal_store(base + my_idx, &vals[i]);   // rank inside the bucket
}

// bucket offsets + chunk lists from the table counts; leaves the table empty for the next registration.
// Four table slots per thread (one 16-byte load), 1024 threads per workgroup: workgroup scan, one atomic triple per workgroup.
// A bucket is cut every 64 queries; a last piece of <= 16 queries is a LIGHT chunk (the k-NN wavefront scans its
// candidates with four parts of its lanes: about half the time of a full chunk) and goes to the second list, which
// grows from the top of the buffer downwards -- the k-NN kernel pairs light chunks so that all chunks run in one round.
template <bool BATCH>
__global__ __launch_bounds__(1024) void bin_offsets_kernel(BinTable bt, uint32_t* __restrict__ chunk_start, uint32_t chunk_cap,
                                                           DevState* __restrict__ st, BatchView bv,
                                                           unsigned long long* __restrict__ packed_ctr /* scan binned ahead: its own counters */) {
  __shared__ uint32_t wq[16], wc[16], wl[16], base_q, base_c, base_l;
  if (BATCH) {
    const uint32_t h = bv.active[blockIdx.y];
    st += h; chunk_start += (size_t)h * bv.bs;
    bt.key += (size_t)h * bv.table_stride; bt.cnt += (size_t)h * bv.table_stride; bt.off += (size_t)h * bv.table_stride;
  }
  const uint32_t t4 = blockIdx.x * blockDim.x + threadIdx.x;  // slots 4*t4 .. 4*t4+3
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint4 c4 = reinterpret_cast<const uint4*>(bt.cnt)[t4];
  const uint32_t cnt[4] = {c4.x, c4.y, c4.z, c4.w};
  uint32_t tq = 0, tc = 0, tl = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const uint32_t r = cnt[k] & 63u;
    tq += cnt[k]; tc += (cnt[k] >> 6) + (r > 16u ? 1u : 0u); tl += (r > 0u && r <= 16u) ? 1u : 0u;
  }
  uint32_t iq = tq, ic = tc, il = tl;  // inclusive scans over the wavefront
  if (__ballot(tq != 0)) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const uint32_t a = (uint32_t)__shfl_up((int)iq, d, 64), b = (uint32_t)__shfl_up((int)ic, d, 64), c = (uint32_t)__shfl_up((int)il, d, 64);
      if (lane >= d) { iq += a; ic += b; il += c; }
    }
  }
  if (lane == 63) { wq[wave] = iq; wc[wave] = ic; wl[wave] = il; }
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t sq = 0, sc = 0, sl = 0;
    for (int w = 0; w < 16; ++w) {
      const uint32_t a = wq[w], b = wc[w], c = wl[w];
      wq[w] = sq; wc[w] = sc; wl[w] = sl; sq += a; sc += b; sl += c;
    }
    // one atomic for the three ranges (21 bits each: scans of fewer than 2^21 points, checked by the host)
    unsigned long long old = 0ull;
    if (sq) old = atomicAdd((!BATCH && packed_ctr) ? packed_ctr : &st->bin_packed, (unsigned long long)sq | ((unsigned long long)sc << 21) | ((unsigned long long)sl << 42));
    base_q = (uint32_t)(old & 0x1FFFFFull); base_c = (uint32_t)((old >> 21) & 0x1FFFFFull); base_l = (uint32_t)(old >> 42);
  }
  __syncthreads();
  if (tq) {
    uint32_t off = base_q + wq[wave] + (iq - tq);
    uint32_t* o = chunk_start + base_c + wc[wave] + (ic - tc);
    uint32_t li = base_l + wl[wave] + (il - tl);  // index in the light list: entry chunk_cap - 1 - li
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (!cnt[k]) continue;
      bt.off[4 * t4 + k] = off;
      const uint32_t full = cnt[k] >> 6, r = cnt[k] & 63u;
      for (uint32_t c2 = 0; c2 < full; ++c2) o[c2] = (off + 64u * c2) | (63u << 26);
      o += full;
      if (r > 16u) *o++ = (off + 64u * full) | ((r - 1u) << 26);
      else if (r > 0u) chunk_start[chunk_cap - 1u - li++] = (off + 64u * full) | ((r - 1u) << 26);
      off += cnt[k];
      bt.key[4 * t4 + k] = 0xFFFFFFFFu;
    }
    reinterpret_cast<uint4*>(bt.cnt)[t4] = make_uint4(0, 0, 0, 0);
  }
}

// queries into their binned positions: one 16-byte record {x, y, z, query index} per position -- one scattered store per query
// where the round-3 layout (three coordinate arrays + the position -> query map) took four, and one load in the k-NN kernel
typedef float f4v __attribute__((ext_vector_type(4)));
template <bool BATCH>
__global__ __launch_bounds__(256) void bin_place_kernel(BinTable bt, const float* __restrict__ scan, uint32_t n,
                                                        const uint32_t* __restrict__ qslot, const uint32_t* __restrict__ qrank,
                                                        float4* __restrict__ binned, const DevState* __restrict__ st_if_rebin, BatchView bv) {
  if (BATCH) {
    const size_t h = bv.active[blockIdx.y];
    bt.off += h * bv.table_stride; qslot += h * bv.bs; qrank += h * bv.bs; binned += h * bv.bs;
  }
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (st_if_rebin && st_if_rebin->reg_done) return;  // re-binning round of a registration that has converged: nothing to place
  const uint32_t sl = qslot[i];
  if (sl == 0xFFFFFFFFu) return;
  const uint32_t pos = bt.off[sl] + qrank[i];
  const f4v rec = {scan[3 * i], scan[3 * i + 1], scan[3 * i + 2], __uint_as_float(i)};
  __builtin_nontemporal_store(rec, reinterpret_cast<f4v*>(binned) + pos);
}

// ------------------------------------------------------------------------------------------------
// exact 5-NN inside the query's cube: candidates = the (clamped) 3x3x3 cell neighbourhood.
// Exactness: one cell >= sqrt(3*planeRes) (local_map.cpp: cells_per_cube), so every map point within
// the reference's acceptance radius lies in that neighbourhood; if the 5th best found is farther than
// the gate the match is rejected either way (LidarSlam.cpp:741).  Keys = (d2 float bits << 32 | index)
// give the total order "ascending d2, ties by ascending canonical index".
// ------------------------------------------------------------------------------------------------
struct Top5 {
  unsigned long long b0, b1, b2, b3, b4;
  __device__ __forceinline__ void init() { b0 = b1 = b2 = b3 = b4 = ~0ull; }
  // five keys at once: 9-comparator sorting network (a third of the instructions of five insertions)
  __device__ __forceinline__ void set5(unsigned long long e0, unsigned long long e1, unsigned long long e2, unsigned long long e3,
                                       unsigned long long e4) {
#define SO_CX(a, b) { const unsigned long long lo = a < b ? a : b, hi = a < b ? b : a; a = lo; b = hi; }
    SO_CX(e0, e1) SO_CX(e3, e4) SO_CX(e2, e4) SO_CX(e2, e3) SO_CX(e1, e4) SO_CX(e0, e3) SO_CX(e0, e2) SO_CX(e1, e3) SO_CX(e1, e2)
#undef SO_CX
    b0 = e0; b1 = e1; b2 = e2; b3 = e3; b4 = e4;
  }
  __device__ __forceinline__ void insert(unsigned long long k) {
    if (k < b4) {
      b4 = k;
      if (b4 < b3) { unsigned long long t = b3; b3 = b4; b4 = t;
        if (b3 < b2) { t = b2; b2 = b3; b3 = t;
          if (b2 < b1) { t = b1; b1 = b2; b2 = t;
            if (b1 < b0) { t = b0; b0 = b1; b1 = t; } } } }
    }
  }
};

// nanoflann::L2Distance::compute, flann/octree.h:93-102: float differences, squares and sum in double
// (std::pow(float,int) promotes), narrowed to float.
__device__ __forceinline__ float l2_d2(float qx, float qy, float qz, float px, float py, float pz) {
  const float dx = qx - px, dy = qy - py, dz = qz - pz;
  return (float)((double)dx * (double)dx + (double)dy * (double)dy + (double)dz * (double)dz);
}

__device__ __forceinline__ uint32_t knn27(const DevMapView& m, const CellRef& c, float qx, float qy, float qz, Top5& top) {
  const uint32_t* tbl = m.cell_start + (size_t)c.slot * m.ncell1;
  const int nc = m.nc;
  const int x0 = c.cx > 0 ? c.cx - 1 : 0, x1 = c.cx < nc - 1 ? c.cx + 1 : nc - 1;
  uint32_t seen = 0;
  for (int dz = -1; dz <= 1; ++dz) {
    const int z = c.cz + dz;
    if (z < 0 || z >= nc) continue;
    for (int dy = -1; dy <= 1; ++dy) {
      const int y = c.cy + dy;
      if (y < 0 || y >= nc) continue;
      const uint32_t* row = tbl + ((size_t)z * nc + y) * nc;
      const uint32_t beg = row[x0], end = row[x1 + 1];
      seen += end - beg;
      for (uint32_t i = beg; i < end; ++i) {
        const float4 p = m.pts[i];
        const float d2 = l2_d2(qx, qy, qz, p.x, p.y, p.z);
        top.insert(((unsigned long long)__float_as_uint(d2) << 32) | i);
      }
    }
  }
  return seen;
}

// ------------------------------------------------------------------------------------------------
// plane fit in registers
// ------------------------------------------------------------------------------------------------
// cyclic Jacobi on a symmetric 3x3 (restates the RESULT of Eigen::SelfAdjointEigenSolver<Matrix3d>,
// utils/superodom_utils.h:150: ascending eigenvalues + eigenvector of the smallest one).
__device__ __forceinline__ void jacobi_rot(double& app, double& aqq, double& apq, double& arp, double& arq,
                                           double& v0p, double& v0q, double& v1p, double& v1q, double& v2p, double& v2q) {
  if (apq == 0.0) return;
  // t = tan(phi) of the annihilating rotation, smaller root: with d = aqq - app, b = 2 apq,
  //   t = sgn(d) b / (|d| + sqrt(d^2 + b^2))   (== sgn(theta) / (|theta| + sqrt(theta^2 + 1)), theta = d / b)
  // one sqrt, one division and one rsqrt per rotation instead of three divisions and two square roots.
  const double d = aqq - app, b2 = 2.0 * apq;
  const double t = (d >= 0 ? b2 : -b2) / (fabs(d) + sqrt(d * d + b2 * b2));
  const double c = rsqrt(t * t + 1.0), s = t * c;
  app -= t * apq; aqq += t * apq; apq = 0.0;
  const double rp = c * arp - s * arq, rq = s * arp + c * arq;
  arp = rp; arq = rq;
  double a, b;
  a = c * v0p - s * v0q; b = s * v0p + c * v0q; v0p = a; v0q = b;
  a = c * v1p - s * v1q; b = s * v1p + c * v1q; v1p = a; v1q = b;
  a = c * v2p - s * v2q; b = s * v2p + c * v2q; v2p = a; v2q = b;
}

__device__ __forceinline__ void eig3_sym(double a00, double a01, double a02, double a11, double a12, double a22,
                                         double ev[3], double nrm[3]) {
  double v00 = 1, v01 = 0, v02 = 0, v10 = 0, v11 = 1, v12 = 0, v20 = 0, v21 = 0, v22 = 1;  // v[row][col]
  for (int sweep = 0; sweep < 16; ++sweep) {
    const double off = a01 * a01 + a02 * a02 + a12 * a12;
    const double dg = a00 * a00 + a11 * a11 + a22 * a22;
    if (off <= 1e-30 * dg || off == 0.0) break;  // off-diagonal below 1e-15 of the diagonal: converged in fp64
    jacobi_rot(a00, a11, a01, a02, a12, v00, v01, v10, v11, v20, v21);  // (p,q)=(0,1), r=2
    jacobi_rot(a00, a22, a02, a01, a12, v00, v02, v10, v12, v20, v22);  // (0,2), r=1
    jacobi_rot(a11, a22, a12, a01, a02, v01, v02, v11, v12, v21, v22);  // (1,2), r=0
  }
  // ascending sort, keep the eigenvector of the smallest eigenvalue
  double e0 = a00, e1 = a11, e2 = a22;
  double n0 = v00, n1 = v10, n2 = v20;  // column 0
  if (e1 < e0 && e1 <= e2) { n0 = v01; n1 = v11; n2 = v21; }
  else if (e2 < e0 && e2 < e1) { n0 = v02; n1 = v12; n2 = v22; }
  double t;
  if (e0 > e1) { t = e0; e0 = e1; e1 = t; }
  if (e1 > e2) { t = e1; e1 = e2; e2 = t; }
  if (e0 > e1) { t = e0; e0 = e1; e1 = t; }
  ev[0] = e0; ev[1] = e1; ev[2] = e2;
  nrm[0] = n0; nrm[1] = n1; nrm[2] = n2;
}

// a / b in ~9 instructions instead of the ~28 of the IEEE sequence: v_rcp_f64, two Newton steps, one residual correction
// (within 1 ulp of the correctly rounded quotient).  The plane-fit pass executes ~26 divisions per query: a third of
// its instructions.  Used only there -- never in the LM controller or the k-NN certification.
__device__ __forceinline__ double fdiv(double a, double b) {
  double r = __builtin_amdgcn_rcp(b);
  r = __builtin_fma(__builtin_fma(-b, r, 1.0), r, r);
  r = __builtin_fma(__builtin_fma(-b, r, 1.0), r, r);
  const double q = a * r;
  return __builtin_fma(__builtin_fma(-b, q, a), r, q);
}

// Same result without iterations (the cyclic Jacobi above costs ~1500 fp64 instructions per query, two thirds of the
// plane-fit pass): the spectrum of a 5-point scatter matrix is lambda0 << lambda1 <= lambda2 for anything that can pass
// the gates, so
//   lambda0    = smallest root of the characteristic cubic by Newton from 0 (monotone from below for a polynomial with
//                real roots; the matrix is first scaled to unit max-norm),
//   lambda1,2  = roots of the deflated quadratic,
//   n          = the largest of the three row cross products of (A - lambda0 I), normalised.
// Eigenvalues agree with the Jacobi values to ~1e-14 relative, the normal to ~1e-15 when lambda0 is separated; only the
// gates (LidarSlam.cpp:772) and the float observability labels consume them.
__device__ __forceinline__ void eig3_sym_direct(double a00, double a01, double a02, double a11, double a12, double a22,
                                                double ev[3], double nrm[3]) {
  const double mx = fmax(fmax(fmax(fabs(a00), fabs(a11)), fabs(a22)), fmax(fmax(fabs(a01), fabs(a02)), fabs(a12)));
  if (!(mx > 0.0)) { ev[0] = ev[1] = ev[2] = 0.0; nrm[0] = 1.0; nrm[1] = 0.0; nrm[2] = 0.0; return; }
  const double is = fdiv(1.0, mx);
  a00 *= is; a01 *= is; a02 *= is; a11 *= is; a12 *= is; a22 *= is;
  // p(l) = -l^3 + c2 l^2 - c1 l + c0
  const double c2 = a00 + a11 + a22;
  const double m00 = a11 * a22 - a12 * a12, m11 = a00 * a22 - a02 * a02, m22 = a00 * a11 - a01 * a01;
  const double c1 = m00 + m11 + m22;
  const double c0 = a00 * m00 - a01 * (a01 * a22 - a12 * a02) + a02 * (a01 * a12 - a11 * a02);
  double l = 0.0;
#pragma unroll 1
  for (int it = 0; it < 60; ++it) {  // 2-3 iterations when lambda0 is separated; linear convergence only towards a double root
    const double f = ((-l + c2) * l - c1) * l + c0;      // p(l)
    const double df = (-3.0 * l + 2.0 * c2) * l - c1;    // p'(l) < 0 left of the smallest root
    if (!(df < 0.0) || !(f > 0.0)) break;   // at (or, by rounding, just past) the root
    const double step = fdiv(f, df);         // < 0: the iterate moves right, never beyond the root (p is convex there)
    l -= step;
    if (!(-step > 4e-16)) break;             // the matrix has unit max-norm: below the noise of p(l)
  }
  if (!(l > 0.0)) l = fmax(l, 0.0);
  // deflate: l1 + l2 = c2 - l, l1 l2 = c1 - l (c2 - l)
  const double sm = c2 - l, pr = c1 - l * sm;
  double disc = sm * sm - 4.0 * pr;
  disc = disc > 0.0 ? sqrt(disc) : 0.0;
  const double l2 = 0.5 * (sm + disc);
  const double l1 = (l2 > 0.0) ? fdiv(pr, l2) : 0.0;  // the smaller root from the product: no cancellation
  ev[0] = l * mx; ev[1] = l1 * mx; ev[2] = l2 * mx;
  // null vector of (A - l I): largest cross product of its rows
  const double r00 = a00 - l, r11 = a11 - l, r22 = a22 - l;
  const double x0 = a01 * a12 - a02 * r11, x1 = a02 * a01 - r00 * a12, x2 = r00 * r11 - a01 * a01;     // row0 x row1
  const double y0 = a01 * r22 - a02 * a12, y1 = a02 * a02 - r00 * r22, y2 = r00 * a12 - a01 * a02;     // row0 x row2
  const double z0 = r11 * r22 - a12 * a12, z1 = a12 * a02 - a01 * r22, z2 = a01 * a12 - r11 * a02;     // row1 x row2
  const double nx = x0 * x0 + x1 * x1 + x2 * x2, ny = y0 * y0 + y1 * y1 + y2 * y2, nz = z0 * z0 + z1 * z1 + z2 * z2;
  double v0 = x0, v1 = x1, v2 = x2, nn = nx;
  if (ny > nn) { v0 = y0; v1 = y1; v2 = y2; nn = ny; }
  if (nz > nn) { v0 = z0; v1 = z1; v2 = z2; nn = nz; }
  if (!(nn > 0.0)) { nrm[0] = 1.0; nrm[1] = 0.0; nrm[2] = 0.0; return; }
  const double inv = rsqrt(nn);
  nrm[0] = v0 * inv; nrm[1] = v1 * inv; nrm[2] = v2 * inv;
}

// least squares A x = -1 (A = 5x3 neighbour coordinates) by column-pivoted Householder QR
// (restates matA0.colPivHouseholderQr().solve(matB0), LidarSlam.cpp:798-806).
__device__ __forceinline__ bool plane_ls5(const float nb[15], double x[3]) {
  double A[3][5], b[5];
  int perm[3] = {0, 1, 2};
#pragma unroll
  for (int i = 0; i < 5; ++i) {
    A[0][i] = (double)nb[3 * i]; A[1][i] = (double)nb[3 * i + 1]; A[2][i] = (double)nb[3 * i + 2];
    b[i] = -1.0;
  }
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    double nrm[3] = {0, 0, 0};
#pragma unroll
    for (int j = k; j < 3; ++j) {
      double s = 0;
#pragma unroll
      for (int i = k; i < 5; ++i) s += A[j][i] * A[j][i];
      nrm[j] = s;
    }
    int piv = k;
    double best = nrm[k];
#pragma unroll
    for (int j = k + 1; j < 3; ++j)
      if (nrm[j] > best) { best = nrm[j]; piv = j; }
#pragma unroll
    for (int j = k + 1; j < 3; ++j)
      if (piv == j) {
#pragma unroll
        for (int i = 0; i < 5; ++i) { const double t = A[k][i]; A[k][i] = A[j][i]; A[j][i] = t; }
        const int t = perm[k]; perm[k] = perm[j]; perm[j] = t;
      }
    double alpha = sqrt(best);
    if (alpha != 0.0) {
      if (A[k][k] > 0) alpha = -alpha;
      double v[5];
      double vn2 = 0;
#pragma unroll
      for (int i = k; i < 5; ++i) v[i] = A[k][i];
      v[k] -= alpha;
#pragma unroll
      for (int i = k; i < 5; ++i) vn2 += v[i] * v[i];
      if (vn2 != 0.0) {
#pragma unroll
        for (int j = k + 1; j < 3; ++j) {
          double dot = 0;
#pragma unroll
          for (int i = k; i < 5; ++i) dot += v[i] * A[j][i];
          const double f = fdiv(2.0 * dot, vn2);
#pragma unroll
          for (int i = k; i < 5; ++i) A[j][i] -= f * v[i];
        }
        double dot = 0;
#pragma unroll
        for (int i = k; i < 5; ++i) dot += v[i] * b[i];
        const double f = fdiv(2.0 * dot, vn2);
#pragma unroll
        for (int i = k; i < 5; ++i) b[i] -= f * v[i];
        A[k][k] = alpha;
      }
    }
  }
  const double y2 = fdiv(b[2], A[2][2]);
  const double y1 = fdiv(b[1] - A[2][1] * y2, A[1][1]);
  const double y0 = fdiv(b[0] - A[1][0] * y1 - A[2][0] * y2, A[0][0]);
#pragma unroll
  for (int a = 0; a < 3; ++a) x[a] = (perm[0] == a) ? y0 : ((perm[1] == a) ? y1 : y2);
  return isfinite(x[0]) && isfinite(x[1]) && isfinite(x[2]);
}

// FeatureObservabilityAnalysis, LidarSlam.cpp:574-693: float arithmetic on float-cast inputs; returns
// the three labels the histogram counts (rot#1, rot#2, trans#1; LidarSlam.cpp:336-339).
__device__ __forceinline__ void observability(const double pw[3], const double ev[3], const double nrm[3], const Pose& pose,
                                              int& o0, int& o1, int& o2) {
  const float px = (float)pw[0], py = (float)pw[1], pz = (float)pw[2];
  const float nx = (float)nrm[0], ny = (float)nrm[1], nz = (float)nrm[2];
  const double l1 = sqrt(ev[2]), l2 = sqrt(ev[1]), l3 = sqrt(ev[0]);
  const double planar_2 = fdiv(l2 - l3, l1);
  const float qf[4] = {(float)pose.q[0], (float)pose.q[1], (float)pose.q[2], (float)pose.q[3]};
  float ax[3][3];
  quat_rotate<float>(qf, 1.f, 0.f, 0.f, ax[0][0], ax[0][1], ax[0][2]);
  quat_rotate<float>(qf, 0.f, 1.f, 0.f, ax[1][0], ax[1][1], ax[1][2]);
  quat_rotate<float>(qf, 0.f, 0.f, 1.f, ax[2][0], ax[2][1], ax[2][2]);
  const float cx = py * nz - pz * ny, cy = pz * nx - px * nz, cz = px * ny - py * nx;
  float rot[6], tr[3];
  const float psq = (float)(planar_2 * planar_2);
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const float v = cx * ax[a][0] + cy * ax[a][1] + cz * ax[a][2];
    rot[2 * a] = v; rot[2 * a + 1] = -v;
    tr[a] = psq * fabsf(nx * ax[a][0] + ny * ax[a][1] + nz * ax[a][2]);
  }
  // descending order, ties keep the lower label (stable insertion sort in libstdc++ for n < 16)
  int b1 = 0;
#pragma unroll
  for (int a = 1; a < 6; ++a) if (rot[a] > rot[b1]) b1 = a;
  int b2 = -1;
#pragma unroll
  for (int a = 0; a < 6; ++a) if (a != b1 && (b2 < 0 || rot[a] > rot[b2])) b2 = a;
  int t1 = 0;
#pragma unroll
  for (int a = 1; a < 3; ++a) if (tr[a] > tr[t1]) t1 = a;
  o0 = b1; o1 = b2; o2 = 6 + t1;
}

// ComputePlaneDistanceParameters after the neighbour search (LidarSlam.cpp:533-571) with the reference's own algorithms --
// column-pivoted Householder for the plane, optionally cyclic Jacobi for the PCA.  Production runs plane_fit5 (plane_fit.h:
// the closed form of the same least-squares problem); this one is kept behind SOICP_ABLATE = 4096 / 512 (PROF kernels) for A/B
// runs and for the gate-edge test.
__device__ __forceinline__ int plane_from_neighbours(const float nb[15], const double pw[3], const Pose& pose,
                                                     const MatchParams& mp, double nd[4], double& coeff, int obs[3], bool jacobi_eig = false) {
  // PCA (LidarSlam.cpp:756-775, utils/superodom_utils.h:143-151)
  double mx = 0, my = 0, mz = 0;
#pragma unroll
  for (int j = 0; j < 5; ++j) { mx += (double)nb[3 * j]; my += (double)nb[3 * j + 1]; mz += (double)nb[3 * j + 2]; }
  mx = fdiv(mx, 5.0); my = fdiv(my, 5.0); mz = fdiv(mz, 5.0);
  double s00 = 0, s01 = 0, s02 = 0, s11 = 0, s12 = 0, s22 = 0;
#pragma unroll
  for (int j = 0; j < 5; ++j) {
    const double a = (double)nb[3 * j] - mx, b = (double)nb[3 * j + 1] - my, c = (double)nb[3 * j + 2] - mz;
    s00 += a * a; s01 += a * b; s02 += a * c; s11 += b * b; s12 += b * c; s22 += c * c;
  }
  double ev[3], nrm[3];
  if (jacobi_eig) eig3_sym(s00, s01, s02, s11, s12, s22, ev, nrm);  // test switch (SOICP_ABLATE=512, PROF instantiation): the iterative reference solver
  else eig3_sym_direct(s00, s01, s02, s11, s12, s22, ev, nrm);
  if (ev[0] < 1e-6 || fdiv(ev[1], ev[2]) < 0.1) return SO_MATCH_BAD_PCA;  // LidarSlam.cpp:772
  double x[3];
  if (!plane_ls5(nb, x)) return SO_MATCH_INVALID;                    // LidarSlam.cpp:809-812
  const double nn = sqrt(x[0] * x[0] + x[1] * x[1] + x[2] * x[2]);
  const double d = fdiv(1.0, nn);                                    // LidarSlam.cpp:815
  const double n0 = fdiv(x[0], nn), n1 = fdiv(x[1], nn), n2 = fdiv(x[2], nn);  // LidarSlam.cpp:816
  double sum = 0;
  bool too_far = false;
#pragma unroll
  for (int j = 0; j < 5; ++j) {
    const double dist = fabs(n0 * (double)nb[3 * j] + n1 * (double)nb[3 * j + 1] + n2 * (double)nb[3 * j + 2] + d);
    too_far |= dist > mp.max_point_dist;                             // LidarSlam.cpp:832
    sum += dist;
  }
  if (too_far) return SO_MATCH_MSE;
  const double mean_abs = fdiv(sum, 5.0);
  if (pw[0] * nrm[0] + pw[1] * nrm[1] + pw[2] * nrm[2] < 0) { nrm[0] = -nrm[0]; nrm[1] = -nrm[1]; nrm[2] = -nrm[2]; }  // :553-561
  observability(pw, ev, nrm, pose, obs[0], obs[1], obs[2]);
  coeff = 1.0 - sqrt(fdiv(mean_abs, (double)mp.sq_max_dist_f));           // LidarSlam.cpp:568
  nd[0] = n0; nd[1] = n1; nd[2] = n2; nd[3] = d;
  return SO_MATCH_SUCCESS;
}

// ------------------------------------------------------------------------------------------------
// knn_plane_kernel -- wave-cooperative exact 5-NN (the plane fit follows in eval_kernel<true>).
//
// A wavefront owns one CHUNK of the spatially sorted scan: <= 64 queries that shared one half-cell octant of the map
// grid when the scan was sorted.  The wave forms the union of its lanes' search-ball cell ranges (ballot / readlane,
// wave-uniform), stages that block's points once in an LDS tile (coalesced 16-byte loads, block-local coordinates,
// canonical index alongside) and streams them back as WAVE-UNIFORM broadcast operands: four ds_read_b128 feed four
// candidates to all 64 lanes; the VALU only touches per-lane query data.  Selection is branch-free: a 32-bit key =
// (approximate fp32 d2 with its low 11 mantissa bits replaced by the candidate's position) runs through a sorted
// 8-register network of v_med3_i32 (no divergence, no 64-bit compares).
// Each lane then re-ranks its 8 survivors with the reference's exact arithmetic (octree.h:93-102, fp64 squares narrowed
// to float; ties by canonical index) and CERTIFIES the result: every point that was not re-ranked has exact d2 >= R2,
// R2 = min(squared distance to the faces of the scanned block, 8th key with its index bits cleared minus the fp32 error
// bound); exact 5th distance below R2 => the list is the exact 5-NN.
// Two passes: NEAR (radius = half a cell => 2x2x2 cells for an octant chunk, a third of the candidates) and, only for
// lanes it could not certify, FULL (the reference's gate radius sqrt(3*planeRes), where "not inside the gate ball" is a
// certain TOO_FAR).  What is still uncertified (8 near-equidistant candidates, or a block with more than 2048
// candidates) falls back to the per-lane exact scan knn27().  Result: bit-identical neighbour lists to the oracle.
// LIGHT chunks (<= 16 queries: 44 % of the chunks of a 128-beam sweep) are PACKED four to a wavefront, one per row of 16
// lanes (round 4, SO_KNN_PACK / MatchParams::pack_light): the same two passes with the block bounds, the row table and the
// candidate count of every row in VGPRs (DPP row all-reduces), a quarter of the tile per row, the full pass staged by all 64
// lanes for one row after the other, and a wave-cooperative exact scan for what is left.  6.65 M -> 5.00 M VALU and 2.20 M ->
// 1.57 M SALU wave-instructions per sweep (PMC), same lists bit for bit; the sweep's time did not follow (23 us: it is ended
// by its heaviest single-chunk wavefronts), the batched sweeps' did (+2.4 % batch64).
// ------------------------------------------------------------------------------------------------
constexpr int kKeyIdxBits = 11;                                  // a key addresses up to 2048 candidates of one group
constexpr uint32_t kKeyIdxMask = (1u << kKeyIdxBits) - 1u;
constexpr uint32_t kGroupMaxCand = 1u << kKeyIdxBits;
constexpr uint32_t kTileCand = 384;                               // candidates staged in LDS at a time (7.8 KB per wavefront: 4 workgroups per CU = 132 KB)

// v_med3_i32 has no clang builtin; it is a pure VALU op (no memory, no wait states needed).
__device__ __forceinline__ int32_t imed3(int32_t a, int32_t b, int32_t c) {
  int32_t r;
  asm("v_med3_i32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}

// Keys are float bit patterns compared as SIGNED integers: for non-negative floats that is the float order, and a
// (rounding-induced) slightly negative approximate d2 sorts in front of everything, which is what we want.
constexpr int32_t kKeyEmpty = 0x7FFFFFFF;

struct Net8 {  // ascending: a0 <= a1 <= ... <= a7
  int32_t a0, a1, a2, a3, a4, a5, a6, a7;
  __device__ __forceinline__ void init() { a0 = a1 = a2 = a3 = a4 = a5 = a6 = a7 = kKeyEmpty; }
  __device__ __forceinline__ void push(int32_t k) {  // new a_s = med3(a_{s-1}, a_s, k); a0 = min(a0, k)
    a7 = imed3(a6, a7, k);
    a6 = imed3(a5, a6, k);
    a5 = imed3(a4, a5, k);
    a4 = imed3(a3, a4, k);
    a3 = imed3(a2, a3, k);
    a2 = imed3(a1, a2, k);
    a1 = imed3(a0, a1, k);
    a0 = a0 < k ? a0 : k;
  }
};

// Approximate d2 in BLOCK-LOCAL coordinates (origin = centre of the group's home cell, |coords| <= ~2 m):
//   d2a = (|c|^2 + |q|^2) - 2 q.c      -> 1 add + 3 fma per candidate instead of 3 sub + mul + 2 fma.
// |c|^2 is precomputed at staging time, -2q and |q|^2 once per lane per group.  With magnitudes <= ~8 the fp32
// rounding error is <= ~4e-6 m^2 (absolute); the certification margin below accounts for it (kApproxAbsErr).
constexpr float kApproxAbsErr = 2e-5f;
typedef float float2v __attribute__((ext_vector_type(2)));
// two candidates per instruction: v_pk_add_f32 + 3 x v_pk_fma_f32 (packed fp32 runs at twice the scalar fp32 rate)
__device__ __forceinline__ float2v approx_d2_pair(float2v m2qx, float2v m2qy, float2v m2qz, float2v qq, float2v cx, float2v cy,
                                                  float2v cz, float2v cc) {
  float2v v = cc + qq;
  v = __builtin_elementwise_fma(m2qx, cx, v);
  v = __builtin_elementwise_fma(m2qy, cy, v);
  v = __builtin_elementwise_fma(m2qz, cz, v);
  return v;
}
// key = distance bits where keep_mask is set, the candidate's position elsewhere: ONE v_bfi_b32 with the (wave-uniform)
// position in an SGPR.  (Left to itself the compiler emits v_and + v_add3 per key.)
__device__ __forceinline__ int32_t make_key(float d2a, uint32_t jloc_uniform, uint32_t keep_mask) {
  int32_t r;
  asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(r) : "v"(keep_mask), "v"(d2a), "s"(jloc_uniform));
  return r;
}
// same with a per-lane position (VGPR operand)
__device__ __forceinline__ int32_t make_key_v(float d2a, uint32_t jloc, uint32_t keep_mask) {
  int32_t r;
  asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(r) : "v"(keep_mask), "v"(d2a), "v"(jloc));
  return r;
}
__device__ __forceinline__ int32_t approx_key(float m2qx, float m2qy, float m2qz, float qq, float cx, float cy, float cz,
                                              float cc, uint32_t jloc, uint32_t keep_mask) {
  float v = cc + qq;
  v = __builtin_fmaf(m2qx, cx, v);
  v = __builtin_fmaf(m2qy, cy, v);
  v = __builtin_fmaf(m2qz, cz, v);
  // v_bfi_b32: distance bits where keep_mask is set, the candidate's position in the tile elsewhere
  return (int32_t)((__float_as_uint(v) & keep_mask) | (jloc & ~keep_mask));
}

__device__ __forceinline__ void publish_state_to(DevState* dst, const DevState* st, unsigned long long seq, int tid, int nthreads);  // below

#ifndef SO_KNN_FILTER
#define SO_KNN_FILTER 1  // stage only the candidates within the search radius of the group's bounding box (see knn_plane_kernel)
#endif
// wave-wide minimum of v over the DPP network (row_shr 1, 2, 4, 8 inside the rows of 16, row_bcast:15 / :31 across rows): no LDS,
// the result is wave-uniform (read from lane 63).  Lanes that must not take part pass +inf.
__device__ __forceinline__ float wave_min_f32(float v) {
  const int inf = 0x7F800000;
#define SO_DPP_MIN(ctrl, rows) v = fminf(v, __int_as_float(__builtin_amdgcn_update_dpp(inf, __float_as_int(v), ctrl, rows, 0xF, false)))
  SO_DPP_MIN(0x111, 0xF); SO_DPP_MIN(0x112, 0xF); SO_DPP_MIN(0x114, 0xF); SO_DPP_MIN(0x118, 0xF);
  SO_DPP_MIN(0x142, 0xA); SO_DPP_MIN(0x143, 0xC);
#undef SO_DPP_MIN
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

// All-reduce over a ROW of 16 lanes (DPP row_ror:1,2,4,8 -- rotations inside the row, no LDS): every lane of the row gets
// the row's result.  The packed light chunks of knn_plane_kernel live one per row.
__device__ __forceinline__ int row_min_i32(int v) {
  v = min(v, __builtin_amdgcn_update_dpp(v, v, 0x121, 0xF, 0xF, false)); v = min(v, __builtin_amdgcn_update_dpp(v, v, 0x122, 0xF, 0xF, false));
  v = min(v, __builtin_amdgcn_update_dpp(v, v, 0x124, 0xF, 0xF, false)); v = min(v, __builtin_amdgcn_update_dpp(v, v, 0x128, 0xF, 0xF, false));
  return v;
}
__device__ __forceinline__ int row_max_i32(int v) {
  v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x121, 0xF, 0xF, false)); v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x122, 0xF, 0xF, false));
  v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x124, 0xF, 0xF, false)); v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x128, 0xF, 0xF, false));
  return v;
}
__device__ __forceinline__ float row_min_f32(float v) {
#define SO_ROR_F(ctrl) __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), ctrl, 0xF, 0xF, false))
  v = fminf(v, SO_ROR_F(0x121)); v = fminf(v, SO_ROR_F(0x122)); v = fminf(v, SO_ROR_F(0x124)); v = fminf(v, SO_ROR_F(0x128));
#undef SO_ROR_F
  return v;
}
__device__ __forceinline__ unsigned long long wave_min_u64(unsigned long long v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const unsigned long long o = __shfl_xor(v, off, 64);
    v = o < v ? o : v;
  }
  return v;
}
#ifndef SO_KNN_PACK
#define SO_KNN_PACK 1  // four light chunks (<= 16 queries each) per wavefront, one per row of 16 lanes (see knn_plane_kernel)
#endif
constexpr uint32_t kPartTile = kTileCand / 4;  // candidates a packed chunk may keep in its quarter of the wavefront's tile

// PROF : the profiling / test-hook instantiation (per-wavefront stamps, SOICP_ABLATE switches, kernel statistics); the
//        production instantiation carries none of it (the sweep is instruction-issue bound).
// BATCH: so_icp_register_batch -- blockIdx.y picks the hypothesis, see BatchView.
// BEGIN: first launch of a registration whose scan was binned ahead (MatchParams::begin): an instantiation of its own, so that the
//        others do not carry the prologue's arguments in their scalar registers (the kernel sits at its register budget).
// BEGIN: 0 = a sweep behind the registration's prologue; 1 = first launch of a registration whose scan was binned ahead (MatchParams::begin);
//        2 = that, for a CHAINED registration (RegBeginArgs::chain_expect): the guess comes from DevState::T_chain
template <bool PROF, bool BATCH, int BEGIN = 0>
__global__ __launch_bounds__(256, 4) void knn_plane_kernel(const float4* __restrict__ binned /* {x, y, z, query index} per binned position */,
                                                        const uint32_t* __restrict__ chunk_start,
                                                        const DevState* __restrict__ st,
                                                        const float4* __restrict__ mpts,
                                                        const uint32_t* __restrict__ mcell_start, DevMapView map,
                                                        MatchParams mp, CorrBuffers corr, uint32_t* __restrict__ nbr5,
                                                        int32_t* __restrict__ hist, BatchView bv) {
  __shared__ int32_t lh[24];
  __shared__ __attribute__((aligned(16))) float tiles[4][4][kTileCand + 16];  // per wavefront: x[], y[], z[], |c|^2 (block-local)
  __shared__ uint32_t tcanon[4][kTileCand + 16];  // per wavefront: canonical map index of the staged candidate
  __shared__ uint32_t rowtab[4][2][72];  // per wavefront: exclusive candidate offsets [33] and first canonical index [32] of the block's x-runs
                                         // (packed light chunks: four tables of 17 + 16 entries, one per row of 16 lanes)
  if (BATCH) {
    const size_t h = bv.active[blockIdx.y];
    st += h; binned += h * bv.bs; chunk_start += h * bv.bs;
    corr.status += h * bv.bs; nbr5 += h * 5 * bv.bs; hist += h * (kHistReplicas * kHistStride);
  }
  uint32_t* const leftover_ctr = (BATCH && mp.packed_leftover)
      ? reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(mp.packed_leftover) + (size_t)bv.active[blockIdx.y] * sizeof(DevState)) : mp.packed_leftover;
  // (MatchParams::begin: first launch of a registration whose scan was binned ahead -- prologue, pose and counters from the arguments)
  constexpr bool begin = BEGIN != 0 && !BATCH;
  constexpr bool chained_begin = BEGIN == 2 && !BATCH;
  if (!begin && st->reg_done) return;  // the registration already converged: this launch is a no-op
  if (!BATCH && mp.chain_expect && st->done_count != mp.chain_expect) return;  // chained registration whose predecessor was not over: no-op
  // BEGIN launch of a chained registration: valid only if the registration in front of it was over, and its guess is what that one's
  // last solve left in DevState::T_chain (= its result o the delta it carried: EvalParams::chain_delta) -- fields no prologue writes,
  // read like any other launch reads st->T
  // (an instantiation of its own: selecting between the two sources at run time cost this kernel, which has no register to spare, 60 - 368
  //  bytes of scratch in every form that was tried)
  if (chained_begin && st->done_count != mp.begin_args.chain_expect) return;
  // the report of the previous outer iteration, left to this launch by its solve (MatchParams::publish_prev)
  if (!BATCH && !begin && mp.publish_prev && blockIdx.x == 0 && st->outer_iter > 0)
    publish_state_to(mp.hring[(st->outer_iter - 1) & 1], st, mp.seq_base | (unsigned long long)st->outer_iter, (int)threadIdx.x, 256);
  // the work-list counters of bin_offsets_kernel: kept queries | normal chunks << 21 | light chunks << 42
  const unsigned long long pk = begin ? *mp.begin_ctr : st->bin_packed;
  if (begin) {
    if (blockIdx.x == 0) {
      hist[threadIdx.x] = 0; hist[256 + threadIdx.x] = 0;
      reg_begin_state(mp.begin_state, mp.begin_args, (int)threadIdx.x);
      if (threadIdx.x == 0) mp.begin_state->bin_packed = pk;
    }
    // the sampling rule's DROPPED status bytes (the rule does not depend on the pose; every other query gets its status from the sweep)
    if (mp.begin_max_surface_features >= 0 && mp.begin_n > (uint32_t)mp.begin_max_surface_features)
      for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < mp.begin_n; i += gridDim.x * 256u)
        if (!sampling_keeps(i, mp.begin_n, mp.begin_max_surface_features)) corr.status[i] = SO_MATCH_DROPPED;
  }
  const uint32_t n_kept = (uint32_t)(pk & 0x1FFFFFull), n_normal = (uint32_t)((pk >> 21) & 0x1FFFFFull), n_light = (uint32_t)(pk >> 42);
  // Logical order of the work list: [first half of the light chunks][normal chunks][second half of the light chunks].
  // Wavefront w takes positions w, w + 4096, ...: with up to 8 192 chunks the wavefronts that get a second chunk are the
  // ones whose first chunk is light, and their second chunk is light too -- two light chunks cost about as much as one
  // full chunk, so the whole sweep runs in ONE round of resident wavefronts (a second round ran on a mostly empty chip).
  // SO_KNN_PACK (round 4): a light chunk leaves three quarters of a wavefront idle and pays a whole wavefront's prologue, group
  // set-up and epilogue (44 % of the chunks, 40 % of the sweep's instructions for 12 % of its queries).  Four of them now share
  // one wavefront, one per ROW of 16 lanes, each with its own block, row table and quarter of the LDS tile: work list =
  // [packed items: light chunks 4 m .. 4 m + 3][normal chunks].
  // (MatchParams::pack_light = 0 -- the host's choice for a sweep that starts with the full pass, or after sweeps in which the packed
  //  near pass left too many queries to the exact scan -- gives the round-3 list: [half of the light chunks][normal][other half])
  const float cell_w = (float)(1.0 / map.inv_cell);
  const bool first_pass_is_near = 0.5f * cell_w < 0.8f * (sqrtf(mp.sq_max_dist_f) * 1.0005f + 1e-4f) && !mp.skip_near_pass && !(PROF && (mp.ablate & 256));
  const bool pack = SO_KNN_PACK && mp.pack_light && first_pass_is_near;
  const uint32_t n_packed = pack ? (n_light + 3u) >> 2 : 0u;
  const uint32_t n_chunks = n_normal + (pack ? n_packed : n_light), n_light1 = pack ? 0u : (n_light + 1u) >> 1;
  const Pose pose = pose_from_array(chained_begin ? st->T_chain : (begin ? mp.begin_args.pose : st->T));
  if (PROF) {  // kernel statistics (group passes, fallback lanes, candidates scanned): profiling instantiation only
    if (threadIdx.x < 24) lh[threadIdx.x] = 0;
    __syncthreads();
  }
  const int lane_k = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));  // (wave-uniform, and the compiler is told so: the LDS bases below live in SGPRs)
  float* tx = tiles[wv][0];
  float* ty = tiles[wv][1];
  float* tz = tiles[wv][2];
  float* tc = tiles[wv][3];
  uint32_t* ti = tcanon[wv];
  uint32_t* rowoff = rowtab[wv][0];
  uint32_t* rowbeg = rowtab[wv][1];
  const int abl = PROF ? mp.ablate : 0;  // profiling / test switches exist only in the PROF instantiation (launched when SOICP_ABLATE is set)
  const bool stamp = PROF && (abl & 128) != 0 && mp.kdbg != nullptr;
  unsigned long long ts[4] = {0, 0, 0, 0}, acc[5] = {0, 0, 0, 0, 0}, t_first = 0, n_mine = 0, t_maxchunk = 0;
  unsigned long long n_cand_total = 0, n_q_total = 0, n_groups_total = 0, n_pass2 = 0, max_info = 0, n_fb_total = 0;
  unsigned long long c_first = 0;
  if (stamp) { t_first = wall_clock64(); c_first = clock64(); }
  const int nc = map.nc;
  const float cell = (float)(1.0 / map.inv_cell);
  const float inv_cellf = (float)map.inv_cell;
  // Two search radii.  NEAR = half a cell: a query's near ball touches 2 cells per axis, and because a chunk is one
  // HALF-cell octant of the sorted scan (scan_keys_kernel) its union block is 2x2x2 cells instead of 3x3x3 -- a third of
  // the candidates.  A lane whose exact 5th distance lies inside the scanned block's coverage is finished (on a map
  // voxelised at planeRes practically all of them); the others run the FULL pass with the reference's gate radius
  // sqrt(3*planeRes) (LidarSlam.cpp:526,741), where "not found inside the gate ball" is a certain TOO_FAR.
  const float r_gate = sqrtf(mp.sq_max_dist_f) * 1.0005f + 1e-4f;
  const float r_near = 0.5f * cell;
  const int first_pass = (r_near < 0.8f * r_gate && !(abl & 256) && !mp.skip_near_pass) ? 0 : 1;
  // one wavefront per chunk of the work list (a second / further chunk when the list is longer than the grid)
  // (the body is instantiated twice -- packed light chunks / one chunk per wavefront -- so that neither path carries the other's
  //  live values: the kernel sits at its 128-register budget)
  auto do_item = [&](auto packed_tag, const uint32_t chunk) {
  constexpr bool packed = decltype(packed_tag)::value;
  // The lane index of an item is opaque to the optimiser: whatever is derived from it (row, lane in the row, quad offsets, tile
  // and row-table addresses) is formed per item.  Hoisted out of the item loop those values were SPILLED IN THE KERNEL'S PROLOGUE
  // BY EVERY WAVEFRONT, working or idle -- 7 MB of scratch writes per sweep in the PMC traffic (16.8 MB against 9.8 MB
  // algorithmic) for values a handful of integer operations rebuild.
  int lane = lane_k;
  asm volatile("" : "+v"(lane));
  if (stamp) ts[0] = wall_clock64();
  uint32_t j = 0;
  bool valid_q = false;
  // A chunk of at most 32 queries is served by BOTH halves of the wavefront: lanes l and l + 32 hold the same query and
  // scan alternate candidate quads, then exchange their eight survivors (the average chunk has 27 queries).
  bool split = false, split4 = false;  // <= 16 queries: four parts of 16 lanes
  const uint32_t part = (uint32_t)lane >> 4, lane16 = (uint32_t)lane & 15u;
  if constexpr (packed) {  // four light chunks, one per row of 16 lanes
    const uint32_t li = chunk * 4u + part;
    const uint32_t desc = li < n_light ? chunk_start[mp.chunk_cap - 1u - li] : 0u;
    const uint32_t start = desc & 0x03FFFFFFu, count = li < n_light ? (desc >> 26) + 1u : 0u;
    j = start + lane16;
    valid_q = (lane16 < count) && (j < n_kept);
  } else {
    const uint32_t entry = pack ? chunk - n_packed
                         : (chunk < n_light1 ? mp.chunk_cap - 1u - chunk
                            : (chunk < n_light1 + n_normal ? chunk - n_light1 : mp.chunk_cap - 1u - (chunk - n_normal)));
    const uint32_t desc = __builtin_amdgcn_readfirstlane(chunk_start[entry]);
    const uint32_t start = desc & 0x03FFFFFFu, count = (desc >> 26) + 1u;
    split = count <= 32u && !(abl & 1024);
    split4 = count <= 16u && split && !(abl & 2048);
    const int ql = split4 ? (lane & 15) : (split ? (lane & 31) : lane);
    j = start + (uint32_t)ql;
    valid_q = (ql < (int)count) && (j < n_kept);
  }
  // quad offset of this part of the wavefront, and candidates consumed per trip by all parts together
  const uint32_t hoff = split4 ? (uint32_t)(lane >> 4) << 2 : (split ? (uint32_t)(lane >> 5) << 2 : 0u);
  const uint32_t sstep = split4 ? 16u : 8u;
  double pw[3] = {0, 0, 0};
  float qx = 0, qy = 0, qz = 0;
  float ux = 0, uy = 0, uz = 0;  // cube-local coordinates of the query (for the coverage test)
  int wcube0 = 0, wcube1 = 0, wcube2 = 0;  // world id of the query's cube
  CellRef c;
  c.slot = -1; c.cx = c.cy = c.cz = 0;
  uint32_t oi = 0;  // the query's index in the scan (results are filed under it): fetched with the coordinates, used at the very end
  if (valid_q) {
    const float4 rec = binned[j];
    oi = __float_as_uint(rec.w);
    quat_rotate<double>(pose.q, (double)rec.x, (double)rec.y, (double)rec.z, pw[0], pw[1], pw[2]);  // LidarSlam.cpp:397-398
    pw[0] += pose.t[0]; pw[1] += pose.t[1]; pw[2] += pose.t[2];
    qx = (float)pw[0]; qy = (float)pw[1]; qz = (float)pw[2];                                            // LidarSlam.cpp:728-731
    int w[3] = {0, 0, 0};
    c = locate(map, qx, qy, qz, w);
    if (c.slot >= 0) {
      ux = (float)((double)qx - (w[0] * 50.0 - 25.0)); uy = (float)((double)qy - (w[1] * 50.0 - 25.0)); uz = (float)((double)qz - (w[2] * 50.0 - 25.0));
      wcube0 = w[0]; wcube1 = w[1]; wcube2 = w[2];
    }
  }
  const uint32_t ckey = (valid_q && c.slot >= 0) ? (((uint32_t)c.slot << 18) | ((uint32_t)c.cz << 12) | ((uint32_t)c.cy << 6) | (uint32_t)c.cx)
                                                 : 0xFFFFFFFFu;
  Top5 top;
  top.init();
  bool resolved = (ckey == 0xFFFFFFFFu);  // no cube: nothing to search
  bool too_far_certain = false, need_exact = false;
  int n_groups = 0;
  uint32_t n_scanned = 0, n_left_stat = 0;
  if (stamp) { ts[1] = wall_clock64(); acc[0] += ts[1] - ts[0]; }
  // exact re-rank of a lane's survivors + certification (used by the packed near pass and by the group passes)
  // returns 0: not certified (the lane stays pending), 1: exact 5-NN in `top`, 2: certainly beyond the gate, 3: needs the exact
  // per-lane scan.  (Values in and out, no reference captures of the lane's flags: those must stay in registers.)
  auto certify = [qx, qy, qz, mpts, &mp](int pass, const uint32_t (&gs)[8], int32_t k6, int32_t k8, float cov2, Top5& top) -> int {
    // First the five best approximate keys only.  Every candidate that is NOT re-ranked has exact d2 >= R2:
    //   in-block outsiders: approximate d2 >= L (the first key left out, index bits cleared), exact >= L - kApproxAbsErr;
    //   points of the cube outside the block: farther than the block boundary (cov2).
    unsigned long long e[5];
#pragma unroll
    for (int t = 0; t < 5; ++t) {
      e[t] = ~0ull;
      if (gs[t] != 0xFFFFFFFFu) {
        const float4 p = mpts[gs[t]];
        e[t] = ((unsigned long long)__float_as_uint(l2_d2(qx, qy, qz, p.x, p.y, p.z)) << 32) | gs[t];
      }
    }
    top.set5(e[0], e[1], e[2], e[3], e[4]);
    const double cov = (double)cov2 * (1.0 - 1e-6);
    double R2 = cov;
    if (k6 != kKeyEmpty) R2 = fmin(R2, (double)__uint_as_float((uint32_t)k6 & ~kKeyIdxMask) * (1.0 - 1e-6) - (double)kApproxAbsErr);
    bool have5 = top.b4 != ~0ull;
    double d5 = (double)__uint_as_float((uint32_t)(top.b4 >> 32));
    bool exact = have5 && d5 < R2;
    bool far = !exact && R2 > (double)mp.sq_max_dist_f;
    // a 5th and a 6th candidate too close to call on approximate keys (a few lanes in a thousand): re-rank all eight
    if (__ballot(!exact && !far && gs[5] != 0xFFFFFFFFu)) {
      if (!exact && !far) {
#pragma unroll
        for (int t = 5; t < 8; ++t) {
          if (gs[t] != 0xFFFFFFFFu) {
            const float4 p = mpts[gs[t]];
            top.insert(((unsigned long long)__float_as_uint(l2_d2(qx, qy, qz, p.x, p.y, p.z)) << 32) | gs[t]);
          }
        }
        R2 = cov;
        if (k8 != kKeyEmpty) R2 = fmin(R2, (double)__uint_as_float((uint32_t)k8 & ~kKeyIdxMask) * (1.0 - 1e-6) - (double)kApproxAbsErr);
        have5 = top.b4 != ~0ull;
        d5 = (double)__uint_as_float((uint32_t)(top.b4 >> 32));
        exact = have5 && d5 < R2;
        far = !exact && R2 > (double)mp.sq_max_dist_f;
      }
    }
    return exact ? 1 : (far ? 2 : (pass == 1 ? 3 : 0));  // (far: the true 5th neighbour is >= R2 > gate, LidarSlam.cpp:741)
  };
  // ---- packed light chunks: the NEAR pass of four chunks at once, one per row of 16 lanes.  What the wave-uniform scalars of
  //      the group passes below are -- block bounds, row table, candidate count, tile -- lives in VGPRs here, uniform over a row
  //      (DPP row all-reduces instead of ballots / readlanes), and every row streams its own quarter of the LDS tile (four LDS
  //      addresses per read, like the split scan).  A row whose block has more than 16 x-runs or keeps more than 64 candidates,
  //      and a lane in another cube than its row's first, is left to the group passes below; so is every lane the near pass
  //      cannot certify (full pass) -- the results are the same exact lists either way.
  constexpr bool near_done = false;  // (group passes only: every lane takes part in its near pass)
  if constexpr (packed) if (first_pass == 0 && !(abl & 2))
  for (int ppass = 0; ppass < 2; ++ppass) {  // near pass, then -- for the rows that still have uncertified lanes -- the full pass (gate radius)
    const bool pend = !resolved && !need_exact;
    if (ppass == 1 && __ballot(pend) == 0ull) break;
    if (ppass == 1) __builtin_amdgcn_s_setprio(3);  // (a wavefront with a second pass ahead is one of the sweep's stragglers: issue priority from here on)
    if (PROF) ++n_groups;
    const int myslot = (int)(ckey >> 18);
    const int gslot = row_min_i32(pend ? myslot : 0x7FFFFFFF);
    const bool mine = pend && myslot == gslot;
    const float r_cover = ppass == 0 ? r_near : r_gate;
    const int lo_x = max(0, (int)floorf((ux - r_cover) * inv_cellf)), hi_x = min(nc - 1, (int)floorf((ux + r_cover) * inv_cellf));
    const int lo_y = max(0, (int)floorf((uy - r_cover) * inv_cellf)), hi_y = min(nc - 1, (int)floorf((uy + r_cover) * inv_cellf));
    const int lo_z = max(0, (int)floorf((uz - r_cover) * inv_cellf)), hi_z = min(nc - 1, (int)floorf((uz + r_cover) * inv_cellf));
    const int bx0 = row_min_i32(mine ? lo_x : 0x7FFFFFFF), bx1 = row_max_i32(mine ? hi_x : -1);
    const int by0 = row_min_i32(mine ? lo_y : 0x7FFFFFFF), by1 = row_max_i32(mine ? hi_y : -1);
    const int bz0 = row_min_i32(mine ? lo_z : 0x7FFFFFFF), bz1 = row_max_i32(mine ? hi_z : -1);
    const int nyr = by1 - by0 + 1, nrows = nyr * (bz1 - bz0 + 1);
    bool part_ok = gslot != 0x7FFFFFFF && nrows >= 1 && nrows <= 16;  // (uniform over the row)
    // squared distance from the query to the faces of the block (a face on the cube's boundary has nothing of the cube behind
    // it), capped by the filter radius: formed here, while the bounds are at hand (six registers less across the scan)
    float cov2p;
    {
      float cv = 1e15f;
      if (bx0 > 0) cv = fminf(cv, ux - (float)bx0 * cell);
      if (bx1 < nc - 1) cv = fminf(cv, (float)(bx1 + 1) * cell - ux);
      if (by0 > 0) cv = fminf(cv, uy - (float)by0 * cell);
      if (by1 < nc - 1) cv = fminf(cv, (float)(by1 + 1) * cell - uy);
      if (bz0 > 0) cv = fminf(cv, uz - (float)bz0 * cell);
      if (bz1 < nc - 1) cv = fminf(cv, (float)(bz1 + 1) * cell - uz);
      cv = fmaxf(cv - 1e-4f, 0.f);  // cell membership of a map point is decided in fp64 on its own coordinates: keep a margin
      if (ppass == 1) cv = 1e15f;   // (the full pass's block contains the lane's whole gate ball by construction)
      cov2p = fminf(cv * cv, r_cover * r_cover);  // (candidates beyond r_cover of every lane of the group are not staged)
    }
    // row table of the part's block: lane r of the row fetches the bounds of x-run r, inclusive scan over the row
    uint32_t vb = 0, vl = 0;
    if (part_ok && (int)lane16 < nrows) {
      const int zq = (int)(((float)lane16 + 0.5f) * __builtin_amdgcn_rcpf((float)nyr));  // lane16 / nyr (see the group passes)
      const int z = bz0 + zq, y = by0 + ((int)lane16 - zq * nyr);
      const uint32_t* row = mcell_start + (size_t)gslot * map.ncell1 + ((size_t)z * nc + y) * nc;
      vb = row[bx0]; vl = row[bx1 + 1] - vb;
    }
    uint32_t inc = vl;
    inc += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)inc, 0x111, 0xF, 0xF, true);
    inc += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)inc, 0x112, 0xF, 0xF, true);
    inc += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)inc, 0x114, 0xF, 0xF, true);
    inc += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)inc, 0x118, 0xF, 0xF, true);
    const uint32_t total = (uint32_t)row_max_i32((int)inc);  // (the inclusive scan is monotone: its maximum is the row's total)
    part_ok = part_ok && total <= 1024u;
    uint32_t* prowoff = rowoff + part * 17u;   // [17] exclusive offsets (total from entry nrows on)
    uint32_t* prowbeg = rowbeg + part * 16u;   // [16] first canonical index of the x-run
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    prowoff[lane16] = ((int)lane16 < nrows) ? inc - vl : total; prowbeg[lane16] = vb;
    if (lane16 == 0) prowoff[16] = total;
    // block-local frame of the part: origin at the centre of the block's middle cell, in the group's cube (the lanes of a group
    // share the cube; the staging lanes below need the ROW's origin whatever their own query is, so it is reduced over the row)
    const int gz = (bz0 + bz1) >> 1, gy = (by0 + by1) >> 1, gx = (bx0 + bx1) >> 1;
    const int w0r = row_max_i32(mine ? wcube0 : -0x7FFFFFFF), w1r = row_max_i32(mine ? wcube1 : -0x7FFFFFFF), w2r = row_max_i32(mine ? wcube2 : -0x7FFFFFFF);
    const double rox = (w0r * 50.0 - 25.0) + ((double)gx + 0.5) * (double)cell;
    const double roy = (w1r * 50.0 - 25.0) + ((double)gy + 0.5) * (double)cell;
    const double roz = (w2r * 50.0 - 25.0) + ((double)gz + 0.5) * (double)cell;
    const float lqx = (float)((double)qx - rox), lqy = (float)((double)qy - roy), lqz = (float)((double)qz - roz);
    const float m2qx = -2.f * lqx, m2qy = -2.f * lqy, m2qz = -2.f * lqz;
    const float qq = __builtin_fmaf(lqz, lqz, __builtin_fmaf(lqy, lqy, lqx * lqx));
    const float pinf = __int_as_float(0x7F800000);
    const float bl0 = row_min_f32(mine ? lqx : pinf), bl1 = row_min_f32(mine ? lqy : pinf), bl2 = row_min_f32(mine ? lqz : pinf);
    const float bh0 = -row_min_f32(mine ? -lqx : pinf), bh1 = -row_min_f32(mine ? -lqy : pinf), bh2 = -row_min_f32(mine ? -lqz : pinf);
    const float dk = r_cover + 2e-4f, dk2 = dk * dk;
    const uint32_t tbase = part * kPartTile;
    const uint32_t tot_eff = part_ok ? total : 0u;
    const uint32_t tmax = max(max((uint32_t)__builtin_amdgcn_readlane((int)tot_eff, 0), (uint32_t)__builtin_amdgcn_readlane((int)tot_eff, 16)),
                              max((uint32_t)__builtin_amdgcn_readlane((int)tot_eff, 32), (uint32_t)__builtin_amdgcn_readlane((int)tot_eff, 48)));
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (PROF) n_scanned += tmax;
    uint32_t w = 0;  // candidates the part has kept so far (uniform over the row)
    // Staging.  A group of lanes -- width 16 = every row for itself (near pass: all four rows are busy), or width 64 = the
    // whole wavefront for ONE row after the other (full pass: one or two rows still have lanes, their blocks hold hundreds of
    // points) -- enumerates a row's block, `width` candidates per step.  Four steps' loads are issued together (four in flight
    // per lane; a step per memory round trip made a 100-candidate block seven dependent round trips), the candidates are
    // filtered against the row's box and compacted into the row's quarter of the tile in enumeration order.
    auto stage = [&](const uint32_t tot, const uint32_t tend, const uint32_t gl /*lane in the group*/, const uint32_t width, const uint32_t shift,
                     const uint32_t* poff, const uint32_t* pbeg, const uint32_t tb, const double sx, const double sy, const double sz,
                     const float l0, const float l1, const float l2, const float h0, const float h1, const float h2) -> uint32_t {
      uint32_t kept = 0;
      const unsigned long long gmask = width == 64u ? ~0ull : 0xFFFFull;
      for (uint32_t t0 = 0; t0 < tend; t0 += 4u * width) {
        float px_[4], py_[4], pz_[4];
        uint32_t cn[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const uint32_t t = t0 + width * (uint32_t)u + gl;
          cn[u] = 0xFFFFFFFFu; px_[u] = 0.f; py_[u] = 0.f; pz_[u] = 0.f;
          if (t < tot) {
            int r = 0;
#pragma unroll
            for (int step = 8; step >= 1; step >>= 1) r = (r + step < 16 && poff[r + step] <= t) ? r + step : r;
            cn[u] = pbeg[r] + (t - poff[r]);
            const float4 p = mpts[cn[u]];
            px_[u] = p.x; py_[u] = p.y; pz_[u] = p.z;
          }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          if (t0 + width * (uint32_t)u >= tend) break;  // (uniform)
          bool kp = false;
          float lx = 0.f, ly = 0.f, lz = 0.f, lc = 0.f;
          if (cn[u] != 0xFFFFFFFFu) {
            lx = (float)((double)px_[u] - sx); ly = (float)((double)py_[u] - sy); lz = (float)((double)pz_[u] - sz);
            lc = __builtin_fmaf(lz, lz, __builtin_fmaf(ly, ly, lx * lx));
            const float ex = fmaxf(fmaxf(l0 - lx, lx - h0), 0.f), ey = fmaxf(fmaxf(l1 - ly, ly - h1), 0.f), ez = fmaxf(fmaxf(l2 - lz, lz - h2), 0.f);
            kp = __builtin_fmaf(ez, ez, __builtin_fmaf(ey, ey, ex * ex)) <= dk2;
          }
          const unsigned long long mg = (__ballot(kp) >> shift) & gmask;
          const uint32_t pos = kept + (uint32_t)__popcll(mg & ((1ull << gl) - 1ull));
          if (kp && pos < kPartTile) { tx[tb + pos] = lx; ty[tb + pos] = ly; tz[tb + pos] = lz; tc[tb + pos] = lc; ti[tb + pos] = cn[u]; }
          kept += (uint32_t)__popcll(mg);
        }
      }
      return kept;
    };
    if (ppass == 0) {
      w = stage(tot_eff, tmax, lane16, 16u, part * 16u, prowoff, prowbeg, tbase, rox, roy, roz, bl0, bl1, bl2, bh0, bh1, bh2);
    } else {
      for (int rr = 0; rr < 4; ++rr) {  // one row after the other, all 64 lanes on it (row-uniform values from the row's first lane)
        const int src = rr * 16;
        const uint32_t tot_r = (uint32_t)__builtin_amdgcn_readlane((int)tot_eff, src);
        if (!tot_r) continue;
#define SO_RL_F(v) __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src))
#define SO_RL_D(v) __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), src), __builtin_amdgcn_readlane(__double2loint(v), src))
        const uint32_t wr = stage(tot_r, tot_r, (uint32_t)lane, 64u, 0u, rowoff + rr * 17, rowbeg + rr * 16, (uint32_t)rr * kPartTile,
                                  SO_RL_D(rox), SO_RL_D(roy), SO_RL_D(roz), SO_RL_F(bl0), SO_RL_F(bl1), SO_RL_F(bl2), SO_RL_F(bh0), SO_RL_F(bh1), SO_RL_F(bh2));
#undef SO_RL_F
#undef SO_RL_D
        if ((int)part == rr) w = wr;
      }
    }
    if (PROF && lane16 == 0 && gslot != 0x7FFFFFFF) {  // statistics: rows with work / left to the group passes (x-runs, kept candidates) / kept candidates
      atomicAdd(&lh[20], 1);
      if (!(part_ok && w <= kPartTile)) atomicAdd(&lh[nrows > 16 || nrows < 1 ? 21 : 22], 1);
      atomicAdd(&lh[23], (int)w);
    }
    part_ok = part_ok && w <= kPartTile;
    const uint32_t wk = part_ok ? w : 0u;
    for (uint32_t sl = wk + lane16; sl < kPartTile; sl += 16u) {  // the rest of the part's quarter: entries that lose against every real candidate
      tx[tbase + sl] = 0.f; ty[tbase + sl] = 0.f; tz[tbase + sl] = 0.f; tc[tbase + sl] = 3.0e38f; ti[tbase + sl] = 0xFFFFFFFFu;
    }
    const uint32_t wmax = (max(max((uint32_t)__builtin_amdgcn_readlane((int)wk, 0), (uint32_t)__builtin_amdgcn_readlane((int)wk, 16)),
                               max((uint32_t)__builtin_amdgcn_readlane((int)wk, 32), (uint32_t)__builtin_amdgcn_readlane((int)wk, 48))) + 3u) & ~3u;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const uint32_t keep = ~kKeyIdxMask;
    const float2v pqx = {m2qx, m2qx}, pqy = {m2qy, m2qy}, pqz = {m2qz, m2qz}, pqq = {qq, qq};
    Net8 net;
    net.init();
    if (!(abl & 8))
    for (uint32_t jl = 0; jl < wmax; jl += 4u) {  // every row walks its own quarter of the tile: four LDS addresses per read
      const uint32_t a = tbase + jl;
      const float4 X = *reinterpret_cast<const float4*>(tx + a), Y = *reinterpret_cast<const float4*>(ty + a);
      const float4 Z = *reinterpret_cast<const float4*>(tz + a), C = *reinterpret_cast<const float4*>(tc + a);
      const float2v d01 = approx_d2_pair(pqx, pqy, pqz, pqq, float2v{X.x, X.y}, float2v{Y.x, Y.y}, float2v{Z.x, Z.y}, float2v{C.x, C.y});
      const float2v d23 = approx_d2_pair(pqx, pqy, pqz, pqq, float2v{X.z, X.w}, float2v{Y.z, Y.w}, float2v{Z.z, Z.w}, float2v{C.z, C.w});
      net.push(make_key(d01.x, jl, keep));
      net.push(make_key(d01.y, jl + 1u, keep));
      net.push(make_key(d23.x, jl + 2u, keep));
      net.push(make_key(d23.y, jl + 3u, keep));
    }
    const int32_t ks[8] = {net.a0, net.a1, net.a2, net.a3, net.a4, net.a5, net.a6, net.a7};
    uint32_t gs[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const uint32_t jl = (uint32_t)ks[t] & kKeyIdxMask;
      gs[t] = (ks[t] == kKeyEmpty || jl >= wk) ? 0xFFFFFFFFu : ti[tbase + (jl < kPartTile ? jl : 0u)];
    }
    if (mine && part_ok) {
      if (!(abl & 4)) {
        const int v = certify(ppass, gs, net.a5, net.a7, cov2p, top);
        resolved = v == 1 || v == 2; too_far_certain = v == 2; need_exact = v == 3;
      } else resolved = true;
    } else if (mine && ppass == 1) {
      need_exact = true;  // (a row whose full-pass block has too many x-runs or keeps more than its quarter of the tile)
    }
  }
  // One query (lane L's) at a time, the WHOLE wavefront on it: the (clamped) 3 x 3 x 3 cells around the query -- every map point
  // inside the gate ball lies there (one cell >= the gate radius) -- as <= 9 x-runs, their points dealt to the 64 lanes with
  // the loads of a lane issued together, exact distances, lane-local top 5, five wavefront minima.  (The per-lane scan
  // knn27() walks those ~300 points through dependent loads: 60 us for one lane, which ended the whole sweep.)
  auto coop_exact_scan = [&](const int L) {
      const float sqx = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(qx), L)), sqy = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(qy), L));
      const float sqz = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(qz), L));
      const int sslot = __builtin_amdgcn_readlane(c.slot, L), scx = __builtin_amdgcn_readlane(c.cx, L);
      const int scy = __builtin_amdgcn_readlane(c.cy, L), scz = __builtin_amdgcn_readlane(c.cz, L);
      const int x0 = scx > 0 ? scx - 1 : 0, x1 = scx < nc - 1 ? scx + 1 : nc - 1;
      uint32_t vb = 0, vl = 0;
      if (lane < 9) {
        const int y = scy + (lane % 3) - 1, z = scz + (lane / 3) - 1;
        if (y >= 0 && y < nc && z >= 0 && z < nc) {
          const uint32_t* row = mcell_start + (size_t)sslot * map.ncell1 + ((size_t)z * nc + y) * nc;
          vb = row[x0]; vl = row[x1 + 1] - vb;
        }
      }
      uint32_t inc = vl;  // inclusive scan over lanes 0..15 (the nine runs sit in the first row of 16 lanes)
      inc += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)inc, 0x111, 0xF, 0xF, true);
      inc += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)inc, 0x112, 0xF, 0xF, true);
      inc += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)inc, 0x114, 0xF, 0xF, true);
      inc += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)inc, 0x118, 0xF, 0xF, true);
      const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)inc, 15);
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
      if (lane < 16) { rowoff[lane] = lane < 9 ? inc - vl : total; rowbeg[lane] = vb; }
      if (lane == 0) rowoff[16] = total;
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
      Top5 loc;
      loc.init();
      for (uint32_t t0 = 0; t0 < total; t0 += 256u) {
        float ax_[4], ay_[4], az_[4];
        uint32_t cn[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const uint32_t t = t0 + 64u * (uint32_t)u + (uint32_t)lane;
          cn[u] = 0xFFFFFFFFu; ax_[u] = ay_[u] = az_[u] = 0.f;
          if (t < total) {
            int r = 0;
#pragma unroll
            for (int step = 8; step >= 1; step >>= 1) r = (r + step < 16 && rowoff[r + step] <= t) ? r + step : r;
            cn[u] = rowbeg[r] + (t - rowoff[r]);
            const float4 p = mpts[cn[u]];
            ax_[u] = p.x; ay_[u] = p.y; az_[u] = p.z;
          }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
          if (cn[u] != 0xFFFFFFFFu) loc.insert(((unsigned long long)__float_as_uint(l2_d2(sqx, sqy, sqz, ax_[u], ay_[u], az_[u])) << 32) | cn[u]);
      }
      unsigned long long m5[5];
#pragma unroll
      for (int t = 0; t < 5; ++t) {
        m5[t] = wave_min_u64(loc.b0);
        if (loc.b0 == m5[t] && m5[t] != ~0ull) { loc.b0 = loc.b1; loc.b1 = loc.b2; loc.b2 = loc.b3; loc.b3 = loc.b4; loc.b4 = ~0ull; }  // (keys are unique)
      }
      if (lane == L) {  // (fewer than five points in reach: b4 stays "none" and the gate below says TOO_FAR, like the per-lane scan)
        top.b0 = m5[0]; top.b1 = m5[1]; top.b2 = m5[2]; top.b3 = m5[3]; top.b4 = m5[4];
        too_far_certain = false; resolved = true;
      }
  };
  if constexpr (packed) {
    // what the packed near pass could not finish -- a 5th neighbour beyond half a cell, a row with too many x-runs or kept
    // candidates, a lane in another cube than its row -- goes to the exact per-lane scan of the 27 cells below (the group passes
    // are not instantiated here: registers and code size).  The host watches the count and turns the packing off for sweeps
    // where it is not rare.
    unsigned long long left = __ballot(valid_q && c.slot >= 0 && !resolved);
    if (left) __builtin_amdgcn_s_setprio(3);
    if (left && lane == 0 && mp.packed_leftover) atomicAdd(leftover_ctr, (uint32_t)__popcll(left));
    if (PROF) n_left_stat = (uint32_t)__popcll(left);
    while (left) {
      const int L = __ffsll((long long)left) - 1;
      left &= left - 1ull;
      coop_exact_scan(L);
    }
  }
  if constexpr (!packed)
  for (int pass = first_pass; pass < 2; ++pass) {
  bool pending = !resolved && !need_exact && !(pass == 0 && near_done);
  unsigned long long todo = __ballot(pending);
  if (abl & 2) todo = 0;
  if (!todo) continue;  // (a packed wavefront may have nothing left for the near pass and still lanes for the full pass)
  if (pass == 1 && first_pass == 0) __builtin_amdgcn_s_setprio(3);  // (a second pass: one of the sweep's stragglers -- issue priority from here on)
  if (stamp && pass == 1) ++n_pass2;
  const float r_cover = pass == 0 ? r_near : r_gate;
  // per-lane cell range that contains the lane's search ball (clamped to the cube: nothing of the cube lies beyond it)
  const int lo_x = max(0, (int)floorf((ux - r_cover) * inv_cellf)), hi_x = min(nc - 1, (int)floorf((ux + r_cover) * inv_cellf));
  const int lo_y = max(0, (int)floorf((uy - r_cover) * inv_cellf)), hi_y = min(nc - 1, (int)floorf((uy + r_cover) * inv_cellf));
  const int lo_z = max(0, (int)floorf((uz - r_cover) * inv_cellf)), hi_z = min(nc - 1, (int)floorf((uz + r_cover) * inv_cellf));
  uint32_t g0 = 0xFFFFFFFFu, g1 = g0, g2 = g0, g3 = g0, g4 = g0, g5 = g0, g6 = g0, g7 = g0;  // canonical indices of the 8 survivors
  int32_t k6 = kKeyEmpty, k8 = kKeyEmpty;  // 6th and 8th key of the lane's group pass
  float cov2 = 0.f;            // squared distance from the query to the boundary of the scanned block
  bool scanned = false;
  while (todo) {
    const int leader = __ffsll((long long)todo) - 1;
    const uint32_t k = __builtin_amdgcn_readlane(ckey, leader);  // wave-uniform (SGPR) cell key of the leader
    const int gslot = (int)(k >> 18);
    // ---- group = every pending lane of the leader's cube; block = union of the lanes' search-ball cell ranges.
    //      A chunk is spatially compact (its queries shared one half-cell octant when the scan was sorted and a rigid
    //      pose update keeps them together), so the union is 2..3 cells per axis (near pass) or 3..4 (full pass).  If
    //      it is too large for the key's index field the group shrinks to the lanes around / of the leader's own cell.
    bool mine = pending && ((int)(ckey >> 18) == gslot);
    int bx0, bx1, by0, by1, bz0, bz1;
    uint32_t total = 0;
    uint32_t vb = 0, vl = 0;
    int nrows = 0, nyr = 0;
    const int lcx = (int)(k & 63u), lcy = (int)((k >> 6) & 63u), lcz = (int)((k >> 12) & 63u);
    for (int attempt = 0; attempt < 3; ++attempt) {
      if (attempt == 1)
        mine = mine && abs(c.cx - lcx) <= 1 && abs(c.cy - lcy) <= 1 && abs(c.cz - lcz) <= 1;
      if (attempt == 2) mine = mine && (ckey == k);
      bx0 = __builtin_amdgcn_readlane(lo_x, leader); bx1 = __builtin_amdgcn_readlane(hi_x, leader);
      by0 = __builtin_amdgcn_readlane(lo_y, leader); by1 = __builtin_amdgcn_readlane(hi_y, leader);
      bz0 = __builtin_amdgcn_readlane(lo_z, leader); bz1 = __builtin_amdgcn_readlane(hi_z, leader);
      while (__ballot(mine && lo_x < bx0)) --bx0;   // wave-uniform min / max over the group, a few ballots each
      while (__ballot(mine && hi_x > bx1)) ++bx1;
      while (__ballot(mine && lo_y < by0)) --by0;
      while (__ballot(mine && hi_y > by1)) ++by1;
      while (__ballot(mine && lo_z < bz0)) --bz0;
      while (__ballot(mine && hi_z > bz1)) ++bz1;
      nyr = by1 - by0 + 1;
      nrows = nyr * (bz1 - bz0 + 1);
      total = 0xFFFFFFFFu;
      if (nrows <= 32) {  // row table: lane r fetches the bounds of x-run r, prefix sums by shuffles
        vb = 0; vl = 0;
        if (lane < nrows) {
          // lane / nyr without the ~30-instruction integer division: (lane + 0.5) / nyr is never within 0.5 / 32 of an
          // integer, far more than the error of the float reciprocal (lane < 32, nyr <= 32)
          const int zq = (int)(((float)lane + 0.5f) * __builtin_amdgcn_rcpf((float)nyr));
          const int z = bz0 + zq, y = by0 + (lane - zq * nyr);
          const uint32_t* row = mcell_start + (size_t)gslot * map.ncell1 + ((size_t)z * nc + y) * nc;
          vb = row[bx0]; vl = row[bx1 + 1] - vb;
        }
        // inclusive scan over lanes 0..31 on the DPP network (row_shr 1,2,4,8 inside the rows of 16, row_bcast:15 carries
        // row 0's total into row 1): no LDS round trips
        uint32_t inc = vl;
        inc += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)inc, 0x111, 0xF, 0xF, true);
        inc += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)inc, 0x112, 0xF, 0xF, true);
        inc += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)inc, 0x114, 0xF, 0xF, true);
        inc += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)inc, 0x118, 0xF, 0xF, true);
        inc += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)inc, 0x142, 0xA, 0xF, false);
        total = __builtin_amdgcn_readlane(inc, 31);
        if (lane < 32) { rowoff[lane] = inc - vl; rowbeg[lane] = vb; }
        if (lane == 0) rowoff[32] = total;
      }
      if (total <= kGroupMaxCand) break;
    }
    pending = pending && !mine;
    todo = __ballot(pending);
    ++n_groups;
    if (total > kGroupMaxCand) {  // still too many candidates for the key's index field: exact per-lane scan for these lanes
      need_exact = need_exact || mine;
      continue;
    }
    n_scanned += total;
    // The SIMDs are issue-bound with ~5 resident wavefronts each: a wavefront with an above-average scan (dense cells, or
    // the full pass after a near pass) would finish long after its neighbours and set the kernel time.  Give it issue
    // priority so that the stragglers are the light chunks instead.
    if (n_scanned > 256u) __builtin_amdgcn_s_setprio(3);
    else if (n_scanned > 128u) __builtin_amdgcn_s_setprio(2);
    if (stamp) { ts[2] = wall_clock64(); acc[1] += ts[2] - ts[1]; }
    const int gz = (bz0 + bz1) >> 1, gy = (by0 + by1) >> 1, gx = (bx0 + bx1) >> 1;
    // block-local frame: origin at the centre of the block's middle cell (world coordinates, fp64)
    const double ox = ((int)__builtin_amdgcn_readlane(wcube0, leader) * 50.0 - 25.0) + ((double)gx + 0.5) * (double)cell;
    const double oy = ((int)__builtin_amdgcn_readlane(wcube1, leader) * 50.0 - 25.0) + ((double)gy + 0.5) * (double)cell;
    const double oz = ((int)__builtin_amdgcn_readlane(wcube2, leader) * 50.0 - 25.0) + ((double)gz + 0.5) * (double)cell;
    const float lqx = (float)((double)qx - ox), lqy = (float)((double)qy - oy), lqz = (float)((double)qz - oz);
    const float m2qx = -2.f * lqx, m2qy = -2.f * lqy, m2qz = -2.f * lqz;
    const float qq = __builtin_fmaf(lqz, lqz, __builtin_fmaf(lqy, lqy, lqx * lqx));
    const uint32_t keep = ~kKeyIdxMask;
    const float2v pqx = {m2qx, m2qx}, pqy = {m2qy, m2qy}, pqz = {m2qz, m2qz}, pqq = {qq, qq};
    Net8 net;
    net.init();
    // Candidate FILTER: the block is whole cells, the lanes sit in one corner of it, so
    // most of its points are farther from EVERY lane than the radius this pass can certify anyway (r_cover: half a cell in
    // the near pass, the gate radius in the full pass).  A candidate is staged only if it lies within r_cover (+ margin) of
    // the bounding box of the group's queries -- about half of the block's points on a surface map -- and the selection
    // network runs over the kept ones only.  Exactness: a dropped point is farther than r_cover from every query of the
    // group, so "every point that was not re-ranked has exact d2 >= R2" holds with R2 <= r_cover^2 (cov2 below).
    bool filt = SO_KNN_FILTER != 0;
    float bl0 = 0.f, bl1 = 0.f, bl2 = 0.f, bh0 = 0.f, bh1 = 0.f, bh2 = 0.f;
    if (filt) {
      const float pinf = __int_as_float(0x7F800000);
      bl0 = wave_min_f32(mine ? lqx : pinf); bl1 = wave_min_f32(mine ? lqy : pinf); bl2 = wave_min_f32(mine ? lqz : pinf);
      bh0 = -wave_min_f32(mine ? -lqx : pinf); bh1 = -wave_min_f32(mine ? -lqy : pinf); bh2 = -wave_min_f32(mine ? -lqz : pinf);
    }
    const float dk = r_cover + 2e-4f, dk2 = dk * dk;  // (block-local fp32 coordinates: errors ~1e-6 m, the margin covers them a hundredfold)
    uint32_t total_eff = total;
    if (filt) {
      // the WHOLE enumeration is filtered into the one tile; a group whose kept candidates do not fit it (dense cells in the
      // full pass) falls back to the unfiltered stream of pieces below
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
      uint32_t w = 0;  // kept candidates so far (wave-uniform)
      for (uint32_t t0 = 0; t0 < total; t0 += 64) {
        const uint32_t t = t0 + (uint32_t)lane;
        bool kp = false;
        float lx = 0.f, ly = 0.f, lz = 0.f, lc = 0.f;
        uint32_t canon = 0xFFFFFFFFu;
        if (t < total) {
          int r = 0;
#pragma unroll
          for (int step = 16; step >= 1; step >>= 1) r = (r + step < 32 && rowoff[r + step] <= t) ? r + step : r;
          canon = rowbeg[r] + (t - rowoff[r]);
          const float4 p = mpts[canon];
          lx = (float)((double)p.x - ox); ly = (float)((double)p.y - oy); lz = (float)((double)p.z - oz);
          lc = __builtin_fmaf(lz, lz, __builtin_fmaf(ly, ly, lx * lx));
          const float ex = fmaxf(fmaxf(bl0 - lx, lx - bh0), 0.f), ey = fmaxf(fmaxf(bl1 - ly, ly - bh1), 0.f), ez = fmaxf(fmaxf(bl2 - lz, lz - bh2), 0.f);
          kp = __builtin_fmaf(ez, ez, __builtin_fmaf(ey, ey, ex * ex)) <= dk2;
        }
        const unsigned long long m = __ballot(kp);
        const uint32_t nk = (uint32_t)__popcll(m);
        if (w + nk > kTileCand) { filt = false; break; }
        if (kp) {
          const uint32_t pos = w + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
          tx[pos] = lx; ty[pos] = ly; tz[pos] = lz; tc[pos] = lc; ti[pos] = canon;
        }
        w += nk;
      }
      if (filt) {
        if (lane < 16) { tx[w + lane] = 0.f; ty[w + lane] = 0.f; tz[w + lane] = 0.f; tc[w + lane] = 3.0e38f; ti[w + lane] = 0xFFFFFFFFu; }  // padding entries lose
        total_eff = w;
      }
    }
    const uint32_t n_enum = filt ? total_eff : total;  // what the scan below walks: the kept candidates in the tile, or the raw enumeration
    // The group's candidate enumeration [0, total) is streamed through the LDS tile in pieces of kTileCand; the
    // selection network simply continues across pieces (keys carry the position in the whole enumeration).
    for (uint32_t base = 0; base < n_enum; base += kTileCand) {
      const uint32_t cnt = (n_enum - base < kTileCand) ? n_enum - base : kTileCand;
      // stage with coalesced 16-byte loads (position -> row by a 5-step binary search over the row offsets)
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
      if (filt) {
        // (the tile already holds the kept candidates)
      } else
      if (!(abl & 16))
      for (uint32_t t = lane; t < cnt + 16; t += 64) {
        float lx = 0.f, ly = 0.f, lz = 0.f, lc = 3.0e38f;  // padding entries lose against every real candidate
        uint32_t canon = 0xFFFFFFFFu;
        if (t < cnt) {
          const uint32_t e = base + t;
          int r = 0;
#pragma unroll
          for (int step = 16; step >= 1; step >>= 1) r = (r + step < 32 && rowoff[r + step] <= e) ? r + step : r;
          canon = rowbeg[r] + (e - rowoff[r]);
          const float4 p = mpts[canon];
          lx = (float)((double)p.x - ox); ly = (float)((double)p.y - oy); lz = (float)((double)p.z - oz);
          lc = __builtin_fmaf(lz, lz, __builtin_fmaf(ly, ly, lx * lx));
        }
        tx[t] = lx; ty[t] = ly; tz[t] = lz; tc[t] = lc; ti[t] = canon;
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
      if (!(abl & 8) && split) {
        // two quads per trip, one per half of the wavefront (two LDS addresses per read)
        for (uint32_t jl = 0; jl < cnt; jl += sstep) {
          const uint32_t a = jl + hoff;
          const float4 X = *reinterpret_cast<const float4*>(tx + a), Y = *reinterpret_cast<const float4*>(ty + a);
          const float4 Z = *reinterpret_cast<const float4*>(tz + a), C = *reinterpret_cast<const float4*>(tc + a);
          const float2v d01 = approx_d2_pair(pqx, pqy, pqz, pqq, float2v{X.x, X.y}, float2v{Y.x, Y.y}, float2v{Z.x, Z.y}, float2v{C.x, C.y});
          const float2v d23 = approx_d2_pair(pqx, pqy, pqz, pqq, float2v{X.z, X.w}, float2v{Y.z, Y.w}, float2v{Z.z, Z.w}, float2v{C.z, C.w});
          const uint32_t e = base + a;  // per lane
          net.push(make_key_v(d01.x, e, keep));
          net.push(make_key_v(d01.y, e + 1, keep));
          net.push(make_key_v(d23.x, e + 2, keep));
          net.push(make_key_v(d23.y, e + 3, keep));
        }
      } else if (!(abl & 8)) {
        // uniform addresses: four broadcast ds_read_b128 feed four candidates; the next quad is fetched while this one
        // runs through the selection network (software pipeline, no wait between LDS issue and use)
        float4 X = *reinterpret_cast<const float4*>(tx), Y = *reinterpret_cast<const float4*>(ty);
        float4 Z = *reinterpret_cast<const float4*>(tz), C = *reinterpret_cast<const float4*>(tc);
        for (uint32_t jl = 0; jl < cnt; jl += 4) {
          const uint32_t nx = (jl + 4 < cnt) ? jl + 4 : jl;  // last iteration re-reads (harmless)
          const float4 Xn = *reinterpret_cast<const float4*>(tx + nx), Yn = *reinterpret_cast<const float4*>(ty + nx);
          const float4 Zn = *reinterpret_cast<const float4*>(tz + nx), Cn = *reinterpret_cast<const float4*>(tc + nx);
          const float2v d01 = approx_d2_pair(pqx, pqy, pqz, pqq, float2v{X.x, X.y}, float2v{Y.x, Y.y}, float2v{Z.x, Z.y}, float2v{C.x, C.y});
          const float2v d23 = approx_d2_pair(pqx, pqy, pqz, pqq, float2v{X.z, X.w}, float2v{Y.z, Y.w}, float2v{Z.z, Z.w}, float2v{C.z, C.w});
          const uint32_t e = base + jl;
          net.push(make_key(d01.x, e, keep));
          net.push(make_key(d01.y, e + 1, keep));
          net.push(make_key(d23.x, e + 2, keep));
          net.push(make_key(d23.y, e + 3, keep));
          X = Xn; Y = Yn; Z = Zn; C = Cn;
        }
      }
    }
    if (split) {  // the other half's eight survivors: both halves end up with the same merged set
      const int32_t o0 = __shfl_xor(net.a0, 32, 64), o1 = __shfl_xor(net.a1, 32, 64), o2 = __shfl_xor(net.a2, 32, 64), o3 = __shfl_xor(net.a3, 32, 64);
      const int32_t o4 = __shfl_xor(net.a4, 32, 64), o5 = __shfl_xor(net.a5, 32, 64), o6 = __shfl_xor(net.a6, 32, 64), o7 = __shfl_xor(net.a7, 32, 64);
      net.push(o0); net.push(o1); net.push(o2); net.push(o3); net.push(o4); net.push(o5); net.push(o6); net.push(o7);
      if (split4) {
        const int32_t p0 = __shfl_xor(net.a0, 16, 64), p1 = __shfl_xor(net.a1, 16, 64), p2 = __shfl_xor(net.a2, 16, 64), p3 = __shfl_xor(net.a3, 16, 64);
        const int32_t p4 = __shfl_xor(net.a4, 16, 64), p5 = __shfl_xor(net.a5, 16, 64), p6 = __shfl_xor(net.a6, 16, 64), p7 = __shfl_xor(net.a7, 16, 64);
        net.push(p0); net.push(p1); net.push(p2); net.push(p3); net.push(p4); net.push(p5); net.push(p6); net.push(p7);
      }
    }
    if (stamp) { ts[3] = wall_clock64(); acc[2] += ts[3] - ts[2]; }
    // survivors: position in the enumeration -> canonical index (all lanes compute, owners commit).  A single-piece
    // group still has its candidates' indices in LDS; a streamed one goes back through the row table.
    const int32_t ks[8] = {net.a0, net.a1, net.a2, net.a3, net.a4, net.a5, net.a6, net.a7};
    uint32_t gi[8];
    if (filt || total <= kTileCand) {
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        const uint32_t jl = (uint32_t)ks[t] & kKeyIdxMask;
        gi[t] = (ks[t] == kKeyEmpty || jl >= total_eff) ? 0xFFFFFFFFu : ti[jl < kTileCand ? jl : 0];
      }
    } else {
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        const uint32_t jl = (uint32_t)ks[t] & kKeyIdxMask;
        int r = 0;
#pragma unroll
        for (int step = 16; step >= 1; step >>= 1) r = (r + step < 32 && rowoff[r + step] <= jl) ? r + step : r;
        gi[t] = (ks[t] == kKeyEmpty || jl >= total) ? 0xFFFFFFFFu : rowbeg[r] + (jl - rowoff[r]);
      }
    }
    if (mine) {
      g0 = gi[0]; g1 = gi[1]; g2 = gi[2]; g3 = gi[3]; g4 = gi[4]; g5 = gi[5]; g6 = gi[6]; g7 = gi[7];
      k6 = net.a5; k8 = net.a7;
      scanned = true;
      if (pass == 1) {
        cov2 = 1e30f;  // the block contains the lane's whole gate ball by construction
      } else {
        // distance to the faces of the scanned block; a face on the cube's boundary has nothing of the cube behind it
        float cv = 1e15f;
        if (bx0 > 0) cv = fminf(cv, ux - (float)bx0 * cell);
        if (bx1 < nc - 1) cv = fminf(cv, (float)(bx1 + 1) * cell - ux);
        if (by0 > 0) cv = fminf(cv, uy - (float)by0 * cell);
        if (by1 < nc - 1) cv = fminf(cv, (float)(by1 + 1) * cell - uy);
        if (bz0 > 0) cv = fminf(cv, uz - (float)bz0 * cell);
        if (bz1 < nc - 1) cv = fminf(cv, (float)(bz1 + 1) * cell - uz);
        cv = fmaxf(cv - 1e-4f, 0.f);  // cell membership of a map point is decided in fp64 on its own coordinates: keep a margin
        cov2 = cv * cv;
      }
      if (filt) cov2 = fminf(cov2, r_cover * r_cover);  // (candidates beyond r_cover of every lane of the group were not staged)
    }
    if (stamp) { ts[1] = wall_clock64(); acc[3] += ts[1] - ts[3]; }
  }
  // exact re-rank of the survivors + certification
  if (scanned && !need_exact && !(abl & 4)) {
    const uint32_t gs[8] = {g0, g1, g2, g3, g4, g5, g6, g7};
    const int v = certify(pass, gs, k6, k8, cov2, top);
    if (v == 1 || v == 2) resolved = true;
    if (v == 2) too_far_certain = true;
    if (v == 3) need_exact = true;
  } else if (scanned && (abl & 4)) {
    resolved = true;
  }
  }  // pass loop
  if (stamp) ts[2] = wall_clock64();
  __builtin_amdgcn_s_setprio(0);
  if (PROF && lane == 0) { atomicAdd(&lh[16], n_groups); atomicAdd(&lh[18], (int)(n_scanned >> 4)); atomicAdd(&lh[19], 1); }

  if constexpr (!packed) {
    // What the group passes left (8 near-equidistant candidates, a block with more than 2048 candidates): the same exact answer
    // from the wave-cooperative scan -- round 5; the per-lane scan it replaces cost 60 us per lane and ended whole sweeps (open
    // scene, 0.5 m / 5 degree guesses: one chunk of 85 us in a 50 us sweep)
    unsigned long long left = __ballot(valid_q && (!split || lane < (split4 ? 16 : 32)) && c.slot >= 0 && (need_exact || !resolved));
    if (PROF && left && lane == 0) atomicAdd(&lh[17], (int)__popcll(left));
    while (left) {
      const int L = __ffsll((long long)left) - 1;
      left &= left - 1ull;
      coop_exact_scan(L);
    }
  }
  if (valid_q && (!split || lane < (split4 ? 16 : 32))) {
    int status;
    if (c.slot < 0) {
      status = SO_MATCH_NOT_ENOUGH;  // LidarSlam.cpp:736-739
    } else {
      const float d2_4 = __uint_as_float((uint32_t)(top.b4 >> 32));
      if ((abl & 1) || too_far_certain || top.b4 == ~0ull || (double)d2_4 > (double)mp.sq_max_dist_f) {
        status = SO_MATCH_TOO_FAR;   // LidarSlam.cpp:741-744 (d2[4] stays FLT_MAX with < 5 points)
      } else {
        status = SO_MATCH_PENDING;   // five neighbours inside the gate: the plane fit runs in plane_eval_kernel
        uint32_t* o = nbr5 + (size_t)5 * oi;  // (streaming stores: read by the next launch only -- nothing to write back at kernel end)
        __builtin_nontemporal_store((uint32_t)top.b0, o); __builtin_nontemporal_store((uint32_t)top.b1, o + 1);
        __builtin_nontemporal_store((uint32_t)top.b2, o + 2); __builtin_nontemporal_store((uint32_t)top.b3, o + 3);
        __builtin_nontemporal_store((uint32_t)top.b4, o + 4);
      }
    }
    __builtin_nontemporal_store((uint8_t)status, &corr.status[oi]);
  }
  if (stamp) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    ts[3] = wall_clock64(); acc[4] += ts[3] - ts[2]; ++n_mine;
    const unsigned long long nfb = packed ? n_left_stat : __popcll(__ballot(valid_q && c.slot >= 0 && (need_exact || !resolved)));
    if (ts[3] - ts[0] > t_maxchunk) { t_maxchunk = ts[3] - ts[0]; max_info = ((unsigned long long)n_scanned << 32) | (nfb << 16) | (unsigned long long)n_groups; }
    n_fb_total += nfb;
    n_cand_total += n_scanned; n_groups_total += n_groups; n_q_total += __popcll(__ballot(valid_q));
  }
  };  // do_item
  for (uint32_t chunk = blockIdx.x * 4 + wv; chunk < n_chunks; chunk += gridDim.x * 4) {
    if (SO_KNN_PACK && chunk < n_packed) do_item(std::true_type{}, chunk); else do_item(std::false_type{}, chunk);  // (n_packed = 0 unless `pack`)
  }
  if (stamp && lane_k == 0) {  // one record per wavefront, no atomics (they would perturb the measurement)
    unsigned long long* d = mp.kdbg + ((size_t)(begin ? 0 : (st->outer_iter & 1)) * gridDim.x * 4 + blockIdx.x * 4 + wv) * 16;
    d[0] = t_first; d[1] = wall_clock64();
    for (int i = 0; i < 5; ++i) d[2 + i] = acc[i];
    d[15] = clock64() - c_first;  // shader-clock ticks over the wavefront's life (d[1] - d[0] = the same span at 100 MHz)
    d[7] = n_mine; d[8] = t_maxchunk; d[9] = n_cand_total; d[10] = n_q_total; d[11] = n_groups_total; d[12] = n_pass2; d[13] = max_info;
    d[14] = n_fb_total | ((unsigned long long)(__builtin_amdgcn_s_getreg(63492) & 0xFFFFu) << 32) | ((unsigned long long)(__builtin_amdgcn_s_getreg(63508) & 0xFu) << 48);  // + HW_ID[15:0] (wave, simd, cu, se), XCC_ID: where the wavefront ran
  }
  if (PROF) {
    __syncthreads();
    if (threadIdx.x >= 16 && threadIdx.x < 24 && lh[threadIdx.x])  // kernel statistics only; the histograms are built by the fit pass
      atomicAdd(&hist[(blockIdx.x % kHistReplicas) * kHistStride + threadIdx.x], lh[threadIdx.x]);
  }
}

// ------------------------------------------------------------------------------------------------
// knn_query_wave_kernel -- the sweep of a SMALL scan (round 6): one WAVEFRONT per scan point, no binning at all.
//
// The stock operating point of the node (config/os1_128.yaml:26-28, livox_mid360.yaml:26-28: max_surface_features 2000 / 4000 of a
// pre-filtered cloud of 9 - 13 k points) keeps a few thousand queries.  The chunked sweep above is built for 131 072 of them: three
// binning launches the host cannot enqueue as fast as the device finishes them (21 us), then wavefronts that serve 27 queries each
// through six dependent memory round trips -- 60 workgroups' worth of work on a 256-unit chip.  With <= 4 096 kept queries every
// query can have a wavefront of its own, all resident at once: sampling rule (LidarSlam.cpp:346-359), world transform (:397-398),
// cube + cell (LocalMap.h:488-507), then the (clamped) 3 x 3 x 3 cells around the query -- every map point inside the gate ball lies
// there, one cell >= the gate radius -- as <= 9 x-runs dealt to the 64 lanes, exact distances (octree.h:93-102), lane-local top 5,
// five wavefront minima, distance gate (LidarSlam.cpp:741).  Four dependent round trips, no binning launches, no chunk list; the
// same exact lists (ties by canonical index) and the same status bytes + neighbour lists as knn_plane_kernel leaves for the fit pass.
// BEGIN: first launch of a registration -- the prologue rides on it (workgroup 0), the pose comes from the kernel arguments, and the
//        points the sampling rule drops get their DROPPED status bytes.
// ------------------------------------------------------------------------------------------------
// first scan point of wavefront k's share: a monotone function of k, evaluated identically by wavefronts k and k + 1, so the shares
// [first(k), first(k + 1)) partition the scan whatever the rounding does
__device__ __forceinline__ uint32_t query_wave_first(uint32_t k, uint32_t n, double per_wave) {
  if (per_wave <= 1.0) return k < n ? k : n;  // (no sampling: wavefront k owns point k)
  const double f = ceil((double)k * per_wave);
  return f < (double)n ? (uint32_t)f : n;
}
// DISPATCH (round 6, second step): under the sampling rule only ~max_surface_features of the n points are searched (1 993 of 13 275 at
// the stock operating point), and a wavefront per POINT spent 4 us of every sweep -- searching or not -- on dispatching 3 300
// workgroups that mostly return at once.  The rule keeps point i iff frac(i * rate) + 0.001 <= rate (LidarSlam.cpp:353-359): about
// one point in every run of 1 / rate.  Wavefront k therefore owns the points [first(k), first(k + 1)), first(k) = ceil(k / rate):
// its lanes apply the rule -- the reference's own fp64 test, not a closed form -- to one point each and the wavefront searches the
// points that pass, one after the other (one, as a rule; none or two where the rounding of i * rate falls that way: the shares
// partition the scan, so the set of searched points is exactly the rule's).  ~max_surface_features wavefronts instead of n.
template <bool BEGIN>
__global__ __launch_bounds__(256) void knn_query_wave_kernel(const float* __restrict__ scan, uint32_t n, uint32_t n_waves, double per_wave,
                                                             const DevState* __restrict__ st, DevState* st_begin,
                                                             RegBeginArgs a, int32_t* __restrict__ hist, const float4* __restrict__ mpts,
                                                             const uint32_t* __restrict__ mcell_start, DevMapView map, MatchParams mp, int max_surface_features,
                                                             uint8_t* __restrict__ status, uint32_t* __restrict__ nbr5) {
  __shared__ uint32_t rowtab[4][2][20];  // per wavefront: exclusive candidate offsets [17] and first canonical index [16] of the nine x-runs
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const uint32_t k = blockIdx.x * 4u + (uint32_t)wv;  // this wavefront's share of the scan
  const uint32_t i_lo = query_wave_first(k, n, per_wave), i_hi = k < n_waves ? query_wave_first(k + 1u, n, per_wave) : i_lo;
  // every lane's own point of the share and the pose are requested BEFORE the "already converged?" word of the state block is
  // looked at: one memory round trip for the three instead of two (a sweep is five dependent round trips and little else)
  const uint32_t il0 = i_lo + (uint32_t)lane < n ? i_lo + (uint32_t)lane : n - 1u;
  float sx = scan[3 * il0], sy = scan[3 * il0 + 1], sz = scan[3 * il0 + 2];
  double T7[7];
#pragma unroll
  for (int t = 0; t < 7; ++t) T7[t] = BEGIN ? (a.chain_expect ? st->T_chain[t] : a.pose[t]) : st->T[t];
  if (mp.chain_expect && st->done_count != mp.chain_expect) return;  // chained registration whose predecessor was not over: no-op
  if (BEGIN) {
    if (a.chain_expect && st->done_count != a.chain_expect) return;  // (its guess: what the last solve of the registration in front left in T_chain)
    if (blockIdx.x == 0) {
      hist[threadIdx.x] = 0; hist[256 + threadIdx.x] = 0;
      reg_begin_state(st_begin, a, (int)threadIdx.x);
    }
  } else {
    if (st->reg_done) return;  // the registration already converged: this launch is a no-op
    // the report of the previous outer iteration, left to this launch by its solve (MatchParams::publish_prev): by a workgroup of its own
    // behind the searching ones (the launch has one more) -- the write-back + system fence + PCIe stores take 2.7 us, which in front of
    // workgroup 0's searches made a 9 us sweep a 12 us one
    if (mp.publish_prev && blockIdx.x + 1u == gridDim.x && st->outer_iter > 0)
      publish_state_to(mp.hring[(st->outer_iter - 1) & 1], st, mp.seq_base | (unsigned long long)st->outer_iter, (int)threadIdx.x, 256);
  }
  if (i_lo >= i_hi) return;
  const Pose pose = pose_from_array(T7);
  uint32_t* rowoff = rowtab[wv][0];
  uint32_t* rowbeg = rowtab[wv][1];
  const int nc = map.nc;
  for (uint32_t base = i_lo; base < i_hi; base += 64u) {  // (one trip unless n / max_surface_features > 64)
    const uint32_t il = base + (uint32_t)lane;
    if (base != i_lo) { const uint32_t ic = il < n ? il : n - 1u; sx = scan[3 * ic]; sy = scan[3 * ic + 1]; sz = scan[3 * ic + 2]; }
    const bool keep = il < i_hi && sampling_keeps(il, n, max_surface_features);
    if (BEGIN && il < i_hi && !keep) status[il] = SO_MATCH_DROPPED;  // (the rule does not depend on the pose: once per registration)
    unsigned long long todo = __ballot(keep);
    while (todo) {
      const int li = __builtin_ctzll(todo);  // wavefront-uniform: the lane that holds the next point of the share the rule keeps
      const uint32_t i = base + (uint32_t)li;
      todo &= todo - 1ull;
      const float fx = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(sx), li));
      const float fy = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(sy), li));
      const float fz = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(sz), li));
      double pw[3];
      quat_rotate<double>(pose.q, (double)fx, (double)fy, (double)fz, pw[0], pw[1], pw[2]);  // LidarSlam.cpp:397-398
      pw[0] += pose.t[0]; pw[1] += pose.t[1]; pw[2] += pose.t[2];
      const float qx = (float)pw[0], qy = (float)pw[1], qz = (float)pw[2];                   // LidarSlam.cpp:728-731
      const CellRef c = locate(map, qx, qy, qz);
      if (c.slot < 0) {  // outside the window / no tree: LidarSlam.cpp:736-739
        if (lane == 0) __builtin_nontemporal_store((uint8_t)SO_MATCH_NOT_ENOUGH, &status[i]);
        continue;
      }
      const int x0 = c.cx > 0 ? c.cx - 1 : 0, x1 = c.cx < nc - 1 ? c.cx + 1 : nc - 1;
      uint32_t vb = 0, vl = 0;
      if (lane < 9) {
        const int y = c.cy + (lane % 3) - 1, z = c.cz + (lane / 3) - 1;
        if (y >= 0 && y < nc && z >= 0 && z < nc) {
          const uint32_t* row = mcell_start + (size_t)c.slot * map.ncell1 + ((size_t)z * nc + y) * nc;
          vb = row[x0]; vl = row[x1 + 1] - vb;
        }
      }
      uint32_t inc = vl;  // inclusive scan over lanes 0..15 (the nine runs sit in the first row of 16 lanes)
      inc += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)inc, 0x111, 0xF, 0xF, true);
      inc += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)inc, 0x112, 0xF, 0xF, true);
      inc += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)inc, 0x114, 0xF, 0xF, true);
      inc += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)inc, 0x118, 0xF, 0xF, true);
      const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)inc, 15);
      __builtin_amdgcn_wave_barrier();  // (a second point of the share: the table of the first has been read by every lane)
      if (lane < 16) { rowoff[lane] = lane < 9 ? inc - vl : total; rowbeg[lane] = vb; }
      if (lane == 0) rowoff[16] = total;
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
      Top5 loc;
      loc.init();
      for (uint32_t t0 = 0; t0 < total; t0 += 256u) {  // four loads of a lane in flight: a block of <= 256 points is one round trip
        float ax_[4], ay_[4], az_[4];
        uint32_t cn[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const uint32_t t = t0 + 64u * (uint32_t)u + (uint32_t)lane;
          cn[u] = 0xFFFFFFFFu; ax_[u] = ay_[u] = az_[u] = 0.f;
          if (t < total) {
            int r = 0;
#pragma unroll
            for (int step = 8; step >= 1; step >>= 1) r = (r + step < 16 && rowoff[r + step] <= t) ? r + step : r;
            cn[u] = rowbeg[r] + (t - rowoff[r]);
            const float4 p = mpts[cn[u]];
            ax_[u] = p.x; ay_[u] = p.y; az_[u] = p.z;
          }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
          if (cn[u] != 0xFFFFFFFFu) loc.insert(((unsigned long long)__float_as_uint(l2_d2(qx, qy, qz, ax_[u], ay_[u], az_[u])) << 32) | cn[u]);
      }
      unsigned long long m5[5];
#pragma unroll
      for (int t = 0; t < 5; ++t) {
        m5[t] = wave_min_u64(loc.b0);
        if (loc.b0 == m5[t] && m5[t] != ~0ull) { loc.b0 = loc.b1; loc.b1 = loc.b2; loc.b2 = loc.b3; loc.b3 = loc.b4; loc.b4 = ~0ull; }  // (keys are unique)
      }
      if (lane == 0) {
        const float d2_4 = __uint_as_float((uint32_t)(m5[4] >> 32));
        int stq = SO_MATCH_PENDING;  // five neighbours inside the gate: the plane fit runs in slot 0 of the solve
        if (m5[4] == ~0ull || (double)d2_4 > (double)mp.sq_max_dist_f) stq = SO_MATCH_TOO_FAR;  // LidarSlam.cpp:741-744 (d2[4] stays FLT_MAX with < 5 points)
        else {
          uint32_t* o = nbr5 + (size_t)5 * i;
          __builtin_nontemporal_store((uint32_t)m5[0], o); __builtin_nontemporal_store((uint32_t)m5[1], o + 1);
          __builtin_nontemporal_store((uint32_t)m5[2], o + 2); __builtin_nontemporal_store((uint32_t)m5[3], o + 3);
          __builtin_nontemporal_store((uint32_t)m5[4], o + 4);
        }
        __builtin_nontemporal_store((uint8_t)stq, &status[i]);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// LM evaluation: fused cost + J^T J + J^T r
// ------------------------------------------------------------------------------------------------
constexpr int kNAcc = 29;  // cost, count, Jtr[6], JtJ[21]

// Which queries a workgroup of the evaluation / solve launches walks: thread `tid` of (virtual) workgroup vb, trip q, grid of V
// workgroups -- 256 consecutive scan points per workgroup and trip.  With V >= n / 256 every workgroup makes one trip over
// the same 256 points whatever V is, so launches with different grids (the per-evaluation launches use 256 workgroups for
// slots >= 1, the persistent and batched solves the grid of slot 0) add up the same records: bit-identical sums.
// (Round 4 measured the alternative -- the scan dealt in 64-point segments, segment (4 q + wave) V + vb, so that every
//  workgroup samples the whole sweep: the fit pass's wait for its slowest workgroup did not shrink (3.3 -> 4.5 us in the
//  in-kernel stamps: it is the start skew of the 256 workgroups, not rejected ring segments -- 94 % of this scene's points
//  are accepted), and the sums became grid-dependent.  Not kept.)
__device__ __forceinline__ uint32_t query_of(uint32_t vb, uint32_t V, int tid, uint32_t q) { return vb * 256u + (uint32_t)tid + q * V * 256u; }

// slot 0 evaluates at the outer pose T (lm_begin); slots >= 1 evaluate the candidate requested by the LM
// controller and are no-ops once the controller has finished (or the registration has converged).
__device__ __forceinline__ bool eval_slot_active(const DevState* st, int slot) {
  return !st->reg_done && (slot == 0 || st->lm_more);
}

// The Ceres-equivalent LM controller plus the outer ICP bookkeeping (LidarSlam.cpp:119-148, 242-251).
// Executed by ONE thread on an LDS copy of the controller state.
struct LmCtl {       // LDS copy of the DevState fields the controller reads (prefetched with the state: no global
  double T[7];       // round trips inside the single-thread controller)
  int32_t lm_max, outer_iter, max_outer, pad;
};

// A solve has ended: T_w_lidar <- optimised pose, iteration statistics, termination rule of the outer loop (one thread).
// Returns 1 when the registration is over too (the caller then stores the final normal equations).
__device__ __forceinline__ int lm_solve_finished(DevState* st, const LmCtl& ctl, const double x[7], double count, int lm_iterations,
                                                 int num_successful, int termination, double initial_cost, double x_cost, const LmSums& sums) {
  for (int i = 0; i < 7; ++i) st->T[i] = x[i];
  const int o = ctl.outer_iter;
  DevIterStats& is = st->iters[o < 16 ? o : 15];
  relative_motion(ctl.T, x, is.translation_norm, is.rotation_norm);
  is.num_surf = (int32_t)count; is.lm_iterations = lm_iterations; is.num_successful = num_successful;
  is.termination = termination; is.initial_cost = initial_cost; is.final_cost = x_cost;
  for (int h = 0; h < 7; ++h) is.reject_hist[h] = (int32_t)sums.hist[h];
  for (int h = 0; h < 9; ++h) is.obs_hist[h] = (int32_t)sums.hist[7 + h];
  for (int i = 0; i < 7; ++i) is.pose_after[i] = x[i];
  st->outer_iter = o + 1;
  st->n_iterations = o + 1;
  if (num_successful == 1 || o + 1 >= ctl.max_outer) {  // LidarSlam.cpp:141
    st->reg_done = 1;
    for (int i = 0; i < 7; ++i) st->T_final[i] = x[i];  // (what a chained registration behind this one starts from)
    st->done_count = st->done_count + 1u;
    return 1;
  }
  return 0;
}

__device__ __forceinline__ int lm_control_regs(int slot, DevState* st, LmState& S, const LmSums& sums, const LmCtl& ctl, bool persist,
                                               int* reg_done_out = nullptr) {
  if (reg_done_out) *reg_done_out = 0;
  // persist (solve_kernel): the next pose travels in the hand-off record and nobody reads lm_more, so a pass that is
  // followed by another one issues no global store at all (a store would have to drain before the next barrier)
  int more;
  double unused_pose[7];
  double* next_pose = persist ? unused_pose : st->eval_pose;
  if (slot == 0) more = lm_begin(S, ctl.T, sums, ctl.lm_max, next_pose);
#ifdef SO_LM_STAMPS
  else more = lm_feed(S, sums, next_pose, slot == 1 ? st->dbg : nullptr);
#else
  else more = lm_feed(S, sums, next_pose);
#endif
  if (!persist || !more) st->lm_more = more;
  if (more) return 1;
  const int rd = lm_solve_finished(st, ctl, S.x, S.count, S.lm_iterations, S.num_successful, S.termination, S.initial_cost, S.x_cost, sums);
  if (rd) {
    if (reg_done_out) *reg_done_out = 1;
    const bool have = S.count > 0;
    for (int i = 0; i < 36; ++i) st->JtJ[i] = have ? S.H[i] : 0.0;
    for (int i = 0; i < 6; ++i) st->Jtr[i] = have ? S.g[i] : 0.0;
  }
  return 0;
}

typedef unsigned int u4v __attribute__((ext_vector_type(4)));
// 16-byte agent-scope (sc1) store / load: one access, bypassing the non-coherent per-XCD L2 state [MI355X guide, G16]
__device__ __forceinline__ void store16_sc1(u4v* p, u4v v) { asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ u4v load16_sc1(const u4v* p) {
  u4v v;
  asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  return v;
}

// system-scope 16-byte store / load (sc0 sc1): the peer-exchange inboxes are written over xGMI by other devices
__device__ __forceinline__ void store16_sys(u4v* p, u4v v) { asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ u4v load16_sys(const u4v* p) {
  u4v v;
  asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  return v;
}

#define SO_LM_INLINE __forceinline__  // (out of line, the callee-saved registers go through scratch: +1 us per call)
// hand / want / pose_out (persistent solve): the controller's thread publishes the hand-off record {next pose, more?}
// straight from its registers, BEFORE the state goes back to LDS -- the other workgroups are already evaluating the
// next pose while this thread is still tidying up.
__device__ SO_LM_INLINE int lm_control(int slot, DevState* st, LmState& S_lds, const LmSums& sums_lds, const LmCtl& ctl, bool persist = false,
                                       u4v* hand = nullptr, unsigned long long want = 0, double* pose_out = nullptr, int* reg_done_out = nullptr) {
  // register copies: the controller is one thread's serial fp64 chain, and every LDS round trip inside it (~100 cycles,
  // nothing to overlap with) would sit on the critical path of the whole device
  // (the sums stay in LDS: they are read once, where a successful step adopts them; H is symmetric: only its upper
  //  triangle lives in registers, the mirror image is rebuilt on the way out -- the full state plus the sums would not
  //  fit the 256 architectural registers and the spill traffic would sit on the same critical path)
  LmState S;
#pragma unroll
  for (int i = 0; i < 7; ++i) { S.x[i] = S_lds.x[i]; S.cand[i] = S_lds.cand[i]; }
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    S.g[i] = S_lds.g[i]; S.scale[i] = S_lds.scale[i]; S.diag[i] = S_lds.diag[i];
#pragma unroll
    for (int j = i; j < 6; ++j) { S.H[6 * i + j] = S_lds.H[6 * i + j]; S.H[6 * j + i] = S.H[6 * i + j]; }
  }
  S.x_cost = S_lds.x_cost; S.x_norm = S_lds.x_norm; S.inv_radius = S_lds.inv_radius; S.decrease_factor = S_lds.decrease_factor;
  S.inv_model_cost_change = S_lds.inv_model_cost_change; S.step_norm = S_lds.step_norm; S.cand_norm = S_lds.cand_norm;
  S.model_cost_change = S_lds.model_cost_change; S.initial_cost = S_lds.initial_cost; S.count = S_lds.count;
  S.iter = S_lds.iter; S.max_iter = S_lds.max_iter; S.reuse_diagonal = S_lds.reuse_diagonal; S.invalid_steps = S_lds.invalid_steps;
  S.num_successful = S_lds.num_successful; S.termination = S_lds.termination; S.done = S_lds.done; S.lm_iterations = S_lds.lm_iterations;
  const int more_ = lm_control_regs(slot, st, S, sums_lds, ctl, persist, reg_done_out);
  if (hand) {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const unsigned long long val = (k < 7) ? (unsigned long long)__double_as_longlong(S.cand[k < 7 ? k : 0]) : (unsigned long long)more_;
      const u4v v = {(unsigned int)val, (unsigned int)(val >> 32), (unsigned int)want, (unsigned int)(want >> 32)};
      store16_sc1(hand + k, v);
    }
#pragma unroll
    for (int k = 0; k < 7; ++k) pose_out[k] = S.cand[k];
  }
#pragma unroll
  for (int i = 0; i < 7; ++i) { S_lds.x[i] = S.x[i]; S_lds.cand[i] = S.cand[i]; }
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    S_lds.g[i] = S.g[i]; S_lds.scale[i] = S.scale[i]; S_lds.diag[i] = S.diag[i];
#pragma unroll
    for (int j = i; j < 6; ++j) { S_lds.H[6 * i + j] = S.H[6 * i + j]; S_lds.H[6 * j + i] = S.H[6 * i + j]; }
  }
  S_lds.x_cost = S.x_cost; S_lds.x_norm = S.x_norm; S_lds.inv_radius = S.inv_radius; S_lds.decrease_factor = S.decrease_factor;
  S_lds.inv_model_cost_change = S.inv_model_cost_change; S_lds.step_norm = S.step_norm; S_lds.cand_norm = S.cand_norm;
  S_lds.model_cost_change = S.model_cost_change; S_lds.initial_cost = S.initial_cost; S_lds.count = S.count;
  S_lds.iter = S.iter; S_lds.max_iter = S.max_iter; S_lds.reuse_diagonal = S.reuse_diagonal; S_lds.invalid_steps = S.invalid_steps;
  S_lds.num_successful = S.num_successful; S_lds.termination = S.termination; S_lds.done = S.done; S_lds.lm_iterations = S.lm_iterations;
#ifdef SO_LM_STAMPS
  if (slot == 1) st->dbg[7] = wall_clock64();
#endif
  return more_;
}

// ---- Round 5: the controller across the lanes of ONE wavefront (persistent solve, single registration).
// The one-thread controller above is an instruction stream of ~750 fp64 operations and LDS moves on one lane -- 3.7 us per pass
// (in-kernel stamps), 2.5 of them in front of the hand-off every other workgroup is waiting for.  Most of that stream is WIDE
// work done element after element: ~70 state words in and out of LDS, 27 sums unpacked, 21 entries scaled twice, six
// reciprocal-square-root scale factors in the first pass, eight hand-off chunks.  Here lane l < 36 owns element (l / 6, l % 6) of
// H, lane 36 + j element j of g: unpacking, scaling and the damped matrix are one or two instructions for the whole wavefront,
// the six Jacobi scale factors one sqrt + division, the hand-off one store.  What is a dependent chain -- the 6x6 Cholesky
// (six pivots through rsqrt), the substitutions, the model cost change, the quaternion update -- stays the code of
// lm_solver.h, run by every lane on the same (gathered) operands, so every element goes through the operations of the
// one-thread controller in the same order: identical bits (the batched solve and the per-evaluation launches keep the
// one-thread form; tests/test_gpu_configs.py compares them bit for bit).  The state's home is the LDS copy; what the next
// lm_feed needs beyond the sums (lm_after_candidate) and the store-back run AFTER the hand-off has been published.
__device__ __forceinline__ int lm_control_wave(int slot, DevState* st, LmState& S, const LmSums& sums, const LmCtl& ctl, double* gather /* LDS, >= 56 doubles */,
                                               u4v* hand, unsigned long long want, double* pose_out, int* reg_done_out, int lane,
                                               unsigned long long* dbg = nullptr) {
  const bool is_mat = lane < 36, is_vec = lane >= 36 && lane < 42;
  const int li = is_mat ? lane / 6 : 0, lj = is_mat ? lane - 6 * li : (is_vec ? lane - 36 : 0);
  const int la = li < lj ? li : lj, lb = li < lj ? lj : li;          // (row, column) of the element's upper-triangle twin
  const int tri = 6 * la - (la * (la - 1)) / 2 + (lb - la);           // its index in LmSums::JtJ
  const bool is_diag = is_mat && li == lj;
  // ---- uniform state (every lane holds the same value), element state (one per lane)
  double x[7], cand[7];
  double x_cost, x_norm, inv_radius, decrease_factor, mcc_state, initial_cost, count, inv_mcc, step_norm, cand_norm;
  int iter, max_iter, reuse_diagonal, invalid_steps, num_successful, termination, done, lm_iterations;
  double Hl = 0, gl = 0, diag_l = 0, sa = 1.0, sb = 1.0;              // H element / g element / diag (diagonal lanes) / scale[min(i,j)], scale[max(i,j)]
  int more = 0;
  bool propose = false;
  if (slot == 0) {  // ---------------- lm_begin
#pragma unroll
    for (int i = 0; i < 7; ++i) { x[i] = ctl.T[i]; cand[i] = x[i]; }
    iter = 0; max_iter = ctl.lm_max; reuse_diagonal = 0; invalid_steps = 0; num_successful = 0; termination = 0; done = 0; lm_iterations = 0;
    inv_radius = LmConst::kInitialInvRadius; decrease_factor = 2.0; mcc_state = 0; inv_mcc = 0; step_norm = 0; cand_norm = 0;
    count = sums.count; x_cost = sums.cost; initial_cost = x_cost; x_norm = 0;
    if (is_mat) Hl = sums.JtJ[tri];
    if (is_vec) gl = sums.Jtr[lj];
    if (!(count > 0)) { termination = 4; done = 1; }  // no residual blocks: nothing to minimise (scale 1, diag 0 go to the state below)
    else {
      // jacobi_scaling, fixed at iteration 0: 1 / (1 + sqrt(H_jj)) on the diagonal lanes, handed to the element lanes through the state
      const double sc = 1.0 / (1.0 + sqrt(Hl));
      if (is_diag) S.scale[li] = sc;
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      sa = S.scale[is_vec ? lj : la]; sb = S.scale[lb];
      double n2 = 0;
#pragma unroll
      for (int i = 0; i < 7; ++i) n2 = SO_FMA(x[i], x[i], n2);
      x_norm = sqrt(n2);
      propose = true;
    }
  } else {          // ---------------- lm_feed
#pragma unroll
    for (int i = 0; i < 7; ++i) { x[i] = S.x[i]; cand[i] = S.cand[i]; }
    x_cost = S.x_cost; x_norm = S.x_norm; inv_radius = S.inv_radius; decrease_factor = S.decrease_factor; mcc_state = S.model_cost_change;
    initial_cost = S.initial_cost; count = S.count; inv_mcc = S.inv_model_cost_change; step_norm = S.step_norm; cand_norm = S.cand_norm;
    iter = S.iter; max_iter = S.max_iter; reuse_diagonal = S.reuse_diagonal; invalid_steps = S.invalid_steps; num_successful = S.num_successful;
    termination = S.termination; done = S.done; lm_iterations = S.lm_iterations;
    if (is_mat) { Hl = S.H[lane]; sa = S.scale[la]; sb = S.scale[lb]; }
    if (is_vec) { gl = S.g[lj]; sa = S.scale[lj]; }
    if (is_diag) diag_l = S.diag[li];
    if (!done) {
      SO_LM_STAMP(dbg, 0);
      const double cand_cost = sums.cost;
      if (step_norm <= LmConst::kParameterTolerance * (x_norm + LmConst::kParameterTolerance)) { termination = 2; done = 1; }  // ParameterToleranceReached
      else {
        const double cost_change = x_cost - cand_cost;
        if (fabs(cost_change) <= LmConst::kFunctionTolerance * x_cost) { termination = 1; done = 1; }  // FunctionToleranceReached
        else {
          const double rel = cost_change * inv_mcc;  // StepQuality
          SO_LM_STAMP(dbg, 1);
          if (rel > LmConst::kMinRelativeDecrease) {  // HandleSuccessfulStep
#pragma unroll
            for (int i = 0; i < 7; ++i) x[i] = cand[i];
            x_norm = cand_norm; x_cost = cand_cost;
            if (is_mat) Hl = sums.JtJ[tri];
            if (is_vec) gl = sums.Jtr[lj];
            num_successful++;
            const double u = 2.0 * rel - 1.0;
            double f = 1.0 - u * u * u;
            if (f < 1.0 / 3.0) f = 1.0 / 3.0;
            inv_radius = inv_radius * f;
            if (inv_radius < LmConst::kMinInvRadius) inv_radius = LmConst::kMinInvRadius;
            decrease_factor = 2.0; reuse_diagonal = 0;
            if (iter >= max_iter) { termination = 0; done = 1; }
            else propose = true;  // (gradient test below)
          } else {  // StepRejected
            inv_radius = inv_radius * decrease_factor; decrease_factor *= 2.0; reuse_diagonal = 1;
            propose = true;
          }
          SO_LM_STAMP(dbg, 2);
        }
      }
    }
  }
  // gradient_max_norm <= gradient_tolerance after lm_begin / an accepted step (lm_gradient_converged: the fast exit on the lanes of
  // g[0..2], the full test -- g gathered to every lane -- only when it does not decide)
  if (propose && reuse_diagonal == 0) {
    const double xi = lane == 36 ? x[0] : (lane == 37 ? x[1] : x[2]);
    const bool big = lane >= 36 && lane < 39 && fabs(gl) > 1e-6 && fabs(xi) < 1e5;
    if (__ballot(big) == 0ull) {
      if (is_vec) gather[48 + lj] = gl;
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      double gu[6];
#pragma unroll
      for (int i = 0; i < 6; ++i) gu[i] = gather[48 + i];
      if (lm_gradient_max_norm(x, gu) <= LmConst::kGradientTolerance) { termination = 3; done = 1; propose = false; }
    }
  }
  // ---------------- lm_propose
  double scale_u[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) scale_u[i] = 1.0;
  if (propose) {
#pragma unroll
    for (int i = 0; i < 6; ++i) scale_u[i] = S.scale[i];
  }
  while (propose) {
    if (iter >= max_iter) { termination = 0; done = 1; break; }                           // MaxSolverIterationsReached
    if (inv_radius >= LmConst::kMaxInvRadius) { termination = 5; done = 1; break; }       // MinTrustRegionRadiusReached
    iter++; lm_iterations = iter;
    if (!reuse_diagonal) {
      double v = Hl * sa * sa;  // (diagonal lanes: sa == sb == scale[j])
      v = v < LmConst::kMinLmDiagonal ? LmConst::kMinLmDiagonal : v;
      diag_l = v > LmConst::kMaxLmDiagonal ? LmConst::kMaxLmDiagonal : v;
    }
    const double Hs = Hl * sa * sb;                       // Hs = S H S
    const double damp = diag_l * inv_radius;
    const double Al = is_diag ? Hs + damp : Hs;           // + lm_diagonal^2 = diag / radius
    if (is_mat) gather[lane] = Al;
    if (is_diag) gather[42 + li] = Hs;
    if (is_vec) gather[36 + lj] = gl * sa;                // gs = S g
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    // (only the lower triangle travels into registers -- lm_chol6 reads nothing else --, and the model cost change reads Hs from
    //  the gather area again)
    double A[36], gs[6], y[6], step[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
#pragma unroll
      for (int j = 0; j <= i; ++j) A[6 * i + j] = gather[6 * i + j];
      gs[i] = gather[36 + i];
    }
    SO_LM_STAMP(dbg, 3);
    const bool ok = lm_chol6(A, gs, y);
    SO_LM_STAMP(dbg, 4);
    reuse_diagonal = 1;
    double mcc = 0;
    if (ok) {
      double sHs = 0, sg = 0;
#pragma unroll
      for (int i = 0; i < 6; ++i) { step[i] = -y[i]; sg = SO_FMA(step[i], gs[i], sg); }
#pragma unroll
      for (int i = 0; i < 6; ++i) {  // Hs: off-diagonal entries = those of the damped matrix, the diagonal comes undamped (gather[42 + i])
        double r = 0;
#pragma unroll
        for (int j = i + 1; j < 6; ++j) r = SO_FMA(gather[6 * i + j], step[j], r);
        sHs = SO_FMA(step[i], SO_FMA(gather[42 + i], step[i], 2.0 * r), sHs);
      }
      mcc = SO_FMA(-0.5, sHs, -sg);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // (the next round of the loop rewrites the gather area)
    if (!ok || !(mcc > 0.0)) {  // HandleInvalidStep
      if (++invalid_steps >= LmConst::kMaxConsecutiveInvalidSteps) { termination = 5; done = 1; break; }
      inv_radius *= 2.0;
      continue;
    }
    SO_LM_STAMP(dbg, 5);
    invalid_steps = 0;
    mcc_state = mcc;
    double delta[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) delta[i] = step[i] * scale_u[i];
    pose_plus(x, delta, cand);
    SO_LM_STAMP(dbg, 6);
    more = 1;
    break;
  }
  // ---------------- hand-off {next pose, more?}: one 16-byte chunk per lane 0..7
  {
    double hv = cand[0];
    hv = lane == 1 ? cand[1] : hv; hv = lane == 2 ? cand[2] : hv; hv = lane == 3 ? cand[3] : hv;
    hv = lane == 4 ? cand[4] : hv; hv = lane == 5 ? cand[5] : hv; hv = lane == 6 ? cand[6] : hv;
    const unsigned long long val = lane == 7 ? (unsigned long long)more : (unsigned long long)__double_as_longlong(hv);
    if (lane < 8) {
      const u4v v = {(unsigned int)val, (unsigned int)(val >> 32), (unsigned int)want, (unsigned int)(want >> 32)};
      store16_sc1(hand + lane, v);
    }
    if (lane < 7) pose_out[lane] = hv;
  }
  // ---------------- behind the hand-off: what the next lm_feed needs, the state back to its home
  if (more) {  // lm_after_candidate
    inv_mcc = 1.0 / mcc_state;
    double sn = 0, n2 = 0;
#pragma unroll
    for (int i = 0; i < 7; ++i) sn = SO_FMA(x[i] - cand[i], x[i] - cand[i], sn);
#pragma unroll
    for (int i = 0; i < 7; ++i) n2 = SO_FMA(cand[i], cand[i], n2);
    step_norm = sqrt(sn); cand_norm = sqrt(n2);
  }
  if (is_mat) S.H[lane] = Hl;
  if (is_vec) S.g[lj] = gl;
  if (is_diag) S.diag[li] = diag_l;
  if (slot == 0 && termination == 4 && lane < 6) S.scale[lane] = 1.0;
  if (lane == 0) {
#pragma unroll
    for (int i = 0; i < 7; ++i) { S.x[i] = x[i]; S.cand[i] = cand[i]; }
    S.x_cost = x_cost; S.x_norm = x_norm; S.inv_radius = inv_radius; S.decrease_factor = decrease_factor; S.model_cost_change = mcc_state;
    S.initial_cost = initial_cost; S.count = count; S.inv_model_cost_change = inv_mcc; S.step_norm = step_norm; S.cand_norm = cand_norm;
    S.iter = iter; S.max_iter = max_iter; S.reuse_diagonal = reuse_diagonal; S.invalid_steps = invalid_steps; S.num_successful = num_successful;
    S.termination = termination; S.done = done; S.lm_iterations = lm_iterations;
  }
#ifdef SO_LM_STAMPS
  if (slot == 1 && lane == 0) st->dbg[7] = wall_clock64();
#endif
  if (!more) {  // the solve is over
    int rd = 0;
    if (lane == 0) {
      st->lm_more = 0;
      rd = lm_solve_finished(st, ctl, x, count, lm_iterations, num_successful, termination, initial_cost, x_cost, sums);
      *reg_done_out = rd;
    }
    rd = __builtin_amdgcn_readfirstlane(rd);
    if (rd) {  // final normal equations: H and g as the state holds them, element by element
      const bool have = count > 0;
      if (is_mat) st->JtJ[lane] = have ? Hl : 0.0;
      if (is_vec) st->Jtr[lj] = have ? gl : 0.0;
    }
  } else if (lane == 0) *reg_done_out = 0;
  return more;
}

// threads [first, first+10) fetch the controller's inputs
__device__ __forceinline__ void load_ctl(LmCtl& ctl, const DevState* st, int tid, int first) {
  const int k = tid - first;
  if (k >= 0 && k < 7) ctl.T[k] = st->T[k];
  else if (k == 7) ctl.lm_max = st->lm_max;
  else if (k == 8) ctl.outer_iter = st->outer_iter;
  else if (k == 9) ctl.max_outer = st->max_outer;
}

// End of a solve: publish the state block to the host mirror (see EvalParams::hring).  Called by every thread of the
// controller's workgroup after the controller's global stores; `outer` = index of the outer iteration that just ended.
__device__ __forceinline__ void publish_state(const DevState* st, const EvalParams& ep, int outer, int tid, int nthreads) {
  publish_state_to(ep.hring[outer & 1], st, ep.seq_base | (unsigned long long)(outer + 1), tid, nthreads);
}
__device__ __forceinline__ void publish_state_to(DevState* dst, const DevState* st, unsigned long long seq, int tid, int nthreads) {
  if (!dst) return;
  __threadfence_block();  // the controller thread's stores to *st are visible to the workgroup
  __syncthreads();
  const unsigned long long* src = reinterpret_cast<const unsigned long long*>(st);
  unsigned long long* out = reinterpret_cast<unsigned long long*>(dst);
  constexpr int kWords = (int)(offsetof(DevState, seq) / 8);
  for (int i = tid; i < kWords; i += nthreads) out[i] = src[i];
  __threadfence_system();
  __syncthreads();
  if (tid == 0) __hip_atomic_store(&dst->seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// cooperative copy of the controller state between global memory and LDS (sizeof(LmState) is a multiple of 8)
__device__ __forceinline__ void copy_words(double* dst, const double* src, int n_doubles, int tid, int nthreads) {
  for (int i = tid; i < n_doubles; i += nthreads) dst[i] = src[i];
}
static_assert(sizeof(LmState) % 8 == 0 && sizeof(LmSums) % 8 == 0, "controller state must be double-aligned");

constexpr int kPartStride = SO_SOLVE_BLOCKS;  // partials[a][workgroup]: transposed so that the last workgroup reads it coalesced
static_assert(kEvalBlocks <= kPartStride && kFitBlocksMax <= kPartStride && kNAcc <= kSumsStride, "partials table");
constexpr int kRedStride = kNAcc + 2;  // 31 doubles per record in LDS: 62-dword rows, so 32 consecutive rows start in 32 different bank pairs

// ---- first level of every reduction: the kNAcc accumulators of the 256 threads of a workgroup -> one record.
// Round 5: in registers.  A reduce-scatter butterfly over the wavefront -- 32 values x 64 lanes -> lane L ends up with the
// wavefront's total of value L >> 1 -- whose first two stages are gfx950's v_permlane32_swap / v_permlane16_swap (swap the
// upper half / the odd rows of one register with the lower half / the even rows of another: after the swap ONE v_add_f64 has
// summed value a over the two halves in the lower lanes and value b in the upper lanes, no selects), the in-row stages DPP moves
// of the half each lane gives away (row_ror:8, row_half_mirror, quad_perm -- partner masks 8, 7, 2, 1 span the row), ~125 VALU
// instructions, no LDS, no barrier; then the four wavefronts' totals meet in part[4][32] (one barrier).  The round-4 form staged
// all 256 x 29 accumulators through 60 KB of LDS (1.3 - 1.8 us per pass by the in-kernel stamps) and is what kept a second
// workgroup off the compute unit.  Fixed tree: identical bits on every launch, in every instantiation (single / BATCH virtual
// workgroups / per-evaluation launches all come through here).
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xF, 0xF, false);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xF, 0xF, false);
  return __hiloint2double(hi, lo);
}
// (a, b) -> a summed over the lane pairs {L, L ^ 32} in lanes 0..31, b in lanes 32..63
__device__ __forceinline__ double swap32_add(double a, double b) {
  const auto lo = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(a), (unsigned)__double2loint(b), false, false);
  const auto hi = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(b), false, false);
  return __hiloint2double((int)hi[0], (int)lo[0]) + __hiloint2double((int)hi[1], (int)lo[1]);
}
// (a, b) -> a summed over {L, L ^ 16} in the even rows of 16 lanes, b in the odd rows
__device__ __forceinline__ double swap16_add(double a, double b) {
  const auto lo = __builtin_amdgcn_permlane16_swap((unsigned)__double2loint(a), (unsigned)__double2loint(b), false, false);
  const auto hi = __builtin_amdgcn_permlane16_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(b), false, false);
  return __hiloint2double((int)hi[0], (int)lo[0]) + __hiloint2double((int)hi[1], (int)lo[1]);
}
// in-row stage with partner L ^ mask (CTRL = the DPP pattern that reads that partner): lanes with `up` clear keep a, the others b
template <int CTRL>
__device__ __forceinline__ double row_stage_add(double a, double b, bool up) {
  const double keep = up ? b : a, give = up ? a : b;
  return keep + dpp_f64<CTRL>(give);
}
// lane L returns the wavefront's total of acc[L >> 1] (lanes with L >> 1 >= kNAcc: 0)
__device__ __forceinline__ double wave_reduce_scatter(const double (&acc)[kNAcc]) {
  static_assert(kNAcc > 16 && kNAcc <= 32, "butterfly over 32 value slots");
  const int lane = (int)(threadIdx.x & 63u);
  double u[16], w[8], x[4], y[2];
#pragma unroll
  for (int a = 0; a < 16; ++a) u[a] = swap32_add(acc[a], (a + 16 < kNAcc) ? acc[a + 16] : 0.0);   // lane bit 5 picks a / a + 16
#pragma unroll
  for (int a = 0; a < 8; ++a) w[a] = swap16_add(u[a], u[a + 8]);                                    // bit 4: a / a + 8
  const bool b3 = (lane & 8) != 0, b2 = (lane & 4) != 0, b1 = (lane & 2) != 0;
#pragma unroll
  for (int a = 0; a < 4; ++a) x[a] = row_stage_add<0x128>(w[a], w[a + 4], b3);                      // partner L ^ 8 (row_ror:8), bit 3: a / a + 4
#pragma unroll
  for (int a = 0; a < 2; ++a) y[a] = row_stage_add<0x141>(x[a], x[a + 2], b2);                      // partner L ^ 7 (row_half_mirror), bit 2: a / a + 2
  const double z = row_stage_add<0x4E>(y[0], y[1], b1);                                             // partner L ^ 2 (quad_perm [2,3,0,1]), bit 1
  return z + dpp_f64<0xB1>(z);                                                                      // partner L ^ 1 (quad_perm [1,0,3,2]): both lanes of the pair
}
// the workgroup's record: thread a < kNAcc returns the total of value a (the four wavefronts' totals added in wavefront order)
__device__ __forceinline__ double wg_reduce(const double (&acc)[kNAcc], double (*part)[32], int tid) {
  const double t = wave_reduce_scatter(acc);
  const int lane = tid & 63, wave = tid >> 6;
  if (!(lane & 1) && (lane >> 1) < kNAcc) part[wave][lane >> 1] = t;
  __syncthreads();
  double tot = 0;
  if (tid < kNAcc) tot = ((part[0][tid] + part[1][tid]) + part[2][tid]) + part[3][tid];
  return tot;
}

// second level (the per-evaluation launches): sum 256 records of kNAcc doubles held in red[256][kRedStride]: thread (a, c) adds rows 32c..32c+31 of value a in
// order, then thread (a, 0) adds the 8 partial sums in order -- a fixed tree, identical on every launch.
// Thread (a, c) belongs to wavefront c / 2 and so do the rows 32c..32c+31: when every thread has written ITS OWN row, a
// wavefront reads only what it wrote itself -- the caller needs a wavefront-scope fence between the two, not a barrier.
__device__ __forceinline__ double reduce_records(double (*red)[kRedStride], double (*part)[32], int tid) {
  const int a = tid & 31, cch = tid >> 5;
  if (a < kNAcc) {
    double s = 0;
#pragma unroll 8
    for (int r = 0; r < 32; ++r) s += red[cch * 32 + r][a];
    part[cch][a] = s;
  }
  __syncthreads();
  double tot = 0;
  if (tid < kNAcc) {
#pragma unroll
    for (int cc = 0; cc < 8; ++cc) tot += part[cc][tid];
  }
  return tot;
}

struct EvalShared {
  double red[256][kRedStride];  // 60 KB: per-thread accumulators, later the workgroup records
  double part[8][32];
  double part_odd[8][32];       // (BATCH: the odd virtual workgroups of a pass -- one barrier per virtual workgroup, see eval_pass)
  LmSums sums;
  LmState S;
  LmCtl ctl;
  double pose[7];
  int more;
  int32_t lh[16];
  int32_t hpart[8][16];
  int reg_done;  // set by the controller thread when the solve it just finished ends the registration
  bool is_last;
  unsigned long long peer_seq;  // peer exchange: running pass number of this context (loaded when the launch starts)
};
// the (at most) two accepted correspondences a thread of a persistent solve owns, kept in LDS between the passes
struct CorrCache {
  double nx[2][256], ny[2][256], nz[2][256], nw[2][256], c[2][256];
  float px[2][256], py[2][256], pz[2][256];
};
enum { kPassNotLast = 0, kPassMore = 1, kPassDone = 2, kPassSums = 3 };

// One evaluation pass of this workgroup at `pose`: accumulate, reduce, hand off.  Returns kPassNotLast in every
// workgroup but the one that arrives last; that one reduces the partial records and (fuse_lm) runs the LM controller:
// kPassMore = another evaluation is requested at sh.S.cand, kPassDone = the solve ended (state published),
// kPassSums = !fuse_lm, the sums are in `out` for the all-reduce.
// PROF  : profiling / test-hook instantiation (phase stamps, SOICP_ABLATE switches), launched only when SOICP_ABLATE is set.
// BATCH : so_icp_register_batch.  The real workgroup stands in for the VIRTUAL workgroups [span.vb0, span.vb1) of the grid of
//         span.V workgroups a single registration of the same scan launches: it walks their query sets one after the other,
//         reduces each through LDS exactly like a workgroup of that grid would and pushes one record per virtual workgroup,
//         so the controller's workgroup adds the same numbers in the same order -- the sums, and with them every decision
//         and the pose, are bit-identical to the single registration, whatever the number of real workgroups.
struct WgSpan { uint32_t vb0, vb1, V; bool ctl; };
// PEER  : the instantiation with the peer exchange compiled in (sharded registration, N > 1); the single-device launches carry
//         none of its index arithmetic and arguments
template <bool FIT, bool PERSIST = false, bool PROF = false, bool BATCH = false, bool PEER = false>
__device__ __forceinline__ int eval_pass(int slot, int fuse_lm, const Pose& pose, const float* __restrict__ spx,
                                         const float* __restrict__ spy, const float* __restrict__ spz,
                                         const CorrBuffers& corr, DevState* __restrict__ st, const EvalParams& ep,
                                         double* __restrict__ partials, uint32_t* __restrict__ ticket,
                                         int32_t* __restrict__ hist, LmSums* __restrict__ out,
                                         const float4* __restrict__ mpts, const uint32_t* __restrict__ nbr5,
                                         const MatchParams& mp, EvalShared& sh, unsigned long long pass_tag = 0,
                                         CorrCache* cc = nullptr, u4v* hand = nullptr, unsigned long long want = 0,
                                         const WgSpan span = WgSpan{0, 0, 0, false}) {
  // PERSIST (solve_kernel): workgroup 0 is the finisher of every pass of the launch, so the controller state stays in
  // its LDS from pass to pass and goes to memory only when the solve ends; the workgroup records are PUSHED (tagged
  // 16-byte chunks, see below) instead of stored + counted
  constexpr bool keep_state = PERSIST;
  double (*red)[kRedStride] = sh.red;
  double (*part)[32] = sh.part;
  LmSums& sh_sums = sh.sums;
  LmState& sh_S = sh.S;
  LmCtl& sh_ctl = sh.ctl;
  int& sh_more = sh.more;
  int32_t* lh = sh.lh;
  bool& is_last = sh.is_last;
  const int tid = threadIdx.x;
  static_assert(!BATCH || PERSIST, "batched hypotheses run in the persistent solve launch");
  const int abl = PROF ? ep.ablate : 0;
  const bool stamp = PROF && (abl & 128) != 0;
  const uint32_t V = BATCH ? span.V : gridDim.x;                    // workgroups of the (virtual) grid
  const uint32_t vb_begin = BATCH ? span.vb0 : blockIdx.x, vb_end = BATCH ? span.vb1 : blockIdx.x + 1u;
  const bool is_ctl = BATCH ? span.ctl : (blockIdx.x == 0);         // this workgroup collects the records and runs the controller
  unsigned long long t_begin = 0, t_loop = 0, t_red = 0, t_ticket = 0, t_loaded = 0, t_sums = 0, t_lm = 0;
  if (stamp) t_begin = wall_clock64();
  if (FIT) {
    if (tid < 16) lh[tid] = 0;
    __syncthreads();
  }
  const uint32_t n_kept = (abl & 64) ? 0u : ep.n_queries;  // every query of the scan, original order
  const uint32_t qs = ep.q_stride;
  double acc[kNAcc];
  // R(q) as Eigen::Quaternion::toRotationMatrix (lidarOptimization.cpp:70)
  const double qx = pose.q[0], qy = pose.q[1], qz = pose.q[2], qw = pose.q[3];
  const double tx = 2 * qx, ty = 2 * qy, tz = 2 * qz;
  const double twx = tx * qw, twy = ty * qw, twz = tz * qw, txx = tx * qx, txy = ty * qx, txz = tz * qx;
  const double tyy = ty * qy, tyz = tz * qy, tzz = tz * qz;
  const double R00 = 1 - (tyy + tzz), R01 = txy - twz, R02 = txz + twy;
  const double R10 = txy + twz, R11 = 1 - (txx + tzz), R12 = tyz - twx;
  const double R20 = txz - twy, R21 = tyz + twx, R22 = 1 - (txx + tyy);
  const ObsAxes axes = obs_axes(pose);  // (FIT) sensor axes of the observability analysis: constant over the pass
  // residual, robust weight, Jacobian row and the 29 sums of one accepted correspondence (w* = R p + t)
  // (s / a^2 as a product with the reciprocal formed once per pass: the IEEE division sequence is ~28 instructions per
  //  correspondence and evaluation; the quotient feeds the weights only -- the comparison s <= a^2 is on s itself)
  const double inv_a2 = 1.0 / ep.a2, rho_out = (ep.variant == 0) ? ep.a2 / 6.0 : ep.a2 / 3.0;
  auto tail = [&](double px, double py, double pz, double wx, double wy, double wz, const double4& nd, double c) {
    // (fused multiply-adds from here on: no thresholds downstream, see SO_FMA)
    const double r = SO_FMA(nd.x, wx, SO_FMA(nd.y, wy, SO_FMA(nd.z, wz, nd.w)));  // lidarOptimization.cpp:61
    // ScaledLoss(TukeyLoss(a), c): rho, rho' [UPSTREAM ceres loss_function.cc]; corrector with rho''<=0
    const double s = r * r;
    double rho0, rho1;
    if (s <= ep.a2) {
      const double v = 1.0 - s * inv_a2, v2 = v * v;
      rho0 = rho_out * (1.0 - v2 * v); rho1 = (ep.variant == 0) ? 0.5 * v2 : v2;
    } else {
      rho0 = rho_out; rho1 = 0;
    }
    const double w = c * rho1;
    // J = [n^T, -n^T R [p]x] = [n^T, (p x R^T n)^T]  (lidarOptimization.cpp:68-74)
    const double m0 = SO_FMA(R00, nd.x, SO_FMA(R10, nd.y, R20 * nd.z));
    const double m1 = SO_FMA(R01, nd.x, SO_FMA(R11, nd.y, R21 * nd.z));
    const double m2 = SO_FMA(R02, nd.x, SO_FMA(R12, nd.y, R22 * nd.z));
    const double J[6] = {nd.x, nd.y, nd.z, SO_FMA(py, m2, -(pz * m1)), SO_FMA(pz, m0, -(px * m2)), SO_FMA(px, m1, -(py * m0))};
    acc[0] = SO_FMA(0.5 * c, rho0, acc[0]);
    acc[1] += 1.0;
    const double wr = w * r;
    int k = 8;
#pragma unroll
    for (int a = 0; a < 6; ++a) {
      acc[2 + a] = SO_FMA(J[a], wr, acc[2 + a]);
      const double wj = w * J[a];
#pragma unroll
      for (int b = a; b < 6; ++b) { acc[k] = SO_FMA(wj, J[b], acc[k]); ++k; }
    }
  };
  // one query: (FIT) plane fit from the prefetched neighbour coordinates, then residual / Jacobian / sums.
  // trip = 0 / 1: the query is one of the two this thread owns in every pass of a persistent solve -- an accepted
  // correspondence is then also left in the LDS cache, where the later passes of the launch find it.
  auto body = [&](const uint32_t j, int status, const float* nb, int trip) {
    if (!FIT && status != SO_MATCH_SUCCESS) return;
    if (FIT && status == SO_MATCH_DROPPED) return;
    const float fx = spx[(size_t)j * qs], fy = spy[(size_t)j * qs], fz = spz[(size_t)j * qs];
    const double px = (double)fx, py = (double)fy, pz = (double)fz;
    double wx, wy, wz;
    quat_rotate<double>(pose.q, px, py, pz, wx, wy, wz);                   // lidarOptimization.cpp:59 == LidarSlam.cpp:397-398
    wx += pose.t[0]; wy += pose.t[1]; wz += pose.t[2];
    double c;
    double4 nd;
    if (FIT) {
      if (PERSIST && !BATCH && trip >= 0) {  // (the coordinates go to the cache before the fit: three registers less to carry through it)
        cc->px[trip][tid] = fx; cc->py[trip][tid] = fy; cc->pz[trip][tid] = fz;
      }
      double fnd[4] = {0, 0, 0, 0}, fc = 0;
      int obs[3] = {0, 0, 0};
      if (status == SO_MATCH_PENDING) {
        const double pw[3] = {wx, wy, wz};
        // (test switches of the PROF instantiation: 4096 = the reference's column-pivoted Householder factorisation instead of the
        //  closed form of plane_fit.h, 512 = that plus the cyclic Jacobi eigen-solver)
        if (PROF && (mp.ablate & (512 | 4096))) status = plane_from_neighbours(nb, pw, pose, mp, fnd, fc, obs, (mp.ablate & 512) != 0);
        else status = plane_fit5(nb, pw, axes, mp.sq_max_dist_f, mp.max_point_dist, fnd, fc, obs);
      }
      if (status != SO_MATCH_SUCCESS) { fc = 0; fnd[0] = fnd[1] = fnd[2] = fnd[3] = 0; }
      nd = make_double4(fnd[0], fnd[1], fnd[2], fnd[3]);
      c = fc;
      // streaming stores: the record is re-read at most by a later launch, and dirty L2 lines would have to be written
      // back when this kernel ends (the next launch waits for that)
      __builtin_nontemporal_store(nd.x, &corr.nd[j].x); __builtin_nontemporal_store(nd.y, &corr.nd[j].y);
      __builtin_nontemporal_store(nd.z, &corr.nd[j].z); __builtin_nontemporal_store(nd.w, &corr.nd[j].w);
      __builtin_nontemporal_store(c, &corr.coeff[j]); __builtin_nontemporal_store((uint8_t)status, &corr.status[j]);
      atomicAdd(&lh[status], 1);                                          // MatchRejectionHistogramPlane, LidarSlam.cpp:341
      if (status == SO_MATCH_SUCCESS) { atomicAdd(&lh[7 + obs[0]], 1); atomicAdd(&lh[7 + obs[1]], 1); atomicAdd(&lh[7 + obs[2]], 1); }
      if (status != SO_MATCH_SUCCESS) return;
      if (PERSIST && !BATCH && trip >= 0) {
        cc->nx[trip][tid] = nd.x; cc->ny[trip][tid] = nd.y; cc->nz[trip][tid] = nd.z; cc->nw[trip][tid] = nd.w;
        cc->c[trip][tid] = c;  // >= 0 marks the entry as an accepted correspondence
      }
    } else {
      c = corr.coeff[j];
      nd = corr.nd[j];
    }
    tail(px, py, pz, wx, wy, wz, nd, c);
  };
  double mine = 0;
  u4v* rec = reinterpret_cast<u4v*>(partials);        // PERSIST: the record table of the pushed workgroup records
  const unsigned int tag = (unsigned int)pass_tag;
  // BATCH, plain evaluation: the two correspondence records a thread owns in a virtual workgroup are fetched one virtual
  // workgroup AHEAD, status and record together (a rejected query's record is valid memory, just not used): the loads
  // are in flight during the sums and the reduction of the workgroup before, where a pass used to sit through two
  // dependent round trips per virtual workgroup with two wavefronts per SIMD to hide them (measured: 4 us each).
  struct Fetched { int st; double4 nd; double c; float x, y, z; };
  Fetched cur[2], nxt[2];
  auto fetch2 = [&](uint32_t vb_, Fetched (&o)[2]) {
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const uint32_t j = query_of(vb_, V, tid, (uint32_t)q);
      const uint32_t js = j < n_kept ? j : 0u;
      o[q].st = j < n_kept ? (int)corr.status[js] : (int)SO_MATCH_DROPPED;
      o[q].nd = corr.nd[js]; o[q].c = corr.coeff[js];
      o[q].x = spx[(size_t)js * qs]; o[q].y = spy[(size_t)js * qs]; o[q].z = spz[(size_t)js * qs];
    }
  };
  if (BATCH && !FIT && vb_begin < vb_end) fetch2(vb_begin, cur);
  for (uint32_t vb = vb_begin, trip_ = 0; BATCH ? (vb < vb_end) : (trip_ < 1u); ++vb, ++trip_) {  // (exactly one trip unless BATCH)
#pragma unroll
  for (int a = 0; a < kNAcc; ++a) acc[a] = 0;
  if (FIT) {
    if (PERSIST && !BATCH) { cc->c[0][tid] = -1.0; cc->c[1][tid] = -1.0; }  // (each thread touches only its own column: no barrier)
    // Two queries per trip with their gathers issued together: status + neighbour indices of both (one round trip),
    // then the ten neighbour points (one round trip), then the two fits.  With one wavefront per SIMD nothing else
    // hides the latency of the dependent index -> point loads (measured: 3 us of a 12 us pass).
    bool first = true;
    if (BATCH) {
      // one query per trip (same order of accumulation): half the live registers, so that two workgroups fit a compute unit
      // and their wavefronts hide each other's gather and fp64 latencies -- a batch has the parallelism the single
      // registration lacks
      for (uint32_t q = 0, j; (j = query_of(vb, V, tid, q)) < n_kept; ++q) {
        const int stq = corr.status[j];
        uint32_t iq[5];
#pragma unroll
        for (int t = 0; t < 5; ++t) iq[t] = nbr5[(size_t)5 * j + t];
        float nbq[15];
#pragma unroll
        for (int t = 0; t < 5; ++t) {
          const float4 a = mpts[stq == SO_MATCH_PENDING ? iq[t] : 0u];
          nbq[3 * t] = a.x; nbq[3 * t + 1] = a.y; nbq[3 * t + 2] = a.z;
        }
        body(j, stq, nbq, -1);
      }
    } else
    for (uint32_t q = 0, jA; (jA = query_of(vb, V, tid, q)) < n_kept; q += 2) {
      const uint32_t jB = query_of(vb, V, tid, q + 1u);
      const bool hasB = jB < n_kept;
      const uint32_t jBs = hasB ? jB : jA;
      const int stA = corr.status[jA];
      const int stB = hasB ? (int)corr.status[jBs] : SO_MATCH_NOT_ENOUGH;
      uint32_t iA[5], iB[5];
#pragma unroll
      for (int t = 0; t < 5; ++t) { iA[t] = nbr5[(size_t)5 * jA + t]; iB[t] = nbr5[(size_t)5 * jBs + t]; }
      float nbA[15], nbB[15];
#pragma unroll
      for (int t = 0; t < 5; ++t) {  // indices of a query that is not PENDING are stale: read point 0 instead
        const float4 a = mpts[stA == SO_MATCH_PENDING ? iA[t] : 0u], b = mpts[stB == SO_MATCH_PENDING ? iB[t] : 0u];
        nbA[3 * t] = a.x; nbA[3 * t + 1] = a.y; nbA[3 * t + 2] = a.z;
        nbB[3 * t] = b.x; nbB[3 * t + 1] = b.y; nbB[3 * t + 2] = b.z;
      }
      body(jA, stA, nbA, first ? 0 : -1);
      if (hasB) body(jB, stB, nbB, first ? 1 : -1);
      first = false;
    }
  } else if (PERSIST && !BATCH) {
    // the two queries this thread fitted first: out of the LDS cache (no memory round trip on the pass's critical path)
#pragma unroll
    for (int trip = 0; trip < 2; ++trip) {
      const double c = cc->c[trip][tid];
      const double4 nd = make_double4(cc->nx[trip][tid], cc->ny[trip][tid], cc->nz[trip][tid], cc->nw[trip][tid]);
      const double px = (double)cc->px[trip][tid], py = (double)cc->py[trip][tid], pz = (double)cc->pz[trip][tid];
      if (c >= 0.0) {
        double wx, wy, wz;
        quat_rotate<double>(pose.q, px, py, pz, wx, wy, wz);
        wx += pose.t[0]; wy += pose.t[1]; wz += pose.t[2];
        tail(px, py, pz, wx, wy, wz, nd, c);
      }
    }
    for (uint32_t q = 2, j; (j = query_of(vb, V, tid, q)) < n_kept; ++q) body(j, corr.status[j], nullptr, -1);
  } else if (BATCH) {
    if (vb + 1u < vb_end) fetch2(vb + 1u, nxt);
#pragma unroll
    for (int q = 0; q < 2; ++q)
      if (cur[q].st == SO_MATCH_SUCCESS) {  // (the arithmetic of body(), on the record fetched ahead)
        const double px = (double)cur[q].x, py = (double)cur[q].y, pz = (double)cur[q].z;
        double wx, wy, wz;
        quat_rotate<double>(pose.q, px, py, pz, wx, wy, wz);
        wx += pose.t[0]; wy += pose.t[1]; wz += pose.t[2];
        tail(px, py, pz, wx, wy, wz, cur[q].nd, cur[q].c);
      }
    for (uint32_t q = 2, j; (j = query_of(vb, V, tid, q)) < n_kept; ++q) body(j, corr.status[j], nullptr, -1);
    cur[0] = nxt[0]; cur[1] = nxt[1];
  } else {
    for (uint32_t q = 0, j; (j = query_of(vb, V, tid, q)) < n_kept; ++q) body(j, corr.status[j], nullptr, -1);
  }
  if (stamp) t_loop = wall_clock64();
  // workgroup reduction in registers (wave_reduce_scatter), fixed order
  // BATCH: the wavefront totals alternate between two buffers, so that ONE barrier per virtual workgroup (the one inside
  // wg_reduce) orders everything: a wavefront that writes buffer b again has passed the barrier of the virtual
  // workgroup in between, which the threads that read b reach only after their reads
  mine = wg_reduce(acc, (BATCH && (trip_ & 1u)) ? sh.part_odd : part, tid);
  if (PERSIST) {
    // Push model: the record of this workgroup for this pass is kNAcc chunks of 16 bytes {value (8), pass tag (4), one
    // histogram counter of the fit pass (4)}, each written with ONE sc1 dwordx4 store -- fire and forget: no drain, no
    // arrival counter.  Workgroup 0 polls the table until every chunk carries this pass's tag: one memory round trip
    // from "last record written" to "all sums in hand" instead of two (arrival counter, then the records).
    // Table layout [i][g][a] for workgroup w = 32 g + i: a writer's 29 chunks are contiguous, and polling thread
    // t = 29 g + a reads chunk i at table[232 i + t] (coalesced over the workgroup); it owns value a of the 32 workgroups
    // of group g and adds them IN REGISTERS, in the order of reduce_records (rows 32 g .. 32 g + 31, then the 8 groups).
    // A stale chunk is always the previous pass's (every pass rewrites every chunk), so a 32-bit tag suffices.
    static_assert(SO_SOLVE_BLOCKS == 256, "record table: 8 groups of 32 workgroups");
    static_assert(kNAcc <= kRecordChunksMax, "record table");
    // The correspondence stores of the fit loop are long on their way (the LDS reduction came in between); retiring
    // them HERE, in the compiler's scoreboard too, keeps it from placing its own vmcnt waits between the hand-written
    // polling loads below (which it cannot see), where each would cost a full memory round trip.
    if (FIT) __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0) expcnt(7) lgkmcnt(15)
    if (tid < kNAcc) {
      const unsigned long long b = (unsigned long long)__double_as_longlong(mine);
      // MatchRejectionHistogramPlane + observability bins (BATCH: lh counts every virtual workgroup of this real one: they
      // travel with the record of the last -- integer sums, any split gives the same totals)
      const unsigned int extra = (FIT && tid < 16 && (!BATCH || vb + 1u == vb_end)) ? (unsigned int)lh[tid] : 0u;
      const u4v v = {(unsigned int)b, (unsigned int)(b >> 32), tag, extra};
      const uint32_t w = vb;
      store16_sc1(rec + ((w & 31u) * 8u + (w >> 5)) * kNAcc + tid, v);
    }
  }
  }  // virtual workgroups
  if (PERSIST) {
    if (stamp) t_red = wall_clock64();
    if (!is_ctl) return kPassNotLast;
    constexpr int kPoll = 8 * kNAcc;  // 232 polling threads
    const bool poller = tid < kPoll;
    const int g = poller ? tid / kNAcc : 0, a = poller ? tid - g * kNAcc : 0;
    const u4v* base = rec + (poller ? tid : 0);
    if (!BATCH) {
    u4v r[32];
    const unsigned long long t0 = wall_clock64();
    bool ok;
    for (;;) {
#pragma unroll
      for (int i = 0; i < 32; ++i)
        asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(r[i]) : "v"(base + (size_t)i * kPoll) : "memory");
      asm volatile("s_waitcnt vmcnt(0)"
                   : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]), "+v"(r[8]), "+v"(r[9]),
                     "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15]), "+v"(r[16]), "+v"(r[17]), "+v"(r[18]),
                     "+v"(r[19]), "+v"(r[20]), "+v"(r[21]), "+v"(r[22]), "+v"(r[23]), "+v"(r[24]), "+v"(r[25]), "+v"(r[26]), "+v"(r[27]),
                     "+v"(r[28])
                   :: "memory");
      asm volatile("" : "+v"(r[29]), "+v"(r[30]), "+v"(r[31]) :: "memory");
      ok = true;
#pragma unroll
      for (int i = 0; i < 32; ++i) ok = ok && (r[i].z == tag || (uint32_t)(32 * g + i) >= V);
      ok = ok || !poller;
      if (abl & 8192) { ok = false; break; }  // test hook: behave as if the records never arrived (the launch is abandoned)
      if (__ballot(!ok) == 0ull) break;
      if (wall_clock64() - t0 > ep.timeout_ticks) break;  // give up instead of hanging the device (EvalParams::timeout_ticks)
    }
    if (!__syncthreads_and(ok ? 1 : 0)) return kPassNotLast;  // timeout: the solve is abandoned, the host reports the missing publication
    if (stamp) t_ticket = t_loaded = wall_clock64();
    {
      double sum = 0;
      int hsum = 0;
#pragma unroll
      for (int i = 0; i < 32; ++i) {
        const bool have = (uint32_t)(32 * g + i) < V;
        sum += have ? __longlong_as_double((long long)(((unsigned long long)r[i].y << 32) | r[i].x)) : 0.0;
        if (FIT) hsum += have ? (int)r[i].w : 0;
      }
      if (poller) {
        part[g][a] = sum;
        if (FIT && a < 16) sh.hpart[g][a] = hsum;
      }
    }
    } else {
      // BATCH: the same sums in the same order, eight records at a time (a quarter of the registers -- two workgroups fit a
      // compute unit -- and a pass lasts tens of microseconds here: the four round trips do not matter, polling gently does)
      double sum = 0;
      int hsum = 0;
      bool ok = true;
      const unsigned long long t0 = wall_clock64();
      for (int i0 = 0; i0 < 32; i0 += 8) {
        u4v r[8];
        for (;;) {
#pragma unroll
          for (int k = 0; k < 8; ++k)
            asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(r[k]) : "v"(base + (size_t)(i0 + k) * kPoll) : "memory");
          asm volatile("s_waitcnt vmcnt(0)" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]) :: "memory");
          bool good = true;
#pragma unroll
          for (int k = 0; k < 8; ++k) good = good && (r[k].z == tag || (uint32_t)(32 * g + i0 + k) >= V);
          good = good || !poller;
          if (abl & 8192) good = false;  // test hook: behave as if the records never arrived
          if (__ballot(!good) == 0ull) break;
          if ((abl & 8192) || wall_clock64() - t0 > ep.timeout_ticks) { ok = ok && good; break; }  // give up (EvalParams::timeout_ticks)
          __builtin_amdgcn_s_sleep(32);
        }
        if (__ballot(!ok) != 0ull) break;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const bool have = (uint32_t)(32 * g + i0 + k) < V;
          sum += have ? __longlong_as_double((long long)(((unsigned long long)r[k].y << 32) | r[k].x)) : 0.0;
          if (FIT) hsum += have ? (int)r[k].w : 0;
        }
      }
      if (!__syncthreads_and(ok ? 1 : 0)) return kPassNotLast;  // timeout: the solve of this hypothesis is abandoned, the host reports it
      if (poller) {
        part[g][a] = sum;
        if (FIT && a < 16) sh.hpart[g][a] = hsum;
      }
    }
    __syncthreads();
    double* o = reinterpret_cast<double*>(&sh_sums);
    if (tid < kNAcc) {
      double tot = 0;
#pragma unroll
      for (int cc8 = 0; cc8 < 8; ++cc8) tot += part[cc8][tid];
      o[tid] = tot;  // cost, count, Jtr[6], JtJ[21]
    } else if (FIT && tid >= 32 && tid < 48) {  // the 16 counters; later passes of the solve keep them in LDS
      int h = 0;
#pragma unroll
      for (int cc8 = 0; cc8 < 8; ++cc8) h += sh.hpart[cc8][tid - 32];
      o[kNAcc + (tid - 32)] = (double)h;
    }
    __syncthreads();
    if (PEER && ep.peer_world > 1) {
      // ---- peer exchange: this rank's record to every rank's inbox, then everybody's records out of the own inbox.
      //      Double-buffered by the parity of the pass number: a rank can run at most one pass ahead of another (it
      //      needs that rank's record of the pass before), so a chunk is never overwritten before it was read.
      const int nx = FIT ? (kNAcc + 16) : kNAcc;  // (the histogram counters travel with the fit pass only)
      const unsigned int xtag = (unsigned int)(sh.peer_seq + 1ull);
      const int par = (int)(xtag & 1u);
      if (tid < nx) {
        const unsigned long long b = (unsigned long long)__double_as_longlong(o[tid]);
        const u4v v = {(unsigned int)b, (unsigned int)(b >> 32), xtag, (unsigned int)ep.peer_rank};
        for (int p = 0; p < ep.peer_world; ++p)
          store16_sys(reinterpret_cast<u4v*>(ep.peer_inbox[p]) + (size_t)(par * kPeerMaxWorld + ep.peer_rank) * kPeerChunks + tid, v);
      }
      const u4v* own = reinterpret_cast<const u4v*>(ep.peer_inbox[ep.peer_rank]) + (size_t)par * kPeerMaxWorld * kPeerChunks;
      const int total = nx * ep.peer_world;  // <= 360 chunks: two per thread at most
      const int c0 = tid, c1 = tid + 256;
      const bool h0 = c0 < total, h1 = c1 < total;
      const int s0 = h0 ? c0 / nx : 0, a0 = h0 ? c0 - s0 * nx : 0, s1 = h1 ? c1 / nx : 0, a1 = h1 ? c1 - s1 * nx : 0;
      bool d0 = !h0, d1 = !h1;
      unsigned long long v0 = 0, v1 = 0;
      const unsigned long long tp = wall_clock64();
      bool okx = true;
      for (;;) {
        if (!d0) { const u4v v = load16_sys(own + (size_t)s0 * kPeerChunks + a0); if (v.z == xtag) { d0 = true; v0 = ((unsigned long long)v.y << 32) | v.x; } }
        if (!d1) { const u4v v = load16_sys(own + (size_t)s1 * kPeerChunks + a1); if (v.z == xtag) { d1 = true; v1 = ((unsigned long long)v.y << 32) | v.x; } }
        if (__syncthreads_and((d0 && d1) ? 1 : 0)) break;
        if (__syncthreads_or((wall_clock64() - tp > ep.timeout_ticks) ? 1 : 0)) { okx = false; break; }  // a rank is missing -- give up (uniformly), the host reports it
      }
      if (!__syncthreads_and(okx ? 1 : 0)) return kPassNotLast;
      double (*xs)[48] = reinterpret_cast<double (*)[48]>(&red[0][0]);  // (the reduction buffer is free here)
      if (h0) xs[s0][a0] = __longlong_as_double((long long)v0);
      if (h1) xs[s1][a1] = __longlong_as_double((long long)v1);
      __syncthreads();
      if (tid < nx) {
        double tot = 0;
        for (int r = 0; r < ep.peer_world; ++r) tot += xs[r][tid];  // rank order: the same bits on every rank
        o[tid] = tot;
      }
      if (tid == 0) sh.peer_seq = (unsigned long long)xtag;
      __syncthreads();
    }
    if (stamp) t_sums = wall_clock64();
    // (BATCH too, round 6: the one-thread controller on the LDS copy of its state was what spilled -- 316 bytes of scratch per lane in a launch
    //  capped at 256 registers --, in the serial code every virtual workgroup's pass waits for)
    if (tid < 64) {  // the controller's wavefront (publishes the hand-off)
#ifdef SO_LM_STAMPS
      const int more_ = lm_control_wave(slot, st, sh_S, sh_sums, sh_ctl, &sh.part_odd[0][0], hand, want, sh.pose, &sh.reg_done, tid, (slot == 1 && tid == 0) ? st->dbg : nullptr);
#else
      const int more_ = lm_control_wave(slot, st, sh_S, sh_sums, sh_ctl, &sh.part_odd[0][0], hand, want, sh.pose, &sh.reg_done, tid);
#endif
      if (tid == 0) sh_more = more_;
    }
    __syncthreads();
    unsigned long long t_ctl = 0;
    if (stamp) t_ctl = wall_clock64();
    if (!sh_more) {  // the solve is over: histogram replicas cleared for the next outer iteration, state to memory, publish
      hist[tid] = 0; hist[256 + tid] = 0;
      if (PEER && ep.peer_world > 1 && tid == 0) st->peer_seq = sh.peer_seq;
      if (tid < (int)(sizeof(LmState) / 8))
        __hip_atomic_store(reinterpret_cast<double*>(&st->S) + tid, reinterpret_cast<const double*>(&sh_S)[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      // the guess of a chained registration behind this one (so_icp_register_sequence): this result o the delta the launch carries
      if (!BATCH && ep.chain_next && sh.reg_done && tid == 0) {
        double tf[7], nx[7];
        for (int i = 0; i < 7; ++i) tf[i] = sh_S.x[i];  // (= DevState::T_final)
        pose_compose(tf, ep.chain_delta, nx);
        for (int i = 0; i < 7; ++i) st->T_chain[i] = nx[i];
      }
      // (a solve that does not end the registration may leave the report to the next k-NN launch, see EvalParams)
      if (!ep.defer_publish || sh.reg_done) publish_state(st, ep, sh_ctl.outer_iter, tid, 256);
    }
    if (stamp && tid == 0 && (FIT || slot == 1)) {
      t_lm = wall_clock64();
      unsigned long long* d = st->dbg + (FIT ? 0 : 8);
      d[0] = t_loop - t_begin; d[1] = t_red - t_loop; d[2] = t_ticket - t_red; d[3] = t_loaded - t_ticket; d[4] = t_sums - t_loaded;
      d[5] = t_lm - t_sums; d[6] = t_lm - t_begin; d[7] = t_ctl - t_sums;
    }
    return sh_more ? kPassMore : kPassDone;
  }
  // Hand-off to the last workgroup WITHOUT release/acquire fences (each costs microseconds on gfx950): the 29
  // partial sums are 8-byte agent-scope relaxed atomic stores (write-through, sc1), drained with vmcnt(0) by the
  // storing wave before the arrival ticket; the consumer reads them with agent-scope relaxed atomic loads
  // (sc1: served by L2, never a stale L1 line).  [MI355X guide, G16 "8-B agent atomics both sides"]
  if (tid < kNAcc)
    __hip_atomic_store(&partials[(size_t)tid * kPartStride + blockIdx.x], mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (FIT && tid >= 32 && tid < 48 && lh[tid - 32])  // histograms: device-scope atomics on 16 replicas, visible before the ticket
    __hip_atomic_fetch_add(&hist[(blockIdx.x % kHistReplicas) * kHistStride + (tid - 32)], lh[tid - 32], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (stamp) t_red = wall_clock64();
  // Arrival: one fire-and-forget add per workgroup, spread over 16 counters on separate lines (256 adds on ONE word
  // serialise for ~2 us); workgroup 0 is the designated finisher and polls the 16 counters with one 16-lane load.
  if (tid == 0) __hip_atomic_fetch_add(&ticket[(blockIdx.x & (kArriveCounters - 1)) * kArriveStrideWords], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (blockIdx.x != 0) return kPassNotLast;
  if (tid < 64) {
    const uint32_t lanes = kArriveCounters;
    const uint32_t expect = (tid < (int)lanes) ? (gridDim.x + lanes - 1u - (uint32_t)tid) / lanes : 0u;  // workgroups b with b % 16 == tid
    const unsigned long long t0 = wall_clock64();
    bool ok = false;
    for (;;) {
      uint32_t v = 0;
      if (tid < (int)lanes) v = __hip_atomic_load(&ticket[tid * kArriveStrideWords], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (__ballot(v != expect) == 0ull) { ok = true; break; }
      if (wall_clock64() - t0 > ep.timeout_ticks) break;  // give up instead of hanging the device (EvalParams::timeout_ticks)
    }
    if (tid < (int)lanes) __hip_atomic_store(&ticket[tid * kArriveStrideWords], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // re-arm
    if (tid == 0) is_last = ok;
  }
  __syncthreads();
  if (!is_last) return kPassNotLast;  // timeout: the solve is abandoned, the host reports the missing publication
  if (stamp) t_ticket = wall_clock64();
  // thread b fetches workgroup b's record from the transposed table partials[a][b] (coalesced: 4 lines per wave
  // load).  The compiler serialises agent-scope atomic loads with a vmcnt(0) after each (6 us measured), so the 29
  // sc1 loads are issued back to back, the controller state is fetched behind them, and ONE wait covers all.
  {
    constexpr int kRec = (SO_SOLVE_BLOCKS + 255) / 256;  // records per thread
    double r[kRec][kNAcc];
    bool have[kRec];
#pragma unroll
    for (int q = 0; q < kRec; ++q) {
      have[q] = (uint32_t)(tid + 256 * q) < gridDim.x;  // no divergence around the asm: a register copy before the wait would read garbage
      const double* rec = partials + (have[q] ? tid + 256 * q : 0);
#pragma unroll
      for (int a = 0; a < kNAcc; ++a)
        asm volatile("global_load_dwordx2 %0, %1, off sc1" : "=v"(r[q][a]) : "v"(rec + a * kPartStride) : "memory");
    }
    if (fuse_lm) {  // (coherent loads: cheap insurance, one word per thread)
      if (!keep_state && tid < (int)(sizeof(LmState) / 8))
        reinterpret_cast<double*>(&sh_S)[tid] = __hip_atomic_load(reinterpret_cast<const double*>(&st->S) + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      load_ctl(sh_ctl, st, tid, 128);
    }
#pragma unroll
    for (int q = 0; q < kRec; ++q)
      asm volatile("s_waitcnt vmcnt(0)"
                   : "+v"(r[q][0]), "+v"(r[q][1]), "+v"(r[q][2]), "+v"(r[q][3]), "+v"(r[q][4]), "+v"(r[q][5]), "+v"(r[q][6]), "+v"(r[q][7]),
                     "+v"(r[q][8]), "+v"(r[q][9]), "+v"(r[q][10]), "+v"(r[q][11]), "+v"(r[q][12]), "+v"(r[q][13]), "+v"(r[q][14]),
                     "+v"(r[q][15]), "+v"(r[q][16]), "+v"(r[q][17]), "+v"(r[q][18]), "+v"(r[q][19]), "+v"(r[q][20]), "+v"(r[q][21]),
                     "+v"(r[q][22]), "+v"(r[q][23]), "+v"(r[q][24]), "+v"(r[q][25]), "+v"(r[q][26]), "+v"(r[q][27]), "+v"(r[q][28])
                   :: "memory");
#pragma unroll
    for (int a = 0; a < kNAcc; ++a) {
      double v = have[0] ? r[0][a] : 0.0;
#pragma unroll
      for (int q = 1; q < kRec; ++q) v += have[q] ? r[q][a] : 0.0;  // fixed order: record tid, tid + 256, ...
      red[tid][a] = v;
    }
  }
  __syncthreads();
  if (stamp) t_loaded = wall_clock64();
  const double total = reduce_records(red, part, tid);
  double* o = reinterpret_cast<double*>(&sh_sums);
  if (tid < kNAcc) {
    o[tid] = total;  // cost, count, Jtr[6], JtJ[21]
  } else if (tid >= 32 && tid < 48) {
    int h = 0;
#pragma unroll
    for (int r = 0; r < kHistReplicas; ++r)
      h += __hip_atomic_load(&hist[r * kHistStride + (tid - 32)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    o[kNAcc + (tid - 32)] = (double)h;
  }
  __syncthreads();
  copy_words(reinterpret_cast<double*>(out), o, (int)(sizeof(LmSums) / 8), tid, 256);
  if (stamp) t_sums = wall_clock64();
  if (!fuse_lm) return kPassSums;  // sharded map: the sums are all-reduced first, lm_step_kernel runs the controller
  if (tid == 0) sh_more = (abl & 32) ? 1 : lm_control(slot, st, sh_S, sh_sums, sh_ctl);  // sh_S / sh_ctl arrived with the partials
  __syncthreads();
  unsigned long long t_ctl = 0;
  if (stamp) t_ctl = wall_clock64();
  // the solve is over: clear the histogram replicas for the next outer iteration (ResetDistanceParameters, LidarSlam.cpp:847-852)
  if (!sh_more) { hist[tid] = 0; hist[256 + tid] = 0; }
  if ((!keep_state || !sh_more) && tid < (int)(sizeof(LmState) / 8))
    __hip_atomic_store(reinterpret_cast<double*>(&st->S) + tid, reinterpret_cast<const double*>(&sh_S)[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (!sh_more) publish_state(st, ep, sh_ctl.outer_iter, tid, 256);
  if (stamp && tid == 0 && (FIT || slot == 1)) {  // (the plain-evaluation record is the one of slot 1: never the pass that publishes)
    t_lm = wall_clock64();
    unsigned long long* d = st->dbg + (FIT ? 0 : 8);
    d[0] = t_loop - t_begin; d[1] = t_red - t_loop; d[2] = t_ticket - t_red; d[3] = t_loaded - t_ticket; d[4] = t_sums - t_loaded;
    d[5] = t_lm - t_sums; d[6] = t_lm - t_begin; d[7] = t_ctl - t_sums;
  }
  return sh_more ? kPassMore : kPassDone;
}

static_assert(sizeof(LmState) / 8 <= 256, "controller state is moved one word per thread");
// FIT = true : the slot-0 launch of an outer iteration.  Dense over the queries (full lane occupancy for the heavy
//              fp64 work): plane fit from the five neighbour indices left by knn_plane_kernel (PCA gate, 5x3 LS
//              plane, inlier gate, coefficient, observability labels -> correspondence record + histograms), then
//              the first evaluation at the outer pose.
// FIT = false: evaluations at the poses requested by the LM controller.
template <bool FIT, bool PROF>
__global__ __launch_bounds__(256, SO_SOLVE_BLOCKS / 256) void eval_kernel(int slot, int fuse_lm, const float* __restrict__ spx,
                                                   const float* __restrict__ spy, const float* __restrict__ spz,
                                                   CorrBuffers corr, DevState* __restrict__ st, EvalParams ep,
                                                   double* __restrict__ partials, uint32_t* __restrict__ ticket,
                                                   int32_t* __restrict__ hist, LmSums* __restrict__ out,
                                                   const float4* __restrict__ mpts, const uint32_t* __restrict__ nbr5,
                                                   MatchParams mp) {
  __shared__ EvalShared sh;
  if (!eval_slot_active(st, slot)) return;
  const Pose pose = pose_from_array(slot == 0 ? st->T : st->eval_pose);
  (void)eval_pass<FIT, false, PROF>(slot, fuse_lm, pose, spx, spy, spz, corr, st, ep, partials, ticket, hist, out, mpts, nbr5, mp, sh);
}

// The whole solve of one outer iteration in ONE launch (single device): slot 0 = plane fit + first evaluation, then up
// to lm_max evaluations at the poses the controller requests.  Between two evaluations the workgroups wait for the
// controller's workgroup to publish {next pose, more?} and then an epoch word, all with agent-scope (sc1) stores / loads
// -- the XCDs' L2s are not coherent with each other, so nothing that crosses workgroups inside this launch goes
// through plain loads.  Requires every workgroup to be resident (<= 256 workgroups of 256 threads, 60 KB LDS each: two
// fit on a CU); a wait that exceeds 50 ms gives up (the host then reports the missing publication).
// BATCH (so_icp_register_batch): bv.wg_per_hyp consecutive workgroups serve hypothesis active[blockIdx.x / wg_per_hyp] --
// its own state block, correspondence records, record table and hand-off record; the first of them is its controller.
// Each stands in for an equal share of the bv.v_grid workgroups of the single-registration launch (eval_pass, WgSpan), so
// the hypotheses advance independently of each other inside the one launch and every one of them reproduces the single
// registration bit for bit.  The grid never exceeds the compute units (every workgroup resident).
template <bool PROF, bool BATCH, bool PEER = false>
__global__ __launch_bounds__(256, BATCH ? 2 : 1) void solve_kernel(int lm_max, const float* __restrict__ spx, const float* __restrict__ spy,
                                                    const float* __restrict__ spz, CorrBuffers corr, DevState* __restrict__ st,
                                                    EvalParams ep, double* __restrict__ partials, uint32_t* __restrict__ ticket,
                                                    int32_t* __restrict__ hist, LmSums* __restrict__ out,
                                                    const float4* __restrict__ mpts, const uint32_t* __restrict__ nbr5,
                                                    MatchParams mp, BatchView bv) {
  __shared__ EvalShared sh;
  __shared__ CorrCache cache;  // 26 KB next to the 64 KB of EvalShared: gfx950 has 160 KB of LDS per CU, one workgroup each here
  WgSpan span{0, 0, 0, false};
  if (BATCH) {
    const uint32_t G = bv.wg_per_hyp, hi = blockIdx.x / G, sub = blockIdx.x - hi * G;
    const size_t h = bv.active[hi];
    st += h; corr.nd += h * bv.bs; corr.coeff += h * bv.bs; corr.status += h * bv.bs; nbr5 += h * 5 * bv.bs;
    partials += h * bv.partial_stride; ticket += h * bv.sync_stride; hist += h * (kHistReplicas * kHistStride);
    span.V = bv.v_grid;
    span.vb0 = (uint32_t)(((unsigned long long)sub * bv.v_grid) / G);
    span.vb1 = (uint32_t)(((unsigned long long)(sub + 1u) * bv.v_grid) / G);
    span.ctl = sub == 0;
  }
  if (st->reg_done) return;
  if (!BATCH && ep.chain_expect && st->done_count != ep.chain_expect) return;  // chained registration whose predecessor was not over: no-op
  const int tid = threadIdx.x;
  // hand-off record: 8 chunks of 16 bytes {value, epoch}, each written / read with ONE sc1 dwordx4 access (atomic as a
  // unit), so a reader that sees the expected epoch in a chunk has that chunk's value: no second round trip, no
  // publisher-side wait between data and flag.  Chunks 0..6 = next pose, chunk 7 = "another evaluation follows".
  u4v* hand = reinterpret_cast<u4v*>(ticket + kHandoffWordOffset);
  // epoch base of this launch: a kernel argument (the host counts its solve launches; 32 epochs per launch), larger than
  // every epoch an earlier launch left in the hand-off record -- no memory round trip before the first pass can start
  const unsigned long long e0 = ep.epoch_base;
  Pose pose = pose_from_array(st->T);
  if (tid == 0) {  // the controller's inputs are constant over the launch (read here, next to the pose: no fetch inside a pass)
    sh.peer_seq = st->peer_seq;
    for (int i = 0; i < 3; ++i) sh.ctl.T[i] = pose.t[i];
    for (int i = 0; i < 4; ++i) sh.ctl.T[3 + i] = pose.q[i];
    sh.ctl.lm_max = st->lm_max; sh.ctl.outer_iter = st->outer_iter; sh.ctl.max_outer = st->max_outer;
  }
  const unsigned long long tag0 = (e0 + 1ull) << 5;  // pass tags: unique over launches (every launch advances the epoch) and passes (slot <= 16)
  int code = eval_pass<true, true, PROF, BATCH, PEER>(0, 1, pose, spx, spy, spz, corr, st, ep, partials, ticket, hist, out, mpts, nbr5, mp, sh, tag0, &cache, hand, e0 + 1ull, span);
  for (int slot = 1; slot <= lm_max; ++slot) {
    __syncthreads();
    const unsigned long long want = e0 + (unsigned long long)slot;
    if (code == kPassMore || code == kPassDone) {
      // this workgroup ran the controller, whose thread has already published {request, more?} with epoch `want` and
      // left them in sh.pose / sh.more
    } else if (tid < 64) {  // wait for the controller's workgroup: lanes 0..7 poll one chunk each
      bool done = tid >= 8;
      unsigned long long val = 0;
      const unsigned long long t0 = wall_clock64();
      bool ok = true;
      for (;;) {
        if (!done) {
          const u4v v = load16_sc1(hand + tid);
          const unsigned long long tag = ((unsigned long long)v.w << 32) | v.z;
          if (tag - e0 >= (unsigned long long)slot && tag - e0 <= 64ull) { done = true; val = ((unsigned long long)v.y << 32) | v.x; }
        }
        if (__ballot(!done) == 0ull) break;
        // (the controller's workgroup may itself be waiting for the other ranks' records: twice its patience + the local 50 ms)
        if (wall_clock64() - t0 > 2ull * ep.timeout_ticks + 5000000ull) { ok = false; break; }  // give up instead of hanging the device
        __builtin_amdgcn_s_sleep(1);
      }
      if (tid < 7) sh.pose[tid] = __longlong_as_double((long long)val);
      if (tid == 7) sh.more = ok ? (int)val : -1;
    }
    __syncthreads();
    if (sh.more != 1) return;  // solve ended (or timeout)
    pose = pose_from_array(sh.pose);
    __syncthreads();
    code = eval_pass<false, true, PROF, BATCH, PEER>(slot, 1, pose, spx, spy, spz, corr, st, ep, partials, ticket, hist, out, mpts, nbr5, mp, sh, tag0 + (unsigned long long)slot, &cache, hand, want + 1ull, span);
  }
}

// controller as its own launch (used when the sums pass through the RCCL all-reduce between eval and control)
__global__ __launch_bounds__(64) void lm_step_kernel(int slot, DevState* st, const LmSums* __restrict__ sums_in, int32_t* __restrict__ hist, EvalParams ep) {
  __shared__ LmSums sh_sums;
  __shared__ LmState sh_S;
  __shared__ LmCtl sh_ctl;
  __shared__ int sh_more;
  if (!eval_slot_active(st, slot)) return;
  const int tid = threadIdx.x;
  copy_words(reinterpret_cast<double*>(&sh_sums), reinterpret_cast<const double*>(sums_in), (int)(sizeof(LmSums) / 8), tid, 64);
  copy_words(reinterpret_cast<double*>(&sh_S), reinterpret_cast<const double*>(&st->S), (int)(sizeof(LmState) / 8), tid, 64);
  load_ctl(sh_ctl, st, tid, 32);
  __syncthreads();
  if (tid == 0) sh_more = lm_control(slot, st, sh_S, sh_sums, sh_ctl);
  __syncthreads();
  copy_words(reinterpret_cast<double*>(&st->S), reinterpret_cast<const double*>(&sh_S), (int)(sizeof(LmState) / 8), tid, 64);
  if (!sh_more) {
    for (int i = tid; i < kHistReplicas * kHistStride; i += 64) hist[i] = 0;
    publish_state(st, ep, sh_ctl.outer_iter, tid, 64);
  }
}

// ------------------------------------------------------------------------------------------------
// Seam B kernels
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void knn_only_kernel(const float* __restrict__ q, uint32_t nq, int k, DevMapView map,
                                                       float gate_d2, float* __restrict__ nbr, float* __restrict__ d2o,
                                                       int32_t* __restrict__ idxo, uint8_t* __restrict__ found,
                                                       uint32_t* __restrict__ fb_list, uint32_t* __restrict__ fb_count) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nq) return;
  const float qx = q[3 * i], qy = q[3 * i + 1], qz = q[3 * i + 2];
  const CellRef c = locate(map, qx, qy, qz);
  if (c.slot < 0) {  // LocalMap.h:499-507 `return false`
    found[i] = 0;
    for (int t = 0; t < k; ++t) {
      d2o[(size_t)i * k + t] = 0; if (idxo) idxo[(size_t)i * k + t] = -1;
      nbr[((size_t)i * k + t) * 3] = 0; nbr[((size_t)i * k + t) * 3 + 1] = 0; nbr[((size_t)i * k + t) * 3 + 2] = 0;
    }
    return;
  }
  found[i] = 1;
  Top5 top;
  top.init();
  knn27(map, c, qx, qy, qz, top);
  const unsigned long long b[5] = {top.b0, top.b1, top.b2, top.b3, top.b4};
  const unsigned long long kth = b[k - 1];
  // exact only if the k-th best lies inside the radius the 27-cell block is guaranteed to cover
  const bool exact = (kth != ~0ull) && (__uint_as_float((uint32_t)(kth >> 32)) <= gate_d2);
  if (!exact) { fb_list[atomicAdd(fb_count, 1u)] = i; return; }
  for (int t = 0; t < k; ++t) {
    const uint32_t id = (uint32_t)b[t];
    d2o[(size_t)i * k + t] = __uint_as_float((uint32_t)(b[t] >> 32));
    if (idxo) idxo[(size_t)i * k + t] = (int32_t)id;
    { const float4 p = map.pts[id]; nbr[((size_t)i * k + t) * 3] = p.x; nbr[((size_t)i * k + t) * 3 + 1] = p.y; nbr[((size_t)i * k + t) * 3 + 2] = p.z; }
  }
}

// one WAVEFRONT per query: 64 lanes stride over every point of the query's cube, lane-local top-5,
// then five rounds of wavefront-min to merge (exact k-NN for far / sparse queries).
__global__ __launch_bounds__(256) void knn_fallback_kernel(const float* __restrict__ q, const uint32_t* __restrict__ fb_list,
                                                           uint32_t n_fb, int k, DevMapView map, float* __restrict__ nbr,
                                                           float* __restrict__ d2o, int32_t* __restrict__ idxo) {
  const uint32_t w = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  if (w >= n_fb) return;
  const uint32_t i = fb_list[w];
  const float qx = q[3 * i], qy = q[3 * i + 1], qz = q[3 * i + 2];
  const CellRef c = locate(map, qx, qy, qz);
  const uint32_t* tbl = map.cell_start + (size_t)c.slot * map.ncell1;
  const uint32_t beg = tbl[0], end = tbl[map.ncell1 - 1];
  Top5 top;
  top.init();
  for (uint32_t p = beg + lane; p < end; p += 64) {
    const float4 mp = map.pts[p];
    const float d2 = l2_d2(qx, qy, qz, mp.x, mp.y, mp.z);
    top.insert(((unsigned long long)__float_as_uint(d2) << 32) | p);
  }
  for (int t = 0; t < k; ++t) {
    const unsigned long long m = wave_min_u64(top.b0);
    if (top.b0 == m && m != ~0ull) { top.b0 = top.b1; top.b1 = top.b2; top.b2 = top.b3; top.b3 = top.b4; top.b4 = ~0ull; }  // keys are unique
    if (lane == 0) {
      if (m != ~0ull) {
        const uint32_t id = (uint32_t)m;
        d2o[(size_t)i * k + t] = __uint_as_float((uint32_t)(m >> 32));
        if (idxo) idxo[(size_t)i * k + t] = (int32_t)id;
        { const float4 p = map.pts[id]; nbr[((size_t)i * k + t) * 3] = p.x; nbr[((size_t)i * k + t) * 3 + 1] = p.y; nbr[((size_t)i * k + t) * 3 + 2] = p.z; }
      } else {  // fewer than k points in the cube: nanoflann.h:87-100 buffer state
        d2o[(size_t)i * k + t] = (t == k - 1) ? 3.402823466e+38f : 0.f;
        if (idxo) idxo[(size_t)i * k + t] = (int32_t)beg;
        { const float4 p = map.pts[beg]; nbr[((size_t)i * k + t) * 3] = p.x; nbr[((size_t)i * k + t) * 3 + 1] = p.y; nbr[((size_t)i * k + t) * 3 + 2] = p.z; }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// launch wrappers
// ------------------------------------------------------------------------------------------------
static inline dim3 grid_for(uint32_t n, int block) { return dim3((n + block - 1) / block); }

void launch_reg_begin(DevState* st, const double pose[7], int max_outer, int lm_max, int32_t* hist, hipStream_t s) {
  RegBeginArgs a{};  // (chain_expect = 0: the guess is the argument, not DevState::T_chain)
  for (int i = 0; i < 7; ++i) a.pose[i] = pose[i];
  a.max_outer = max_outer; a.lm_max = lm_max;
  hipLaunchKernelGGL(reg_begin_kernel, dim3(1), dim3(512), 0, s, st, a, hist);
}
static const BatchView kNoBatch{nullptr, nullptr, 0, 0, 0, 0, 0, 0};
void launch_scan_keys(const float* d_scan, uint32_t n, DevState* st, const double pose[7], int max_outer, int lm_max, int32_t* hist,
                      const DevMapView& map, int max_sf, int rank, int world, uint32_t* keys, uint32_t* vals, uint8_t* status,
                      const BinTable& bin, hipStream_t s, bool rebin, const BatchView* bv, uint32_t n_hyp, bool qsplit, uint32_t n_total,
                      unsigned long long* prebin_ctr) {
  RegBeginArgs a{};
  if (bv) {  // (the hypotheses' prologue arguments are in bv->begin)
    if (!n || !n_hyp) return;
    hipLaunchKernelGGL(scan_keys_kernel<true>, dim3((n + 255u) / 256u, n_hyp), dim3(256), 0, s, d_scan, n, st, a, hist, map, max_sf, rank, world,
                       keys, vals, status, bin, 0, *bv, 0, n, nullptr);
    return;
  }
  if (!n) { if (!rebin && !prebin_ctr) launch_reg_begin(st, pose, max_outer, lm_max, hist, s); return; }
  for (int i = 0; i < 7; ++i) a.pose[i] = pose[i];
  a.max_outer = max_outer; a.lm_max = lm_max;
  hipLaunchKernelGGL(scan_keys_kernel<false>, grid_for(n, 256), dim3(256), 0, s, d_scan, n, st, a, hist, map, max_sf, rank, world, keys, vals, status,
                     bin, rebin ? 1 : 0, kNoBatch, qsplit ? 1 : 0, qsplit ? n_total : n, prebin_ctr);
}
void launch_reg_begin_prebinned(DevState* st, const double pose[7], int max_outer, int lm_max, int32_t* hist, const unsigned long long* ctr,
                                uint8_t* status, uint32_t n, int max_sf, hipStream_t s) {
  RegBeginArgs a{};  // (chain_expect = 0: the guess is the argument, not DevState::T_chain)
  for (int i = 0; i < 7; ++i) a.pose[i] = pose[i];
  a.max_outer = max_outer; a.lm_max = lm_max;
  const bool sampling = max_sf >= 0 && n > (uint32_t)max_sf;
  hipLaunchKernelGGL(reg_begin_prebinned_kernel, sampling ? grid_for(n, 256) : dim3(1), dim3(256), 0, s, st, a, hist, ctr, status, sampling ? n : 0u, max_sf);
}
void launch_bin_offsets(const BinTable& bt, uint32_t* chunk_start, uint32_t chunk_cap, DevState* st, hipStream_t s, const BatchView* bv,
                        uint32_t n_hyp, unsigned long long* packed_ctr) {
  const uint32_t gx = (1u << bt.log2_size) / 4096u;
  if (bv) { if (n_hyp) hipLaunchKernelGGL(bin_offsets_kernel<true>, dim3(gx, n_hyp), dim3(1024), 0, s, bt, chunk_start, chunk_cap, st, *bv, nullptr); }
  else hipLaunchKernelGGL(bin_offsets_kernel<false>, dim3(gx), dim3(1024), 0, s, bt, chunk_start, chunk_cap, st, kNoBatch, packed_ctr);
}
void launch_bin_place(const BinTable& bt, const float* d_scan, uint32_t n, const uint32_t* qslot, const uint32_t* qrank, float4* binned,
                      hipStream_t s, const DevState* st_if_rebin, const BatchView* bv, uint32_t n_hyp) {
  if (!n) return;
  if (bv) { if (n_hyp) hipLaunchKernelGGL(bin_place_kernel<true>, dim3((n + 255u) / 256u, n_hyp), dim3(256), 0, s, bt, d_scan, n, qslot, qrank, binned, nullptr, *bv); }
  else hipLaunchKernelGGL(bin_place_kernel<false>, grid_for(n, 256), dim3(256), 0, s, bt, d_scan, n, qslot, qrank, binned, st_if_rebin, kNoBatch);
}
void launch_knn_plane(const float4* binned,
                      const uint32_t* chunk_start, const DevState* st, const DevMapView& map, const MatchParams& mp,
                      CorrBuffers corr, uint32_t* nbr5, int32_t* hist, hipStream_t s, hipEvent_t ev_start, hipEvent_t ev_stop,
                      const BatchView* bv, uint32_t n_hyp) {
  if (bv) {
    // B x ~4 800 chunks: 256 workgroups per hypothesis, every wavefront walks ~5 chunks (no tail to hide with 64 hypotheses in
    // flight, and a quarter of the workgroup prologues of the one-round grid of the single registration)
    if (!n_hyp) return;
    if (mp.ablate) hipLaunchKernelGGL((knn_plane_kernel<true, true>), dim3(kKnnBlocks / 4, n_hyp), dim3(256), 0, s, binned, chunk_start, st,
                                      map.pts, map.cell_start, map, mp, corr, nbr5, hist, *bv);
    else hipLaunchKernelGGL((knn_plane_kernel<false, true>), dim3(kKnnBlocks / 4, n_hyp), dim3(256), 0, s, binned, chunk_start, st,
                            map.pts, map.cell_start, map, mp, corr, nbr5, hist, *bv);
    return;
  }
  // The production instantiation carries no profiling code; SOICP_ABLATE != 0 selects the instrumented one.
  // Timing events ride on the kernel's own dispatch packet (no marker packets: separate hipEventRecord calls cost
  // ~3.7 us of stream time each, 8 % of a registration when every sweep is timed)
  auto* k = mp.ablate ? knn_plane_kernel<true, false>
                      : (mp.begin ? (mp.begin_args.chain_expect ? knn_plane_kernel<false, false, 2> : knn_plane_kernel<false, false, 1>) : knn_plane_kernel<false, false>);  // (the host never sets begin with ablate)
  if (ev_start && ev_stop)
    hipExtLaunchKernelGGL(k, dim3(kKnnBlocks), dim3(256), 0, s, ev_start, ev_stop, 0, binned,
                          chunk_start, st, map.pts, map.cell_start, map, mp, corr, nbr5, hist, kNoBatch);
  else
    hipLaunchKernelGGL(k, dim3(kKnnBlocks), dim3(256), 0, s, binned, chunk_start, st, map.pts,
                       map.cell_start, map, mp, corr, nbr5, hist, kNoBatch);
}
void launch_knn_query_waves(const float* d_scan, uint32_t n, DevState* st, const double pose[7], int max_outer, int lm_max, bool begin, int32_t* hist,
                            const DevMapView& map, const MatchParams& mp, int max_sf, uint8_t* status, uint32_t* nbr5, hipStream_t s,
                            hipEvent_t ev_start, hipEvent_t ev_stop, uint32_t chain_expect) {
  if (!n) return;
  RegBeginArgs a{};
  if (begin) { for (int i = 0; i < 7; ++i) a.pose[i] = pose[i]; a.max_outer = max_outer; a.lm_max = lm_max; }
  if (begin) a.chain_expect = chain_expect;
  // one wavefront per run of points the sampling rule keeps about one of (knn_query_wave_kernel); max_surface_features == 0 keeps nothing
  const bool sampled = max_sf > 0 && n > (uint32_t)max_sf;
  const double per_wave = sampled ? (double)n / (double)max_sf : 1.0;
  const uint32_t n_waves = sampled ? (uint32_t)max_sf + 1u : n;  // first(max_sf) = ceil(max_sf * (n / max_sf)) >= n - 1: the last share is short or empty
  auto* k = begin ? knn_query_wave_kernel<true> : knn_query_wave_kernel<false>;
  const dim3 grid((n_waves + 3u) / 4u + ((!begin && mp.publish_prev) ? 1u : 0u));  // (+ the workgroup that publishes the deferred report)
  if (ev_start && ev_stop)
    hipExtLaunchKernelGGL(k, grid, dim3(256), 0, s, ev_start, ev_stop, 0, d_scan, n, n_waves, per_wave, st, st, a, hist, map.pts, map.cell_start, map, mp, max_sf, status, nbr5);
  else
    hipLaunchKernelGGL(k, grid, dim3(256), 0, s, d_scan, n, n_waves, per_wave, st, st, a, hist, map.pts, map.cell_start, map, mp, max_sf, status, nbr5);
}
uint32_t solve_grid(uint32_t n_upper, uint32_t max_blocks) {
  uint32_t blocks = (n_upper + 255u) / 256u;
  blocks = blocks < 1 ? 1 : (blocks > (uint32_t)kFitBlocksMax ? (uint32_t)kFitBlocksMax : blocks);
  if (max_blocks >= 1 && blocks > max_blocks) blocks = max_blocks;
  return blocks;
}
void launch_eval(int slot, bool fuse_lm, const float* spx, const float* spy, const float* spz, const CorrBuffers& corr,
                 DevState* st, const EvalParams& ep, double* partials, uint32_t* ticket, int32_t* hist, LmSums* sums,
                 const DevMapView& map, const uint32_t* nbr5, const MatchParams& mp, uint32_t n_upper, hipStream_t s) {
  const bool prof = ep.ablate != 0 || mp.ablate != 0;
  if (slot == 0) {  // plane fit + first evaluation, dense over the queries
    const uint32_t blocks = solve_grid(n_upper, 0);
    auto* k = prof ? eval_kernel<true, true> : eval_kernel<true, false>;
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, s, slot, fuse_lm ? 1 : 0, spx, spy, spz, corr, st, ep, partials, ticket, hist, sums, map.pts, nbr5, mp);
  } else {
    auto* k = prof ? eval_kernel<false, true> : eval_kernel<false, false>;
    hipLaunchKernelGGL(k, dim3(kEvalBlocks), dim3(256), 0, s, slot, fuse_lm ? 1 : 0, spx, spy, spz, corr, st, ep, partials, ticket, hist, sums, map.pts, nbr5, mp);
  }
}
void launch_solve(int lm_max, const float* spx, const float* spy, const float* spz, const CorrBuffers& corr, DevState* st,
                  const EvalParams& ep, double* partials, uint32_t* ticket, int32_t* hist, LmSums* sums, const DevMapView& map,
                  const uint32_t* nbr5, const MatchParams& mp, uint32_t n_upper, uint32_t max_blocks, hipStream_t s) {
  // every workgroup must be resident for the whole launch (94 KB of LDS: one per compute unit)
  const uint32_t blocks = solve_grid(n_upper, max_blocks);
  const bool prof = ep.ablate != 0 || mp.ablate != 0;
  auto* k = ep.peer_world > 1 ? (prof ? solve_kernel<true, false, true> : solve_kernel<false, false, true>)
                              : (prof ? solve_kernel<true, false, false> : solve_kernel<false, false, false>);
  hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, s, lm_max, spx, spy, spz, corr, st, ep, partials, ticket, hist, sums, map.pts, nbr5, mp, kNoBatch);
}
uint32_t solve_batch_resident_blocks(uint32_t n_cus, int wg_per_cu) {
  int nb = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, solve_kernel<false, true>, 256, 0) != hipSuccess || nb < 1) { (void)hipGetLastError(); nb = 1; }
  if (nb > 2) nb = 2;  // (registers: two wavefronts per SIMD; LDS: two 66 KB workgroups)
  if (wg_per_cu >= 1 && wg_per_cu < nb) nb = wg_per_cu;
  return (uint32_t)nb * n_cus;
}
void launch_solve_batch(int lm_max, const float* spx, const float* spy, const float* spz, const CorrBuffers& corr, DevState* st,
                        const EvalParams& ep, double* partials, uint32_t* ticket, int32_t* hist, const DevMapView& map,
                        const uint32_t* nbr5, const MatchParams& mp, const BatchView& bv, uint32_t n_hyp, hipStream_t s) {
  if (!n_hyp) return;
  const bool prof = ep.ablate != 0 || mp.ablate != 0;
  auto* k = prof ? solve_kernel<true, true> : solve_kernel<false, true>;
  hipLaunchKernelGGL(k, dim3(n_hyp * bv.wg_per_hyp), dim3(256), 0, s, lm_max, spx, spy, spz, corr, st, ep, partials, ticket, hist, (LmSums*)nullptr, map.pts,
                     nbr5, mp, bv);
}
void launch_lm_step(int slot, DevState* st, const LmSums* sums, int32_t* hist, const EvalParams& ep, hipStream_t s) {
  hipLaunchKernelGGL(lm_step_kernel, dim3(1), dim3(64), 0, s, slot, st, sums, hist, ep);
}
// peer exchange self-test (so_icp_peer_connect): the same stores and loads as the solve's exchange, one chunk per rank pair
__global__ __launch_bounds__(64) void peer_selftest_kernel(EvalParams ep, uint32_t tag, int32_t* __restrict__ ok_out) {
  const int tid = threadIdx.x;
  const size_t base = (size_t)2 * kPeerMaxWorld * kPeerChunks;
  if (tid < ep.peer_world) {
    const u4v v = {0x50454552u, (unsigned int)ep.peer_rank, tag, (unsigned int)tid};
    store16_sys(reinterpret_cast<u4v*>(ep.peer_inbox[tid]) + base + ep.peer_rank, v);
  }
  bool done = tid >= ep.peer_world;
  const unsigned long long t0 = wall_clock64();
  bool ok = true;
  for (;;) {
    if (!done) {
      const u4v v = load16_sys(reinterpret_cast<const u4v*>(ep.peer_inbox[ep.peer_rank]) + base + tid);
      if (v.z == tag && v.x == 0x50454552u && v.y == (unsigned int)tid) done = true;
    }
    if (__ballot(!done) == 0ull) break;
    if (wall_clock64() - t0 > 200000000ull) { ok = false; break; }  // 2 s
    __builtin_amdgcn_s_sleep(8);
  }
  if (tid == 0) *ok_out = ok ? 1 : 0;
}
void launch_peer_selftest(void* const inbox[8], int rank, int world, uint32_t tag, int32_t* d_ok, hipStream_t s) {
  EvalParams ep{};
  for (int i = 0; i < 8; ++i) ep.peer_inbox[i] = inbox[i];
  ep.peer_rank = rank; ep.peer_world = world;
  hipLaunchKernelGGL(peer_selftest_kernel, dim3(1), dim3(64), 0, s, ep, tag, d_ok);
}
void launch_knn_only(const float* q, uint32_t nq, int k, const DevMapView& map, float gate_d2, float* nbr, float* d2,
                     int32_t* idx, uint8_t* found, uint32_t* fb_list, uint32_t* fb_count, hipStream_t s) {
  if (!nq) return;
  hipLaunchKernelGGL(knn_only_kernel, grid_for(nq, 256), dim3(256), 0, s, q, nq, k, map, gate_d2, nbr, d2, idx, found, fb_list, fb_count);
}
void launch_knn_fallback(const float* q, const uint32_t* fb_list, uint32_t n_fb, int k, const DevMapView& map, float* nbr,
                         float* d2, int32_t* idx, hipStream_t s) {
  if (!n_fb) return;
  hipLaunchKernelGGL(knn_fallback_kernel, dim3((n_fb + 3) / 4), dim3(256), 0, s, q, fb_list, n_fb, k, map, nbr, d2, idx);
}

}  // namespace soicp
