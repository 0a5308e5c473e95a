// kernels.h -- launch interface of the gfx950 kernels (kernels.hip).  Host-only declarations.
#pragma once
#include <hip/hip_runtime_api.h>

#include <cstddef>
#include <cstdint>

#include "lm_solver.h"
#include "so_math.h"

namespace soicp {

// read-only view of the HBM-resident map shard (layout: local_map.h)
struct DevMapView {
  const float4* pts;           // {x, y, z, 0} per map point, canonical order (+ 64 B of readable padding)
  const uint32_t* cell_start;  // n_slots * (nc^3+1)
  const int32_t* cube_slot;    // 4851
  int32_t nc;
  uint32_t ncell1;             // nc^3 + 1
  double inv_cell;
  int32_t origin[3];
  uint32_t n_points;
  uint32_t n_slots;            // occupied cubes (tables)
};

// registration prologue arguments (the guess and the loop bounds travel as kernel arguments; a batch reads them from memory)
// chain_expect != 0 (so_icp_register_sequence): the guess was formed ON THE DEVICE by the last solve of the registration before,
// DevState::T_chain = its result o the delta that solve carried (EvalParams::chain_delta; pose_compose, so_math.h), and the launch is
// valid only if exactly chain_expect registrations have completed on this state block (DevState::done_count) -- it was enqueued behind
// the launches of the registration before, which may turn out to need more.
struct RegBeginArgs { double pose[7]; int32_t max_outer, lm_max; uint32_t chain_expect, pad; };

struct MatchParams {
  float plane_res;        // localMap.planeRes_ (float member, LocalMap.h:761)
  float sq_max_dist_f;    // 3 * planeRes evaluated in float (LidarSlam.cpp:526)
  double max_point_dist;  // planeRes / 2.0 (LidarSlam.cpp:820)
  int32_t ablate;         // profiling / test switches (env SOICP_ABLATE), read only by the PROF instantiations of the kernels,
                          // which are launched when it is non-zero: bit0 skip plane fit, bit1 skip scan, bit2 skip re-rank, ...
  unsigned long long* kdbg;  // profiling only (SOICP_ABLATE bit 7): 2 sweeps x 4*kKnnBlocks wavefront records of 16 stamps, else nullptr
  uint32_t* packed_leftover;  // &DevState::packed_leftover of the registration (hypothesis 0 of a batch), passed beside `st` so that the
                              // kernel's loads through its read-only, restrict-qualified `st` stay scalar loads
  int32_t pack_light;     // 1: four light chunks per wavefront (knn_plane_kernel, SO_KNN_PACK); 0: one chunk per wavefront throughout
  int32_t skip_near_pass; // 1: the sweep starts with the FULL pass (gate radius).  Round 0 of a batch of hypotheses +-0.5 m / +-5 degrees
                          // off: the near pass (half a cell) certifies almost nothing there and its scan is wasted (exact either way)
  uint32_t chunk_cap;     // entries of the chunk list buffer: LIGHT chunks (<= 16 queries) are listed from its top downwards
  uint32_t chain_expect;  // != 0: a launch of a CHAINED registration (enqueued before the registration in front of it had reported): a no-op
                          // unless DevState::done_count == chain_expect, i.e. unless that registration really was over when this one began
  // deferred report: when the solve of outer iteration i-1 left the publication of its state block to the k-NN launch of
  // iteration i (EvalParams::defer_publish), that launch's first workgroup writes it to hring[(i-1) & 1] (see EvalParams)
  struct DevState* hring[2];
  unsigned long long seq_base;
  int32_t publish_prev;
  // begin != 0: this launch is the FIRST kernel of a registration whose scan was binned ahead (so_icp_stage_scan): it carries the
  // registration prologue itself -- one launch less on the path from the call to the first sweep, where the host's enqueue rate is
  // the limit.  Every workgroup takes the pose, the loop state (iteration 0, not done) and the work-list counters from HERE instead
  // of the state block; workgroup 0 writes the prologue into begin_state for the launches behind this one.
  int32_t begin, begin_max_surface_features;
  uint32_t begin_n;                        // points of the scan (the sampling rule's DROPPED status bytes are written by this launch)
  RegBeginArgs begin_args;                 // guess + loop bounds
  const unsigned long long* begin_ctr;     // work-list counters left by the binning
  struct DevState* begin_state;
};

struct EvalParams {
  double a2;       // TukeyLoss a^2 with a = (double)sqrtf(3*planeRes)  (LidarSlam.cpp:271)
  int32_t variant; // 0: Ceres 2.0.0, 1: Ceres >= 2.1
  int32_t ablate;  // profiling / test switches (env SOICP_ABLATE; PROF instantiations only): bit5 skip the LM controller, bit6 skip the point loop
  // the evaluation kernels walk the queries in ORIGINAL scan order (deterministic sums whatever the spatial binning did):
  // spx/spy/spz = scan, scan+1, scan+2 with q_stride 3; correspondence records are indexed by the original query index
  uint32_t n_queries, q_stride;
  // Read-back without a copy engine round trip: when a solve ends, the controller's workgroup stores the whole state
  // block into the pinned, host-coherent mirror hring[outer & 1] and then publishes seq_base | (outer + 1) in its seq
  // word (system-scope release); the host polls that word.  hring[0] == nullptr disables it (host uses hipMemcpyAsync).
  struct DevState* hring[2];
  unsigned long long seq_base;
  // 1: a solve that does NOT end the registration leaves the publication to the next k-NN launch (already enqueued by the
  // host): the L2 write-back + system fence + PCIe stores (2.7 us) then overlap that sweep instead of delaying it
  int32_t defer_publish;
  uint32_t chain_expect;  // see MatchParams::chain_expect
  // chain_next != 0: a chained registration may follow this one (so_icp_register_sequence): the solve that ends the registration leaves
  // DevState::T_chain = result o chain_delta, the guess the first launch of the next registration starts from
  int32_t chain_next;
  double chain_delta[7];
  // persistent solve: epoch base of the launch (hand-off epochs and pass tags count up from it; strictly increasing per context)
  unsigned long long epoch_base;
  // Peer exchange (sharded map, persistent solve): every rank's inbox is mapped into this process (hipIpc, or the plain
  // pointer for contexts of one process).  After a pass's local reduction, workgroup 0 of every rank PUSHES its record
  // (29 sums, + 16 histogram counters in the fit pass) as tagged 16-byte chunks into every rank's inbox with system-scope
  // stores and polls its own inbox until the chunks of all ranks carry the pass's sequence number; the records are added
  // in rank order, so every rank holds bit-identical sums and runs the same controller -- no collective launch, no host.
  void* peer_inbox[8];
  int32_t peer_rank, peer_world;  // peer_world <= 1: off
  // How long (ticks of the 100 MHz wall clock) the waits inside a persistent solve launch hold out before the launch is
  // abandoned and the host reports it.  50 ms without peers (only a missing co-resident workgroup can be late); with the
  // peer exchange a whole RANK can be late -- its process descheduled, its queue waiting for a time slice of a shared
  // device -- so the host passes SOICP_PEER_TIMEOUT_MS (default 1 000 ms) there.
  unsigned long long timeout_ticks;
};
constexpr int kPeerMaxWorld = 8, kPeerChunks = 48;                     // chunks per (parity, source rank): 45 used
constexpr size_t kPeerInboxBytes = (size_t)(2 * kPeerMaxWorld * kPeerChunks + kPeerMaxWorld) * 16;  // + one self-test chunk per source

// per-correspondence record written by the k-NN + plane-fit kernel, read by the evaluation kernel
struct CorrBuffers {
  double4* nd;      // {nx, ny, nz, negative_OA_dot_norm}
  double* coeff;    // residualCoefficient; 0 for rejected points
  uint8_t* status;  // MatchingResult
};

// ---- device-resident registration state: the outer ICP loop and the LM controller advance on the GPU, the
// host only enqueues the (static) kernel sequence and reads this block back (no host round trip per evaluation)
struct DevIterStats {
  double translation_norm, rotation_norm;
  int32_t num_surf, lm_iterations, num_successful, termination;
  double initial_cost, final_cost;
  int32_t reject_hist[7], obs_hist[9];
  double pose_after[7];
};
struct DevState {
  // the host's inputs (kernel arguments of reg_begin_kernel)
  double pose_in[7];
  int32_t max_outer, lm_max, pad0, pad1;
  // device-side control
  int32_t outer_iter, reg_done, lm_more, n_iterations;
  uint32_t packed_leftover;  // k-NN sweeps since the context was created (a running count: the host takes differences): queries of packed light chunks that the packed near pass could not finish (exact per-lane scan)
  uint32_t pad2[3];
  // work-list counters of the hash binning in ONE word (kept queries | normal chunks << 21 | light chunks (<= 16 queries,
  // listed separately) << 42), so that a workgroup of bin_offsets_kernel reserves its three ranges with one atomic round trip
  unsigned long long bin_packed;
  double T[7];          // pose of the current outer iteration (T_w_lidar)
  double eval_pose[7];  // pose the next LM evaluation is requested at
  LmState S;
  double JtJ[36], Jtr[6];
  DevIterStats iters[16];
  // chained registrations (so_icp_register_sequence): the pose the LAST completed registration ended with and the number of registrations
  // completed on this block -- written when a registration ends, never by a prologue, so that the first launch of a chained
  // registration (whose workgroup 0 rewrites the fields above) can read both from every workgroup
  double T_final[7];
  double T_chain[7];     // T_final o the delta the ending solve carried (EvalParams::chain_delta)
  uint32_t done_count, pad3;
  unsigned long long peer_seq;  // peer exchange: passes exchanged so far by this context (tag and double-buffer parity; equal on all ranks)
  unsigned long long dbg[16];  // profiling aid (SOICP_ABLATE bit 7): wall-clock stamps of the last evaluation's phases
  unsigned long long seq;      // host mirror only: publication word (see EvalParams::hring), written last
};
static_assert(sizeof(DevState) % 8 == 0, "DevState is copied in 8-byte words");

constexpr int kHistReplicas = 16;    // histogram atomics are spread over replicas (contention), summed by eval_kernel
constexpr int kHistStride = 32;      // ints per replica: reject[7] obs[9] stats[4]
constexpr int kKnnBlocks = 1024;     // 4096 wavefronts = what the chip holds at 4 per SIMD; one chunk each, and a second one for the
                                     // wavefronts whose first chunk is a light one (see knn_plane_kernel: everything in ONE round)
#ifndef SO_SOLVE_BLOCKS
#define SO_SOLVE_BLOCKS 256
#endif
constexpr int kEvalBlocks = SO_SOLVE_BLOCKS;     // workgroups of the evaluation / solve launches (256 = one per CU, 512 = two)
constexpr int kFitBlocksMax = SO_SOLVE_BLOCKS;
// cross-workgroup synchronisation block of the evaluation kernels: 16 arrival counters on their own 128-byte lines,
// then the 8 x 16-byte hand-off record {value, epoch} of solve_kernel
constexpr int kArriveCounters = 16, kArriveStrideWords = 32, kHandoffWordOffset = kArriveCounters * kArriveStrideWords;
constexpr int kSyncBytes = kHandoffWordOffset * 4 + 8 * 16;
constexpr int kSumsStride = 48;      // doubles per partial record (45 used)
constexpr int kRecordChunksMax = 40; // 16-byte chunks per workgroup record of the persistent solve (29 sums + 8 histogram pairs)

// so_icp_register_batch: B hypotheses (initial poses) of ONE scan advance in the same launches.  Every per-registration
// array exists once per hypothesis with a common element stride; a launch serves the hypotheses listed in `active`
// (blockIdx.y for the binning and k-NN kernels, blockIdx.x / wg_per_hyp for the solve launch).
struct BatchView {
  const uint32_t* active;     // [hypotheses of this launch] index of the hypothesis (device memory)
  const RegBeginArgs* begin;  // [B] prologue arguments (device memory; scan_keys)
  uint32_t bs;                // elements per hypothesis of the per-query arrays (table slot / rank, perm, binned SoA scan, status,
                              // plane records) and of the chunk list (= its capacity); the neighbour lists hold 5 bs
  uint32_t table_stride;      // bin table entries per hypothesis
  uint32_t partial_stride;    // doubles per hypothesis in the record table of the solve launch
  uint32_t sync_stride;       // 32-bit words per hypothesis in the hand-off block
  uint32_t wg_per_hyp;        // solve: real workgroups per hypothesis
  uint32_t v_grid;            // solve: workgroups of the single-registration launch they stand in for
};

void launch_reg_begin(DevState* st, const double pose[7], int max_outer, int lm_max, int32_t* d_hist, hipStream_t s);
// Spatial binning of the scan without a sort: an open-addressing table keyed by the sort key (cube slot | half-cell Morton
// code) counts the queries of every key, a block scan over the table turns the counts into bucket offsets + the chunk
// list, and a scatter places the queries.  The order INSIDE a bucket depends on the order of the atomics -- harmless: it
// only decides which queries of one octant share a wavefront (results are exact per query, evaluation sums run in scan order).
struct BinTable {
  uint32_t* key;   // [size] 0xFFFFFFFF = empty (restored by launch_bin_offsets)
  uint32_t* cnt;   // [size] queries per key (zeroed again by launch_bin_offsets)
  uint32_t* off;   // [size] first binned position of the key
  uint32_t log2_size;
};
// per query: table slot (0xFFFFFFFF = dropped) and rank inside its bucket
// (bv / n_hyp: batched launch over the hypotheses bv->active[0 .. n_hyp), see BatchView; nullptr / 0 = one registration)
void launch_bin_offsets(const BinTable& bt, uint32_t* d_chunk_start, uint32_t chunk_cap, DevState* st, hipStream_t s,
                        const BatchView* bv = nullptr, uint32_t n_hyp = 0,
                        unsigned long long* d_packed_ctr = nullptr /* a scan binned ahead of its registration: its own work-list counters */);
// binned[pos] = {x, y, z, query index (bits)} of the query filed at position pos of its bucket
void launch_bin_place(const BinTable& bt, const float* d_scan_xyz, uint32_t n, const uint32_t* d_qslot, const uint32_t* d_qrank,
                      float4* d_binned, hipStream_t s, const DevState* st_if_rebin = nullptr,
                      const BatchView* bv = nullptr, uint32_t n_hyp = 0);

// scan_keys also runs the registration prologue (reg_begin) in its first workgroup; n == 0 launches the prologue alone
void launch_scan_keys(const float* d_scan_xyz, uint32_t n, DevState* st, const double pose[7], int max_outer, int lm_max,
                      int32_t* d_hist, const DevMapView& map, int max_surface_features, int rank, int world, uint32_t* d_keys,
                      uint32_t* d_vals, uint8_t* d_status /* SO_MATCH_DROPPED for queries that are not processed */,
                      const BinTable& bin /* d_keys / d_vals receive the query's table slot / rank inside its bucket */, hipStream_t s,
                      bool rebin = false /* sharded map, outer iteration >= 1: keys under the CURRENT device-resident pose, no prologue */,
                      const BatchView* bv = nullptr, uint32_t n_hyp = 0,
                      bool qsplit = false /* N > 1 with the QUERIES split: d_scan is this rank's share (64-point segments rank, rank + world,
                                             ... of a scan of n_total points); every query is owned, the sampling rule uses the global index */,
                      uint32_t n_total = 0,
                      unsigned long long* d_prebin_ctr = nullptr /* binning AHEAD of the scan's registration (so_icp_stage_scan): no prologue, no status
                                                                    bytes, `st` untouched; `pose` = the guess of the registration in flight */);
// prologue of a registration whose scan was binned ahead: guess + loop bounds, the work-list counters out of *d_ctr, DROPPED status bytes
// of the points the sampling rule leaves out
void launch_reg_begin_prebinned(DevState* st, const double pose[7], int max_outer, int lm_max, int32_t* d_hist, const unsigned long long* d_ctr,
                                uint8_t* d_status, uint32_t n, int max_surface_features, hipStream_t s);
void launch_knn_plane(const float4* d_binned,
                      const uint32_t* d_chunk_start, const DevState* st, const DevMapView& map, const MatchParams& mp,
                      CorrBuffers corr, uint32_t* d_nbr5 /*5 canonical indices per query*/,
                      int32_t* d_hist /*kHistReplicas*kHistStride*/, hipStream_t s, hipEvent_t ev_start = nullptr,
                      hipEvent_t ev_stop = nullptr /* timing events attached to the dispatch itself */,
                      const BatchView* bv = nullptr, uint32_t n_hyp = 0);
// The sweep of a SMALL scan: one wavefront per scan point, no binning launches, no chunk list (knn_query_wave_kernel).  begin: first
// launch of the registration -- carries the prologue (guess + loop bounds from the arguments) and the DROPPED status bytes of the
// sampling rule.  Leaves what launch_knn_plane leaves: a status byte per processed query, five canonical indices where PENDING.
constexpr uint32_t kQueryWaveMaxKept = 4096;  // kept queries up to which every query gets a wavefront of its own (all resident at once)
void launch_knn_query_waves(const float* d_scan_xyz, uint32_t n, DevState* st, const double pose[7], int max_outer, int lm_max, bool begin,
                            int32_t* d_hist, const DevMapView& map, const MatchParams& mp, int max_surface_features, uint8_t* d_status,
                            uint32_t* d_nbr5, hipStream_t s, hipEvent_t ev_start = nullptr, hipEvent_t ev_stop = nullptr,
                            uint32_t chain_expect = 0 /* begin launch of a chained registration: the guess is DevState::T_chain (RegBeginArgs) */);
void launch_eval(int slot, bool fuse_lm, const float* spx, const float* spy, const float* spz, const CorrBuffers& corr,
                 DevState* st, const EvalParams& ep, double* d_partials, uint32_t* d_ticket, int32_t* d_hist,
                 LmSums* d_sums, const DevMapView& map, const uint32_t* d_nbr5, const MatchParams& mp, uint32_t n_upper,
                 hipStream_t s);
// the whole solve (slot 0 .. lm_max) of one outer iteration in one launch (single device only)
void launch_solve(int lm_max, const float* spx, const float* spy, const float* spz, const CorrBuffers& corr, DevState* st,
                  const EvalParams& ep, double* d_partials, uint32_t* d_ticket, int32_t* d_hist, LmSums* d_sums,
                  const DevMapView& map, const uint32_t* d_nbr5, const MatchParams& mp, uint32_t n_upper, uint32_t max_blocks, hipStream_t s);
// workgroups of the single-registration solve launch for a scan of n_upper queries (the grid a batch stands in for)
uint32_t solve_grid(uint32_t n_upper, uint32_t max_blocks);
// workgroups of the batched solve launch that are certainly resident together on n_cus compute units (occupancy query of the
// BATCH instantiation, at most two per compute unit; wg_per_cu > 0 overrides the per-CU number downwards)
uint32_t solve_batch_resident_blocks(uint32_t n_cus, int wg_per_cu);
// the solves of the hypotheses bv.active[0 .. n_hyp) in one launch of n_hyp * bv.wg_per_hyp workgroups (all resident)
void launch_solve_batch(int lm_max, const float* spx, const float* spy, const float* spz, const CorrBuffers& corr, DevState* st,
                        const EvalParams& ep, double* d_partials, uint32_t* d_ticket, int32_t* d_hist, const DevMapView& map,
                        const uint32_t* d_nbr5, const MatchParams& mp, const BatchView& bv, uint32_t n_hyp, hipStream_t s);
void launch_lm_step(int slot, DevState* st, const LmSums* d_sums, int32_t* d_hist, const EvalParams& ep, hipStream_t s);
// peer exchange self-test: every rank writes a tagged chunk into every inbox and waits (<= 2 s) for all of them in its own; *d_ok = 1 on success
void launch_peer_selftest(void* const inbox[8], int rank, int world, uint32_t tag, int32_t* d_ok, hipStream_t s);
// Seam B
void launch_knn_only(const float* d_q_xyz, uint32_t nq, int k, const DevMapView& map, float gate_d2, float* d_nbr,
                     float* d_d2, int32_t* d_idx, uint8_t* d_found, uint32_t* d_fallback_list,
                     uint32_t* d_fallback_count, hipStream_t s);
void launch_knn_fallback(const float* d_q_xyz, const uint32_t* d_fallback_list, uint32_t n_fallback, int k,
                         const DevMapView& map, float* d_nbr, float* d_d2, int32_t* d_idx, hipStream_t s);

}  // namespace soicp
