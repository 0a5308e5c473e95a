// icp_context.cpp -- host driver behind the C ABI of include/so_icp.h.
//
// Mirrors, on top of the HIP kernels, the control flow of
//   LidarSLAM::Localization / performLocalizationAndMapping   src/LidarProcess/LidarSlam.cpp:30-51, 107-210
// (paths relative to /root/reference/super_odometry/).  There is NO CPU fallback: without a usable
// HIP device so_icp_create() fails and says so.
#include <dlfcn.h>
#include <unistd.h>
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <climits>
#include <condition_variable>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "../../include/so_icp.h"
#include "kernels.h"
#include "lm_solver.h"
#include "deskew_math.h"
#include "device_map.h"
#include "local_map.h"
#include "so_math.h"

using namespace soicp;

static_assert(sizeof(so_icp_sums) == sizeof(LmSums), "so_icp_sums must mirror LmSums");
static_assert(sizeof(LmState) <= sizeof(so_icp_lm_state), "so_icp_lm_state too small");
static_assert(sizeof(LmSums) == 45 * sizeof(double), "LmSums is 45 doubles");

namespace {

thread_local std::string g_create_error;

struct DevBuf {
  void* p = nullptr;
  size_t cap = 0;
  hipError_t reserve(size_t bytes) {
    if (bytes <= cap) return hipSuccess;
    if (p) (void)hipFree(p);
    p = nullptr; cap = 0;
    size_t want = bytes + bytes / 4 + 256;
    hipError_t e = hipMalloc(&p, want);
    if (e == hipSuccess) cap = want;
    return e;
  }
  void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
  template <typename T> T* as() const { return reinterpret_cast<T*>(p); }
};

// RCCL entry points: prototypes and types come from <rccl/rccl.h>; the library itself is resolved lazily with dlopen
// (a single-GPU process never loads librccl -- one collective per evaluation is the only use)
struct Rccl {
  void* lib = nullptr;
  decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
  decltype(&ncclCommInitRank) CommInitRank = nullptr;
  decltype(&ncclAllReduce) AllReduce = nullptr;
  decltype(&ncclAllGather) AllGather = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
};
static_assert(sizeof(ncclUniqueId) == SO_ICP_UNIQUE_ID_BYTES, "SO_ICP_UNIQUE_ID_BYTES must equal sizeof(ncclUniqueId)");

bool rccl_load(Rccl& r, std::string& err) {
  if (r.lib) return true;
  const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  for (const char* n : names) { r.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL); if (r.lib) break; }
  if (!r.lib) { err = std::string("dlopen(librccl) failed: ") + dlerror(); return false; }
  r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(dlsym(r.lib, "ncclGetUniqueId"));
  r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(dlsym(r.lib, "ncclCommInitRank"));
  r.AllReduce = reinterpret_cast<decltype(r.AllReduce)>(dlsym(r.lib, "ncclAllReduce"));
  r.AllGather = reinterpret_cast<decltype(r.AllGather)>(dlsym(r.lib, "ncclAllGather"));
  r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(dlsym(r.lib, "ncclCommDestroy"));
  r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(dlsym(r.lib, "ncclGetErrorString"));
  if (!r.GetUniqueId || !r.CommInitRank || !r.AllReduce || !r.AllGather || !r.CommDestroy) { err = "librccl lacks a required symbol"; return false; }
  return true;
}

struct EventSpan { int kind; hipEvent_t a, b; uint32_t units; };  // kind 0 knn, 1 eval, 2 prep

// In-process shard group (so_icp_comm_init_inprocess): the contexts of ONE process that share a key -- one thread and one
// context per GPU, or several shard contexts on one GPU in a test -- sum their 45-double records through host memory.
// Fixed order (rank 0, 1, ...): every member receives bit-identical sums and takes identical controller decisions.
struct InprocGroup {
  std::mutex mu; std::condition_variable cv;
  int world = 0, arrived = 0, members = 0; unsigned long long generation = 0;
  // A round is identified by (kind, size): members that disagree about what is being summed -- one in the LmSums reduce, another
  // in the per-cube counts after a failed insert -- must not be paired silently; a member that returns early (a failed HIP call
  // before the exchange) or never arrives (wait_seconds) makes the round fail on every member instead of blocking the others for ever.
  int round_kind = -1; size_t round_size = 0; bool aborted = false;
  int wait_seconds = 60;  // patience with a member that has not arrived (first-call code-object load, a debugger): SOICP_GROUP_TIMEOUT_S
  std::vector<LmSums> slot; LmSums total{};
  std::vector<std::vector<int32_t>> islot; std::vector<int32_t> itotal;
  void abort_all() { std::lock_guard<std::mutex> lk(mu); aborted = true; cv.notify_all(); }
  // returns false when the round failed (mismatch, abort, or a member missing for wait_seconds): the group is unusable afterwards
  template <class Publish, class Combine>
  bool round(int kind, size_t size, Publish&& publish, Combine&& combine) {
    std::unique_lock<std::mutex> lk(mu);
    if (aborted) return false;
    if (arrived == 0) { round_kind = kind; round_size = size; }
    else if (round_kind != kind || round_size != size) { aborted = true; cv.notify_all(); return false; }
    publish();
    if (++arrived == world) {
      combine();
      arrived = 0; ++generation;
      cv.notify_all();
      return true;
    }
    const unsigned long long g = generation;
    const bool done = cv.wait_for(lk, std::chrono::seconds(wait_seconds), [&] { return generation != g || aborted; });
    if (!done || aborted) { aborted = true; cv.notify_all(); return false; }
    return true;
  }
  bool allreduce(int rank, LmSums* io) {
    const bool ok = round(0, sizeof(LmSums), [&] { slot[(size_t)rank] = *io; }, [&] {
      double* t = reinterpret_cast<double*>(&total);
      for (size_t k = 0; k < sizeof(LmSums) / sizeof(double); ++k) {
        double acc = 0;
        for (int r = 0; r < world; ++r) acc += reinterpret_cast<const double*>(&slot[(size_t)r])[k];  // fixed order: rank 0, 1, ...
        t[k] = acc;
      }
    });
    if (ok) { std::lock_guard<std::mutex> lk(mu); *io = total; }
    return ok;
  }
  // same for a vector of counters (per-cube point counts after a map insert)
  bool allreduce_i32(int rank, std::vector<int32_t>& io) {
    const bool ok = round(1, io.size(), [&] {
      if (islot.size() != (size_t)world) islot.resize((size_t)world);
      islot[(size_t)rank] = io;
    }, [&] {
      itotal.assign(io.size(), 0);
      for (int r = 0; r < world; ++r)
        for (size_t k = 0; k < io.size() && k < islot[(size_t)r].size(); ++k) itotal[k] += islot[(size_t)r][k];
    });
    if (ok) { std::lock_guard<std::mutex> lk(mu); io = itotal; }
    return ok;
  }
  // all-gather of byte strings of any length (the shards' points at a planeRes change); same discipline as above: the last
  // member to arrive assembles the result, nobody's slot is read after the round
  std::vector<std::vector<uint8_t>> bslot, ball;
  bool allgather_bytes(int rank, const std::vector<uint8_t>& mine, std::vector<std::vector<uint8_t>>& all) {
    const bool ok = round(2, 0, [&] {
      if (bslot.size() != (size_t)world) bslot.resize((size_t)world);
      bslot[(size_t)rank] = mine;
    }, [&] { ball = bslot; });
    if (ok) { std::lock_guard<std::mutex> lk(mu); all = ball; }
    return ok;
  }
};
std::mutex g_groups_mu;
std::vector<std::pair<uint64_t, std::shared_ptr<InprocGroup>>> g_groups;

}  // namespace

struct so_icp_ctx {
  so_icp_config cfg;
  std::string err;
  bool host_only = false;  // device_id < 0: LocalMap bookkeeping only, every compute entry point fails
  LocalMap map;                     // host LocalMap: host-only contexts and sharded (world_size > 1) contexts
  std::unique_ptr<DeviceMap> dmap;  // HBM-resident LocalMap with GPU insert (world_size == 1)
  DevBuf d_world;                   // world-frame copy of the scan for the map insert
  CanonicalMap cm;
  uint64_t uploaded_version = 0;
  hipStream_t stream = nullptr;
  // map shard in HBM
  DevBuf d_mpts, d_cell_start, d_cube_slot;
  DevMapView view{};
  // scan / correspondence buffers
  DevBuf d_scan_own, d_keys0, d_vals0, d_chunks, d_binned, d_nd, d_coeff, d_status, d_nbr5;
  DevBuf d_kdbg;   // profiling only
  DevBuf d_counts; // sharded device map: per-cube counters on their way through the all-reduce
  DevBuf d_small;  // hist[16] int32 | ticket | n_kept | fb_count | LmSums | partials
  int32_t* d_hist = nullptr; uint32_t* d_ticket = nullptr; uint32_t* d_nkept = nullptr; uint32_t* d_fbcount = nullptr;
  LmSums* d_sums = nullptr; double* d_partials = nullptr;
  LmSums* h_sums = nullptr; uint32_t* h_u32 = nullptr;  // pinned
  DevBuf d_state_buf; DevState* d_state = nullptr; DevState* h_state = nullptr;  // device-resident registration state + the pinned mirror read last
  // per-outer-iteration read-backs (double-buffered); mirrors 2, 3: the second pair of so_icp_register_sequence, whose chained
  // registrations alternate between the pairs (the next registration starts reporting before the host has read the last report of this one)
  DevState* h_ring[4] = {nullptr, nullptr, nullptr, nullptr}; hipEvent_t ev_outer[2] = {nullptr, nullptr};
  DevState* d_ring[4] = {nullptr, nullptr, nullptr, nullptr};  // device-side addresses of the pinned mirrors
  bool direct_readback = true; unsigned long long reg_counter = 0;
  bool persistent_solve = true;  // SOICP_PERSISTENT=0: one launch per evaluation
  unsigned long long solve_launches = 0;  // persistent solve launches so far (EvalParams::epoch_base)
  DevBuf d_bin_key, d_bin_cnt, d_bin_off; uint32_t bin_log2 = 0; bool bin_dirty = true;
  int32_t* h_hist = nullptr;  // pinned: per-outer-iteration copy of the histogram replicas (profiling mode)
  std::vector<DevBuf> resident_scans;  // so_icp_upload_scan
  // so_icp_prefilter_announce: the NEXT raw cloud, already on its way to HBM (pf_stage) when so_icp_prefilter_scan is called with the same buffer
  DevBuf pf_stage; std::mutex pf_mu, aux_mu;
  struct PfAnnounced { const void* ptr = nullptr; size_t n = 0, stride = 0; bool on = false; } pf_announced;
  DevBuf pf_in, pf_out, pf_small, pf_w, pf_s, pf_k0, pf_k1, pf_v0, pf_v1, pf_flags, pf_pos, pf_heads, pf_temp;  // so_icp_prefilter_scan
  DevBuf pf_dec;                      // {counters[16], VgDecision, partial statistics}: the pre-filter decided on the device
  VgDecision* h_pf = nullptr;         // pinned read-back of the decision
  uint32_t* h_pf_kept = nullptr;      // pinned: so_icp_transform_cloud's count of kept points
  size_t pf_temp_for = 0, pf_temp_need = 0;  // map_sort_temp_bytes(pf_temp_for) == pf_temp_need (the query costs two library calls)
  bool pf_fast = true;                // SOICP_PREFILTER_FAST=0: statistics read back, decided on the host, then the filter (rounds 1-3)
  hipEvent_t ev_upload = nullptr;     // a scan uploaded through the auxiliary queue: the context's queue waits for it
  hipStream_t pf_stream = nullptr;    // the pre-filter's own queue: the next frame's upload + VoxelGrid run BESIDE the map insert the previous
  // Seam B scratch
  DevBuf d_q, d_nbr, d_d2, d_idx, d_found, d_fblist;
  // persistent LidarSLAM state
  int32_t prev_obs_hist[SO_ICP_N_OBS]{};
  bool have_hist = false;
  int last_pos[3] = {0, 0, 0};
  // so_icp_register_batch: worker contexts register hypotheses concurrently against the PARENT's resident map
  struct Borrow { bool on = false; DevMapView view{}; float plane_res = 0; int pos[3] = {0, 0, 0}; int count_5x5 = 0; } borrow;
  std::vector<so_icp_ctx*> workers;
  // so_icp_register_batch, batched kernels: one set of per-registration arrays per hypothesis (common element stride bs)
  struct BatchBufs {
    uint32_t cap_hyp = 0, bs = 0, table_log2 = 0;
    bool tables_clean = false;
    DevBuf states, begin, active, qslot, qrank, binned, chunks, status, nbr5, nd, coeff, bin_key, bin_cnt, bin_off, partials, sync, hist;
    DevState* h_states = nullptr; RegBeginArgs* h_begin = nullptr; uint32_t* h_active = nullptr;  // pinned
    void release() {
      for (DevBuf* b : {&states, &begin, &active, &qslot, &qrank, &binned, &chunks, &status, &nbr5, &nd, &coeff, &bin_key, &bin_cnt,
                        &bin_off, &partials, &sync, &hist}) b->release();
      if (h_states) (void)hipHostFree(h_states);
      if (h_begin) (void)hipHostFree(h_begin);
      if (h_active) (void)hipHostFree(h_active);
      h_states = nullptr; h_begin = nullptr; h_active = nullptr; cap_hyp = 0; bs = 0; table_log2 = 0; tables_clean = false;
    }
  } batch;
  int batch_degrade = 0;  // 0: two solve workgroups per compute unit, 1: one (after a batched solve that was not co-resident, or
                          // SOICP_BATCH_MODE=one_per_cu), 2: lanes = concurrent sequential registrations (SOICP_BATCH_MODE=lanes)
  bool no_map_shift_once = false;  // retry of a registration: keep the window of the first attempt
  int n_cus = 256;            // compute units of the device: upper bound of the persistent solve launch's workgroups
  int ablate = 0;             // SOICP_ABLATE (profiling / test switches), read at creation
  bool speculate = true;      // enqueue outer iteration i+1 before the report of i is in (SOICP_SPECULATE=0: wait first)
  // (round 5: no event and no stream query accompanies the host's wait for a report in the normal case.  An event record is a
  //  marker packet, and so is what hipStreamQuery enqueues to learn whether the queue has drained: either one landed between the
  //  speculated k-NN sweep and the solve launch enqueued behind it, where the command processor spent ~6 us on it -- the gap
  //  every kernel trace of rounds 3-5 shows in front of the second solve.  The watchdog of that wait is now the clock: the queue
  //  is queried only after kReportWatchdogMs without a report.)
  bool batch_mode = false;    // no kernel timing, tracker state read-only
  bool batch_single = false;  // batch on ONE lane: nothing runs next to it, the persistent solve launch is safe
  bool no_map_shift = false;  // so_icp_register_batch: hypotheses after the first keep the window of the first
  int startup_count = 0;
  double last_time = 0;
  // timing
  std::vector<hipEvent_t> ev_pool; size_t ev_used = 0;
  std::vector<EventSpan> spans;
  bool span_open = false;
  so_icp_timing timing{};
  // RCCL
  Rccl rccl; ncclComm_t comm = nullptr;
  std::shared_ptr<InprocGroup> group;  // so_icp_comm_init_inprocess
  // peer exchange (so_icp_peer_export / _connect / _enable): tagged-chunk push between the ranks' persistent solve launches
  void* peer_own = nullptr;                 // this rank's inbox (uncached / fine-grained device memory)
  void* peer_inbox[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  bool peer_opened[8] = {false, false, false, false, false, false, false, false};  // mapped with hipIpcOpenMemHandle (to be closed)
  bool peer_connected = false, peer_on = false;
  unsigned peer_connects = 0;  // handshakes so far (tag of the self-test chunks)
  // so_icp_stage_scan: the NEXT scans travel to HBM while the current registration runs -- straight from the caller's buffer
  // when that is registered (pinned) host memory (so_icp_host_register: the announcing thread enqueues the DMA on the copy
  // stream and returns; the registration's first kernel waits for it ON THE DEVICE), else through a copy thread that packs
  // the cloud into a pinned buffer first.
  static constexpr int kStageSlots = 3;  // one in use by the registration in flight + two announced ahead
  struct StageSlot {
    const float* src = nullptr; size_t n = 0, stride = 0;  // identity of the staged host buffer
    DevBuf dev; float* pinned = nullptr; size_t pinned_cap = 0;
    int state = 0;  // 0 empty, 1 queued (the copy thread owns it), 2 ready, 3 in use by the registration in flight, -1 failed
    unsigned long long seq = 0;          // announcement number (newer scans have larger ones)
    hipEvent_t ev = nullptr;             // direct path: end of the H2D copy on the copy stream
    bool ev_pending = false;             //   ... which may still be reading the caller's buffer
    bool deferred = false;               // direct path: announced, the copy is not enqueued yet (see stage_issue_deferred)
    hipStream_t tail_stream = nullptr;   // direct path: the copy is enqueued there, what follows it (binning ahead, `ev`) not yet -- see stage_tail
    std::chrono::steady_clock::time_point t_announced;
    std::string err;
    // binned ahead (stage_prebin): the scan's work list, built on the copy queue behind the copy while the registration before it runs
    DevBuf pb_keys, pb_vals, pb_chunks, pb_binned, pb_ctr;
    bool prebinned = false; uint32_t pb_chunk_cap = 0;
  } stage[kStageSlots];
  // Binning ahead (round 5).  A scan announced with so_icp_stage_scan is hash-binned on the copy queue right behind its DMA, under the
  // guess of the registration that enqueues the copy (the latest pose this context knows), so that its own registration starts
  // with the k-NN sweep: scan_keys -> bin_offsets -> bin_place (three dependent launches, ~21 us of a 150 us registration) leave
  // the registration's critical path and run beside the previous registration's solve, which keeps one wavefront per SIMD busy.
  // Chunks binned under a pose one frame old stay spatially compact under the scan's own guess -- the k-NN kernel forms every
  // chunk's candidate block from the queries' actual positions, as it does for the second sweep of any registration; results
  // do not depend on the binning (exact per query, sums in scan order).  Single device, device-resident map only; SOICP_PREBIN=0
  // switches it off.
  bool prebin = true;
  DevBuf d_pbin_key, d_pbin_cnt, d_pbin_off; uint32_t pbin_log2 = 0;
  StageSlot* stage_in_use = nullptr;  // the slot the current registration reads (released when the call returns)
  unsigned long long stage_seq = 0, stage_consumed_seq = 0;  // announcements so far / announcement number of the scan consumed last
  bool stage_quit = false, stage_started = false;
  std::atomic<int> stage_pending{0};      // queued slots the copy thread has not picked up yet
  std::atomic<bool> stage_parked{false};  // the copy thread sleeps on stage_cv (it spins for a while after every job first)
  std::atomic<bool> stage_timed{false};   // ... in the TIMED wait for a DMA-staged scan's 300 us: it looks at the slots again by itself when that
                                          // runs out, so another DMA announcement need not wake it (a futex call on the announcing thread's path)
  std::thread stage_thread; std::mutex stage_mu; std::condition_variable stage_cv;
  struct HostRange { const char* p; size_t bytes; bool owned; };
  std::vector<HostRange> host_ranges;     // so_icp_host_register / so_icp_host_alloc (under stage_mu)
  // so_icp_register_sequence: a copy / binning queue and three scan slots of its own (nothing shared with so_icp_stage_scan's
  // slots and thread), the iterations pre-enqueued per registration, DevState::done_count as of the last report
  hipStream_t seq_stream = nullptr;
  StageSlot seq_slot[kStageSlots];
  DevBuf d_sbin_key, d_sbin_cnt, d_sbin_off; uint32_t sbin_log2 = 0;
  int seq_depth = 2; uint32_t done_count_seen = 0;
  // so_icp_sequence_announce_next: the scan that will START the next so_icp_register_sequence call -- copied and binned beside the LAST
  // registration of the current call (under that registration's guess o delta), adopted by the next call when its scans[0] is this buffer
  struct SeqNext { const void* next_scan = nullptr; size_t next_n = 0; double delta[7] = {0, 0, 0, 0, 0, 0, 1}; bool announced = false;  // for the coming call to stage
                   const void* scan = nullptr; size_t n = 0;                                                                             // staged by the last call
                   bool staged = false; int slot = 0; bool binned = false, needs_event = false; const float* d_scan = nullptr; } seq_next;
  bool seq_chain = true;                  // SOICP_SEQ_CHAIN=0: so_icp_register_sequence runs one registration after the other (same results)
  bool query_waves = true;                // SOICP_QUERY_WAVES=0: a small scan (<= 4 096 kept queries) is binned and swept in chunks like a large one
  bool knn_pack = true;                   // SOICP_KNN_PACK=0: one chunk per wavefront throughout (round-3 work list)
  bool knn_list_fits = false;             // the last registration's work list (normal + light chunks) fitted the k-NN grid one chunk per wavefront:
                                          // packing four light chunks into a wavefront then only lengthens the longest wavefronts (a 13 k-point
                                          // voxel-filtered scan: sweeps 20.5 + 18.6 -> 17.2 + 16.9 us unpacked)
  uint32_t packed_leftover_seen = 0;      // DevState::packed_leftover (a running count) as of the last report
  int knn_pack_hold = 0;                  // registrations left without packing after one in which the packed near pass left > 3 % of
                                          // the queries to the exact per-lane scan (sparse map, far-off guess): then it is not a saving
  static constexpr int kBatchRoundsTracked = 16;
  float batch_survivors[kBatchRoundsTracked] = {};  // so_icp_register_batch: share of round r's list still active after it, last batch (chaining of rounds)
  bool batch_chain = true;                // (SOICP_BATCH_CHAIN=0: report + synchronisation after every round, as in round 3)
  hipStream_t copy_stream = nullptr;
  bool retried = false;       // the current registration is the repeat of an abandoned one
  unsigned long long peer_timeout_ticks = 100000000ull;  // 1 s at 100 MHz: patience of a solve launch with the peer exchange (SOICP_PEER_TIMEOUT_MS)
  bool scan_staged = false;   // the scan of the current registration came from a stage slot
  bool query_split = false;   // world_size > 1, SO_ICP_SHARD_QUERIES: map replicated, the scan's 64-point segments dealt to the ranks
  DevBuf d_sub;               //   this rank's share of the current scan, gathered

  ~so_icp_ctx();
};

namespace {

#define HIP_TRY(ctx, expr)                                                                         \
  do {                                                                                             \
    hipError_t e__ = (expr);                                                                       \
    if (e__ != hipSuccess) {                                                                       \
      (ctx)->err = std::string(#expr) + ": " + hipGetErrorString(e__);                             \
      return SO_ICP_E_HIP;                                                                         \
    }                                                                                              \
  } while (0)

int fail(so_icp_ctx* c, int code, const std::string& msg) { c->err = msg; return code; }
#define NEED_DEVICE(c)                                                                                        \
  do {                                                                                                        \
    if ((c)->host_only)                                                                                       \
      return fail((c), SO_ICP_E_HIP, "host-only context (device_id < 0): no compute path -- libsoicp has no CPU fallback"); \
  } while (0)

hipEvent_t next_event(so_icp_ctx* c) {
  if (c->ev_used == c->ev_pool.size()) {
    hipEvent_t e;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    c->ev_pool.push_back(e);
  }
  return c->ev_pool[c->ev_used++];
}
// time_kernels: 1 = bracket only the dominant (k-NN) kernel -- cheap enough to stay on inside a timed region;
//               2 = bracket every kernel (events cost a few microseconds of pipeline bubble each)
void span_begin(so_icp_ctx* c, int kind, uint32_t units) {
  if (c->batch_mode || !c->cfg.time_kernels || (c->cfg.time_kernels == 1 && kind != 0)) return;
  EventSpan s{kind, next_event(c), next_event(c), units};
  if (!s.a || !s.b) return;
  (void)hipEventRecord(s.a, c->stream);
  c->spans.push_back(s);
  c->span_open = true;
}
void span_end(so_icp_ctx* c) {
  if (!c->span_open || c->spans.empty()) return;
  c->span_open = false;
  (void)hipEventRecord(c->spans.back().b, c->stream);
}
void spans_collect(so_icp_ctx* c) {  // stream must be idle
  for (const EventSpan& s : c->spans) {
    float ms = 0;
    if (hipEventElapsedTime(&ms, s.a, s.b) != hipSuccess) continue;
    if (s.kind == 0) { c->timing.knn_ms_total += ms; c->timing.knn_launches++; c->timing.knn_queries += s.units; c->timing.knn_map_points += c->view.n_points; }
    else if (s.kind == 1) { c->timing.eval_ms_total += ms; c->timing.eval_launches++; c->timing.eval_points += s.units; }
    else { c->timing.prep_ms_total += ms; c->timing.prep_launches++; }
  }
  c->spans.clear();
  c->ev_used = 0;
}

// Sharded device map: every rank has inserted the same cloud into its shard; the per-cube point counts of the FULL map
// (get5x5LocalMapFeatureSize, LocalMap.h:292-318, and the <= 50 check of LidarSlam.cpp:113-116 read them) are the sums of
// the ranks' owned counts -- one small collective per insert (per scan), never per registration.
int exchange_map_counts(so_icp_ctx* c) {
  if (!c->dmap || !c->dmap->sharded()) return SO_ICP_OK;
  std::vector<int32_t> v;
  c->dmap->owned_counts(v);
  if (c->group) {
    if (!c->group->allreduce_i32(c->cfg.rank, v))
      return fail(c, SO_ICP_E_RCCL, "in-process group: the map-count exchange failed (a member returned early, is in another exchange, or did not arrive)");
  } else if (c->comm) {
    HIP_TRY(c, c->d_counts.reserve(v.size() * sizeof(int32_t)));
    HIP_TRY(c, hipMemcpyAsync(c->d_counts.p, v.data(), v.size() * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
    const ncclResult_t nrc = c->rccl.AllReduce(c->d_counts.p, c->d_counts.p, v.size(), ncclInt32, ncclSum, c->comm, c->stream);
    if (nrc != ncclSuccess) return fail(c, SO_ICP_E_RCCL, std::string("ncclAllReduce(map counts): ") + (c->rccl.GetErrorString ? c->rccl.GetErrorString(nrc) : "?"));
    HIP_TRY(c, hipMemcpyAsync(v.data(), c->d_counts.p, v.size() * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
  }  // (a shard context without a communicator -- tests that drive the ranks one by one -- reports its own owned counts)
  c->dmap->set_full_counts(v);
  return SO_ICP_OK;
}

// Sharded device map, planeRes change: the shards are cut along the cell grid, which follows planeRes.  Every rank hands out
// the points it owns, all ranks gather all of them (in-process group, or RCCL all-gather of the padded byte strings), and each
// re-cuts its shard on the new grid (DeviceMap::reshard).  Collective: every rank must make the same so_icp_set_resolution call.
int reshard_for_resolution(so_icp_ctx* c, float line_res, float plane_res) {
  std::vector<uint8_t> mine;
  if (c->dmap->export_owned(mine, c->err) < 0) return SO_ICP_E_HIP;
  std::vector<std::vector<uint8_t>> all;
  if (c->group) {
    if (!c->group->allgather_bytes(c->cfg.rank, mine, all))
      return fail(c, SO_ICP_E_RCCL, "in-process group: the exchange of the shards' points failed (a member returned early, is in another exchange, or did not arrive)");
  } else {
    const int W = c->cfg.world_size;
    auto nccl_fail = [&](const char* what, ncclResult_t r) { return fail(c, SO_ICP_E_RCCL, std::string(what) + ": " + (c->rccl.GetErrorString ? c->rccl.GetErrorString(r) : "?")); };
    // lengths first, then the strings padded to the longest
    std::vector<unsigned long long> len((size_t)W, 0ull);
    len[(size_t)c->cfg.rank] = mine.size();
    HIP_TRY(c, c->d_counts.reserve((size_t)W * sizeof(unsigned long long)));
    HIP_TRY(c, hipMemcpyAsync(c->d_counts.p, len.data(), (size_t)W * sizeof(unsigned long long), hipMemcpyHostToDevice, c->stream));
    ncclResult_t r = c->rccl.AllGather(c->d_counts.as<unsigned long long>() + c->cfg.rank, c->d_counts.p, 1, ncclUint64, c->comm, c->stream);
    if (r != ncclSuccess) return nccl_fail("ncclAllGather(shard sizes)", r);
    HIP_TRY(c, hipMemcpyAsync(len.data(), c->d_counts.p, (size_t)W * sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    size_t longest = 16;
    for (unsigned long long v : len) longest = std::max(longest, (size_t)v);
    longest = (longest + 15) & ~(size_t)15;
    DevBuf send, recv;
    std::vector<uint8_t> host(longest * (size_t)W);
    hipError_t e = send.reserve(longest);
    if (e == hipSuccess) e = recv.reserve(longest * (size_t)W);
    if (e == hipSuccess && !mine.empty()) e = hipMemcpyAsync(send.p, mine.data(), mine.size(), hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) {
      r = c->rccl.AllGather(send.p, recv.p, longest, ncclUint8, c->comm, c->stream);
      if (r != ncclSuccess) { send.release(); recv.release(); return nccl_fail("ncclAllGather(shard points)", r); }
      e = hipMemcpyAsync(host.data(), recv.p, host.size(), hipMemcpyDeviceToHost, c->stream);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    send.release(); recv.release();
    HIP_TRY(c, e);
    all.resize((size_t)W);
    for (int k = 0; k < W; ++k) all[(size_t)k].assign(host.begin() + (long)(longest * (size_t)k), host.begin() + (long)(longest * (size_t)k + (size_t)len[(size_t)k]));
  }
  const int rc = c->dmap->reshard(all, line_res, plane_res, c->err);
  if (rc < 0) return rc == -1 ? SO_ICP_E_NOMEM : SO_ICP_E_HIP;
  c->uploaded_version = 0;
  return SO_ICP_OK;
}

float map_plane_res(const so_icp_ctx* c) { return c->dmap ? c->dmap->plane_res() : c->map.plane_res(); }
void map_shift(so_icp_ctx* c, const double t[3], int pos[3]) { if (c->dmap) c->dmap->shift(t, pos); else c->map.shift(t, pos); }
int map_count_5x5(const so_icp_ctx* c, const int pos[3]) { return c->dmap ? c->dmap->count_5x5(pos) : c->map.count_5x5(pos); }
const int* map_origin(const so_icp_ctx* c) { return c->dmap ? c->dmap->origin() : c->map.origin(); }

int upload_map(so_icp_ctx* c) {
  if (c->dmap) { const int rv = c->dmap->view(c->view, c->err); return rv == 0 ? SO_ICP_OK : (rv == -1 ? SO_ICP_E_NOMEM : SO_ICP_E_HIP); }
  if (c->uploaded_version == c->map.version()) return SO_ICP_OK;
  c->map.build_canonical(c->query_split ? 0 : c->cfg.rank, c->query_split ? 1 : c->cfg.world_size, c->cm);
  const CanonicalMap& m = c->cm;
  const size_t n = m.n_points();
  HIP_TRY(c, c->d_mpts.reserve((n + 16) * 16));
  HIP_TRY(c, c->d_cell_start.reserve((m.cell_start.size() + 1) * 4));
  HIP_TRY(c, c->d_cube_slot.reserve(kMapNum * 4));
  if (n) {
    HIP_TRY(c, hipMemcpyAsync(c->d_mpts.p, m.xyzw.data(), n * 16, hipMemcpyHostToDevice, c->stream));
  }
  if (!m.cell_start.empty())
    HIP_TRY(c, hipMemcpyAsync(c->d_cell_start.p, m.cell_start.data(), m.cell_start.size() * 4, hipMemcpyHostToDevice, c->stream));
  HIP_TRY(c, hipMemcpyAsync(c->d_cube_slot.p, m.cube_slot.data(), kMapNum * 4, hipMemcpyHostToDevice, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));  // host vectors may be rebuilt right after
  DevMapView& v = c->view;
  v.pts = c->d_mpts.as<float4>();
  v.cell_start = c->d_cell_start.as<uint32_t>(); v.cube_slot = c->d_cube_slot.as<int32_t>();
  v.nc = m.nc; v.ncell1 = (uint32_t)((size_t)m.nc * m.nc * m.nc + 1); v.inv_cell = 1.0 / m.cell;
  v.origin[0] = c->map.origin()[0]; v.origin[1] = c->map.origin()[1]; v.origin[2] = c->map.origin()[2];
  v.n_points = (uint32_t)n;
  v.n_slots = (uint32_t)m.n_slots;
  c->uploaded_version = c->map.version();
  return SO_ICP_OK;
}

int reserve_scan_buffers(so_icp_ctx* c, size_t n) {
  const size_t m = n + 256;
  HIP_TRY(c, c->d_keys0.reserve(m * 4));
  HIP_TRY(c, c->d_vals0.reserve(m * 4)); HIP_TRY(c, c->d_chunks.reserve(m * 4));
  HIP_TRY(c, c->d_binned.reserve(m * 16));
  HIP_TRY(c, c->d_nd.reserve(m * 32)); HIP_TRY(c, c->d_coeff.reserve(m * 8)); HIP_TRY(c, c->d_status.reserve(m));
  HIP_TRY(c, c->d_nbr5.reserve(m * 20));
  return SO_ICP_OK;
}

MatchParams match_params(float plane_res, int ablate) {
  MatchParams mp;
  mp.plane_res = plane_res;
  mp.sq_max_dist_f = 3 * plane_res;           // float product (LidarSlam.cpp:526)
  mp.max_point_dist = (double)plane_res / 2.0; // LidarSlam.cpp:820
  mp.ablate = ablate;  // SOICP_ABLATE, read when the context is created (a getenv per registration is a walk over environ)
  mp.kdbg = nullptr;
  mp.skip_near_pass = 0;
  mp.pack_light = 1;
  mp.packed_leftover = nullptr;
  mp.begin = 0; mp.begin_max_surface_features = -1; mp.begin_n = 0;
  mp.begin_args = RegBeginArgs{};
  mp.begin_ctr = nullptr; mp.begin_state = nullptr;
  mp.chain_expect = 0;
  return mp;
}
EvalParams eval_params(float plane_res, int variant, int ablate) {
  EvalParams ep;
  const double a = (double)sqrtf(3 * plane_res);  // std::sqrt(float) then TukeyLoss(double a) (LidarSlam.cpp:271)
  ep.a2 = a * a;
  ep.variant = variant;
  ep.ablate = ablate;
  ep.hring[0] = ep.hring[1] = nullptr;
  ep.seq_base = 0;
  ep.n_queries = 0; ep.q_stride = 1;
  ep.defer_publish = 0;
  ep.chain_expect = 0; ep.chain_next = 0;
  for (double& d : ep.chain_delta) d = 0;
  ep.epoch_base = 0;
  for (void*& p : ep.peer_inbox) p = nullptr;
  ep.peer_rank = 0; ep.peer_world = 0;
  ep.timeout_ticks = 5000000ull;  // 50 ms
  return ep;
}

// LidarSLAM::EstimateLidarUncertainty, LidarSlam.cpp:915-964
void uncertainty_from_hist(const int32_t* H, double u[6]) {
  const double tt = (double)H[6] + H[7] + H[8];
  const double tr = (double)H[0] + H[1] + H[2] + H[3] + H[4] + H[5];
  if (tt == 0 || tr == 0) { for (int i = 0; i < 6; ++i) u[i] = 0; return; }
  u[0] = std::fmin(H[6] / tt * 3, 1.0); u[1] = std::fmin(H[7] / tt * 3, 1.0); u[2] = std::fmin(H[8] / tt * 3, 1.0);
  u[3] = std::fmin((H[0] + H[1]) / tr * 3, 1.0); u[4] = std::fmin((H[2] + H[3]) / tr * 3, 1.0); u[5] = std::fmin((H[4] + H[5]) / tr * 3, 1.0);
}

// LidarSLAM::MannualYawCorrection, LidarSlam.cpp:891-913; tf2::Matrix3x3::getRPY and tf2::Quaternion::setRPY
// [UPSTREAM tf2] written out.
void yaw_correction(double T[7], const double last[7], double yaw_ratio) {
  double tn, rn;
  relative_motion(last, T, tn, rn);
  const float translation_norm = (float)tn;
  const double x = T[3], y = T[4], z = T[5], w = T[6];
  const double d = x * x + y * y + z * z + w * w, s = 2.0 / d;
  const double xs = x * s, ys = y * s, zs = z * s, wx = w * xs, wy = w * ys, wz = w * zs;
  const double xx = x * xs, xy = x * ys, xz = x * zs, yy = y * ys, yz = y * zs, zz = z * zs;
  const double m00 = 1.0 - (yy + zz), m01 = xy - wz, m02 = xz + wy, m10 = xy + wz, m20 = xz - wy, m21 = yz + wx, m22 = 1.0 - (xx + yy);
  double roll, pitch, yaw;
  if (std::fabs(m20) >= 1) {
    yaw = 0;
    const double delta = std::atan2(m01, m02);
    pitch = (m20 < 0) ? M_PI / 2.0 : -M_PI / 2.0;
    roll = delta;
  } else {
    pitch = -std::asin(m20);
    roll = std::atan2(m21 / std::cos(pitch), m22 / std::cos(pitch));
    yaw = std::atan2(m10 / std::cos(pitch), m00 / std::cos(pitch));
  }
  const double cyaw = yaw + translation_norm * yaw_ratio * M_PI / 180;
  const double hy = cyaw * 0.5, hp = pitch * 0.5, hr = roll * 0.5;
  const double cy = std::cos(hy), sy = std::sin(hy), cp = std::cos(hp), sp = std::sin(hp), cr = std::cos(hr), sr = std::sin(hr);
  double q[4] = {sr * cp * cy - cr * sp * sy, cr * sp * cy + sr * cp * sy, cr * cp * sy - sr * sp * cy, cr * cp * cy + sr * sp * sy};
  const double n = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  for (int i = 0; i < 4; ++i) T[3 + i] = q[i] / n;
}

// The registration's results out of the state block the device published: pose (LidarSlam.cpp:135-136), per-iteration
// statistics (:242-251), final normal equations, post-processing (:155-157, 198-210).
void fill_result(so_icp_ctx* c, const DevState& H, const double pose_in[7], so_icp_stats* st, double pose_out[7], bool update_tracker) {
  double T[7];
  std::memcpy(T, H.T, sizeof(T));
  st->n_iterations = H.n_iterations;
  for (int it = 0; it < H.n_iterations && it < SO_ICP_MAX_OUTER; ++it) {
    so_icp_iter_stats& is = st->iterations[it];
    const DevIterStats& d = H.iters[it];
    is.translation_norm = d.translation_norm; is.rotation_norm = d.rotation_norm;
    is.num_surf_from_scan = d.num_surf; is.lm_iterations = d.lm_iterations; is.num_successful_steps = d.num_successful;
    is.termination = d.termination; is.initial_cost = d.initial_cost; is.final_cost = d.final_cost;
    std::memcpy(is.reject_hist, d.reject_hist, sizeof(is.reject_hist));
    std::memcpy(is.obs_hist, d.obs_hist, sizeof(is.obs_hist));
    std::memcpy(is.pose_after, d.pose_after, sizeof(is.pose_after));
  }
  if (H.n_iterations > 0 && update_tracker) {
    std::memcpy(c->prev_obs_hist, H.iters[H.n_iterations - 1].obs_hist, sizeof(c->prev_obs_hist));
    c->have_hist = true;
  }
  std::memcpy(st->JtJ, H.JtJ, sizeof(st->JtJ));
  std::memcpy(st->Jtr, H.Jtr, sizeof(st->Jtr));
  yaw_correction(T, pose_in, c->cfg.yaw_ratio);  // performPostOptimizationProcessing, :155-157 (last_T_w_lidar = the guess, :53-57)
  relative_motion(pose_in, T, st->total_translation, st->total_rotation);
  relative_motion(pose_in, T, st->translation_from_last, st->rotation_from_last);
  st->prediction_source = 0;
  std::memcpy(pose_out, T, sizeof(T));
}

// LidarSLAM::performLocalizationAndMapping (LidarSlam.cpp:107-152) with the loop state resident on the device:
// the host enqueues, per outer iteration, the STATIC sequence
//     clear histograms -> knn_plane -> [ eval(slot) -> (all-reduce) -> lm_step(slot) ] x (1 + lm_max)
// and every kernel consults DevState (reg_done / lm_more) to turn itself into a no-op once the controller has
// finished -- no host round trip per evaluation.  One small read-back per outer iteration tells the host when to stop
// enqueuing.
void stage_issue_deferred(so_icp_ctx* c, const double* prebin_pose);  // (so_icp_stage_scan machinery, below)
void stage_issue_deferred_copy(so_icp_ctx* c);
constexpr int kRetryWithoutPersistentSolve = -1000;  // internal: never leaves register_core
int register_core_once(so_icp_ctx* c, const float* d_scan, size_t n, const double pose_in[7], double pose_out[7], so_icp_stats* st) {
  const auto t_begin = std::chrono::steady_clock::now();
  so_icp_stats local;
  if (!st) st = &local;
  std::memset(st, 0, sizeof(*st));
  // (kernels.hip: bin_offsets_kernel / knn_plane_kernel)
  if (n >= ((size_t)1 << 21)) return fail(c, SO_ICP_E_UNSUPPORTED, "scan of 2^21 points or more: the work-list counters hold 21 bits each (chunk descriptors 26)");
  st->flags = (c->retried ? SO_ICP_FLAG_RETRIED : 0u) | (!c->dmap && !c->borrow.on ? SO_ICP_FLAG_HOST_MAP : 0u) |
              (c->cfg.world_size > 1 ? SO_ICP_FLAG_SHARDED : 0u) | (c->query_split ? SO_ICP_FLAG_QUERY_SPLIT : 0u) |
              (c->scan_staged ? SO_ICP_FLAG_STAGED_SCAN : 0u) | (c->direct_readback ? 0u : SO_ICP_FLAG_COPY_READBACK);
  double T[7];
  std::memcpy(T, pose_in, sizeof(T));  // LidarSlam.cpp:53-57 (T_w_initial_guess = last_T_w_lidar = T_w_lidar = the guess)
  std::memcpy(pose_out, pose_in, sizeof(T));
  if (c->have_hist) uncertainty_from_hist(c->prev_obs_hist, st->uncertainty);  // LidarSlam.cpp:47
  int pos[3];
  if (c->borrow.on) std::memcpy(pos, c->borrow.pos, sizeof(pos));  // window, count and map view were fixed by the batch driver
  else if (!c->no_map_shift && !c->no_map_shift_once) { map_shift(c, T, pos); std::memcpy(c->last_pos, pos, sizeof(pos)); }  // LidarSlam.cpp:363
  else std::memcpy(pos, c->last_pos, sizeof(pos));
  c->no_map_shift_once = false;
  st->pos_in_localmap[0] = pos[0]; st->pos_in_localmap[1] = pos[1]; st->pos_in_localmap[2] = pos[2];
  st->laser_cloud_surf_from_map_num = c->borrow.on ? c->borrow.count_5x5 : map_count_5x5(c, pos);  // LidarSlam.cpp:367
  st->laser_cloud_surf_stack_num = (int32_t)n;
  st->startup_count = c->startup_count;
  if (!(st->laser_cloud_surf_from_map_num > 50)) return SO_ICP_NOT_ENOUGH_MAP_FEATURES;  // LidarSlam.cpp:113-116
  int rc = SO_ICP_OK;
  if (c->borrow.on) c->view = c->borrow.view; else rc = upload_map(c);
  if (rc) return rc;
  // SO_ICP_SHARD_QUERIES: this rank registers ITS share of the scan -- the 64-point segments rank, rank + world, ... (a 128-ring
  // sweep in ring-major order gives every rank two 22.5-degree sectors of every ring: spatially compact, so its k-NN chunks are
  // as full as the whole scan's) -- gathered into one array by a strided device copy; from here on the registration is a
  // single-device one over n_own points, except that the sums of every evaluation are exchanged with the other ranks.
  const bool qsplit = c->query_split && !c->batch_mode && !c->borrow.on;
  const size_t n_total = n;
  if (qsplit) {
    const size_t W = (size_t)c->cfg.world_size, r = (size_t)c->cfg.rank, s_full = n / 64, tail = n % 64;
    const size_t own_full = s_full > r ? (s_full - r + W - 1) / W : 0;
    const bool own_tail = tail != 0 && (s_full % W) == r;  // (the partial last segment is segment number s_full)
    const size_t n_own = own_full * 64 + (own_tail ? tail : 0);
    HIP_TRY(c, c->d_sub.reserve((n_own + 64) * 12));
    if (own_full) HIP_TRY(c, hipMemcpy2DAsync(c->d_sub.p, 768, d_scan + r * 192, W * 768, 768, own_full, hipMemcpyDeviceToDevice, c->stream));
    if (own_tail) HIP_TRY(c, hipMemcpyAsync(c->d_sub.as<float>() + own_full * 192, d_scan + s_full * 192, tail * 12, hipMemcpyDeviceToDevice, c->stream));
    d_scan = c->d_sub.as<float>();
    n = n_own;
  }
  rc = reserve_scan_buffers(c, n);
  if (rc) return rc;
  const auto t_icp = std::chrono::steady_clock::now();  // TicToc t_opt, LidarSlam.cpp:118

  const int max_outer = std::min(c->cfg.max_iterations > 0 ? c->cfg.max_iterations : 4, SO_ICP_MAX_OUTER);
  const int lm_max = std::min(c->cfg.lm_max_iterations > 0 ? c->cfg.lm_max_iterations : 4, 16);
  DevState* ds = c->d_state;
  hipStream_t s = c->stream;
  // ---- once per registration: prologue (the guess and the loop bounds travel as kernel arguments, no H2D copy),
  //      sampling rule, spatial sort (locality survives the small pose updates), chunk list + gather
  span_begin(c, 2, (uint32_t)n);
  BinTable bt{nullptr, nullptr, nullptr, 0};
  // a scan that was binned ahead of this call (so_icp_stage_scan, so_icp_ctx::prebin): its work list is in the slot
  const bool may_prebin = c->prebin && c->dmap && c->cfg.world_size <= 1 && !c->batch_mode && !c->borrow.on && !qsplit;
  so_icp_ctx::StageSlot* pb = (may_prebin && n && c->scan_staged && c->stage_in_use && c->stage_in_use->prebinned && c->stage_in_use->n == n &&
                               c->stage_in_use->dev.as<float>() == d_scan) ? c->stage_in_use : nullptr;
  const float4* d_binned = c->d_binned.as<float4>();
  const uint32_t* d_chunks = c->d_chunks.as<uint32_t>();
  uint32_t chunk_cap = (uint32_t)(c->d_chunks.cap / 4);
  // (the prologue rides on the first k-NN launch -- MatchParams::begin -- unless that is the instrumented instantiation, whose
  //  statistics share the histogram block the prologue clears)
  const bool begin_in_knn = pb && c->ablate == 0;
  // A SMALL scan -- the stock operating point of the node: max_surface_features 2000 / 4000 of a pre-filtered cloud -- is not binned at all:
  // every kept query gets a wavefront of its own (knn_query_wave_kernel), the prologue rides on the first sweep.  Single device,
  // single registration; the instrumented build keeps the chunked sweep (its stamps describe that kernel).
  const int max_sf_cfg = c->cfg.max_surface_features;
  const size_t kept_upper = (max_sf_cfg >= 0 && n > (size_t)max_sf_cfg) ? (size_t)max_sf_cfg + 2 : n;  // (the rule keeps ~ rate * n points)
  const bool query_waves = c->query_waves && n && kept_upper <= kQueryWaveMaxKept && c->cfg.world_size <= 1 && !c->batch_mode && !c->borrow.on &&
                           !qsplit && c->ablate == 0;
  if (query_waves) {
    st->flags |= SO_ICP_FLAG_QUERY_WAVES;
  } else if (pb) {
    if (!begin_in_knn)
      launch_reg_begin_prebinned(ds, pose_in, max_outer, lm_max, c->d_hist, pb->pb_ctr.as<unsigned long long>(), c->d_status.as<uint8_t>(), (uint32_t)n,
                                 c->cfg.max_surface_features, s);
    d_binned = pb->pb_binned.as<float4>(); d_chunks = pb->pb_chunks.as<uint32_t>(); chunk_cap = pb->pb_chunk_cap;
    st->flags |= SO_ICP_FLAG_BINNED_AHEAD;
  } else if (n) {
    // hash binning: keys + per-key counts (scan_keys), bucket offsets + chunk list (bin_offsets), placement (bin_place).
    // The table has >= 2 slots per query; bin_offsets leaves it empty again.
    uint32_t lg = 16;
    while ((1ull << lg) < 2 * (unsigned long long)n) ++lg;
    if (c->bin_log2 != lg || c->bin_dirty) {
      const size_t T = (size_t)1 << lg;
      HIP_TRY(c, c->d_bin_key.reserve(T * 4)); HIP_TRY(c, c->d_bin_cnt.reserve(T * 4)); HIP_TRY(c, c->d_bin_off.reserve(T * 4));
      HIP_TRY(c, hipMemsetAsync(c->d_bin_key.p, 0xFF, T * 4, s));
      HIP_TRY(c, hipMemsetAsync(c->d_bin_cnt.p, 0, T * 4, s));
      c->bin_log2 = lg;
    }
    bt = BinTable{c->d_bin_key.as<uint32_t>(), c->d_bin_cnt.as<uint32_t>(), c->d_bin_off.as<uint32_t>(), lg};
    c->bin_dirty = true;  // until bin_offsets has been enqueued behind scan_keys
    launch_scan_keys(d_scan, (uint32_t)n, ds, pose_in, max_outer, lm_max, c->d_hist, c->view, c->cfg.max_surface_features, c->cfg.rank,
                     c->cfg.world_size, c->d_keys0.as<uint32_t>(), c->d_vals0.as<uint32_t>(), c->d_status.as<uint8_t>(), bt, s, false, nullptr, 0,
                     qsplit, (uint32_t)n_total);
    launch_bin_offsets(bt, c->d_chunks.as<uint32_t>(), (uint32_t)(c->d_chunks.cap / 4), ds, s);
    c->bin_dirty = false;
    launch_bin_place(bt, d_scan, (uint32_t)n, c->d_keys0.as<uint32_t>(), c->d_vals0.as<uint32_t>(), c->d_binned.as<float4>(), s);
  } else {
    launch_scan_keys(d_scan, 0, ds, pose_in, max_outer, lm_max, c->d_hist, c->view, c->cfg.max_surface_features, c->cfg.rank,
                     c->cfg.world_size, nullptr, nullptr, nullptr, bt, s);  // (empty scan: the prologue alone)
  }
  span_end(c);
  HIP_TRY(c, hipGetLastError());  // a refused launch would otherwise surface as a 50 ms wait or "state was not published"
  const float plane_res_now = c->borrow.on ? c->borrow.plane_res : map_plane_res(c);
  MatchParams mp = match_params(plane_res_now, c->ablate);
  mp.chunk_cap = chunk_cap;
  mp.pack_light = (c->knn_pack && c->knn_pack_hold == 0 && !c->knn_list_fits) ? 1 : 0;
  mp.packed_leftover = &c->d_state->packed_leftover;
  if (c->knn_pack_hold > 0) --c->knn_pack_hold;
  if (mp.ablate & 128) {  // profiling: per-workgroup phase stamps of the k-NN sweeps
    HIP_TRY(c, c->d_kdbg.reserve((size_t)2 * kKnnBlocks * 4 * 16 * sizeof(unsigned long long)));
    HIP_TRY(c, hipMemsetAsync(c->d_kdbg.p, 0, (size_t)2 * kKnnBlocks * 4 * 16 * sizeof(unsigned long long), s));
    mp.kdbg = c->d_kdbg.as<unsigned long long>();
  }
  EvalParams ep = eval_params(plane_res_now, c->cfg.tukey_variant, c->ablate);
  // read-back: the controller's workgroup publishes the state block straight into the pinned mirrors (polled below);
  // SOICP_READBACK=copy (or the controller ablated away) falls back to hipMemcpyAsync + event
  const bool direct_rb = c->direct_readback && !(ep.ablate & 32);
  const unsigned long long seq_base = (++c->reg_counter) << 8;
  if (direct_rb) { ep.hring[0] = c->d_ring[0]; ep.hring[1] = c->d_ring[1]; ep.seq_base = seq_base; }
  ep.n_queries = (uint32_t)n; ep.q_stride = 3;  // evaluation kernels read the scan itself, in its own order
  ep.defer_publish = 0;
  mp.hring[0] = ep.hring[0]; mp.hring[1] = ep.hring[1]; mp.seq_base = ep.seq_base; mp.publish_prev = 0;
  CorrBuffers corr{c->d_nd.as<double4>(), c->d_coeff.as<double>(), c->d_status.as<uint8_t>()};
  std::vector<size_t> knn_span_of_outer, eval_span_first;
  // One outer iteration = knn_plane -> [ eval(slot) -> (all-reduce -> lm_step) ] x (1 + lm_max) -> state read-back.
  // (The histogram replicas are cleared by reg_begin and again by the controller when a solve ends:
  //  ResetDistanceParameters, LidarSlam.cpp:847-852.)
  // time_kernels == 1 samples every 3rd registration (a period coprime to the scan rotation of typical benchmarks, so
  // that every scan of the rotation gets timed): even dispatch-attached events cost ~5 us of stream time per
  // timed launch (completion-signal handling), which would otherwise sit inside every step of a throughput run
  const bool timed = !c->batch_mode && (c->cfg.time_kernels >= 2 || (c->cfg.time_kernels == 1 && (c->timing.registrations % 3) == 0));
  // part A: correspondences + plane fit + first evaluation; part B: the remaining evaluations + read-back
  // (concurrent hypotheses: two persistent launches could each hold part of the CUs and wait for the rest -- one launch per
  //  evaluation there; only workgroup 0 of a launch ever waits, for workgroups that finish unconditionally)
  const bool peer = c->peer_on && c->cfg.world_size > 1 && c->persistent_solve && !c->batch_mode && !(ep.ablate & 32);
  if (peer) { for (int r = 0; r < 8; ++r) ep.peer_inbox[r] = c->peer_inbox[r]; ep.peer_rank = c->cfg.rank; ep.peer_world = c->cfg.world_size; ep.timeout_ticks = c->peer_timeout_ticks; }
  // (peer exchange: the ranks' persistent solve launches trade their records themselves, see EvalParams::peer_inbox)
  const bool exchange = !peer && (c->comm != nullptr || c->group != nullptr);  // the sums pass through a collective between evaluation and controller
  const bool persistent = c->persistent_solve && !exchange && (!c->batch_mode || c->batch_single) && !(ep.ablate & 32);  // (ablated controller: per-evaluation launches)
  if (!persistent) st->flags |= SO_ICP_FLAG_PER_EVAL_LAUNCHES;
  // deferred report (see EvalParams::defer_publish): possible when the host always has the next k-NN launch in the queue
  // before it waits for a report
  const bool defer_reports = persistent && direct_rb && c->speculate;
  mp.publish_prev = defer_reports ? 1 : 0;
  auto enqueue_eval = [&](int slot) -> int {
    span_begin(c, 1, (uint32_t)n);
    const bool fuse_lm = !exchange;  // single device: the last workgroup of eval runs the LM controller itself
    launch_eval(slot, fuse_lm, d_scan, d_scan + 1, d_scan + 2, corr, ds, ep, c->d_partials,
                c->d_ticket, c->d_hist, c->d_sums, c->view, c->d_nbr5.as<uint32_t>(), mp, (uint32_t)n, s);
    span_end(c);
    if (!fuse_lm && c->group) {  // in-process group: through host memory (every member calls this the same number of times)
      HIP_TRY(c, hipMemcpyAsync(c->h_sums, c->d_sums, sizeof(LmSums), hipMemcpyDeviceToHost, s));
      HIP_TRY(c, hipStreamSynchronize(s));
      if (!c->group->allreduce(c->cfg.rank, c->h_sums))
        return fail(c, SO_ICP_E_RCCL, "in-process group: the exchange of the normal-equation sums failed (a member returned early or did not arrive)");
      HIP_TRY(c, hipMemcpyAsync(c->d_sums, c->h_sums, sizeof(LmSums), hipMemcpyHostToDevice, s));
      launch_lm_step(slot, ds, c->d_sums, c->d_hist, ep, s);
    } else if (!fuse_lm) {  // per-evaluation collective: 45 fp64 summed over the shards (xGMI, latency-bound), then the controller
      const ncclResult_t nrc = c->rccl.AllReduce(c->d_sums, c->d_sums, sizeof(LmSums) / sizeof(double), ncclDouble, ncclSum, c->comm, s);
      if (nrc != ncclSuccess) return fail(c, SO_ICP_E_RCCL, std::string("ncclAllReduce: ") + (c->rccl.GetErrorString ? c->rccl.GetErrorString(nrc) : "?"));
      launch_lm_step(slot, ds, c->d_sums, c->d_hist, ep, s);
    }
    return SO_ICP_OK;
  };
  auto enqueue_outer_a = [&](int it) -> int {
    if (it > 0 && c->cfg.world_size > 1 && n && !qsplit) {
      // sharded map: ownership follows the query's cell under the CURRENT pose, so the scan is re-binned at the start of
      // every outer iteration (a 1 degree correction at 50 m moves a point by more than the one-cell halo of a shard)
      launch_scan_keys(d_scan, (uint32_t)n, ds, pose_in, max_outer, lm_max, c->d_hist, c->view, c->cfg.max_surface_features, c->cfg.rank,
                       c->cfg.world_size, c->d_keys0.as<uint32_t>(), c->d_vals0.as<uint32_t>(), c->d_status.as<uint8_t>(), bt, s, true);
      launch_bin_offsets(bt, c->d_chunks.as<uint32_t>(), (uint32_t)(c->d_chunks.cap / 4), ds, s);
      launch_bin_place(bt, d_scan, (uint32_t)n, c->d_keys0.as<uint32_t>(), c->d_vals0.as<uint32_t>(), c->d_binned.as<float4>(), s, ds);
    }
    // processPlannerFeatures: every kept query in parallel (LidarSlam.cpp:323-344)
    knn_span_of_outer.push_back(c->spans.size());
    hipEvent_t ka = nullptr, kb = nullptr;
    if (timed) {  // the events ride on the dispatch packet (hipExtLaunchKernelGGL), no marker packets
      ka = next_event(c); kb = next_event(c);
      if (ka && kb) c->spans.push_back(EventSpan{0, ka, kb, (uint32_t)n});
    }
    MatchParams mp_it = mp;
    if (it == 0 && begin_in_knn && !query_waves) {
      mp_it.begin = 1; mp_it.begin_args.max_outer = max_outer; mp_it.begin_args.lm_max = lm_max; mp_it.begin_max_surface_features = c->cfg.max_surface_features;
      mp_it.begin_n = (uint32_t)n; std::memcpy(mp_it.begin_args.pose, pose_in, sizeof(mp_it.begin_args.pose));
      mp_it.begin_ctr = pb->pb_ctr.as<unsigned long long>(); mp_it.begin_state = ds;
    }
    if (query_waves)
      launch_knn_query_waves(d_scan, (uint32_t)n, ds, pose_in, max_outer, lm_max, it == 0, c->d_hist, c->view, mp, c->cfg.max_surface_features,
                             c->d_status.as<uint8_t>(), c->d_nbr5.as<uint32_t>(), s, ka, kb);
    else
      launch_knn_plane(d_binned, d_chunks, ds, c->view, mp_it, corr, c->d_nbr5.as<uint32_t>(), c->d_hist, s, ka, kb);
    if (c->cfg.time_kernels >= 2)  // kernel statistics of this sweep (profiling mode only)
      HIP_TRY(c, hipMemcpyAsync(c->h_hist + (size_t)it * kHistReplicas * kHistStride, c->d_hist,
                                kHistReplicas * kHistStride * sizeof(int32_t), hipMemcpyDeviceToHost, s));
    // setupOptimizationProblem + solveOptimizationProblem (LidarSlam.cpp:213-240): 1 + lm_max fused evaluations
    eval_span_first.push_back(c->spans.size());
    HIP_TRY(c, hipGetLastError());
    if (persistent) return SO_ICP_OK;  // the solve launch belongs to part B: only the k-NN sweep is speculated
    return enqueue_eval(0);
  };
  auto enqueue_outer_b = [&](int it) -> int {
    const bool deferred = defer_reports && it + 1 < max_outer;  // the k-NN launch of it + 1 will be enqueued before the host waits
    if (persistent) {  // the whole solve in one launch (workgroups hand the next pose to each other on the device)
      EvalParams ep_it = ep;
      ep_it.defer_publish = deferred ? 1 : 0;
      ep_it.epoch_base = (++c->solve_launches) << 5;
      span_begin(c, 1, (uint32_t)n);
      launch_solve(lm_max, d_scan, d_scan + 1, d_scan + 2, corr, ds, ep_it, c->d_partials, c->d_ticket,
                   c->d_hist, c->d_sums, c->view, c->d_nbr5.as<uint32_t>(), mp, (uint32_t)n, (uint32_t)c->n_cus, s);
      span_end(c);
    } else {
      for (int slot = 1; slot <= lm_max; ++slot) { const int r = enqueue_eval(slot); if (r) return r; }
    }
    // the whole state block (pose, per-iteration statistics, final normal equations) into this iteration's pinned mirror
    if (!direct_rb) HIP_TRY(c, hipMemcpyAsync(c->h_ring[it & 1], ds, sizeof(DevState), hipMemcpyDeviceToHost, s));
    // (a deferred report is complete only after the NEXT k-NN launch: the event is recorded behind that one, see the loop)
    // (the pinned mirrors are polled; the event is the watchdog of that wait only where the stream itself cannot serve as one)
    if (!direct_rb) HIP_TRY(c, hipEventRecord(c->ev_outer[it & 1], s));
    HIP_TRY(c, hipGetLastError());
    return SO_ICP_OK;
  };
  // wait until outer iteration `it` has been reported
  auto await_outer = [&](int it) -> int {
    if (!direct_rb) { HIP_TRY(c, hipEventSynchronize(c->ev_outer[it & 1])); return SO_ICP_OK; }
    volatile unsigned long long* seq = &c->h_ring[it & 1]->seq;
    const unsigned long long want = seq_base | (unsigned long long)(it + 1);
    constexpr int kReportWatchdogMs = 5;  // (a registration lasts 0.15 ms; the waits inside a solve launch give up after 50 ms)
    auto next_check = std::chrono::steady_clock::now() + std::chrono::milliseconds(kReportWatchdogMs);
    for (unsigned spin = 1;; ++spin) {
      if (*seq == want) break;
      if ((spin & 0x3FFu) != 0) continue;
      const auto now = std::chrono::steady_clock::now();
      if (now < next_check) continue;
      next_check = now + std::chrono::milliseconds(1);
      // watchdog: everything enqueued so far -- the launch that reports this iteration included -- has completed
      if (hipStreamQuery(s) != hipErrorNotReady) {
        (void)hipGetLastError();  // (hipErrorNotReady of the earlier polls, or the error the synchronize below reports)
        // the iteration's launches have all completed: either it was a no-op (converged earlier: cannot happen for the
        // iteration the host waits on) or a kernel failed -- report instead of spinning forever
        if (*seq == want) break;
        HIP_TRY(c, hipStreamSynchronize(s));
        if (*seq == want) break;
        if (persistent && peer) {
          // The ranks' pass counters (DevState::peer_seq) and inboxes can no longer be assumed equal: chunks of the failed attempt
          // still carry tags the next registration would reuse.  The peer path is left until the caller repeats the collective
          // handshake (so_icp_peer_export clears the inbox and the counter, _connect, _enable).
          c->peer_on = false; c->peer_connected = false;
          return fail(c, SO_ICP_E_HIP, "peer exchange: a solve launch was abandoned (a rank's records did not arrive within SOICP_PEER_TIMEOUT_MS, or the "
                                       "workgroups were not co-resident); the peer exchange is now disabled on this rank until so_icp_peer_export / "
                                       "_connect / _enable are repeated on every rank");
        }
        if (persistent) {
          // The persistent solve launch needs all of its workgroups resident at once.  If the device could not provide that
          // (compute units held by another process, a partitioned device, ...) its waits gave up after 50 ms: fall back to
          // one launch per evaluation for the rest of this context's life and run the registration again.
          c->persistent_solve = false;
          c->err = "persistent solve launch did not complete (workgroups not co-resident?): using per-evaluation launches from now on";
          return kRetryWithoutPersistentSolve;
        }
        return fail(c, SO_ICP_E_HIP, "registration state was not published by the device");
      }
    }
    std::atomic_thread_fence(std::memory_order_acquire);
    return SO_ICP_OK;
  };
  // The host stays ahead of what it knows: part A of iteration it+1 (the k-NN sweep; with a sharded map also the fit
  // evaluation) is enqueued before the report of iteration it is awaited, so the device never idles on a host round trip;
  // part B (the solve launch / the remaining evaluations) follows as soon as the report says "not converged" -- the device
  // is then busy with part A for tens of microseconds.  If iteration it did converge, the speculated launch is a no-op
  // (every kernel consults DevState::reg_done) that drains while the host post-processes.
  // (SOICP_SPECULATE=0 enqueues nothing ahead: every launch of a profiled run is then a real one.)
  int last = 0;
  if ((rc = enqueue_outer_a(0))) return rc;
  // the NEXT scan's DMA goes out right behind this registration's first launch (the rest of what the copy queue does for that
  // scan follows below, once the launches that are not urgent -- the first sweep lasts 20 us -- are in the queue as well)
  if (!c->batch_mode) stage_issue_deferred_copy(c);
  if ((rc = enqueue_outer_b(0))) return rc;
  for (int it = 0;; ++it) {
    if (c->speculate && it + 1 < max_outer && (rc = enqueue_outer_a(it + 1))) return rc;
    // this registration's launches are in the queue and the host is about to idle: the moment for the NEXT scan's DMA
    if (!c->batch_mode && it == 0) {
      stage_issue_deferred(c, (may_prebin && !query_waves) ? pose_in : nullptr);  // (the next scan is binned behind its copy, under this registration's guess; a stream of small scans is not binned at all)
    }
    if ((rc = await_outer(it))) return rc;
    last = it;
    if (c->h_ring[it & 1]->reg_done || it + 1 >= max_outer) break;
    if (!c->speculate && (rc = enqueue_outer_a(it + 1))) return rc;  // SOICP_SPECULATE=0: no launch that could turn out a no-op
    if ((rc = enqueue_outer_b(it + 1))) return rc;
  }
  c->h_state = c->h_ring[last & 1];
  const DevState& H = *c->h_state;
  if (mp.pack_light) c->timing.knn_pack_registrations++;
  // (the device's count runs on from registration to registration -- nothing on the device clears it beside the sweeps that add to it)
  const uint32_t packed_left = H.packed_leftover >= c->packed_leftover_seen ? H.packed_leftover - c->packed_leftover_seen : H.packed_leftover;
  c->packed_leftover_seen = H.packed_leftover;
  if (mp.pack_light && (double)packed_left > 0.03 * (double)n * (double)std::max(H.n_iterations, 1)) { c->knn_pack_hold = 32; c->timing.knn_pack_holds++; }
  // (scans of a stream have one size: the list of this registration decides the packing of the next -- results do not depend on it)
  c->knn_list_fits = ((H.bin_packed >> 21) & 0x1FFFFFull) + (H.bin_packed >> 42) <= (unsigned long long)kKnnBlocks * 4ull;
  if (!c->batch_mode && !c->borrow.on) c->done_count_seen = H.done_count;  // (registrations completed on the context's state block: so_icp_register_sequence)
  fill_result(c, H, pose_in, st, pose_out, !c->batch_mode);
  st->time_elapsed_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_icp).count();  // :199-200
  if (timed) {  // keep only the launches that did real work (no-op launches after convergence are excluded)
    std::vector<EventSpan> real;
    for (size_t i = 0; i < c->spans.size(); ++i) {
      const EventSpan& sp = c->spans[i];
      bool keep = (sp.kind == 2);
      for (int it = 0; it < H.n_iterations && !keep; ++it) {
        if (sp.kind == 0 && it < (int)knn_span_of_outer.size() && i == knn_span_of_outer[it]) keep = true;
        // (a persistent solve launch is ONE span per outer iteration, the per-evaluation schedule 1 + #LM iterations)
        if (sp.kind == 1 && it < (int)eval_span_first.size() && i >= eval_span_first[it] &&
            i < eval_span_first[it] + (persistent ? 1 : 1 + (size_t)std::max(H.iters[it].lm_iterations, 0))) keep = true;
      }
      if (keep) { EventSpan r = sp; r.units = (uint32_t)(H.bin_packed & 0x1FFFFFull); real.push_back(r); }
    }
    c->spans.swap(real);
    // profiling mode brackets the solve launches too: the last one has published its result but its stop event may not
    // have signalled yet (hipEventElapsedTime would refuse it) -- wait for the stream there; the k-NN events of mode 1
    // completed long ago
    if (c->cfg.time_kernels >= 2) HIP_TRY(c, hipStreamSynchronize(s));
    spans_collect(c);
    for (int it = 0; it < H.n_iterations && c->cfg.time_kernels >= 2; ++it)
      for (int r = 0; r < kHistReplicas; ++r) {
        const int32_t* hh = c->h_hist + ((size_t)it * kHistReplicas + r) * kHistStride;
        c->timing.knn_group_passes += hh[16]; c->timing.knn_fallback_lanes += hh[17];
        c->timing.knn_packed_rows += hh[20]; c->timing.knn_packed_rows_too_many_runs += hh[21]; c->timing.knn_packed_rows_tile_full += hh[22];
        c->timing.knn_packed_kept += hh[23];
        c->timing.knn_candidates_scanned += (int64_t)hh[18] * 16;
      }
  }
  c->timing.registrations++;
  c->timing.host_ms_total += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
  return SO_ICP_OK;
}

int register_core(so_icp_ctx* c, const float* d_scan, size_t n, const double pose_in[7], double pose_out[7], so_icp_stats* st) {
  int rc = register_core_once(c, d_scan, n, pose_in, pose_out, st);
  if (rc == kRetryWithoutPersistentSolve) {
    c->no_map_shift_once = true;  // the window was already placed for this scan
    c->retried = true;            // so_icp_stats::flags tells the caller (the context stays on per-evaluation launches)
    rc = register_core_once(c, d_scan, n, pose_in, pose_out, st);
    c->retried = false;
    if (rc == kRetryWithoutPersistentSolve) rc = fail(c, SO_ICP_E_HIP, "registration state was not published by the device");
  }
  // The other members of an in-process group must not wait for this one's next exchange -- when this one FAILED MID-SEQUENCE (a
  // device or exchange error).  A call refused on its arguments (scan too large, bad stride: checked before anything is
  // enqueued or exchanged, and refused alike on every member, which all pass the same scan) leaves the group usable.
  if ((rc == SO_ICP_E_HIP || rc == SO_ICP_E_RCCL) && c->group) c->group->abort_all();
  return rc;
}

// wait = false: the caller enqueues the scan's consumers on the same stream and does not return to ITS caller before they
// have completed (so_icp_register), so the source buffer outlives the copy without a host-side wait here
// The queue of the host-in / host-out steps around Localization() (pre-filter, de-skew, registered scan): they touch nothing the
// map insert of the previous frame uses, so they need not wait behind it in the context's queue.
static hipStream_t aux_stream(so_icp_ctx* c) {
  std::lock_guard<std::mutex> lk(c->aux_mu);  // (created on first use, and so_icp_prefilter_announce may be that use, on the callback's thread)
  if (!c->pf_stream && hipStreamCreateWithFlags(&c->pf_stream, hipStreamNonBlocking) != hipSuccess) { (void)hipGetLastError(); return c->stream; }
  return c->pf_stream;
}

int upload_scan_impl(so_icp_ctx* c, const float* xyz, size_t n, size_t stride_bytes, DevBuf& dst, bool wait = true) {
  if (stride_bytes == 0) stride_bytes = 12;
  if (stride_bytes % 4) return fail(c, SO_ICP_E_INVALID, "stride_bytes must be a multiple of 4");
  HIP_TRY(c, dst.reserve((n + 64) * 12));
  if (!n) return SO_ICP_OK;
  if (stride_bytes == 12) {
    // The map insert of the previous Localization() may still be in the context's queue: the upload then goes through the
    // auxiliary queue, beside it, and the context's queue waits for the copy's event (nothing in flight reads `dst`: the
    // registration that used it has reported, the insert's only reader of it finished before that call returned).
    hipStream_t s2 = (!wait && c->dmap && c->dmap->insert_in_flight()) ? aux_stream(c) : c->stream;
    if (s2 != c->stream && !c->ev_upload && hipEventCreateWithFlags(&c->ev_upload, hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); s2 = c->stream; }
    HIP_TRY(c, hipMemcpyAsync(dst.p, xyz, n * 12, hipMemcpyHostToDevice, s2));
    if (s2 != c->stream) { HIP_TRY(c, hipEventRecord(c->ev_upload, s2)); HIP_TRY(c, hipStreamWaitEvent(c->stream, c->ev_upload, 0)); }
    if (wait) HIP_TRY(c, hipStreamSynchronize(c->stream));
  } else {
    std::vector<float> packed(n * 3);
    const size_t sf = stride_bytes / 4;
    for (size_t i = 0; i < n; ++i) { packed[3 * i] = xyz[i * sf]; packed[3 * i + 1] = xyz[i * sf + 1]; packed[3 * i + 2] = xyz[i * sf + 2]; }
    HIP_TRY(c, hipMemcpyAsync(dst.p, packed.data(), n * 12, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
  }
  return SO_ICP_OK;
}

// ---- so_icp_stage_scan ---------------------------------------------------------------------------------------------
// Slot life: empty -> (queued, copy thread) -> ready -> in use by the registration that consumes it -> empty.
// Everything below runs under stage_mu: so_icp_stage_scan may come from another thread than the registration calls.
bool stage_any_queued(const so_icp_ctx* c) {
  for (const so_icp_ctx::StageSlot& sl : c->stage) if (sl.state == 1) return true;
  return false;
}
// the copy of a direct slot has left the caller's buffer (host-side wait; a no-op in steady state: the copy was enqueued
// a registration ago)
void stage_finish_direct(so_icp_ctx::StageSlot& sl) {
  sl.deferred = false;  // (a copy that was never enqueued reads nothing)
  if (sl.tail_stream) {  // (a copy whose event was never recorded -- the registration that enqueued it failed in between: wait for the queue)
    (void)hipStreamSynchronize(sl.tail_stream);
    sl.tail_stream = nullptr; sl.prebinned = false; sl.ev_pending = false;
  }
  if (sl.ev_pending) { (void)hipEventSynchronize(sl.ev); sl.ev_pending = false; }
}
// enqueue the DMA of a direct slot on the copy stream (under stage_mu)
// the buffers stage_prebin needs for a scan of n points, reserved by the ANNOUNCING thread (so_icp_stage_scan, under stage_mu) next to the
// slot's scan buffer: a hipMalloc / hipFree is a device-wide synchronisation and must not sit in a registration's critical path,
// where stage_prebin runs.  A failure only means "not binned ahead".
static inline uint32_t prebin_table_log2(size_t n) {
  uint32_t lg = 16;
  while ((1ull << lg) < 2 * (unsigned long long)n) ++lg;
  return lg;
}
void stage_prebin_reserve(so_icp_ctx* c, so_icp_ctx::StageSlot& sl, size_t n) {
  if (!c->prebin || !n || n >= ((size_t)1 << 21)) return;
  const size_t m = n + 256, T = (size_t)1 << prebin_table_log2(n);
  bool ok = sl.pb_keys.reserve(m * 4) == hipSuccess && sl.pb_vals.reserve(m * 4) == hipSuccess && sl.pb_chunks.reserve(m * 4) == hipSuccess &&
            sl.pb_binned.reserve(m * 16) == hipSuccess && sl.pb_ctr.reserve(64) == hipSuccess;
  if (ok && (c->d_pbin_key.cap < T * 4 || c->d_pbin_cnt.cap < T * 4 || c->d_pbin_off.cap < T * 4)) {
    c->pbin_log2 = 0;  // (a table that moves is cleared again before its next use)
    ok = c->d_pbin_key.reserve(T * 4) == hipSuccess && c->d_pbin_cnt.reserve(T * 4) == hipSuccess && c->d_pbin_off.reserve(T * 4) == hipSuccess;
  }
  if (!ok) (void)hipGetLastError();
}
// bin the slot's scan on the copy queue, behind its copy (so_icp_ctx::prebin); a failure only means "not binned ahead".  Runs on the
// registration thread, inside a registration: allocates nothing -- a scan whose buffers were not reserved at its announcement is not binned ahead
void stage_prebin(so_icp_ctx* c, so_icp_ctx::StageSlot& sl, const double* pose) {
  sl.prebinned = false;
  const size_t n = sl.n;
  if (!c->prebin || !pose || !n || n >= ((size_t)1 << 21)) return;
  const uint32_t lg = prebin_table_log2(n);
  const size_t m = n + 256, T = (size_t)1 << lg;
  hipStream_t s = c->copy_stream;
  bool ok = sl.pb_keys.cap >= m * 4 && sl.pb_vals.cap >= m * 4 && sl.pb_chunks.cap >= m * 4 && sl.pb_binned.cap >= m * 16 && sl.pb_ctr.cap >= 64 &&
            c->d_pbin_key.cap >= T * 4 && c->d_pbin_cnt.cap >= T * 4 && c->d_pbin_off.cap >= T * 4;
  if (ok && c->pbin_log2 != lg) {  // (bin_offsets leaves the table empty again)
    ok = hipMemsetAsync(c->d_pbin_key.p, 0xFF, T * 4, s) == hipSuccess && hipMemsetAsync(c->d_pbin_cnt.p, 0, T * 4, s) == hipSuccess;
    c->pbin_log2 = ok ? lg : 0;
  }
  if (!ok) { (void)hipGetLastError(); c->pbin_log2 = 0; return; }
  const BinTable bt{c->d_pbin_key.as<uint32_t>(), c->d_pbin_cnt.as<uint32_t>(), c->d_pbin_off.as<uint32_t>(), lg};
  sl.pb_chunk_cap = (uint32_t)(sl.pb_chunks.cap / 4);
  launch_scan_keys(sl.dev.as<float>(), (uint32_t)n, c->d_state, pose, 0, 0, c->d_hist, c->view, c->cfg.max_surface_features, 0, 1,
                   sl.pb_keys.as<uint32_t>(), sl.pb_vals.as<uint32_t>(), nullptr, bt, s, false, nullptr, 0, false, 0, sl.pb_ctr.as<unsigned long long>());
  launch_bin_offsets(bt, sl.pb_chunks.as<uint32_t>(), sl.pb_chunk_cap, c->d_state, s, nullptr, 0, sl.pb_ctr.as<unsigned long long>());
  launch_bin_place(bt, sl.dev.as<float>(), (uint32_t)n, sl.pb_keys.as<uint32_t>(), sl.pb_vals.as<uint32_t>(), sl.pb_binned.as<float4>(), s);
  if (hipGetLastError() != hipSuccess) { c->pbin_log2 = 0; return; }  // (a refused launch may have left the table dirty: cleared before its next use)
  sl.prebinned = true;
}
// The DMA of a direct slot in two steps (both under stage_mu): stage_issue_copy enqueues the copy, stage_tail what follows it on the
// copy queue -- the binning ahead (only when the REGISTRATION THREAD calls it from inside a registration, with that registration's
// guess: `pose`; the copy thread and the announcing thread pass nullptr -- they must not read the context's map view) and the event the consuming
// registration waits for.  A registration in flight calls them apart (the copy right behind its first launch, the tail once its
// other launches are in the queue: the copy is the long pole -- 34 us for a 131 072-point scan -- and the binning should land in
// the shadow of the first solve, not beside the second sweep); everybody else calls stage_issue = both at once.
hipError_t stage_issue_copy(so_icp_ctx* c, so_icp_ctx::StageSlot& sl) {
  sl.deferred = false;
  sl.prebinned = false;
  const hipError_t e = hipMemcpyAsync(sl.dev.p, sl.src, sl.n * 12, hipMemcpyHostToDevice, c->copy_stream);
  if (e == hipSuccess) { sl.tail_stream = c->copy_stream; return e; }
  sl.state = -1; sl.err = std::string("so_icp_stage_scan: ") + hipGetErrorString(e);
  return e;
}
hipError_t stage_tail(so_icp_ctx* c, so_icp_ctx::StageSlot& sl, const double* pose = nullptr) {
  if (!sl.tail_stream) return hipSuccess;
  stage_prebin(c, sl, pose);
  const hipError_t e = hipEventRecord(sl.ev, c->copy_stream);
  if (e == hipSuccess) { sl.tail_stream = nullptr; sl.ev_pending = true; return e; }
  stage_finish_direct(sl);  // (waits for the copy queue instead)
  return hipSuccess;
}
hipError_t stage_issue(so_icp_ctx* c, so_icp_ctx::StageSlot& sl) {
  const hipError_t e = stage_issue_copy(c, sl);
  return e == hipSuccess ? stage_tail(c, sl) : e;
}
// WHEN a DMA-staged scan travels.  A copy that is enqueued while the registration thread is enqueuing its launches slows
// them down: the command processor fetches every dispatch packet and its arguments from host memory over the same PCIe link
// the copy saturates with reads (measured, MI355X: registration core +10..15 us with the copy started by the announcement
// right in front of the registration call, +0 with it started here).  So an announced copy waits for the registration in
// flight to have its launches in the queue -- the host then idles for tens of microseconds, waiting for the first solve's
// report -- and is enqueued from there.  Without a registration to piggy-back on, the copy thread enqueues it after 300 us.
void stage_issue_deferred(so_icp_ctx* c, const double* prebin_pose) {
  if (!c->stage_started) return;
  std::lock_guard<std::mutex> lk(c->stage_mu);
  for (so_icp_ctx::StageSlot& sl : c->stage) {
    if (sl.state == 2 && sl.deferred) { if (stage_issue_copy(c, sl) == hipSuccess) (void)stage_tail(c, sl, prebin_pose); }
    else if (sl.state == 2 && sl.tail_stream) (void)stage_tail(c, sl, prebin_pose);
  }
}
void stage_issue_deferred_copy(so_icp_ctx* c) {  // (the registration in flight: copy now, stage_issue_deferred for the rest later)
  if (!c->stage_started) return;
  std::lock_guard<std::mutex> lk(c->stage_mu);
  for (so_icp_ctx::StageSlot& sl : c->stage) if (sl.state == 2 && sl.deferred) (void)stage_issue_copy(c, sl);
}
bool host_range_registered(const so_icp_ctx* c, const void* p, size_t bytes) {
  const char* q = static_cast<const char*>(p);
  for (const so_icp_ctx::HostRange& r : c->host_ranges) if (q >= r.p && q + bytes <= r.p + r.bytes) return true;
  return false;
}

// Copy thread: one per context, started on first use, for sources that are NOT registered host memory.  Pack (strided input,
// e.g. 32-byte pcl::PointXYZI) into the slot's pinned buffer, hipMemcpyAsync to the slot's HBM buffer, wait, mark the slot
// ready.  A copy from pinned memory runs on an SDMA engine, whereas the runtime may serve a pageable source with a blit
// kernel that competes for compute units with the persistent solve launch of the registration in flight.
void stage_worker(so_icp_ctx* c) {
  (void)hipSetDevice(c->cfg.device_id);
  std::unique_lock<std::mutex> lk(c->stage_mu);
  for (;;) {
    if (!(c->stage_quit || stage_any_queued(c))) {
      // DMA-staged scans waiting for a registration to enqueue them (stage_issue_deferred): after 300 us this thread does it
      {
        const auto now = std::chrono::steady_clock::now();
        bool waiting = false;
        auto deadline = now + std::chrono::hours(1);
        for (so_icp_ctx::StageSlot& q : c->stage) {
          if (!(q.state == 2 && q.deferred)) continue;
          const auto due = q.t_announced + std::chrono::microseconds(300);
          if (due <= now) (void)stage_issue(c, q); else { waiting = true; deadline = std::min(deadline, due); }
        }
        if (waiting) {
          c->stage_pending.store(0, std::memory_order_relaxed);
          c->stage_timed.store(true);
          c->stage_parked.store(true);
          c->stage_cv.wait_until(lk, deadline, [&] { return c->stage_quit || stage_any_queued(c); });
          c->stage_parked.store(false);
          c->stage_timed.store(false);
          continue;
        }
      }
      // Nothing queued.  A registration stream announces the next scan within a few hundred microseconds: spin that long
      // on the pending counter (no futex wake-up on the announcing thread's path), then park on the condition variable
      // (a 10 Hz node finds the thread parked and pays one notify per frame).
      c->stage_pending.store(0, std::memory_order_relaxed);
      lk.unlock();
      const auto t0 = std::chrono::steady_clock::now();
      bool got = false;
      while (std::chrono::steady_clock::now() - t0 < std::chrono::microseconds(400)) {
        if (c->stage_pending.load(std::memory_order_acquire) > 0) { got = true; break; }
#if defined(__x86_64__) || defined(__i386__)
        __builtin_ia32_pause();
#else
        std::this_thread::yield();
#endif
      }
      lk.lock();
      if (!got) {
        auto any_deferred = [&] { for (const so_icp_ctx::StageSlot& q : c->stage) if (q.state == 2 && q.deferred) return true; return false; };
        c->stage_parked.store(true);
        c->stage_cv.wait(lk, [&] { return c->stage_quit || stage_any_queued(c) || any_deferred(); });
        c->stage_parked.store(false);
      }
      continue;
    }
    if (c->stage_quit) return;
    so_icp_ctx::StageSlot* pick = nullptr;  // oldest queued announcement first
    for (so_icp_ctx::StageSlot& q : c->stage) if (q.state == 1 && (!pick || q.seq < pick->seq)) pick = &q;
    so_icp_ctx::StageSlot& sl = *pick;
    const float* src = sl.src; const size_t n = sl.n, stride = sl.stride; const unsigned long long seq = sl.seq;
    lk.unlock();
    std::string err;
    hipError_t e = sl.dev.reserve((n + 64) * 12);
    if (e == hipSuccess && n) {
      if (sl.pinned_cap < n * 12) {
        if (sl.pinned) (void)hipHostFree(sl.pinned);
        sl.pinned = nullptr; sl.pinned_cap = 0;
        e = hipHostMalloc(reinterpret_cast<void**>(&sl.pinned), n * 12 + 4096);
        if (e == hipSuccess) sl.pinned_cap = n * 12 + 4096;
      }
      // pack and copy in pieces of 16 384 points (192 KB): the DMA of a piece runs while the next one is being packed, so a
      // scan is in HBM after max(pack, DMA) + one piece instead of pack + DMA (it matters for the one copy nothing hides:
      // the first of a run)
      const size_t sf = stride / 4, piece = 16384;
      for (size_t i0 = 0; i0 < n && e == hipSuccess; i0 += piece) {
        const size_t m = std::min(piece, n - i0);
        float* dst = sl.pinned + 3 * i0;
        if (sf == 3) std::memcpy(dst, src + 3 * i0, m * 12);
        else for (size_t i = 0; i < m; ++i) { const float* p = src + (i0 + i) * sf; dst[3 * i] = p[0]; dst[3 * i + 1] = p[1]; dst[3 * i + 2] = p[2]; }
        e = hipMemcpyAsync(sl.dev.as<float>() + 3 * i0, dst, m * 12, hipMemcpyHostToDevice, c->copy_stream);
      }
      if (e == hipSuccess) e = hipStreamSynchronize(c->copy_stream);
    }
    if (e != hipSuccess) err = std::string("so_icp_stage_scan: ") + hipGetErrorString(e);
    lk.lock();
    if (sl.seq == seq && sl.state == 1) { sl.state = err.empty() ? 2 : -1; sl.err = err; }
    c->stage_cv.notify_all();
  }
}

// The staged copy of (xyz, n, stride) -- the NEWEST announcement of that buffer --, waiting for the copy thread if it is still
// on its way; nullptr = not staged.  A staged copy is consumed by the call that takes it: the caller may refill the same host
// buffer for a later frame, and a later call without a new so_icp_stage_scan must not see the old contents.  (The HBM buffer
// itself stays valid until the slot is staged again, i.e. for the whole call that took it.)  A copy that the announcing
// thread enqueued itself (registered host memory) is awaited by the registration's stream, not by the host.
const float* take_staged(so_icp_ctx* c, const float* xyz, size_t n, size_t stride_bytes, int* rc) {
  *rc = SO_ICP_OK;
  std::unique_lock<std::mutex> lk(c->stage_mu);
  if (!c->stage_started) return nullptr;
  so_icp_ctx::StageSlot* best = nullptr;
  for (so_icp_ctx::StageSlot& sl : c->stage) {
    if (sl.state == 0 || sl.state == 3 || sl.src != xyz || sl.n != n || sl.stride != stride_bytes) continue;
    if (sl.seq < c->stage_consumed_seq) {
      // announced before a scan that has been consumed since: its frame was skipped, and the caller may have refilled the
      // buffer meanwhile (allowed once a later so_icp_stage_scan has returned) -- never served
      c->stage_cv.wait(lk, [&] { return sl.state != 1; });
      stage_finish_direct(sl);
      sl.src = nullptr; sl.state = 0;
      continue;
    }
    if (!best || sl.seq > best->seq) best = &sl;
  }
  if (!best) return nullptr;
  for (so_icp_ctx::StageSlot& sl : c->stage) {  // older announcements of the same buffer: superseded
    if (&sl == best || sl.state == 0 || sl.state == 3 || sl.src != xyz || sl.n != n || sl.stride != stride_bytes) continue;
    c->stage_cv.wait(lk, [&] { return sl.state != 1; });
    stage_finish_direct(sl);
    sl.src = nullptr; sl.state = 0;
  }
  so_icp_ctx::StageSlot& sl = *best;
  if (sl.state == 1) {
    const auto t0 = std::chrono::steady_clock::now();
    c->stage_cv.wait(lk, [&] { return sl.state != 1; });
    c->timing.stage_wait_ms_total += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  }
  const int state = sl.state;
  sl.src = nullptr;
  if (state == 2) {
    if (sl.deferred) {  // (no registration came by to enqueue it: the first scan of a run)
      sl.src = xyz;
      if (stage_issue(c, sl) != hipSuccess) { c->err = sl.err; *rc = SO_ICP_E_HIP; sl.src = nullptr; sl.state = 0; return nullptr; }
      sl.src = nullptr;
    }
    if (sl.tail_stream) (void)stage_tail(c, sl);  // (no registration finished what it had begun: not binned ahead)
    // (in a stream of registrations the copy ended long ago -- it was enqueued a registration earlier: then no barrier packet
    //  in front of this registration's first kernel either)
    if (sl.ev_pending && hipEventQuery(sl.ev) == hipSuccess) sl.ev_pending = false;
    else (void)hipGetLastError();
    if (sl.ev_pending && hipStreamWaitEvent(c->stream, sl.ev, 0) != hipSuccess) {
      (void)hipGetLastError();
      stage_finish_direct(sl);  // (cannot order the streams on the device: wait here)
    }
    sl.state = 3; c->stage_in_use = &sl; c->stage_consumed_seq = sl.seq;
    return sl.dev.as<float>();  // (so_icp_stage_scan -- possibly on another thread -- leaves a slot in use alone)
  }
  sl.state = 0;
  if (state == -1) { c->err = sl.err; *rc = SO_ICP_E_HIP; }
  return nullptr;
}
// A call that reads (xyz, n, stride) itself -- map seeding -- without consuming a staged copy of it: the copy is dropped, so that
// a later frame whose buffer happens to have the same address and size is never served the old contents.
void drop_staged(so_icp_ctx* c, const float* xyz, size_t n, size_t stride_bytes) {
  std::unique_lock<std::mutex> lk(c->stage_mu);
  if (!c->stage_started) return;
  for (so_icp_ctx::StageSlot& sl : c->stage) {
    if (sl.state == 0 || sl.state == 3 || sl.src != xyz || sl.n != n || sl.stride != stride_bytes) continue;
    c->stage_cv.wait(lk, [&] { return sl.state != 1; });
    stage_finish_direct(sl);
    sl.src = nullptr; sl.state = 0;
  }
}
void release_staged(so_icp_ctx* c) {
  if (!c->stage_in_use) return;
  std::lock_guard<std::mutex> lk(c->stage_mu);
  c->stage_in_use->ev_pending = false;  // (the registration that read the slot has completed, and the copy before it)
  c->stage_in_use->state = 0;
  c->stage_in_use = nullptr;
}

// the scan of this call in HBM: the staged copy when the caller announced it, else a plain upload into d_scan_own
// (enqueued on the registration's own stream in front of its kernels: no host-side wait)
int resolve_scan(so_icp_ctx* c, const float* xyz, size_t n, size_t stride_bytes, const float** d_scan) {
  int rc = SO_ICP_OK;
  c->scan_staged = false;
  if (const float* staged = take_staged(c, xyz, n, stride_bytes ? stride_bytes : 12, &rc)) { c->scan_staged = true; *d_scan = staged; return SO_ICP_OK; }
  if (rc) return rc;
  rc = upload_scan_impl(c, xyz, n, stride_bytes, c->d_scan_own, /*wait=*/false);
  *d_scan = c->d_scan_own.as<float>();
  return rc;
}


// ---- so_icp_register_batch: B hypotheses of one scan in the same launches ------------------------------------------
// One binning launch sequence over (queries x hypotheses), then rounds of { one k-NN launch over the chunks of every
// hypothesis still iterating, one persistent solve launch in which every such hypothesis owns a group of workgroups
// and runs its own LM controller (kernels.hip: solve_kernel<BATCH>) , one read-back of the state blocks }.  A hypothesis
// is an independent registration: it leaves the rounds when its own termination rule fires (LidarSlam.cpp:141), and the
// workgroups it held go to the others in the next round.  Results are bit-identical to so_icp_register per hypothesis.
constexpr int kBatchMaxConcurrent = 64;
int batch_reserve(so_icp_ctx* c, uint32_t B, size_t n, uint32_t lg) {
  so_icp_ctx::BatchBufs& b = c->batch;
  const uint32_t bs = (uint32_t)(((n + 256 + 63) / 64) * 64);
  const size_t T = (size_t)1 << lg;
  const size_t partial_bytes = (size_t)kFitBlocksMax * kRecordChunksMax * 16;
  if (B > b.cap_hyp || bs > b.bs || lg != b.table_log2) {
    const uint32_t cap = std::max(B, b.cap_hyp), nbs = std::max(bs, b.bs);
    b.release();
    HIP_TRY(c, b.states.reserve((size_t)cap * sizeof(DevState))); HIP_TRY(c, b.begin.reserve((size_t)cap * sizeof(RegBeginArgs)));
    HIP_TRY(c, b.active.reserve((size_t)cap * 4));
    for (DevBuf* d : {&b.qslot, &b.qrank, &b.chunks}) HIP_TRY(c, d->reserve((size_t)cap * nbs * 4));
    HIP_TRY(c, b.binned.reserve((size_t)cap * nbs * 16));
    HIP_TRY(c, b.status.reserve((size_t)cap * nbs)); HIP_TRY(c, b.nbr5.reserve((size_t)cap * nbs * 20));
    HIP_TRY(c, b.nd.reserve((size_t)cap * nbs * 32)); HIP_TRY(c, b.coeff.reserve((size_t)cap * nbs * 8));
    for (DevBuf* d : {&b.bin_key, &b.bin_cnt, &b.bin_off}) HIP_TRY(c, d->reserve((size_t)cap * T * 4));
    HIP_TRY(c, b.partials.reserve((size_t)cap * partial_bytes)); HIP_TRY(c, b.sync.reserve((size_t)cap * kSyncBytes));
    HIP_TRY(c, b.hist.reserve((size_t)cap * kHistReplicas * kHistStride * 4));
    HIP_TRY(c, hipHostMalloc(reinterpret_cast<void**>(&b.h_states), (size_t)cap * sizeof(DevState)));
    HIP_TRY(c, hipHostMalloc(reinterpret_cast<void**>(&b.h_begin), (size_t)cap * sizeof(RegBeginArgs)));
    HIP_TRY(c, hipHostMalloc(reinterpret_cast<void**>(&b.h_active), (size_t)cap * 4));
    // tags / epochs of the record tables and hand-off blocks count up from zero; the state blocks start cleared
    HIP_TRY(c, hipMemsetAsync(b.partials.p, 0, (size_t)cap * partial_bytes, c->stream));
    HIP_TRY(c, hipMemsetAsync(b.sync.p, 0, (size_t)cap * kSyncBytes, c->stream));
    HIP_TRY(c, hipMemsetAsync(b.states.p, 0, (size_t)cap * sizeof(DevState), c->stream));
    HIP_TRY(c, hipMemsetAsync(b.hist.p, 0, (size_t)cap * kHistReplicas * kHistStride * 4, c->stream));
    b.cap_hyp = cap; b.bs = nbs; b.table_log2 = lg; b.tables_clean = false;
  }
  if (!b.tables_clean) {  // (bin_offsets leaves the tables empty again)
    HIP_TRY(c, hipMemsetAsync(b.bin_key.p, 0xFF, (size_t)b.cap_hyp * T * 4, c->stream));
    HIP_TRY(c, hipMemsetAsync(b.bin_cnt.p, 0, (size_t)b.cap_hyp * T * 4, c->stream));
  }
  return SO_ICP_OK;
}

int register_batch_group(so_icp_ctx* c, const float* d_scan, size_t n, const double* poses_in, int B, double* poses_out, so_icp_stats* stats,
                         int32_t* hyp_rc, const int pos[3], int count_5x5) {
  const auto t_begin = std::chrono::steady_clock::now();
  std::vector<so_icp_stats> local;
  if (!stats) { local.resize((size_t)B); stats = local.data(); }
  for (int h = 0; h < B; ++h) {
    so_icp_stats* st = stats + h;
    std::memset(st, 0, sizeof(*st));
    st->flags = (!c->dmap ? SO_ICP_FLAG_HOST_MAP : 0u) | (c->direct_readback ? 0u : SO_ICP_FLAG_COPY_READBACK);
    std::memcpy(poses_out + 7 * (size_t)h, poses_in + 7 * (size_t)h, 7 * sizeof(double));
    if (c->have_hist) uncertainty_from_hist(c->prev_obs_hist, st->uncertainty);
    st->pos_in_localmap[0] = pos[0]; st->pos_in_localmap[1] = pos[1]; st->pos_in_localmap[2] = pos[2];
    st->laser_cloud_surf_from_map_num = count_5x5; st->laser_cloud_surf_stack_num = (int32_t)n; st->startup_count = c->startup_count;
    hyp_rc[h] = SO_ICP_OK;
  }
  if (!(count_5x5 > 50)) { for (int h = 0; h < B; ++h) hyp_rc[h] = SO_ICP_NOT_ENOUGH_MAP_FEATURES; return SO_ICP_OK; }  // LidarSlam.cpp:113-116
  if (n >= ((size_t)1 << 21)) return fail(c, SO_ICP_E_UNSUPPORTED, "scan of 2^21 points or more: the work-list counters hold 21 bits each (chunk descriptors 26)");
  const int max_outer = std::min(c->cfg.max_iterations > 0 ? c->cfg.max_iterations : 4, SO_ICP_MAX_OUTER);
  const int lm_max = std::min(c->cfg.lm_max_iterations > 0 ? c->cfg.lm_max_iterations : 4, 16);
  uint32_t lg = 16;
  while ((1ull << lg) < 2 * (unsigned long long)n) ++lg;
  int rc = batch_reserve(c, (uint32_t)B, n, lg);
  if (rc) return rc;
  so_icp_ctx::BatchBufs& b = c->batch;
  hipStream_t s = c->stream;
  for (int h = 0; h < B; ++h) {
    std::memcpy(b.h_begin[h].pose, poses_in + 7 * (size_t)h, 7 * sizeof(double));
    b.h_begin[h].max_outer = max_outer; b.h_begin[h].lm_max = lm_max; b.h_begin[h].chain_expect = 0; b.h_begin[h].pad = 0;  // (a hypothesis starts from its own guess)
    b.h_active[h] = (uint32_t)h;
  }
  HIP_TRY(c, hipMemcpyAsync(b.begin.p, b.h_begin, (size_t)B * sizeof(RegBeginArgs), hipMemcpyHostToDevice, s));
  HIP_TRY(c, hipMemcpyAsync(b.active.p, b.h_active, (size_t)B * 4, hipMemcpyHostToDevice, s));
  const float plane_res_now = map_plane_res(c);
  MatchParams mp = match_params(plane_res_now, c->ablate);
  mp.chunk_cap = b.bs;
  mp.hring[0] = mp.hring[1] = nullptr; mp.seq_base = 0; mp.publish_prev = 0;
  mp.packed_leftover = &b.states.as<DevState>()->packed_leftover;
  EvalParams ep = eval_params(plane_res_now, c->cfg.tukey_variant, c->ablate);
  ep.n_queries = (uint32_t)n; ep.q_stride = 3;
  ep.timeout_ticks = 20000000ull;  // 200 ms: a pass of one hypothesis on a few workgroups lasts up to a millisecond
  const uint32_t v_grid = solve_grid((uint32_t)n, (uint32_t)c->n_cus);
  const uint32_t resident = solve_batch_resident_blocks((uint32_t)c->n_cus, c->batch_degrade >= 1 ? 1 : 0);
  if (resident < (uint32_t)B) {  // (the driver below sizes its groups by the resident workgroups; this is the second line of defence)
    c->err = "so_icp_register_batch: fewer resident solve workgroups (" + std::to_string(resident) + ") than hypotheses in the group (" + std::to_string(B) + ")";
    return kRetryWithoutPersistentSolve;  // degrade (fewer workgroups per hypothesis is not possible: one each) -> concurrent sequential registrations
  }
  BatchView bv{b.active.as<uint32_t>(), b.begin.as<RegBeginArgs>(), b.bs, (uint32_t)((size_t)1 << lg),
               (uint32_t)((size_t)kFitBlocksMax * kRecordChunksMax * 2), (uint32_t)(kSyncBytes / 4), 1u, v_grid};
  const BinTable bt{b.bin_key.as<uint32_t>(), b.bin_cnt.as<uint32_t>(), b.bin_off.as<uint32_t>(), lg};
  DevState* ds = b.states.as<DevState>();
  CorrBuffers corr{b.nd.as<double4>(), b.coeff.as<double>(), b.status.as<uint8_t>()};
  // ---- binning of the scan under every hypothesis' pose: (queries x hypotheses) in three launches
  b.tables_clean = false;
  const double zero_pose[7] = {0, 0, 0, 0, 0, 0, 1};
  launch_scan_keys(d_scan, (uint32_t)n, ds, zero_pose, max_outer, lm_max, b.hist.as<int32_t>(), c->view, c->cfg.max_surface_features, 0, 1,
                   b.qslot.as<uint32_t>(), b.qrank.as<uint32_t>(), b.status.as<uint8_t>(), bt, s, false, &bv, (uint32_t)B);
  launch_bin_offsets(bt, b.chunks.as<uint32_t>(), b.bs, ds, s, &bv, (uint32_t)B);
  b.tables_clean = true;
  launch_bin_place(bt, d_scan, (uint32_t)n, b.qslot.as<uint32_t>(), b.qrank.as<uint32_t>(), b.binned.as<float4>(), s, nullptr, &bv, (uint32_t)B);
  HIP_TRY(c, hipGetLastError());
  std::vector<uint32_t> act((size_t)B);
  for (int h = 0; h < B; ++h) act[(size_t)h] = (uint32_t)h;
  // Rounds are CHAINED -- enqueued on the same list without the host looking at the report in between -- while most of the list is
  // expected to go on: a hypothesis that has finished makes its workgroups of a later round return at once (reg_done), so a stale
  // list costs launches, never results.  After round 0 always (one outer iteration cannot meet the convergence test of most
  // guesses, and a list that shrinks by less than half keeps its workgroups per hypothesis anyway); after a later round when
  // the previous batch of this context found three quarters of that round's list still active (batch_survivors).  Every
  // report + synchronisation left out is 35 us in which the device sits idle (measured: 4 per batch of 5.7 ms).
  for (int it = 0; it < max_outer && !act.empty();) {
    const uint32_t n_act = (uint32_t)act.size();
    if (it > 0) {  // (round 0 uses the identity list uploaded with the poses; the stream was synchronised by the last read-back)
      for (uint32_t k = 0; k < n_act; ++k) b.h_active[k] = act[k];
      HIP_TRY(c, hipMemcpyAsync(b.active.p, b.h_active, (size_t)n_act * 4, hipMemcpyHostToDevice, s));
    }
    // workgroups per hypothesis: the resident grid split evenly (a power of two, never more than the grid they stand in for)
    uint32_t G = 1;
    while (2u * G * n_act <= resident && 2u * G <= v_grid) G *= 2u;
    bv.wg_per_hyp = G;
    const int first = it;
    for (;;) {
      MatchParams mp_it = mp;
      mp_it.skip_near_pass = it == 0 ? 1 : 0;  // round 0 starts with the full k-NN pass (hypotheses +-0.5 m / +-5 degrees off: the near pass certifies almost nothing)
      mp_it.pack_light = (c->knn_pack && !mp_it.skip_near_pass) ? 1 : 0;
      launch_knn_plane(b.binned.as<float4>(), b.chunks.as<uint32_t>(), ds, c->view, mp_it, corr,
                       b.nbr5.as<uint32_t>(), b.hist.as<int32_t>(), s, nullptr, nullptr, &bv, n_act);
      EvalParams ep_it = ep;
      ep_it.epoch_base = (++c->solve_launches) << 5;
      launch_solve_batch(lm_max, d_scan, d_scan + 1, d_scan + 2, corr, ds, ep_it, b.partials.as<double>(), b.sync.as<uint32_t>(), b.hist.as<int32_t>(),
                         c->view, b.nbr5.as<uint32_t>(), mp, bv, n_act, s);
      HIP_TRY(c, hipGetLastError());
      ++it;
      const bool chain = c->batch_chain && it < max_outer && it - 1 < so_icp_ctx::kBatchRoundsTracked && (it - 1 == 0 || c->batch_survivors[it - 1] >= 0.75f);
      if (!chain) break;
    }
    // at a synchronisation point the host needs two words per hypothesis (outer_iter, reg_done); the whole state blocks (280 KB
    // for 64 hypotheses) are read once, after the last round
    static_assert(offsetof(DevState, reg_done) == offsetof(DevState, outer_iter) + 4, "the round report reads outer_iter and reg_done together");
    HIP_TRY(c, hipMemcpy2DAsync(reinterpret_cast<char*>(b.h_states) + offsetof(DevState, outer_iter), sizeof(DevState),
                                reinterpret_cast<const char*>(ds) + offsetof(DevState, outer_iter), sizeof(DevState), 8, (size_t)B,
                                hipMemcpyDeviceToHost, s));
    HIP_TRY(c, hipStreamSynchronize(s));
    std::vector<uint32_t> next;
    uint32_t alive_after[so_icp_ctx::kBatchRoundsTracked] = {};
    for (uint32_t h : act) {
      const DevState& H = b.h_states[h];
      // a hypothesis of the list ran the rounds first .. it-1 unless it finished on the way (then outer_iter says where)
      const bool ran_all = H.outer_iter == it, finished_early = H.reg_done && H.outer_iter > first && H.outer_iter < it;
      if (!ran_all && !finished_early) {  // the hypothesis' solve did not finish (a wait inside the launch gave up)
        c->err = "so_icp_register_batch: the solve of hypothesis " + std::to_string(h) + " did not complete in round " + std::to_string(H.outer_iter) +
                 " (workgroups not co-resident: compute units held by another process?)";
        return kRetryWithoutPersistentSolve;
      }
      for (int r = first; r < it && r < so_icp_ctx::kBatchRoundsTracked; ++r)
        if (!(H.reg_done && H.outer_iter <= r + 1)) ++alive_after[r];
      if (!H.reg_done && it < max_outer) next.push_back(h);
    }
    for (int r = first; r < it && r < so_icp_ctx::kBatchRoundsTracked; ++r) c->batch_survivors[r] = (float)alive_after[r] / (float)n_act;
    act.swap(next);
  }
  HIP_TRY(c, hipMemcpyAsync(b.h_states, ds, (size_t)B * sizeof(DevState), hipMemcpyDeviceToHost, s));
  HIP_TRY(c, hipStreamSynchronize(s));
  const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
  if ((c->ablate & 128) && c->h_state)  // profiling: so_icp_debug_stamps shows the phase stamps of hypothesis 0's last solve
    for (int i = 0; i < 16; ++i) c->h_state->dbg[i] = b.h_states[0].dbg[i];
  for (int h = 0; h < B; ++h) {
    fill_result(c, b.h_states[h], poses_in + 7 * (size_t)h, stats + h, poses_out + 7 * (size_t)h, false);
    stats[h].time_elapsed_ms = ms;  // (of the whole group: the hypotheses advance together)
  }
  return SO_ICP_OK;
}

}  // namespace

so_icp_ctx::~so_icp_ctx() {
  if (group) {  // the registry forgets a group when its last member goes
    std::lock_guard<std::mutex> lk(g_groups_mu);
    if (--group->members <= 0)
      for (size_t i = 0; i < g_groups.size(); ++i)
        if (g_groups[i].second == group) { g_groups.erase(g_groups.begin() + (long)i); break; }
  }
  if (stage_started) {
    { std::lock_guard<std::mutex> lk(stage_mu); stage_quit = true; }
    stage_pending.fetch_add(1);
    stage_cv.notify_all();
    if (stage_thread.joinable()) stage_thread.join();
  }
  if (copy_stream) (void)hipStreamSynchronize(copy_stream);
  if (seq_stream) (void)hipStreamSynchronize(seq_stream);
  for (StageSlot& sl : seq_slot) { sl.dev.release(); for (DevBuf* b : {&sl.pb_keys, &sl.pb_vals, &sl.pb_chunks, &sl.pb_binned, &sl.pb_ctr}) b->release(); if (sl.ev) (void)hipEventDestroy(sl.ev); }
  for (DevBuf* b : {&d_sbin_key, &d_sbin_cnt, &d_sbin_off}) b->release();
  if (seq_stream) (void)hipStreamDestroy(seq_stream);
  for (StageSlot& sl : stage) { sl.dev.release(); for (DevBuf* b : {&sl.pb_keys, &sl.pb_vals, &sl.pb_chunks, &sl.pb_binned, &sl.pb_ctr}) b->release(); if (sl.pinned) (void)hipHostFree(sl.pinned); if (sl.ev) (void)hipEventDestroy(sl.ev); }
  for (const HostRange& r : host_ranges) { if (r.owned) (void)hipHostFree(const_cast<char*>(r.p)); else (void)hipHostUnregister(const_cast<char*>(r.p)); }
  for (int r = 0; r < 8; ++r) if (peer_opened[r] && peer_inbox[r]) (void)hipIpcCloseMemHandle(peer_inbox[r]);
  if (peer_own) (void)hipFree(peer_own);
  if (copy_stream) (void)hipStreamDestroy(copy_stream);
  for (so_icp_ctx* w : workers) delete w;
  batch.release();
  if (comm && rccl.CommDestroy) rccl.CommDestroy(comm);
  for (DevBuf* b : {&d_world, &d_mpts, &d_cell_start, &d_cube_slot, &d_scan_own, &d_keys0, &d_vals0, &d_chunks,
                    &d_binned, &d_nd, &d_coeff, &d_status, &d_nbr5, &d_small, &d_q, &d_nbr, &d_d2, &d_idx,
                    &d_found, &d_fblist, &d_kdbg, &pf_stage, &pf_in, &pf_out, &pf_small, &pf_w, &pf_s, &pf_k0, &pf_k1, &pf_v0, &pf_v1, &pf_flags, &pf_pos,
                    &pf_heads, &pf_temp, &pf_dec, &d_bin_key, &d_bin_cnt, &d_bin_off, &d_pbin_key, &d_pbin_cnt, &d_pbin_off, &d_counts, &d_sub})
    b->release();
  for (DevBuf& b : resident_scans) b.release();
  d_state_buf.release();
  for (DevState* h : h_ring) if (h) (void)hipHostFree(h);
  for (hipEvent_t e : ev_outer) if (e) (void)hipEventDestroy(e);
  if (h_hist) (void)hipHostFree(h_hist);
  if (h_sums) (void)hipHostFree(h_sums);
  if (h_u32) (void)hipHostFree(h_u32);
  if (h_pf) (void)hipHostFree(h_pf);
  if (h_pf_kept) (void)hipHostFree(h_pf_kept);
  for (hipEvent_t e : ev_pool) (void)hipEventDestroy(e);
  dmap.reset();  // (waits for a deferred insert on `stream`)
  if (pf_stream) { (void)hipStreamSynchronize(pf_stream); (void)hipStreamDestroy(pf_stream); }
  if (ev_upload) (void)hipEventDestroy(ev_upload);
  if (stream) (void)hipStreamDestroy(stream);
}

// =================================================================================================
// C ABI
// =================================================================================================
extern "C" {

int so_icp_abi_version(void) { return SO_ICP_ABI_VERSION; }

void so_icp_default_config(so_icp_config* cfg) {
  if (!cfg) return;
  std::memset(cfg, 0, sizeof(*cfg));
  cfg->abi_version = SO_ICP_ABI_VERSION;
  cfg->device_id = 0; cfg->rank = 0; cfg->world_size = 1;
  cfg->max_iterations = 5;          // config/os1_128.yaml:27 (code default 4, LidarSlam.h:273)
  cfg->lm_max_iterations = 4;       // LidarSlam.cpp:232
  cfg->max_surface_features = 2000; // config/os1_128.yaml:28
  cfg->k = 5;                       // LidarSlam.h:277
  cfg->tukey_variant = 0;
  cfg->time_kernels = 0;
  cfg->line_res = 0.1f; cfg->plane_res = 0.2f;  // config/os1_128.yaml mapping_{line,plane}_resolution
  cfg->yaw_ratio = 0.0;
  cfg->velocity_failure_threshold = 30.0;
}

int so_icp_device_available(void) {
  int n = 0;
  return hipGetDeviceCount(&n) == hipSuccess && n > 0;
}

int so_icp_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
  return n;
}

const char* so_icp_last_error(const so_icp_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

so_icp_ctx* so_icp_create(const so_icp_config* cfg) {
  g_create_error.clear();
  if (!cfg || cfg->abi_version != SO_ICP_ABI_VERSION) { g_create_error = "so_icp_create: bad config / ABI version"; return nullptr; }
  if (cfg->k != 5) { g_create_error = "so_icp_create: only k = 5 (LocalizationPlaneDistanceNbrNeighbors) is supported"; return nullptr; }
  if (cfg->world_size < 1 || cfg->rank < 0 || cfg->rank >= cfg->world_size) { g_create_error = "so_icp_create: bad rank/world_size"; return nullptr; }
  if (cfg->device_id < 0) {  // host-only: map bookkeeping for tools/tests; compute calls return SO_ICP_E_HIP
    so_icp_ctx* h = new (std::nothrow) so_icp_ctx();
    if (!h) { g_create_error = "out of memory"; return nullptr; }
    h->cfg = *cfg; h->host_only = true;
    h->map.set_resolution(cfg->line_res, cfg->plane_res);
    return h;
  }
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev <= 0) {
    g_create_error = "so_icp_create: no HIP device available (libsoicp has no CPU fallback)";
    return nullptr;
  }
  if (cfg->device_id < 0 || cfg->device_id >= ndev) { g_create_error = "so_icp_create: device_id out of range"; return nullptr; }
  if ((e = hipSetDevice(cfg->device_id)) != hipSuccess) { g_create_error = std::string("hipSetDevice: ") + hipGetErrorString(e); return nullptr; }
  so_icp_ctx* c = new (std::nothrow) so_icp_ctx();
  if (!c) { g_create_error = "out of memory"; return nullptr; }
  c->cfg = *cfg;
  c->map.set_resolution(cfg->line_res, cfg->plane_res);
  auto bail = [&](const std::string& m) { g_create_error = m; delete c; return (so_icp_ctx*)nullptr; };
  if ((e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking)) != hipSuccess) return bail(std::string("hipStreamCreate: ") + hipGetErrorString(e));
  {  // the persistent solve launch holds one workgroup per compute unit: never ask for more than the device has
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, cfg->device_id) == hipSuccess && cus > 0) c->n_cus = cus;
    if (cfg->solve_workgroups >= 1 && cfg->solve_workgroups < c->n_cus) c->n_cus = cfg->solve_workgroups;  // leave compute units to others
    if (const char* ev = std::getenv("SOICP_SOLVE_WORKGROUPS")) { const int w = std::atoi(ev); if (w >= 1 && w < c->n_cus) c->n_cus = w; }
  }
  const size_t partial_bytes = std::max((size_t)kFitBlocksMax * kSumsStride * sizeof(double),
                                        (size_t)kFitBlocksMax * kRecordChunksMax * 16);  // partial sums / tagged records of solve_kernel
  const size_t small_bytes = 4096 + sizeof(LmSums) + 256 + partial_bytes + 256 + kSyncBytes;
  if ((e = c->d_small.reserve(small_bytes)) != hipSuccess) return bail(std::string("hipMalloc: ") + hipGetErrorString(e));
  if ((e = hipMemset(c->d_small.p, 0, c->d_small.cap)) != hipSuccess) return bail(std::string("hipMemset: ") + hipGetErrorString(e));
  char* base = c->d_small.as<char>();
  c->d_hist = reinterpret_cast<int32_t*>(base);            // kHistReplicas x kHistStride ints (2 KB)
  c->d_nkept = reinterpret_cast<uint32_t*>(base + 2112);   // Seam B scratch counters
  c->d_fbcount = reinterpret_cast<uint32_t*>(base + 2176);
  c->d_sums = reinterpret_cast<LmSums*>(base + 4096);
  c->d_partials = reinterpret_cast<double*>(base + 4096 + ((sizeof(LmSums) + 255) / 256) * 256);
  c->d_ticket = reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(c->d_partials) + ((partial_bytes + 255) / 256) * 256);  // arrival counters + hand-off record (kSyncBytes)
  if ((e = hipHostMalloc(reinterpret_cast<void**>(&c->h_sums), sizeof(LmSums))) != hipSuccess) return bail(std::string("hipHostMalloc: ") + hipGetErrorString(e));
  if ((e = hipHostMalloc(reinterpret_cast<void**>(&c->h_u32), 64)) != hipSuccess) return bail(std::string("hipHostMalloc: ") + hipGetErrorString(e));
  if ((e = c->d_state_buf.reserve(sizeof(DevState))) != hipSuccess) return bail(std::string("hipMalloc: ") + hipGetErrorString(e));
  if ((e = hipMemset(c->d_state_buf.p, 0, sizeof(DevState))) != hipSuccess) return bail(std::string("hipMemset: ") + hipGetErrorString(e));
  c->d_state = c->d_state_buf.as<DevState>();
  for (int i = 0; i < 4; ++i) {
    if ((e = hipHostMalloc(reinterpret_cast<void**>(&c->h_ring[i]), sizeof(DevState), hipHostMallocMapped | hipHostMallocCoherent)) != hipSuccess)
      return bail(std::string("hipHostMalloc: ") + hipGetErrorString(e));
    std::memset(c->h_ring[i], 0, sizeof(DevState));
    if ((e = hipHostGetDevicePointer(reinterpret_cast<void**>(&c->d_ring[i]), c->h_ring[i], 0)) != hipSuccess) return bail(std::string("hipHostGetDevicePointer: ") + hipGetErrorString(e));
    if (i < 2 && (e = hipEventCreateWithFlags(&c->ev_outer[i], hipEventDisableTiming)) != hipSuccess) return bail(std::string("hipEventCreate: ") + hipGetErrorString(e));
  }
  c->h_state = c->h_ring[0];
  if ((e = hipHostMalloc(reinterpret_cast<void**>(&c->h_hist), (size_t)SO_ICP_MAX_OUTER * kHistReplicas * kHistStride * sizeof(int32_t))) != hipSuccess)
    return bail(std::string("hipHostMalloc: ") + hipGetErrorString(e));
  if (const char* ev = std::getenv("SOICP_READBACK")) c->direct_readback = std::string(ev) != "copy";
  if (const char* ev = std::getenv("SOICP_SPECULATE")) c->speculate = std::atoi(ev) != 0;
  if (const char* ev = std::getenv("SOICP_PREFILTER_FAST")) c->pf_fast = std::atoi(ev) != 0;
  if (const char* ev = std::getenv("SOICP_ABLATE")) c->ablate = std::atoi(ev);
  if (const char* ev = std::getenv("SOICP_PEER_TIMEOUT_MS")) { const long ms = std::atol(ev); if (ms >= 1 && ms <= 60000) c->peer_timeout_ticks = (unsigned long long)ms * 100000ull; }
  if (const char* ev = std::getenv("SOICP_PERSISTENT")) c->persistent_solve = std::atoi(ev) != 0;
  if (const char* ev = std::getenv("SOICP_KNN_PACK")) c->knn_pack = std::atoi(ev) != 0;
  if (const char* ev = std::getenv("SOICP_QUERY_WAVES")) c->query_waves = std::atoi(ev) != 0;
  if (const char* ev = std::getenv("SOICP_PREBIN")) c->prebin = std::atoi(ev) != 0;
  if (const char* ev = std::getenv("SOICP_SEQ_CHAIN")) c->seq_chain = std::atoi(ev) != 0;
  if (const char* ev = std::getenv("SOICP_BATCH_CHAIN")) c->batch_chain = std::string(ev) != "0";
  if (const char* ev = std::getenv("SOICP_BATCH_MODE")) {  // "one_per_cu": one solve workgroup per compute unit (several processes on one device); "lanes"
    if (std::string(ev) == "one_per_cu") c->batch_degrade = 1;
    if (std::string(ev) == "lanes") c->batch_degrade = 2;
  }
  const bool want_dmap = !(std::getenv("SOICP_HOST_MAP") && std::atoi(std::getenv("SOICP_HOST_MAP")));
  if (want_dmap) {  // world_size > 1: this rank's shard of the map, resident and updated on the device like the whole map is
    c->query_split = cfg->world_size > 1 && cfg->shard_mode == SO_ICP_SHARD_QUERIES;
    if (c->query_split) c->dmap = std::make_unique<DeviceMap>(c->stream, 0, 1);  // the whole map on every rank
    else c->dmap = std::make_unique<DeviceMap>(c->stream, cfg->rank, cfg->world_size);
    if (!c->dmap->supported_resolution(cfg->plane_res)) c->dmap.reset();  // leaf keys hold 9 bits per axis
    else { std::string e2; c->dmap->set_resolution(cfg->line_res, cfg->plane_res, e2); }
  }
  return c;
}

void so_icp_destroy(so_icp_ctx* ctx) {
  if (!ctx) return;
  if (ctx->host_only) { delete ctx; return; }
  (void)hipSetDevice(ctx->cfg.device_id);
  if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
  delete ctx;
}

int so_icp_set_resolution(so_icp_ctx* c, float line_res, float plane_res) {
  if (!c || !(plane_res > 0) || !(line_res > 0)) return SO_ICP_E_INVALID;
  if (c->dmap && !c->dmap->supported_resolution(plane_res)) return fail(c, SO_ICP_E_UNSUPPORTED, "device map needs plane_res >= 0.05 (leaf coordinates of the grouping keys hold 10 bits)");
  // ("non-empty" must be the same decision on every rank: the FULL map's count, which the ranks share after any insert under the
  //  communicator -- a rank whose own shard happens to be empty still takes part in the exchange)
  if (c->dmap && c->dmap->sharded() && c->cfg.world_size > 1 && plane_res != map_plane_res(c) &&
      ((c->group || c->comm) ? c->dmap->size() : c->dmap->size_local()) > 0) {
    // A shard holds the leaves within one CELL of the bricks it owns, and cell size and bricks follow planeRes: after a
    // change the resident subset would no longer cover the gate balls of the rank's queries (wrong neighbours, silently).
    // The shards are re-cut from every rank's points -- a collective step; without a communicator it cannot be done.
    NEED_DEVICE(c);
    HIP_TRY(c, hipSetDevice(c->cfg.device_id));
    if (!c->group && !c->comm)
      return fail(c, SO_ICP_E_UNSUPPORTED, "so_icp_set_resolution: planeRes cannot change under a sharded, non-empty map without a communicator "
                                            "(the shards are cut along the cell grid that follows planeRes; re-cutting them needs the other ranks' points)");
    const int rc = reshard_for_resolution(c, line_res, plane_res);
    if (rc) return rc;
    c->map.set_resolution(line_res, plane_res);
    c->cfg.line_res = line_res; c->cfg.plane_res = plane_res;
    return SO_ICP_OK;
  }
  if (c->dmap) {
    NEED_DEVICE(c);
    HIP_TRY(c, hipSetDevice(c->cfg.device_id));
    if (c->dmap->set_resolution(line_res, plane_res, c->err) < 0) return SO_ICP_E_HIP;
  }
  if (plane_res != c->map.plane_res()) c->uploaded_version = 0;  // cell size follows planeRes
  c->map.set_resolution(line_res, plane_res);
  c->cfg.line_res = line_res; c->cfg.plane_res = plane_res;
  return SO_ICP_OK;
}
int so_icp_set_max_surface_features(so_icp_ctx* c, int v) { if (!c) return SO_ICP_E_INVALID; c->cfg.max_surface_features = v; return SO_ICP_OK; }
int so_icp_set_max_iterations(so_icp_ctx* c, int v) { if (!c || v < 1) return SO_ICP_E_INVALID; c->cfg.max_iterations = v; return SO_ICP_OK; }

int so_icp_map_set_origin(so_icp_ctx* c, const double t[3], int o[3]) {
  if (!c || !t) return SO_ICP_E_INVALID;
  if (c->dmap) c->dmap->set_origin(t); else c->map.set_origin(t);
  if (o) { o[0] = map_origin(c)[0]; o[1] = map_origin(c)[1]; o[2] = map_origin(c)[2]; }
  return SO_ICP_OK;
}
int so_icp_map_get_origin(so_icp_ctx* c, int o[3]) {
  if (!c || !o) return SO_ICP_E_INVALID;
  o[0] = map_origin(c)[0]; o[1] = map_origin(c)[1]; o[2] = map_origin(c)[2];
  return SO_ICP_OK;
}
int so_icp_map_shift(so_icp_ctx* c, const double t[3], int pos[3]) {
  if (!c || !t || !pos) return SO_ICP_E_INVALID;
  map_shift(c, t, pos);
  return SO_ICP_OK;
}
int so_icp_map_add_surf(so_icp_ctx* c, const float* xyz, size_t n, size_t stride_bytes) {
  if (!c || (!xyz && n)) return SO_ICP_E_INVALID;
  if (stride_bytes == 0) stride_bytes = 12;
  if (stride_bytes % 4) return fail(c, SO_ICP_E_INVALID, "stride_bytes must be a multiple of 4");
  if (c->dmap) {  // bin + VoxelGrid + index rebuild on the device (map_kernels.hip)
    HIP_TRY(c, hipSetDevice(c->cfg.device_id));
    const int r = c->dmap->add_surf_host(xyz, n, stride_bytes / 4, c->err);
    if (r < 0) { if (c->group) c->group->abort_all(); return r == -1 ? SO_ICP_E_NOMEM : SO_ICP_E_HIP; }
    const int xr = exchange_map_counts(c);
    return xr ? xr : r;
  }
  return c->map.add_surf(xyz, n, stride_bytes / 4);
}
int so_icp_map_count_5x5(so_icp_ctx* c, const int pos[3], int* n_edge, int* n_surf) {
  if (!c || !pos) return SO_ICP_E_INVALID;
  if (n_edge) *n_edge = 0;
  if (n_surf) *n_surf = map_count_5x5(c, pos);
  return SO_ICP_OK;
}
int so_icp_map_export(so_icp_ctx* c, float* xyz, size_t cap, size_t* n_out, int only_5x5, const int pos[3]) {
  if (!c || (only_5x5 && !pos)) return SO_ICP_E_INVALID;
  const int zero[3] = {0, 0, 0};
  if (c->dmap) HIP_TRY(c, hipSetDevice(c->cfg.device_id));
  const size_t n = c->dmap ? c->dmap->export_points(xyz, cap, only_5x5 != 0, pos ? pos : zero, c->err)
                           : c->map.export_points(xyz, cap, only_5x5 != 0, pos ? pos : zero);
  if (n_out) *n_out = n;
  return SO_ICP_OK;
}
int so_icp_map_export_records(so_icp_ctx* c, void* out, size_t stride_bytes, size_t cap, size_t* n_out, int only_5x5, const int pos[3]) {
  if (!c || (only_5x5 && !pos)) return SO_ICP_E_INVALID;
  if (stride_bytes < 12 || stride_bytes % 4) return fail(c, SO_ICP_E_INVALID, "records: float x y z at 0 4 8, stride a multiple of 4");
  const int zero[3] = {0, 0, 0};
  size_t n = 0;
  if (c->dmap) {
    HIP_TRY(c, hipSetDevice(c->cfg.device_id));
    c->err.clear();
    n = c->dmap->export_records(out, stride_bytes, cap, only_5x5 != 0, pos ? pos : zero, c->err);
    if (!c->err.empty()) return SO_ICP_E_HIP;
  } else {  // host-side map (sharded ranks, host-only contexts): through the packed export
    n = c->map.export_points(nullptr, 0, only_5x5 != 0, pos ? pos : zero);
    if (out && n <= cap && n) {
      std::vector<float> xyz(3 * n);
      c->map.export_points(xyz.data(), n, only_5x5 != 0, pos ? pos : zero);
      std::memset(out, 0, n * stride_bytes);
      for (size_t i = 0; i < n; ++i) std::memcpy(static_cast<char*>(out) + i * stride_bytes, &xyz[3 * i], 12);
    }
  }
  if (n_out) *n_out = n;
  return SO_ICP_OK;
}
int so_icp_map_size(so_icp_ctx* c, size_t* n, size_t* n_rank) {
  if (!c) return SO_ICP_E_INVALID;
  if (n) *n = c->dmap ? c->dmap->size() : c->map.size();
  if (n_rank) { NEED_DEVICE(c); const int rc = upload_map(c); if (rc) return rc; *n_rank = c->view.n_points; }
  return SO_ICP_OK;
}
int so_icp_map_insert_stats(so_icp_ctx* c, unsigned* device_built, unsigned* handed_back) {
  if (!c) return SO_ICP_E_INVALID;
  unsigned a = 0, b = 0;
  if (c->dmap) { std::string e; (void)c->dmap->settle(e); c->dmap->fast_stats(a, b); }
  if (device_built) *device_built = a;
  if (handed_back) *handed_back = b;
  return SO_ICP_OK;
}
int so_icp_map_clear(so_icp_ctx* c) { if (!c) return SO_ICP_E_INVALID; if (c->dmap) c->dmap->clear(); else c->map.clear(); return SO_ICP_OK; }

int so_icp_knn_surf(so_icp_ctx* c, const float* q, size_t nq, int k, float* nbr, float* d2, int32_t* idx, uint8_t* found) {
  if (!c || (!q && nq) || !nbr || !d2 || !found) return SO_ICP_E_INVALID;
  NEED_DEVICE(c);
  if (k < 1 || k > 5) return fail(c, SO_ICP_E_UNSUPPORTED, "k must be in [1,5]");
  if (c->cfg.world_size != 1) return fail(c, SO_ICP_E_UNSUPPORTED, "Seam B needs the whole map on one device (world_size == 1)");
  if (!nq) return SO_ICP_OK;
  HIP_TRY(c, hipSetDevice(c->cfg.device_id));
  int rc = upload_map(c);
  if (rc) return rc;
  HIP_TRY(c, c->d_q.reserve(nq * 12)); HIP_TRY(c, c->d_nbr.reserve(nq * k * 12)); HIP_TRY(c, c->d_d2.reserve(nq * k * 4));
  HIP_TRY(c, c->d_idx.reserve(nq * k * 4)); HIP_TRY(c, c->d_found.reserve(nq)); HIP_TRY(c, c->d_fblist.reserve(nq * 4));
  HIP_TRY(c, hipMemcpyAsync(c->d_q.p, q, nq * 12, hipMemcpyHostToDevice, c->stream));
  HIP_TRY(c, hipMemsetAsync(c->d_fbcount, 0, 4, c->stream));
  // the 27-cell block certainly covers a ball of one cell edge around the query
  const double cover = (1.0 / c->view.inv_cell) * (1.0 - 1e-5);
  const float gate = (float)(cover * cover);
  launch_knn_only(c->d_q.as<float>(), (uint32_t)nq, k, c->view, gate, c->d_nbr.as<float>(), c->d_d2.as<float>(), c->d_idx.as<int32_t>(),
                  c->d_found.as<uint8_t>(), c->d_fblist.as<uint32_t>(), c->d_fbcount, c->stream);
  HIP_TRY(c, hipMemcpyAsync(c->h_u32, c->d_fbcount, 4, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  const uint32_t n_fb = c->h_u32[0];
  launch_knn_fallback(c->d_q.as<float>(), c->d_fblist.as<uint32_t>(), n_fb, k, c->view, c->d_nbr.as<float>(), c->d_d2.as<float>(), c->d_idx.as<int32_t>(), c->stream);
  HIP_TRY(c, hipMemcpyAsync(nbr, c->d_nbr.p, nq * k * 12, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipMemcpyAsync(d2, c->d_d2.p, nq * k * 4, hipMemcpyDeviceToHost, c->stream));
  if (idx) HIP_TRY(c, hipMemcpyAsync(idx, c->d_idx.p, nq * k * 4, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipMemcpyAsync(found, c->d_found.p, nq, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  return SO_ICP_OK;
}

int so_icp_upload_scan(so_icp_ctx* c, const float* xyz, size_t n, size_t stride_bytes, void** d_out) {
  if (!c || (!xyz && n) || !d_out) return SO_ICP_E_INVALID;
  NEED_DEVICE(c);
  HIP_TRY(c, hipSetDevice(c->cfg.device_id));
  DevBuf b;
  const int rc = upload_scan_impl(c, xyz, n, stride_bytes, b);
  if (rc) { b.release(); return rc; }
  c->resident_scans.push_back(b);
  *d_out = b.p;
  return SO_ICP_OK;
}

int so_icp_free_scan(so_icp_ctx* c, void* d_scan) {
  if (!c) return SO_ICP_E_INVALID;
  NEED_DEVICE(c);
  for (size_t i = 0; i < c->resident_scans.size(); ++i)
    if (c->resident_scans[i].p == d_scan) {
      HIP_TRY(c, hipStreamSynchronize(c->stream));
      c->resident_scans[i].release();
      c->resident_scans.erase(c->resident_scans.begin() + i);
      return SO_ICP_OK;
    }
  return fail(c, SO_ICP_E_INVALID, "so_icp_free_scan: unknown scan pointer");
}

int so_icp_register_dev(so_icp_ctx* c, const void* d_scan, size_t n, const double pose_in[7], double pose_out[7], so_icp_stats* st) {
  if (!c || !pose_in || !pose_out || (!d_scan && n)) return SO_ICP_E_INVALID;
  NEED_DEVICE(c);
  HIP_TRY(c, hipSetDevice(c->cfg.device_id));
  return register_core(c, static_cast<const float*>(d_scan), n, pose_in, pose_out, st);
}

int so_icp_register(so_icp_ctx* c, const float* xyz, size_t n, size_t stride_bytes, const double pose_in[7], double pose_out[7], so_icp_stats* st) {
  if (!c || !pose_in || !pose_out || (!xyz && n)) return SO_ICP_E_INVALID;
  NEED_DEVICE(c);
  HIP_TRY(c, hipSetDevice(c->cfg.device_id));
  const float* d_scan = nullptr;
  int rc = resolve_scan(c, xyz, n, stride_bytes, &d_scan);  // the copy announced with so_icp_stage_scan, or a plain upload
  if (rc) return rc;
  rc = register_core(c, d_scan, n, pose_in, pose_out, st);
  c->scan_staged = false;
  release_staged(c);
  return rc;
}

// ---- so_icp_register_sequence ---------------------------------------------------------------------------------------------
// A run of scans whose guesses chain: guess_0 = pose0, guess_k = T_(k-1) o delta_k, T_(k-1) = the pose registration k-1 ended with
// (laserMapping.cpp:345-372: T_w_lidar = T_w_lidar * prediction; pose_compose, so_math.h).  Between two so_icp_register calls of a
// stream the device idles for ~11 us: the host reads the last report, returns, is called again and enqueues the first launch of the
// next registration.  Here the NEXT registration's launches are in the queue before the current one has
// reported: the guess is formed on the device, by the solve that ends the registration in front (DevState::T_chain) -- provided the
// registration in front of it was over by then (DevState::done_count): a registration gets `seq_depth` outer iterations enqueued
// ahead (what the last one needed); one that needs more makes the launches behind it no-ops, the host finishes it with further
// launches and starts the next one again, unchained.  Every registration is the one so_icp_register runs from guesses_out[k]: same
// kernels, same arguments, same sums -- identical bits (tests/test_gpu_sequence.py).
// Chained path: single device, device-resident map, persistent solve, direct read-back, yaw_ratio 0 (the stock configurations:
// MannualYawCorrection, LidarSlam.cpp:891-913, is then the identity up to rounding; the chain starts from the optimised pose itself,
// iterations[last].pose_after); everything else -- and SOICP_SEQ_CHAIN=0 -- runs the registrations one after the other with guesses
// composed on the host by the same arithmetic.
namespace {
struct SeqRun {
  const float* d_scan = nullptr; size_t n = 0;
  so_icp_ctx::StageSlot* slot = nullptr;   // host scan (its HBM copy) and / or the work list binned ahead; nullptr: resident scan swept by query waves
  bool query_waves = false, binned = false, enqueued = false, chained = false, needs_event = false;
  bool copied = false;                      // the H2D copy of this (host) scan is already in the sequence's queue (issued one registration early)
  bool timed = false;                       // time_kernels 1: this registration's sweeps carry timing events
  struct KnnEv { int it; hipEvent_t a, b; };
  std::vector<KnnEv> knn_ev;
  uint32_t chain_expect = 0;
  double guess[7];                          // exact for an unchained start, the host's prediction for a chained one
  int pos[3] = {0, 0, 0}; int count_5x5 = 0;
  unsigned long long seq_base = 0; int ring = 0; int enq_iters = 0;
  MatchParams mp; EvalParams ep;
  const float4* d_binned = nullptr; const uint32_t* d_chunks = nullptr;
};
inline bool cube_stable(const so_icp_ctx* c, const double t[3], double margin) {
  // the window would not roll for a pose here (LocalMap.h:169-287: the sensor's block stays >= 3 blocks from the border), and no
  // pose within `margin` of it lies in another block: placing the window for the PREDICTED guess is placing it for the actual one
  const int* o = map_origin(c);
  const int dim[3] = {kMapW, kMapH, kMapD};
  for (int a = 0; a < 3; ++a) {
    const int lo = cube_coord(t[a] - margin, o[a]), hi = cube_coord(t[a] + margin, o[a]);
    if (lo != hi || lo < 3 || lo >= dim[a] - 3) return false;
  }
  return true;
}
}  // namespace

int so_icp_sequence_announce_next(so_icp_ctx* c, const float* scan, size_t n, const double delta[7]) {
  if (!c || (scan && !delta)) return SO_ICP_E_INVALID;
  NEED_DEVICE(c);
  so_icp_ctx::SeqNext& nx = c->seq_next;
  if (!scan || !n) {  // withdrawn: the announcement, and a copy the last call staged (which must have left its buffer before the caller reuses it)
    if (nx.staged && c->seq_stream) { HIP_TRY(c, hipSetDevice(c->cfg.device_id)); HIP_TRY(c, hipStreamSynchronize(c->seq_stream)); }
    nx = so_icp_ctx::SeqNext{};
    return SO_ICP_OK;
  }
  // (what the LAST call staged for the coming call to adopt -- nx.staged and its slot -- stays: this names the scan BEHIND the coming call)
  nx.next_scan = scan; nx.next_n = n; nx.announced = true;
  std::memcpy(nx.delta, delta, sizeof(nx.delta));
  return SO_ICP_OK;
}

int so_icp_register_sequence(so_icp_ctx* c, int count, const void* const* scans, const size_t* n_points, size_t stride_bytes, int scans_on_device,
                             const double pose0[7], const double* deltas, double* poses_out, double* guesses_out, so_icp_stats* stats, int* n_done) {
  if (n_done) *n_done = 0;
  if (!c || count < 0 || (count && (!scans || !n_points || !pose0 || !poses_out)) || (count > 1 && !deltas)) return SO_ICP_E_INVALID;
  NEED_DEVICE(c);
  HIP_TRY(c, hipSetDevice(c->cfg.device_id));
  if (stride_bytes == 0) stride_bytes = 12;
  for (int k = 0; k < count; ++k) if (!scans[k] && n_points[k]) return SO_ICP_E_INVALID;
  std::vector<so_icp_stats> local_stats;
  if (!stats) { local_stats.resize((size_t)count); stats = local_stats.data(); }
  const bool fast = c->seq_chain && c->dmap && c->cfg.world_size <= 1 && !c->batch_mode && !c->borrow.on && c->persistent_solve && c->direct_readback &&
                    c->speculate && c->ablate == 0 && c->cfg.time_kernels <= 1 && c->cfg.yaw_ratio == 0.0 && !c->comm && !c->group && !c->query_split &&
                    (scans_on_device || stride_bytes == 12) && count > 1;
  // the pose the chain continues from: the optimised pose of the registration, before MannualYawCorrection (fill_result)
  auto chain_from = [&](int k, double T[7]) {
    const so_icp_stats& s = stats[k];
    if (s.n_iterations > 0) std::memcpy(T, s.iterations[std::min(s.n_iterations, SO_ICP_MAX_OUTER) - 1].pose_after, 7 * sizeof(double));
    else std::memcpy(T, poses_out + 7 * (size_t)k, 7 * sizeof(double));
  };
  auto run_plain = [&](int k, const double guess[7]) -> int {  // one registration through the ordinary entry points
    if (guesses_out) std::memcpy(guesses_out + 7 * (size_t)k, guess, 7 * sizeof(double));
    return scans_on_device ? so_icp_register_dev(c, scans[k], n_points[k], guess, poses_out + 7 * (size_t)k, &stats[k])
                           : so_icp_register(c, static_cast<const float*>(scans[k]), n_points[k], stride_bytes, guess, poses_out + 7 * (size_t)k, &stats[k]);
  };
  if (!fast) {
    double guess[7];
    std::memcpy(guess, pose0, sizeof(guess));
    for (int k = 0; k < count; ++k) {
      if (k) { double T[7]; chain_from(k - 1, T); pose_compose(T, deltas + 7 * (size_t)k, guess); }
      const int rc = run_plain(k, guess);
      if (rc) return rc;
      if (n_done) *n_done = k + 1;
    }
    return SO_ICP_OK;
  }

  // ---------------- chained path
  hipStream_t s = c->stream;
  if (!c->seq_stream) HIP_TRY(c, hipStreamCreateWithFlags(&c->seq_stream, hipStreamNonBlocking));
  struct Drain {  // an early return must not leave copies reading the caller's buffers, nor launches of this call in the queue
    so_icp_ctx* c; bool ok = false;
    ~Drain() { if (!ok) { (void)hipStreamSynchronize(c->seq_stream); (void)hipStreamSynchronize(c->stream); (void)hipGetLastError(); c->ev_used = 0; } }
  } drain{c};
  size_t n_max = 0;
  for (int k = 0; k < count; ++k) {
    if (n_points[k] >= ((size_t)1 << 21)) return fail(c, SO_ICP_E_UNSUPPORTED, "scan of 2^21 points or more: the work-list counters hold 21 bits each (chunk descriptors 26)");
    n_max = std::max(n_max, n_points[k]);
  }
  { const int rc = reserve_scan_buffers(c, n_max); if (rc) return rc; }  // (once, for the longest scan: nothing is re-allocated under a registration in flight)
  { const int rc = upload_map(c); if (rc) return rc; }                   // (the binning ahead of scan 0 reads the map view before the first prepare())
  const int max_outer = std::min(c->cfg.max_iterations > 0 ? c->cfg.max_iterations : 4, SO_ICP_MAX_OUTER);
  const int lm_max = std::min(c->cfg.lm_max_iterations > 0 ? c->cfg.lm_max_iterations : 4, 16);
  const int max_sf = c->cfg.max_surface_features;
  DevState* ds = c->d_state;
  CorrBuffers corr{c->d_nd.as<double4>(), c->d_coeff.as<double>(), c->d_status.as<uint8_t>()};
  std::vector<SeqRun> runs((size_t)count);
  // scan 0 may already be in HBM with its work list: the call before this one staged it beside its last registration (so_icp_sequence_announce_next)
  so_icp_ctx::SeqNext adopted = c->seq_next;
  const bool adopt = adopted.staged && adopted.scan == scans[0] && adopted.n == n_points[0];
  const int slot_base = adopt ? adopted.slot : 0;
  c->seq_next.staged = false;
  for (int k = 0; k < count; ++k) {
    SeqRun& r = runs[(size_t)k];
    r.n = n_points[k];
    const size_t kept_upper = (max_sf >= 0 && r.n > (size_t)max_sf) ? (size_t)max_sf + 2 : r.n;
    r.query_waves = c->query_waves && r.n && kept_upper <= kQueryWaveMaxKept;
    r.ring = (k & 1) * 2;
    if (!scans_on_device || !r.query_waves) r.slot = &c->seq_slot[(slot_base + k) % so_icp_ctx::kStageSlots];
  }
  // the scan's way to HBM and its work list, on the sequence's own queue: copy (host scans), scan_keys -> bin_offsets -> bin_place under
  // `pose` (scans swept in chunks), one event.  Slot k % 3: its last user, scan k - 3, was collected before scan k - 1 was enqueued.
  auto stage_scan = [&](int k, const double pose[7]) -> int {
    SeqRun& r = runs[(size_t)k];
    r.d_scan = static_cast<const float*>(scans[k]);
    r.binned = false; r.needs_event = false;
    if (!r.slot || !r.n) { if (!scans_on_device) r.d_scan = nullptr; return SO_ICP_OK; }
    so_icp_ctx::StageSlot& sl = *r.slot;
    if (!sl.ev) HIP_TRY(c, hipEventCreateWithFlags(&sl.ev, hipEventDisableTiming));
    if (!scans_on_device) {
      if (!r.copied) {
        HIP_TRY(c, sl.dev.reserve((r.n + 64) * 12));
        HIP_TRY(c, hipMemcpyAsync(sl.dev.p, scans[k], r.n * 12, hipMemcpyHostToDevice, c->seq_stream));
        r.copied = true;
      }
      r.d_scan = sl.dev.as<float>();
      r.needs_event = true;
    }
    if (!r.query_waves) {
      if (!c->prebin) return SO_ICP_OK;  // (binned by the registration itself: never chained)
      const uint32_t lg = prebin_table_log2(r.n);
      const size_t m = r.n + 256, T = (size_t)1 << lg;
      HIP_TRY(c, sl.pb_keys.reserve(m * 4)); HIP_TRY(c, sl.pb_vals.reserve(m * 4)); HIP_TRY(c, sl.pb_chunks.reserve(m * 4));
      HIP_TRY(c, sl.pb_binned.reserve(m * 16)); HIP_TRY(c, sl.pb_ctr.reserve(64));
      if (c->d_sbin_key.cap < T * 4 || c->sbin_log2 != lg) {
        HIP_TRY(c, c->d_sbin_key.reserve(T * 4)); HIP_TRY(c, c->d_sbin_cnt.reserve(T * 4)); HIP_TRY(c, c->d_sbin_off.reserve(T * 4));
        HIP_TRY(c, hipMemsetAsync(c->d_sbin_key.p, 0xFF, T * 4, c->seq_stream)); HIP_TRY(c, hipMemsetAsync(c->d_sbin_cnt.p, 0, T * 4, c->seq_stream));
        c->sbin_log2 = lg;
      }
      const BinTable bt{c->d_sbin_key.as<uint32_t>(), c->d_sbin_cnt.as<uint32_t>(), c->d_sbin_off.as<uint32_t>(), lg};
      sl.pb_chunk_cap = (uint32_t)(sl.pb_chunks.cap / 4);
      launch_scan_keys(r.d_scan, (uint32_t)r.n, ds, pose, 0, 0, c->d_hist, c->view, max_sf, 0, 1, sl.pb_keys.as<uint32_t>(), sl.pb_vals.as<uint32_t>(), nullptr,
                       bt, c->seq_stream, false, nullptr, 0, false, 0, sl.pb_ctr.as<unsigned long long>());
      launch_bin_offsets(bt, sl.pb_chunks.as<uint32_t>(), sl.pb_chunk_cap, ds, c->seq_stream, nullptr, 0, sl.pb_ctr.as<unsigned long long>());
      launch_bin_place(bt, r.d_scan, (uint32_t)r.n, sl.pb_keys.as<uint32_t>(), sl.pb_vals.as<uint32_t>(), sl.pb_binned.as<float4>(), c->seq_stream);
      if (hipGetLastError() != hipSuccess) { c->sbin_log2 = 0; return fail(c, SO_ICP_E_HIP, "so_icp_register_sequence: the binning launches were refused"); }
      r.binned = true; r.needs_event = true;
      r.d_binned = sl.pb_binned.as<float4>(); r.d_chunks = sl.pb_chunks.as<uint32_t>();
    }
    if (r.needs_event) HIP_TRY(c, hipEventRecord(sl.ev, c->seq_stream));
    return SO_ICP_OK;
  };
  // The copy of a host scan TWO registrations ahead (its slot's last user, scan k - 3, has been collected): the binning of scan k + 1 is then
  // enqueued with its scan long in HBM and runs beside the FIRST SWEEP of registration k -- hundreds of short wavefronts next to a sweep that
  // leaves 60 % of its issue slots empty -- instead of behind a 34 us copy, beside the first solve, whose one wavefront per SIMD it slowed
  // by ~5 us (41 - 43 us against 36 - 37 for the second solve of the same registration: profiles/r06/sequence_timeline_flag_wait.txt).
  auto copy_ahead = [&](int k) -> int {
    if (k >= count || scans_on_device) return SO_ICP_OK;
    SeqRun& r = runs[(size_t)k];
    if (!r.slot || !r.n || r.copied) return SO_ICP_OK;
    so_icp_ctx::StageSlot& sl = *r.slot;
    HIP_TRY(c, sl.dev.reserve((r.n + 64) * 12));
    HIP_TRY(c, hipMemcpyAsync(sl.dev.p, scans[k], r.n * 12, hipMemcpyHostToDevice, c->seq_stream));
    r.copied = true;
    return SO_ICP_OK;
  };
  // host side of a registration's start (register_core_once): window, map view, parameters.  false + rc == 0: cannot be started this way
  auto prepare = [&](int k, const double guess[7], bool chained, int* rc_out) -> bool {
    SeqRun& r = runs[(size_t)k];
    *rc_out = SO_ICP_OK;
    std::memcpy(r.guess, guess, sizeof(r.guess));
    if (!r.query_waves && !r.binned) return false;
    if (!c->no_map_shift) {
      if (chained && !cube_stable(c, guess, 1.0)) return false;
      map_shift(c, guess, r.pos); std::memcpy(c->last_pos, r.pos, sizeof(r.pos));
    } else std::memcpy(r.pos, c->last_pos, sizeof(r.pos));
    r.count_5x5 = map_count_5x5(c, r.pos);
    if (!(r.count_5x5 > 50)) { if (!chained) *rc_out = SO_ICP_NOT_ENOUGH_MAP_FEATURES; return false; }  // LidarSlam.cpp:113-116
    if ((*rc_out = upload_map(c))) return false;
    const float plane_res_now = map_plane_res(c);
    r.mp = match_params(plane_res_now, 0);
    r.mp.chunk_cap = r.binned ? r.slot->pb_chunk_cap : 0;
    r.mp.pack_light = (c->knn_pack && c->knn_pack_hold == 0 && !c->knn_list_fits) ? 1 : 0;
    r.mp.packed_leftover = &ds->packed_leftover;
    if (c->knn_pack_hold > 0) --c->knn_pack_hold;
    r.ep = eval_params(plane_res_now, c->cfg.tukey_variant, 0);
    r.seq_base = (++c->reg_counter) << 8;
    r.ep.hring[0] = c->d_ring[r.ring]; r.ep.hring[1] = c->d_ring[r.ring + 1]; r.ep.seq_base = r.seq_base;
    r.ep.n_queries = (uint32_t)r.n; r.ep.q_stride = 3; r.ep.defer_publish = 0;
    r.mp.hring[0] = r.ep.hring[0]; r.mp.hring[1] = r.ep.hring[1]; r.mp.seq_base = r.seq_base; r.mp.publish_prev = 0;
    r.chained = chained;
    // (every seventh registration -- a period coprime to the scan rotation of the benchmarks: an event pair on a dispatch was measured at
    //  ~8 us of queue time here, where no idle moment between back-to-back chained launches hides it: 4 % of the rate at every third)
    r.timed = c->cfg.time_kernels == 1 && (k % 7) == 0;
    r.knn_ev.clear();
    r.chain_expect = chained ? c->done_count_seen + 1u : 0u;  // (exactly the registration in front of this one completes in between)
    r.mp.chain_expect = r.chain_expect; r.ep.chain_expect = r.chain_expect;
    return true;
  };
  // outer iterations [it0, it1) of run k into the queue; `last_publishes`: the solve of it1 - 1 reports by itself (nothing of this
  // registration is enqueued behind it yet), the others leave their report to the sweep behind them (EvalParams::defer_publish)
  auto enqueue = [&](int k, int it0, int it1) -> int {
    SeqRun& r = runs[(size_t)k];
    // Scan and work list come from the other queue.  The host WATCHES that queue's event for the scan (the copy went out a registration
    // ago, the binning a moment ago: tens of microseconds, and the registration in front has only just begun) and enqueues this
    // registration's launches once it has fired -- then nothing has to order the two queues on the device.  Measured alternatives: a
    // barrier packet in front of the first launch (hipStreamWaitEvent) costs 5.6 us of command-processor time between two registrations;
    // a first launch that polls a flag in device memory costs nothing -- and deadlocks when it is dispatched before the binning
    // kernels it waits for (its 4 096 spinning wavefronts fill the chip: seen once, on a first call whose allocations had held the host up).
    if (it0 == 0 && r.needs_event) {
      const auto t_w = std::chrono::steady_clock::now();
      bool fired = false;
      for (unsigned spin = 0;; ++spin) {
        const hipError_t q = hipEventQuery(r.slot->ev);
        if (q == hipSuccess) { fired = true; break; }
        if (q != hipErrorNotReady) break;
        if ((spin & 15u) == 15u && std::chrono::steady_clock::now() - t_w > std::chrono::microseconds(400)) break;
      }
      (void)hipGetLastError();
      if (!fired) HIP_TRY(c, hipStreamWaitEvent(s, r.slot->ev, 0));  // (the other queue is late: let the device order the two)
    }
    for (int it = it0; it < it1; ++it) {
      MatchParams mp_it = r.mp;
      mp_it.publish_prev = (it > it0) ? 1 : 0;  // (the solve in front of this sweep deferred its report)
      hipEvent_t ka = nullptr, kb = nullptr;
      if (r.timed) {  // (the events ride on the dispatch packet, no marker packets)
        ka = next_event(c); kb = next_event(c);
        if (ka && kb) r.knn_ev.push_back(SeqRun::KnnEv{it, ka, kb}); else ka = kb = nullptr;
      }
      if (r.query_waves) {
        launch_knn_query_waves(r.d_scan, (uint32_t)r.n, ds, r.guess, max_outer, lm_max, it == 0, c->d_hist, c->view, mp_it, max_sf, c->d_status.as<uint8_t>(),
                               c->d_nbr5.as<uint32_t>(), s, ka, kb, it == 0 ? r.chain_expect : 0u);
      } else {
        if (it == 0) {
          mp_it.begin = 1; mp_it.begin_args.max_outer = max_outer; mp_it.begin_args.lm_max = lm_max; mp_it.begin_max_surface_features = max_sf;
          mp_it.begin_n = (uint32_t)r.n; std::memcpy(mp_it.begin_args.pose, r.guess, sizeof(mp_it.begin_args.pose));
          mp_it.begin_args.chain_expect = r.chain_expect; mp_it.begin_args.pad = 0;
          mp_it.begin_ctr = r.slot->pb_ctr.as<unsigned long long>(); mp_it.begin_state = ds;
        }
        launch_knn_plane(r.d_binned, r.d_chunks, ds, c->view, mp_it, corr, c->d_nbr5.as<uint32_t>(), c->d_hist, s, ka, kb);
      }
      EvalParams ep_it = r.ep;
      ep_it.defer_publish = (it + 1 < it1) ? 1 : 0;
      ep_it.epoch_base = (++c->solve_launches) << 5;
      if (k + 1 < count) { ep_it.chain_next = 1; std::memcpy(ep_it.chain_delta, deltas + 7 * (size_t)(k + 1), sizeof(ep_it.chain_delta)); }  // (whichever solve ends this registration forms the next guess)
      launch_solve(lm_max, r.d_scan, r.d_scan + 1, r.d_scan + 2, corr, ds, ep_it, c->d_partials, c->d_ticket, c->d_hist, c->d_sums, c->view,
                   c->d_nbr5.as<uint32_t>(), r.mp, (uint32_t)r.n, (uint32_t)c->n_cus, s);
    }
    HIP_TRY(c, hipGetLastError());
    r.enq_iters = it1; r.enqueued = true;
    return SO_ICP_OK;
  };
  auto await = [&](int k, int it) -> int {
    SeqRun& r = runs[(size_t)k];
    volatile unsigned long long* seq = &c->h_ring[r.ring + (it & 1)]->seq;
    const unsigned long long want = r.seq_base | (unsigned long long)(it + 1);
    auto next_check = std::chrono::steady_clock::now() + std::chrono::milliseconds(5);
    for (unsigned spin = 1;; ++spin) {
      if (*seq == want) break;
      if ((spin & 0x3FFu) != 0) continue;
      const auto now = std::chrono::steady_clock::now();
      if (now < next_check) continue;
      next_check = now + std::chrono::milliseconds(1);
      if (hipStreamQuery(s) != hipErrorNotReady) {  // everything enqueued has completed and the report is not there
        (void)hipGetLastError();
        if (*seq == want) break;
        HIP_TRY(c, hipStreamSynchronize(s));
        if (*seq == want) break;
        return fail(c, SO_ICP_E_HIP, "so_icp_register_sequence: registration state was not published by the device");
      }
    }
    std::atomic_thread_fence(std::memory_order_acquire);
    return SO_ICP_OK;
  };
  const int depth0 = std::max(1, std::min(c->seq_depth, max_outer));
  int rc = SO_ICP_OK;
  // scan 0: an ordinary start from pose0
  if (adopt) {
    SeqRun& r0 = runs[0];
    r0.d_scan = adopted.d_scan; r0.binned = adopted.binned; r0.needs_event = adopted.needs_event;
    if (r0.binned) { r0.d_binned = r0.slot->pb_binned.as<float4>(); r0.d_chunks = r0.slot->pb_chunks.as<uint32_t>(); }
  } else if ((rc = stage_scan(0, pose0))) return rc;
  if ((rc = copy_ahead(1))) return rc;
  bool started = prepare(0, pose0, false, &rc);
  if (rc) return rc;
  if (started && (rc = enqueue(0, 0, depth0))) return rc;
  for (int k = 0; k < count; ++k) {
    SeqRun& r = runs[(size_t)k];
    so_icp_stats* st = &stats[k];
    double* pose_out = poses_out + 7 * (size_t)k;
    if (!started) {
      // this scan cannot be started from here (an empty scan, too little map, a window about to roll, ...): the ordinary entry point,
      // from the guess the chain arithmetic gives -- same results, and the next scan starts a new chain
      double guess[7];
      if (k == 0) std::memcpy(guess, pose0, sizeof(guess)); else { double T[7]; chain_from(k - 1, T); pose_compose(T, deltas + 7 * (size_t)k, guess); }
      if ((rc = hipStreamSynchronize(c->seq_stream)) != hipSuccess) return fail(c, SO_ICP_E_HIP, "so_icp_register_sequence: copy queue");
      if ((rc = run_plain(k, guess))) return rc;
    } else {
      const auto t_icp = std::chrono::steady_clock::now();
      // the NEXT scan: on its way to HBM, binned under the host's prediction of its guess, and -- the point of this entry -- its
      // registration enqueued behind this one's launches
      bool next_started = false;
      if (k + 1 < count) {
        double pred[7];
        pose_compose(r.guess, deltas + 7 * (size_t)(k + 1), pred);  // (this registration will move r.guess by centimetres: good enough to bin under and to place the window)
        if ((rc = stage_scan(k + 1, pred))) return rc;
        if ((rc = copy_ahead(k + 2))) return rc;
        int prc = 0;
        next_started = prepare(k + 1, pred, true, &prc);
        if (prc) return prc;
        if (next_started && (rc = enqueue(k + 1, 0, std::max(1, std::min(c->seq_depth, max_outer))))) return rc;
      } else if (c->seq_next.announced && !scans_on_device) {
        // the scan that will start the NEXT call: on its way to HBM and binned beside this call's last registration
        so_icp_ctx::SeqNext& nx = c->seq_next;
        nx.announced = false;
        double pred[7];
        pose_compose(r.guess, nx.delta, pred);
        SeqRun rn;  // (a run record of its own, outside `runs`: the vector must not move under the references held here)
        nx.scan = nx.next_scan; nx.n = nx.next_n;
        rn.n = nx.n;
        const size_t kept_upper = (max_sf >= 0 && rn.n > (size_t)max_sf) ? (size_t)max_sf + 2 : rn.n;
        rn.query_waves = c->query_waves && rn.n && kept_upper <= kQueryWaveMaxKept;
        nx.slot = (slot_base + count) % so_icp_ctx::kStageSlots;
        rn.slot = &c->seq_slot[nx.slot];
        const void* one[1] = {nx.scan};
        {  // (stage_scan() of the lambda above, for a scan that is not in `scans`; a failure only means "not staged ahead")
          SeqRun& rr = rn;
          rr.d_scan = nullptr; rr.binned = false; rr.needs_event = false;
          so_icp_ctx::StageSlot& sl = *rr.slot;
          bool ok = rr.n > 0;
          if (ok && !sl.ev) ok = hipEventCreateWithFlags(&sl.ev, hipEventDisableTiming) == hipSuccess;
          if (ok) ok = sl.dev.reserve((rr.n + 64) * 12) == hipSuccess && hipMemcpyAsync(sl.dev.p, one[0], rr.n * 12, hipMemcpyHostToDevice, c->seq_stream) == hipSuccess;
          if (ok) { rr.d_scan = sl.dev.as<float>(); rr.needs_event = true; }
          if (ok && !rr.query_waves && c->prebin) {
            const uint32_t lg = prebin_table_log2(rr.n);
            const size_t m = rr.n + 256, T = (size_t)1 << lg;
            ok = sl.pb_keys.reserve(m * 4) == hipSuccess && sl.pb_vals.reserve(m * 4) == hipSuccess && sl.pb_chunks.reserve(m * 4) == hipSuccess &&
                 sl.pb_binned.reserve(m * 16) == hipSuccess && sl.pb_ctr.reserve(64) == hipSuccess && c->d_sbin_key.cap >= T * 4 && c->sbin_log2 == lg;
            if (ok) {
              const BinTable bt{c->d_sbin_key.as<uint32_t>(), c->d_sbin_cnt.as<uint32_t>(), c->d_sbin_off.as<uint32_t>(), lg};
              sl.pb_chunk_cap = (uint32_t)(sl.pb_chunks.cap / 4);
              launch_scan_keys(rr.d_scan, (uint32_t)rr.n, ds, pred, 0, 0, c->d_hist, c->view, max_sf, 0, 1, sl.pb_keys.as<uint32_t>(), sl.pb_vals.as<uint32_t>(), nullptr,
                               bt, c->seq_stream, false, nullptr, 0, false, 0, sl.pb_ctr.as<unsigned long long>());
              launch_bin_offsets(bt, sl.pb_chunks.as<uint32_t>(), sl.pb_chunk_cap, ds, c->seq_stream, nullptr, 0, sl.pb_ctr.as<unsigned long long>());
              launch_bin_place(bt, rr.d_scan, (uint32_t)rr.n, sl.pb_keys.as<uint32_t>(), sl.pb_vals.as<uint32_t>(), sl.pb_binned.as<float4>(), c->seq_stream);
              if (hipGetLastError() != hipSuccess) { c->sbin_log2 = 0; ok = false; } else rr.binned = true;
            }
          }
          if (ok && rr.needs_event) ok = hipEventRecord(sl.ev, c->seq_stream) == hipSuccess;
          (void)hipGetLastError();
          nx.staged = ok; nx.binned = rr.binned; nx.needs_event = rr.needs_event; nx.d_scan = rr.d_scan;
        }
      }
      // this registration's reports
      int last = 0;
      for (int it = 0;; ++it) {
        if ((rc = await(k, it))) return rc;
        last = it;
        if (c->h_ring[r.ring + (it & 1)]->reg_done || it + 1 >= max_outer) break;
        if (it + 1 >= r.enq_iters) {
          // it needs more outer iterations than were enqueued ahead: the chained registration behind it has found it unfinished
          // and turned itself off (DevState::done_count); one iteration at a time from here, the next scan starts again afterwards
          if (next_started) { next_started = false; runs[(size_t)k + 1].enqueued = false; c->timing.seq_chain_breaks++; }
          if ((rc = enqueue(k, it + 1, it + 2))) return rc;
        }
      }
      const DevState& H = *c->h_ring[r.ring + (last & 1)];
      c->h_state = c->h_ring[r.ring + (last & 1)];
      std::memset(st, 0, sizeof(*st));
      st->flags = (r.slot && !scans_on_device ? SO_ICP_FLAG_STAGED_SCAN : 0u) | (r.binned ? SO_ICP_FLAG_BINNED_AHEAD : 0u) |
                  (r.query_waves ? SO_ICP_FLAG_QUERY_WAVES : 0u) | (r.chained ? SO_ICP_FLAG_CHAINED : 0u);
      if (c->have_hist) uncertainty_from_hist(c->prev_obs_hist, st->uncertainty);  // LidarSlam.cpp:47
      st->pos_in_localmap[0] = r.pos[0]; st->pos_in_localmap[1] = r.pos[1]; st->pos_in_localmap[2] = r.pos[2];
      st->laser_cloud_surf_from_map_num = r.count_5x5;
      st->laser_cloud_surf_stack_num = (int32_t)r.n;
      st->startup_count = c->startup_count;
      double guess[7];
      std::memcpy(guess, H.pose_in, sizeof(guess));  // (what the device formed for a chained registration; the host's own argument otherwise)
      if (guesses_out) std::memcpy(guesses_out + 7 * (size_t)k, guess, sizeof(guess));
      const uint32_t packed_left = H.packed_leftover >= c->packed_leftover_seen ? H.packed_leftover - c->packed_leftover_seen : H.packed_leftover;
      c->packed_leftover_seen = H.packed_leftover;
      if (r.mp.pack_light) c->timing.knn_pack_registrations++;
      if (r.mp.pack_light && (double)packed_left > 0.03 * (double)r.n * (double)std::max(H.n_iterations, 1)) { c->knn_pack_hold = 32; c->timing.knn_pack_holds++; }
      if (!r.query_waves) c->knn_list_fits = ((H.bin_packed >> 21) & 0x1FFFFFull) + (H.bin_packed >> 42) <= (unsigned long long)kKnnBlocks * 4ull;
      c->done_count_seen = H.done_count;
      c->seq_depth = std::max(1, H.n_iterations);
      for (const SeqRun::KnnEv& e : r.knn_ev) {  // sweeps that did real work (a launch behind the converged iteration was a no-op); they ended long ago
        float ms = 0;
        if (e.it < H.n_iterations && hipEventElapsedTime(&ms, e.a, e.b) == hipSuccess) {
          c->timing.knn_ms_total += ms; c->timing.knn_launches++;
          c->timing.knn_queries += r.query_waves ? (int64_t)r.n : (int64_t)(H.bin_packed & 0x1FFFFFull); c->timing.knn_map_points += c->view.n_points;
        }
      }
      (void)hipGetLastError();
      fill_result(c, H, guess, st, pose_out, true);
      st->time_elapsed_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_icp).count();
      c->timing.registrations++;
      if (r.chained) {
        c->timing.seq_chained++;
        // the window was placed for the predicted guess: the actual one must lie in the same block (cube_stable saw to it)
        const int* o = map_origin(c);
        if (cube_coord(guess[0], o[0]) != r.pos[0] || cube_coord(guess[1], o[1]) != r.pos[1] || cube_coord(guess[2], o[2]) != r.pos[2])
          return fail(c, SO_ICP_E_HIP, "so_icp_register_sequence: a chained guess left the map block its window was placed for");
      }
      if (k + 1 < count && !next_started && !runs[(size_t)k + 1].enqueued) {
        // an ordinary start of the next scan, from the exact guess (after a broken chain: its copy and work list are where they were)
        SeqRun& nx = runs[(size_t)k + 1];
        double T[7], g[7];
        chain_from(k, T); pose_compose(T, deltas + 7 * (size_t)(k + 1), g);
        int prc = 0;
        next_started = prepare(k + 1, g, false, &prc);
        if (prc && prc != SO_ICP_NOT_ENOUGH_MAP_FEATURES) return prc;
        if (next_started && (rc = enqueue(k + 1, 0, std::max(1, std::min(c->seq_depth, max_outer))))) return rc;
        (void)nx;
      }
      started = next_started;
      if (n_done) *n_done = k + 1;
      continue;
    }
    // (after a plain registration: the next scan starts a chain of its own)
    if (n_done) *n_done = k + 1;
    started = false;
    if (k + 1 < count) {
      double T[7], g[7];
      chain_from(k, T); pose_compose(T, deltas + 7 * (size_t)(k + 1), g);
      if ((rc = stage_scan(k + 1, g))) return rc;
      int prc = 0;
      started = prepare(k + 1, g, false, &prc);
      if (prc && prc != SO_ICP_NOT_ENOUGH_MAP_FEATURES) return prc;
      if (started && (rc = enqueue(k + 1, 0, std::max(1, std::min(c->seq_depth, max_outer))))) return rc;
    }
  }
  c->scan_staged = false;
  c->ev_used = 0;  // (the timing events of this call are free again)
  drain.ok = true;
  return SO_ICP_OK;
}

int so_icp_stage_scan(so_icp_ctx* c, const float* xyz, size_t n, size_t stride_bytes) {
  if (!c || (!xyz && n)) return SO_ICP_E_INVALID;
  NEED_DEVICE(c);
  if (stride_bytes == 0) stride_bytes = 12;
  if (stride_bytes % 4 && stride_bytes != SIZE_MAX) return fail(c, SO_ICP_E_INVALID, "stride_bytes must be a multiple of 4");
  bool queued = false, dma_only = false;
  {
    // (this entry point may be called from ANOTHER thread than the registration calls -- the node's feature callback --,
    //  so everything it touches lives under stage_mu)
    std::unique_lock<std::mutex> lk(c->stage_mu);
    HIP_TRY(c, hipSetDevice(c->cfg.device_id));
    if (!c->stage_started) {
      HIP_TRY(c, hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
      c->stage_thread = std::thread(stage_worker, c);
      c->stage_started = true;
    }
    // The same buffer announced again supersedes its older copy (so_icp.h: this is also how a caller takes a staged buffer
    // back -- so_icp_stage_cancel); failed slots are recycled.
    for (so_icp_ctx::StageSlot& sl : c->stage) {
      if (sl.state == 1 && sl.src == xyz) c->stage_cv.wait(lk, [&] { return sl.state != 1; });
      if ((sl.state == 2 && sl.src == xyz) || sl.state == -1) { stage_finish_direct(sl); sl.src = nullptr; sl.state = 0; }
    }
    if (n == 0 && stride_bytes == SIZE_MAX) return SO_ICP_OK;  // (so_icp_stage_cancel: withdraw only)
    // Slot: an empty one; else a ready copy that was announced BEFORE the scan consumed last (its frame was skipped: it would
    // never be asked for again), oldest first; else, with no registration in flight, the oldest ready copy gives way (a caller
    // that announces scans it never registers cannot block the slots).  Otherwise every slot holds a scan that is still
    // needed -- one being registered, two announced ahead of it: SO_ICP_STAGE_DECLINED (soft: that scan is uploaded by its
    // own registration call).
    so_icp_ctx::StageSlot* pick = nullptr;
    bool in_flight = false;
    for (so_icp_ctx::StageSlot& sl : c->stage) { if (sl.state == 0 && !pick) pick = &sl; in_flight = in_flight || sl.state == 3; }
    if (!pick)
      for (so_icp_ctx::StageSlot& sl : c->stage)
        if (sl.state == 2 && (sl.seq < c->stage_consumed_seq || !in_flight) && (!pick || sl.seq < pick->seq)) pick = &sl;
    if (!pick) { c->timing.stage_declined++; return SO_ICP_STAGE_DECLINED; }
    so_icp_ctx::StageSlot& sl = *pick;
    stage_finish_direct(sl);  // (an evicted copy must have left its caller's buffer)
    sl.src = xyz; sl.n = n; sl.stride = stride_bytes; sl.err.clear(); sl.seq = ++c->stage_seq; sl.ev_pending = false;
    sl.prebinned = false;  // (the slot's work list belongs to the scan it held before)
    if (stride_bytes == 12 && n && host_range_registered(c, xyz, n * 12)) {
      // registered (pinned) host memory, packed xyz: no pack, no copy thread -- the DMA reads the caller's buffer itself
      sl.state = 0;  // (until the copy is enqueued: an error below leaves the slot empty)
      HIP_TRY(c, sl.dev.reserve((n + 64) * 12));
      stage_prebin_reserve(c, sl, n);  // (what the binning ahead of this scan will need: allocated here, off the registration's path)
      if (!sl.ev) HIP_TRY(c, hipEventCreateWithFlags(&sl.ev, hipEventDisableTiming));
      sl.state = 2;
      sl.deferred = true; sl.t_announced = std::chrono::steady_clock::now(); queued = true; dma_only = true;  // (the copy thread is the time-out)
      c->timing.staged_direct++;
    } else {
      sl.state = 1; queued = true;
      c->timing.staged_copied++;
    }
  }
  if (queued) {
    c->stage_pending.fetch_add(1, std::memory_order_release);
    // (a copy thread in its timed wait re-reads the slots when the time is up -- at most 300 us from now, the patience a DMA
    //  announcement is entitled to anyway: no wake-up for it; a scan for the copy thread itself, or a thread parked for good, is woken)
    if (c->stage_parked.load() && !(dma_only && c->stage_timed.load())) c->stage_cv.notify_all();
  }
  return SO_ICP_OK;
}

// Pin a host buffer of the caller (hipHostRegister): scans announced from inside it with stride 12 go to HBM by DMA straight
// from the buffer.  The node side keeps its feature clouds in a few such buffers.
int so_icp_stage_cancel(so_icp_ctx* c, const float* xyz) {
  if (!c || !xyz) return SO_ICP_E_INVALID;
  if (c->host_only || !c->stage_started) return SO_ICP_OK;
  return so_icp_stage_scan(c, xyz, 0, SIZE_MAX);
}

int so_icp_host_register(so_icp_ctx* c, const void* ptr, size_t bytes) {
  if (!c || !ptr || !bytes) return SO_ICP_E_INVALID;
  NEED_DEVICE(c);
  HIP_TRY(c, hipSetDevice(c->cfg.device_id));
  std::lock_guard<std::mutex> lk(c->stage_mu);
  if (host_range_registered(c, ptr, bytes)) return SO_ICP_OK;
  HIP_TRY(c, hipHostRegister(const_cast<void*>(ptr), bytes, hipHostRegisterDefault));
  c->host_ranges.push_back(so_icp_ctx::HostRange{static_cast<const char*>(ptr), bytes, false});
  return SO_ICP_OK;
}
// Pinned host memory from the runtime's own allocator (hipHostMalloc) for the caller's clouds: the fastest DMA source.
int so_icp_host_alloc(so_icp_ctx* c, size_t bytes, void** out) {
  if (!c || !out || !bytes) return SO_ICP_E_INVALID;
  NEED_DEVICE(c);
  HIP_TRY(c, hipSetDevice(c->cfg.device_id));
  void* p = nullptr;
  HIP_TRY(c, hipHostMalloc(&p, bytes));
  std::lock_guard<std::mutex> lk(c->stage_mu);
  c->host_ranges.push_back(so_icp_ctx::HostRange{static_cast<const char*>(p), bytes, true});
  *out = p;
  return SO_ICP_OK;
}
int so_icp_host_free(so_icp_ctx* c, void* ptr) { return so_icp_host_unregister(c, ptr); }
int so_icp_host_unregister(so_icp_ctx* c, const void* ptr) {
  if (!c || !ptr) return SO_ICP_E_INVALID;
  NEED_DEVICE(c);
  HIP_TRY(c, hipSetDevice(c->cfg.device_id));
  std::unique_lock<std::mutex> lk(c->stage_mu);
  for (size_t i = 0; i < c->host_ranges.size(); ++i) {
    if (c->host_ranges[i].p != static_cast<const char*>(ptr)) continue;
    for (so_icp_ctx::StageSlot& sl : c->stage) stage_finish_direct(sl);  // no copy may still be reading the buffer
    if (c->copy_stream) HIP_TRY(c, hipStreamSynchronize(c->copy_stream));
    if (c->host_ranges[i].owned) HIP_TRY(c, hipHostFree(const_cast<void*>(ptr))); else HIP_TRY(c, hipHostUnregister(const_cast<void*>(ptr)));
    c->host_ranges.erase(c->host_ranges.begin() + (long)i);
    return SO_ICP_OK;
  }
  return fail(c, SO_ICP_E_INVALID, "so_icp_host_unregister / so_icp_host_free: this pointer did not come from so_icp_host_register / so_icp_host_alloc");
}

int so_icp_register_batch(so_icp_ctx* c, const float* xyz, const void* d_scan, size_t n, size_t stride_bytes, const double* poses_in,
                          int n_hyp, double* poses_out, so_icp_stats* stats, int32_t* rc_out) {
  if (!c || !poses_in || !poses_out || n_hyp < 0 || (!xyz && !d_scan && n)) return SO_ICP_E_INVALID;
  NEED_DEVICE(c);
  HIP_TRY(c, hipSetDevice(c->cfg.device_id));
  if (c->cfg.world_size != 1)  // (hypotheses are independent: replicate the map and split THEM over the ranks -- bench.py's batch64)
    return fail(c, SO_ICP_E_UNSUPPORTED, "so_icp_register_batch needs the whole map on one device (world_size == 1)");
  if (n_hyp == 0) return 0;
  const float* scan = static_cast<const float*>(d_scan);
  if (!scan) {
    const int rc = upload_scan_impl(c, xyz, n, stride_bytes, c->d_scan_own);
    if (rc) return rc;
    scan = c->d_scan_own.as<float>();
  }
  // the map window is placed once, for hypothesis 0 (LidarSlam.cpp:363); every hypothesis sees that map
  int pos[3];
  map_shift(c, poses_in, pos);
  std::memcpy(c->last_pos, pos, sizeof(pos));
  int rc = upload_map(c);
  if (rc) return rc;
  if (n > 0 && c->batch_degrade < 2) {
    // batched kernels: groups of up to kBatchMaxConcurrent hypotheses advance together (kernels.hip, BatchView)
    const int count = map_count_5x5(c, pos);
    std::vector<int32_t> hrc((size_t)n_hyp, 0);
    for (int base = 0; base < n_hyp;) {
      // a group never holds more hypotheses than solve workgroups can be resident together (one workgroup per hypothesis at
      // least): on a device with few compute units -- SOICP_SOLVE_WORKGROUPS, a partitioned device, one workgroup per unit after
      // a failed co-residency wait -- the batch goes through in smaller groups instead of failing
      const int cap = (int)std::min<uint32_t>((uint32_t)kBatchMaxConcurrent, solve_batch_resident_blocks((uint32_t)c->n_cus, c->batch_degrade >= 1 ? 1 : 0));
      if (cap < 1) { c->batch_degrade = 2; break; }
      const int B = std::min(cap, n_hyp - base);
      rc = register_batch_group(c, scan, n, poses_in + 7 * (size_t)base, B, poses_out + 7 * (size_t)base, stats ? stats + base : nullptr,
                                hrc.data() + base, pos, count);
      // A batched solve launch needs its workgroups resident together.  If the device could not provide that (shared with another
      // process), the group is repeated with one workgroup per compute unit; if that fails too the context falls back to
      // concurrent sequential registrations (below) for the rest of its life.  so_icp_last_error keeps the notice.
      if (rc == kRetryWithoutPersistentSolve) {
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        c->batch.tables_clean = false;
        if (++c->batch_degrade <= 1) continue;  // (the same group again)
        break;
      }
      if (rc < 0) return rc;
      base += B;
    }
    if (c->batch_degrade < 2) {
      int ok = 0;
      for (int h = 0; h < n_hyp; ++h) { if (rc_out) rc_out[h] = hrc[(size_t)h]; if (hrc[(size_t)h] == SO_ICP_OK) ++ok; }
      return ok;
    }
  }
  // SOICP_BATCH_MODE=lanes, empty scans, and a device that cannot keep the batched solve resident: the hypotheses as concurrent
  // sequential registrations on worker contexts (own stream / buffers / state each, one launch per evaluation)
  const int lanes = std::max(1, std::min(16, n_hyp));
  // worker contexts: own stream / buffers / device state, no map of their own (they borrow this context's resident map)
  while ((int)c->workers.size() < lanes - 1) {
    so_icp_config wc = c->cfg;
    wc.time_kernels = 0;
    so_icp_ctx* w = so_icp_create(&wc);
    if (!w) return fail(c, SO_ICP_E_HIP, "so_icp_register_batch: worker context: " + g_create_error);
    w->dmap.reset();
    c->workers.push_back(w);
  }
  so_icp_ctx::Borrow bw;
  bw.on = true; bw.view = c->view; bw.plane_res = map_plane_res(c);
  std::memcpy(bw.pos, pos, sizeof(pos));
  bw.count_5x5 = map_count_5x5(c, pos);
  std::vector<so_icp_ctx*> lane_ctx(1, c);
  for (int l = 1; l < lanes; ++l) lane_ctx.push_back(c->workers[l - 1]);
  for (so_icp_ctx* w : lane_ctx) {
    w->borrow = bw; w->batch_mode = true; w->batch_single = (lanes == 1);
    std::memcpy(w->prev_obs_hist, c->prev_obs_hist, sizeof(c->prev_obs_hist));
    w->have_hist = c->have_hist; w->startup_count = c->startup_count;
    w->cfg.max_iterations = c->cfg.max_iterations; w->cfg.lm_max_iterations = c->cfg.lm_max_iterations;
    w->cfg.max_surface_features = c->cfg.max_surface_features;
  }
  std::vector<int> lane_rc(lanes, 0);
  std::vector<int> hyp_rc((size_t)n_hyp, 0);
  auto run_lane = [&](int l) {
    so_icp_ctx* w = lane_ctx[l];
    if (hipSetDevice(c->cfg.device_id) != hipSuccess) { lane_rc[l] = SO_ICP_E_HIP; return; }
    for (int h = l; h < n_hyp; h += lanes) {
      so_icp_stats local;
      const int r = register_core(w, scan, n, poses_in + 7 * (size_t)h, poses_out + 7 * (size_t)h, stats ? stats + h : &local);
      hyp_rc[h] = r;
      if (r < 0) { lane_rc[l] = r; return; }
    }
  };
  std::vector<std::thread> th;
  for (int l = 1; l < lanes; ++l) th.emplace_back(run_lane, l);
  run_lane(0);
  for (std::thread& t : th) t.join();
  int ok = 0, err = 0;
  for (int l = 0; l < lanes; ++l) {
    lane_ctx[l]->borrow.on = false; lane_ctx[l]->batch_mode = false;
    if (lane_rc[l] < 0 && !err) { err = lane_rc[l]; if (l > 0) c->err = "worker: " + lane_ctx[l]->err; }
  }
  for (int h = 0; h < n_hyp; ++h) { if (rc_out) rc_out[h] = hyp_rc[h]; if (hyp_rc[h] == SO_ICP_OK) ++ok; }
  return err ? err : ok;
}

// symmetric 3x3 eigen-decomposition (cyclic Jacobi), ascending eigenvalues, eigenvectors in the columns of V
static void eig3_host(const double A[9], double ev[3], double V[9]) {
  double a[9]; std::memcpy(a, A, sizeof(a));
  for (int i = 0; i < 9; ++i) V[i] = (i % 4 == 0) ? 1.0 : 0.0;
  for (int sweep = 0; sweep < 32; ++sweep) {
    const double off = a[1] * a[1] + a[2] * a[2] + a[5] * a[5];
    if (off <= 1e-300) break;
    for (int p = 0; p < 2; ++p)
      for (int q = p + 1; q < 3; ++q) {
        const double apq = a[3 * p + q];
        if (apq == 0.0) continue;
        const double theta = (a[3 * q + q] - a[3 * p + p]) / (2.0 * apq);
        const double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
        const double cs = 1.0 / std::sqrt(t * t + 1.0), sn = t * cs;
        for (int k = 0; k < 3; ++k) {  // A <- A G
          const double akp = a[3 * k + p], akq = a[3 * k + q];
          a[3 * k + p] = cs * akp - sn * akq; a[3 * k + q] = sn * akp + cs * akq;
        }
        for (int k = 0; k < 3; ++k) {  // A <- G^T A
          const double apk = a[3 * p + k], aqk = a[3 * q + k];
          a[3 * p + k] = cs * apk - sn * aqk; a[3 * q + k] = sn * apk + cs * aqk;
        }
        for (int k = 0; k < 3; ++k) {
          const double vkp = V[3 * k + p], vkq = V[3 * k + q];
          V[3 * k + p] = cs * vkp - sn * vkq; V[3 * k + q] = sn * vkp + cs * vkq;
        }
      }
  }
  int idx[3] = {0, 1, 2};
  for (int i = 0; i < 3; ++i) ev[i] = a[4 * i];
  for (int i = 0; i < 2; ++i)
    for (int j = 0; j < 2 - i; ++j)
      if (ev[idx[j]] > ev[idx[j + 1]]) std::swap(idx[j], idx[j + 1]);
  double e2[3], V2[9];
  for (int c2 = 0; c2 < 3; ++c2) { e2[c2] = ev[idx[c2]]; for (int r = 0; r < 3; ++r) V2[3 * r + c2] = V[3 * r + idx[c2]]; }
  std::memcpy(ev, e2, sizeof(e2)); std::memcpy(V, V2, sizeof(V2));
}

int so_icp_registration_error(const so_icp_stats* st, so_icp_registration_error_t* out) {
  if (!st || !out) return SO_ICP_E_INVALID;
  std::memset(out, 0, sizeof(*out));
  // (J^T J)^-1 by Cholesky: L L^T = H, then solve for the six unit vectors
  double L[36];
  for (int j = 0; j < 6; ++j) {
    double d = st->JtJ[6 * j + j];
    for (int k = 0; k < j; ++k) d -= L[6 * j + k] * L[6 * j + k];
    if (!(d > 0.0) || !std::isfinite(d)) return SO_ICP_E_INVALID;
    L[6 * j + j] = std::sqrt(d);
    for (int i = j + 1; i < 6; ++i) {
      double s = st->JtJ[6 * i + j];
      for (int k = 0; k < j; ++k) s -= L[6 * i + k] * L[6 * j + k];
      L[6 * i + j] = s / L[6 * j + j];
    }
  }
  for (int c2 = 0; c2 < 6; ++c2) {
    double z[6], x[6];
    for (int i = 0; i < 6; ++i) { double s = (i == c2) ? 1.0 : 0.0; for (int k = 0; k < i; ++k) s -= L[6 * i + k] * z[k]; z[i] = s / L[6 * i + i]; }
    for (int i = 5; i >= 0; --i) { double s = z[i]; for (int k = i + 1; k < 6; ++k) s -= L[6 * k + i] * x[k]; x[i] = s / L[6 * i + i]; }
    for (int r = 0; r < 6; ++r) out->covariance[6 * r + c2] = x[r];
  }
  for (int r = 0; r < 6; ++r)  // symmetrise the rounding
    for (int c2 = r + 1; c2 < 6; ++c2) { const double m = 0.5 * (out->covariance[6 * r + c2] + out->covariance[6 * c2 + r]); out->covariance[6 * r + c2] = out->covariance[6 * c2 + r] = m; }
  double P[9], O[9], ev[3], V[9];
  for (int r = 0; r < 3; ++r) for (int c2 = 0; c2 < 3; ++c2) { P[3 * r + c2] = out->covariance[6 * r + c2]; O[3 * r + c2] = out->covariance[6 * (r + 3) + c2 + 3]; }
  eig3_host(P, ev, V);
  out->position_error = std::sqrt(ev[2]);
  for (int r = 0; r < 3; ++r) out->position_error_direction[r] = V[3 * r + 2];
  out->pos_inverse_condition_num = std::sqrt(ev[0]) / std::sqrt(ev[2]);
  eig3_host(O, ev, V);
  out->orientation_error_deg = std::sqrt(ev[2]) * 180.0 / M_PI;
  for (int r = 0; r < 3; ++r) out->orientation_error_direction[r] = V[3 * r + 2];
  out->ori_inverse_condition_num = std::sqrt(ev[0]) / std::sqrt(ev[2]);
  return SO_ICP_OK;
}

int so_icp_localization(so_icp_ctx* c, int initialization, const double T_in[7], const float* xyz, size_t n, size_t stride_bytes,
                        double time_laser_odometry, double pose_out[7], so_icp_stats* st) {
  if (!c || !T_in || !pose_out || (!xyz && n)) return SO_ICP_E_INVALID;
  if (stride_bytes == 0) stride_bytes = 12;
  if (stride_bytes % 4) return fail(c, SO_ICP_E_INVALID, "stride_bytes must be a multiple of 4");
  if (!c->host_only) HIP_TRY(c, hipSetDevice(c->cfg.device_id));  // (the caller may sit on another device / thread)
  const size_t sf = stride_bytes / 4;
  auto transform_and_add = [&](const double T[7]) -> int {  // transformAndAddToMap, LidarSlam.cpp:60-80; TransformPoint, superodom_utils.h:119-123
    std::vector<float> w(n * 3);
    for (size_t i = 0; i < n; ++i) {
      double ox, oy, oz;
      quat_rotate<double>(T + 3, (double)xyz[i * sf], (double)xyz[i * sf + 1], (double)xyz[i * sf + 2], ox, oy, oz);
      w[3 * i] = (float)(ox + T[0]); w[3 * i + 1] = (float)(oy + T[1]); w[3 * i + 2] = (float)(oz + T[2]);
    }
    if (c->dmap) { const int r = c->dmap->add_surf_host(w.data(), n, 3, c->err); return r < 0 ? (r == -1 ? SO_ICP_E_NOMEM : SO_ICP_E_HIP) : exchange_map_counts(c); }
    return c->map.add_surf(w.data(), n, 3) < 0 ? fail(c, SO_ICP_E_INVALID, "LocalMap insert failed") : SO_ICP_OK;
  };
  auto transform_and_add_dev = [&](const float* d_scan, const double T[7]) -> int {  // same, entirely on the device
    // (one launch transforms the scan, finds every point's cube and lays the insert round out on the device; the insert is
    //  enqueued without a read-back and -- unless SOICP_MAP_FAST=sync -- completes behind this call: device_map.h, settle)
    if (const int rs = c->dmap->settle(c->err); rs < 0) return rs == -1 ? SO_ICP_E_NOMEM : SO_ICP_E_HIP;  // (before d_world may be re-allocated)
    HIP_TRY(c, c->d_world.reserve((n + 64) * 12));
    const int r = c->dmap->add_scan_dev(d_scan, n, T, c->d_world.as<float>(), c->dmap->defer_enabled() && !c->dmap->sharded(), c->err);
    return r < 0 ? (r == -1 ? SO_ICP_E_NOMEM : SO_ICP_E_HIP) : exchange_map_counts(c);
  };
  if (!initialization) {  // initializeMapping, LidarSlam.cpp:83-94
    std::memcpy(pose_out, T_in, 7 * sizeof(double));
    if (st) std::memset(st, 0, sizeof(*st));
    if (!c->host_only) drop_staged(c, xyz, n, stride_bytes);
    if (c->dmap) c->dmap->set_origin(T_in); else c->map.set_origin(T_in);
    const int r = transform_and_add(T_in);
    if (r) return r;
    c->last_time = time_laser_odometry;
    return SO_ICP_MAP_SEEDED;
  }
  so_icp_stats local;
  if (!st) st = &local;
  NEED_DEVICE(c);
  // (the scan the registration ran on -- staged slot or d_scan_own -- is still resident afterwards: the insert reuses it)
  const float* d_scan = nullptr;
  int rc = resolve_scan(c, xyz, n, stride_bytes, &d_scan);
  if (rc) return rc;
  rc = register_core(c, d_scan, n, T_in, pose_out, st);
  c->scan_staged = false;
  if (rc != SO_ICP_OK) { release_staged(c); return rc; }  // NOT_ENOUGH: the reference returns before the post-processing (LidarSlam.cpp:113-116)
  // checkMotionThresholds, LidarSlam.cpp:173-195: always accepts; only the startupCount side effect survives
  const double dt = time_laser_odometry - c->last_time;
  if (st->translation_from_last / dt > c->cfg.velocity_failure_threshold) c->startup_count = 5;
  st->startup_count = c->startup_count;
  int r;
  if (c->dmap) r = transform_and_add_dev(d_scan, pose_out);  // LidarSlam.cpp:163-167
  else r = transform_and_add(pose_out);
  release_staged(c);
  if (r) return r;
  c->last_time = time_laser_odometry;
  return SO_ICP_OK;
}

// featureExtraction::removePointDistortion, featureExtraction.cpp:223-314 (kernel: map_kernels.hip deskew_kernel)
static int deskew_core(so_icp_ctx* c, hipStream_t s, void* d_points, size_t n, size_t stride, size_t time_off, double t0, const so_icp_stamped_pose* poses,
                       size_t n_poses, int imu, const double T_i_l[7], so_icp_deskew_info* info) {
  static_assert(sizeof(so_icp_stamped_pose) == kStampedPoseDoubles * sizeof(double), "stamped pose = 8 doubles");
  const double* tab = reinterpret_cast<const double*>(poses);
  for (size_t k = 0; k + 1 < n_poses; ++k)
    if (!(poses[k].time < poses[k + 1].time)) return fail(c, SO_ICP_E_INVALID, "pose buffer times must increase strictly (the reference keeps them in a std::map)");
  DeskewFrames f;
  f.imu = imu ? 1 : 0;
  if (T_i_l) { for (int k = 0; k < 3; ++k) f.i_l.t[k] = T_i_l[k]; for (int k = 0; k < 4; ++k) f.i_l.q[k] = T_i_l[3 + k]; }
  else { f.i_l.t[0] = f.i_l.t[1] = f.i_l.t[2] = 0; f.i_l.q[0] = f.i_l.q[1] = f.i_l.q[2] = 0; f.i_l.q[3] = 1; }
  f.l_i = rigid_inverse(f.i_l);  // parameter.cpp:193
  bool clamped_start = false;
  Rigid start = interpolated_pose(tab, (uint32_t)n_poses, t0, &clamped_start);  // :279
  if (imu) start.t[0] = start.t[1] = start.t[2] = 0;                             // extractPose, :231-235
  f.w_original_inv = rigid_inverse(start);
  const Rigid sensor = imu ? rigid_mul(start, f.i_l) : start;                    // :284-290
  if (info) {
    std::memset(info, 0, sizeof(*info));
    for (int k = 0; k < 4; ++k) info->q_w_original_l[k] = sensor.q[k];
    for (int k = 0; k < 3; ++k) info->t_w_original_l[k] = sensor.t[k];
  }
  if (!n) return SO_ICP_OK;
  std::vector<double> host_tab(tab, tab + n_poses * kStampedPoseDoubles);
  if (imu) for (size_t k = 0; k < n_poses; ++k) host_tab[k * 8 + 1] = host_tab[k * 8 + 2] = host_tab[k * 8 + 3] = 0.0;
  HIP_TRY(c, c->pf_small.reserve(host_tab.size() * sizeof(double) + 64));
  HIP_TRY(c, hipMemcpyAsync(c->pf_small.p, host_tab.data(), host_tab.size() * sizeof(double), hipMemcpyHostToDevice, s));
  uint32_t* d_cnt = reinterpret_cast<uint32_t*>(c->pf_small.as<uint8_t>() + host_tab.size() * sizeof(double));
  HIP_TRY(c, hipMemsetAsync(d_cnt, 0, 8, s));
  launch_deskew(static_cast<uint8_t*>(d_points), (uint32_t)n, (uint32_t)stride, (uint32_t)time_off, t0, c->pf_small.as<double>(), (uint32_t)n_poses, f, d_cnt, s);
  HIP_TRY(c, hipGetLastError());
  uint32_t cnt = 0;
  HIP_TRY(c, hipMemcpyAsync(&cnt, d_cnt, 4, hipMemcpyDeviceToHost, s));
  HIP_TRY(c, hipStreamSynchronize(s));  // also keeps host_tab alive until the upload has been consumed
  if (info) info->n_clamped = cnt;
  return SO_ICP_OK;
}

static int deskew_check(so_icp_ctx* c, const void* points, size_t n, size_t stride, size_t time_off, const so_icp_stamped_pose* poses, size_t n_poses) {
  if (!c || (!points && n) || !poses || !n_poses) return SO_ICP_E_INVALID;
  if (stride < 16 || stride % 4 || time_off % 4 || time_off < 12 || time_off + 4 > stride)
    return fail(c, SO_ICP_E_INVALID, "records: x y z at 0 4 8, a float time at a 4-byte aligned offset in [12, stride - 4], stride a multiple of 4");
  if (n >= ((size_t)1 << 31) || n_poses >= ((size_t)1 << 24)) return fail(c, SO_ICP_E_UNSUPPORTED, "too many points / poses");
  return SO_ICP_OK;
}

int so_icp_deskew_scan_dev(so_icp_ctx* c, void* d_points, size_t n, size_t stride, size_t time_off, double t0, const so_icp_stamped_pose* poses,
                           size_t n_poses, int imu, const double T_i_l[7], so_icp_deskew_info* info) {
  const int rc = deskew_check(c, d_points, n, stride, time_off, poses, n_poses);
  if (rc) return rc;
  NEED_DEVICE(c);
  HIP_TRY(c, hipSetDevice(c->cfg.device_id));
  return deskew_core(c, c->stream, d_points, n, stride, time_off, t0, poses, n_poses, imu, T_i_l, info);  // (the caller's device buffer: its queue)
}

int so_icp_deskew_scan(so_icp_ctx* c, void* points, size_t n, size_t stride, size_t time_off, double t0, const so_icp_stamped_pose* poses,
                       size_t n_poses, int imu, const double T_i_l[7], so_icp_deskew_info* info) {
  int rc = deskew_check(c, points, n, stride, time_off, poses, n_poses);
  if (rc) return rc;
  NEED_DEVICE(c);
  HIP_TRY(c, hipSetDevice(c->cfg.device_id));
  hipStream_t s = aux_stream(c);
  if (n) {
    HIP_TRY(c, c->pf_in.reserve(n * stride + 64));
    HIP_TRY(c, hipMemcpyAsync(c->pf_in.p, points, n * stride, hipMemcpyHostToDevice, s));
  }
  rc = deskew_core(c, s, c->pf_in.p, n, stride, time_off, t0, poses, n_poses, imu, T_i_l, info);
  if (rc || !n) return rc;
  HIP_TRY(c, hipMemcpyAsync(points, c->pf_in.p, n * stride, hipMemcpyDeviceToHost, s));
  HIP_TRY(c, hipStreamSynchronize(s));
  return SO_ICP_OK;
}

// laserMapping::publishTopic's registered scan, laserMapping.cpp:464-493 (kernel: map_kernels.hip transform_cloud_kernel)
int so_icp_transform_cloud(so_icp_ctx* c, void* points, size_t n, size_t stride, const double T[7], uint8_t* keep, size_t* n_kept) {
  if (!c || (!points && n) || !T) return SO_ICP_E_INVALID;
  if (stride < 12 || stride % 4) return fail(c, SO_ICP_E_INVALID, "records: float x y z at 0 4 8, stride a multiple of 4");
  if (n >= ((size_t)1 << 31)) return fail(c, SO_ICP_E_UNSUPPORTED, "too many points");
  NEED_DEVICE(c);
  HIP_TRY(c, hipSetDevice(c->cfg.device_id));
  if (n_kept) *n_kept = 0;
  if (!n) return SO_ICP_OK;
  hipStream_t s = aux_stream(c);
  HIP_TRY(c, c->pf_in.reserve(n * stride + 64));
  HIP_TRY(c, c->pf_flags.reserve(n + 64));
  HIP_TRY(c, c->pf_small.reserve(256));
  if (!c->h_pf_kept) HIP_TRY(c, hipHostMalloc(reinterpret_cast<void**>(&c->h_pf_kept), 64));
  HIP_TRY(c, hipMemcpyAsync(c->pf_in.p, points, n * stride, hipMemcpyHostToDevice, s));
  HIP_TRY(c, hipMemsetAsync(c->pf_small.p, 0, 8, s));
  launch_transform_cloud(c->pf_in.as<uint8_t>(), (uint32_t)n, (uint32_t)stride, pose_from_array(T), c->pf_flags.as<uint8_t>(), c->pf_small.as<uint32_t>(), s);
  HIP_TRY(c, hipGetLastError());
  // the records and the count first (the count through a pinned word: a copy to pageable memory is staged and synchronised by the
  // runtime); the flags only when a point was dropped -- points within 0.1 m of the world origin, next to never (lmap.cpp:476) --: a
  // caller's std::vector of flags is pageable memory, and its copy cost as much as the records'
  HIP_TRY(c, hipMemcpyAsync(points, c->pf_in.p, n * stride, hipMemcpyDeviceToHost, s));
  HIP_TRY(c, hipMemcpyAsync(c->h_pf_kept, c->pf_small.p, 4, hipMemcpyDeviceToHost, s));
  HIP_TRY(c, hipStreamSynchronize(s));
  const uint32_t kept = *c->h_pf_kept;
  if (keep) {
    if (kept == (uint32_t)n) std::memset(keep, 1, n);
    else { HIP_TRY(c, hipMemcpyAsync(keep, c->pf_flags.p, n, hipMemcpyDeviceToHost, s)); HIP_TRY(c, hipStreamSynchronize(s)); }
  }
  if (n_kept) *n_kept = kept;
  return SO_ICP_OK;
}

int so_icp_download_scan(so_icp_ctx* c, const void* d_scan, size_t n, float* out_xyz) {
  if (!c || (!d_scan && n) || (!out_xyz && n)) return SO_ICP_E_INVALID;
  NEED_DEVICE(c);
  if (!n) return SO_ICP_OK;
  HIP_TRY(c, hipMemcpyAsync(out_xyz, d_scan, n * 12, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  return SO_ICP_OK;
}

int so_icp_localization_dev(so_icp_ctx* c, int initialization, const double T_in[7], const void* d_scan, size_t n,
                            double time_laser_odometry, double pose_out[7], so_icp_stats* st) {
  if (!c || !T_in || !pose_out || (!d_scan && n)) return SO_ICP_E_INVALID;
  NEED_DEVICE(c);
  HIP_TRY(c, hipSetDevice(c->cfg.device_id));
  if (!c->dmap) {  // host-side LocalMap (sharded ranks): the insert needs the points on the host
    std::vector<float> h(n * 3);
    const int rc = so_icp_download_scan(c, d_scan, n, h.data());
    if (rc) return rc;
    return so_icp_localization(c, initialization, T_in, h.data(), n, 12, time_laser_odometry, pose_out, st);
  }
  auto transform_and_add_dev = [&](const double T[7]) -> int {  // transformAndAddToMap (LidarSlam.cpp:60-80) on the device
    if (const int rs = c->dmap->settle(c->err); rs < 0) return rs == -1 ? SO_ICP_E_NOMEM : SO_ICP_E_HIP;
    HIP_TRY(c, c->d_world.reserve((n + 64) * 12));
    const int r = c->dmap->add_scan_dev(static_cast<const float*>(d_scan), n, T, c->d_world.as<float>(), c->dmap->defer_enabled() && !c->dmap->sharded(), c->err);
    return r < 0 ? (r == -1 ? SO_ICP_E_NOMEM : SO_ICP_E_HIP) : exchange_map_counts(c);
  };
  if (!initialization) {  // initializeMapping, LidarSlam.cpp:83-94
    std::memcpy(pose_out, T_in, 7 * sizeof(double));
    if (st) std::memset(st, 0, sizeof(*st));
    c->dmap->set_origin(T_in);
    const int r = transform_and_add_dev(T_in);
    if (r) return r;
    c->last_time = time_laser_odometry;
    return SO_ICP_MAP_SEEDED;
  }
  so_icp_stats local;
  if (!st) st = &local;
  const int rc = register_core(c, static_cast<const float*>(d_scan), n, T_in, pose_out, st);
  if (rc != SO_ICP_OK) return rc;
  const double dt = time_laser_odometry - c->last_time;  // checkMotionThresholds, LidarSlam.cpp:173-195
  if (st->translation_from_last / dt > c->cfg.velocity_failure_threshold) c->startup_count = 5;
  st->startup_count = c->startup_count;
  const int r = transform_and_add_dev(pose_out);  // LidarSlam.cpp:163-167
  if (r) return r;
  c->last_time = time_laser_odometry;
  return SO_ICP_OK;
}

// laserMapping::adjustVoxelSize (laserMapping.cpp:598-651) on the device: cloud statistics -> resolution choice ->
// pcl::VoxelGrid of the surf cloud at planeRes; the resolutions are pushed into the context like the node does.
// The pre-filter as ONE enqueue: statistics -> decision and leaf grid on the device (vg_decide_kernel) -> VoxelGrid -> one
// read-back (decision + number of leaves).  kPrefilterHostPath: a case the device leaves to the host (the statistic within
// the rounding band of a threshold, a leaf grid that overflows int32): the caller goes on with the host-decided sequence.
constexpr int kPrefilterHostPath = 1000;
static int prefilter_reserve_work(so_icp_ctx* c, size_t n) {
  const size_t cap = n + 1024;
  HIP_TRY(c, c->pf_w.reserve(cap * 16)); HIP_TRY(c, c->pf_s.reserve(cap * 16));
  for (DevBuf* b : {&c->pf_k0, &c->pf_k1, &c->pf_v0, &c->pf_v1, &c->pf_flags, &c->pf_pos, &c->pf_heads}) HIP_TRY(c, b->reserve((cap + 1) * 4));
  if (c->pf_temp_for != cap) { c->pf_temp_need = map_sort_temp_bytes(cap) + 256; c->pf_temp_for = cap; }
  HIP_TRY(c, c->pf_temp.reserve(c->pf_temp_need));
  HIP_TRY(c, c->pf_out.reserve((n + 64) * 12));
  return SO_ICP_OK;
}
static int prefilter_fast(so_icp_ctx* c, hipStream_t s, size_t n, uint32_t sf, int auto_voxel_size, float line_res, float plane_res,
                          so_icp_prefilter_info& li, void** d_out, size_t* n_out) {
  constexpr int kStatBlocks = 256;
  constexpr size_t kDecOff = 64, kPartOff = 512;
  static_assert(kDecOff + sizeof(VgDecision) <= kPartOff, "layout of pf_dec");
  constexpr uint32_t kScanRecords = 1024;  // look-back records of the filter's fused scan: 2 048 points each
  constexpr size_t kStateOff = kPartOff + kStatBlocks * 10 * sizeof(double);
  HIP_TRY(c, c->pf_dec.reserve(kStateOff + kScanRecords * sizeof(unsigned long long) + 64));
  if (!c->h_pf) HIP_TRY(c, hipHostMalloc(reinterpret_cast<void**>(&c->h_pf), sizeof(VgDecision)));
  uint32_t* d_counters = c->pf_dec.as<uint32_t>();
  VgDecision* d_dec = reinterpret_cast<VgDecision*>(c->pf_dec.as<uint8_t>() + kDecOff);
  double* d_part = reinterpret_cast<double*>(c->pf_dec.as<uint8_t>() + kPartOff);
  int rc = prefilter_reserve_work(c, n);
  if (rc) return rc;
  VgCandidates cand;
  cand.line_res[0] = 0.1f; cand.plane_res[0] = 0.2f;            // laserMapping.cpp:622-626
  cand.line_res[1] = line_res; cand.plane_res[1] = plane_res;
  cand.line_res[2] = 0.4f; cand.plane_res[2] = 0.8f;            // :627-631
  for (int k = 0; k < 3; ++k) cand.inv_leaf[k] = 1.0f / cand.plane_res[k];
  launch_vg_stats(c->pf_in.as<float>(), (uint32_t)n, sf, d_part, kStatBlocks, s);
  unsigned long long* d_state = reinterpret_cast<unsigned long long*>(c->pf_dec.as<uint8_t>() + kStateOff);
  launch_vg_decide(d_part, kStatBlocks, (uint32_t)n, auto_voxel_size, cand, d_dec, d_counters, d_state, kScanRecords, s);
  VoxelFilterArgs a{};
  a.d_decision = d_dec; a.scan_state = d_state; a.n_scan_state = kScanRecords;
  a.d_xyz = c->pf_in.as<float>(); a.n = (uint32_t)n; a.stride_floats = sf;
  a.wpts = c->pf_w.as<float4>(); a.spts = c->pf_s.as<float4>();
  a.keys0 = c->pf_k0.as<uint32_t>(); a.keys1 = c->pf_k1.as<uint32_t>(); a.vals0 = c->pf_v0.as<uint32_t>(); a.vals1 = c->pf_v1.as<uint32_t>();
  a.flags = c->pf_flags.as<uint32_t>(); a.pos = c->pf_pos.as<uint32_t>(); a.heads = c->pf_heads.as<uint32_t>();
  a.d_n_cent = d_counters; a.d_out = c->pf_out.as<float>();
  a.temp = c->pf_temp.p; a.temp_bytes = c->pf_temp.cap;
  launch_voxel_filter(a, s);
  HIP_TRY(c, hipGetLastError());
  HIP_TRY(c, hipMemcpyAsync(c->h_pf, d_dec, sizeof(VgDecision), hipMemcpyDeviceToHost, s));
  HIP_TRY(c, hipStreamSynchronize(s));
  const VgDecision& H = *c->h_pf;
  if (H.flags) return kPrefilterHostPath;
  if (auto_voxel_size) {
    li.statistic_in_input_order = 0;
    li.average_distance = (double)H.average_distance;
    li.count_far_points = (int32_t)H.acc[3];
    li.increase_blind_radius = li.count_far_points > 3000;
  }
  li.line_res = H.line_res; li.plane_res = H.plane_res;
  rc = so_icp_set_resolution(c, li.line_res, li.plane_res);  // lmap.cpp:648-649
  if (rc) return rc;
  *d_out = c->pf_out.p; *n_out = H.n_leaves;
  return SO_ICP_OK;
}

int so_icp_prefilter_announce(so_icp_ctx* c, const float* xyz, size_t n, size_t stride_bytes) {
  if (!c) return SO_ICP_E_INVALID;
  NEED_DEVICE(c);
  HIP_TRY(c, hipSetDevice(c->cfg.device_id));
  if (stride_bytes == 0) stride_bytes = 12;
  if (stride_bytes % 4) return fail(c, SO_ICP_E_INVALID, "stride_bytes must be a multiple of 4");
  hipStream_t s = aux_stream(c);
  std::lock_guard<std::mutex> lk(c->pf_mu);
  if (!xyz || !n) {  // withdrawn: a copy under way must have left the caller's buffer before the caller reuses it
    if (c->pf_announced.on) HIP_TRY(c, hipStreamSynchronize(s));
    c->pf_announced = so_icp_ctx::PfAnnounced{};
    return SO_ICP_OK;
  }
  // (the pre-filter's queue: whatever still reads pf_stage -- nothing does, a taken buffer became pf_in -- or writes it is in front of this copy)
  HIP_TRY(c, c->pf_stage.reserve(n * stride_bytes + 64));
  HIP_TRY(c, hipMemcpyAsync(c->pf_stage.p, xyz, n * stride_bytes, hipMemcpyHostToDevice, s));
  c->pf_announced.ptr = xyz; c->pf_announced.n = n; c->pf_announced.stride = stride_bytes; c->pf_announced.on = true;
  return SO_ICP_OK;
}

int so_icp_prefilter_scan(so_icp_ctx* c, const float* xyz, size_t n, size_t stride_bytes, int auto_voxel_size, float line_res,
                          float plane_res, void** d_out, size_t* n_out, so_icp_prefilter_info* info) {
  if (!c || (!xyz && n) || !d_out || !n_out) return SO_ICP_E_INVALID;
  NEED_DEVICE(c);
  HIP_TRY(c, hipSetDevice(c->cfg.device_id));
  if (stride_bytes == 0) stride_bytes = 12;
  if (stride_bytes % 4) return fail(c, SO_ICP_E_INVALID, "stride_bytes must be a multiple of 4");
  const uint32_t sf = (uint32_t)(stride_bytes / 4);
  // The pre-filter reads the caller's cloud and writes its own buffers: nothing the map insert of the previous frame (still in the
  // context's queue when Localization() returned) touches -- that insert's first kernel, the only reader of the previous filtered
  // cloud, had finished before Localization() returned.  On its own queue it runs beside the insert instead of behind it; the call
  // returns after its own read-back, so the registration that follows finds the filtered cloud complete.
  hipStream_t s = aux_stream(c);
  so_icp_prefilter_info li;
  std::memset(&li, 0, sizeof(li));
  li.line_res = line_res; li.plane_res = plane_res;
  *d_out = nullptr; *n_out = 0;
  if (!n) { if (info) *info = li; return so_icp_set_resolution(c, line_res, plane_res); }
  // raw cloud -> device (with its stride) -- unless it was announced (so_icp_prefilter_announce): then its copy went into the queue long
  // ago (34 us for a 131 072-point sweep, beside the registration of the frame before) and the two buffers change places
  bool announced = false;
  {
    std::lock_guard<std::mutex> lk(c->pf_mu);
    announced = c->pf_announced.on && c->pf_announced.ptr == (const void*)xyz && c->pf_announced.n == n && c->pf_announced.stride == stride_bytes &&
                c->pf_stage.p != nullptr;
    c->pf_announced.on = false;  // (taken, or not meant for this call: a copy still in this queue ends before this call's read-back does)
    if (announced) std::swap(c->pf_in, c->pf_stage);
  }
  if (!announced) {
    HIP_TRY(c, c->pf_in.reserve(n * stride_bytes + 64));
    HIP_TRY(c, hipMemcpyAsync(c->pf_in.p, xyz, n * stride_bytes, hipMemcpyHostToDevice, s));
  }
  if (c->pf_fast) {
    const int frc = prefilter_fast(c, s, n, sf, auto_voxel_size, line_res, plane_res, li, d_out, n_out);
    if (frc != kPrefilterHostPath) { li.reserved = announced ? 1 : 0; if (frc == SO_ICP_OK && info) *info = li; return frc; }
    std::memset(&li, 0, sizeof(li)); li.line_res = line_res; li.plane_res = plane_res;
  }
  // statistics + bounding box (fp64 tree sums; the reference accumulates |x|,|y|,|z| in float in input order --
  // the statistic only feeds the 25 / 65 thresholds and the 3000-far-points flag)
  constexpr int kStatBlocks = 256;
  HIP_TRY(c, c->pf_small.reserve(kStatBlocks * 10 * sizeof(double) + 128));
  launch_vg_stats(c->pf_in.as<float>(), (uint32_t)n, sf, c->pf_small.as<double>(), kStatBlocks, s);
  std::vector<double> part((size_t)kStatBlocks * 10);
  HIP_TRY(c, hipMemcpyAsync(part.data(), c->pf_small.p, part.size() * sizeof(double), hipMemcpyDeviceToHost, s));
  HIP_TRY(c, hipStreamSynchronize(s));
  double acc[10] = {0, 0, 0, 0, 3.0e38, 3.0e38, 3.0e38, -3.0e38, -3.0e38, -3.0e38};
  for (int b = 0; b < kStatBlocks; ++b)
    for (int k = 0; k < 10; ++k) {
      const double v = part[(size_t)b * 10 + k];
      acc[k] = k < 4 ? acc[k] + v : (k < 7 ? std::min(acc[k], v) : std::max(acc[k], v));
    }
  if (auto_voxel_size) {
    float ax = (float)(acc[0] / (double)n), ay = (float)(acc[1] / (double)n), az = (float)(acc[2] / (double)n);
    // The reference sums |x|, |y|, |z| in FLOAT in input order (laserMapping.cpp:604-611); the tree sums above are the exact sums
    // to ~1e-16.  A sequential float sum of n non-negative terms is within n 2^-24 of the exact one (relative), so the
    // reference's statistic lies within 3 n 2^-24 (+ the roundings of the divisions and the product) of this one: unless the
    // value is that close to a threshold, the resolution it chooses is decided.  Inside the band the reference's own
    // accumulation is run (one wavefront, ~3 ns per point) and ITS value decides -- and is reported.
    const double stat64 = (acc[0] / (double)n) * (acc[1] / (double)n) * (acc[2] / (double)n);
    const double band = 3.1 * (double)n * 5.9604644775390625e-8 + 1e-6;
    li.statistic_in_input_order = 0;
    if (std::fabs(stat64 - 25.0) <= 25.0 * band || std::fabs(stat64 - 65.0) <= 65.0 * band) {
      float* d3 = reinterpret_cast<float*>(c->pf_small.as<double>() + (size_t)kStatBlocks * 10);
      launch_vg_stats_inorder(c->pf_in.as<float>(), (uint32_t)n, sf, d3, s);
      float h3[3] = {0, 0, 0};
      HIP_TRY(c, hipMemcpyAsync(h3, d3, sizeof(h3), hipMemcpyDeviceToHost, s));
      HIP_TRY(c, hipStreamSynchronize(s));
      const float fn = (float)n;  // average /= laserCloudSurfLast->points.size()  (Eigen: the scalar becomes a float, one division per axis)
      ax = h3[0] / fn; ay = h3[1] / fn; az = h3[2] / fn;
      li.statistic_in_input_order = 1;
    }
    li.average_distance = (double)(ax * ay * az);       // laserMapping.cpp:620-621 (float product)
    li.count_far_points = (int32_t)acc[3];
    li.increase_blind_radius = li.count_far_points > 3000;
    if (li.average_distance < 25) { li.line_res = 0.1f; li.plane_res = 0.2f; }
    else if (li.average_distance > 65) { li.line_res = 0.4f; li.plane_res = 0.8f; }
  }
  int rc = so_icp_set_resolution(c, li.line_res, li.plane_res);  // lmap.cpp:648-649
  if (rc) return rc;
  // pcl::VoxelGrid::applyFilter: bounding box -> min_b / div_b; "leaf size too small" passes the cloud through
  const float leaf = li.plane_res, inv = 1.0f / leaf;
  const float mn[3] = {(float)acc[4], (float)acc[5], (float)acc[6]}, mx[3] = {(float)acc[7], (float)acc[8], (float)acc[9]};
  const int64_t dx = (int64_t)((mx[0] - mn[0]) * inv) + 1, dy = (int64_t)((mx[1] - mn[1]) * inv) + 1, dz = (int64_t)((mx[2] - mn[2]) * inv) + 1;
  HIP_TRY(c, c->pf_out.reserve((n + 64) * 12));
  if (dx * dy * dz > (int64_t)INT32_MAX) {
    if (sf == 3) HIP_TRY(c, hipMemcpyAsync(c->pf_out.p, c->pf_in.p, n * 12, hipMemcpyDeviceToDevice, s));
    else HIP_TRY(c, hipMemcpy2DAsync(c->pf_out.p, 12, c->pf_in.p, stride_bytes, 12, n, hipMemcpyDeviceToDevice, s));
    HIP_TRY(c, hipStreamSynchronize(s));
    *d_out = c->pf_out.p; *n_out = n;
    li.reserved = announced ? 1 : 0;
    if (info) *info = li;
    return SO_ICP_OK;
  }
  VoxelFilterArgs a{};
  for (int k = 0; k < 3; ++k) {
    a.min_b[k] = (int)std::floor(mn[k] * inv);
    a.div_b[k] = (int)std::floor(mx[k] * inv) - a.min_b[k] + 1;
  }
  const size_t cap = n + 1024;
  HIP_TRY(c, c->pf_w.reserve(cap * 16)); HIP_TRY(c, c->pf_s.reserve(cap * 16));
  for (DevBuf* b : {&c->pf_k0, &c->pf_k1, &c->pf_v0, &c->pf_v1, &c->pf_flags, &c->pf_pos, &c->pf_heads}) HIP_TRY(c, b->reserve((cap + 1) * 4));
  const size_t tb = map_sort_temp_bytes(cap) + 256;
  HIP_TRY(c, c->pf_temp.reserve(tb));
  HIP_TRY(c, hipMemsetAsync(c->pf_small.p, 0, 64, s));
  a.d_xyz = c->pf_in.as<float>(); a.n = (uint32_t)n; a.stride_floats = sf; a.inv_leaf = inv;
  a.wpts = c->pf_w.as<float4>(); a.spts = c->pf_s.as<float4>();
  a.keys0 = c->pf_k0.as<uint32_t>(); a.keys1 = c->pf_k1.as<uint32_t>(); a.vals0 = c->pf_v0.as<uint32_t>(); a.vals1 = c->pf_v1.as<uint32_t>();
  a.flags = c->pf_flags.as<uint32_t>(); a.pos = c->pf_pos.as<uint32_t>(); a.heads = c->pf_heads.as<uint32_t>();
  a.d_n_cent = c->pf_small.as<uint32_t>(); a.d_out = c->pf_out.as<float>();
  a.temp = c->pf_temp.p; a.temp_bytes = c->pf_temp.cap;
  launch_voxel_filter(a, s);
  uint32_t n_leaves = 0;
  HIP_TRY(c, hipMemcpyAsync(&n_leaves, c->pf_small.p, 4, hipMemcpyDeviceToHost, s));
  HIP_TRY(c, hipStreamSynchronize(s));
  *d_out = c->pf_out.p; *n_out = n_leaves;
  li.reserved = announced ? 1 : 0;
  if (info) *info = li;
  return SO_ICP_OK;
}

int so_icp_comm_unique_id(uint8_t id[SO_ICP_UNIQUE_ID_BYTES]) {
  if (!id) return SO_ICP_E_INVALID;
  Rccl r;
  std::string err;
  if (!rccl_load(r, err)) { g_create_error = err; return SO_ICP_E_RCCL; }
  ncclUniqueId u;
  std::memset(&u, 0, sizeof(u));
  const ncclResult_t rc = r.GetUniqueId(&u);
  if (rc != ncclSuccess) { g_create_error = "ncclGetUniqueId failed"; return SO_ICP_E_RCCL; }
  std::memcpy(id, &u, SO_ICP_UNIQUE_ID_BYTES);
  return SO_ICP_OK;
}

int so_icp_comm_init(so_icp_ctx* c, const uint8_t id[SO_ICP_UNIQUE_ID_BYTES]) {
  if (!c || !id) return SO_ICP_E_INVALID;
  NEED_DEVICE(c);
  if (!rccl_load(c->rccl, c->err)) return SO_ICP_E_RCCL;
  HIP_TRY(c, hipSetDevice(c->cfg.device_id));
  ncclUniqueId u;
  std::memcpy(&u, id, SO_ICP_UNIQUE_ID_BYTES);
  const ncclResult_t rc = c->rccl.CommInitRank(&c->comm, c->cfg.world_size, u, c->cfg.rank);
  if (rc != ncclSuccess) { c->comm = nullptr; return fail(c, SO_ICP_E_RCCL, std::string("ncclCommInitRank: ") + (c->rccl.GetErrorString ? c->rccl.GetErrorString(rc) : "?")); }
  return SO_ICP_OK;
}

int so_icp_comm_init_inprocess(so_icp_ctx* c, uint64_t group_key) {
  if (!c) return SO_ICP_E_INVALID;
  NEED_DEVICE(c);
  if (c->comm) return fail(c, SO_ICP_E_INVALID, "so_icp_comm_init_inprocess: the context already has an RCCL communicator");
  std::lock_guard<std::mutex> lk(g_groups_mu);
  std::shared_ptr<InprocGroup> g;
  for (auto& kv : g_groups) if (kv.first == group_key) g = kv.second;
  if (!g) {
    g = std::make_shared<InprocGroup>();
    if (const char* ev = std::getenv("SOICP_GROUP_TIMEOUT_S")) { const int t = std::atoi(ev); if (t >= 1) g->wait_seconds = t; }
    g->world = c->cfg.world_size; g->slot.resize((size_t)g->world);
    g_groups.emplace_back(group_key, g);
  }
  if (g->world != c->cfg.world_size) return fail(c, SO_ICP_E_INVALID, "so_icp_comm_init_inprocess: world_size differs from the group's");
  if (g->members >= g->world) return fail(c, SO_ICP_E_INVALID, "so_icp_comm_init_inprocess: the group is complete already (use a new key)");
  g->members++;
  c->group = g;
  return SO_ICP_OK;
}

// ---- peer exchange -------------------------------------------------------------------------------------------------
namespace {
struct PeerHandle { hipIpcMemHandle_t ipc; uint64_t pid; uint64_t ptr; };
static_assert(sizeof(PeerHandle) == SO_ICP_PEER_HANDLE_BYTES, "SO_ICP_PEER_HANDLE_BYTES");
}  // namespace

int so_icp_peer_export(so_icp_ctx* c, uint8_t handle[SO_ICP_PEER_HANDLE_BYTES]) {
  if (!c || !handle) return SO_ICP_E_INVALID;
  NEED_DEVICE(c);
  if (c->cfg.world_size < 2 || c->cfg.world_size > kPeerMaxWorld) return fail(c, SO_ICP_E_UNSUPPORTED, "peer exchange: world_size must be 2..8");
  HIP_TRY(c, hipSetDevice(c->cfg.device_id));
  PeerHandle h;
  std::memset(&h, 0, sizeof(h));
  if (!c->peer_own) {
    // memory another device writes while a kernel of this one polls it: uncached (else fine-grained) device memory
    hipError_t e = hipExtMallocWithFlags(&c->peer_own, kPeerInboxBytes, hipDeviceMallocUncached);
    if (e != hipSuccess) { (void)hipGetLastError(); e = hipExtMallocWithFlags(&c->peer_own, kPeerInboxBytes, hipDeviceMallocFinegrained); }
    if (e != hipSuccess) { (void)hipGetLastError(); e = hipMalloc(&c->peer_own, kPeerInboxBytes); }
    if (e != hipSuccess) { c->peer_own = nullptr; return fail(c, SO_ICP_E_HIP, std::string("peer exchange: inbox allocation: ") + hipGetErrorString(e)); }
  }
  // A (new) handshake starts from an empty inbox -- stale pass records and self-test chunks of an earlier connection must not
  // satisfy the polls of this one -- and from pass number zero on every rank.  The exchange of the handles that follows
  // is the barrier between these clears and the first remote store.
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  HIP_TRY(c, hipMemset(c->peer_own, 0, kPeerInboxBytes));
  HIP_TRY(c, hipMemset(&c->d_state->peer_seq, 0, sizeof(unsigned long long)));
  c->peer_on = false; c->peer_connected = false;
  if (hipIpcGetMemHandle(&h.ipc, c->peer_own) != hipSuccess) {
    (void)hipGetLastError();  // contexts of ONE process need no IPC handle (pid + pointer below); across processes connect() will refuse
    std::memset(&h.ipc, 0, sizeof(h.ipc));
  }
  h.pid = (uint64_t)getpid(); h.ptr = (uint64_t)(uintptr_t)c->peer_own;
  std::memcpy(handle, &h, sizeof(h));
  return SO_ICP_OK;
}

int so_icp_peer_connect(so_icp_ctx* c, const uint8_t* handles, int* self_test_ok) {
  if (!c || !handles || !self_test_ok) return SO_ICP_E_INVALID;
  NEED_DEVICE(c);
  *self_test_ok = 0;
  if (!c->peer_own) return fail(c, SO_ICP_E_INVALID, "so_icp_peer_connect: call so_icp_peer_export first");
  HIP_TRY(c, hipSetDevice(c->cfg.device_id));
  const int world = c->cfg.world_size;
  for (int r = 0; r < world; ++r) {
    PeerHandle h;
    std::memcpy(&h, handles + (size_t)r * SO_ICP_PEER_HANDLE_BYTES, sizeof(h));
    if (r == c->cfg.rank) { c->peer_inbox[r] = c->peer_own; continue; }
    if (h.pid == (uint64_t)getpid()) { c->peer_inbox[r] = (void*)(uintptr_t)h.ptr; continue; }  // same address space
    void* p = nullptr;
    const hipError_t e = hipIpcOpenMemHandle(&p, h.ipc, hipIpcMemLazyEnablePeerAccess);
    if (e != hipSuccess) { (void)hipGetLastError(); c->err = std::string("peer exchange: hipIpcOpenMemHandle(rank ") + std::to_string(r) + "): " + hipGetErrorString(e); return SO_ICP_OK; }  // self_test_ok stays 0
    c->peer_inbox[r] = p; c->peer_opened[r] = true;
  }
  c->peer_connected = true;
  // self-test with the very stores / loads of the solve's exchange (every rank runs it; waits up to 2 s for the others)
  int32_t* d_ok = reinterpret_cast<int32_t*>(c->d_fbcount);
  HIP_TRY(c, hipMemsetAsync(d_ok, 0, 4, c->stream));
  launch_peer_selftest(c->peer_inbox, c->cfg.rank, world, 0x7E570000u + (++c->peer_connects & 0xFFFFu), d_ok, c->stream);  // (every rank connects equally often)
  HIP_TRY(c, hipMemcpyAsync(c->h_u32, d_ok, 4, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  *self_test_ok = c->h_u32[0] == 1 ? 1 : 0;
  if (!*self_test_ok) c->err = "peer exchange: self-test chunks did not arrive from every rank";
  return SO_ICP_OK;
}

int so_icp_peer_enable(so_icp_ctx* c, int on) {
  if (!c) return SO_ICP_E_INVALID;
  NEED_DEVICE(c);
  if (on && !c->peer_connected) return fail(c, SO_ICP_E_INVALID, "so_icp_peer_enable: not connected");
  c->peer_on = on != 0;
  return SO_ICP_OK;
}

int so_icp_cells_per_cube(float plane_res, double* cell_size) { return cells_per_cube(plane_res, cell_size); }

int so_icp_shard_owner_of_point(const float p[3], const int origin[3], float plane_res, int world_size) {
  if (!p || !origin) return SO_ICP_E_INVALID;
  const int ci = cube_coord((double)p[0], origin[0]), cj = cube_coord((double)p[1], origin[1]), ck = cube_coord((double)p[2], origin[2]);
  if (!(ci >= 0 && ci < kMapW && cj >= 0 && cj < kMapH && ck >= 0 && ck < kMapD)) return 0;  // counted by rank 0
  double cell;
  const int nc = cells_per_cube(plane_res, &cell);
  const int w[3] = {ci - origin[0], cj - origin[1], ck - origin[2]};
  int g[3];
  for (int a = 0; a < 3; ++a) {
    const int v = (int)std::floor(((double)p[a] - (w[a] * kCube - kHalfCube)) * (1.0 / cell));
    g[a] = v < 0 ? 0 : (v >= nc ? nc - 1 : v);
  }
  return shard_owner_of_cell(w[0], w[1], w[2], g[0], g[1], g[2], world_size);
}

int so_icp_shard_histogram(const float* scan_xyz, size_t n, size_t stride_bytes, const double pose[7], const int origin[3], float plane_res,
                           int world_size, int64_t* counts) {
  if ((!scan_xyz && n) || !pose || !origin || !counts || world_size < 1) return SO_ICP_E_INVALID;
  if (stride_bytes == 0) stride_bytes = 12;
  if (stride_bytes % 4) return SO_ICP_E_INVALID;
  const size_t sf = stride_bytes / 4;
  for (int r = 0; r < world_size; ++r) counts[r] = 0;
  for (size_t i = 0; i < n; ++i) {  // the queries' world positions exactly as scan_keys_kernel forms them (LidarSlam.cpp:397-398, 728-731)
    double wx, wy, wz;
    quat_rotate<double>(pose + 3, (double)scan_xyz[i * sf], (double)scan_xyz[i * sf + 1], (double)scan_xyz[i * sf + 2], wx, wy, wz);
    const float q[3] = {(float)(wx + pose[0]), (float)(wy + pose[1]), (float)(wz + pose[2])};
    counts[so_icp_shard_owner_of_point(q, origin, plane_res, world_size)]++;
  }
  return SO_ICP_OK;
}

int so_icp_lm_begin(so_icp_lm_state* s, const double x0[7], const so_icp_sums* sums, int max_iterations, double next_pose[7]) {
  if (!s || !x0 || !sums || !next_pose) return SO_ICP_E_INVALID;
  LmState* S = reinterpret_cast<LmState*>(s);
  return lm_begin(*S, x0, *reinterpret_cast<const LmSums*>(sums), max_iterations, next_pose);
}
int so_icp_lm_feed(so_icp_lm_state* s, const so_icp_sums* sums, double next_pose[7]) {
  if (!s || !sums || !next_pose) return SO_ICP_E_INVALID;
  return lm_feed(*reinterpret_cast<LmState*>(s), *reinterpret_cast<const LmSums*>(sums), next_pose);
}
int so_icp_lm_result(const so_icp_lm_state* s, double pose[7], so_icp_iter_stats* st) {
  if (!s || !pose) return SO_ICP_E_INVALID;
  const LmState* S = reinterpret_cast<const LmState*>(s);
  std::memcpy(pose, S->x, 7 * sizeof(double));
  if (st) {
    st->num_surf_from_scan = (int32_t)S->count; st->lm_iterations = S->lm_iterations; st->num_successful_steps = S->num_successful;
    st->termination = S->termination; st->initial_cost = S->initial_cost; st->final_cost = S->x_cost;
  }
  return SO_ICP_OK;
}

int so_icp_get_timing(so_icp_ctx* c, so_icp_timing* t) { if (!c || !t) return SO_ICP_E_INVALID; *t = c->timing; return SO_ICP_OK; }
int so_icp_reset_timing(so_icp_ctx* c) { if (!c) return SO_ICP_E_INVALID; std::memset(&c->timing, 0, sizeof(c->timing)); return SO_ICP_OK; }
int so_icp_set_time_kernels(so_icp_ctx* c, int mode) {
  if (!c || mode < 0 || mode > 2) return SO_ICP_E_INVALID;
  c->cfg.time_kernels = mode;
  return SO_ICP_OK;
}
int so_icp_debug_stamps(so_icp_ctx* c, uint64_t out[16]) {
  if (!c || !out) return SO_ICP_E_INVALID;
  NEED_DEVICE(c);
  for (int i = 0; i < 16; ++i) out[i] = c->h_state->dbg[i];
  return SO_ICP_OK;
}
int so_icp_debug_knn_stamps(so_icp_ctx* c, uint64_t* out, size_t capacity_words, size_t* n_words) {
  if (!c || !n_words) return SO_ICP_E_INVALID;
  NEED_DEVICE(c);
  const size_t have = c->d_kdbg.p ? (size_t)2 * kKnnBlocks * 4 * 16 : 0;
  *n_words = have;
  if (!out || !have) return SO_ICP_OK;
  if (capacity_words < have) return fail(c, SO_ICP_E_INVALID, "so_icp_debug_knn_stamps: buffer too small");
  HIP_TRY(c, hipMemcpy(out, c->d_kdbg.p, have * sizeof(uint64_t), hipMemcpyDeviceToHost));
  return SO_ICP_OK;
}
int so_icp_debug_match_status(so_icp_ctx* c, uint8_t* out, size_t n) {
  if (!c || (!out && n)) return SO_ICP_E_INVALID;
  NEED_DEVICE(c);
  if (!n) return SO_ICP_OK;
  if (n > c->d_status.cap) return fail(c, SO_ICP_E_INVALID, "so_icp_debug_match_status: more entries than the last scan had");
  HIP_TRY(c, hipSetDevice(c->cfg.device_id));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  HIP_TRY(c, hipMemcpy(out, c->d_status.p, n, hipMemcpyDeviceToHost));
  return SO_ICP_OK;
}
int so_icp_debug_neighbours(so_icp_ctx* c, uint32_t* out, size_t n) {
  if (!c || (!out && n)) return SO_ICP_E_INVALID;
  NEED_DEVICE(c);
  if (!n) return SO_ICP_OK;
  if (n * 20 > c->d_nbr5.cap) return fail(c, SO_ICP_E_INVALID, "so_icp_debug_neighbours: more entries than the last scan had");
  HIP_TRY(c, hipSetDevice(c->cfg.device_id));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  HIP_TRY(c, hipMemcpy(out, c->d_nbr5.p, n * 20, hipMemcpyDeviceToHost));
  return SO_ICP_OK;
}
int so_icp_synchronize(so_icp_ctx* c) { if (!c) return SO_ICP_E_INVALID; NEED_DEVICE(c); HIP_TRY(c, hipStreamSynchronize(c->stream)); return SO_ICP_OK; }

}  // extern "C"
