// icp_context.cpp -- host driver behind the C ABI of include/so_icp.h.
//
// Mirrors, on top of the HIP kernels, the control flow of
//   LidarSLAM::Localization / performLocalizationAndMapping   src/LidarProcess/LidarSlam.cpp:30-51, 107-210
// (paths relative to /root/reference/super_odometry/).  There is NO CPU fallback: without a usable
// HIP device so_icp_create() fails and says so.
#include <dlfcn.h>
#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "../../include/so_icp.h"
#include "kernels.h"
#include "lm_solver.h"
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

// RCCL entry points, resolved lazily (one collective per evaluation; no link-time dependency)
struct Uid { char internal[SO_ICP_UNIQUE_ID_BYTES]; };  // == ncclUniqueId (rccl.h: char internal[128])
struct Rccl {
  void* lib = nullptr;
  int (*GetUniqueId)(void*) = nullptr;
  int (*CommInitRank)(void**, int, Uid /*ncclUniqueId by value*/, int) = nullptr;
  int (*AllReduce)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
  int (*CommDestroy)(void*) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
};

bool rccl_load(Rccl& r, std::string& err) {
  if (r.lib) return true;
  const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  for (const char* n : names) { r.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL); if (r.lib) break; }
  if (!r.lib) { err = std::string("dlopen(librccl) failed: ") + dlerror(); return false; }
  r.GetUniqueId = reinterpret_cast<int (*)(void*)>(dlsym(r.lib, "ncclGetUniqueId"));
  r.CommInitRank = reinterpret_cast<int (*)(void**, int, Uid, int)>(dlsym(r.lib, "ncclCommInitRank"));
  r.AllReduce = reinterpret_cast<int (*)(const void*, void*, size_t, int, int, void*, hipStream_t)>(dlsym(r.lib, "ncclAllReduce"));
  r.CommDestroy = reinterpret_cast<int (*)(void*)>(dlsym(r.lib, "ncclCommDestroy"));
  r.GetErrorString = reinterpret_cast<const char* (*)(int)>(dlsym(r.lib, "ncclGetErrorString"));
  if (!r.GetUniqueId || !r.CommInitRank || !r.AllReduce || !r.CommDestroy) { err = "librccl lacks a required symbol"; return false; }
  return true;
}
constexpr int kNcclDouble = 8, kNcclSum = 0;  // rccl.h: ncclFloat64 = 8, ncclSum = 0

struct EventSpan { int kind; hipEvent_t a, b; uint32_t units; };  // kind 0 knn, 1 eval, 2 prep

}  // namespace

struct so_icp_ctx {
  so_icp_config cfg;
  std::string err;
  bool host_only = false;  // device_id < 0: LocalMap bookkeeping only, every compute entry point fails
  LocalMap map;
  CanonicalMap cm;
  uint64_t uploaded_version = 0;
  hipStream_t stream = nullptr;
  // map shard in HBM
  DevBuf d_mpts, d_cell_start, d_cube_slot;
  DevMapView view{};
  // scan / correspondence buffers
  DevBuf d_scan_own, d_keys0, d_keys1, d_vals0, d_vals1, d_chunks, d_sort_tmp, d_spx, d_spy, d_spz, d_nd, d_coeff, d_status;
  DevBuf d_small;  // hist[16] int32 | ticket | n_kept | fb_count | LmSums | partials
  int32_t* d_hist = nullptr; uint32_t* d_ticket = nullptr; uint32_t* d_nkept = nullptr; uint32_t* d_fbcount = nullptr;
  LmSums* d_sums = nullptr; double* d_partials = nullptr;
  LmSums* h_sums = nullptr; uint32_t* h_u32 = nullptr;  // pinned
  std::vector<DevBuf> resident_scans;  // so_icp_upload_scan
  // Seam B scratch
  DevBuf d_q, d_nbr, d_d2, d_idx, d_found, d_fblist;
  // persistent LidarSLAM state
  int32_t prev_obs_hist[SO_ICP_N_OBS]{};
  bool have_hist = false;
  int startup_count = 0;
  double last_time = 0;
  // timing
  std::vector<hipEvent_t> ev_pool; size_t ev_used = 0;
  std::vector<EventSpan> spans;
  so_icp_timing timing{};
  // RCCL
  Rccl rccl; void* comm = nullptr;

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
void span_begin(so_icp_ctx* c, int kind, uint32_t units) {
  if (!c->cfg.time_kernels) return;
  EventSpan s{kind, next_event(c), next_event(c), units};
  if (!s.a || !s.b) return;
  (void)hipEventRecord(s.a, c->stream);
  c->spans.push_back(s);
}
void span_end(so_icp_ctx* c) {
  if (!c->cfg.time_kernels || c->spans.empty()) return;
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

int upload_map(so_icp_ctx* c) {
  if (c->uploaded_version == c->map.version()) return SO_ICP_OK;
  c->map.build_canonical(c->cfg.rank, c->cfg.world_size, c->cm);
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
  c->uploaded_version = c->map.version();
  return SO_ICP_OK;
}

int reserve_scan_buffers(so_icp_ctx* c, size_t n) {
  const size_t m = n + 256;
  HIP_TRY(c, c->d_keys0.reserve(m * 4)); HIP_TRY(c, c->d_keys1.reserve(m * 4));
  HIP_TRY(c, c->d_vals0.reserve(m * 4)); HIP_TRY(c, c->d_vals1.reserve(m * 4)); HIP_TRY(c, c->d_chunks.reserve(m * 4));
  HIP_TRY(c, c->d_sort_tmp.reserve(sort_temp_bytes(m) + 256));
  HIP_TRY(c, c->d_spx.reserve(m * 4)); HIP_TRY(c, c->d_spy.reserve(m * 4)); HIP_TRY(c, c->d_spz.reserve(m * 4));
  HIP_TRY(c, c->d_nd.reserve(m * 32)); HIP_TRY(c, c->d_coeff.reserve(m * 8)); HIP_TRY(c, c->d_status.reserve(m));
  return SO_ICP_OK;
}

MatchParams match_params(float plane_res) {
  MatchParams mp;
  mp.plane_res = plane_res;
  mp.sq_max_dist_f = 3 * plane_res;           // float product (LidarSlam.cpp:526)
  mp.max_point_dist = (double)plane_res / 2.0; // LidarSlam.cpp:820
  static const int ablate = std::getenv("SOICP_ABLATE") ? std::atoi(std::getenv("SOICP_ABLATE")) : 0;
  mp.ablate = ablate;
  return mp;
}
EvalParams eval_params(float plane_res, int variant) {
  EvalParams ep;
  const double a = (double)sqrtf(3 * plane_res);  // std::sqrt(float) then TukeyLoss(double a) (LidarSlam.cpp:271)
  ep.a2 = a * a;
  ep.variant = variant;
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

// one fused evaluation at `pose`: kernel -> (all-reduce) -> pinned host copy
int evaluate_at(so_icp_ctx* c, const double pose[7], uint32_t n_kept, LmSums& out) {
  const Pose P = pose_from_array(pose);
  CorrBuffers corr{c->d_nd.as<double4>(), c->d_coeff.as<double>(), c->d_status.as<uint8_t>()};
  span_begin(c, 1, n_kept);
  launch_eval(c->d_spx.as<float>(), c->d_spy.as<float>(), c->d_spz.as<float>(), corr, n_kept, P,
              eval_params(c->map.plane_res(), c->cfg.tukey_variant), c->d_partials, c->d_ticket, c->d_hist, c->d_sums, c->stream);
  span_end(c);
  if (c->comm) {  // per-evaluation collective: 45 fp64 summed over the shards (xGMI, latency-bound)
    const int rc = c->rccl.AllReduce(c->d_sums, c->d_sums, sizeof(LmSums) / sizeof(double), kNcclDouble, kNcclSum, c->comm, c->stream);
    if (rc != 0) return fail(c, SO_ICP_E_RCCL, std::string("ncclAllReduce: ") + (c->rccl.GetErrorString ? c->rccl.GetErrorString(rc) : "?"));
  }
  HIP_TRY(c, hipMemcpyAsync(c->h_sums, c->d_sums, sizeof(LmSums), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  out = *c->h_sums;
  return SO_ICP_OK;
}

int register_core(so_icp_ctx* c, const float* d_scan, size_t n, const double pose_in[7], double pose_out[7], so_icp_stats* st) {
  const auto t_begin = std::chrono::steady_clock::now();
  so_icp_stats local;
  if (!st) st = &local;
  std::memset(st, 0, sizeof(*st));
  double T[7], T_init[7], T_last[7];
  std::memcpy(T, pose_in, sizeof(T)); std::memcpy(T_init, pose_in, sizeof(T)); std::memcpy(T_last, pose_in, sizeof(T));  // LidarSlam.cpp:53-57
  std::memcpy(pose_out, pose_in, sizeof(T));
  if (c->have_hist) uncertainty_from_hist(c->prev_obs_hist, st->uncertainty);  // LidarSlam.cpp:47
  int pos[3];
  c->map.shift(T, pos);                                                       // LidarSlam.cpp:363
  st->pos_in_localmap[0] = pos[0]; st->pos_in_localmap[1] = pos[1]; st->pos_in_localmap[2] = pos[2];
  st->laser_cloud_surf_from_map_num = c->map.count_5x5(pos);                  // LidarSlam.cpp:367
  st->laser_cloud_surf_stack_num = (int32_t)n;
  st->startup_count = c->startup_count;
  if (!(st->laser_cloud_surf_from_map_num > 50)) return SO_ICP_NOT_ENOUGH_MAP_FEATURES;  // LidarSlam.cpp:113-116
  int rc = upload_map(c);
  if (rc) return rc;
  rc = reserve_scan_buffers(c, n);
  if (rc) return rc;
  const auto t_icp = std::chrono::steady_clock::now();  // TicToc t_opt, LidarSlam.cpp:118

  // ---- once per registration: sampling, spatial sort (locality survives the small pose updates) ----
  uint32_t n_kept = 0, n_chunks = 0;
  if (c->cfg.time_kernels) HIP_TRY(c, hipMemsetAsync(c->d_hist + 16, 0, 4 * sizeof(int32_t), c->stream));
  if (n) {
    span_begin(c, 2, (uint32_t)n);
    HIP_TRY(c, hipMemsetAsync(c->d_nkept, 0, 8, c->stream));  // n_kept, n_chunks
    launch_scan_keys(d_scan, (uint32_t)n, pose_from_array(T), c->view, c->cfg.max_surface_features, c->cfg.rank,
                     c->cfg.world_size, c->d_keys0.as<uint32_t>(), c->d_vals0.as<uint32_t>(), c->d_nkept, c->stream);
    launch_sort_pairs(c->d_sort_tmp.p, c->d_sort_tmp.cap, c->d_keys0.as<uint32_t>(), c->d_keys1.as<uint32_t>(),
                      c->d_vals0.as<uint32_t>(), c->d_vals1.as<uint32_t>(), (uint32_t)n, c->stream);
    HIP_TRY(c, hipMemcpyAsync(c->h_u32, c->d_nkept, 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    n_kept = c->h_u32[0];
    launch_chunk_heads(c->d_keys1.as<uint32_t>(), n_kept, c->d_chunks.as<uint32_t>(), c->d_nkept + 1, c->stream);
    HIP_TRY(c, hipMemcpyAsync(c->h_u32 + 1, c->d_nkept + 1, 4, hipMemcpyDeviceToHost, c->stream));
    launch_gather_scan(d_scan, c->d_vals1.as<uint32_t>(), n_kept, c->d_spx.as<float>(), c->d_spy.as<float>(), c->d_spz.as<float>(), c->stream);
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    n_chunks = c->h_u32[1];
    span_end(c);
  }

  const int max_outer = std::min(c->cfg.max_iterations > 0 ? c->cfg.max_iterations : 4, SO_ICP_MAX_OUTER);
  const int lm_max = c->cfg.lm_max_iterations > 0 ? c->cfg.lm_max_iterations : 4;
  const MatchParams mp = match_params(c->map.plane_res());
  CorrBuffers corr{c->d_nd.as<double4>(), c->d_coeff.as<double>(), c->d_status.as<uint8_t>()};
  LmState S;
  std::memset(&S, 0, sizeof(S));
  bool final_sums_valid = false;
  for (int it = 0; it < max_outer; ++it) {
    so_icp_iter_stats& is = st->iterations[it];
    st->n_iterations = it + 1;
    // processPlannerFeatures: every (kept) query in parallel (LidarSlam.cpp:323-344)
    HIP_TRY(c, hipMemsetAsync(c->d_hist, 0, 16 * sizeof(int32_t), c->stream));  // ResetDistanceParameters, :847-852
    span_begin(c, 0, n_kept);
    launch_knn_plane(c->d_spx.as<float>(), c->d_spy.as<float>(), c->d_spz.as<float>(), n_kept, c->d_keys1.as<uint32_t>(), c->d_chunks.as<uint32_t>(),
                     n_chunks, pose_from_array(T), c->view, mp, corr, c->d_hist, c->stream);
    span_end(c);
    // setupOptimizationProblem + solveOptimizationProblem (LidarSlam.cpp:213-240)
    double prev[7];
    std::memcpy(prev, T, sizeof(T));
    LmSums sums;
    rc = evaluate_at(c, T, n_kept, sums);
    if (rc) return rc;
    for (int h = 0; h < SO_ICP_N_REJECT; ++h) is.reject_hist[h] = (int32_t)sums.hist[h];
    for (int h = 0; h < SO_ICP_N_OBS; ++h) is.obs_hist[h] = (int32_t)sums.hist[7 + h];
    double next[7];
    int more = lm_begin(S, T, sums, lm_max, next);
    while (more) {
      rc = evaluate_at(c, next, n_kept, sums);
      if (rc) return rc;
      more = lm_feed(S, sums, next);
    }
    std::memcpy(T, S.x, sizeof(T));  // LidarSlam.cpp:135-136
    final_sums_valid = S.count > 0;
    is.num_surf_from_scan = (int32_t)S.count;
    is.lm_iterations = S.lm_iterations;
    is.num_successful_steps = S.num_successful;
    is.termination = S.termination;
    is.initial_cost = S.initial_cost; is.final_cost = S.x_cost;
    relative_motion(prev, T, is.translation_norm, is.rotation_norm);  // recordIterationStats, :242-251
    std::memcpy(is.pose_after, T, sizeof(T));
    std::memcpy(c->prev_obs_hist, is.obs_hist, sizeof(c->prev_obs_hist));
    c->have_hist = true;
    if (S.num_successful == 1 || it == max_outer - 1) break;  // LidarSlam.cpp:141
  }
  if (final_sums_valid) {  // normal equations at the returned pose (S.H/S.g always belong to S.x)
    std::memcpy(st->JtJ, S.H, sizeof(st->JtJ));
    std::memcpy(st->Jtr, S.g, sizeof(st->Jtr));
  }
  yaw_correction(T, T_last, c->cfg.yaw_ratio);  // performPostOptimizationProcessing, :155-157
  st->time_elapsed_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_icp).count();  // :199-200
  relative_motion(T_init, T, st->total_translation, st->total_rotation);
  relative_motion(T_last, T, st->translation_from_last, st->rotation_from_last);
  st->prediction_source = 0;
  std::memcpy(pose_out, T, sizeof(T));
  if (c->cfg.time_kernels) {
    HIP_TRY(c, hipMemcpyAsync(c->h_u32 + 4, c->d_hist + 16, 4 * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    c->timing.knn_group_passes += c->h_u32[4]; c->timing.knn_fallback_lanes += c->h_u32[5];
    c->timing.knn_candidates_scanned += (int64_t)c->h_u32[6] * 16;
    spans_collect(c);
  }
  c->timing.registrations++;
  c->timing.host_ms_total += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
  return SO_ICP_OK;
}

int upload_scan_impl(so_icp_ctx* c, const float* xyz, size_t n, size_t stride_bytes, DevBuf& dst) {
  if (stride_bytes == 0) stride_bytes = 12;
  if (stride_bytes % 4) return fail(c, SO_ICP_E_INVALID, "stride_bytes must be a multiple of 4");
  HIP_TRY(c, dst.reserve((n + 64) * 12));
  if (!n) return SO_ICP_OK;
  if (stride_bytes == 12) {
    HIP_TRY(c, hipMemcpyAsync(dst.p, xyz, n * 12, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
  } else {
    std::vector<float> packed(n * 3);
    const size_t sf = stride_bytes / 4;
    for (size_t i = 0; i < n; ++i) { packed[3 * i] = xyz[i * sf]; packed[3 * i + 1] = xyz[i * sf + 1]; packed[3 * i + 2] = xyz[i * sf + 2]; }
    HIP_TRY(c, hipMemcpyAsync(dst.p, packed.data(), n * 12, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
  }
  return SO_ICP_OK;
}

}  // namespace

so_icp_ctx::~so_icp_ctx() {
  if (comm && rccl.CommDestroy) rccl.CommDestroy(comm);
  for (DevBuf* b : {&d_mpts, &d_cell_start, &d_cube_slot, &d_scan_own, &d_keys0, &d_keys1, &d_vals0, &d_vals1, &d_chunks,
                    &d_sort_tmp, &d_spx, &d_spy, &d_spz, &d_nd, &d_coeff, &d_status, &d_small, &d_q, &d_nbr, &d_d2, &d_idx,
                    &d_found, &d_fblist})
    b->release();
  for (DevBuf& b : resident_scans) b.release();
  if (h_sums) (void)hipHostFree(h_sums);
  if (h_u32) (void)hipHostFree(h_u32);
  for (hipEvent_t e : ev_pool) (void)hipEventDestroy(e);
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
  const size_t small_bytes = 512 + sizeof(LmSums) + 256 + (size_t)kEvalBlocks * kSumsStride * sizeof(double);
  if ((e = c->d_small.reserve(small_bytes)) != hipSuccess) return bail(std::string("hipMalloc: ") + hipGetErrorString(e));
  if ((e = hipMemset(c->d_small.p, 0, c->d_small.cap)) != hipSuccess) return bail(std::string("hipMemset: ") + hipGetErrorString(e));
  char* base = c->d_small.as<char>();
  c->d_hist = reinterpret_cast<int32_t*>(base);            // 16 histogram bins + 4 kernel statistics (128 B reserved)
  c->d_ticket = reinterpret_cast<uint32_t*>(base + 128);
  c->d_nkept = reinterpret_cast<uint32_t*>(base + 192);
  c->d_fbcount = reinterpret_cast<uint32_t*>(base + 256);
  c->d_sums = reinterpret_cast<LmSums*>(base + 512);
  c->d_partials = reinterpret_cast<double*>(base + 512 + ((sizeof(LmSums) + 255) / 256) * 256);
  if ((e = hipHostMalloc(reinterpret_cast<void**>(&c->h_sums), sizeof(LmSums))) != hipSuccess) return bail(std::string("hipHostMalloc: ") + hipGetErrorString(e));
  if ((e = hipHostMalloc(reinterpret_cast<void**>(&c->h_u32), 64)) != hipSuccess) return bail(std::string("hipHostMalloc: ") + hipGetErrorString(e));
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
  if (plane_res != c->map.plane_res()) c->uploaded_version = 0;  // cell size follows planeRes
  c->map.set_resolution(line_res, plane_res);
  c->cfg.line_res = line_res; c->cfg.plane_res = plane_res;
  return SO_ICP_OK;
}
int so_icp_set_max_surface_features(so_icp_ctx* c, int v) { if (!c) return SO_ICP_E_INVALID; c->cfg.max_surface_features = v; return SO_ICP_OK; }
int so_icp_set_max_iterations(so_icp_ctx* c, int v) { if (!c || v < 1) return SO_ICP_E_INVALID; c->cfg.max_iterations = v; return SO_ICP_OK; }

int so_icp_map_set_origin(so_icp_ctx* c, const double t[3], int o[3]) {
  if (!c || !t) return SO_ICP_E_INVALID;
  c->map.set_origin(t);
  if (o) { o[0] = c->map.origin()[0]; o[1] = c->map.origin()[1]; o[2] = c->map.origin()[2]; }
  return SO_ICP_OK;
}
int so_icp_map_get_origin(so_icp_ctx* c, int o[3]) {
  if (!c || !o) return SO_ICP_E_INVALID;
  o[0] = c->map.origin()[0]; o[1] = c->map.origin()[1]; o[2] = c->map.origin()[2];
  return SO_ICP_OK;
}
int so_icp_map_shift(so_icp_ctx* c, const double t[3], int pos[3]) {
  if (!c || !t || !pos) return SO_ICP_E_INVALID;
  c->map.shift(t, pos);
  return SO_ICP_OK;
}
int so_icp_map_add_surf(so_icp_ctx* c, const float* xyz, size_t n, size_t stride_bytes) {
  if (!c || (!xyz && n)) return SO_ICP_E_INVALID;
  if (stride_bytes == 0) stride_bytes = 12;
  if (stride_bytes % 4) return fail(c, SO_ICP_E_INVALID, "stride_bytes must be a multiple of 4");
  return c->map.add_surf(xyz, n, stride_bytes / 4);
}
int so_icp_map_count_5x5(so_icp_ctx* c, const int pos[3], int* n_edge, int* n_surf) {
  if (!c || !pos) return SO_ICP_E_INVALID;
  if (n_edge) *n_edge = 0;
  if (n_surf) *n_surf = c->map.count_5x5(pos);
  return SO_ICP_OK;
}
int so_icp_map_export(so_icp_ctx* c, float* xyz, size_t cap, size_t* n_out, int only_5x5, const int pos[3]) {
  if (!c || (only_5x5 && !pos)) return SO_ICP_E_INVALID;
  const int zero[3] = {0, 0, 0};
  const size_t n = c->map.export_points(xyz, cap, only_5x5 != 0, pos ? pos : zero);
  if (n_out) *n_out = n;
  return SO_ICP_OK;
}
int so_icp_map_size(so_icp_ctx* c, size_t* n, size_t* n_rank) {
  if (!c) return SO_ICP_E_INVALID;
  if (n) *n = c->map.size();
  if (n_rank) { NEED_DEVICE(c); const int rc = upload_map(c); if (rc) return rc; *n_rank = c->view.n_points; }
  return SO_ICP_OK;
}
int so_icp_map_clear(so_icp_ctx* c) { if (!c) return SO_ICP_E_INVALID; c->map.clear(); return SO_ICP_OK; }

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
  const double cover = c->cm.cell * (1.0 - 1e-5);
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
  const int rc = upload_scan_impl(c, xyz, n, stride_bytes, c->d_scan_own);
  if (rc) return rc;
  return register_core(c, c->d_scan_own.as<float>(), n, pose_in, pose_out, st);
}

int so_icp_localization(so_icp_ctx* c, int initialization, const double T_in[7], const float* xyz, size_t n, size_t stride_bytes,
                        double time_laser_odometry, double pose_out[7], so_icp_stats* st) {
  if (!c || !T_in || !pose_out || (!xyz && n)) return SO_ICP_E_INVALID;
  if (stride_bytes == 0) stride_bytes = 12;
  if (stride_bytes % 4) return fail(c, SO_ICP_E_INVALID, "stride_bytes must be a multiple of 4");
  const size_t sf = stride_bytes / 4;
  auto transform_and_add = [&](const double T[7]) {  // transformAndAddToMap, LidarSlam.cpp:60-80; TransformPoint, superodom_utils.h:119-123
    std::vector<float> w(n * 3);
    for (size_t i = 0; i < n; ++i) {
      double ox, oy, oz;
      quat_rotate<double>(T + 3, (double)xyz[i * sf], (double)xyz[i * sf + 1], (double)xyz[i * sf + 2], ox, oy, oz);
      w[3 * i] = (float)(ox + T[0]); w[3 * i + 1] = (float)(oy + T[1]); w[3 * i + 2] = (float)(oz + T[2]);
    }
    c->map.add_surf(w.data(), n, 3);
  };
  if (!initialization) {  // initializeMapping, LidarSlam.cpp:83-94
    std::memcpy(pose_out, T_in, 7 * sizeof(double));
    if (st) std::memset(st, 0, sizeof(*st));
    c->map.set_origin(T_in);
    transform_and_add(T_in);
    c->last_time = time_laser_odometry;
    return SO_ICP_MAP_SEEDED;
  }
  so_icp_stats local;
  if (!st) st = &local;
  const int rc = so_icp_register(c, xyz, n, stride_bytes, T_in, pose_out, st);
  if (rc != SO_ICP_OK) return rc;  // NOT_ENOUGH: the reference returns before the post-processing (LidarSlam.cpp:113-116)
  // checkMotionThresholds, LidarSlam.cpp:173-195: always accepts; only the startupCount side effect survives
  const double dt = time_laser_odometry - c->last_time;
  if (st->translation_from_last / dt > c->cfg.velocity_failure_threshold) c->startup_count = 5;
  st->startup_count = c->startup_count;
  transform_and_add(pose_out);  // LidarSlam.cpp:163-167
  c->last_time = time_laser_odometry;
  return SO_ICP_OK;
}

int so_icp_comm_unique_id(uint8_t id[SO_ICP_UNIQUE_ID_BYTES]) {
  if (!id) return SO_ICP_E_INVALID;
  Rccl r;
  std::string err;
  if (!rccl_load(r, err)) { g_create_error = err; return SO_ICP_E_RCCL; }
  Uid u;
  std::memset(&u, 0, sizeof(u));
  const int rc = r.GetUniqueId(&u);
  if (rc != 0) { g_create_error = "ncclGetUniqueId failed"; return SO_ICP_E_RCCL; }
  std::memcpy(id, &u, SO_ICP_UNIQUE_ID_BYTES);
  return SO_ICP_OK;
}

int so_icp_comm_init(so_icp_ctx* c, const uint8_t id[SO_ICP_UNIQUE_ID_BYTES]) {
  if (!c || !id) return SO_ICP_E_INVALID;
  NEED_DEVICE(c);
  if (!rccl_load(c->rccl, c->err)) return SO_ICP_E_RCCL;
  HIP_TRY(c, hipSetDevice(c->cfg.device_id));
  Uid u;
  std::memcpy(&u, id, SO_ICP_UNIQUE_ID_BYTES);
  const int rc = c->rccl.CommInitRank(&c->comm, c->cfg.world_size, u, c->cfg.rank);
  if (rc != 0) { c->comm = nullptr; return fail(c, SO_ICP_E_RCCL, std::string("ncclCommInitRank: ") + (c->rccl.GetErrorString ? c->rccl.GetErrorString(rc) : "?")); }
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
int so_icp_synchronize(so_icp_ctx* c) { if (!c) return SO_ICP_E_INVALID; NEED_DEVICE(c); HIP_TRY(c, hipStreamSynchronize(c->stream)); return SO_ICP_OK; }

}  // extern "C"
