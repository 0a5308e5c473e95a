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
};

struct MatchParams {
  float plane_res;        // localMap.planeRes_ (float member, LocalMap.h:761)
  float sq_max_dist_f;    // 3 * planeRes evaluated in float (LidarSlam.cpp:526)
  double max_point_dist;  // planeRes / 2.0 (LidarSlam.cpp:820)
  int32_t ablate;         // profiling only (env SOICP_ABLATE): bit0 skip plane fit, bit1 skip scan, bit2 skip re-rank
};

struct EvalParams {
  double a2;       // TukeyLoss a^2 with a = (double)sqrtf(3*planeRes)  (LidarSlam.cpp:271)
  int32_t variant; // 0: Ceres 2.0.0, 1: Ceres >= 2.1
};

// per-correspondence record written by the k-NN + plane-fit kernel, read by the evaluation kernel
struct CorrBuffers {
  double4* nd;      // {nx, ny, nz, negative_OA_dot_norm}
  double* coeff;    // residualCoefficient; 0 for rejected points
  uint8_t* status;  // MatchingResult
};

constexpr int kEvalBlocks = 256;     // one workgroup per CU
constexpr int kSumsStride = 48;      // doubles per partial record (45 used)
constexpr uint32_t kKeyDropped = 0xFFFFFFFFu;   // not sampled / not owned by this rank
constexpr uint32_t kKeyNoCube = 0xFFFFFFFEu;    // processed, but cube outside window / no tree

size_t sort_temp_bytes(size_t n);

void launch_scan_keys(const float* d_scan_xyz, uint32_t n, const Pose& pose, const DevMapView& map,
                      int max_surface_features, int rank, int world, uint32_t* d_keys, uint32_t* d_vals,
                      uint32_t* d_n_kept, hipStream_t s);
void launch_sort_pairs(void* d_temp, size_t temp_bytes, const uint32_t* keys_in, uint32_t* keys_out,
                       const uint32_t* vals_in, uint32_t* vals_out, uint32_t n, hipStream_t s);
void launch_gather_scan(const float* d_scan_xyz, const uint32_t* d_perm, uint32_t n_kept, float* spx, float* spy,
                        float* spz, hipStream_t s);
void launch_chunk_heads(const uint32_t* d_keys_sorted, uint32_t n_kept, uint32_t* d_chunk_start, uint32_t* d_n_chunks,
                        hipStream_t s);
void launch_knn_plane(const float* spx, const float* spy, const float* spz, uint32_t n_kept, const uint32_t* d_keys_sorted,
                      const uint32_t* d_chunk_start, uint32_t n_chunks, const Pose& pose, const DevMapView& map,
                      const MatchParams& mp, CorrBuffers corr, int32_t* d_hist /*20*/, hipStream_t s);
void launch_eval(const float* spx, const float* spy, const float* spz, const CorrBuffers& corr, uint32_t n_kept,
                 const Pose& pose, const EvalParams& ep, double* d_partials, uint32_t* d_ticket,
                 const int32_t* d_hist, LmSums* d_sums, hipStream_t s);
// Seam B
void launch_knn_only(const float* d_q_xyz, uint32_t nq, int k, const DevMapView& map, float gate_d2, float* d_nbr,
                     float* d_d2, int32_t* d_idx, uint8_t* d_found, uint32_t* d_fallback_list,
                     uint32_t* d_fallback_count, hipStream_t s);
void launch_knn_fallback(const float* d_q_xyz, const uint32_t* d_fallback_list, uint32_t n_fallback, int k,
                         const DevMapView& map, float* d_nbr, float* d_d2, int32_t* d_idx, hipStream_t s);

}  // namespace soicp
