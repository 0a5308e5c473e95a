// kernels.hip -- gfx950 (MI355X / CDNA4) kernels of the scan-to-map ICP hot path.
//
//   scan_keys_kernel    sampling rule + world transform + spatial sort key        (LidarSlam.cpp:346-359, 397-398)
//   gather_scan_kernel  AoS scan -> spatially sorted SoA (once per registration)
//   knn_plane_kernel    per query: cube-restricted exact 5-NN over the hashed-voxel cell grid, then in
//                       registers the PCA gate, the 5x3 LS plane, inlier gate, fit coefficient and the
//                       observability labels                                       (LidarSlam.cpp:514-572)
//   eval_kernel         per LM evaluation: residual, Tukey*coeff weight, 6-DoF Jacobian and the fp64
//                       21+6+1+1 normal-equation sums, wavefront-shuffle reduced, deterministic
//                       two-stage finish by the last workgroup                     (lidarOptimization.cpp:55-80)
//   knn_only / knn_fallback   Seam B (LocalMap::nearestKSearchSurf, LocalMap.h:481-525)
//
// Nothing here is a dense contraction, so no MFMA: these are HBM/latency-bound gather-scan-reduce
// kernels (64-wide wavefronts, fp64 VALU for the parts the reference computes in double).
// Paths in citations are relative to /root/reference/super_odometry/{src,include/super_odometry}.
#include <hip/hip_runtime.h>

#include <cstring>  // rocprim/iterator/texture_cache_iterator.hpp calls ::memset on the host path

#include <rocprim/rocprim.hpp>

#include "kernels.h"

namespace soicp {

#define SO_MATCH_SUCCESS 0
#define SO_MATCH_NOT_ENOUGH 1
#define SO_MATCH_TOO_FAR 2
#define SO_MATCH_BAD_PCA 3
#define SO_MATCH_INVALID 4
#define SO_MATCH_MSE 5

// ------------------------------------------------------------------------------------------------
// map addressing
// ------------------------------------------------------------------------------------------------
struct CellRef {
  int slot;        // -1: outside window / no tree
  int cx, cy, cz;  // cell inside the cube
};

// cube index exactly as LocalMap::nearestKSearchSurf (LocalMap.h:488-507); then the cell of the
// hashed-voxel grid inside that cube.
__device__ __forceinline__ CellRef locate(const DevMapView& m, float qx, float qy, float qz, int* wcube = nullptr) {
  CellRef r;
  const int ci = cube_coord((double)qx, m.origin[0]);
  const int cj = cube_coord((double)qy, m.origin[1]);
  const int ck = cube_coord((double)qz, m.origin[2]);
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
__global__ __launch_bounds__(256) void scan_keys_kernel(const float* __restrict__ scan, uint32_t n, Pose pose,
                                                        DevMapView map, int max_surface_features, int rank, int world,
                                                        uint32_t* __restrict__ keys, uint32_t* __restrict__ vals,
                                                        uint32_t* __restrict__ n_kept) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t key = kKeyDropped;
  bool process = true;
  if (max_surface_features > 0 && n > (uint32_t)max_surface_features) {  // calculateSamplingRate / shouldProcessPoint
    const double rate = 1.0 * max_surface_features / n;
    const double rem = fmod((double)i * rate, 1.0);
    if (rem + 0.001 > rate) process = false;
  }
  if (process) {
    const float px = scan[3 * i], py = scan[3 * i + 1], pz = scan[3 * i + 2];
    double wx, wy, wz;
    quat_rotate<double>(pose.q, (double)px, (double)py, (double)pz, wx, wy, wz);
    const float qx = (float)(wx + pose.t[0]), qy = (float)(wy + pose.t[1]), qz = (float)(wz + pose.t[2]);
    int w[3];
    const CellRef c = locate(map, qx, qy, qz, w);
    if (c.slot < 0) {
      key = (rank == 0) ? kKeyNoCube : kKeyDropped;  // counted once (NOT_ENOUGH_NEIGHBORS) by rank 0
    } else {
      int owner = 0;
      if (world > 1) owner = (int)(brick_hash(w[0], w[1], w[2], c.cx / kBrickCells, c.cy / kBrickCells, c.cz / kBrickCells) % (uint32_t)world);
      if (owner == rank) key = ((uint32_t)c.slot << 18) | morton3((uint32_t)c.cx, (uint32_t)c.cy, (uint32_t)c.cz);
    }
  }
  keys[i] = key;
  vals[i] = i;
  if (key != kKeyDropped) atomicAdd(n_kept, 1u);  // coalesced by the compiler into one add per wave
}

__global__ __launch_bounds__(256) void gather_scan_kernel(const float* __restrict__ scan, const uint32_t* __restrict__ perm,
                                                          uint32_t n_kept, float* __restrict__ spx,
                                                          float* __restrict__ spy, float* __restrict__ spz) {
  const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n_kept) return;
  const uint32_t i = perm[j];
  spx[j] = scan[3 * i]; spy[j] = scan[3 * i + 1]; spz[j] = scan[3 * i + 2];
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
        const float d2 = l2_d2(qx, qy, qz, m.x[i], m.y[i], m.z[i]);
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
  const double theta = (aqq - app) / (2.0 * apq);
  const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
  const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
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
  for (int sweep = 0; sweep < 32; ++sweep) {
    const double off = a01 * a01 + a02 * a02 + a12 * a12;
    const double dg = a00 * a00 + a11 * a11 + a22 * a22;
    if (off <= 1e-40 * dg || off == 0.0) break;
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
          const double f = 2.0 * dot / vn2;
#pragma unroll
          for (int i = k; i < 5; ++i) A[j][i] -= f * v[i];
        }
        double dot = 0;
#pragma unroll
        for (int i = k; i < 5; ++i) dot += v[i] * b[i];
        const double f = 2.0 * dot / vn2;
#pragma unroll
        for (int i = k; i < 5; ++i) b[i] -= f * v[i];
        A[k][k] = alpha;
      }
    }
  }
  const double y2 = b[2] / A[2][2];
  const double y1 = (b[1] - A[2][1] * y2) / A[1][1];
  const double y0 = (b[0] - A[1][0] * y1 - A[2][0] * y2) / A[0][0];
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
  const double planar_2 = (l2 - l3) / l1;
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

// ComputePlaneDistanceParameters after the neighbour search (LidarSlam.cpp:533-571)
__device__ __forceinline__ int plane_from_neighbours(const float nb[15], const double pw[3], const Pose& pose,
                                                     const MatchParams& mp, double nd[4], double& coeff, int obs[3]) {
  // PCA (LidarSlam.cpp:756-775, utils/superodom_utils.h:143-151)
  double mx = 0, my = 0, mz = 0;
#pragma unroll
  for (int j = 0; j < 5; ++j) { mx += (double)nb[3 * j]; my += (double)nb[3 * j + 1]; mz += (double)nb[3 * j + 2]; }
  mx /= 5.0; my /= 5.0; mz /= 5.0;
  double s00 = 0, s01 = 0, s02 = 0, s11 = 0, s12 = 0, s22 = 0;
#pragma unroll
  for (int j = 0; j < 5; ++j) {
    const double a = (double)nb[3 * j] - mx, b = (double)nb[3 * j + 1] - my, c = (double)nb[3 * j + 2] - mz;
    s00 += a * a; s01 += a * b; s02 += a * c; s11 += b * b; s12 += b * c; s22 += c * c;
  }
  double ev[3], nrm[3];
  eig3_sym(s00, s01, s02, s11, s12, s22, ev, nrm);
  if (ev[0] < 1e-6 || ev[1] / ev[2] < 0.1) return SO_MATCH_BAD_PCA;  // LidarSlam.cpp:772
  double x[3];
  if (!plane_ls5(nb, x)) return SO_MATCH_INVALID;                    // LidarSlam.cpp:809-812
  const double nn = sqrt(x[0] * x[0] + x[1] * x[1] + x[2] * x[2]);
  const double d = 1.0 / nn;                                         // LidarSlam.cpp:815
  const double n0 = x[0] / nn, n1 = x[1] / nn, n2 = x[2] / nn;       // LidarSlam.cpp:816
  double sum = 0;
  bool too_far = false;
#pragma unroll
  for (int j = 0; j < 5; ++j) {
    const double dist = fabs(n0 * (double)nb[3 * j] + n1 * (double)nb[3 * j + 1] + n2 * (double)nb[3 * j + 2] + d);
    too_far |= dist > mp.max_point_dist;                             // LidarSlam.cpp:832
    sum += dist;
  }
  if (too_far) return SO_MATCH_MSE;
  const double mean_abs = sum / 5.0;
  if (pw[0] * nrm[0] + pw[1] * nrm[1] + pw[2] * nrm[2] < 0) { nrm[0] = -nrm[0]; nrm[1] = -nrm[1]; nrm[2] = -nrm[2]; }  // :553-561
  observability(pw, ev, nrm, pose, obs[0], obs[1], obs[2]);
  coeff = 1.0 - sqrt(mean_abs / (double)mp.sq_max_dist_f);           // LidarSlam.cpp:568
  nd[0] = n0; nd[1] = n1; nd[2] = n2; nd[3] = d;
  return SO_MATCH_SUCCESS;
}

__global__ __launch_bounds__(256) void knn_plane_kernel(const float* __restrict__ spx, const float* __restrict__ spy,
                                                        const float* __restrict__ spz, uint32_t n_kept, Pose pose,
                                                        DevMapView map, MatchParams mp, CorrBuffers corr,
                                                        int32_t* __restrict__ hist) {
  __shared__ int32_t lh[16];
  if (threadIdx.x < 16) lh[threadIdx.x] = 0;
  __syncthreads();
  const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j < n_kept) {
    double pw[3];
    quat_rotate<double>(pose.q, (double)spx[j], (double)spy[j], (double)spz[j], pw[0], pw[1], pw[2]);  // LidarSlam.cpp:397-398
    pw[0] += pose.t[0]; pw[1] += pose.t[1]; pw[2] += pose.t[2];
    const float qx = (float)pw[0], qy = (float)pw[1], qz = (float)pw[2];  // LidarSlam.cpp:728-731
    int status;
    double nd[4] = {0, 0, 0, 0}, coeff = 0;
    int obs[3] = {0, 0, 0};
    const CellRef c = locate(map, qx, qy, qz);
    if (c.slot < 0) {
      status = SO_MATCH_NOT_ENOUGH;  // LidarSlam.cpp:736-739
    } else {
      Top5 top;
      top.init();
      knn27(map, c, qx, qy, qz, top);
      const float d2_4 = __uint_as_float((uint32_t)(top.b4 >> 32));
      if (top.b4 == ~0ull || (double)d2_4 > (double)mp.sq_max_dist_f) {
        status = SO_MATCH_TOO_FAR;   // LidarSlam.cpp:741-744 (d2[4] stays FLT_MAX with < 5 points)
      } else {
        float nb[15];
        const uint32_t id[5] = {(uint32_t)top.b0, (uint32_t)top.b1, (uint32_t)top.b2, (uint32_t)top.b3, (uint32_t)top.b4};
#pragma unroll
        for (int t = 0; t < 5; ++t) { nb[3 * t] = map.x[id[t]]; nb[3 * t + 1] = map.y[id[t]]; nb[3 * t + 2] = map.z[id[t]]; }
        status = plane_from_neighbours(nb, pw, pose, mp, nd, coeff, obs);
      }
    }
    if (status != SO_MATCH_SUCCESS) { coeff = 0; nd[0] = nd[1] = nd[2] = nd[3] = 0; }
    corr.nd[j] = make_double4(nd[0], nd[1], nd[2], nd[3]);
    corr.coeff[j] = coeff;
    corr.status[j] = (uint8_t)status;
    atomicAdd(&lh[status], 1);
    if (status == SO_MATCH_SUCCESS) { atomicAdd(&lh[7 + obs[0]], 1); atomicAdd(&lh[7 + obs[1]], 1); atomicAdd(&lh[7 + obs[2]], 1); }
  }
  __syncthreads();
  if (threadIdx.x < 16 && lh[threadIdx.x]) atomicAdd(&hist[threadIdx.x], lh[threadIdx.x]);
}

// ------------------------------------------------------------------------------------------------
// LM evaluation: fused cost + J^T J + J^T r
// ------------------------------------------------------------------------------------------------
constexpr int kNAcc = 29;  // cost, count, Jtr[6], JtJ[21]

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}

__global__ __launch_bounds__(256) void eval_kernel(const float* __restrict__ spx, const float* __restrict__ spy,
                                                   const float* __restrict__ spz, CorrBuffers corr, uint32_t n_kept,
                                                   Pose pose, EvalParams ep, double* __restrict__ partials,
                                                   uint32_t* __restrict__ ticket, const int32_t* __restrict__ hist,
                                                   LmSums* __restrict__ out) {
  __shared__ double red[4][kNAcc + 3];
  __shared__ double fin[8][32];
  __shared__ bool is_last;
  double acc[kNAcc];
#pragma unroll
  for (int a = 0; a < kNAcc; ++a) acc[a] = 0;
  // R(q) as Eigen::Quaternion::toRotationMatrix (lidarOptimization.cpp:70)
  const double qx = pose.q[0], qy = pose.q[1], qz = pose.q[2], qw = pose.q[3];
  const double tx = 2 * qx, ty = 2 * qy, tz = 2 * qz;
  const double twx = tx * qw, twy = ty * qw, twz = tz * qw, txx = tx * qx, txy = ty * qx, txz = tz * qx;
  const double tyy = ty * qy, tyz = tz * qy, tzz = tz * qz;
  const double R00 = 1 - (tyy + tzz), R01 = txy - twz, R02 = txz + twy;
  const double R10 = txy + twz, R11 = 1 - (txx + tzz), R12 = tyz - twx;
  const double R20 = txz - twy, R21 = tyz + twx, R22 = 1 - (txx + tyy);
  for (uint32_t j = blockIdx.x * blockDim.x + threadIdx.x; j < n_kept; j += gridDim.x * blockDim.x) {
    const double c = corr.coeff[j];
    if (corr.status[j] != SO_MATCH_SUCCESS) continue;
    const double4 nd = corr.nd[j];
    const double px = (double)spx[j], py = (double)spy[j], pz = (double)spz[j];
    double wx, wy, wz;
    quat_rotate<double>(pose.q, px, py, pz, wx, wy, wz);                   // lidarOptimization.cpp:59
    wx += pose.t[0]; wy += pose.t[1]; wz += pose.t[2];
    const double r = nd.x * wx + nd.y * wy + nd.z * wz + nd.w;             // lidarOptimization.cpp:61
    // ScaledLoss(TukeyLoss(a), c): rho, rho' [UPSTREAM ceres loss_function.cc]; corrector with rho''<=0
    const double s = r * r;
    double rho0, rho1;
    if (s <= ep.a2) {
      const double v = 1.0 - s / ep.a2, v2 = v * v;
      if (ep.variant == 0) { rho0 = ep.a2 / 6.0 * (1.0 - v2 * v); rho1 = 0.5 * v2; }
      else { rho0 = ep.a2 / 3.0 * (1.0 - v2 * v); rho1 = v2; }
    } else {
      rho0 = (ep.variant == 0) ? ep.a2 / 6.0 : ep.a2 / 3.0; rho1 = 0;
    }
    const double w = c * rho1;
    // J = [n^T, -n^T R [p]x] = [n^T, (p x R^T n)^T]  (lidarOptimization.cpp:68-74)
    const double m0 = R00 * nd.x + R10 * nd.y + R20 * nd.z;
    const double m1 = R01 * nd.x + R11 * nd.y + R21 * nd.z;
    const double m2 = R02 * nd.x + R12 * nd.y + R22 * nd.z;
    const double J[6] = {nd.x, nd.y, nd.z, py * m2 - pz * m1, pz * m0 - px * m2, px * m1 - py * m0};
    acc[0] += 0.5 * c * rho0;
    acc[1] += 1.0;
    const double wr = w * r;
    int k = 8;
#pragma unroll
    for (int a = 0; a < 6; ++a) {
      acc[2 + a] += J[a] * wr;
      const double wj = w * J[a];
#pragma unroll
      for (int b = a; b < 6; ++b) acc[k++] += wj * J[b];
    }
  }
  // wavefront shuffle reduction -> LDS -> one partial record per workgroup
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int a = 0; a < kNAcc; ++a) {
    const double v = wave_sum(acc[a]);
    if (lane == 0) red[wave][a] = v;
  }
  __syncthreads();
  if (threadIdx.x < kNAcc) {
    const double v = ((red[0][threadIdx.x] + red[1][threadIdx.x]) + red[2][threadIdx.x]) + red[3][threadIdx.x];
    partials[(size_t)blockIdx.x * kSumsStride + threadIdx.x] = v;
  }
  // last-workgroup finish (agent-scope release by the producers, acquire by the consumer)
  __syncthreads();
  if (threadIdx.x == 0) {
    __threadfence();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // keep the write-back ahead of the ticket (ROCm 7.2 hazard)
    const uint32_t t = atomicAdd(ticket, 1u);
    is_last = (t == gridDim.x - 1);
    if (is_last) __threadfence();
  }
  __syncthreads();
  if (!is_last) return;
  {
    const int v = threadIdx.x & 31, chunk = threadIdx.x >> 5;  // 8 chunks x 32 values, fixed order
    double s = 0;
    if (v < kNAcc)
      for (uint32_t b = chunk; b < gridDim.x; b += 8) s += __builtin_nontemporal_load(&partials[(size_t)b * kSumsStride + v]);
    fin[chunk][v] = s;
  }
  __syncthreads();
  if (threadIdx.x < kNAcc) {
    double s = 0;
#pragma unroll
    for (int cidx = 0; cidx < 8; ++cidx) s += fin[cidx][threadIdx.x];
    double* o = reinterpret_cast<double*>(out);
    o[threadIdx.x] = s;  // LmSums layout: cost, count, Jtr[6], JtJ[21]
  } else if (threadIdx.x >= 32 && threadIdx.x < 48) {
    reinterpret_cast<double*>(out)[kNAcc + (threadIdx.x - 32)] = (double)hist[threadIdx.x - 32];
  }
  if (threadIdx.x == 0) *ticket = 0;  // re-arm for the next launch on this stream
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
    nbr[((size_t)i * k + t) * 3] = map.x[id]; nbr[((size_t)i * k + t) * 3 + 1] = map.y[id]; nbr[((size_t)i * k + t) * 3 + 2] = map.z[id];
  }
}

__device__ __forceinline__ unsigned long long wave_min_u64(unsigned long long v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const unsigned long long o = __shfl_xor(v, off, 64);
    v = o < v ? o : v;
  }
  return v;
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
    const float d2 = l2_d2(qx, qy, qz, map.x[p], map.y[p], map.z[p]);
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
        nbr[((size_t)i * k + t) * 3] = map.x[id]; nbr[((size_t)i * k + t) * 3 + 1] = map.y[id]; nbr[((size_t)i * k + t) * 3 + 2] = map.z[id];
      } else {  // fewer than k points in the cube: nanoflann.h:87-100 buffer state
        d2o[(size_t)i * k + t] = (t == k - 1) ? 3.402823466e+38f : 0.f;
        if (idxo) idxo[(size_t)i * k + t] = (int32_t)beg;
        nbr[((size_t)i * k + t) * 3] = map.x[beg]; nbr[((size_t)i * k + t) * 3 + 1] = map.y[beg]; nbr[((size_t)i * k + t) * 3 + 2] = map.z[beg];
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// launch wrappers
// ------------------------------------------------------------------------------------------------
static inline dim3 grid_for(uint32_t n, int block) { return dim3((n + block - 1) / block); }

size_t sort_temp_bytes(size_t n) {
  size_t bytes = 0;
  (void)rocprim::radix_sort_pairs(nullptr, bytes, (const uint32_t*)nullptr, (uint32_t*)nullptr, (const uint32_t*)nullptr,
                                  (uint32_t*)nullptr, n, 0, 32, (hipStream_t)0);
  return bytes;
}

void launch_scan_keys(const float* d_scan, uint32_t n, const Pose& pose, const DevMapView& map, int max_sf, int rank,
                      int world, uint32_t* keys, uint32_t* vals, uint32_t* n_kept, hipStream_t s) {
  if (!n) return;
  hipLaunchKernelGGL(scan_keys_kernel, grid_for(n, 256), dim3(256), 0, s, d_scan, n, pose, map, max_sf, rank, world, keys, vals, n_kept);
}
void launch_sort_pairs(void* tmp, size_t tmp_bytes, const uint32_t* ki, uint32_t* ko, const uint32_t* vi, uint32_t* vo,
                       uint32_t n, hipStream_t s) {
  if (!n) return;
  (void)rocprim::radix_sort_pairs(tmp, tmp_bytes, ki, ko, vi, vo, (size_t)n, 0, 32, s);
}
void launch_gather_scan(const float* d_scan, const uint32_t* perm, uint32_t n_kept, float* spx, float* spy, float* spz, hipStream_t s) {
  if (!n_kept) return;
  hipLaunchKernelGGL(gather_scan_kernel, grid_for(n_kept, 256), dim3(256), 0, s, d_scan, perm, n_kept, spx, spy, spz);
}
void launch_knn_plane(const float* spx, const float* spy, const float* spz, uint32_t n_kept, const Pose& pose,
                      const DevMapView& map, const MatchParams& mp, CorrBuffers corr, int32_t* hist, hipStream_t s) {
  if (!n_kept) return;
  hipLaunchKernelGGL(knn_plane_kernel, grid_for(n_kept, 256), dim3(256), 0, s, spx, spy, spz, n_kept, pose, map, mp, corr, hist);
}
void launch_eval(const float* spx, const float* spy, const float* spz, const CorrBuffers& corr, uint32_t n_kept,
                 const Pose& pose, const EvalParams& ep, double* partials, uint32_t* ticket, const int32_t* hist,
                 LmSums* sums, hipStream_t s) {
  hipLaunchKernelGGL(eval_kernel, dim3(kEvalBlocks), dim3(256), 0, s, spx, spy, spz, corr, n_kept, pose, ep, partials, ticket, hist, sums);
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
