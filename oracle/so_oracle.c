/*
 * so_oracle.c -- CPU ORACLE (test infrastructure; see so_oracle.h header for the contract and
 * the pinning status: k-NN pinned against the reference's own octree.h; Ceres/Eigen/PCL/tf2
 * stages restated from published algorithms = "PARITY UNPINNED" for those stages).
 *
 * Citations `X.cpp:line` are relative to /root/reference/super_odometry/{src,include/super_odometry}/...
 *   LS  = src/LidarProcess/LidarSlam.cpp            LM  = include/super_odometry/LidarProcess/LocalMap.h
 *   oct = include/super_odometry/flann/octree.h     nf  = include/super_odometry/flann/nanoflann.h
 *   lopt= src/LaserMapping/lidarOptimization.cpp    plp = src/LidarProcess/pose_local_parameterization.cpp
 *   sutil = include/super_odometry/utils/superodom_utils.h
 */
#include "so_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ============================================================================================ */
/* small math                                                                                   */
/* ============================================================================================ */
static void cross3(const double a[3], const double b[3], double o[3]) {
  o[0] = a[1] * b[2] - a[2] * b[1];
  o[1] = a[2] * b[0] - a[0] * b[2];
  o[2] = a[0] * b[1] - a[1] * b[0];
}
/* Eigen::Quaternion::_transformVector: v + w*(2 u x v) + u x (2 u x v), u = q.vec() */
static void quat_rotate(const double q[4] /*x y z w*/, const double v[3], double o[3]) {
  double uv[3], uuv[3];
  cross3(q, v, uv);
  uv[0] += uv[0]; uv[1] += uv[1]; uv[2] += uv[2];
  cross3(q, uv, uuv);
  o[0] = v[0] + q[3] * uv[0] + uuv[0];
  o[1] = v[1] + q[3] * uv[1] + uuv[1];
  o[2] = v[2] + q[3] * uv[2] + uuv[2];
}
static void quat_rotate_f(const float q[4], const float v[3], float o[3]) {
  float uv[3], uuv[3];
  uv[0] = q[1] * v[2] - q[2] * v[1]; uv[1] = q[2] * v[0] - q[0] * v[2]; uv[2] = q[0] * v[1] - q[1] * v[0];
  uv[0] += uv[0]; uv[1] += uv[1]; uv[2] += uv[2];
  uuv[0] = q[1] * uv[2] - q[2] * uv[1]; uuv[1] = q[2] * uv[0] - q[0] * uv[2]; uuv[2] = q[0] * uv[1] - q[1] * uv[0];
  o[0] = v[0] + q[3] * uv[0] + uuv[0];
  o[1] = v[1] + q[3] * uv[1] + uuv[1];
  o[2] = v[2] + q[3] * uv[2] + uuv[2];
}
static void quat_mul(const double a[4], const double b[4], double o[4]) { /* Eigen order x y z w */
  o[3] = a[3] * b[3] - a[0] * b[0] - a[1] * b[1] - a[2] * b[2];
  o[0] = a[3] * b[0] + a[0] * b[3] + a[1] * b[2] - a[2] * b[1];
  o[1] = a[3] * b[1] + a[1] * b[3] + a[2] * b[0] - a[0] * b[2];
  o[2] = a[3] * b[2] + a[2] * b[3] + a[0] * b[1] - a[1] * b[0];
}
static void quat_normalize(double q[4]) {
  double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  q[0] /= n; q[1] /= n; q[2] /= n; q[3] /= n;
}
static void quat_to_R(const double q[4], double R[9] /*row-major*/) { /* Eigen toRotationMatrix */
  double tx = 2 * q[0], ty = 2 * q[1], tz = 2 * q[2];
  double twx = tx * q[3], twy = ty * q[3], twz = tz * q[3];
  double txx = tx * q[0], txy = ty * q[0], txz = tz * q[0];
  double tyy = ty * q[1], tyz = tz * q[1], tzz = tz * q[2];
  R[0] = 1 - (tyy + tzz); R[1] = txy - twz; R[2] = txz + twy;
  R[3] = txy + twz; R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
  R[6] = txz - twy; R[7] = tyz + twx; R[8] = 1 - (txx + tyy);
}
/* pose^-1 * pose2 as (translation norm, rotation angle): LS:203-208, 246-249 (Twist.h:172-185) */
static void relative_motion(const double a[7], const double b[7], double *tn, double *rn) {
  double qi[4] = {-a[3], -a[4], -a[5], a[6]};
  double dt[3] = {b[0] - a[0], b[1] - a[1], b[2] - a[2]}, t[3], dq[4];
  quat_rotate(qi, dt, t);
  quat_mul(qi, b + 3, dq);
  if (dq[3] < 0) { dq[0] = -dq[0]; dq[1] = -dq[1]; dq[2] = -dq[2]; dq[3] = -dq[3]; } /* matrix->quaternion yields w >= 0 */
  *tn = sqrt(t[0] * t[0] + t[1] * t[1] + t[2] * t[2]);
  *rn = 2 * atan2(sqrt(dq[0] * dq[0] + dq[1] * dq[1] + dq[2] * dq[2]), dq[3]);
}

/* ============================================================================================ */
/* 3x3 symmetric eigen-solver (restates Eigen::SelfAdjointEigenSolver<Matrix3d> results:         */
/* ascending eigenvalues, orthonormal eigenvectors; cyclic Jacobi in fp64)                       */
/* ============================================================================================ */
void orc_eig3_sym(const double S[9], double ev[3], double V[9]) {
  double a[3][3] = {{S[0], S[1], S[2]}, {S[3], S[4], S[5]}, {S[6], S[7], S[8]}};
  double v[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
  for (int sweep = 0; sweep < 64; ++sweep) {
    double off = a[0][1] * a[0][1] + a[0][2] * a[0][2] + a[1][2] * a[1][2];
    double diag = a[0][0] * a[0][0] + a[1][1] * a[1][1] + a[2][2] * a[2][2];
    if (off <= 1e-40 * diag || off == 0.0) break;
    for (int p = 0; p < 2; ++p)
      for (int q = p + 1; q < 3; ++q) {
        if (a[p][q] == 0.0) continue;
        double theta = (a[q][q] - a[p][p]) / (2.0 * a[p][q]);
        double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
        double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
        for (int k = 0; k < 3; ++k) { /* A <- A J */
          double akp = a[k][p], akq = a[k][q];
          a[k][p] = c * akp - s * akq;
          a[k][q] = s * akp + c * akq;
        }
        for (int k = 0; k < 3; ++k) { /* A <- J^T A */
          double apk = a[p][k], aqk = a[q][k];
          a[p][k] = c * apk - s * aqk;
          a[q][k] = s * apk + c * aqk;
        }
        for (int k = 0; k < 3; ++k) {
          double vkp = v[k][p], vkq = v[k][q];
          v[k][p] = c * vkp - s * vkq;
          v[k][q] = s * vkp + c * vkq;
        }
      }
  }
  int idx[3] = {0, 1, 2};
  double d[3] = {a[0][0], a[1][1], a[2][2]};
  for (int i = 0; i < 2; ++i)
    for (int j = 0; j < 2 - i; ++j)
      if (d[idx[j]] > d[idx[j + 1]]) { int t = idx[j]; idx[j] = idx[j + 1]; idx[j + 1] = t; }
  for (int j = 0; j < 3; ++j) {
    ev[j] = d[idx[j]];
    for (int i = 0; i < 3; ++i) V[3 * j + i] = v[i][idx[j]];
  }
}

/* ============================================================================================ */
/* 5x3 least squares  A x = -1  by column-pivoted Householder QR (LS:798-806,                    */
/* Eigen::ColPivHouseholderQR restated).  Returns 0 if x is not finite.                          */
/* ============================================================================================ */
int orc_plane_ls5(const double Ain[15], double x[3]) {
  double A[5][3], b[5];
  int perm[3] = {0, 1, 2};
  for (int i = 0; i < 5; ++i) {
    for (int j = 0; j < 3; ++j) A[i][j] = Ain[3 * i + j];
    b[i] = -1.0;
  }
  for (int k = 0; k < 3; ++k) {
    /* pivot: remaining column with the largest remaining squared norm */
    int piv = k;
    double best = -1.0;
    for (int j = k; j < 3; ++j) {
      double s = 0;
      for (int i = k; i < 5; ++i) s += A[i][j] * A[i][j];
      if (s > best) { best = s; piv = j; }
    }
    if (piv != k) {
      for (int i = 0; i < 5; ++i) { double t = A[i][k]; A[i][k] = A[i][piv]; A[i][piv] = t; }
      int t = perm[k]; perm[k] = perm[piv]; perm[piv] = t;
    }
    /* Householder vector for column k, rows k..4 */
    double alpha = 0;
    for (int i = k; i < 5; ++i) alpha += A[i][k] * A[i][k];
    alpha = sqrt(alpha);
    if (alpha == 0.0) continue;
    if (A[k][k] > 0) alpha = -alpha;
    double v[5] = {0, 0, 0, 0, 0};
    for (int i = k; i < 5; ++i) v[i] = A[i][k];
    v[k] -= alpha;
    double vnorm2 = 0;
    for (int i = k; i < 5; ++i) vnorm2 += v[i] * v[i];
    if (vnorm2 == 0.0) continue;
    for (int j = k; j < 3; ++j) {
      double dot = 0;
      for (int i = k; i < 5; ++i) dot += v[i] * A[i][j];
      double f = 2.0 * dot / vnorm2;
      for (int i = k; i < 5; ++i) A[i][j] -= f * v[i];
    }
    double dot = 0;
    for (int i = k; i < 5; ++i) dot += v[i] * b[i];
    double f = 2.0 * dot / vnorm2;
    for (int i = k; i < 5; ++i) b[i] -= f * v[i];
  }
  double y[3];
  for (int k = 2; k >= 0; --k) {
    double s = b[k];
    for (int j = k + 1; j < 3; ++j) s -= A[k][j] * y[j];
    y[k] = s / A[k][k];
  }
  for (int k = 0; k < 3; ++k) x[perm[k]] = y[k];
  return isfinite(x[0]) && isfinite(x[1]) && isfinite(x[2]);
}

/* ============================================================================================ */
/* LocalMap                                                                                      */
/* ============================================================================================ */
typedef struct {
  float *xyz; /* AoS xyz, n points (pcl::PointXYZI minus intensity, LM:40) */
  size_t n, cap;
  /* exact-search acceleration grid (oracle-internal; replaces the reference's octree) */
  int grid_valid, gn;
  double gcell, gmin[3];
  int32_t *gstart; /* gn^3+1 */
  int32_t *gidx;   /* point indices sorted by cell */
} orc_cube;

struct orc_map {
  orc_cube *cubes[ORC_MAP_NUM]; /* map_, LM:758 */
  int origin[3];                /* origin_, LM:763 */
  float lineRes, planeRes;      /* LM:760-761 */
};

static void cube_free(orc_cube *c) {
  if (!c) return;
  free(c->xyz); free(c->gstart); free(c->gidx); free(c);
}
static void cube_push(orc_cube *c, const float p[3]) {
  if (c->n == c->cap) {
    c->cap = c->cap ? c->cap * 2 : 256;
    c->xyz = (float *)realloc(c->xyz, c->cap * 3 * sizeof(float));
  }
  memcpy(c->xyz + 3 * c->n, p, 3 * sizeof(float));
  c->n++;
  c->grid_valid = 0;
}

orc_map *orc_map_create(void) {
  orc_map *m = (orc_map *)calloc(1, sizeof(orc_map));
  /* LM:141-144: origin_ = (W*0.5, H*0.5, D*0.5) truncated to int */
  m->origin[0] = (int)(ORC_MAP_W * 0.5); m->origin[1] = (int)(ORC_MAP_H * 0.5); m->origin[2] = (int)(ORC_MAP_D * 0.5);
  m->lineRes = 0.2f; m->planeRes = 0.4f; /* LM:760-761 defaults */
  return m;
}
void orc_map_destroy(orc_map *m) {
  if (!m) return;
  for (int i = 0; i < ORC_MAP_NUM; ++i) cube_free(m->cubes[i]);
  free(m);
}
void orc_map_set_resolution(orc_map *m, float lineRes, float planeRes) { m->lineRes = lineRes; m->planeRes = planeRes; }
void orc_map_get_origin(const orc_map *m, int o[3]) { o[0] = m->origin[0]; o[1] = m->origin[1]; o[2] = m->origin[2]; }

/* LM:146-164 / 488-497: int((c + 25.0)/50.0) (+origin), then -- when c + 25.0 < 0 */
static int cube_coord(double c, int origin) {
  int i = (int)((c + ORC_HALF_CUBE) / ORC_CUBE) + origin;
  if (c + ORC_HALF_CUBE < 0) i--;
  return i;
}
void orc_map_set_origin(orc_map *m, const double t[3], int o[3]) {
  for (int a = 0; a < 3; ++a) m->origin[a] = -cube_coord(t[a], 0);
  if (o) orc_map_get_origin(m, o);
}
#define CIDX(i, j, k) ((i) + ORC_MAP_W * (j) + ORC_MAP_W * ORC_MAP_H * (k))

void orc_map_shift(orc_map *m, const double t[3], int pos[3]) { /* LM:169-287 */
  int ci = cube_coord(t[0], m->origin[0]);
  int cj = cube_coord(t[1], m->origin[1]);
  int ck = cube_coord(t[2], m->origin[2]);
  orc_cube **M = m->cubes;
  while (ci < 3) {
    for (int j = 0; j < ORC_MAP_H; ++j) for (int k = 0; k < ORC_MAP_D; ++k) {
      cube_free(M[CIDX(ORC_MAP_W - 1, j, k)]);
      for (int i = ORC_MAP_W - 1; i >= 1; --i) M[CIDX(i, j, k)] = M[CIDX(i - 1, j, k)];
      M[CIDX(0, j, k)] = NULL;
    }
    ci++; m->origin[0]++;
  }
  while (ci >= ORC_MAP_W - 3) {
    for (int j = 0; j < ORC_MAP_H; ++j) for (int k = 0; k < ORC_MAP_D; ++k) {
      cube_free(M[CIDX(0, j, k)]);
      for (int i = 0; i < ORC_MAP_W - 1; ++i) M[CIDX(i, j, k)] = M[CIDX(i + 1, j, k)];
      M[CIDX(ORC_MAP_W - 1, j, k)] = NULL;
    }
    ci--; m->origin[0]--;
  }
  while (cj < 3) {
    for (int i = 0; i < ORC_MAP_W; ++i) for (int k = 0; k < ORC_MAP_D; ++k) {
      cube_free(M[CIDX(i, ORC_MAP_H - 1, k)]);
      for (int j = ORC_MAP_H - 1; j >= 1; --j) M[CIDX(i, j, k)] = M[CIDX(i, j - 1, k)];
      M[CIDX(i, 0, k)] = NULL;
    }
    cj++; m->origin[1]++;
  }
  while (cj >= ORC_MAP_H - 3) {
    for (int i = 0; i < ORC_MAP_W; ++i) for (int k = 0; k < ORC_MAP_D; ++k) {
      cube_free(M[CIDX(i, 0, k)]);
      for (int j = 0; j < ORC_MAP_H - 1; ++j) M[CIDX(i, j, k)] = M[CIDX(i, j + 1, k)];
      M[CIDX(i, ORC_MAP_H - 1, k)] = NULL;
    }
    cj--; m->origin[1]--;
  }
  while (ck < 3) {
    for (int i = 0; i < ORC_MAP_W; ++i) for (int j = 0; j < ORC_MAP_H; ++j) {
      cube_free(M[CIDX(i, j, ORC_MAP_D - 1)]);
      for (int k = ORC_MAP_D - 1; k >= 1; --k) M[CIDX(i, j, k)] = M[CIDX(i, j, k - 1)];
      M[CIDX(i, j, 0)] = NULL;
    }
    ck++; m->origin[2]++;
  }
  while (ck >= ORC_MAP_D - 3) {
    for (int i = 0; i < ORC_MAP_W; ++i) for (int j = 0; j < ORC_MAP_H; ++j) {
      cube_free(M[CIDX(i, j, 0)]);
      for (int k = 0; k < ORC_MAP_D - 1; ++k) M[CIDX(i, j, k)] = M[CIDX(i, j, k + 1)];
      M[CIDX(i, j, ORC_MAP_D - 1)] = NULL;
    }
    ck--; m->origin[2]--;
  }
  pos[0] = ci; pos[1] = cj; pos[2] = ck;
}

int orc_map_count_5x5(const orc_map *m, const int pos[3]) { /* LM:292-318 */
  int n = 0;
  for (int i = pos[0] - 2; i <= pos[0] + 2; ++i)
    for (int j = pos[1] - 2; j <= pos[1] + 2; ++j)
      for (int k = pos[2] - 1; k <= pos[2] + 1; ++k)
        if (i >= 0 && i < ORC_MAP_W && j >= 0 && j < ORC_MAP_H && k >= 0 && k < ORC_MAP_D) {
          const orc_cube *c = m->cubes[CIDX(i, j, k)];
          if (c) n += (int)c->n;
        }
  return n;
}
size_t orc_map_size(const orc_map *m) {
  size_t n = 0;
  for (int i = 0; i < ORC_MAP_NUM; ++i) if (m->cubes[i]) n += m->cubes[i]->n;
  return n;
}
size_t orc_map_cube_size(const orc_map *m, int ci) { return (ci >= 0 && ci < ORC_MAP_NUM && m->cubes[ci]) ? m->cubes[ci]->n : 0; }
size_t orc_map_export(const orc_map *m, float *xyz, size_t cap) {
  size_t n = 0;
  for (int i = 0; i < ORC_MAP_NUM; ++i) {
    const orc_cube *c = m->cubes[i];
    if (!c) continue;
    for (size_t j = 0; j < c->n && n < cap; ++j, ++n) memcpy(xyz + 3 * n, c->xyz + 3 * j, 12);
  }
  return n;
}

/* pcl::VoxelGrid<PointXYZI>::applyFilter (PCL 1.12.1, [UPSTREAM]) restated: leaf index =
 * floor(x * inv_leaf) - min_b per axis in float; centroids accumulated in float in input order
 * (upstream's within-leaf order is unspecified), emitted in ascending linear leaf index. */
typedef struct { uint32_t idx; uint32_t pt; } vg_pair;
static int vg_cmp(const void *a, const void *b) {
  const vg_pair *x = (const vg_pair *)a, *y = (const vg_pair *)b;
  if (x->idx != y->idx) return x->idx < y->idx ? -1 : 1;
  return x->pt < y->pt ? -1 : (x->pt > y->pt ? 1 : 0);
}
size_t orc_voxel_grid(const float *xyz, size_t n, float leaf, float *out) {
  if (n == 0) return 0;
  float inv = 1.0f / leaf;
  float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
  for (size_t i = 0; i < n; ++i)
    for (int a = 0; a < 3; ++a) {
      float v = xyz[3 * i + a];
      if (v < mn[a]) mn[a] = v;
      if (v > mx[a]) mx[a] = v;
    }
  int64_t dx = (int64_t)((mx[0] - mn[0]) * inv) + 1, dy = (int64_t)((mx[1] - mn[1]) * inv) + 1, dz = (int64_t)((mx[2] - mn[2]) * inv) + 1;
  if (dx * dy * dz > (int64_t)INT32_MAX) { /* "Leaf size is too small": output = input */
    memcpy(out, xyz, n * 12);
    return n;
  }
  int minb[3], maxb[3], divb[3];
  for (int a = 0; a < 3; ++a) {
    minb[a] = (int)floorf(mn[a] * inv);
    maxb[a] = (int)floorf(mx[a] * inv);
    divb[a] = maxb[a] - minb[a] + 1;
  }
  int mul[3] = {1, divb[0], divb[0] * divb[1]};
  vg_pair *pairs = (vg_pair *)malloc(n * sizeof(vg_pair));
  for (size_t i = 0; i < n; ++i) {
    int ijk[3];
    for (int a = 0; a < 3; ++a) ijk[a] = (int)(floorf(xyz[3 * i + a] * inv) - (float)minb[a]);
    pairs[i].idx = (uint32_t)(ijk[0] * mul[0] + ijk[1] * mul[1] + ijk[2] * mul[2]);
    pairs[i].pt = (uint32_t)i;
  }
  qsort(pairs, n, sizeof(vg_pair), vg_cmp);
  size_t no = 0, i = 0;
  while (i < n) {
    size_t j = i;
    float s[3] = {0, 0, 0};
    while (j < n && pairs[j].idx == pairs[i].idx) {
      const float *p = xyz + 3 * pairs[j].pt;
      s[0] += p[0]; s[1] += p[1]; s[2] += p[2];
      ++j;
    }
    float cnt = (float)(j - i);
    out[3 * no + 0] = s[0] / cnt; out[3 * no + 1] = s[1] / cnt; out[3 * no + 2] = s[2] / cnt;
    ++no;
    i = j;
  }
  free(pairs);
  return no;
}

static int map_cube_index(const orc_map *m, const float p[3]) { /* LM:596-610 (same rule as :488-502) */
  int ci = cube_coord((double)p[0], m->origin[0]);
  int cj = cube_coord((double)p[1], m->origin[1]);
  int ck = cube_coord((double)p[2], m->origin[2]);
  if (!(ci >= 0 && ci < ORC_MAP_W && cj >= 0 && cj < ORC_MAP_H && ck >= 0 && ck < ORC_MAP_D)) return -1;
  return CIDX(ci, cj, ck);
}

static int map_add(orc_map *m, const float *xyz, size_t n, size_t stride, int filter) {
  if (stride == 0) stride = 3;
  unsigned char *touched = (unsigned char *)calloc(ORC_MAP_NUM, 1);
  int inserted = 0;
  for (size_t i = 0; i < n; ++i) {
    const float *p = xyz + i * stride;
    int ci = map_cube_index(m, p);
    if (ci < 0) continue;
    if (!m->cubes[ci]) m->cubes[ci] = (orc_cube *)calloc(1, sizeof(orc_cube));
    cube_push(m->cubes[ci], p);
    touched[ci] = 1;
    inserted++;
  }
  if (filter) {
    for (int ci = 0; ci < ORC_MAP_NUM; ++ci) {
      if (!touched[ci]) continue;
      orc_cube *c = m->cubes[ci];
      float *out = (float *)malloc(c->n * 12);
      size_t no = orc_voxel_grid(c->xyz, c->n, m->planeRes, out);
      free(c->xyz);
      c->xyz = out; c->n = no; c->cap = c->n; c->grid_valid = 0;
    }
  }
  free(touched);
  return inserted;
}
int orc_map_add_surf(orc_map *m, const float *xyz, size_t n, size_t stride) { return map_add(m, xyz, n, stride, 1); }
int orc_map_add_surf_raw(orc_map *m, const float *xyz, size_t n, size_t stride) { return map_add(m, xyz, n, stride, 0); }

/* ============================================================================================ */
/* exact k-NN inside one cube                                                                    */
/* ============================================================================================ */
/* oct:93-102: float differences, squares via std::pow(float,int) -> double, summed in double,
 * narrowed to float on return. */
static inline float l2_compute(const float q[3], const float p[3]) {
  float d1 = q[0] - p[0], d2 = q[1] - p[1], d3 = q[2] - p[2];
  return (float)((double)d1 * (double)d1 + (double)d2 * (double)d2 + (double)d3 * (double)d3);
}
/* sorted insert = nf:117-147 with an explicit (d2, index) total order: the reference keeps the
 * earlier-VISITED point on ties (strict > at nf:124); visit order is octree-internal, so the exact
 * oracle defines ties by ascending in-cube index. */
typedef struct { float d2[16]; int64_t idx[16]; int k, count; } knn_set;
static inline void knn_init(knn_set *s, int k) {
  s->k = k; s->count = 0;
  for (int i = 0; i < k; ++i) { s->d2[i] = 0; s->idx[i] = 0; }
  if (k) s->d2[k - 1] = FLT_MAX; /* nf:99 */
}
static inline void knn_add(knn_set *s, float d, int64_t id) {
  int i;
  for (i = s->count; i > 0; --i) {
    int worse = (s->d2[i - 1] > d) || (s->d2[i - 1] == d && s->idx[i - 1] > id);
    if (worse) {
      if (i < s->k) { s->d2[i] = s->d2[i - 1]; s->idx[i] = s->idx[i - 1]; }
    } else break;
  }
  if (i < s->k) { s->d2[i] = d; s->idx[i] = id; }
  if (s->count < s->k) s->count++;
}

#define ORC_GRID_CELL 1.0
static void cube_build_grid(orc_cube *c, const orc_map *m, int cube_ind) {
  int ci = cube_ind % ORC_MAP_W, cj = (cube_ind / ORC_MAP_W) % ORC_MAP_H, ck = cube_ind / (ORC_MAP_W * ORC_MAP_H);
  c->gn = (int)(ORC_CUBE / ORC_GRID_CELL);
  c->gcell = ORC_GRID_CELL;
  c->gmin[0] = (ci - m->origin[0]) * ORC_CUBE - ORC_HALF_CUBE;
  c->gmin[1] = (cj - m->origin[1]) * ORC_CUBE - ORC_HALF_CUBE;
  c->gmin[2] = (ck - m->origin[2]) * ORC_CUBE - ORC_HALF_CUBE;
  size_t ncell = (size_t)c->gn * c->gn * c->gn;
  free(c->gstart); free(c->gidx);
  c->gstart = (int32_t *)calloc(ncell + 1, sizeof(int32_t));
  c->gidx = (int32_t *)malloc((c->n ? c->n : 1) * sizeof(int32_t));
  int32_t *cell_of = (int32_t *)malloc((c->n ? c->n : 1) * sizeof(int32_t));
  for (size_t i = 0; i < c->n; ++i) {
    int g[3];
    for (int a = 0; a < 3; ++a) {
      int v = (int)floor(((double)c->xyz[3 * i + a] - c->gmin[a]) / c->gcell);
      g[a] = v < 0 ? 0 : (v >= c->gn ? c->gn - 1 : v);
    }
    cell_of[i] = (g[2] * c->gn + g[1]) * c->gn + g[0];
    c->gstart[cell_of[i] + 1]++;
  }
  for (size_t k = 0; k < ncell; ++k) c->gstart[k + 1] += c->gstart[k];
  int32_t *fill = (int32_t *)malloc(ncell * sizeof(int32_t));
  memcpy(fill, c->gstart, ncell * sizeof(int32_t));
  for (size_t i = 0; i < c->n; ++i) c->gidx[fill[cell_of[i]]++] = (int32_t)i; /* ascending index inside a cell */
  free(fill); free(cell_of);
  c->grid_valid = 1;
}
static void map_ensure_grids(orc_map *m) {
  for (int i = 0; i < ORC_MAP_NUM; ++i)
    if (m->cubes[i] && !m->cubes[i]->grid_valid) cube_build_grid(m->cubes[i], m, i);
}

static void knn_cube_brute(const orc_cube *c, const float q[3], knn_set *s) {
  for (size_t i = 0; i < c->n; ++i) knn_add(s, l2_compute(q, c->xyz + 3 * i), (int64_t)i);
}
/* exact shell-expanding search: after all cells at Chebyshev distance <= s from the query's cell
 * were visited, every unvisited point is farther than s*cell, so we may stop once the current k-th
 * distance is strictly below that bound. */
static void knn_cube_grid(const orc_cube *c, const float q[3], knn_set *s) {
  int g[3];
  for (int a = 0; a < 3; ++a) {
    int v = (int)floor(((double)q[a] - c->gmin[a]) / c->gcell);
    g[a] = v < 0 ? 0 : (v >= c->gn ? c->gn - 1 : v);
  }
  int gn = c->gn;
  for (int sh = 0; sh < gn; ++sh) {
    int z0 = g[2] - sh, z1 = g[2] + sh, y0 = g[1] - sh, y1 = g[1] + sh, x0 = g[0] - sh, x1 = g[0] + sh;
    for (int z = (z0 < 0 ? 0 : z0); z <= (z1 >= gn ? gn - 1 : z1); ++z)
      for (int y = (y0 < 0 ? 0 : y0); y <= (y1 >= gn ? gn - 1 : y1); ++y) {
        int on_zy_face = (z == z0 || z == z1 || y == y0 || y == y1);
        for (int x = (x0 < 0 ? 0 : x0); x <= (x1 >= gn ? gn - 1 : x1); ++x) {
          if (!on_zy_face && x != x0 && x != x1) continue; /* interior of the shell: already visited */
          size_t cell = ((size_t)z * gn + y) * gn + x;
          for (int32_t t = c->gstart[cell]; t < c->gstart[cell + 1]; ++t) {
            int32_t i = c->gidx[t];
            knn_add(s, l2_compute(q, c->xyz + 3 * i), (int64_t)i);
          }
        }
      }
    if (s->count == s->k) {
      double bound = (double)sh * c->gcell * (1.0 - 1e-6);
      if ((double)s->d2[s->k - 1] < bound * bound) return;
    }
    if (z0 <= 0 && y0 <= 0 && x0 <= 0 && z1 >= gn - 1 && y1 >= gn - 1 && x1 >= gn - 1) return;
  }
}

/* Oracle-B: the k-NN of a cube answered by an EXTERNAL engine -- oracle/_ref/libref_octree.so, the reference's
 * own flann/octree.h compiled where it lies (Octree::knnNeighbors, oct:509-519, 1004-1055, with its two pruning bugs) --
 * while everything else stays this restatement.  use_grid_knn == 2 selects it; the hook owns one tree per cube. */
static orc_knn_hook_t g_knn_hook = 0;
void orc_set_knn_hook(orc_knn_hook_t h) { g_knn_hook = h; }

int orc_knn_surf(const orc_map *m, const float q[3], int k, int use_grid, float *nbr, float *d2, int64_t *idx, int *cube_out) {
  int ci = map_cube_index(m, q); /* LM:488-502 */
  if (cube_out) *cube_out = ci;
  if (ci < 0) return 0;
  const orc_cube *c = m->cubes[ci];
  if (!c || c->n == 0) return 0; /* LM:506: no tree */
  if (use_grid == 2 && g_knn_hook) { /* LM:516-523 with the stock octree: indices value-initialised to 0, distances resized */
    int64_t id[16]; float dd[16];
    for (int i = 0; i < k; ++i) { id[i] = 0; dd[i] = 0; }
    g_knn_hook(ci, c->xyz, c->n, q, k, id, dd);
    for (int i = 0; i < k; ++i) {
      if (d2) d2[i] = dd[i];
      if (idx) idx[i] = id[i];
      if (nbr) memcpy(nbr + 3 * i, c->xyz + 3 * id[i], 12);
    }
    return 1;
  }
  knn_set s;
  knn_init(&s, k);
  if (use_grid && c->grid_valid) knn_cube_grid(c, q, &s);
  else knn_cube_brute(c, q, &s);
  for (int i = 0; i < k; ++i) {
    if (d2) d2[i] = s.d2[i];
    if (idx) idx[i] = s.idx[i];
    if (nbr) memcpy(nbr + 3 * i, c->xyz + 3 * s.idx[i], 12); /* LM:522-523 (unfilled idx stays 0) */
  }
  return 1;
}

/* ============================================================================================ */
/* per-point correspondence  (ComputePlaneDistanceParameters, LS:514-572)                        */
/* ============================================================================================ */
static void observability(const double pFinal[3], const double evals[3], const double normal[3],
                          const double pose[7], int32_t obs[4]) {
  /* LS:574-693. feature.pt is a pcl::PointNormal (float); directions are cast to float. */
  float pt[3] = {(float)pFinal[0], (float)pFinal[1], (float)pFinal[2]};
  float nf[3] = {(float)normal[0], (float)normal[1], (float)normal[2]};
  double l1 = sqrt(evals[2]), l2 = sqrt(evals[1]), l3 = sqrt(evals[0]); /* LS:605-607 */
  double planar_2 = (l2 - l3) / l1;                                     /* LS:620 */
  float qf[4] = {(float)pose[3], (float)pose[4], (float)pose[5], (float)pose[6]}; /* LS:629-631 */
  float ax[3][3];
  const float ex[3] = {1, 0, 0}, ey[3] = {0, 1, 0}, ez[3] = {0, 0, 1};
  quat_rotate_f(qf, ex, ax[0]); quat_rotate_f(qf, ey, ax[1]); quat_rotate_f(qf, ez, ax[2]);
  float cr[3] = {pt[1] * nf[2] - pt[2] * nf[1], pt[2] * nf[0] - pt[0] * nf[2], pt[0] * nf[1] - pt[1] * nf[0]}; /* LS:684 */
  float rot[6];
  for (int a = 0; a < 3; ++a) {
    float v = cr[0] * ax[a][0] + cr[1] * ax[a][1] + cr[2] * ax[a][2];
    rot[2 * a] = v; rot[2 * a + 1] = -v; /* LS:687-692 */
  }
  float planar_sq = (float)(planar_2 * planar_2); /* LS:644 */
  float tr[3];
  for (int a = 0; a < 3; ++a) tr[a] = planar_sq * fabsf(nf[0] * ax[a][0] + nf[1] * ax[a][1] + nf[2] * ax[a][2]); /* LS:646-651 */
  /* descending, stable (libstdc++ std::sort on <16 elements is an insertion sort): LS:671-672 */
  int ro[6] = {0, 1, 2, 3, 4, 5}, to[3] = {0, 1, 2};
  for (int i = 1; i < 6; ++i) { int v = ro[i], j = i; while (j > 0 && rot[v] > rot[ro[j - 1]]) { ro[j] = ro[j - 1]; --j; } ro[j] = v; }
  for (int i = 1; i < 3; ++i) { int v = to[i], j = i; while (j > 0 && tr[v] > tr[to[j - 1]]) { to[j] = to[j - 1]; --j; } to[j] = v; }
  obs[0] = ro[0]; obs[1] = ro[1]; obs[2] = 6 + to[0]; obs[3] = 6 + to[1]; /* LS:675-678 */
}

void orc_plane_match(const orc_map *m, const double pose[7], const float ps[3], const orc_config *cfg, orc_corr *out) {
  memset(out, 0, sizeof(*out));
  out->status = ORC_UNKNOWN;
  double pInit[3] = {(double)ps[0], (double)ps[1], (double)ps[2]}, pFinal[3];
  quat_rotate(pose + 3, pInit, pFinal); /* LS:397-398, Twist.h:187 */
  pFinal[0] += pose[0]; pFinal[1] += pose[1]; pFinal[2] += pose[2];
  memcpy(out->p, pInit, sizeof(pInit));
  const float planeRes = m->planeRes;
  const double square_max_dist = (double)(3 * planeRes); /* LS:526: float product widened */
  float q[3] = {(float)pFinal[0], (float)pFinal[1], (float)pFinal[2]}; /* LS:728-731 */
  int k = cfg->k > 0 ? cfg->k : 5;
  int64_t idx[16];
  if (!orc_knn_surf(m, q, k, cfg->use_grid_knn, out->nbr, out->d2, idx, NULL)) { out->status = ORC_NOT_ENOUGH_NEIGHBORS; return; } /* LS:736-739 */
  if ((double)out->d2[k - 1] > square_max_dist) { out->status = ORC_NEIGHBORS_TOO_FAR; return; }                                  /* LS:741-744 */
  /* PCA, LS:756-775 + sutil:143-151 */
  double mean[3] = {0, 0, 0}, S[9] = {0};
  for (int j = 0; j < 5; ++j) for (int a = 0; a < 3; ++a) mean[a] += (double)out->nbr[3 * j + a];
  for (int a = 0; a < 3; ++a) mean[a] /= 5.0;
  for (int j = 0; j < 5; ++j) {
    double c[3] = {(double)out->nbr[3 * j] - mean[0], (double)out->nbr[3 * j + 1] - mean[1], (double)out->nbr[3 * j + 2] - mean[2]};
    for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) S[3 * a + b] += c[a] * c[b];
  }
  double ev[3], V[9];
  orc_eig3_sym(S, ev, V);
  memcpy(out->eig, ev, sizeof(ev));
  if (ev[0] < 1e-6 || ev[1] / ev[2] < 0.1) { out->status = ORC_BAD_PCA_STRUCTURE; return; } /* LS:772 */
  /* plane LS, LS:798-816 */
  double A[15], x[3];
  for (int j = 0; j < 15; ++j) A[j] = (double)out->nbr[j];
  if (!orc_plane_ls5(A, x)) { out->status = ORC_INVALID_NUMERICAL; return; }
  double nrm = sqrt(x[0] * x[0] + x[1] * x[1] + x[2] * x[2]);
  double d = 1.0 / nrm;
  double n[3] = {x[0] / nrm, x[1] / nrm, x[2] / nrm};
  const double max_point_distance = (double)planeRes / 2.0; /* LS:820 */
  double sum = 0;
  for (int j = 0; j < 5; ++j) {
    double dist = fabs(n[0] * (double)out->nbr[3 * j] + n[1] * (double)out->nbr[3 * j + 1] + n[2] * (double)out->nbr[3 * j + 2] + d);
    if (dist > max_point_distance) { out->status = ORC_MSE_TOO_LARGE; return; } /* LS:832-835 */
    sum += dist;
  }
  double mean_abs = sum / 5.0; /* LS:841 ("meanSquareDist" is a mean ABSOLUTE distance) */
  /* normal orientation (vs world origin!) LS:553-561 */
  double normal[3] = {V[0], V[1], V[2]};
  if (pFinal[0] * normal[0] + pFinal[1] * normal[1] + pFinal[2] * normal[2] < 0) { normal[0] = -normal[0]; normal[1] = -normal[1]; normal[2] = -normal[2]; }
  observability(pFinal, ev, normal, pose, out->obs);
  out->coeff = 1.0 - sqrt(mean_abs / square_max_dist); /* LS:568 */
  memcpy(out->n, n, sizeof(n));
  out->d = d;
  out->status = ORC_SUCCESS;
}

/* ============================================================================================ */
/* residual / Jacobian / loss / parameterisation                                                 */
/* ============================================================================================ */
void orc_residual_jacobian(const double pose[7], const double p[3], const double n[3], double d, double *r, double J[6]) {
  double pw[3];
  quat_rotate(pose + 3, p, pw); /* lopt:59 */
  pw[0] += pose[0]; pw[1] += pose[1]; pw[2] += pose[2];
  *r = n[0] * pw[0] + n[1] * pw[1] + n[2] * pw[2] + d; /* lopt:61 */
  if (!J) return;
  double R[9];
  quat_to_R(pose + 3, R);
  /* skew(p), lopt:152-162; dp_by_so3 = [I, -R*skew(p)], lopt:68-71 */
  double sk[9] = {0, -p[2], p[1], p[2], 0, -p[0], -p[1], p[0], 0};
  double M[9];
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) {
    double s = 0;
    for (int k = 0; k < 3; ++k) s += R[3 * i + k] * sk[3 * k + j];
    M[3 * i + j] = -s;
  }
  J[0] = n[0]; J[1] = n[1]; J[2] = n[2];
  for (int j = 0; j < 3; ++j) J[3 + j] = n[0] * M[j] + n[1] * M[3 + j] + n[2] * M[6 + j]; /* lopt:74 */
}
/* plp:7-23 + utils/utility.h:12-24 */
void orc_pose_plus(const double x[7], const double dlt[6], double out[7]) {
  out[0] = x[0] + dlt[0]; out[1] = x[1] + dlt[1]; out[2] = x[2] + dlt[2];
  double dq[4] = {dlt[3] / 2.0, dlt[4] / 2.0, dlt[5] / 2.0, 1.0};
  quat_mul(x + 3, dq, out + 3);
  quat_normalize(out + 3);
}
/* ceres::TukeyLoss / ScaledLoss [UPSTREAM Ceres 2.0.0 loss_function.cc] */
void orc_tukey_scaled(double s, double a, double coeff, int variant, double rho[3]) {
  double a2 = a * a;
  if (s <= a2) {
    double v = 1.0 - s / a2, v2 = v * v;
    if (variant == 0) { rho[0] = a2 / 6.0 * (1.0 - v2 * v); rho[1] = 0.5 * v2; rho[2] = -1.0 / a2 * v; }
    else { rho[0] = a2 / 3.0 * (1.0 - v2 * v); rho[1] = v2; rho[2] = -2.0 / a2 * v; }
  } else {
    rho[0] = (variant == 0) ? a2 / 6.0 : a2 / 3.0; rho[1] = 0; rho[2] = 0;
  }
  rho[0] *= coeff; rho[1] *= coeff; rho[2] *= coeff; /* ScaledLoss */
}
static double tukey_a(float planeRes) { return (double)sqrtf(3 * planeRes); } /* LS:271: float sqrt of a float product */

/* Evaluate one residual block the way ceres::ResidualBlock::Evaluate does with a loss whose
 * rho'' <= 0: cost = rho/2, residual and Jacobian scaled by sqrt(rho') (corrector.cc). */
static void eval_block(const orc_corr *c, const double pose[7], double a, int variant, double *cost, double *r_out, double J_out[6]) {
  double r, J[6], rho[3];
  orc_residual_jacobian(pose, c->p, c->n, c->d, &r, J_out ? J : NULL);
  orc_tukey_scaled(r * r, a, c->coeff, variant, rho);
  *cost = 0.5 * rho[0];
  if (r_out) {
    double sc = sqrt(rho[1]);
    *r_out = sc * r;
    if (J_out) for (int j = 0; j < 6; ++j) J_out[j] = sc * J[j];
  }
}

void orc_evaluate(const orc_corr *c, size_t n, const double pose[7], float planeRes, int variant,
                  double *cost, double JtJ[36], double Jtr[6], int *count) {
  double a = tukey_a(planeRes), ctot = 0;
  int cnt = 0;
  memset(JtJ, 0, 36 * sizeof(double)); memset(Jtr, 0, 6 * sizeof(double));
  for (size_t i = 0; i < n; ++i) {
    if (c[i].status != ORC_SUCCESS) continue;
    double ci, r, J[6];
    eval_block(&c[i], pose, a, variant, &ci, &r, J);
    ctot += ci; cnt++;
    for (int p = 0; p < 6; ++p) { Jtr[p] += J[p] * r; for (int q = 0; q < 6; ++q) JtJ[6 * p + q] += J[p] * J[q]; }
  }
  *cost = ctot; if (count) *count = cnt;
}

/* dense Householder QR least squares: min || M y - rhs ||, M is rows x 6 (row-major). */
static int qr_solve6(double *M, double *rhs, int rows, double y[6]) {
  for (int k = 0; k < 6; ++k) {
    double alpha = 0;
    for (int i = k; i < rows; ++i) alpha += M[6 * i + k] * M[6 * i + k];
    alpha = sqrt(alpha);
    if (alpha == 0) return 0;
    if (M[6 * k + k] > 0) alpha = -alpha;
    double vk = M[6 * k + k] - alpha, vn2 = vk * vk;
    for (int i = k + 1; i < rows; ++i) vn2 += M[6 * i + k] * M[6 * i + k];
    if (vn2 == 0) return 0;
    for (int j = k + 1; j < 6; ++j) {
      double dot = vk * M[6 * k + j];
      for (int i = k + 1; i < rows; ++i) dot += M[6 * i + k] * M[6 * i + j];
      double f = 2 * dot / vn2;
      M[6 * k + j] -= f * vk;
      for (int i = k + 1; i < rows; ++i) M[6 * i + j] -= f * M[6 * i + k];
    }
    double dot = vk * rhs[k];
    for (int i = k + 1; i < rows; ++i) dot += M[6 * i + k] * rhs[i];
    double f = 2 * dot / vn2;
    rhs[k] -= f * vk;
    for (int i = k + 1; i < rows; ++i) rhs[i] -= f * M[6 * i + k];
    M[6 * k + k] = alpha;
  }
  for (int k = 5; k >= 0; --k) {
    double s = rhs[k];
    for (int j = k + 1; j < 6; ++j) s -= M[6 * k + j] * y[j];
    y[k] = s / M[6 * k + k];
  }
  for (int k = 0; k < 6; ++k) if (!isfinite(y[k])) return 0;
  return 1;
}

/* Restatement of ceres::internal::TrustRegionMinimizer + LevenbergMarquardtStrategy +
 * DenseQRSolver [UPSTREAM Ceres 2.0.0] with the options of LS:231-236 (everything else default).
 * See DESIGN.md "Ceres restatement" for the constants. */
void orc_lm_solve(const orc_corr *c_all, size_t n_all, double pose[7], float planeRes, const orc_config *cfg, orc_iter_stats *st) {
  const double a = tukey_a(planeRes);
  const int variant = cfg->tukey_variant;
  const int max_it = cfg->lm_max_iterations > 0 ? cfg->lm_max_iterations : 4;
  const double min_diag = 1e-6, max_diag = 1e32, max_radius = 1e16, min_radius = 1e-32;
  const double f_tol = 1e-6, g_tol = 1e-10, p_tol = 1e-8, min_rel_dec = 1e-3;
  /* compact accepted correspondences */
  size_t A = 0;
  const orc_corr **c = (const orc_corr **)malloc((n_all ? n_all : 1) * sizeof(*c));
  for (size_t i = 0; i < n_all; ++i) if (c_all[i].status == ORC_SUCCESS) c[A++] = &c_all[i];
  st->num_surf = (int32_t)A; st->lm_iterations = 0; st->num_successful_steps = 0; st->termination = 0;
  st->initial_cost = st->final_cost = 0;
  if (A == 0) { st->termination = 4; free(c); return; }
  int rows = (int)A + 6;
  double *J = (double *)malloc((size_t)rows * 6 * sizeof(double)); /* scaled Jacobian (A x 6) */
  double *r = (double *)malloc((size_t)rows * sizeof(double));
  double *M = (double *)malloc((size_t)rows * 6 * sizeof(double));
  double *rhs = (double *)malloc((size_t)rows * sizeof(double));
  double x[7], cand[7], scale[6], grad[6];
  memcpy(x, pose, sizeof(x));
  double x_cost = 0, radius = 1e4, decrease_factor = 2.0, diag[6];
  int reuse_diagonal = 0, have_scale = 0, invalid_steps = 0;

#define EVAL_FULL()                                                                      \
  do {                                                                                   \
    x_cost = 0;                                                                          \
    memset(grad, 0, sizeof(grad));                                                       \
    for (size_t i = 0; i < A; ++i) {                                                     \
      double ci;                                                                         \
      eval_block(c[i], x, a, variant, &ci, &r[i], &J[6 * i]);                            \
      x_cost += ci;                                                                      \
      for (int j = 0; j < 6; ++j) grad[j] += J[6 * i + j] * r[i];                        \
    }                                                                                    \
    if (!have_scale) { /* jacobi_scaling, fixed at iteration 0 */                        \
      for (int j = 0; j < 6; ++j) {                                                      \
        double s2 = 0;                                                                   \
        for (size_t i = 0; i < A; ++i) s2 += J[6 * i + j] * J[6 * i + j];                \
        scale[j] = 1.0 / (1.0 + sqrt(s2));                                               \
      }                                                                                  \
      have_scale = 1;                                                                    \
    }                                                                                    \
    for (size_t i = 0; i < A; ++i) for (int j = 0; j < 6; ++j) J[6 * i + j] *= scale[j]; \
  } while (0)

  EVAL_FULL();
  st->initial_cost = x_cost;
  double x_norm = 0;
  for (int j = 0; j < 7; ++j) x_norm += x[j] * x[j];
  x_norm = sqrt(x_norm);
  /* gradient max-norm as |x - Plus(x, -g)|_inf */
  double gmax;
  {
    double ng[6], xp[7];
    for (int j = 0; j < 6; ++j) ng[j] = -grad[j];
    orc_pose_plus(x, ng, xp);
    gmax = 0;
    for (int j = 0; j < 7; ++j) { double v = fabs(x[j] - xp[j]); if (v > gmax) gmax = v; }
  }
  if (gmax <= g_tol) { st->termination = 3; goto done; }

  for (int iter = 1;; ++iter) {
    /* FinalizeIterationAndCheckIfMinimizerCanContinue for the previous iteration */
    if (iter - 1 >= max_it) { st->termination = 0; break; }
    if (radius <= min_radius) { st->termination = 5; break; }
    st->lm_iterations = iter;
    /* LevenbergMarquardtStrategy::ComputeStep */
    if (!reuse_diagonal) {
      for (int j = 0; j < 6; ++j) {
        double s2 = 0;
        for (size_t i = 0; i < A; ++i) s2 += J[6 * i + j] * J[6 * i + j];
        diag[j] = fmin(fmax(s2, min_diag), max_diag);
      }
    }
    double lm_diag[6];
    for (int j = 0; j < 6; ++j) lm_diag[j] = sqrt(diag[j] / radius);
    /* DenseQRSolver: [J; D] y = [r; 0]; step = -y */
    memcpy(M, J, A * 6 * sizeof(double));
    memcpy(rhs, r, A * sizeof(double));
    for (int j = 0; j < 6; ++j) {
      for (int k = 0; k < 6; ++k) M[6 * (A + j) + k] = (j == k) ? lm_diag[j] : 0.0;
      rhs[A + j] = 0;
    }
    double y[6], step[6], delta[6];
    int ok = qr_solve6(M, rhs, rows, y);
    reuse_diagonal = 1;
    double model_cost_change = 0;
    if (ok) {
      for (int j = 0; j < 6; ++j) step[j] = -y[j];
      /* model_cost_change = -(J s)^T (r + J s / 2) */
      for (size_t i = 0; i < A; ++i) {
        double js = 0;
        for (int j = 0; j < 6; ++j) js += J[6 * i + j] * step[j];
        model_cost_change -= js * (r[i] + js / 2.0);
      }
    }
    if (!ok || !(model_cost_change > 0.0)) { /* HandleInvalidStep */
      if (++invalid_steps >= 5) { st->termination = 5; break; }
      radius *= 0.5; reuse_diagonal = 1;
      continue;
    }
    invalid_steps = 0;
    for (int j = 0; j < 6; ++j) delta[j] = step[j] * scale[j];
    orc_pose_plus(x, delta, cand);
    double cand_cost = 0;
    for (size_t i = 0; i < A; ++i) { double ci; eval_block(c[i], cand, a, variant, &ci, NULL, NULL); cand_cost += ci; }
    /* ParameterToleranceReached */
    double step_norm = 0;
    for (int j = 0; j < 7; ++j) step_norm += (x[j] - cand[j]) * (x[j] - cand[j]);
    step_norm = sqrt(step_norm);
    if (step_norm <= p_tol * (x_norm + p_tol)) { st->termination = 2; break; }
    /* FunctionToleranceReached */
    double cost_change = x_cost - cand_cost;
    if (fabs(cost_change) <= f_tol * x_cost) { st->termination = 1; break; }
    double rel_dec = cost_change / model_cost_change;
    if (rel_dec > min_rel_dec) { /* HandleSuccessfulStep */
      memcpy(x, cand, sizeof(x));
      x_norm = 0;
      for (int j = 0; j < 7; ++j) x_norm += x[j] * x[j];
      x_norm = sqrt(x_norm);
      EVAL_FULL();
      st->num_successful_steps++;
      radius = radius / fmax(1.0 / 3.0, 1.0 - pow(2.0 * rel_dec - 1.0, 3));
      radius = fmin(max_radius, radius);
      decrease_factor = 2.0; reuse_diagonal = 0;
      {
        double ng[6], xp[7];
        for (int j = 0; j < 6; ++j) ng[j] = -grad[j];
        orc_pose_plus(x, ng, xp);
        gmax = 0;
        for (int j = 0; j < 7; ++j) { double v = fabs(x[j] - xp[j]); if (v > gmax) gmax = v; }
      }
      /* FinalizeIterationAndCheckIfMinimizerCanContinue [UPSTREAM ceres 2.0.0 trust_region_minimizer.cc] tests
       * MaxSolverIterationsReached BEFORE GradientToleranceReached: when both fire on the last iteration the summary
       * says NO_CONVERGENCE (0), not CONVERGENCE by gradient (3) */
      if (iter >= max_it) { st->termination = 0; break; }
      if (gmax <= g_tol) { st->termination = 3; break; }
    } else { /* HandleUnsuccessfulStep */
      radius = radius / decrease_factor;
      decrease_factor *= 2.0; reuse_diagonal = 1;
    }
  }
done:
  st->final_cost = x_cost;
  memcpy(pose, x, sizeof(x));
  free(J); free(r); free(M); free(rhs); free(c);
#undef EVAL_FULL
}

/* LS:915-964 */
void orc_uncertainty_from_hist(const int32_t H[ORC_N_OBS], double u[6]) {
  double tt = (double)H[6] + H[7] + H[8];
  double tr = (double)H[0] + H[1] + H[2] + H[3] + H[4] + H[5];
  if (tt == 0 || tr == 0) { for (int i = 0; i < 6; ++i) u[i] = 0; return; }
  u[0] = fmin(H[6] / tt * 3, 1.0); u[1] = fmin(H[7] / tt * 3, 1.0); u[2] = fmin(H[8] / tt * 3, 1.0);
  u[3] = fmin((H[0] + H[1]) / tr * 3, 1.0); u[4] = fmin((H[2] + H[3]) / tr * 3, 1.0); u[5] = fmin((H[4] + H[5]) / tr * 3, 1.0);
}
/* LS:346-359 */
int orc_should_process(size_t index, size_t n_points, int max_surface_features) {
  /* `num_points > OptSet.max_surface_features` compares a size_t with an int (LS:347): a negative setting converts to a
   * huge unsigned value (no sampling), 0 gives rate 0 and drops every point (0 + 0.001 > 0) */
  if (max_surface_features < 0 || n_points <= (size_t)max_surface_features) return 1;
  double rate = 1.0 * max_surface_features / n_points;
  double rem = fmod(index * rate, 1.0);
  return !(rem + 0.001 > rate);
}
/* LS:891-913 with tf2::Matrix3x3::getRPY / tf2::Quaternion::setRPY restated [UPSTREAM tf2] */
void orc_yaw_correction(double pose[7], const double last[7], double yaw_ratio) {
  double tn, rn;
  relative_motion(last, pose, &tn, &rn);
  float translation_norm = (float)tn; /* LS:895 */
  double x = pose[3], y = pose[4], z = pose[5], w = pose[6];
  double d = x * x + y * y + z * z + w * w, s = 2.0 / d;
  double xs = x * s, ys = y * s, zs = z * s, wx = w * xs, wy = w * ys, wz = w * zs;
  double xx = x * xs, xy = x * ys, xz = x * zs, yy = y * ys, yz = y * zs, zz = z * zs;
  double m00 = 1.0 - (yy + zz), m10 = xy + wz, m20 = xz - wy, m21 = yz + wx, m22 = 1.0 - (xx + yy);
  double m01 = xy - wz, m02 = xz + wy;
  double roll, pitch, yaw;
  if (fabs(m20) >= 1) {
    yaw = 0;
    double delta = atan2(m01, m02);
    if (m20 < 0) { pitch = M_PI / 2.0; roll = delta; }
    else { pitch = -M_PI / 2.0; roll = delta; }
  } else {
    pitch = -asin(m20);
    roll = atan2(m21 / cos(pitch), m22 / cos(pitch));
    yaw = atan2(m10 / cos(pitch), m00 / cos(pitch));
  }
  double cyaw = yaw + translation_norm * yaw_ratio * M_PI / 180;
  double hy = cyaw * 0.5, hp = pitch * 0.5, hr = roll * 0.5;
  double cy = cos(hy), sy = sin(hy), cp = cos(hp), sp = sin(hp), cr = cos(hr), sr = sin(hr);
  double q[4] = {sr * cp * cy - cr * sp * sy, cr * sp * cy + sr * cp * sy, cr * cp * sy - sr * sp * cy, cr * cp * cy + sr * sp * sy};
  quat_normalize(q);
  memcpy(pose + 3, q, sizeof(q));
}

/* ============================================================================================ */
/* performLocalizationAndMapping, LS:107-152 (+ post-processing :155-210 minus the map insert)   */
/* ============================================================================================ */
int orc_register(orc_map *m, const float *scan, size_t n, size_t stride, const double pose_in[7], const orc_config *cfg,
                 const int32_t prev_obs_hist[ORC_N_OBS], double pose_out[7], orc_stats *st, orc_corr *last_corrs) {
  if (stride == 0) stride = 3;
  memset(st, 0, sizeof(*st));
  double T[7], T_init[7], T_last[7];
  memcpy(T, pose_in, sizeof(T)); memcpy(T_init, pose_in, sizeof(T)); memcpy(T_last, pose_in, sizeof(T)); /* LS:53-57 */
  memcpy(pose_out, pose_in, sizeof(T));
  if (prev_obs_hist) orc_uncertainty_from_hist(prev_obs_hist, st->uncertainty); /* LS:47 */
  orc_map_shift(m, T, st->pos_in_map);                                          /* LS:363 */
  st->surf_from_map_num = orc_map_count_5x5(m, st->pos_in_map);                 /* LS:367 */
  st->surf_stack_num = (int32_t)n;
  if (!(st->surf_from_map_num > 50)) return 1;                                  /* LS:113-116, 379-381 */
  if (cfg->use_grid_knn == 1) map_ensure_grids(m);
  int max_outer = cfg->max_iterations > 0 ? cfg->max_iterations : 4;
  if (max_outer > ORC_MAX_OUTER) max_outer = ORC_MAX_OUTER;
  orc_corr *corrs = last_corrs ? last_corrs : (orc_corr *)malloc((n ? n : 1) * sizeof(orc_corr));
  for (int it = 0; it < max_outer; ++it) {
    orc_iter_stats *is = &st->iters[it];
    st->n_iterations = it + 1;
    /* processPlannerFeatures, LS:323-344 (serial in the reference; queries are independent) */
#pragma omp parallel for schedule(dynamic, 256) if (cfg->use_grid_knn != 2) /* Oracle-B: the hook builds its trees lazily -- serial, like the reference loop (LS:328) */
    for (size_t i = 0; i < n; ++i) {
      if (!orc_should_process(i, n, cfg->max_surface_features)) { memset(&corrs[i], 0, sizeof(orc_corr)); corrs[i].status = -1; continue; }
      orc_plane_match(m, T, scan + i * stride, cfg, &corrs[i]);
    }
    for (size_t i = 0; i < n; ++i) {
      if (corrs[i].status < 0) continue;
      if (corrs[i].status == ORC_SUCCESS) { is->obs_hist[corrs[i].obs[0]]++; is->obs_hist[corrs[i].obs[1]]++; is->obs_hist[corrs[i].obs[2]]++; }
      is->reject_hist[corrs[i].status]++;
    }
    double prev[7];
    memcpy(prev, T, sizeof(T));
    orc_lm_solve(corrs, n, T, m->planeRes, cfg, is); /* LS:130-136 */
    relative_motion(prev, T, &is->translation_norm, &is->rotation_norm); /* LS:246-249 */
    memcpy(is->pose_after, T, sizeof(T));
    if (is->num_successful_steps == 1 || it == max_outer - 1) break; /* LS:141 */
  }
  /* final normal equations at the returned pose (for a16-style covariance on the caller side) */
  {
    double cost; int cnt;
    orc_evaluate(corrs, n, T, m->planeRes, cfg->tukey_variant, &cost, st->JtJ, st->Jtr, &cnt);
  }
  orc_yaw_correction(T, T_last, cfg->yaw_ratio); /* LS:157 */
  relative_motion(T_init, T, &st->total_translation, &st->total_rotation);        /* LS:201-204 */
  relative_motion(T_last, T, &st->translation_from_last, &st->rotation_from_last); /* LS:205-208 */
  memcpy(pose_out, T, sizeof(T));
  if (!last_corrs) free(corrs);
  return 0;
}

int orc_transform_and_add(orc_map *m, const float *scan, size_t n, size_t stride, const double pose[7]) { /* LS:60-80, sutil:119-123 */
  if (stride == 0) stride = 3;
  float *w = (float *)malloc((n ? n : 1) * 12);
  for (size_t i = 0; i < n; ++i) {
    double p[3] = {(double)scan[i * stride], (double)scan[i * stride + 1], (double)scan[i * stride + 2]}, o[3];
    quat_rotate(pose + 3, p, o);
    w[3 * i] = (float)(o[0] + pose[0]); w[3 * i + 1] = (float)(o[1] + pose[1]); w[3 * i + 2] = (float)(o[2] + pose[2]);
  }
  int r = orc_map_add_surf(m, w, n, 3);
  free(w);
  return r;
}

void orc_set_num_threads(int n) {
#ifdef _OPENMP
  omp_set_num_threads(n > 0 ? n : 1);
#else
  (void)n;
#endif
}

int orc_num_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

/* ============================================================================================
 * featureExtraction::removePointDistortion (src/FeatureExtraction/featureExtraction.cpp:223-314), the de-skew in front of
 * the feature extraction whose planar cloud this path registers.  Transformd algebra: include/super_odometry/utils/Twist.h
 * (:165-172 inverse, :180-185 product through Eigen::Transform, :187 point).  Eigen 3.4 routines written out:
 * QuaternionBase::toRotationMatrix, quaternionbase_assign_impl<Matrix3>, normalized(), slerp().  [UPSTREAM Eigen]
 * ============================================================================================ */
typedef struct { double q[4]; double p[3]; } dsk_tf; /* rot x y z w, pos */

static void dsk_unit(const double q[4], double o[4]) {
  double n2 = q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3];
  if (n2 > 0) { double n = sqrt(n2); for (int i = 0; i < 4; ++i) o[i] = q[i] / n; }
  else for (int i = 0; i < 4; ++i) o[i] = q[i];
}
static void dsk_rotmat(const double q[4], double R[3][3]) {
  const double x = q[0], y = q[1], z = q[2], w = q[3];
  const double tx = 2 * x, ty = 2 * y, tz = 2 * z;
  const double twx = tx * w, twy = ty * w, twz = tz * w, txx = tx * x, txy = ty * x, txz = tz * x, tyy = ty * y, tyz = tz * y, tzz = tz * z;
  R[0][0] = 1 - (tyy + tzz); R[0][1] = txy - twz; R[0][2] = txz + twy;
  R[1][0] = txy + twz; R[1][1] = 1 - (txx + tzz); R[1][2] = tyz - twx;
  R[2][0] = txz - twy; R[2][1] = tyz + twx; R[2][2] = 1 - (txx + tyy);
}
static void dsk_quat_of(const double R[3][3], double q[4]) {
  double t = R[0][0] + R[1][1] + R[2][2];
  if (t > 0) {
    t = sqrt(t + 1.0);
    q[3] = 0.5 * t;
    t = 0.5 / t;
    q[0] = (R[2][1] - R[1][2]) * t; q[1] = (R[0][2] - R[2][0]) * t; q[2] = (R[1][0] - R[0][1]) * t;
  } else {
    int i = 0;
    if (R[1][1] > R[0][0]) i = 1;
    if (R[2][2] > R[i][i]) i = 2;
    int j = (i + 1) % 3, k = (j + 1) % 3;
    t = sqrt(R[i][i] - R[j][j] - R[k][k] + 1.0);
    q[i] = 0.5 * t;
    t = 0.5 / t;
    q[3] = (R[k][j] - R[j][k]) * t; q[j] = (R[j][i] + R[i][j]) * t; q[k] = (R[k][i] + R[i][k]) * t;
  }
}
static dsk_tf dsk_mul(const dsk_tf *a, const dsk_tf *b) { /* Twist::operator* */
  double ua[4], ub[4], A[3][3], B[3][3], Cm[3][3], q[4];
  dsk_unit(a->q, ua); dsk_unit(b->q, ub);
  dsk_rotmat(ua, A); dsk_rotmat(ub, B);
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) Cm[i][j] = A[i][0] * B[0][j] + A[i][1] * B[1][j] + A[i][2] * B[2][j];
  dsk_tf o;
  dsk_quat_of(Cm, q);
  dsk_unit(q, o.q);
  for (int i = 0; i < 3; ++i) o.p[i] = A[i][0] * b->p[0] + A[i][1] * b->p[1] + A[i][2] * b->p[2] + a->p[i];
  return o;
}
static dsk_tf dsk_inv(const dsk_tf *a) { /* Twist::inverse */
  dsk_tf o;
  double R[3][3];
  o.q[0] = -a->q[0]; o.q[1] = -a->q[1]; o.q[2] = -a->q[2]; o.q[3] = a->q[3];
  dsk_rotmat(o.q, R);
  for (int i = 0; i < 3; ++i) o.p[i] = -(R[i][0] * a->p[0] + R[i][1] * a->p[1] + R[i][2] * a->p[2]);
  return o;
}
static void dsk_slerp(const double a[4], const double b[4], double t, double o[4]) { /* QuaternionBase::slerp */
  const double one = 1.0 - DBL_EPSILON;
  double d = a[0] * b[0] + a[1] * b[1] + a[2] * b[2] + a[3] * b[3], ad = fabs(d), s0, s1;
  if (ad >= one) { s0 = 1.0 - t; s1 = t; }
  else { double th = acos(ad), sn = sin(th); s0 = sin((1.0 - t) * th) / sn; s1 = sin(t * th) / sn; }
  if (d < 0) s1 = -s1;
  for (int i = 0; i < 4; ++i) o[i] = s0 * a[i] + s1 * b[i];
}
/* getInterpolatedPoseAtTime, :257-276 */
static dsk_tf dsk_pose_at(const double *poses, size_t n, int imu, double ts, int *beyond) {
  size_t after = 0;
  while (after < n && !(poses[8 * after] > ts)) ++after; /* std::map::upper_bound */
  dsk_tf r;
  if (after == n || after == 0) {
    size_t e = after == n ? n - 1 : 0;
    if (after == n) *beyond = 1;
    for (int i = 0; i < 3; ++i) r.p[i] = imu ? 0.0 : poses[8 * e + 1 + i];
    for (int i = 0; i < 4; ++i) r.q[i] = poses[8 * e + 4 + i];
    return r;
  }
  const double *pb = poses + 8 * (after - 1), *pa = poses + 8 * after;
  double ratio = (ts - pb[0]) / (pa[0] - pb[0]);
  dsk_slerp(pb + 4, pa + 4, ratio, r.q);
  for (int i = 0; i < 3; ++i) r.p[i] = imu ? (1 - ratio) * 0.0 + ratio * 0.0 : (1 - ratio) * pb[1 + i] + ratio * pa[1 + i];
  return r;
}

size_t orc_deskew(void *points, size_t n, size_t stride, size_t time_off, double t0, const double *poses, size_t n_poses, int imu,
                  const double T_i_l[7], double start_sensor[7]) {
  dsk_tf il, li;
  if (T_i_l) { for (int i = 0; i < 3; ++i) il.p[i] = T_i_l[i]; for (int i = 0; i < 4; ++i) il.q[i] = T_i_l[3 + i]; }
  else { il.p[0] = il.p[1] = il.p[2] = 0; il.q[0] = il.q[1] = il.q[2] = 0; il.q[3] = 1; }
  li = dsk_inv(&il);
  int beyond = 0;
  dsk_tf w_original = dsk_pose_at(poses, n_poses, imu, t0, &beyond);              /* :279-282 */
  dsk_tf sensor = imu ? dsk_mul(&w_original, &il) : w_original;                    /* :283-290 */
  if (start_sensor) { for (int i = 0; i < 3; ++i) start_sensor[i] = sensor.p[i]; for (int i = 0; i < 4; ++i) start_sensor[3 + i] = sensor.q[i]; }
  dsk_tf w_original_inv = dsk_inv(&w_original);
  size_t n_beyond = 0;
  for (size_t k = 0; k < n; ++k) {                                                 /* :292-312 */
    float *xyz = (float *)((char *)points + k * stride);
    if (!isfinite(xyz[0]) || !isfinite(xyz[1]) || !isfinite(xyz[2])) continue;
    float tm;
    memcpy(&tm, (char *)points + k * stride + time_off, 4);
    double ts = tm + t0;
    beyond = 0;
    dsk_tf w_current = dsk_pose_at(poses, n_poses, imu, ts, &beyond);
    n_beyond += (size_t)beyond;
    dsk_tf oc = dsk_mul(&w_original_inv, &w_current), fin = oc;
    if (imu) { dsk_tf tmp = dsk_mul(&li, &oc); fin = dsk_mul(&tmp, &il); }
    /* T_final * pt: rot * p + pos, Eigen's _transformVector */
    double v[3] = {xyz[0], xyz[1], xyz[2]}, *u = fin.q;
    double uv[3] = {2 * (u[1] * v[2] - u[2] * v[1]), 2 * (u[2] * v[0] - u[0] * v[2]), 2 * (u[0] * v[1] - u[1] * v[0])};
    double r[3] = {v[0] + u[3] * uv[0] + (u[1] * uv[2] - u[2] * uv[1]), v[1] + u[3] * uv[1] + (u[2] * uv[0] - u[0] * uv[2]),
                   v[2] + u[3] * uv[2] + (u[0] * uv[1] - u[1] * uv[0])};
    xyz[0] = (float)(r[0] + fin.p[0]); xyz[1] = (float)(r[1] + fin.p[1]); xyz[2] = (float)(r[2] + fin.p[2]);
  }
  return n_beyond;
}
