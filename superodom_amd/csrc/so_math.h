// so_math.h -- pose / quaternion arithmetic shared by the host driver and the HIP kernels.
// Every routine names the reference arithmetic it reproduces (paths relative to
// /root/reference/super_odometry/).  Pose layout = pose_parameters[7] (src/LidarProcess/LidarSlam.cpp:7-9):
// {tx,ty,tz,qx,qy,qz,qw}.
#pragma once
#include <math.h>
#include <stdint.h>

#if defined(__HIP__)
#include <hip/hip_runtime.h>
#define SO_HD __host__ __device__ __forceinline__
#else
#define SO_HD inline
#endif

namespace soicp {

struct Pose {
  double t[3];
  double q[4];  // x y z w
};

SO_HD Pose pose_from_array(const double p[7]) {
  Pose o;
  o.t[0] = p[0]; o.t[1] = p[1]; o.t[2] = p[2];
  o.q[0] = p[3]; o.q[1] = p[4]; o.q[2] = p[5]; o.q[3] = p[6];
  return o;
}
SO_HD void pose_to_array(const Pose& p, double o[7]) {
  o[0] = p.t[0]; o[1] = p.t[1]; o[2] = p.t[2];
  o[3] = p.q[0]; o[4] = p.q[1]; o[5] = p.q[2]; o[6] = p.q[3];
}

// Eigen::QuaternionBase::_transformVector (what `T_w_lidar * pos`, utils/Twist.h:187, and
// `q_w_curr * curr_point`, LaserMapping/lidarOptimization.cpp:59, execute):
//   uv = 2 * (u x v);  result = v + w * uv + u x uv
template <typename T>
SO_HD void quat_rotate(const T q[4], T vx, T vy, T vz, T& ox, T& oy, T& oz) {
  T ux = q[1] * vz - q[2] * vy, uy = q[2] * vx - q[0] * vz, uz = q[0] * vy - q[1] * vx;
  ux += ux; uy += uy; uz += uz;
  ox = vx + q[3] * ux + (q[1] * uz - q[2] * uy);
  oy = vy + q[3] * uy + (q[2] * ux - q[0] * uz);
  oz = vz + q[3] * uz + (q[0] * uy - q[1] * ux);
}

// a * b + c with one rounding.  Used where the restated arithmetic has no thresholds downstream (the LM controller's
// linear algebra and the normal-equation sums): half the serial fp64 operations of the unfused form; the plane fit and
// its gates keep the reference's unfused arithmetic (-ffp-contract=off).
#define SO_FMA(a, b, c) __builtin_fma((a), (b), (c))

SO_HD void quat_mul(const double a[4], const double b[4], double o[4]) {
  o[3] = a[3] * b[3] - a[0] * b[0] - a[1] * b[1] - a[2] * b[2];
  o[0] = a[3] * b[0] + a[0] * b[3] + a[1] * b[2] - a[2] * b[1];
  o[1] = a[3] * b[1] + a[1] * b[3] + a[2] * b[0] - a[0] * b[2];
  o[2] = a[3] * b[2] + a[2] * b[3] + a[0] * b[1] - a[1] * b[0];
}

// PoseLocalParameterization::Plus, src/LidarProcess/pose_local_parameterization.cpp:7-23 with
// Utility::deltaQ (include/super_odometry/utils/utility.h:12-24): p += dp; q = normalize(q (x) [1, dtheta/2]).
SO_HD void pose_plus(const double x[7], const double d[6], double o[7]) {
  o[0] = x[0] + d[0]; o[1] = x[1] + d[1]; o[2] = x[2] + d[2];
  const double dq[4] = {d[3] / 2.0, d[4] / 2.0, d[5] / 2.0, 1.0};
  double q[4];  // x.q (x) [dq, 1]
  const double* a = x + 3;
  q[3] = SO_FMA(-a[2], dq[2], SO_FMA(-a[1], dq[1], SO_FMA(-a[0], dq[0], a[3])));
  q[0] = SO_FMA(-a[2], dq[1], SO_FMA(a[1], dq[2], SO_FMA(a[3], dq[0], a[0])));
  q[1] = SO_FMA(-a[0], dq[2], SO_FMA(a[2], dq[0], SO_FMA(a[3], dq[1], a[1])));
  q[2] = SO_FMA(-a[1], dq[0], SO_FMA(a[0], dq[1], SO_FMA(a[3], dq[2], a[2])));
  // Eigen normalized() divides coefficient-wise; one reciprocal + four products differ from that by <= 1 ulp per
  // coefficient and cost a quarter of the serial fp64 latency inside the device-side controller
#if defined(__HIP_DEVICE_COMPILE__)
  const double inv = rsqrt(SO_FMA(q[3], q[3], SO_FMA(q[2], q[2], SO_FMA(q[1], q[1], q[0] * q[0]))));  // one long fp64 operation instead of sqrt + division
#else
  const double inv = 1.0 / sqrt(SO_FMA(q[3], q[3], SO_FMA(q[2], q[2], SO_FMA(q[1], q[1], q[0] * q[0]))));
#endif
  o[3] = q[0] * inv; o[4] = q[1] * inv; o[5] = q[2] * inv; o[6] = q[3] * inv;
}

// a o d: the pose a followed by the relative motion d expressed in a's frame -- what `T_w_lidar = T_w_lidar * prediction`
// (laserMapping.cpp:345-372, Twist::operator*, utils/Twist.h:181-185) computes, here without Eigen's detour through the 4x4 affine
// matrix: t = a.t + R(a.q) d.t, q = normalize(a.q (x) d.q).  Only + - * / sqrt, unfused (-ffp-contract=off on both sides): the host and the
// device form the same bits (so_icp_register_sequence composes the guess of a chained registration on the device).
SO_HD void pose_compose(const double a[7], const double d[7], double o[7]) {
  double rx, ry, rz;
  quat_rotate<double>(a + 3, d[0], d[1], d[2], rx, ry, rz);
  o[0] = a[0] + rx; o[1] = a[1] + ry; o[2] = a[2] + rz;
  double q[4];
  quat_mul(a + 3, d + 3, q);
  const double nrm = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  o[3] = q[0] / nrm; o[4] = q[1] / nrm; o[5] = q[2] / nrm; o[6] = q[3] / nrm;
}

// (a^-1 * b).pos.norm() and 2*atan2(|vec|, w): LidarSlam.cpp:201-208, 246-249 (Twist.h:172-185).
SO_HD void relative_motion(const double a[7], const double b[7], double& tn, double& rn) {
  const double qi[4] = {-a[3], -a[4], -a[5], a[6]};
  double tx, ty, tz, dq[4];
  quat_rotate<double>(qi, b[0] - a[0], b[1] - a[1], b[2] - a[2], tx, ty, tz);
  quat_mul(qi, b + 3, dq);
  if (dq[3] < 0) { dq[0] = -dq[0]; dq[1] = -dq[1]; dq[2] = -dq[2]; dq[3] = -dq[3]; }
  tn = sqrt(tx * tx + ty * ty + tz * tz);
  rn = 2.0 * atan2(sqrt(dq[0] * dq[0] + dq[1] * dq[1] + dq[2] * dq[2]), dq[3]);
}

// LocalMap cube coordinate, LocalMap.h:488-497: int((c + 25.0)/50.0) + origin, then -- if c + 25.0 < 0.
SO_HD int cube_coord(double c, int origin) {
  int i = (int)((c + 25.0) / 50.0) + origin;
  if (c + 25.0 < 0) i--;
  return i;
}

// 6-bit-per-axis Morton interleave (sort key locality inside one 50 m cube).
SO_HD uint32_t part1by2(uint32_t x) {
  x &= 0x3ff;
  x = (x ^ (x << 16)) & 0xff0000ff;
  x = (x ^ (x << 8)) & 0x0300f00f;
  x = (x ^ (x << 4)) & 0x030c30c3;
  x = (x ^ (x << 2)) & 0x09249249;
  return x;
}
SO_HD uint32_t morton3(uint32_t x, uint32_t y, uint32_t z) { return part1by2(x) | (part1by2(y) << 1) | (part1by2(z) << 2); }

// Shard ownership granule: a brick of kBrickCells^3 cells.  Hash over WORLD cube ids + brick coordinates,
// stable under LocalMap::shiftMap.
#ifndef SO_BRICK_CELLS
#define SO_BRICK_CELLS 4
#endif
constexpr int kBrickCells = SO_BRICK_CELLS;
SO_HD uint32_t brick_hash(int wx, int wy, int wz, int bx, int by, int bz) {
  uint32_t h = 2166136261u;
  const uint32_t v[6] = {(uint32_t)wx, (uint32_t)wy, (uint32_t)wz, (uint32_t)bx, (uint32_t)by, (uint32_t)bz};
  for (int i = 0; i < 6; ++i) { h ^= v[i] + 0x9e3779b9u + (h << 6) + (h >> 2); h *= 16777619u; }
  h ^= h >> 15; h *= 0x2c1b3c6du; h ^= h >> 12;
  return h;
}

}  // namespace soicp
