// deskew_math.h -- arithmetic of featureExtraction::removePointDistortion
// (/root/reference/super_odometry/src/FeatureExtraction/featureExtraction.cpp:223-314), shared by the host entry point and
// the HIP kernel.  Transformd products follow include/super_odometry/utils/Twist.h:165-187 through Eigen 3.4's
// quaternion <-> matrix conversions [UPSTREAM Eigen, written out]; slerp is Eigen::QuaternionBase::slerp.
#pragma once
#include "so_math.h"

namespace soicp {

struct Rigid {     // Transformd: rot (x y z w), pos
  double q[4];
  double t[3];
};

SO_HD void quat_normalized(const double q[4], double o[4]) {  // Eigen 3.4 normalized(): unchanged when the norm is 0
  const double n2 = q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3];
  if (n2 > 0) { const double n = sqrt(n2); o[0] = q[0] / n; o[1] = q[1] / n; o[2] = q[2] / n; o[3] = q[3] / n; }
  else { o[0] = q[0]; o[1] = q[1]; o[2] = q[2]; o[3] = q[3]; }
}
SO_HD void quat_to_matrix(const double q[4], double m[9]) {  // QuaternionBase::toRotationMatrix, row-major
  const double tx = 2 * q[0], ty = 2 * q[1], tz = 2 * q[2];
  const double twx = tx * q[3], twy = ty * q[3], twz = tz * q[3];
  const double txx = tx * q[0], txy = ty * q[0], txz = tz * q[0], tyy = ty * q[1], tyz = tz * q[1], tzz = tz * q[2];
  m[0] = 1 - (tyy + tzz); m[1] = txy - twz; m[2] = txz + twy;
  m[3] = txy + twz; m[4] = 1 - (txx + tzz); m[5] = tyz - twx;
  m[6] = txz - twy; m[7] = tyz + twx; m[8] = 1 - (txx + tyy);
}
SO_HD void matrix_to_quat(const double a[9], double q[4]) {  // quaternionbase_assign_impl<Matrix3>::run; branches written without indexed registers
  double t = a[0] + a[4] + a[8];
  if (t > 0) {
    t = sqrt(t + 1.0);
    q[3] = 0.5 * t;
    t = 0.5 / t;
    q[0] = (a[7] - a[5]) * t; q[1] = (a[2] - a[6]) * t; q[2] = (a[3] - a[1]) * t;
  } else if (!(a[4] > a[0]) && !(a[8] > a[0])) {  // i = 0, j = 1, k = 2
    t = sqrt(a[0] - a[4] - a[8] + 1.0);
    q[0] = 0.5 * t; t = 0.5 / t;
    q[3] = (a[7] - a[5]) * t; q[1] = (a[3] + a[1]) * t; q[2] = (a[6] + a[2]) * t;
  } else if (a[4] > a[0] && !(a[8] > a[4])) {     // i = 1, j = 2, k = 0
    t = sqrt(a[4] - a[8] - a[0] + 1.0);
    q[1] = 0.5 * t; t = 0.5 / t;
    q[3] = (a[2] - a[6]) * t; q[2] = (a[7] + a[5]) * t; q[0] = (a[1] + a[3]) * t;
  } else {                                         // i = 2, j = 0, k = 1
    t = sqrt(a[8] - a[0] - a[4] + 1.0);
    q[2] = 0.5 * t; t = 0.5 / t;
    q[3] = (a[3] - a[1]) * t; q[0] = (a[2] + a[6]) * t; q[1] = (a[5] + a[7]) * t;
  }
}
// Twist::operator* (Twist.h:180-185)
SO_HD Rigid rigid_mul(const Rigid& a, const Rigid& b) {
  double qa[4], qb[4], ra[9], rb[9], rc[9];
  quat_normalized(a.q, qa); quat_normalized(b.q, qb);
  quat_to_matrix(qa, ra); quat_to_matrix(qb, rb);
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) rc[3 * i + j] = ra[3 * i] * rb[j] + ra[3 * i + 1] * rb[3 + j] + ra[3 * i + 2] * rb[6 + j];
  Rigid o;
  double q[4];
  matrix_to_quat(rc, q);
  quat_normalized(q, o.q);
  for (int i = 0; i < 3; ++i) o.t[i] = ra[3 * i] * b.t[0] + ra[3 * i + 1] * b.t[1] + ra[3 * i + 2] * b.t[2] + a.t[i];
  return o;
}
// Twist::inverse (Twist.h:165-172)
SO_HD Rigid rigid_inverse(const Rigid& a) {
  Rigid o;
  o.q[0] = -a.q[0]; o.q[1] = -a.q[1]; o.q[2] = -a.q[2]; o.q[3] = a.q[3];
  double r[9];
  quat_to_matrix(o.q, r);
  for (int i = 0; i < 3; ++i) o.t[i] = -(r[3 * i] * a.t[0] + r[3 * i + 1] * a.t[1] + r[3 * i + 2] * a.t[2]);
  return o;
}
// Eigen::QuaternionBase::slerp
SO_HD void quat_slerp(const double a[4], const double b[4], double t, double o[4]) {
  const double one = 1.0 - 2.220446049250313e-16;
  const double d = a[0] * b[0] + a[1] * b[1] + a[2] * b[2] + a[3] * b[3];
  const double ad = fabs(d);
  double s0, s1;
  if (ad >= one) { s0 = 1.0 - t; s1 = t; }
  else {
    const double theta = acos(ad), st = sin(theta);
    s0 = sin((1.0 - t) * theta) / st;
    s1 = sin(t * theta) / st;
  }
  if (d < 0) s1 = -s1;
  for (int k = 0; k < 4; ++k) o[k] = s0 * a[k] + s1 * b[k];
}

// one stamped pose of the buffer the scan is de-skewed against: 8 doubles {time, px, py, pz, qx, qy, qz, qw}
constexpr int kStampedPoseDoubles = 8;

// getInterpolatedPoseAtTime (featureExtraction.cpp:257-276): the first entry with time > ts is "after"; before the first
// entry its pose is returned; otherwise slerp / lerp between the neighbours.  ts at or beyond the last entry has no
// "after" (the reference dereferences end() there -- it only runs de-skew once a later measurement has arrived,
// :185-201); this restatement returns the last pose and reports the point (*clamped).
template <typename Table>
SO_HD Rigid interpolated_pose(const Table& tab, uint32_t n, double ts, bool* clamped) {
  uint32_t lo = 0, hi = n;  // upper_bound
  while (lo < hi) {
    const uint32_t mid = (lo + hi) >> 1;
    if (tab[mid * kStampedPoseDoubles] > ts) hi = mid; else lo = mid + 1;
  }
  Rigid r;
  uint32_t at = lo;
  if (at == n) { at = n - 1; *clamped = true; }
  if (at == 0 || lo == n) {
    for (int k = 0; k < 3; ++k) r.t[k] = tab[at * kStampedPoseDoubles + 1 + k];
    for (int k = 0; k < 4; ++k) r.q[k] = tab[at * kStampedPoseDoubles + 4 + k];
    return r;
  }
  const uint32_t b = at - 1;
  const double tb = tab[b * kStampedPoseDoubles], ta = tab[at * kStampedPoseDoubles];
  const double ratio = (ts - tb) / (ta - tb);
  double qb[4], qa[4];
  for (int k = 0; k < 4; ++k) { qb[k] = tab[b * kStampedPoseDoubles + 4 + k]; qa[k] = tab[at * kStampedPoseDoubles + 4 + k]; }
  quat_slerp(qb, qa, ratio, r.q);
  for (int k = 0; k < 3; ++k) r.t[k] = (1 - ratio) * tab[b * kStampedPoseDoubles + 1 + k] + ratio * tab[at * kStampedPoseDoubles + 1 + k];
  return r;
}

struct DeskewFrames {  // per-scan constants (featureExtraction.cpp:279-290)
  Rigid w_original_inv;  // T_w_original.inverse()
  Rigid i_l, l_i;        // T_i_l, T_l_i (src/parameter/parameter.cpp:192-193); used when the pose buffer is the IMU's
  int imu;
};

// the transform applied to a point measured at ts (featureExtraction.cpp:297-306)
template <typename Table>
SO_HD Rigid deskew_transform(const Table& tab, uint32_t n, double ts, const DeskewFrames& f, bool* clamped) {
  const Rigid w_current = interpolated_pose(tab, n, ts, clamped);
  const Rigid original_current = rigid_mul(f.w_original_inv, w_current);
  if (!f.imu) return original_current;
  return rigid_mul(rigid_mul(f.l_i, original_current), f.i_l);
}

}  // namespace soicp
