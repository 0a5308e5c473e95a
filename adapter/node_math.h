// node_math.h -- the pose arithmetic laserMapping.cpp performs through Eigen 3.4 / tf2 on its own (outside
// LidarSLAM::Localization), written out so that the node shim (laser_mapping_soicp.cpp) compiles without either library
// and produces the same doubles: setInitialGuess (laserMapping.cpp:265-381), updatePoseAndPublish (:726-765),
// utils::extractRollPitch (src/utils/superodom_utils.cpp:187-195), Transformd products (include/super_odometry/utils/Twist.h:165-185).
// Each routine names the library routine it restates.  [UPSTREAM Eigen 3.4.0 / tf2 (Humble)]: not in /root/reference,
// parity unpinned upstream -- tests/test_gpu_node.py checks these against an independent numpy restatement.
#pragma once
#include <cmath>

#include "standins.h"

namespace so_node_math {
using so_standins::Quaterniond;
using so_standins::Transformd;
using so_standins::Vector3d;

struct Mat3 { double m[3][3]; };

// Eigen::QuaternionBase::operator* (quat product), conjugate, inverse, normalize / normalized, squaredNorm
inline Quaterniond qmul(const Quaterniond& a, const Quaterniond& b) {
  return Quaterniond(a.w() * b.w() - a.x() * b.x() - a.y() * b.y() - a.z() * b.z(),
                     a.w() * b.x() + a.x() * b.w() + a.y() * b.z() - a.z() * b.y(),
                     a.w() * b.y() + a.y() * b.w() + a.z() * b.x() - a.x() * b.z(),
                     a.w() * b.z() + a.z() * b.w() + a.x() * b.y() - a.y() * b.x());
}
inline Quaterniond qconj(const Quaterniond& q) { return Quaterniond(q.w(), -q.x(), -q.y(), -q.z()); }
inline double qnorm2(const Quaterniond& q) { return q.x() * q.x() + q.y() * q.y() + q.z() * q.z() + q.w() * q.w(); }
inline Quaterniond qinverse(const Quaterniond& q) {  // Eigen: conjugate / squaredNorm, zero quaternion if the norm is 0
  const double n2 = qnorm2(q);
  if (n2 > 0) return Quaterniond(q.w() / n2, -q.x() / n2, -q.y() / n2, -q.z() / n2);
  return Quaterniond(0, 0, 0, 0);
}
inline Quaterniond qnormalized(const Quaterniond& q) {  // Eigen 3.4 MatrixBase::normalized: unchanged if the norm is 0
  const double n2 = qnorm2(q);
  if (n2 > 0) { const double n = std::sqrt(n2); return Quaterniond(q.w() / n, q.x() / n, q.y() / n, q.z() / n); }
  return q;
}
// Eigen::QuaternionBase::_transformVector: uv = 2 (u x v); v + w uv + u x uv
inline Vector3d qrot(const Quaterniond& q, const Vector3d& v) {
  double ux = q.y() * v.z() - q.z() * v.y(), uy = q.z() * v.x() - q.x() * v.z(), uz = q.x() * v.y() - q.y() * v.x();
  ux += ux; uy += uy; uz += uz;
  return Vector3d(v.x() + q.w() * ux + (q.y() * uz - q.z() * uy), v.y() + q.w() * uy + (q.z() * ux - q.x() * uz),
                  v.z() + q.w() * uz + (q.x() * uy - q.y() * ux));
}
// Eigen::QuaternionBase::toRotationMatrix
inline Mat3 to_matrix(const Quaterniond& q) {
  const double tx = 2 * q.x(), ty = 2 * q.y(), tz = 2 * q.z();
  const double twx = tx * q.w(), twy = ty * q.w(), twz = tz * q.w();
  const double txx = tx * q.x(), txy = ty * q.x(), txz = tz * q.x(), tyy = ty * q.y(), tyz = tz * q.y(), tzz = tz * q.z();
  Mat3 r;
  r.m[0][0] = 1 - (tyy + tzz); r.m[0][1] = txy - twz; r.m[0][2] = txz + twy;
  r.m[1][0] = txy + twz; r.m[1][1] = 1 - (txx + tzz); r.m[1][2] = tyz - twx;
  r.m[2][0] = txz - twy; r.m[2][1] = tyz + twx; r.m[2][2] = 1 - (txx + tyy);
  return r;
}
// Eigen::internal::quaternionbase_assign_impl<Matrix3>::run (Ken Shoemake's method as Eigen writes it)
inline Quaterniond from_matrix(const Mat3& a) {
  double q[4];  // x y z w
  double t = a.m[0][0] + a.m[1][1] + a.m[2][2];
  if (t > 0) {
    t = std::sqrt(t + 1.0);
    q[3] = 0.5 * t;
    t = 0.5 / t;
    q[0] = (a.m[2][1] - a.m[1][2]) * t; q[1] = (a.m[0][2] - a.m[2][0]) * t; q[2] = (a.m[1][0] - a.m[0][1]) * t;
  } else {
    int i = 0;
    if (a.m[1][1] > a.m[0][0]) i = 1;
    if (a.m[2][2] > a.m[i][i]) i = 2;
    const int j = (i + 1) % 3, k = (j + 1) % 3;
    t = std::sqrt(a.m[i][i] - a.m[j][j] - a.m[k][k] + 1.0);
    q[i] = 0.5 * t;
    t = 0.5 / t;
    q[3] = (a.m[k][j] - a.m[j][k]) * t; q[j] = (a.m[j][i] + a.m[i][j]) * t; q[k] = (a.m[k][i] + a.m[i][k]) * t;
  }
  return Quaterniond(q[3], q[0], q[1], q[2]);
}
inline Mat3 mmul(const Mat3& a, const Mat3& b) {
  Mat3 r;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) r.m[i][j] = a.m[i][0] * b.m[0][j] + a.m[i][1] * b.m[1][j] + a.m[i][2] * b.m[2][j];
  return r;
}
inline Vector3d mvec(const Mat3& a, const Vector3d& v) {
  return Vector3d(a.m[0][0] * v.x() + a.m[0][1] * v.y() + a.m[0][2] * v.z(), a.m[1][0] * v.x() + a.m[1][1] * v.y() + a.m[1][2] * v.z(),
                  a.m[2][0] * v.x() + a.m[2][1] * v.y() + a.m[2][2] * v.z());
}
// Twist::operator* (Twist.h:180-185): both sides through transform() (rot.normalized().toRotationMatrix(), :80-85), the
// affine product, back through Twist(Eigen::Transform) (:75-78: Quaternion{linear}.normalized()).
// (Sum order inside the 3x3 products is Eigen's internal one; a different association moves the last bit only.)
inline Transformd tmul(const Transformd& a, const Transformd& b) {
  const Mat3 ra = to_matrix(qnormalized(a.rot)), rb = to_matrix(qnormalized(b.rot));
  Transformd o;
  o.rot = qnormalized(from_matrix(mmul(ra, rb)));
  const Vector3d t = mvec(ra, b.pos);
  o.pos = Vector3d(t.x() + a.pos.x(), t.y() + a.pos.y(), t.z() + a.pos.z());
  return o;
}
// Twist::inverse (Twist.h:165-172): rot.conjugate(); pos = -R(conj) * pos
inline Transformd tinverse(const Transformd& a) {
  Transformd o;
  o.rot = qconj(a.rot);
  const Vector3d t = mvec(to_matrix(o.rot), a.pos);
  o.pos = Vector3d(-t.x(), -t.y(), -t.z());
  return o;
}
// Eigen::AngleAxis<double>::operator=(const QuaternionBase&): angle * axis
inline Vector3d angle_axis_vector(const Quaterniond& q) {
  double n = std::sqrt(q.x() * q.x() + q.y() * q.y() + q.z() * q.z());
  // (Eigen falls back to stableNorm() below the smallest normal double; the plain norm is the same number there for our inputs)
  if (n != 0.0) {
    const double angle = 2.0 * std::atan2(n, std::fabs(q.w()));
    if (q.w() < 0) n = -n;
    return Vector3d(q.x() / n * angle, q.y() / n * angle, q.z() / n * angle);
  }
  return Vector3d(0, 0, 0);  // angle 0, axis (1,0,0)
}
// tf2::Matrix3x3(tf2::Quaternion).getRPY (solution 1) and tf2::Quaternion::setRPY
inline void get_rpy(const Quaterniond& q, double& roll, double& pitch, double& yaw) {
  const double x = q.x(), y = q.y(), z = q.z(), w = q.w();
  const double d = x * x + y * y + z * z + w * w, s = 2.0 / d;
  const double xs = x * s, ys = y * s, zs = z * s, wx = w * xs, wy = w * ys, wz = w * zs;
  const double xx = x * xs, xy = x * ys, xz = x * zs, yy = y * ys, yz = y * zs, zz = z * zs;
  const double m00 = 1.0 - (yy + zz), m01 = xy - wz, m02 = xz + wy, m10 = xy + wz, m20 = xz - wy, m21 = yz + wx, m22 = 1.0 - (xx + yy);
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
}
inline Quaterniond set_rpy(double roll, double pitch, double yaw) {  // tf2::Quaternion::setRPY (not normalised there either)
  const double hy = yaw * 0.5, hp = pitch * 0.5, hr = roll * 0.5;
  const double cy = std::cos(hy), sy = std::sin(hy), cp = std::cos(hp), sp = std::sin(hp), cr = std::cos(hr), sr = std::sin(hr);
  return Quaterniond(cr * cp * cy + sr * sp * sy, sr * cp * cy - cr * sp * sy, cr * sp * cy + sr * cp * sy, cr * cp * sy - sr * sp * cy);
}
// utils::extractRollPitch (superodom_utils.cpp:187-195): roll, pitch of the IMU orientation, yaw zeroed
inline Quaterniond extract_roll_pitch(const Quaterniond& imu) {
  double r, p, y;
  get_rpy(imu, r, p, y);
  return set_rpy(r, p, 0.0);
}

}  // namespace so_node_math
