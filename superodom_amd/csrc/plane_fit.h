// plane_fit.h -- the per-correspondence plane fit of the scan-to-map ICP path, host + device (SO_HD):
//   LidarSLAM::ComputePlaneDistanceParameters after the neighbour search   src/LidarProcess/LidarSlam.cpp:533-571
//     computePCAForFeature         :749-790  (+ utils::ComputePCA, include/super_odometry/utils/superodom_utils.h:143-151)
//     computePlaneQualityMetrics   :792-844  (matA0.colPivHouseholderQr().solve(matB0), d = 1/|x|, n = x/|x|, inlier gate)
//     FeatureObservabilityAnalysis :574-693
//     residualCoefficient          :568
// (paths relative to /root/reference/super_odometry/).
//
// The fit pass of solve_kernel executes this once per query and outer iteration; it was ~1 900 fp64 instructions per query
// (two thirds of the solve's instruction count), 500 of them the 5x3 column-pivoted Householder factorisation.  Round 4
// replaces the factorisation by the closed form of the SAME least-squares problem: with mu = mean of the five points and
// S = sum (p - mu)(p - mu)^T (both already formed for the PCA gate),
//     min_x |A x + 1|^2 = x^T S x + 5 (mu.x + 1)^2      (A^T A = S + 5 mu mu^T, A^T 1 = 5 mu, because sum (p - mu) = 0)
//     x = -5 S^-1 mu / (1 + 5 mu^T S^-1 mu)             (Sherman-Morrison)
// so with w = adj(S) mu and D = det(S) + 5 mu.w:   n = x/|x| = -w/|w|,   d = 1/|x| = D / (5 |w|).
// adj(S) and det(S) are the cofactors the eigen-solver computes anyway.  The PCA gate (lambda0 >= 1e-6, lambda1 >= 0.1 lambda2)
// has already certified S as well conditioned (cond <= ~1e6) when this runs, and the centred formulation is better conditioned
// than QR on the un-centred A: against an 80-bit evaluation of the same problem on 20 000 random clusters up to 100 m from
// the origin the closed form is within 1.4e-13 of the true normal, column-pivoted QR within 8.8e-12
// (tests/test_plane_fit_host.py).  The reference factorisation stays available for A/B: SOICP_ABLATE=4096 (PROF kernels).
#pragma once
#include "so_math.h"

namespace soicp {

#define SO_FIT_SUCCESS 0
#define SO_FIT_BAD_PCA 3
#define SO_FIT_INVALID 4
#define SO_FIT_MSE 5

// a / b: on the device ~9 instructions instead of the ~28 of the IEEE sequence (v_rcp_f64, two Newton steps, one residual
// correction: within 1 ulp of the correctly rounded quotient); plain division on the host
SO_HD double fit_div(double a, double b) {
#if defined(__HIP_DEVICE_COMPILE__)
  double r = __builtin_amdgcn_rcp(b);
  r = __builtin_fma(__builtin_fma(-b, r, 1.0), r, r);
  r = __builtin_fma(__builtin_fma(-b, r, 1.0), r, r);
  const double q = a * r;
  return __builtin_fma(__builtin_fma(-b, q, a), r, q);
#else
  return a / b;
#endif
}
SO_HD double fit_rsqrt(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
  return rsqrt(x);
#else
  return 1.0 / sqrt(x);
#endif
}

// Pose-dependent constants of FeatureObservabilityAnalysis (LidarSlam.cpp:624-638): the sensor axes R_f e_i with R_f the
// FLOAT quaternion of the current pose.  Constant over a pass: computed once per thread, not once per query.
struct ObsAxes { float ax[3][3]; };
SO_HD ObsAxes obs_axes(const Pose& pose) {
  ObsAxes o;
  const float qf[4] = {(float)pose.q[0], (float)pose.q[1], (float)pose.q[2], (float)pose.q[3]};
  quat_rotate<float>(qf, 1.f, 0.f, 0.f, o.ax[0][0], o.ax[0][1], o.ax[0][2]);
  quat_rotate<float>(qf, 0.f, 1.f, 0.f, o.ax[1][0], o.ax[1][1], o.ax[1][2]);
  quat_rotate<float>(qf, 0.f, 0.f, 1.f, o.ax[2][0], o.ax[2][1], o.ax[2][2]);
  return o;
}

// Symmetric 3x3 eigen-decomposition without iterations (restates the RESULT of Eigen::SelfAdjointEigenSolver<Matrix3d>,
// superodom_utils.h:150: ascending eigenvalues + the eigenvector of the smallest): the spectrum of a 5-point scatter matrix
// is lambda0 << lambda1 <= lambda2 for anything that can pass the gates, so
//   lambda0    = smallest root of the characteristic cubic by Newton from 0 (monotone from below for a polynomial with real
//                roots; the matrix is first scaled to unit max-norm),
//   lambda1,2  = roots of the deflated quadratic,
//   n          = the largest of the three row cross products of (A - lambda0 I), normalised.
// Eigenvalues agree with cyclic Jacobi to ~1e-14 relative, the normal to ~1e-15 when lambda0 is separated; only the gates
// (LidarSlam.cpp:772) and the float observability labels consume them.
// Also returns what the closed-form plane needs: the scale mx, the cofactors of the SCALED matrix (adj = {c00, c01, c02, c11,
// c12, c22}) and its determinant.  (Arithmetic and operation order of the eigenvalue part are those of round 3: unfused.)
SO_HD void eig3_sym_direct_adj(double a00, double a01, double a02, double a11, double a12, double a22, double ev[3], double nrm[3],
                               double adj[6], double& det_s, double& scale) {
  const double mx = fmax(fmax(fmax(fabs(a00), fabs(a11)), fabs(a22)), fmax(fmax(fabs(a01), fabs(a02)), fabs(a12)));
  scale = mx; det_s = 0.0;
  adj[0] = adj[1] = adj[2] = adj[3] = adj[4] = adj[5] = 0.0;
  if (!(mx > 0.0)) { ev[0] = ev[1] = ev[2] = 0.0; nrm[0] = 1.0; nrm[1] = 0.0; nrm[2] = 0.0; return; }
  const double is = fit_div(1.0, mx);
  a00 *= is; a01 *= is; a02 *= is; a11 *= is; a12 *= is; a22 *= is;
  // p(l) = -l^3 + c2 l^2 - c1 l + c0
  const double c2 = a00 + a11 + a22;
  const double m00 = a11 * a22 - a12 * a12, m11 = a00 * a22 - a02 * a02, m22 = a00 * a11 - a01 * a01;
  const double c1 = m00 + m11 + m22;
  const double t01 = a01 * a22 - a12 * a02, t02 = a01 * a12 - a11 * a02;  // -cofactor(0,1), +cofactor(0,2)
  const double c0 = a00 * m00 - a01 * t01 + a02 * t02;
  adj[0] = m00; adj[1] = -t01; adj[2] = t02; adj[3] = m11; adj[4] = a01 * a02 - a00 * a12; adj[5] = m22;
  det_s = c0;
  double l = 0.0;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll 1
#endif
  for (int it = 0; it < 60; ++it) {  // 2-3 iterations when lambda0 is separated; linear convergence only towards a double root
    const double f = ((-l + c2) * l - c1) * l + c0;      // p(l)
    const double df = (-3.0 * l + 2.0 * c2) * l - c1;    // p'(l) < 0 left of the smallest root
    if (!(df < 0.0) || !(f > 0.0)) break;   // at (or, by rounding, just past) the root
    const double step = fit_div(f, df);      // < 0: the iterate moves right, never beyond the root (p is convex there)
    l -= step;
    if (!(-step > 4e-16)) break;             // the matrix has unit max-norm: below the noise of p(l)
  }
  if (!(l > 0.0)) l = fmax(l, 0.0);
  // deflate: l1 + l2 = c2 - l, l1 l2 = c1 - l (c2 - l)
  const double sm = c2 - l, pr = c1 - l * sm;
  double disc = sm * sm - 4.0 * pr;
  disc = disc > 0.0 ? sqrt(disc) : 0.0;
  const double l2 = 0.5 * (sm + disc);
  const double l1 = (l2 > 0.0) ? fit_div(pr, l2) : 0.0;  // the smaller root from the product: no cancellation
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
  const double inv = fit_rsqrt(nn);
  nrm[0] = v0 * inv; nrm[1] = v1 * inv; nrm[2] = v2 * inv;
}

// FeatureObservabilityAnalysis, LidarSlam.cpp:574-693: float arithmetic on float-cast inputs; returns the three labels the
// histogram counts (rot#1, rot#2, trans#1; LidarSlam.cpp:336-339).
// The translation label is argmax_a planar^2 * |n.axis_a| in float with ties to the lower label.  planar^2 > 0 is a common
// factor: unless two of the |n.axis_a| are within 2^-20 of each other (or planar^2 is zero / tiny) the label is the argmax of
// |n.axis_a| alone -- rounding of the float products cannot reorder or tie operands that far apart -- and the three fp64 square
// roots + division behind planar^2 (LidarSlam.cpp:605-620, ~80 instructions) are skipped.  Otherwise the reference's
// arithmetic runs as written.  The labels are identical in both cases.
SO_HD void fit_observability(const double pw[3], const double ev[3], const double nrm[3], const ObsAxes& A, int& o0, int& o1, int& o2,
                              bool as_written = false /* tests: always the reference's arithmetic */) {
  const float px = (float)pw[0], py = (float)pw[1], pz = (float)pw[2];
  const float nx = (float)nrm[0], ny = (float)nrm[1], nz = (float)nrm[2];
  const float cx = py * nz - pz * ny, cy = pz * nx - px * nz, cz = px * ny - py * nx;
  float rot[6], f[3];
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
  for (int a = 0; a < 3; ++a) {
    const float v = cx * A.ax[a][0] + cy * A.ax[a][1] + cz * A.ax[a][2];
    rot[2 * a] = v; rot[2 * a + 1] = -v;
    f[a] = fabsf(nx * A.ax[a][0] + ny * A.ax[a][1] + nz * A.ax[a][2]);
  }
  // descending order, ties keep the lower label (stable insertion sort in libstdc++ for n < 16)
  int b1 = 0;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
  for (int a = 1; a < 6; ++a) if (rot[a] > rot[b1]) b1 = a;
  int b2 = -1;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
  for (int a = 0; a < 6; ++a) if (a != b1 && (b2 < 0 || rot[a] > rot[b2])) b2 = a;
  int t1 = 0;
  if (f[1] > f[t1]) t1 = 1;
  if (f[2] > f[t1]) t1 = 2;
  const float fb = f[t1] * (1.0f - 9.5367431640625e-7f);  // 1 - 2^-20
  const bool clear = (t1 == 0 || f[0] < fb) && (t1 == 1 || f[1] < fb) && (t1 == 2 || f[2] < fb);
  if (as_written || !(clear && ev[1] > 1.0001 * ev[0] && ev[0] > 0.0)) {  // rare: the reference's arithmetic as written
    const double l1 = sqrt(ev[2]), l2 = sqrt(ev[1]), l3 = sqrt(ev[0]);
    const double planar_2 = fit_div(l2 - l3, l1);
    const float psq = (float)(planar_2 * planar_2);
    const float tr[3] = {psq * f[0], psq * f[1], psq * f[2]};
    t1 = 0;
    if (tr[1] > tr[t1]) t1 = 1;
    if (tr[2] > tr[t1]) t1 = 2;
  }
  o0 = b1; o1 = b2; o2 = 6 + t1;
}

// One correspondence: PCA gate, plane, inlier gate, coefficient, observability labels.  nb = the five neighbours (float
// world coordinates), pw = the query in the world frame.  Returns the MatchingResult (SO_FIT_*).
//   sq_max_dist_f  = 3 * planeRes evaluated in float (LidarSlam.cpp:526), max_point_dist = planeRes / 2.0 (:820)
SO_HD int plane_fit5(const float nb[15], const double pw[3], const ObsAxes& axes, float sq_max_dist_f, double max_point_dist,
                     double nd[4], double& coeff, int obs[3], bool obs_as_written = false) {
  // PCA (LidarSlam.cpp:756-775, utils/superodom_utils.h:143-151)
  double mx = 0, my = 0, mz = 0;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
  for (int j = 0; j < 5; ++j) { mx += (double)nb[3 * j]; my += (double)nb[3 * j + 1]; mz += (double)nb[3 * j + 2]; }
  mx = fit_div(mx, 5.0); my = fit_div(my, 5.0); mz = fit_div(mz, 5.0);
  double s00 = 0, s01 = 0, s02 = 0, s11 = 0, s12 = 0, s22 = 0;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
  for (int j = 0; j < 5; ++j) {
    const double a = (double)nb[3 * j] - mx, b = (double)nb[3 * j + 1] - my, c = (double)nb[3 * j + 2] - mz;
    s00 += a * a; s01 += a * b; s02 += a * c; s11 += b * b; s12 += b * c; s22 += c * c;
  }
  double ev[3], nrm[3], adj[6], det_s, scale;
  eig3_sym_direct_adj(s00, s01, s02, s11, s12, s22, ev, nrm, adj, det_s, scale);
  if (ev[0] < 1e-6 || fit_div(ev[1], ev[2]) < 0.1) return SO_FIT_BAD_PCA;  // LidarSlam.cpp:772
  // LS plane A x = -1 in closed form (see the head of this file): w = adj(S) mu, D = det(S) + 5 mu.w  (scaled: S / scale)
  const double w0 = __builtin_fma(adj[0], mx, __builtin_fma(adj[1], my, adj[2] * mz));
  const double w1 = __builtin_fma(adj[1], mx, __builtin_fma(adj[3], my, adj[4] * mz));
  const double w2 = __builtin_fma(adj[2], mx, __builtin_fma(adj[4], my, adj[5] * mz));
  const double ww = __builtin_fma(w0, w0, __builtin_fma(w1, w1, w2 * w2));
  const double D = __builtin_fma(scale, det_s, 5.0 * __builtin_fma(w0, mx, __builtin_fma(w1, my, w2 * mz)));
  if (!(ww > 0.0) || !(D > 0.0) || !(D < 1.79e308)) return SO_FIT_INVALID;  // x = -5 w / D not finite / zero (LidarSlam.cpp:809-812)
  const double rs = fit_rsqrt(ww);
  const double n0 = -w0 * rs, n1 = -w1 * rs, n2 = -w2 * rs;  // LidarSlam.cpp:816
  const double d = (0.2 * D) * rs;                           // LidarSlam.cpp:815
  double sum = 0;
  bool too_far = false;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
  for (int j = 0; j < 5; ++j) {
    const double dist = fabs(n0 * (double)nb[3 * j] + n1 * (double)nb[3 * j + 1] + n2 * (double)nb[3 * j + 2] + d);
    too_far |= dist > max_point_dist;                             // LidarSlam.cpp:832
    sum += dist;
  }
  if (too_far) return SO_FIT_MSE;
  const double mean_abs = fit_div(sum, 5.0);
  if (pw[0] * nrm[0] + pw[1] * nrm[1] + pw[2] * nrm[2] < 0) { nrm[0] = -nrm[0]; nrm[1] = -nrm[1]; nrm[2] = -nrm[2]; }  // :553-561
  fit_observability(pw, ev, nrm, axes, obs[0], obs[1], obs[2], obs_as_written);
  coeff = 1.0 - sqrt(fit_div(mean_abs, (double)sq_max_dist_f));           // LidarSlam.cpp:568
  nd[0] = n0; nd[1] = n1; nd[2] = n2; nd[3] = d;
  return SO_FIT_SUCCESS;
}

}  // namespace soicp
