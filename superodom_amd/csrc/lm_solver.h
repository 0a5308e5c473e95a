// lm_solver.h -- the trust-region Levenberg-Marquardt controller that drives the HIP evaluation
// kernel.  It restates what the reference obtains from Ceres through
//   LidarSLAM::solveOptimizationProblem   src/LidarProcess/LidarSlam.cpp:230-240
//     (TRUST_REGION / LEVENBERG_MARQUARDT, max_num_iterations = 4, DENSE_QR, every other option default)
// [UPSTREAM: ceres-solver 2.0.0 trust_region_minimizer.cc, levenberg_marquardt_strategy.cc,
//  dense_qr_solver.cc, trust_region_step_evaluator.cc -- not under /root/reference].
//
// Design: the kernel returns, per evaluation point, the loss-corrected normal equations
//   H = sum w_i J_i J_i^T,  g = sum w_i J_i r_i,  cost = 1/2 sum c_i rho(s_i)      (w_i = c_i rho'(s_i))
// in one fused pass (Ceres evaluates "cost only" at the candidate and re-evaluates with Jacobians after
// acceptance; both collapse into ONE fused evaluation here, so a solve costs 1 + (#iterations) passes).
// Ceres factors the A x 6 Jacobian by QR; with fp64 sums the 6x6 Cholesky route agrees to ~1e-12.
// The controller is a resumable state machine (begin -> [evaluate at next_pose -> feed]*), written
// SO_HD so that the same source runs on the host today and inside a single-wave kernel later.
#pragma once
#include "so_math.h"

#if defined(__HIP__)
#define SO_UNROLL _Pragma("unroll")
#else
#define SO_UNROLL
#endif

#ifndef SO_LM_STAMP
#define SO_LM_STAMP(dbg, i)  // profiling builds (kernels.hip, -DSO_LM_STAMPS) record a device clock here
#endif

namespace soicp {

struct LmSums {      // == so_icp_sums (include/so_icp.h), 45 doubles
  double cost, count;
  double Jtr[6];
  double JtJ[21];    // upper triangle, row-major
  double hist[16];
};

// Round 5: the controller's serial chain carries no division any more.  Ceres keeps the trust-region radius and divides by it
// (lm_diagonal = sqrt(diagonal / radius)), divides the cost change by the model cost change, and forms two norms when the
// candidate's cost arrives.  Here the state holds the RECIPROCAL radius -- every update of the radius is a product (x 1/f on an
// accepted step, x decrease_factor = a power of two on a rejected one, x 2 on an invalid one), so 1/radius is never formed --
// and lm_propose, right after it has the candidate, forms 1 / model_cost_change, |x - cand| and |cand| (inputs of the NEXT
// lm_feed's tolerance tests and step quality): on the device that work runs after the next pose has been handed to the other
// workgroups, i.e. beside their evaluation pass instead of in front of it.  rel = cost_change x (1 / model_cost_change) differs
// from the quotient by <= 1 ulp; it feeds the accept test (> 1e-3) and the radius factor only.
struct LmState {
  double x[7], cand[7];
  double H[36], g[6];
  double scale[6], diag[6];
  double x_cost, x_norm, inv_radius, decrease_factor, model_cost_change, initial_cost, count;
  double inv_model_cost_change, step_norm, cand_norm;  // formed by lm_propose with the candidate: 1 / model_cost_change, |x - cand|, |cand|
  int32_t iter, max_iter, reuse_diagonal, invalid_steps, num_successful, termination, done, lm_iterations;
};

// Ceres defaults in effect (solver.h, ceres 2.0.0)
struct LmConst {
  static constexpr double kInitialRadius = 1e4, kMaxRadius = 1e16, kMinRadius = 1e-32;
  static constexpr double kInitialInvRadius = 1e-4, kMinInvRadius = 1e-16 /* 1 / kMaxRadius */, kMaxInvRadius = 1e32 /* 1 / kMinRadius */;
  static constexpr double kMinRelativeDecrease = 1e-3, kMinLmDiagonal = 1e-6, kMaxLmDiagonal = 1e32;
  static constexpr double kFunctionTolerance = 1e-6, kGradientTolerance = 1e-10, kParameterTolerance = 1e-8;
  static constexpr int kMaxConsecutiveInvalidSteps = 5;
};

SO_HD void lm_unpack(const LmSums& s, double H[36], double g[6]) {
  int k = 0;
  SO_UNROLL
  for (int i = 0; i < 6; ++i)
    SO_UNROLL
    for (int j = i; j < 6; ++j) { H[6 * i + j] = s.JtJ[k]; H[6 * j + i] = s.JtJ[k]; ++k; }
  SO_UNROLL
  for (int i = 0; i < 6; ++i) g[i] = s.Jtr[i];
}

// | x - Plus(x, -g) |_inf : TrustRegionMinimizer::EvaluateGradientAndJacobian
SO_HD double lm_gradient_max_norm(const double x[7], const double g[6]) {
  double ng[6], xp[7], m = 0;
  SO_UNROLL
  for (int i = 0; i < 6; ++i) ng[i] = -g[i];
  pose_plus(x, ng, xp);
  SO_UNROLL
  for (int i = 0; i < 7; ++i) { double v = fabs(x[i] - xp[i]); if (v > m) m = v; }
  return m;
}

// gradient_max_norm <= gradient_tolerance.  Fast exit: a translation component of the gradient above 1e-6 alone puts
// |x - Plus(x, -g)|_inf far above the 1e-10 tolerance (for |x| < 1e5 the rounding of x - g stays below 3e-11), so the
// quaternion update + normalisation (~100 serial fp64 operations of the single-thread controller) is skipped.
SO_HD bool lm_gradient_converged(const double x[7], const double g[6]) {
  SO_UNROLL
  for (int i = 0; i < 3; ++i)
    if (fabs(g[i]) > 1e-6 && fabs(x[i]) < 1e5) return false;
  return lm_gradient_max_norm(x, g) <= LmConst::kGradientTolerance;
}

// Cholesky solve of a 6x6 SPD system; returns false when a pivot is not positive / result not finite.
SO_HD bool lm_chol6(double A[36], const double b[6], double y[6]) {
  double inv[6];  // reciprocal pivots: one division per column instead of one per element
  bool spd = true;  // no early exit: ONE basic block, so that the scheduler overlaps the columns' independent updates
                    // with the serial pivot -> rsqrt -> next pivot chain (a failed pivot poisons y, which is discarded)
  SO_UNROLL
  for (int j = 0; j < 6; ++j) {
    double d = A[6 * j + j];
    SO_UNROLL
    for (int k = 0; k < j; ++k) d = SO_FMA(-A[6 * j + k], A[6 * j + k], d);
    spd = spd && (d > 0.0);
#if defined(__HIP_DEVICE_COMPILE__)
    inv[j] = rsqrt(d);  // only 1/L_jj is used below; one long fp64 operation per column instead of sqrt + division
#else
    inv[j] = 1.0 / sqrt(d);
#endif
    SO_UNROLL
    for (int i = j + 1; i < 6; ++i) {
      double s = A[6 * i + j];
      SO_UNROLL
      for (int k = 0; k < j; ++k) s = SO_FMA(-A[6 * i + k], A[6 * j + k], s);
      A[6 * i + j] = s * inv[j];
    }
  }
  double z[6];
  SO_UNROLL
  for (int i = 0; i < 6; ++i) {
    double s = b[i];
    SO_UNROLL
    for (int k = 0; k < i; ++k) s = SO_FMA(-A[6 * i + k], z[k], s);
    z[i] = s * inv[i];
  }
  SO_UNROLL
  for (int i = 5; i >= 0; --i) {
    double s = z[i];
    SO_UNROLL
    for (int k = i + 1; k < 6; ++k) s = SO_FMA(-A[6 * k + i], y[k], s);
    y[i] = s * inv[i];
  }
  SO_UNROLL
  for (int i = 0; i < 6; ++i) spd = spd && isfinite(y[i]);
  return spd;
}

// What the next lm_feed needs besides the candidate's sums (see LmState); on the device the caller has published the candidate
// before this runs.
SO_HD void lm_after_candidate(LmState& S) {
  S.inv_model_cost_change = 1.0 / S.model_cost_change;
  double sn = 0, n2 = 0;
  SO_UNROLL
  for (int i = 0; i < 7; ++i) sn = SO_FMA(S.x[i] - S.cand[i], S.x[i] - S.cand[i], sn);
  SO_UNROLL
  for (int i = 0; i < 7; ++i) n2 = SO_FMA(S.cand[i], S.cand[i], n2);
  S.step_norm = sqrt(sn);
  S.cand_norm = sqrt(n2);
}

// One pass of the while(FinalizeIterationAndCheckIfMinimizerCanContinue()) loop up to the point
// where the candidate must be evaluated.  Returns 1 (evaluate S.cand) or 0 (solver finished).
SO_HD int lm_propose(LmState& S, double next_pose[7], unsigned long long* dbg = nullptr) {
  for (;;) {
    if (S.iter >= S.max_iter) { S.termination = 0; S.done = 1; return 0; }           // MaxSolverIterationsReached
    if (S.inv_radius >= LmConst::kMaxInvRadius) { S.termination = 5; S.done = 1; return 0; } // MinTrustRegionRadiusReached
    S.iter++;
    S.lm_iterations = S.iter;
    // jacobian_ is column-scaled: Hs = S H S, gs = S g
    if (!S.reuse_diagonal) {
      SO_UNROLL
      for (int j = 0; j < 6; ++j) {
        double v = S.H[7 * j] * S.scale[j] * S.scale[j];
        v = v < LmConst::kMinLmDiagonal ? LmConst::kMinLmDiagonal : v;
        S.diag[j] = v > LmConst::kMaxLmDiagonal ? LmConst::kMaxLmDiagonal : v;
      }
    }
    double A[36], Hs[36], gs[6], y[6], step[6];
    const double inv_radius = S.inv_radius;
    SO_UNROLL
    for (int i = 0; i < 6; ++i) {
      gs[i] = S.g[i] * S.scale[i];
      SO_UNROLL
      for (int j = i; j < 6; ++j) {  // H is symmetric: 21 scaled entries, mirrored
        Hs[6 * i + j] = S.H[6 * i + j] * S.scale[i] * S.scale[j]; Hs[6 * j + i] = Hs[6 * i + j];
        A[6 * i + j] = Hs[6 * i + j]; A[6 * j + i] = Hs[6 * i + j];
      }
      A[7 * i] += S.diag[i] * inv_radius;  // lm_diagonal^2 = diag / radius
    }
    SO_LM_STAMP(dbg, 3);
    const bool ok = lm_chol6(A, gs, y);  // (Hs + D^2) y = gs ; step = -y
    SO_LM_STAMP(dbg, 4);
    S.reuse_diagonal = 1;
    double mcc = 0;
    if (ok) {
      double sHs = 0, sg = 0;
      SO_UNROLL
      for (int i = 0; i < 6; ++i) {
        step[i] = -y[i];
        sg = SO_FMA(step[i], gs[i], sg);
      }
      SO_UNROLL
      for (int i = 0; i < 6; ++i) {  // s^T Hs s over the upper triangle: diagonal once, off-diagonal terms twice
        double r = 0;
        SO_UNROLL
        for (int j = i + 1; j < 6; ++j) r = SO_FMA(Hs[6 * i + j], step[j], r);
        sHs = SO_FMA(step[i], SO_FMA(Hs[7 * i], step[i], 2.0 * r), sHs);
      }
      mcc = SO_FMA(-0.5, sHs, -sg);  // -(J s)^T (r + J s / 2)
    }
    if (!ok || !(mcc > 0.0)) {  // HandleInvalidStep
      if (++S.invalid_steps >= LmConst::kMaxConsecutiveInvalidSteps) { S.termination = 5; S.done = 1; return 0; }
      S.inv_radius *= 2.0;  // radius *= 0.5
      continue;
    }
    SO_LM_STAMP(dbg, 5);
    S.invalid_steps = 0;
    S.model_cost_change = mcc;
    double delta[6];
    SO_UNROLL
    for (int i = 0; i < 6; ++i) delta[i] = step[i] * S.scale[i];
    pose_plus(S.x, delta, S.cand);
    SO_UNROLL
    for (int i = 0; i < 7; ++i) next_pose[i] = S.cand[i];
    SO_LM_STAMP(dbg, 6);
    lm_after_candidate(S);
    return 1;
  }
}

SO_HD int lm_begin(LmState& S, const double x0[7], const LmSums& sums, int max_iterations, double next_pose[7]) {
  SO_UNROLL
  for (int i = 0; i < 7; ++i) { S.x[i] = x0[i]; S.cand[i] = x0[i]; }
  S.iter = 0; S.max_iter = max_iterations; S.reuse_diagonal = 0; S.invalid_steps = 0; S.num_successful = 0;
  S.termination = 0; S.done = 0; S.lm_iterations = 0;
  S.inv_radius = LmConst::kInitialInvRadius; S.decrease_factor = 2.0; S.model_cost_change = 0;
  S.inv_model_cost_change = 0; S.step_norm = 0; S.cand_norm = 0;
  S.count = sums.count; S.x_cost = sums.cost; S.initial_cost = sums.cost;
  lm_unpack(sums, S.H, S.g);
  SO_UNROLL
  for (int j = 0; j < 6; ++j) { S.scale[j] = 1.0; S.diag[j] = 0; }
  if (!(sums.count > 0)) { S.termination = 4; S.done = 1; return 0; }  // no residual blocks: nothing to minimise
  SO_UNROLL
  for (int j = 0; j < 6; ++j) S.scale[j] = 1.0 / (1.0 + sqrt(S.H[7 * j]));  // jacobi_scaling, fixed at iteration 0
  double n2 = 0;
  SO_UNROLL
  for (int i = 0; i < 7; ++i) n2 = SO_FMA(S.x[i], S.x[i], n2);
  S.x_norm = sqrt(n2);
  if (lm_gradient_converged(S.x, S.g)) { S.termination = 3; S.done = 1; return 0; }
  return lm_propose(S, next_pose);
}

SO_HD int lm_feed(LmState& S, const LmSums& sums, double next_pose[7], unsigned long long* dbg = nullptr) {
  if (S.done) return 0;
  SO_LM_STAMP(dbg, 0);
  const double cand_cost = sums.cost;
  // ParameterToleranceReached
  const double sn = S.step_norm;
  if (sn <= LmConst::kParameterTolerance * (S.x_norm + LmConst::kParameterTolerance)) { S.termination = 2; S.done = 1; return 0; }
  // FunctionToleranceReached
  const double cost_change = S.x_cost - cand_cost;
  if (fabs(cost_change) <= LmConst::kFunctionTolerance * S.x_cost) { S.termination = 1; S.done = 1; return 0; }
  const double rel = cost_change * S.inv_model_cost_change;  // TrustRegionStepEvaluator::StepQuality, monotonic
  SO_LM_STAMP(dbg, 1);
  if (rel > LmConst::kMinRelativeDecrease) {             // HandleSuccessfulStep
    SO_UNROLL
    for (int i = 0; i < 7; ++i) S.x[i] = S.cand[i];
    S.x_norm = S.cand_norm;
    S.x_cost = cand_cost;
    lm_unpack(sums, S.H, S.g);
    S.num_successful++;
    const double u = 2.0 * rel - 1.0;
    double f = 1.0 - u * u * u;  // LevenbergMarquardtStrategy::StepAccepted: 1 - pow(2 rho - 1, 3)
    if (f < 1.0 / 3.0) f = 1.0 / 3.0;
    S.inv_radius = S.inv_radius * f;  // radius = radius / f
    if (S.inv_radius < LmConst::kMinInvRadius) S.inv_radius = LmConst::kMinInvRadius;  // radius <= kMaxRadius
    S.decrease_factor = 2.0;
    S.reuse_diagonal = 0;
    // FinalizeIterationAndCheckIfMinimizerCanContinue tests MaxSolverIterationsReached before GradientToleranceReached
    // [UPSTREAM ceres 2.0.0 trust_region_minimizer.cc]: both on the last iteration => termination 0, not 3
    if (S.iter >= S.max_iter) { S.termination = 0; S.done = 1; return 0; }
    if (lm_gradient_converged(S.x, S.g)) { S.termination = 3; S.done = 1; return 0; }
  } else {  // HandleUnsuccessfulStep / StepRejected
    S.inv_radius = S.inv_radius * S.decrease_factor;  // radius = radius / decrease_factor (a power of two: exact)
    S.decrease_factor *= 2.0;
    S.reuse_diagonal = 1;
  }
  SO_LM_STAMP(dbg, 2);
  return lm_propose(S, next_pose, dbg);
}

}  // namespace soicp
