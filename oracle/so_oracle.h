/*
 * so_oracle.h -- CPU ORACLE for the scan-to-map ICP hot path of superxslam/SuperOdom.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it.  Nothing under superodom_amd/ links,
 * imports or calls it; the product path fails loudly when its HIP library is missing.
 *
 * It is a plain-C restatement ("Oracle-A" of SURVEY.md section 8c) of the reference path
 *   LidarSLAM::performLocalizationAndMapping   super_odometry/src/LidarProcess/LidarSlam.cpp:107-152
 * with EXACT cube-restricted 5-NN (= the reference built with DONT_USE_SELF_OCTREE, i.e. the
 * pcl::KdTreeFLANN branch of LidarProcess/LocalMap.h:509-514).  All `file:line` citations
 * below are relative to /root/reference/super_odometry/.
 *
 * PINNING STATUS
 *   - k-NN stage: pinned against the reference's own header-only octree
 *     (include/super_odometry/flann/octree.h + nanoflann.h), compiled verbatim into
 *     oracle/_ref/libref_octree.so by oracle/Makefile and compared in
 *     tests/test_oracle_vs_reference_octree.py (bit-identical d2 arithmetic, identical
 *     neighbour lists on scenes where the stock octree's two pruning bugs are inert).
 *   - Everything that the reference delegates to third-party code that is NOT under
 *     /root/reference -- Ceres 2.0.0 (trust-region LM, TukeyLoss, ScaledLoss, DENSE_QR),
 *     Eigen 3.4.0 (SelfAdjointEigenSolver, colPivHouseholderQr), PCL 1.12.1 (VoxelGrid),
 *     tf2 (getRPY/setRPY) -- is restated from the published algorithms.  The reference
 *     ships no tests, golden vectors or fixtures for this path, so
 *     for those stages this oracle is "PARITY UNPINNED": it is cross-checked only against
 *     numpy/scipy known-answer tests that this repo authors (tests/golden/).
 */
#ifndef SO_ORACLE_H
#define SO_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* LocalMap constants, include/super_odometry/LidarProcess/LocalMap.h:131-138 */
#define ORC_MAP_W 21
#define ORC_MAP_H 21
#define ORC_MAP_D 11
#define ORC_MAP_NUM (ORC_MAP_W * ORC_MAP_H * ORC_MAP_D) /* 4851 */
#define ORC_CUBE 50.0
#define ORC_HALF_CUBE 25.0

/* MatchingResult, include/super_odometry/LidarProcess/LidarSlam.h:85-94 */
enum {
  ORC_SUCCESS = 0,
  ORC_NOT_ENOUGH_NEIGHBORS = 1,
  ORC_NEIGHBORS_TOO_FAR = 2,
  ORC_BAD_PCA_STRUCTURE = 3,
  ORC_INVALID_NUMERICAL = 4,
  ORC_MSE_TOO_LARGE = 5,
  ORC_UNKNOWN = 6,
  ORC_N_REJECT = 7
};
#define ORC_N_OBS 9 /* Feature_observability, LidarSlam.h:96-107 */
#define ORC_MAX_OUTER 16

typedef struct orc_map orc_map;

typedef struct {
  int max_iterations;        /* LocalizationICPMaxIter, LidarSlam.h:273; YAML max_iterations=5 */
  int lm_max_iterations;     /* options.max_num_iterations = 4, LidarSlam.cpp:232 */
  int max_surface_features;  /* OptSet.max_surface_features; <0 means "all", 0 drops every point (LS:346-359) */
  int k;                     /* LocalizationPlaneDistanceNbrNeighbors = 5, LidarSlam.h:277 */
  int tukey_variant;         /* 0: Ceres<=2.0 (rho'=0.5(1-s/a^2)^2), 1: Ceres>=2.1 (rho'=(1-s/a^2)^2) */
  int use_grid_knn;          /* 0: brute force inside the cube; 1: exact uniform-grid search; 2: orc_set_knn_hook engine (Oracle-B) */
  double yaw_ratio;          /* OptSet.yaw_ratio (0.0 in shipped calibrations) */
  double velocity_failure_threshold; /* 30 */
} orc_config;

/* One accepted/rejected correspondence = OptimizationParameter, LidarSlam.h:209-222 (fields used). */
typedef struct {
  double p[3];     /* Xvalue  = scan point, sensor frame */
  double n[3];     /* NormDir = unit plane normal (from the 5x3 LS fit) */
  double d;        /* negative_OA_dot_norm */
  double coeff;    /* residualCoefficient */
  int32_t status;  /* MatchingResult */
  int32_t obs[4];  /* feature.observability */
  float nbr[15];   /* the 5 neighbours (debug / parity) */
  float d2[5];
  double eig[3];   /* ascending eigenvalues of the scatter matrix */
} orc_corr;

typedef struct {
  double translation_norm, rotation_norm; /* IterationStats.msg */
  int32_t num_surf;                       /* accepted correspondences A */
  int32_t lm_iterations;                  /* minimizer iterations executed (excluding iteration 0) */
  int32_t num_successful_steps;           /* ceres summary.num_successful_steps */
  int32_t termination;                    /* 0 max-iter, 1 func-tol, 2 param-tol, 3 grad-tol, 4 no-residuals, 5 fail */
  double initial_cost, final_cost;
  int32_t reject_hist[ORC_N_REJECT];
  int32_t obs_hist[ORC_N_OBS];
  double pose_after[7];
} orc_iter_stats;

typedef struct {
  int32_t surf_from_map_num;   /* laser_cloud_surf_from_map_num */
  int32_t surf_stack_num;      /* laser_cloud_surf_stack_num */
  int32_t n_iterations;        /* outer iterations executed */
  int32_t startup_count;       /* startupCount side effect of checkMotionThresholds */
  int32_t pos_in_map[3];
  double total_translation, total_rotation;
  double translation_from_last, rotation_from_last;
  double uncertainty[6];       /* x y z roll pitch yaw -- from the histogram handed in */
  double JtJ[36], Jtr[6];      /* final (loss-corrected, unscaled) normal equations */
  orc_iter_stats iters[ORC_MAX_OUTER];
} orc_stats;

/* ---- LocalMap (LocalMap.h) -------------------------------------------------------------- */
orc_map *orc_map_create(void);
void orc_map_destroy(orc_map *m);
void orc_map_set_resolution(orc_map *m, float lineRes, float planeRes);
void orc_map_set_origin(orc_map *m, const double t[3], int origin_out[3]);           /* LocalMap.h:146-164 */
void orc_map_shift(orc_map *m, const double t[3], int pos_out[3]);                   /* LocalMap.h:169-287 */
void orc_map_get_origin(const orc_map *m, int origin_out[3]);
/* addSurfPointCloud, LocalMap.h:591-645: bin into cubes, then per touched cube VoxelGrid(planeRes). */
int orc_map_add_surf(orc_map *m, const float *xyz, size_t n, size_t stride_floats);
/* Test helper: bin into cubes, keep order, NO voxel filter (loads an already-filtered map). */
int orc_map_add_surf_raw(orc_map *m, const float *xyz, size_t n, size_t stride_floats);
int orc_map_count_5x5(const orc_map *m, const int pos[3]);                           /* LocalMap.h:292-318 */
size_t orc_map_size(const orc_map *m);
size_t orc_map_export(const orc_map *m, float *xyz, size_t cap); /* cube-index order, then in-cube order */
size_t orc_map_cube_size(const orc_map *m, int cube_ind);

/* nearestKSearchSurf, LocalMap.h:481-525, exact variant.  Returns 1 if the cube is valid and has a
 * tree, else 0.  Unfilled slots keep idx 0 / d2 FLT_MAX exactly like nanoflann.h:87-100. */
int orc_knn_surf(const orc_map *m, const float q[3], int k, int use_grid,
                 float *nbr_xyz /*k*3*/, float *d2 /*k*/, int64_t *idx_in_cube /*k*/, int *cube_ind);

/* PCL VoxelGrid restatement on a bare array (exposed for tests). Returns number of output points. */
size_t orc_voxel_grid(const float *xyz, size_t n, float leaf, float *out_xyz);

/* ---- per-point correspondence, LidarSlam.cpp:514-572 ------------------------------------- */
void orc_plane_match(const orc_map *m, const double pose[7], const float p_sensor[3],
                     const orc_config *cfg, orc_corr *out);

/* ---- numerics exposed for known-answer tests ----------------------------------------------- */
void orc_eig3_sym(const double S[9], double evals[3], double evecs[9] /*col-major: evecs[3*j+i]*/);
int orc_plane_ls5(const double A[15] /*row-major 5x3*/, double x[3]); /* colPivHouseholderQr().solve(-1) */
void orc_residual_jacobian(const double pose[7], const double p[3], const double n[3], double d,
                           double *r, double J6[6]);                  /* lidarOptimization.cpp:55-80 */
void orc_pose_plus(const double x[7], const double delta[6], double out[7]); /* pose_local_parameterization.cpp:7-23 */
void orc_tukey_scaled(double s, double a, double coeff, int variant, double rho[3]);
/* cost + corrected normal equations for a correspondence set (status==SUCCESS only) */
void orc_evaluate(const orc_corr *c, size_t n, const double pose[7], float planeRes, int tukey_variant,
                  double *cost, double JtJ[36], double Jtr[6], int *count);
/* Ceres-style trust-region LM (DENSE_QR on the stacked Jacobian).  pose is updated in place. */
void orc_lm_solve(const orc_corr *c, size_t n, double pose[7], float planeRes, const orc_config *cfg,
                  orc_iter_stats *st);
void orc_uncertainty_from_hist(const int32_t obs_hist[ORC_N_OBS], double u[6]);      /* LidarSlam.cpp:915-964 */
int orc_should_process(size_t index, size_t n_points, int max_surface_features);     /* LidarSlam.cpp:346-359 */
void orc_yaw_correction(double pose[7], const double last_pose[7], double yaw_ratio);/* LidarSlam.cpp:891-913 */

/* ---- whole registration = performLocalizationAndMapping minus the map insert ----------------- */
/* returns 0 ok, 1 not-enough-map-features (pose_out = pose_in, stats mostly empty) */
int orc_register(orc_map *m, const float *scan_xyz, size_t n, size_t stride_floats,
                 const double pose_in[7], const orc_config *cfg, const int32_t prev_obs_hist[ORC_N_OBS],
                 double pose_out[7], orc_stats *stats, orc_corr *last_corrs /*nullable, n entries*/);
/* transformAndAddToMap, LidarSlam.cpp:60-80 */
int orc_transform_and_add(orc_map *m, const float *scan_xyz, size_t n, size_t stride_floats, const double pose[7]);

/* Oracle-B: k-NN of one cube through an external engine (oracle/_ref: the reference's flann/octree.h); selected by
 * orc_config::use_grid_knn == 2.  xyz / n = the cube's points (stable until the next insert into that cube). */
typedef void (*orc_knn_hook_t)(int cube_ind, const float *xyz, size_t n, const float q[3], int k, int64_t *idx, float *d2);
void orc_set_knn_hook(orc_knn_hook_t hook);

/* ---- featureExtraction::removePointDistortion, featureExtraction.cpp:223-314 -----------------------
 * points: records of stride bytes, float x y z at 0 4 8, float time at time_off; rewritten in place.
 * poses: n_poses x 8 doubles {time, px py pz, qx qy qz qw}, strictly increasing times.  imu != 0: the buffer holds IMU
 * orientations (positions ignored) and the result is wrapped in T_l_i * . * T_i_l (T_i_l = {tx ty tz qx qy qz qw}).
 * start_sensor[7] receives {t_w_original_l, q_w_original_l}.  Returns the number of points stamped at / after the last
 * pose (no successor in the buffer: undefined in the reference; the last pose is used). */
size_t orc_deskew(void *points, size_t n, size_t stride, size_t time_off, double lidar_start_time, const double *poses, size_t n_poses,
                  int imu, const double T_i_l[7], double start_sensor[7]);

int orc_num_threads(void);
void orc_set_num_threads(int n);

#ifdef __cplusplus
}
#endif
#endif
