/*
 * so_icp.h -- C ABI of libsoicp: the MI355X-native (gfx950 / HIP) scan-to-map ICP hot path that
 * replaces, behind the reference's own member-function seams, the CPU path
 *
 *     LidarSLAM::Localization -> performLocalizationAndMapping      (Seam A)
 *     LocalMap::nearestKSearchSurf                                  (Seam B)
 *
 * of superxslam/SuperOdom.  The reference has no plugin/FFI interface, so every
 * entry point below names the reference member function it stands in for.  Citations are relative
 * to /root/reference/super_odometry/ with the short names
 *     LS.cpp = src/LidarProcess/LidarSlam.cpp          LS.h = include/super_odometry/LidarProcess/LidarSlam.h
 *     LM.h   = include/super_odometry/LidarProcess/LocalMap.h
 *     lmap.cpp = src/LaserMapping/laserMapping.cpp
 *
 * Conventions: plain pointers and sizes, no C++/torch types; `int` status (0 ok, >0 soft
 * condition, <0 error; text via so_icp_last_error); nothing throws across the boundary; the caller
 * owns every buffer; a context is single-caller (the reference's LidarSLAM is non-re-entrant,
 * LS.cpp:7-9) and owns its HIP stream, device buffers and (optionally) its RCCL communicator.
 * Pose layout = the reference's pose_parameters[7] (LS.cpp:7-9): {tx,ty,tz, qx,qy,qz,qw}.
 * INTEGRATION.md shows the LidarSLAM-side adapter a maintainer would add.
 */
#ifndef SO_ICP_H
#define SO_ICP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SO_ICP_ABI_VERSION 4 /* v4: so_icp_register_sequence; so_icp_timing grew (seq_chained, seq_chain_breaks).  v3: the config, timing and prefilter-info structs grew (shard_mode, solve_workgroups, staging / packing counters, statistic_in_input_order) */

/* LocalMap geometry, LM.h:131-138 */
#define SO_ICP_MAP_W 21
#define SO_ICP_MAP_H 21
#define SO_ICP_MAP_D 11
#define SO_ICP_MAP_NUM 4851
#define SO_ICP_MAX_OUTER 16

/* status codes */
#define SO_ICP_OK 0
#define SO_ICP_NOT_ENOUGH_MAP_FEATURES 1 /* LS.cpp:113-116: surf_from_map_num <= 50, pose = guess */
#define SO_ICP_MAP_SEEDED 2              /* Localization(initialization=false): LS.cpp:45-46, 83-94 */
#define SO_ICP_STAGE_DECLINED 3          /* so_icp_stage_scan: every staging slot holds a scan that is still needed; not staged (soft) */
#define SO_ICP_E_INVALID (-1)
#define SO_ICP_E_HIP (-2)
#define SO_ICP_E_NOMEM (-3)
#define SO_ICP_E_RCCL (-4)
#define SO_ICP_E_UNSUPPORTED (-5)

/* MatchingResult, LS.h:85-94 (index into so_icp_iter_stats.reject_hist) */
enum {
  SO_ICP_MATCH_SUCCESS = 0,
  SO_ICP_MATCH_NOT_ENOUGH_NEIGHBORS = 1,
  SO_ICP_MATCH_NEIGHBORS_TOO_FAR = 2,
  SO_ICP_MATCH_BAD_PCA_STRUCTURE = 3,
  SO_ICP_MATCH_INVALID_NUMERICAL = 4,
  SO_ICP_MATCH_MSE_TOO_LARGE = 5,
  SO_ICP_MATCH_UNKNOWN = 6,
  SO_ICP_N_REJECT = 7
};
#define SO_ICP_N_OBS 9 /* Feature_observability, LS.h:96-107 */

typedef struct so_icp_ctx so_icp_ctx;

/* Knobs the node pushes into LidarSLAM before each call (lmap.cpp:103-120, 648-649, 703-711) plus
 * device placement.  Fill with so_icp_default_config() first. */
typedef struct {
  int32_t abi_version;          /* SO_ICP_ABI_VERSION */
  int32_t device_id;            /* HIP device ordinal (one process per GPU); < 0: host-only context -- LocalMap
                                   bookkeeping works, every compute entry point returns SO_ICP_E_HIP (no CPU fallback) */
  int32_t rank, world_size;     /* map shard owned by this context (brick-hash of the voxel grid) */
  int32_t max_iterations;       /* LocalizationICPMaxIter (LS.h:273; YAML max_iterations: 5) */
  int32_t lm_max_iterations;    /* ceres max_num_iterations = 4 (LS.cpp:232) */
  int32_t max_surface_features; /* OptSet.max_surface_features (LS.cpp:346-359); < 0: use all points (the reference compares
                                   size_t > int: a negative value never samples); 0 drops every point, like the reference */
  int32_t k;                    /* LocalizationPlaneDistanceNbrNeighbors = 5 (LS.h:277); only 5 supported */
  int32_t tukey_variant;        /* 0 = Ceres 2.0.0 TukeyLoss (rho' = 0.5 (1-s/a^2)^2); 1 = Ceres >= 2.1 */
  int32_t time_kernels;         /* HIP-event timing on the context's stream (so_icp_get_timing): 1 = the k-NN kernel only, on every
                                   3rd registration (events attached to the dispatch; cheap enough for a timed region),
                                   2 = every kernel of every registration (adds pipeline bubbles; profiling only) */
  float line_res, plane_res;    /* localMap.lineRes_/planeRes_ (LM.h:760-761; pushed every frame, lmap.cpp:648-649) */
  double yaw_ratio;             /* OptSet.yaw_ratio (LS.cpp:906) */
  double velocity_failure_threshold; /* OptSet.velocity_failure_threshold (LS.cpp:179) */
  int32_t shard_mode;           /* world_size > 1: SO_ICP_SHARD_MAP (0) = the map sharded by brick-hash of the voxel grid, queries follow
                                   their cell's owner (BASELINE configs[3]); SO_ICP_SHARD_QUERIES (1) = the map replicated on every
                                   rank, the scan's 64-point segments dealt round-robin to the ranks -- equal shares whatever the scene,
                                   no halo, no re-binning; the same 45-double exchange per evaluation in both */
  int32_t solve_workgroups;     /* 0 = one workgroup of the persistent solve launch per compute unit (default); n >= 1: at most n -- the
                                   launch needs ALL its workgroups resident at once, so contexts (or ranks) that share a device must
                                   leave each other room: sum of the values <= compute units.  Results do not depend on it beyond the
                                   order of the fp64 sums (SOICP_SOLVE_WORKGROUPS in the environment overrides it: test aid) */
} so_icp_config;
#define SO_ICP_SHARD_MAP 0
#define SO_ICP_SHARD_QUERIES 1

/* super_odometry_msgs/msg/IterationStats.msg + what LS.cpp:242-251 fills + solver summary */
typedef struct {
  double translation_norm, rotation_norm;
  int32_t num_surf_from_scan;   /* accepted correspondences A (planner_num) */
  int32_t num_corner_from_scan; /* always 0: the edge path is dead */
  int32_t lm_iterations;        /* minimizer iterations executed (iteration 0 not counted) */
  int32_t num_successful_steps; /* ceres::Solver::Summary::num_successful_steps (LS.cpp:141) */
  int32_t termination;          /* 0 max-iter, 1 function tol, 2 parameter tol, 3 gradient tol, 4 no residuals, 5 failure */
  int32_t reserved;
  double initial_cost, final_cost;
  int32_t reject_hist[SO_ICP_N_REJECT]; /* MatchRejectionHistogramPlane, LS.cpp:341 */
  int32_t obs_hist[SO_ICP_N_OBS];       /* PlaneFeatureHistogramObs, LS.cpp:336-339 */
  double pose_after[7];
} so_icp_iter_stats;

/* super_odometry_msgs/msg/OptimizationStats.msg: every field the reference fills (LS.cpp:198-210,
 * 242-251, 371-377, 969-974) + the final normal equations (enables EstimateRegistrationError,
 * LS.cpp:854-889, on the caller side without the A x 6 SVD). */
typedef struct {
  int32_t laser_cloud_surf_from_map_num;
  int32_t laser_cloud_corner_from_map_num; /* 0 */
  int32_t laser_cloud_surf_stack_num;
  int32_t laser_cloud_corner_stack_num;    /* 0 */
  int32_t n_iterations;                    /* outer ICP iterations executed */
  int32_t startup_count;                   /* LidarSLAM::startupCount side effect (LS.cpp:181) */
  int32_t pos_in_localmap[3];              /* LS.cpp:363 */
  int32_t prediction_source;               /* LS.cpp:278: reset to 0 */
  double total_translation, total_rotation;
  double translation_from_last, rotation_from_last;
  double time_elapsed_ms;                  /* TicToc over the ICP loop, LS.cpp:118,199-200 */
  double uncertainty[6];                   /* x y z roll pitch yaw, LS.cpp:915-974 (histogram of the previous scan) */
  double JtJ[36], Jtr[6];                  /* loss-corrected normal equations at the returned pose */
  so_icp_iter_stats iterations[SO_ICP_MAX_OUTER];
  uint32_t flags;                          /* SO_ICP_FLAG_*: degraded / non-default modes this registration ran in -- what the
                                              reference would print as a warning (LS.cpp:113-116 style), for the node's log */
  uint32_t reserved;
} so_icp_stats;

/* so_icp_stats::flags */
#define SO_ICP_FLAG_PER_EVAL_LAUNCHES 0x1u   /* the persistent solve launch is not in use (disabled by a failed co-residency
                                                 wait, SOICP_PERSISTENT=0, a sharded map or a concurrent batch): ~2x slower solve */
#define SO_ICP_FLAG_RETRIED 0x2u             /* THIS registration was repeated after an abandoned persistent solve launch */
#define SO_ICP_FLAG_HOST_MAP 0x4u            /* LocalMap lives on the host and is re-uploaded after every insert (planeRes < 0.1
                                                 or SOICP_HOST_MAP=1) */
#define SO_ICP_FLAG_SORT_BINNING 0x8u        /* reserved (the sort-binning path of ABI v2 rounds 1-2 no longer exists; never set) */
#define SO_ICP_FLAG_SHARDED 0x10u            /* world_size > 1: map shard + collective per evaluation */
#define SO_ICP_FLAG_STAGED_SCAN 0x20u        /* the scan came from so_icp_stage_scan (upload overlapped with earlier work) */
#define SO_ICP_FLAG_COPY_READBACK 0x40u      /* state read back with hipMemcpyAsync (SOICP_READBACK=copy) */
#define SO_ICP_FLAG_QUERY_SPLIT 0x80u        /* world_size > 1 with SO_ICP_SHARD_QUERIES: map replicated, this rank registered its share of the scan */
#define SO_ICP_FLAG_BINNED_AHEAD 0x100u      /* the staged scan had been spatially binned behind its copy, beside the registration before it
                                                 (single device; SOICP_PREBIN=0 disables): this registration started with its k-NN sweep */
#define SO_ICP_FLAG_QUERY_WAVES 0x200u       /* small scan (<= 4 096 kept queries, e.g. max_surface_features 2000 / 4000 of the stock configurations): no
                                                 binning, every kept query searched by a wavefront of its own (SOICP_QUERY_WAVES=0 disables): identical results */

#define SO_ICP_FLAG_CHAINED 0x400u           /* so_icp_register_sequence: this registration's launches were enqueued behind the registration before it, before
                                                 that one had reported; its guess (guesses_out) was formed on the device */

/* average kernel durations since the last so_icp_reset_timing (HIP events on the context's stream) */
typedef struct {
  double knn_ms_total;   int64_t knn_launches;   int64_t knn_queries;   int64_t knn_map_points;
  double eval_ms_total;  int64_t eval_launches;  int64_t eval_points;
  double prep_ms_total;  int64_t prep_launches;
  double host_ms_total;  int64_t registrations;
  /* k-NN kernel statistics (time_kernels only): wave group passes, lanes sent to the exact per-lane scan,
   * wave-uniform candidates streamed */
  int64_t knn_group_passes, knn_fallback_lanes, knn_candidates_scanned;
  /* so_icp_stage_scan: host time registrations spent waiting for a staged scan that was still on its way through the copy
   * thread; announcements copied by DMA straight from registered host memory / through the copy thread / declined */
  double stage_wait_ms_total;  int64_t staged_direct, staged_copied, stage_declined;
  /* packed light chunks of the k-NN sweep (time_kernels 2 + SOICP_ABLATE only): rows of 16 lanes with work, rows left to the
   * group passes because their block has more than 16 x-runs / keeps more than its quarter of the tile, candidates kept */
  int64_t knn_packed_rows, knn_packed_rows_too_many_runs, knn_packed_rows_tile_full, knn_packed_kept;
  /* host adaptivity of the packing (every registration counts): registrations whose sweeps packed their light chunks, and how
   * often a registration that left > 3 % of its queries to the exact scan switched the packing off for the next 32 */
  int64_t knn_pack_registrations, knn_pack_holds;
  /* so_icp_register_sequence: registrations whose launches were enqueued behind the registration before them (SO_ICP_FLAG_CHAINED), and how
   * often a registration needed more outer iterations than were enqueued ahead (the chained registration behind it then started again) */
  int64_t seq_chained, seq_chain_breaks;
} so_icp_timing;

/* -------- lifecycle ------------------------------------------------------------------------ */
void so_icp_default_config(so_icp_config *cfg);
so_icp_ctx *so_icp_create(const so_icp_config *cfg); /* NULL on failure: see so_icp_last_error(NULL) */
void so_icp_destroy(so_icp_ctx *ctx);
const char *so_icp_last_error(const so_icp_ctx *ctx); /* ctx may be NULL for creation errors */
int so_icp_abi_version(void);
/* 1 when a HIP device is usable by this library; the product has NO CPU fallback */
int so_icp_device_available(void);
/* HIP devices visible to this process (one process per GPU: device_id = LOCAL_RANK) */
int so_icp_device_count(void);

/* -------- per-frame knobs (public fields the node writes, lmap.cpp:648-649, 703-711) ---------- */
/* planeRes may change between frames (auto_voxel_size, lmap.cpp:604-649): the resident points stay as they are, the index
 * follows, the next insert re-filters the cubes it touches (LM.h:617-641).  world_size > 1: the shards are cut along the
 * cell grid, which follows planeRes, so a CHANGE of planeRes over a non-empty map re-cuts them from every rank's points --
 * a COLLECTIVE step over the communicator (every rank makes the same call; all-gather of the owned points, a few MB);
 * without a communicator the change is refused with SO_ICP_E_UNSUPPORTED. */
int so_icp_set_resolution(so_icp_ctx *ctx, float line_res, float plane_res);
int so_icp_set_max_surface_features(so_icp_ctx *ctx, int max_surface_features);
int so_icp_set_max_iterations(so_icp_ctx *ctx, int max_iterations);

/* -------- LocalMap (LM.h) ------------------------------------------------------------------ */
int so_icp_map_set_origin(so_icp_ctx *ctx, const double t_w_cur[3], int origin_out[3]); /* LocalMap::setOrigin, LM.h:146-164 */
int so_icp_map_shift(so_icp_ctx *ctx, const double t_w_cur[3], int pos_in_map[3]);      /* LocalMap::shiftMap,  LM.h:169-287 */
/* LocalMap::addSurfPointCloud, LM.h:591-645: world-frame points (stride in bytes, 12 for packed xyz,
 * 32 for pcl::PointXYZI); bins into 50 m cubes, VoxelGrid(planeRes) per touched cube, rebuilds the
 * device index.  Returns the number of points that fell inside the 21x21x11 window, or <0.
 * world_size > 1: every rank passes the SAME cloud and keeps its shard of it; with a communicator (RCCL or in-process
 * group) the call is collective -- the per-block point counts of the full map are summed over the ranks. */
int so_icp_map_add_surf(so_icp_ctx *ctx, const float *xyz, size_t n, size_t stride_bytes);
int so_icp_map_count_5x5(so_icp_ctx *ctx, const int pos[3], int *n_edge, int *n_surf);  /* get5x5LocalMapFeatureSize, LM.h:292-318 */
/* getAllLocalMap / get5x5LocalMap (LM.h:646-688): points in the canonical (device) order */
int so_icp_map_export(so_icp_ctx *ctx, float *xyz, size_t cap_points, size_t *n_out, int only_5x5, const int pos[3]);
/* The same points as RECORDS of stride_bytes (float x, y, z first, the remaining bytes zero -- stride 32: pcl::PointXYZI with intensity 0,
 * what pcl::toROSMsg puts into the laser_cloud_map / laser_cloud_surround messages, lmap.cpp:437-462), written straight to `out`: the
 * payload area of the message being assembled.  One gather, one copy (DMA when `out` lies in so_icp_host_alloc / _register memory), one
 * synchronisation -- so_icp_map_export makes a round trip per occupied cube.  out == NULL (or more points than cap_points): the count only. */
int so_icp_map_export_records(so_icp_ctx *ctx, void *out, size_t stride_bytes, size_t cap_points, size_t *n_out, int only_5x5, const int pos[3]);
int so_icp_map_size(so_icp_ctx *ctx, size_t *n_points, size_t *n_points_this_rank);
int so_icp_map_clear(so_icp_ctx *ctx);
int so_icp_map_get_origin(so_icp_ctx *ctx, int origin_out[3]);
/* Diagnostics of the device-resident map's insert (LM.h:591-645 on the device): how many inserts the device laid out itself
 * -- touched cubes, slots and counts worked out by the insert's first kernel, no read-back -- and how many of those it had
 * to hand back to the host (a cube without a slot, more cubes than a round holds, a cube that needs the sort).  Both 0
 * with the host-side map (world_size > 1) or SOICP_MAP_FAST=0. */
int so_icp_map_insert_stats(so_icp_ctx *ctx, unsigned *device_built, unsigned *handed_back);

/* -------- Seam B: LocalMap::nearestKSearchSurf (LM.h:481-525), batched ------------------------ */
/* Host buffers. found[i]=0 reproduces the `return false` paths (cube outside the window, LM.h:499-502,
 * or no tree, LM.h:506).  Unfilled slots reproduce nanoflann.h:87-100 (idx 0 -> first point of the cube,
 * d2 = 0 except d2[k-1] = FLT_MAX). Exact cube-restricted k-NN; ties by ascending canonical index. */
int so_icp_knn_surf(so_icp_ctx *ctx, const float *q_xyz, size_t nq, int k,
                    float *nbr_xyz /* nq*k*3 */, float *d2 /* nq*k */, int32_t *nbr_index /* nq*k, nullable */,
                    uint8_t *found /* nq */);

/* -------- Seam A: LidarSLAM::performLocalizationAndMapping (LS.cpp:107-152 + 155-210) --------- */
/* scan: sensor frame, already voxel-filtered by the node (lmap.cpp:643-645).  Does shiftMap,
 * the <=50-feature check, the outer ICP loop and the post-processing statistics; does NOT insert
 * the scan into the map (so_icp_localization does).  prev uncertainty comes from the previous call. */
int so_icp_register(so_icp_ctx *ctx, const float *scan_xyz, size_t n, size_t stride_bytes,
                    const double pose_in[7], double pose_out[7], so_icp_stats *stats);
/* A RUN of scans whose guesses chain (a node working off a backlog of feature clouds in localization mode; a log replayed at full
 * speed): guess_0 = pose0, guess_k = T_(k-1) o delta_k -- T_(k-1) the pose registration k-1 ended with, delta_k the motion prediction
 * between the two scans in the sensor frame: what laserMapping::selectPosePrediction (lmap.cpp:345-372) does with the LIO / VIO /
 * IMU predictions, `T_w_lidar = T_w_lidar * prediction` (Twist::operator*, utils/Twist.h:181-185): t = t + R(q) dt, q = normalize(q dq).
 * Equivalent to `count` calls of so_icp_register (so_icp_register_dev with scans_on_device) from guesses_out[k], bit for bit.  What it
 * buys: the launches of registration k+1 are enqueued BEHIND those of registration k before k has reported -- the guess is formed on
 * the device --, so the device does not idle for the host's turn-around between two calls (~11 us of a 140 us registration), and
 * scan k+1 is copied and binned beside registration k under the predicted guess.  A registration that turns out to need more outer
 * iterations than were enqueued ahead is finished the ordinary way and the one behind it starts again (so_icp_timing::seq_chain_breaks).
 * The map is NOT updated in between (localization mode, LS.cpp:60-80 `if (!localization_mode)`): use so_icp_localization per scan when it must be.
 * scans[k]: host pointers (packed or strided xyz; memory from so_icp_host_alloc / so_icp_host_register is copied by DMA) or, with
 * scans_on_device, device pointers to packed xyz.  deltas: count x 7 {dx, dy, dz, qx, qy, qz, qw}, row 0 unused.  poses_out: count x 7;
 * guesses_out (count x 7, nullable): the guess each registration started from; stats (count, nullable); n_done (nullable): registrations
 * completed -- on an error or SO_ICP_NOT_ENOUGH_MAP_FEATURES the run stops at that scan and its code is returned.
 * The chained path needs a single device, the device-resident map and yaw_ratio 0 (every stock configuration: MannualYawCorrection,
 * LS.cpp:891-913, is then the identity up to rounding, and the chain continues from the optimised pose itself,
 * stats[k].iterations[last].pose_after); otherwise -- or with SOICP_SEQ_CHAIN=0 -- the registrations run one after the other, same results. */
int so_icp_register_sequence(so_icp_ctx *ctx, int count, const void *const *scans, const size_t *n_points, size_t stride_bytes,
                             int scans_on_device, const double pose0[7], const double *deltas, double *poses_out, double *guesses_out,
                             so_icp_stats *stats, int *n_done);
/* A backlog worked off in several calls: name the scan that will START the next so_icp_register_sequence call (packed xyz in host memory,
 * n points) and the motion prediction from the LAST scan of the coming call to it.  That call then copies and bins it beside its last
 * registration, and the call after it -- whose scans[0] is this very buffer -- starts with its first sweep instead of a copy and a binning
 * nothing hides (55 us for a 131 072-point scan).  The buffer must stay valid and unchanged until that second call returns; scan == NULL
 * withdraws the announcement (and waits for a copy already under way).  Results never depend on it. */
int so_icp_sequence_announce_next(so_icp_ctx *ctx, const float *scan_xyz, size_t n, const double delta[7]);
/* Announce the NEXT scan: the host buffer travels to HBM on the context's copy stream while the caller goes on (typically:
 * while the previous so_icp_register is still running -- the node's feature callback, lmap.cpp:21-25, has the cloud long
 * before process() reaches it).  Packed xyz (stride 12) inside a buffer pinned with so_icp_host_register is read by DMA
 * straight from the caller's memory: this call enqueues the copy and returns, and the registration's first kernel waits for
 * it on the device.  Anything else (pageable memory, strided pcl::PointXYZI records) is packed into a pinned buffer by the
 * context's copy thread first.  A following so_icp_register / so_icp_localization with the SAME (scan_xyz, n, stride_bytes)
 * consumes the staged copy instead of uploading again (so_icp_stats::flags carries SO_ICP_FLAG_STAGED_SCAN); any other call
 * simply ignores it.  The caller's buffer must stay valid and unchanged until that call returns (a DMA copy reads it when the
 * registration in flight has its first launch in the queue, or 300 us after the announcement, whichever comes first); to take a
 * staged buffer back without registering it, announce it again (the older copy is superseded) or call so_icp_stage_cancel.
 * A DMA copy that a registration in flight enqueues is followed, on the copy stream, by the scan's spatial binning under that
 * registration's guess (single device, device-resident map): the consuming registration then starts with its k-NN sweep
 * (SO_ICP_FLAG_BINNED_AHEAD).  The binning only groups queries for the sweep -- results do not depend on it.
 * Three slots: two scans may be announced ahead of the one being registered; a further announcement returns
 * SO_ICP_STAGE_DECLINED (soft: that scan is uploaded by its own registration call).  Announcing a buffer again supersedes
 * its earlier copy; a copy announced BEFORE a scan that has been consumed since (a skipped frame) is never served and gives
 * its slot to the next announcement.  A staged copy is also dropped by a map-seeding so_icp_localization(initialization = 0)
 * on the same buffer.  Unlike the other entry points this one may be called from ANOTHER thread than the registration calls
 * (the node's feature callback, lmap.cpp:21-25). */
int so_icp_stage_scan(so_icp_ctx *ctx, const float *scan_xyz, size_t n, size_t stride_bytes);
/* withdraw every staged copy of this host buffer (returns when nothing reads the buffer any more) */
int so_icp_stage_cancel(so_icp_ctx *ctx, const float *scan_xyz);
/* Pin a host buffer the caller keeps its clouds in (hipHostRegister; the node's copy of the message payload into its feature
 * cloud, pcl::fromROSMsg at lmap.cpp:250-263, then lands in pinned memory): so_icp_stage_scan and so_icp_register copy
 * packed scans that lie inside it by DMA without an intermediate copy.  Unregister before freeing the buffer. */
int so_icp_host_register(so_icp_ctx *ctx, const void *ptr, size_t bytes);
int so_icp_host_unregister(so_icp_ctx *ctx, const void *ptr);
/* Pinned host memory from the HIP runtime's allocator (hipHostMalloc), owned by the context until so_icp_host_free /
 * so_icp_destroy: a pool for the caller's clouds without a registration call per buffer. */
int so_icp_host_alloc(so_icp_ctx *ctx, size_t bytes, void **ptr_out);
int so_icp_host_free(so_icp_ctx *ctx, void *ptr);
/* same, scan already resident in HBM as packed float xyz (n*3 floats, device pointer) */
int so_icp_register_dev(so_icp_ctx *ctx, const void *d_scan_xyz, size_t n,
                        const double pose_in[7], double pose_out[7], so_icp_stats *stats);
/* Batched hypotheses (BASELINE.json configs[4], the "degeneracy / alignment-risk" use): the SAME scan registered from
 * n_hyp initial poses (poses_in = n_hyp x 7).  Every hypothesis is an independent so_icp_register (own binning, own
 * correspondences, own LM solves); the map window is shifted once, for hypothesis 0 (LS.cpp:363), and the context's
 * scan-to-scan state (previous observability histogram, startup counter) is left as it was.  Groups of up to 64
 * hypotheses go through batched kernels -- one binning / k-NN / persistent-solve launch per outer iteration over all
 * hypotheses still running, a group of workgroups and an LM controller per hypothesis -- whose sums follow the summation
 * tree of the single registration: results are bit-identical to one so_icp_register per hypothesis.  A device that cannot
 * keep the batched solve resident falls back by itself (one workgroup per compute unit, then concurrent sequential
 * registrations on worker contexts that borrow this context's resident map); same bits.  d_scan_xyz != NULL
 * takes a scan already resident in HBM (so_icp_upload_scan), else scan_xyz is uploaded once.  Returns the number of
 * hypotheses that returned SO_ICP_OK (>= 0), or a negative error; rc_out[h] (nullable) = status of hypothesis h. */
int so_icp_register_batch(so_icp_ctx *ctx, const float *scan_xyz, const void *d_scan_xyz, size_t n, size_t stride_bytes,
                          const double *poses_in, int n_hyp, double *poses_out, so_icp_stats *stats /* n_hyp, nullable */,
                          int32_t *rc_out /* n_hyp, nullable */);

/* LidarSLAM::EstimateRegistrationError (LS.cpp:854-889) from the final normal equations in so_icp_stats, without the
 * A x 6 SVD: covariance in the tangent space = (J^T J)^-1 with the loss applied (ceres::Covariance,
 * apply_loss_function = true, GetCovarianceBlockInTangentSpace), then the eigen-analysis of its position and
 * orientation blocks (LS.h:127-151).  Host arithmetic only; returns SO_ICP_E_INVALID when J^T J is singular. */
typedef struct {
  double covariance[36];               /* row-major, DoF order X Y Z rX rY rZ */
  double position_error;               /* sqrt(largest eigenvalue of the position block) */
  double position_error_direction[3];
  double pos_inverse_condition_num;    /* sqrt(lambda_min) / sqrt(lambda_max) */
  double orientation_error_deg;        /* Rad2Deg(sqrt(largest eigenvalue of the orientation block)) */
  double orientation_error_direction[3];
  double ori_inverse_condition_num;
} so_icp_registration_error_t;
int so_icp_registration_error(const so_icp_stats *stats, so_icp_registration_error_t *out);

/* upload a scan once and keep it resident in HBM (device pointer owned by the context until
 * so_icp_free_scan / so_icp_destroy) */
int so_icp_upload_scan(so_icp_ctx *ctx, const float *scan_xyz, size_t n, size_t stride_bytes, void **d_scan_out);
int so_icp_free_scan(so_icp_ctx *ctx, void *d_scan);

/* LidarSLAM::Localization (LS.cpp:30-51): initialization==0 seeds the map (LS.cpp:83-94) and returns
 * SO_ICP_MAP_SEEDED; otherwise registers, applies the post-processing and inserts the scan
 * (transformAndAddToMap, LS.cpp:60-80, 163-167). */
int so_icp_localization(so_icp_ctx *ctx, int initialization, const double T_w_lidar[7],
                        const float *planar_xyz, size_t n, size_t stride_bytes, double time_laser_odometry,
                        double pose_out[7], so_icp_stats *stats);

/* same, scan already resident in HBM as packed float xyz (e.g. the output of so_icp_prefilter_scan) */
int so_icp_localization_dev(so_icp_ctx *ctx, int initialization, const double T_w_lidar[7], const void *d_scan_xyz, size_t n,
                            double time_laser_odometry, double pose_out[7], so_icp_stats *stats);
/* copy a resident scan (packed float xyz) back to the host */
int so_icp_download_scan(so_icp_ctx *ctx, const void *d_scan_xyz, size_t n, float *out_xyz);

/* -------- the step before Seam A: laserMapping::adjustVoxelSize (lmap.cpp:598-651) on the device ----------------
 * auto_voxel_size != 0: average_distance = mean|x| * mean|y| * mean|z| of the surf cloud chooses the resolutions
 * (< 25: 0.1 / 0.2, > 65: 0.4 / 0.8, else unchanged), count_far_points (> 3 m) > 3000 raises increase_blind_radius.
 * The choice is ALWAYS the reference's: it accumulates the three sums in float in input order (lmap.cpp:604-611), the device
 * in an fp64 tree; when the statistic is within the reach of those roundings of a threshold, the reference's accumulation
 * itself is run (so_icp_prefilter_info::statistic_in_input_order).
 * Then pcl::VoxelGrid (leaf = planeRes: float leaf coordinates, centroids accumulated in float in input order, output in
 * ascending leaf index; "leaf size too small" passes the cloud through) and localMap.lineRes_/planeRes_ = the result
 * (lmap.cpp:648-649).  *d_filtered_out (packed float xyz, owned by the context, valid until the next call) feeds
 * so_icp_register_dev / so_icp_localization_dev without a host round trip.
 * One enqueue, one read-back: the statistics are reduced, the resolution chosen and the leaf grid laid out on the device
 * (only a statistic inside the rounding band of a threshold, or PCL's "leaf size too small" pass-through, goes through the
 * host); the call runs on a queue of its own, beside the map insert the previous so_icp_localization left in the
 * context's queue, and returns when the filtered cloud is complete. */
typedef struct {
  double average_distance;
  int32_t count_far_points, increase_blind_radius;
  float line_res, plane_res;  /* resolutions in effect after the call */
  int32_t statistic_in_input_order; /* 1: average_distance is the reference's own float accumulation in input order, bit for bit
                                       (taken whenever the value is close enough to 25 / 65 for its roundings to decide);
                                       0: mean|x| mean|y| mean|z| from exact sums -- within 3 n 2^-24 of it, same decision */
  int32_t reserved;                 /* 1: the cloud had been announced (so_icp_prefilter_announce): no copy on this call's path */
} so_icp_prefilter_info;
int so_icp_prefilter_scan(so_icp_ctx *ctx, const float *surf_xyz, size_t n, size_t stride_bytes, int auto_voxel_size,
                          float line_res, float plane_res, void **d_filtered_out, size_t *n_out, so_icp_prefilter_info *info);
/* Announce the NEXT raw surf cloud (laserFeatureInfoHandler has it long before process() reaches it, lmap.cpp:250-263, 768-793): its H2D copy
 * (34 us for a 131 072-point sweep over PCIe) goes into the pre-filter's queue now, beside the registration of the frame before, and the
 * so_icp_prefilter_scan call that names the same buffer (pointer, n, stride) starts with its first kernel; so_icp_prefilter_info::reserved
 * says 1 when that happened.  The buffer must stay valid and unchanged until that call returns; xyz == NULL withdraws the announcement
 * (and waits for a copy under way).  One announcement at a time (a second one replaces the first); any thread.  Results never depend on it. */
int so_icp_prefilter_announce(so_icp_ctx *ctx, const float *surf_xyz, size_t n, size_t stride_bytes);

/* -------- two steps before Seam A: featureExtraction::removePointDistortion
 * (src/FeatureExtraction/featureExtraction.cpp:223-314) on the device.  Every finite point of the sweep is moved to the
 * sensor frame of the sweep start: pose buffer (IMU orientations or VIO odometry, strictly increasing times: the
 * reference keeps them in a std::map) interpolated at lidar_start_time + point.time (slerp / lerp between the two
 * neighbours, the first pose before the buffer starts), T_final = T_w_original^-1 * T_w_current, wrapped in
 * T_l_i * . * T_i_l when the buffer is the IMU's (then positions are ignored: :231-235).  Records: float x y z at byte
 * 0 4 8, float time at time_offset_bytes (point_os::PointcloudXYZITR: stride 32, time at 20); rewritten in place.
 * A point stamped at or after the last pose has no successor in the buffer (undefined in the reference, which only
 * de-skews once a later measurement has arrived, :185-201): it gets the last pose and is counted in n_clamped. */
typedef struct {
  double time;
  double pos[3];
  double rot[4]; /* x y z w */
} so_icp_stamped_pose;
typedef struct {
  double q_w_original_l[4]; /* x y z w: orientation of the sweep-start sensor frame (featureExtraction.cpp:289) */
  double t_w_original_l[3]; /* :290 */
  uint32_t n_clamped, reserved;
} so_icp_deskew_info;
int so_icp_deskew_scan(so_icp_ctx *ctx, void *points /* host, rewritten in place */, size_t n, size_t stride_bytes,
                       size_t time_offset_bytes, double lidar_start_time, const so_icp_stamped_pose *poses, size_t n_poses,
                       int poses_are_imu, const double T_i_l[7] /* tx ty tz qx qy qz qw; NULL = identity */, so_icp_deskew_info *info);
/* same on records already resident in HBM (rewritten there) */
int so_icp_deskew_scan_dev(so_icp_ctx *ctx, void *d_points, size_t n, size_t stride_bytes, size_t time_offset_bytes,
                           double lidar_start_time, const so_icp_stamped_pose *poses, size_t n_poses, int poses_are_imu,
                           const double T_i_l[7], so_icp_deskew_info *info);

/* -------- one step after Seam A: the registered scan laserMapping::publishTopic builds
 * (src/LaserMapping/laserMapping.cpp:464-493 with utils::pointAssociateToMap, src/utils/superodom_utils.cpp:148-158).
 * Records with float x y z at byte 0 4 8, rewritten in place: a point within 0.1 m of the sensor (x*x + y*y + z*z < 0.01 in
 * float) stays as it is, every other point becomes q * p + t (fp64, rounded to float).  keep[i] (nullable) = 1 when the
 * result lies farther than 0.1 m from the world origin -- the node publishes only those; *n_kept (nullable) = their number. */
int so_icp_transform_cloud(so_icp_ctx *ctx, void *points /* host, rewritten in place */, size_t n, size_t stride_bytes,
                           const double T_w_lidar[7], uint8_t *keep, size_t *n_kept);

/* -------- multi-GPU: one process per GPU; the map is sharded by brick-hash of the voxel grid and the 45 fp64 sums of every
 * evaluation are summed over the ranks: by RCCL (below), by an in-process group, or by the solve launches themselves
 * (peer exchange, further below) -------------------------------------------------------------------------------------- */
#define SO_ICP_UNIQUE_ID_BYTES 128
int so_icp_comm_unique_id(uint8_t id[SO_ICP_UNIQUE_ID_BYTES]);                 /* rank 0: ncclGetUniqueId */
int so_icp_comm_init(so_icp_ctx *ctx, const uint8_t id[SO_ICP_UNIQUE_ID_BYTES]); /* all ranks: ncclCommInitRank on ctx->rank/world_size */
/* The same exchange without RCCL, for the shard contexts of ONE process (one thread + one context per GPU of the node, or
 * several shard contexts on one GPU in a test): the contexts that pass the same key form a group of cfg.world_size members
 * and sum their records through host memory in rank order.  Every member must run the same registrations, each from its
 * own thread (a member waits inside so_icp_register until all members have contributed).  The streams of one process share its
 * hardware queues (HIP: GPU_MAX_HW_QUEUES, 4 by default): with the peer exchange enabled, where every member's solve launch
 * must be running at the same time, keep the number of members per DEVICE at or below that. */
int so_icp_comm_init_inprocess(so_icp_ctx *ctx, uint64_t group_key);
/* Peer exchange: the per-evaluation collective replaced by the ranks' persistent solve launches trading their 45-double
 * records themselves -- tagged 16-byte chunks pushed into every rank's inbox (device memory mapped across processes with
 * hipIpc, across the contexts of one process by pointer) with system-scope stores over xGMI, polled by the receiving
 * launch.  One launch per outer iteration survives N > 1 (with a collective per evaluation the solve falls apart into
 * ~25 launches + collectives per registration).  Transport-agnostic like the unique id above:
 *   1. every rank: so_icp_peer_export -> handle;  2. exchange the handles (rank order) by any means;
 *   3. every rank, at about the same time: so_icp_peer_connect (maps the inboxes and runs a self-test with the very
 *      stores / loads of the exchange; *self_test_ok = 0 when a mapping or a chunk is missing: not an error);
 *   4. agree on the minimum of the self-test results and call so_icp_peer_enable(agreed) on every rank -- with 0 the
 *      context keeps the collective path (RCCL / in-process group).  The map-count collective of a map insert still needs
 *      one of those.  While enabled, a registration that loses a rank returns SO_ICP_E_HIP (no silent fall-back: the
 *      decision to leave the peer path would have to be collective). */
#define SO_ICP_PEER_HANDLE_BYTES 80
int so_icp_peer_export(so_icp_ctx *ctx, uint8_t handle[SO_ICP_PEER_HANDLE_BYTES]);
int so_icp_peer_connect(so_icp_ctx *ctx, const uint8_t *handles /* world_size x SO_ICP_PEER_HANDLE_BYTES, rank order */, int *self_test_ok);
int so_icp_peer_enable(so_icp_ctx *ctx, int on);
/* brick-hash ownership of the shard (host logic, testable without a GPU) */
int so_icp_shard_owner_of_point(const float p[3], const int origin[3], float plane_res, int world_size);
int so_icp_cells_per_cube(float plane_res, double *cell_size);
/* how a scan's queries fall to the ranks under `pose` (the ownership rule of the sharded registration; host arithmetic):
 * counts[r] = queries whose map cell rank r owns (queries outside the window count for rank 0, like in the kernels) */
int so_icp_shard_histogram(const float *scan_xyz, size_t n, size_t stride_bytes, const double pose[7], const int origin[3],
                           float plane_res, int world_size, int64_t *counts /* world_size */);

/* -------- host-side Levenberg-Marquardt state machine (the Ceres restatement the device path is
 * driven by; exposed so it can be tested without a GPU) ------------------------------------------ */
typedef struct {
  double cost;      /* 0.5 * sum c_i rho(s_i) */
  double count;     /* accepted correspondences */
  double Jtr[6];
  double JtJ[21];   /* upper triangle, row-major: (0,0..5),(1,1..5),... */
  double hist[16];  /* reject_hist[7] then obs_hist[9] (carried through the same all-reduce) */
} so_icp_sums;      /* 45 doubles */
typedef struct { double opaque[96]; } so_icp_lm_state;
/* begin: sums evaluated at x0. returns 1 and writes next_pose when another evaluation is needed, 0 when done */
int so_icp_lm_begin(so_icp_lm_state *s, const double x0[7], const so_icp_sums *sums_at_x0, int max_iterations, double next_pose[7]);
int so_icp_lm_feed(so_icp_lm_state *s, const so_icp_sums *sums_at_next, double next_pose[7]);
int so_icp_lm_result(const so_icp_lm_state *s, double pose[7], so_icp_iter_stats *stats /* solver fields only */);

/* -------- measurement ------------------------------------------------------------------------ */
int so_icp_get_timing(so_icp_ctx *ctx, so_icp_timing *t);
int so_icp_reset_timing(so_icp_ctx *ctx);
/* switch so_icp_config::time_kernels on a live context (e.g. 1 inside a timed region, 2 for a profiling pass after it) */
int so_icp_set_time_kernels(so_icp_ctx *ctx, int mode);
int so_icp_synchronize(so_icp_ctx *ctx);
/* test aid: MatchingResult (LS.h:85-94) of every query of the last registration's LAST outer iteration, indexed like the scan
 * (254 = not sampled / not owned by this rank) */
int so_icp_debug_match_status(so_icp_ctx *ctx, uint8_t *out, size_t n);
/* test aid: the five neighbours (canonical map indices, nearest first) the LAST k-NN sweep of the last registration left for every query, indexed
 * like the scan: out[5 * n].  Meaningful where that sweep found five neighbours inside the gate (every query the fit pass then judged: status
 * 0, 3, 4, 5); elsewhere the entries are whatever an earlier sweep left */
int so_icp_debug_neighbours(so_icp_ctx *ctx, uint32_t *out, size_t n);
/* profiling aid: wall-clock stamps (100 MHz ticks) of the phases of the last fit / evaluation kernels (SOICP_ABLATE=128) */
int so_icp_debug_stamps(so_icp_ctx *ctx, uint64_t out[16]);
/* profiling aid: per-workgroup records (16 words each, 2 sweeps x workgroups) of the k-NN kernel's phases (SOICP_ABLATE=128);
 * call with out == NULL to query the size in 64-bit words */
int so_icp_debug_knn_stamps(so_icp_ctx *ctx, uint64_t *out, size_t capacity_words, size_t *n_words);

#ifdef __cplusplus
}
#endif
#endif /* SO_ICP_H */
