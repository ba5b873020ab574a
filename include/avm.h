/*
 * avm.h — C ABI of the MI355X-native hot paths of Anticipated-VINS-Mono.
 *
 * The reference has no FFI/plugin interface: both hot paths are C++ member
 * functions on stateful objects called from one thread
 * (vins_estimator/src/estimator_node.cpp:340,360).  This header is the boundary
 * a maintainer would bind instead of
 *
 *   HP-A  void Estimator::optimization()                       vins_estimator/src/estimator.h:47
 *                                                              (body estimator.cpp:661-994)
 *   HP-B  std::pair<std::vector<int>,std::vector<int>>
 *         FeatureSelector::select(image_t&, const Header&, int) vins_estimator/src/feature_selector.h:49-50
 *                                                              (body feature_selector.cpp:74-202)
 *
 * Conventions
 *   - plain pointers + sizes, FP64 everywhere, row-major, no exceptions cross
 *     the ABI; every function returns AVM_OK (0) or a negative avm_status.
 *   - the index tables of a batch (feature tracks, IMU sample counts, prior block
 *     tables, the selector's counts) are validated before any kernel indexes with
 *     them - host tables on the host, device-resident ones by a one-thread-per-window
 *     kernel: AVM_ERR_INVALID, avm_last_error() names the first bad window and the
 *     rule, nothing has been modified.  (avm_window_solve_batch on device-resident
 *     tables reads that kernel's verdict while the pre-integration - which writes
 *     library-owned buffers only and clamps the one entry it indexes with - is already
 *     running; nothing else is launched before the verdict is in.)
 *   - all buffers are caller owned.  `mem` says whether the pointers are host
 *     pointers (the library stages them over PCIe) or device pointers already
 *     resident in HBM (what bench.py times).
 *   - one avm_ctx per host thread (re-entrant, no statics — the reference's
 *     function-local statics at feature_selector.cpp:85,383,424 are NOT
 *     reproduced); the ctx owns device scratch + one HIP stream.
 *   - stream ordering of AVM_MEM_DEVICE buffers: the ctx stream is a BLOCKING stream,
 *     i.e. it is ordered after everything submitted to the legacy default (NULL) stream
 *     before the call (PyTorch's default current stream, plain hipMemcpy), and every
 *     entry point returns only after its own work has finished.  A caller that fills
 *     device buffers on some other non-blocking stream must order that stream itself
 *     before calling in (synchronize it, or record an event and hipStreamWaitEvent on
 *     avm_ctx_stream()).
 *   - quaternions are stored (x,y,z,w) exactly like para_Pose
 *     (estimator.cpp:484-488).
 *
 * Struct layouts in this header are shared verbatim by the CPU oracle
 * (oracle/, test infrastructure only) so the same buffers can be handed to both.
 */
#ifndef AVM_H_
#define AVM_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AVM_WINDOW_SIZE 10               /* parameters.h:14  */
#define AVM_NFRAMES (AVM_WINDOW_SIZE + 1)
#define AVM_SIZE_POSE 7                  /* parameters.h:53  */
#define AVM_SIZE_SPEEDBIAS 9             /* parameters.h:54  */
#define AVM_MAX_ITER_TRACE 16

typedef enum avm_status {
  AVM_OK = 0,
  AVM_ERR_INVALID = -1,     /* bad argument / size */
  AVM_ERR_UNSUPPORTED = -2, /* not available on this host: a selector HORIZON without a kernel instance (2,3,5,10,13 are built),
                               or librccl.so.1 cannot be loaded for avm_comm_* */
  AVM_ERR_NO_DEVICE = -3,   /* no HIP device: there is NO CPU fallback */
  AVM_ERR_HIP = -4,         /* HIP runtime error, see avm_last_error() */
  AVM_ERR_CAPACITY = -5     /* problem larger than avm_config limits */
} avm_status;

typedef enum avm_mem { AVM_MEM_HOST = 0, AVM_MEM_DEVICE = 1 } avm_mem;

/* prior block kinds (replaces the pointer-keyed addr_shift map, estimator.cpp:904-916) */
enum { AVM_BLK_POSE = 0, AVM_BLK_SPEEDBIAS = 1, AVM_BLK_EXPOSE = 2, AVM_BLK_TD = 3 /* para_Td, 1 value */ };

/* marginalization_flag, estimator.h:33-37 */
enum { AVM_MARGIN_OLD = 0, AVM_MARGIN_SECOND_NEW = 1, AVM_MARGIN_NONE = 2 };

/* termination, mirrors ceres::TerminationType as far as this path can reach it */
enum {
  AVM_TERM_NO_CONVERGENCE = 0, /* max_num_iterations reached */
  AVM_TERM_GRADIENT_TOL = 1,
  AVM_TERM_PARAMETER_TOL = 2,
  AVM_TERM_FUNCTION_TOL = 3,
  AVM_TERM_MIN_RADIUS = 4,
  AVM_TERM_FAILURE = 5         /* too many invalid steps / linear solver failure */
};

/* Solver + model options.  Defaults (avm_default_options) are the values the
 * reference runs with: estimator.cpp:794-806, config/euroc/euroc_config.yaml:54-63,
 * parameters.cpp:11, estimator.cpp:17, and the Ceres defaults listed in SURVEY.md §5.9. */
typedef struct avm_options {
  int32_t max_num_iterations;        /* NUM_ITERATIONS = 8 (the wall-clock cap: max_solver_time_s below) */
  int32_t estimate_extrinsic;        /* 0: ex_pose constant (SetParameterBlockConstant, estimator.cpp:677-681); != 0: a variable of the solve */
  int32_t estimate_td;               /* != 0: every vision factor is a ProjectionTdFactor and para_Td is a variable (estimator.cpp:684-688,732-747) */
  int32_t marginalization_flag;      /* AVM_MARGIN_* */
  double focal_length;               /* FOCAL_LENGTH 460; sqrt_info = focal/1.5 * I2 (estimator.cpp:17) */
  double g[3];                       /* G = (0,0,g_norm) parameters.cpp:11,127 */
  double acc_n, gyr_n, acc_w, gyr_w; /* integration_base.h:21-27 */
  double cauchy_a;                   /* CauchyLoss(1.0) estimator.cpp:666 */
  double max_sum_dt;                 /* 10.0: IMU factor skipped above (estimator.cpp:705) */
  /* Ceres trust-region defaults (not overridden by the reference) */
  double initial_trust_region_radius; /* 1e4  */
  double max_trust_region_radius;     /* 1e16 */
  double min_trust_region_radius;     /* 1e-32 */
  double min_relative_decrease;       /* 1e-3 */
  double function_tolerance;          /* 1e-6 */
  double gradient_tolerance;          /* 1e-10 */
  double parameter_tolerance;         /* 1e-8 */
  double min_lm_diagonal;             /* 1e-6 */
  double max_lm_diagonal;             /* 1e32 */
  int32_t max_num_consecutive_invalid_steps; /* 5 */
  int32_t jacobi_scaling;                    /* 1 */
  double marg_eps;                           /* 1e-8 marginalization_factor.h:70 */
  double tr, row;                            /* TR (rolling-shutter read-out time, 0 for a global shutter) and ROW (image height), parameters.cpp:
                                                only read by the td factor (projection_td_factor.cpp:19-20,50-52) */
  double max_solver_time_s;                  /* options.max_solver_time_in_seconds (estimator.cpp:803-806: SOLVER_TIME, x 4/5 under MARGIN_OLD;
                                                config/euroc/euroc_config.yaml:54).  0 (the default) = no cap, which is what the parity
                                                tests and the bench run with.  > 0: checked like Ceres' MaxSolverTimeReached at the top of
                                                every iteration, before the iteration limit, against a device wall clock that starts when the
                                                window's solve starts on the GPU (staging and pre-integration are not counted); the
                                                minimizer then stops at the current point with AVM_TERM_NO_CONVERGENCE.  Non-finite values
                                                and anything above 1e9 s mean "no cap".  Both forms of the solve kernel check it (round 5:
                                                a batch larger than the CU count keeps the throughput form under a cap) */
  double marg_noise_rel;                     /* The eigenvalue clamp of marginalization_factor.cpp:284-285 keeps S > marg_eps.  With
                                                marg_noise_rel > 0 (default 1e-18) an eigenvalue is kept only if it ALSO exceeds the
                                                rounding noise of the variables its eigenvector lives on,
                                                S^2 > marg_noise_rel * v^T diag(s) v, s_i the magnitude the diagonal entry A'_ii was
                                                formed at: the exact zeros of a rank-deficient A' (the first marginalization of a run,
                                                ragged tracks) are dropped as exact arithmetic would drop them, where the reference's own
                                                FP64 run keeps some of them as rounding noise.  The default is
                                                set so that NO genuine direction is dropped: along eight 20-frame streams the states
                                                stay as close to the exact-prior stream as with the literal clamp (round 5; 1e-16, the
                                                default of rounds 2-4, dropped weak genuine directions on 11 of 160 frames).  0 = the
                                                reference-literal clamp, nothing but S > marg_eps (AVM_PRIOR_LITERAL=1 does the same and
                                                also forces the eigen-decomposition path) */
} avm_options;

/* A batch of independent sliding windows, struct-of-arrays over the window index.
 * [B] = n_windows.  Strides are the max_* fields so that one batch can hold
 * ragged windows.  Maps the inputs of Estimator::optimization() (SURVEY §8 A1,A14). */
typedef struct avm_window_batch {
  int32_t n_windows;
  int32_t max_feat;  /* stride of per-feature arrays (>= max n_feat) */
  int32_t max_obs;   /* stride of per-observation arrays (>= max total observations) */
  int32_t max_samp;  /* stride of IMU sample arrays (>= max samples per interval) */
  int32_t max_prior; /* leading dimension of prior_J / prior_r (>= max prior_n) */
  int32_t max_pblk;  /* stride of prior block tables */

  /* ---- state, in/out: para_Pose / para_SpeedBias / para_Ex_Pose / para_Feature (estimator.h:109-115) */
  double* pose;      /* [B][11][7]  x y z qx qy qz qw */
  double* speedbias; /* [B][11][9]  v ba bg */
  double* ex_pose;   /* [B][7]      tic, qic */
  double* inv_depth; /* [B][max_feat]  1/estimated_depth, feature_manager.cpp:184-200 */

  /* ---- feature tracks (already filtered by used_num>=2 && start_frame<WINDOW_SIZE-2, estimator.cpp:715) */
  const int32_t* n_feat;         /* [B] */
  const int32_t* feat_start;     /* [B][max_feat] start_frame (imu_i); must be non-decreasing in e (std::list order) */
  const int32_t* feat_nobs;      /* [B][max_feat] feature_per_frame.size(); frames start..start+nobs-1 */
  const int32_t* feat_obs_begin; /* [B][max_feat] offset of the first observation in obs_xy */
  const double* obs_xy;          /* [B][max_obs][2] normalized-plane point.x, point.y (z == 1) */

  /* ---- IMU raw samples per interval j=0..9 (pre_integrations[j+1], between frames j and j+1) */
  const int32_t* imu_n; /* [B][10] number of push_back() samples */
  const double* imu_dt; /* [B][10][max_samp] */
  const double* imu_acc; /* [B][10][max_samp+1][3]; row 0 = acc_0 of the constructor (integration_base.h:13) */
  const double* imu_gyr; /* [B][10][max_samp+1][3] */
  const double* imu_lin_ba; /* [B][10][3] linearized_ba */
  const double* imu_lin_bg; /* [B][10][3] linearized_bg */

  /* ---- marginalization prior in (last_marginalization_info), prior_n == 0: none */
  const int32_t* prior_n;         /* [B] residual dimension n */
  const int32_t* prior_nblk;      /* [B] number of kept blocks */
  const int32_t* prior_blk_kind;  /* [B][max_pblk] AVM_BLK_* */
  const int32_t* prior_blk_frame; /* [B][max_pblk] frame index the block is applied to (already addr_shift'ed) */
  const double* prior_J;          /* [B][max_prior][max_prior] linearized_jacobians (n x n used) */
  const double* prior_r;          /* [B][max_prior] linearized_residuals */
  const double* prior_x0;         /* [B][max_pblk][9] keep_block_data (pose uses 7, speedbias 9, td 1) */

  /* ---- optional members of the problem; NULL = absent -------------------------------------------------------- */
  /* time offset (opt->estimate_td != 0): per OBSERVATION slot FeaturePerFrame::velocity.x, .y, cur_td, uv.y()
   * (estimator.cpp:734-736) and para_Td */
  const double* obs_vel_td;      /* [B][max_obs][4] */
  double* td;                    /* [B] in/out */
  /* relocalization (estimator.cpp:760-792, 588-604): these five arrays present <=> relocalization_info.  The loop over
   * f_manager.feature / match_points (ids ascending) is resolved by the host into feature INDICES of this batch
   * (ascending, only features with start_frame <= relo_frame_local_index).  relo_n[w] == 0 (no feature matched): no factor
   * references relo_Pose and the solve leaves it alone, but it still comes back through double2vector's gauge fix
   * (relo_t / relo_r of :590-596), as in the reference; a window without relocalization simply ignores its relo_pose. */
  const int32_t* relo_n;         /* [B] number of matched features */
  const int32_t* relo_frame;     /* [B] relo_frame_local_index (only needed by the host for the outputs of :598-604) */
  const int32_t* relo_feat;      /* [B][max_feat] feature index of match k */
  const double* relo_xy;         /* [B][max_feat][2] match_points[k].x, .y */
  double* relo_pose;             /* [B][7] relo_Pose, in: as set by setReloFrame (:1135-1141); out: relo_t | quaternion of
                                    relo_r of double2vector (:590-596), i.e. the optimized loop-frame pose after the gauge fix */
  /* failure_occur re-anchoring of double2vector (estimator.cpp:526-531) */
  const int32_t* failure_occur;  /* [B] != 0: origin_R0 / origin_P0 come from last_pose0 instead of pose[0] */
  const double* last_pose0;      /* [B][7] last_P0, last_R0 as a quaternion x y z w */
} avm_window_batch;

/* new prior produced by the post-solve marginalization (estimator.cpp:817-990) */
typedef struct avm_prior_out {
  int32_t max_prior, max_pblk;
  int32_t* n;         /* [B] */
  int32_t* nblk;      /* [B] */
  int32_t* blk_kind;  /* [B][max_pblk] */
  int32_t* blk_frame; /* [B][max_pblk] frame index AFTER addr_shift */
  double* J;          /* [B][max_prior][max_prior] */
  double* r;          /* [B][max_prior] */
  double* x0;         /* [B][max_pblk][9] */
} avm_prior_out;

/* per-window solve report (subset of ceres::Solver::Summary) */
typedef struct avm_solve_summary {
  int32_t termination;      /* AVM_TERM_* */
  int32_t num_iterations;   /* step attempts (iteration 0 not counted) */
  int32_t num_successful;   /* accepted steps */
  int32_t accept_mask;      /* bit k set: attempt k+1 accepted */
  double initial_cost;
  double final_cost;
  double cost_trace[AVM_MAX_ITER_TRACE]; /* x_cost after each attempt */
  double radius_trace[AVM_MAX_ITER_TRACE];
} avm_solve_summary;

/* One feature-selection problem = the inputs FeatureSelector::select() reads from its
 * members + estimator.  H is a runtime parameter (state_defs.h:8 fixes 13). */
typedef struct avm_fsel_batch {
  int32_t n_problems;
  int32_t horizon;   /* HORIZON; Omega is 9(H+1) square */
  int32_t max_cand;  /* stride of candidate arrays */
  int32_t max_used;  /* stride of already-tracked ("subset") arrays */
  int32_t max_cloud; /* stride of depth cloud arrays */
  int32_t max_features; /* maxFeatures_ */

  /* horizon states state_kkH[0..H] (state_defs.h:15-19): pos, vel, ba, quaternion(x,y,z,w) */
  const double* hor_pos;  /* [P][H+1][3] */
  const double* hor_quat; /* [P][H+1][4] x y z w */
  const int32_t* nr_imu;  /* [P] nrImuMeasurements */
  const double* delta_imu; /* [P] deltaImu */
  /* IMU noise passed to setParameters (std-devs passed where variances are expected — kept) */
  double acc_var, acc_bias_var;
  /* extrinsics + pinhole camera (config/euroc/euroc_config.yaml:11-22,30-42) */
  double q_ic[4]; /* x y z w */
  double t_ic[3];
  double fx, fy, cx, cy, k1, k2, p1, p2;
  int32_t image_width, image_height;

  /* new candidates (image_new), ascending feature id */
  const int32_t* n_cand;  /* [P] */
  const int32_t* cand_id; /* [P][max_cand] */
  const double* cand_xy;  /* [P][max_cand][2] normalized plane x,y (z==1) */
  const double* cand_prob; /* [P][max_cand] fPROB channel (float32-rounded upstream) */
  /* already tracked features present in this frame (subset) */
  const int32_t* n_used;  /* [P] */
  const int32_t* used_id; /* [P][max_used] */
  const double* used_xy;  /* [P][max_used][2] */
  /* depth cloud of initKDTree(): nip points w.r.t camera k+1 and their depths (feature_selector.cpp:396-419) */
  const int32_t* n_cloud;   /* [P] */
  const double* cloud_xy;   /* [P][max_cloud][2] */
  const double* cloud_depth; /* [P][max_cloud] */
} avm_fsel_batch;

typedef struct avm_fsel_out {
  int32_t* n_selected;   /* [P] */
  int32_t* selected_ids; /* [P][max_features] in selection order (blacklist order) */
  double* fvalues;       /* [P][max_features] fMax of each round (nullable) */
  double* min_gap;       /* [P][max_features] (nullable; round 5) fMax of the round minus the largest fValue among the OTHER candidates that
                            took part in it (+inf when there was no other): how firmly the round was decided.  Every round compares FP64
                            log-determinants, so a pick whose gap is of the order of their rounding errors (1e-12 of the value) is decided by
                            rounding - a host can see that here and need not expect another FP64 implementation (the reference's own run
                            included) to make the same pick.  Exact in the forms that score every candidate every round; in the lazy form of
                            large batches exact whenever it is below 1e-8 of the value and a lower bound otherwise.  Candidates the
                            std::map equal-key rule of sortedlogDetUB keeps out of the round are not counted as far as the pick meets them. */
} avm_fsel_out;

/* B4: FeatureSelector::generateFutureHorizon in IMU mode = HorizonGenerator::imu
 * (utility/horizon_generator.cpp:25-69; state_defs.h:15-19,37-41).  Inputs are what
 * setNextStateFromImuPropagation latched (feature_selector.cpp:38-70): state_k_ (tail of the window), state_k1_
 * (IMU-propagated current frame), the body acceleration / angular rate a_k1, w_k1. */
typedef struct avm_fsel_horizon_in {
  int32_t n_problems;
  int32_t horizon;          /* H: states 0..H are produced */
  const double* k_pos;      /* [P][3] state_k_  position */
  const double* k_quat;     /* [P][4] state_k_  attitude x y z w */
  const double* k_ba;       /* [P][3] state_k_  accelerometer bias (held constant over the horizon) */
  const double* k1_pos;     /* [P][3] state_k1_ position */
  const double* k1_vel;     /* [P][3] state_k1_ velocity */
  const double* k1_quat;    /* [P][4] state_k1_ attitude x y z w */
  const double* acc;        /* [P][3] a_k1 */
  const double* gyr;        /* [P][3] w_k1 */
  const int32_t* nr_imu;    /* [P] nrImuMeasurements */
  const double* delta_imu;  /* [P] deltaImu */
} avm_fsel_horizon_in;

/* A7: inputs of ProjectionTdFactor (factor/projection_td_factor.cpp:6-32,34-141; call site estimator.cpp:732-747),
 * one entry per factor: the factor on its own (the solve uses it through avm_window_batch::obs_vel_td when
 * opt->estimate_td != 0), for hosts that assemble problems themselves and for the parity tests. */
typedef struct avm_td_factor_batch {
  int32_t n;
  const double* pose_i;    /* [n][7] x y z qx qy qz qw */
  const double* pose_j;    /* [n][7] */
  const double* ex_pose;   /* [n][7] */
  const double* inv_depth; /* [n] */
  const double* td;        /* [n] current time offset (para_Td) */
  const double* pts_i;     /* [n][2] normalized plane, z == 1 */
  const double* pts_j;     /* [n][2] */
  const double* vel_i;     /* [n][2] image velocity of the feature in frame i (FeaturePerFrame::velocity) */
  const double* vel_j;     /* [n][2] */
  const double* td_i;      /* [n] cur_td when frame i was taken */
  const double* td_j;      /* [n] */
  const double* row_i;     /* [n] image row (uv.y()), before the ROW / 2 shift of the constructor */
  const double* row_j;     /* [n] */
  double tr, row;          /* TR (rolling shutter read-out time), ROW (image height): parameters.cpp */
  double focal_length;     /* sqrt_info = focal_length / 1.5 * I2 (estimator.cpp:17) */
} avm_td_factor_batch;

typedef struct avm_config {
  int32_t device;       /* HIP device ordinal */
  int32_t max_windows;  /* capacity of one avm_window_solve_batch call */
  int32_t max_problems; /* capacity of one avm_fsel_select_batch call */
  int32_t abi_version;  /* AVM_ABI_VERSION of the header the caller was compiled against.  avm_create() refuses any other value
                         * (AVM_ERR_INVALID): the structs of this header carry no size fields, so a host built against an older
                         * header - avm_fsel_out had three members before min_gap - must not get as far as a call that reads them. */
  int32_t reserved[4];
} avm_config;
#define AVM_ABI_VERSION 6 /* bumped whenever a struct of this header changes its layout or an entry point its meaning */

typedef struct avm_ctx avm_ctx;

/* ---- lifecycle -------------------------------------------------------------- */
int avm_default_options(avm_options* opt);
int avm_create(const avm_config* cfg, avm_ctx** out);
void avm_destroy(avm_ctx* ctx);
const char* avm_last_error(const avm_ctx* ctx);
const char* avm_version(void);
int avm_abi_version(void); /* AVM_ABI_VERSION the library was built with */

/* ---- HP-A: Estimator::optimization() for a batch of independent windows ------ */
/* solve in place (states updated like double2vector+vector2double leave them),
 * optionally producing the new prior.  prior_out may be NULL iff
 * opt->marginalization_flag == AVM_MARGIN_NONE.  summary may be NULL. */
int avm_window_solve_batch(avm_ctx* ctx, const avm_options* opt, avm_mem mem,
                           const avm_window_batch* batch, avm_prior_out* prior_out,
                           avm_solve_summary* summary /* [B], host or device per mem */);

/* the single-call form, exactly one Estimator::optimization() (estimator.h:47): the same arguments with
 * batch->n_windows == 1 (AVM_ERR_INVALID otherwise).  One window occupies one compute unit; see DESIGN.md for its latency. */
int avm_window_solve(avm_ctx* ctx, const avm_options* opt, avm_mem mem, const avm_window_batch* window,
                     avm_prior_out* prior_out, avm_solve_summary* summary);

/* A4 only: IntegrationBase for every interval of every window (integration_base.h:13-158).
 * out_* are [B][10][...]: delta (p3,q4 xyzw,v3 = 10), jacobian 15x15, covariance 15x15, sum_dt. */
int avm_imu_preintegrate_batch(avm_ctx* ctx, const avm_options* opt, avm_mem mem,
                               const avm_window_batch* batch, double* out_delta, double* out_jacobian,
                               double* out_covariance, double* out_sum_dt);

/* SURVEY 8(f)1 - the step right before optimization() in solveOdometry() (estimator.cpp:471):
 * FeatureManager::triangulate (feature_manager.cpp:202-257).  Every feature whose inv_depth is <= 0
 * ("no depth yet": estimated_depth = -1 at construction, feature_manager.h:61) gets 1 / depth from the linear
 * multi-view triangulation over all of its observations, in the camera frame of its first observation: the
 * smallest right singular vector v of the (2 nobs) x 4 system, depth = v[2] / v[3]; depths < 0.1 fall back to
 * init_depth (INIT_DEPTH = 5.0, parameters.cpp:113).  In place on batch->inv_depth; reads pose, ex_pose and the
 * feature tables only.  (IntegrationBase::repropagate, the other half of that row, is
 * avm_imu_preintegrate_batch() with the new linearization biases in imu_lin_ba / imu_lin_bg.) */
int avm_triangulate_batch(avm_ctx* ctx, avm_mem mem, avm_window_batch* batch, double init_depth);

/* SURVEY 8(f)1, second half: the dead-reckoning of the newest frame in Estimator::processIMU (estimator.cpp:100-107).
 * Frame AVM_WINDOW_SIZE of every window (which slideWindow() left holding a copy of the previous newest frame) is
 * carried through the raw samples of the last interval (imu_* [B][AVM_WINDOW_SIZE-1], row 0 of acc/gyr = the sample
 * the interval was constructed with): world-frame midpoint integration with the biases of that frame and gravity g;
 * Rs is propagated as a matrix times the rotation matrix of the UNNORMALIZED deltaQ, as the reference does, and turned
 * into the pose quaternion at the end.  In place on batch->pose[.][10] and batch->speedbias[.][10][0..2]. */
int avm_imu_propagate_batch(avm_ctx* ctx, avm_mem mem, avm_window_batch* batch, const double g[3]);

/* B4 (see avm_fsel_horizon_in): constant body acceleration / angular rate propagation of states 2..H with
 * Qimu = deltaQ(w * deltaImu) - UNNORMALIZED and never renormalized in the loop (horizon_generator.cpp:46,54) - and
 * gravity (0, 0, -9.80665).  Writes hor_pos [P][H+1][3] and hor_quat [P][H+1][4] (x y z w), the layout
 * avm_fsel_batch consumes. */
int avm_fsel_horizon_imu(avm_ctx* ctx, avm_mem mem, const avm_fsel_horizon_in* in, double* hor_pos, double* hor_quat);

/* A7: ProjectionTdFactor::Evaluate for n factors: residual [n][2], jac [n][2][20] with columns
 * pose_i 6 | pose_j 6 | ex_pose 6 | inv_depth 1 | td 1 (local 6-column pose blocks, like avm_window_eval_factors). */
int avm_projection_td_eval(avm_ctx* ctx, avm_mem mem, const avm_td_factor_batch* f, double* residual, double* jac);

/* B8, first half: the depth cloud FeatureSelector::initKDTree() builds (feature_selector.cpp:380-433), one cloud per
 * window: every feature of the window (they all pass used_num >= 2 && start_frame < WINDOW_SIZE - 2 by construction)
 * with start_frame <= WINDOW_SIZE * 3 / 4 and solve_flag == 1 (depth = 1 / inv_depth >= 0, feature_manager.cpp:141-159)
 * is lifted to the world with its first observation and its depth, moved into camera k+1 (state_k1_: k1_pos [B][3],
 * k1_quat [B][4] x y z w; extrinsic = the window's ex_pose) and projected to the normalized plane.  Output in feature
 * order, the layout avm_fsel_batch consumes: n_cloud [B], cloud_xy [B][max_cloud][2], cloud_depth [B][max_cloud].
 * (The nearest-neighbour lookup itself, findNNDepth, runs inside avm_fsel_select_batch.) */
int avm_fsel_build_cloud(avm_ctx* ctx, avm_mem mem, const avm_window_batch* windows, const double* k1_pos, const double* k1_quat,
                         int32_t max_cloud, int32_t* n_cloud, double* cloud_xy, double* cloud_depth);

/* the window roll, Estimator::slideWindow (estimator.cpp:996-1107) with FeatureManager::removeBackShiftDepth /
 * removeBack / removeFront (feature_manager.cpp:275-352), applied IN PLACE to the batch so that the next solve can run on
 * the same (device-resident) tables; the prior hand-off is already part of avm_window_solve_batch (blk_frame carries the
 * addr_shift).  frame_count == WINDOW_SIZE is assumed.  The const qualifiers of the batch's table pointers are cast away:
 * the caller owns writable memory of the declared strides.
 *   AVM_MARGIN_OLD:        poses / speed-biases / IMU intervals move down by one (frame 10 keeps its values, interval 9
 *                          becomes empty: imu_n = 0, row 0 of imu_acc / imu_gyr = the last sample pushed, linearization
 *                          biases = those of frame 10); a feature with start_frame != 0 starts one frame earlier, the
 *                          others lose their first observation and are erased when fewer than 2 remain (shift_depth != 0,
 *                          i.e. solver_flag == NON_LINEAR; their depth is re-anchored in the new first frame, init_depth
 *                          where it comes out non-positive) or when none remains (shift_depth == 0).
 *   AVM_MARGIN_SECOND_NEW: frame 10 overwrites frame 9, the samples of interval 9 are appended to interval 8
 *                          (AVM_ERR_CAPACITY if that exceeds max_samp); a feature that starts in frame 10 starts in 9,
 *                          the others lose their frame-9 observation if they were still tracked in frame 9, and are
 *                          erased when none remains.
 * Erasing compacts the per-feature arrays in order (std::list order); observations stay where they are, only
 * feat_obs_begin / feat_nobs change (one element moves for removeFront).  inv_depth holds 1 / estimated_depth. */
int avm_slide_window(avm_ctx* ctx, avm_mem mem, avm_window_batch* windows, int32_t marginalization_flag, int32_t shift_depth, double init_depth);

/* ---- SURVEY 8(f)4 + B4 (ground-truth mode): host-side format adapters.  Pure host bookkeeping like their
 * reference counterparts: no device work, no avm_ctx, usable without a GPU. ---------------------------------------- */

/* EuRoC ground-truth table with the reference's seek cursor (HorizonGenerator::truth_, seek_idx_, horizon_generator.h). */
typedef struct avm_gt avm_gt;
/* HorizonGenerator::loadGroundTruth (horizon_generator.cpp:169-196): first line = header, then
 * timestamp[ns], p(3), q(w x y z), v(3), w(3), a(3); the timestamp is stod(field) * 1e-9.  NULL on I/O / parse error. */
avm_gt* avm_gt_load_csv(const char* data_csv);
/* the same table from memory: rows [n][17], column 0 already in nanoseconds as a double */
avm_gt* avm_gt_from_rows(const double* rows, int32_t n);
void avm_gt_free(avm_gt* gt);
int32_t avm_gt_size(const avm_gt* gt);
int32_t avm_gt_seek(const avm_gt* gt); /* current seek_idx_ */
/* HorizonGenerator::groundTruth (horizon_generator.cpp:73-123) incl. getNextFrameTruth (:200-210): the H future poses
 * from the relative motion of the ground truth, starting at state k (timestamp, k_pos [3], k_quat [4] x y z w).
 * Stateful like the reference: the seek cursor only moves forward (and moves by at least one row per call).
 * Out: hor_pos [H+1][3], hor_quat [H+1][4] (x y z w; the products are not renormalized, as in the reference).
 * AVM_ERR_INVALID where the reference would read past the end of the table. */
int avm_fsel_horizon_ground_truth(avm_gt* gt, int32_t horizon, double timestamp_k, const double* k_pos, const double* k_quat,
                                  double delta_frame, double* hor_pos, double* hor_quat);

/* The feature message decode of estimator_node.cpp:303-321 (sensor_msgs::PointCloud -> image_t): points [n][3] float32
 * (geometry_msgs/Point32), channels[0..5] = id_of_point, u, v, velocity_x, velocity_y, probability (float32 each, [n]).
 * feature_id = int(ch0 + 0.5) / num_cam, camera_id = ... % num_cam; xyz_uv_velocity [n][8] = x y z u v vx vy prob.
 * Output in image_t order: ascending feature_id (std::map), message order within an id.  AVM_ERR_INVALID if a z != 1
 * (ROS_ASSERT(z == 1), :317). */
int avm_image_from_pointcloud(int32_t n_points, const float* points_xyz, const float* const* channels, int32_t num_cam,
                              int32_t* feature_id, int32_t* camera_id, double* xyz_uv_velocity);

/* A5/A6/A8 only: evaluate every factor once at the current state and return
 * residuals/Jacobians (local 6-column pose blocks).  Used by the per-factor parity tests.
 *   proj_r [B][max_obs][2], proj_J [B][max_obs][2][13]  (pose_i 6 | pose_j 6 | inv_depth 1), index = observation slot
 *   imu_r  [B][10][15],     imu_J  [B][10][15][30]      (pose_i 6 | sb_i 9 | pose_j 6 | sb_j 9)
 *   prior_res [B][max_prior]
 * robust loss (Cauchy) correction is applied to proj_* when apply_loss != 0. */
int avm_window_eval_factors(avm_ctx* ctx, const avm_options* opt, avm_mem mem,
                            const avm_window_batch* batch, int apply_loss, double* proj_r, double* proj_J,
                            double* imu_r, double* imu_J, double* prior_res, double* cost /* [B] */);

/* ---- HP-B: FeatureSelector::select() for a batch of independent frames ------- */
int avm_fsel_select_batch(avm_ctx* ctx, avm_mem mem, const avm_fsel_batch* batch, avm_fsel_out* out);
/* the single-call form, one FeatureSelector::select() (feature_selector.h:49-50): frame->n_problems == 1
 * (AVM_ERR_INVALID otherwise); selected_ids has room for frame->max_features ids (selection order), fvalues_opt may be NULL. */
int avm_fsel_select(avm_ctx* ctx, avm_mem mem, const avm_fsel_batch* frame, int32_t* selected_ids, int32_t* n_selected,
                    double* fvalues_opt);

/* how often this ctx's select calls had to fall back from the all-rounds-in-one-launch kernel (csrc/fsel.hip): counters since
 * avm_create.  out[0] = CALLS that had to be re-run in a slower mode (once per call), out[1] = LAUNCHES that reported a timed-out
 * wait or an unfinished frame (a call that falls two modes counts twice here), out[2] = the mode the next call starts in (2 = a
 * team per XCD, 1 = one team over all XCDs, 0 = one launch per round), out[3] = select calls so far.  A degraded call costs at most
 * the 20 ms spin time-out of the failed launch plus the slower mode's run time; the ctx probes the fast mode again after 16 calls,
 * and after 32, 64 ... 4096 if the probes keep failing (back to 16 once a fast-mode call has gone through). */
int avm_fsel_fallback_stats(const avm_ctx* ctx, int64_t out[4]);

/* B8, second half as a parity surface: FeatureSelector::findNNDepth (feature_selector.cpp:437-459) for every candidate of every
 * frame - the depth of the cloud point the reference's kd-tree search returns for the candidate on the normalized plane (exact
 * 1-NN, squared Euclidean distance as nanoflann's L2_Simple_Adaptor, feature_selector.h:143; 1.0 for an empty cloud,
 * feature_selector.cpp:444).  depth [P][max_cand] (entries beyond n_cand: 0).  The same search runs inside
 * avm_fsel_select_batch / avm_fsel_information; this entry exists so that it can be checked against the reference's own nanoflann
 * (tests/golden/nanoflann_nn.npz, nanoflann_nn2.npz).  Round 5: the tree is built and walked as nanoflann builds and walks it
 * (csrc/fsel.hip: fsel_kdtree_kernel, kd_depth), so among cloud points at bit-identical distances the answer is nanoflann's too - the
 * point its traversal meets first; every query of the two fixtures is answered bit for bit.  max_cloud <= 4096 (the tree is built in LDS). */
int avm_fsel_nn_depth(avm_ctx* ctx, avm_mem mem, const avm_fsel_batch* batch, double* depth);

/* B5/B6 only: Omega_kkH (+prior) [P][N][N] and compact Delta_ell position blocks
 * [P][max_cand][3H][3H] (+ valid flag [P][max_cand]); for parity tests. */
int avm_fsel_information(avm_ctx* ctx, avm_mem mem, const avm_fsel_batch* batch, double* omega,
                         double* delta_cand, int32_t* cand_valid);

/* ---- multi-GPU: windows and selector frames are independent, so a job is one process per GPU, each with its
 * own ctx and a contiguous block of the work; the only exchange is a gather of the final states.  The library owns a raw
 * RCCL communicator for it (rccl.h over xGMI; librccl.so.1 is dlopen'ed on first use, a single-GPU host never loads it).
 *   avm_comm_unique_id   rank 0 creates the 128-byte ncclUniqueId; the caller carries it to the other ranks (any channel)
 *   avm_comm_init        collective over all ranks: ncclCommInitRank on this ctx's device
 *   avm_gather_states    ncclAllGather of `count` doubles per rank, device pointers, on the ctx stream; returns when done.
 *                        recv is [n_ranks][count]; rank r's block lands at recv + r * count on every rank
 *   avm_comm_destroy     (also done by avm_destroy) */
#define AVM_COMM_ID_BYTES 128
int avm_comm_unique_id(avm_ctx* ctx, void* id /* AVM_COMM_ID_BYTES */);
int avm_comm_init(avm_ctx* ctx, int32_t n_ranks, int32_t rank, const void* id);
int avm_gather_states(avm_ctx* ctx, const double* send, double* recv, size_t count);
int avm_comm_destroy(avm_ctx* ctx);

/* the HIP stream (hipStream_t) every call on this ctx is enqueued on, for callers that produce device-resident inputs
 * on streams of their own (see "stream ordering" above) */
int avm_ctx_stream(const avm_ctx* ctx, void** stream);

/* ---- timing of the last call on the ctx stream (HIP events), milliseconds ---- */
int avm_last_kernel_ms(const avm_ctx* ctx, const char* which, float* ms);

#ifdef __cplusplus
}
#endif
#endif /* AVM_H_ */
