/*
 * lfvio.h — C-ABI boundary of the MI355X sliding-window solver.
 *
 * This is the single host -> device seam of the drop-in.  The reference has no
 * FFI: its seam is the C++ member `void Estimator::optimization()`
 * (vins_estimator/src/estimator.h:47, body estimator.cpp:676-1009), which talks
 * to Ceres through flat `para_*` arrays (estimator.h:107-113) filled by
 * `vector2double()` (estimator.cpp:488-530).  A re-implemented
 * `optimization()` body packs exactly those arrays, the feature-per-frame
 * observations (feature_manager.h:18-71), the ten `IntegrationBase` results
 * (factor/integration_base.h:188-203) and the marginalization prior
 * (factor/marginalization_factor.h:47-72) into the PODs below and calls the
 * entry points declared here.  See INTEGRATION.md for the reference-side stub.
 *
 * Conventions
 *   - everything is FP64; matrices are ROW-major; quaternions inside pose
 *     blocks are stored [x y z w] exactly like para_Pose (estimator.cpp:492-499);
 *     preintegration delta_q is [x y z w] (Eigen coeffs() order).
 *   - pose block  = [px py pz qx qy qz qw]                  (SIZE_POSE = 7)
 *     speed/bias  = [vx vy vz bax bay baz bgx bgy bgz]      (SIZE_SPEEDBIAS = 9)
 *   - caller owns every buffer; the context owns device memory, streams and
 *     graphs; no pointer is retained across calls.
 *   - every function returns LFVIO_OK (0) or a negative error; on error the
 *     outputs are untouched so the caller can fall back (the reference itself
 *     has no error channel: optimization() is void, estimator.cpp:676).
 *   - one caller at a time per context (the reference holds m_estimator across
 *     processImage(), estimator_node.cpp:215-332).
 */
#ifndef LFVIO_H
#define LFVIO_H

#ifdef __cplusplus
extern "C" {
#endif

#define LFVIO_WINDOW_SIZE 10                      /* parameters.h:12 */
#define LFVIO_NUM_FRAMES (LFVIO_WINDOW_SIZE + 1)  /* estimator.h:107 */
#define LFVIO_SIZE_POSE 7                         /* parameters.h:45 */
#define LFVIO_SIZE_SPEEDBIAS 9                    /* parameters.h:46 */
#define LFVIO_MAX_PRIOR_BLOCKS 24                 /* 11 poses + 11 speed/bias + ex + td */
#define LFVIO_MAX_PRIOR_DIM 172                   /* 11*6 + 11*9 + 6 + 1 tangent dims */
#define LFVIO_MAX_TRACE 64

/* error codes */
#define LFVIO_OK 0
#define LFVIO_ERR_ARG (-1)         /* malformed window (bad CSR, NULL, sizes)  */
#define LFVIO_ERR_DEVICE (-2)      /* HIP runtime error                         */
#define LFVIO_ERR_NONFINITE (-3)   /* non-finite cost / state / prior           */
/* A reduced system that never becomes positive definite is NOT an error of the call: as in Ceres, every failed
 * factorization is an invalid step (mu *= 10), five in a row end the solve with termination = LFVIO_FAILURE and the
 * state of the last accepted step (trust_region_minimizer.cc: HandleInvalidStep). */

/* marginalization flags: Estimator::MarginalizationFlag, estimator.h:58-62 */
#define LFVIO_MARGIN_OLD 0
#define LFVIO_MARGIN_SECOND_NEW 1

/* termination: ceres::TerminationType subset used by TrustRegionMinimizer */
#define LFVIO_CONVERGENCE 0
#define LFVIO_NO_CONVERGENCE 1
#define LFVIO_FAILURE 2

/* parameter-block identity.  The reference identifies prior blocks by the
 * ADDRESS of para_* rows (addr_shift, estimator.cpp:921-933); the ABI uses
 * (kind, frame) tags instead. */
#define LFVIO_BLOCK_POSE 0       /* para_Pose[frame],       global 7 / local 6 */
#define LFVIO_BLOCK_SPEEDBIAS 1  /* para_SpeedBias[frame],  9 / 9              */
#define LFVIO_BLOCK_EX_POSE 2    /* para_Ex_Pose[0],        7 / 6              */
#define LFVIO_BLOCK_TD 3         /* para_Td[0],             1 / 1              */

typedef struct LfvioBlockId {
  int kind;
  int frame;
} LfvioBlockId;

/* One IntegrationBase (factor/integration_base.h:188-203) as consumed by
 * IMUFactor::Evaluate (factor/imu_factor.h:19-200). */
typedef struct LfvioPreintegration {
  double sum_dt;
  double delta_p[3];
  double delta_q[4]; /* x y z w */
  double delta_v[3];
  double linearized_ba[3];
  double linearized_bg[3];
  double jacobian[225];   /* 15x15 row-major, order O_P O_R O_V O_BA O_BG */
  double covariance[225]; /* 15x15 row-major */
} LfvioPreintegration;

/* MarginalizationInfo after marginalize()+getParameterBlocks()
 * (factor/marginalization_factor.cpp:174-319): the kept blocks in order,
 * their linearization points, and linearized_jacobians / linearized_residuals. */
typedef struct LfvioPrior {
  int valid;      /* 0 => no prior (last_marginalization_info == nullptr) */
  int m;          /* marginalized tangent dim (informational)             */
  int n;          /* kept tangent dim = rows = cols of linearized_jacobians */
  int num_blocks; /* kept parameter blocks                                 */
  LfvioBlockId blocks[LFVIO_MAX_PRIOR_BLOCKS]; /* AFTER addr_shift          */
  int block_idx[LFVIO_MAX_PRIOR_BLOCKS];       /* column offset (keep_block_idx - m) */
  double block_x0[LFVIO_MAX_PRIOR_BLOCKS][9];  /* keep_block_data (global size) */
  double linearized_jacobians[LFVIO_MAX_PRIOR_DIM * LFVIO_MAX_PRIOR_DIM]; /* n x n row-major, leading dim n */
  double linearized_residuals[LFVIO_MAX_PRIOR_DIM];
} LfvioPrior;

/* The whole input of one optimization() call. */
typedef struct LfvioWindow {
  /* state, as written by vector2double() (estimator.cpp:488-530) */
  double para_pose[LFVIO_NUM_FRAMES][LFVIO_SIZE_POSE];
  double para_speed_bias[LFVIO_NUM_FRAMES][LFVIO_SIZE_SPEEDBIAS];
  double para_ex_pose[LFVIO_SIZE_POSE];
  double para_td;

  /* flags / solver options (estimator.cpp:690-703, 810-822) */
  int estimate_extrinsic;            /* ESTIMATE_EXTRINSIC != 0                */
  int estimate_td;                   /* ESTIMATE_TD: ProjectionTdFactor vs ProjectionFactor */
  int max_num_iterations;            /* NUM_ITERATIONS                          */
  double max_solver_time_in_seconds; /* <= 0: disabled (parity / bench runs); > 0: honoured by the synchronous entry
                                        points between graph launches (every 2 passes), ignored by *_async    */

  /* globals read by the factors (parameters.h:17-41) */
  double g[3];      /* G                                      */
  double tr;        /* TR (rolling-shutter read-out time)     */
  double row;       /* ROW (image height)                     */
  double sqrt_info; /* FOCAL_LENGTH / 1.5, estimator.cpp:18-19 */

  /* landmarks that pass `used_num >= 2 && start_frame < WINDOW_SIZE - 2`
   * (feature_manager.cpp:36), in f_manager.feature list order, CSR over
   * their feature_per_frame vectors.  Observation o of landmark l is seen in
   * frame start_frame[l] + (o - obs_offset[l]); the first one is the anchor
   * (estimator.cpp:737-745). */
  int num_landmarks;
  int num_observations;            /* obs_offset[num_landmarks]              */
  const int *start_frame;          /* [N]                                    */
  const int *obs_offset;           /* [N+1]                                  */
  const double *inv_depth;         /* [N] para_Feature = 1/estimated_depth   */
  const double *obs_point;         /* [M][3] FeaturePerFrame::point (unit bearing) */
  const double *obs_velocity;      /* [M][3] FeaturePerFrame::velocity       */
  const double *obs_cur_td;        /* [M]    FeaturePerFrame::cur_td         */
  const double *obs_uv_y;          /* [M]    FeaturePerFrame::uv.y()         */

  /* pre_integrations[1..10] (estimator.cpp:717-724); imu[i] links frame i -> i+1 */
  LfvioPreintegration imu[LFVIO_WINDOW_SIZE];

  /* last_marginalization_info (+ parameter blocks); prior->valid==0 or NULL => none */
  const LfvioPrior *prior;
} LfvioWindow;

typedef struct LfvioIterationSummary { /* ceres::IterationSummary subset */
  double cost;
  double cost_change;
  double gradient_max_norm; /* NaN in the entry of a successful iteration that ended the loop (iteration cap, wall-clock cap): Ceres
                             * evaluates the gradient at the accepted point before it tests the cap, the device's next
                             * linearization never happens; the reference reads no summary field.  lfvio_debug_linearize() at the
                             * solution gives it (tests/test_gpu_parity.py::test_kkt_residual_at_the_solution). */
  double step_norm;
  double relative_decrease;
  double trust_region_radius;
  int step_is_valid;
  int step_is_successful;
} LfvioIterationSummary;

/* Output of the solve: the para_* arrays as Ceres leaves them (BEFORE
 * double2vector(), estimator.cpp:830, which stays on the host). */
typedef struct LfvioSolution {
  double para_pose[LFVIO_NUM_FRAMES][LFVIO_SIZE_POSE];
  double para_speed_bias[LFVIO_NUM_FRAMES][LFVIO_SIZE_SPEEDBIAS];
  double para_ex_pose[LFVIO_SIZE_POSE];
  double para_td;
  double *inv_depth; /* [N] caller-allocated */
  int num_iterations; /* summary.iterations.size() (iteration 0 included) */
  int num_successful_steps;
  int num_unsuccessful_steps;
  int termination;
  double initial_cost;
  double final_cost;
  LfvioIterationSummary trace[LFVIO_MAX_TRACE];
} LfvioSolution;

typedef struct lfvio_ctx lfvio_ctx;

/* Create a context on HIP device `device`.  Fails (returns NULL) when no
 * gfx950 device / HIP runtime is usable: there is no CPU fallback. */
lfvio_ctx *lfvio_create(int device);
void lfvio_destroy(lfvio_ctx *ctx);
const char *lfvio_last_error(const lfvio_ctx *ctx);
const char *lfvio_version(void);

/* ceres::Solve replacement for the problem built at estimator.cpp:678-825:
 * upload, run the trust-region loop on the device, download. */
int lfvio_solve(lfvio_ctx *ctx, const LfvioWindow *in, LfvioSolution *out);

/* MarginalizationInfo::{preMarginalize,marginalize,getParameterBlocks}
 * replacement for the factor sets built at estimator.cpp:833-1005.  `in` holds
 * the state AFTER double2vector()+vector2double().  For MARGIN_SECOND_NEW with
 * no prior touching Pose[WINDOW_SIZE-1] the reference does nothing
 * (estimator.cpp:942-943): out->valid is copied from the input prior. */
int lfvio_marginalize(lfvio_ctx *ctx, const LfvioWindow *in, int flag, LfvioPrior *out);

/* ---- device-resident API (throughput / bench; same kernels) -------------
 * upload once, then run the whole optimization() — solve, the gauge fix of
 * double2vector() (estimator.cpp:532-600) and marginalization — without
 * leaving the device.  `batch` independent windows can be resident at once
 * (BASELINE config "512 independent windows"). */
int lfvio_batch_reserve(lfvio_ctx *ctx, int batch, int max_landmarks, int max_observations);
int lfvio_batch_upload(lfvio_ctx *ctx, int slot, const LfvioWindow *in);
/* run optimization() for slots [0, count): flag per call (same for all slots) */
int lfvio_batch_optimize(lfvio_ctx *ctx, int count, int marg_flag);
/* enqueue only (no host sync); stream is the context's stream */
int lfvio_batch_optimize_async(lfvio_ctx *ctx, int count, int marg_flag);
int lfvio_batch_sync(lfvio_ctx *ctx);
/* state returned here is AFTER the gauge fix (what Ps/Rs/Vs... hold after
 * double2vector()), re-expressed through vector2double(). */
int lfvio_batch_download(lfvio_ctx *ctx, int slot, LfvioSolution *sol, LfvioPrior *prior);
/* The same call split in two for ONE resident window (slot 0), for callers that use the state before they need the
 * prior — the reference publishes the pose right after optimization() (estimator_node.cpp:  pubOdometry behind
 * processImage) and reads last_marginalization_info first in the NEXT optimization() (estimator.cpp:700-706):
 *   lfvio_batch_optimize_begin   returns with the solution (after the gauge fix, as lfvio_batch_download gives it) as soon as
 *                                solve + gauge fix are out; the marginalization is still running on the device then
 *                                (lfvio_batch_optimize_pending() == 1).  The device pushes the state into mapped host memory and
 *                                the call polls one word of it: no copy, no stream synchronization in front of the caller.
 *   lfvio_batch_optimize_finish  waits for the rest and delivers the prior (NULL: just wait).  Any other entry point on the
 *                                context waits for the tail first, so forgetting it costs overlap, not correctness; the prior
 *                                must be collected before the slot is uploaded again.
 * lfvio_triangulate / lfvio_shift_depth / lfvio_preintegrate do NOT wait: they run beside the tail on their own stream. */
/* lfvio_batch_upload_chained  the upload of the NEXT window of the same estimator while that marginalization is still
 *                                running: the window's prior is *prior_io (in->prior is ignored), and if a call is in flight
 *                                on the context it is THAT call's prior — the landmark tables of the new window are packed
 *                                on the host while the device finishes, then the prior is collected into *prior_io (as
 *                                lfvio_batch_optimize_finish would) and goes up with the window.  With nothing in flight it
 *                                is lfvio_batch_upload with in->prior = prior_io. */
int lfvio_batch_upload_chained(lfvio_ctx *ctx, int slot, const LfvioWindow *in, LfvioPrior *prior_io);
/* lfvio_batch_upload_chained_device  the same hand-over WITHOUT the host in it (round 5): while the call begun with
 *                                lfvio_batch_optimize_begin is still marginalizing, the next window of the same estimator is
 *                                packed, its copies are enqueued behind that marginalization, and the prior it is producing becomes
 *                                this window's prior where it lies on the device (in->prior is ignored; the block structure is the one
 *                                this library planned for that marginalization, the values never leave the GPU).  Nothing is waited
 *                                for: the call returns as soon as the copies are enqueued, and the lfvio_batch_optimize_begin that
 *                                follows goes out behind them — the device runs window after window back to back, the caller still
 *                                gets every state as early as before (estimator.cpp:700-706: the prior is read by the NEXT
 *                                optimization(), which is exactly where it stays).  The prior of the call that was in flight is no
 *                                longer collectable afterwards (lfvio_batch_optimize_finish delivers the NEWEST call's prior).
 *                                LFVIO_ERR_ARG (nothing enqueued, the call in flight untouched) when there is no call in flight on
 *                                slot 0 or when its marginalization passes the input prior through (MARGIN_SECOND_NEW without a prior
 *                                on the newest pose): use lfvio_batch_upload_chained then.  If that marginalization fails on the
 *                                device — or leaves a prior of another block structure than the one promised — the window runs
 *                                without a prior and the next lfvio_batch_optimize_begin / lfvio_batch_optimize(ctx, 1, flag)
 *                                returns LFVIO_ERR_DEVICE.  Those two are the calls that may follow this upload: any other way
 *                                of optimizing the slot (lfvio_batch_optimize_async, a count above one) is refused with
 *                                LFVIO_ERR_ARG — the verdict on the prior travels with the graph of one window's synchronous call. */
int lfvio_batch_upload_chained_device(lfvio_ctx *ctx, int slot, const LfvioWindow *in);
int lfvio_batch_optimize_begin(lfvio_ctx *ctx, int marg_flag, LfvioSolution *sol);
int lfvio_batch_optimize_finish(lfvio_ctx *ctx, LfvioPrior *prior);
int lfvio_batch_optimize_pending(const lfvio_ctx *ctx);
/* the context's HIP stream (hipStream_t) for event timing by the caller */
void *lfvio_stream(lfvio_ctx *ctx);

/* ---- the landmark-parallel steps either side of optimization() ------------------------
 * lfvio_triangulate: FeatureManager::triangulate (feature_manager.cpp:199-253).  For every landmark with
 * estimated_depth <= 0 (the caller lists only landmarks with used_num >= 2 && start_frame < WINDOW_SIZE - 2, in the
 * CSR form of LfvioWindow): the 2k x 4 system of the k observations in the frame of the first one, its right singular
 * vector of the smallest singular value v (the reference: Eigen::JacobiSVD, ComputeThinV, last column),
 * depth = (v[0:3] / v[3]) . point_0, replaced by init_depth when negative.  Landmarks with estimated_depth > 0 are
 * left alone.  estimated_depth is read and written in place. */
typedef struct {
  int num_landmarks, num_observations;
  const int *start_frame;   /* [N] */
  const int *obs_offset;    /* [N + 1] */
  const double *obs_point;  /* [M][3]  FeaturePerFrame::point as stored (normalized inside, used raw in the dot product) */
  double Ps[LFVIO_NUM_FRAMES][3];
  double Rs[LFVIO_NUM_FRAMES][9]; /* row-major */
  double tic[3], ric[9];
  double init_depth;              /* INIT_DEPTH, parameters.cpp:116 */
} LfvioTriangulateIn;
int lfvio_triangulate(lfvio_ctx *ctx, const LfvioTriangulateIn *in, double *estimated_depth);

/* lfvio_shift_depth: the arithmetic of FeatureManager::removeBackShiftDepth (feature_manager.cpp:271-310) for the n
 * landmarks that started in the marginalized frame and keep >= 2 observations: uv_i is the erased first observation,
 * depth <- || new_R^T (marg_R (uv_i * depth) + marg_P - new_P) ||, or init_depth when that is not > 0.
 * (Erasing observations / features and start_frame-- stay list bookkeeping on the host.) */
int lfvio_shift_depth(lfvio_ctx *ctx, int n, const double *uv_i /* [n][3] */, const double marg_R[9], const double marg_P[3],
                      const double new_R[9], const double new_P[3], double init_depth, double *estimated_depth /* [n] */);

/* lfvio_preintegrate: IntegrationBase::push_back / propagate / midPointIntegration (factor/integration_base.h:29-158) for
 * num_intervals independent keyframe intervals at once — the ten of a window after a bias update (repropagate(),
 * integration_base.h:41-52, called from Estimator::double2vector / solveGyroscopeBias), or one interval as frames arrive.
 * Each interval is the constructor arguments (acc_0, gyr_0, linearized_ba, linearized_bg, integration_base.h:15-27) plus its
 * buffered samples dt_buf / acc_buf / gyr_buf; noise = {ACC_N, GYR_N, ACC_W, GYR_W} (parameters.cpp:94-97).  out[k] is
 * what LfvioWindow::imu[k] takes: delta_p/q/v, sum_dt, jacobian and covariance after the last sample.  An interval with
 * no samples returns the constructor state (identity jacobian, zero covariance). */
typedef struct {
  int num_samples;
  const double *dt;  /* [num_samples]    */
  const double *acc; /* [num_samples][3] */
  const double *gyr; /* [num_samples][3] */
  double acc_0[3], gyr_0[3], linearized_ba[3], linearized_bg[3];
} LfvioImuInterval;
int lfvio_preintegrate(lfvio_ctx *ctx, int num_intervals, const LfvioImuInterval *in, const double noise[4], LfvioPreintegration *out);

/* ---- landmark-sharded API (multi-GPU) ------------------------
 * Every rank passes the same window but linearizes only landmarks [lm_begin, lm_end) (caller order);
 * IMU factors and the prior are added on the rank(s) with add_pose_side != 0 — exactly one rank.
 * The library exposes a device exchange buffer of lfvio_shard_exchange_len() doubles,
 *   [ H_pp packed | g_p | Schur sums | 16 scalars ],  scalars at lfvio_shard_scalar_offset();
 * the CALLER sum-all-reduces it in place (RCCL) where a phase function returns 1:
 *   while (state != 2) {
 *     if (lfvio_shard_linearize(ctx) == 1) all_reduce(buf[0 : len]);            // 151 KB
 *     if (lfvio_shard_solve(ctx)     == 1) all_reduce(buf[scalar_offset : len]); // 128 B
 *     if (lfvio_shard_candidate(ctx) == 1) all_reduce(buf[scalar_offset : len]); // 128 B
 *     lfvio_shard_decide(ctx, &state);   // identical decision on every rank, no broadcast
 *   }
 *   if (lfvio_shard_marg_linearize(ctx, flag) == 1) all_reduce(buf[0 : len]);   // optional: the next prior
 *   lfvio_shard_marg_finish(ctx, flag, &prior);   // identical on every rank (estimator.cpp:833-1005)
 *   lfvio_shard_finish(ctx, &sol);       // pose-side state replicated; inv_depth: own range only;
 *                                        // after the marginalization calls: the state after double2vector()
 */
int lfvio_shard_begin(lfvio_ctx *ctx, const LfvioWindow *in, int lm_begin, int lm_end, int add_pose_side);
int lfvio_shard_exchange_len(void);
int lfvio_shard_scalar_offset(void);
double *lfvio_shard_exchange_ptr(lfvio_ctx *ctx);
int lfvio_shard_linearize(lfvio_ctx *ctx);
int lfvio_shard_solve(lfvio_ctx *ctx);
int lfvio_shard_candidate(lfvio_ctx *ctx);
/* *state: 0 = linearize next, 1 = step rejected (only a new candidate), 2 = terminated */
int lfvio_shard_decide(lfvio_ctx *ctx, int *state);
/* Stream-ordered form of the same loop (RCCL is stream-ordered: make lfvio_stream(ctx) the collective's stream).  One
 * pass is ALWAYS the same seven calls, no host synchronisation in between:
 *   lfvio_shard_enqueue(ctx, 0); all_reduce(buf[0 : len]);
 *   lfvio_shard_enqueue(ctx, 1); all_reduce(buf[scalar_offset : len]);
 *   lfvio_shard_enqueue(ctx, 2); all_reduce(buf[scalar_offset : len]);
 *   lfvio_shard_enqueue(ctx, 3);                        // decision + a 64-byte flag record to pinned memory
 * and lfvio_shard_poll() waits for the oldest record not yet read (returns 1 and the state of lfvio_shard_decide, 0 when
 * nothing is outstanding, < 0 on error), so one pass can be kept in flight behind the decision being read; a pass enqueued
 * behind a terminated loop does nothing.  lfvio_shard_restart() re-arms the resident shard from its uploaded state. */
int lfvio_shard_restart(lfvio_ctx *ctx);
int lfvio_shard_enqueue(lfvio_ctx *ctx, int phase);
int lfvio_shard_poll(lfvio_ctx *ctx, int *state);
int lfvio_shard_marg_linearize(lfvio_ctx *ctx, int flag);
int lfvio_shard_marg_finish(lfvio_ctx *ctx, int flag, LfvioPrior *out);
int lfvio_shard_finish(lfvio_ctx *ctx, LfvioSolution *out);

/* ---- multi-GPU groups: the 8-GPU path behind the C-ABI (SURVEY §8b `lfvio_create(int device_mask)`, §8e) -------------
 * The host side is C++ and the collective is RCCL (ncclAllReduce on each context's own stream), called by the library
 * itself; librccl is dlopen()ed when the first group is created (LFVIO_RCCL_LIB overrides the search), so a single-GPU
 * caller has no RCCL dependency.  Three ways to form a group:
 *   lfvio_group_create(mask)       ONE process (the ROS node: estimator.cpp:484 runs on one thread under m_estimator):
 *                                  one context + stream per device whose bit is set, ncclCommInitAll
 *   lfvio_group_create_rank(...)   one process per GPU (torchrun / mpirun): ncclCommInitRank; rank 0 obtains the id from
 *                                  lfvio_group_unique_id() and the launcher broadcasts its 128 bytes
 *   lfvio_group_create_local(...)  `shards` ranks on ONE device, the all-reduce a device-side sum in rank order instead
 *                                  of RCCL — for tests of the sharded window on a one-GPU box, and the form of choice for a
 *                                  resident BATCH that fills the device: two contexts take half of the windows each
 *                                  (lfvio_group_batch_*, no collective) and their streams run side by side
 * lfvio_group_solve() is what the re-implemented Estimator::optimization() calls in place of lfvio_solve() +
 * lfvio_marginalize(): every rank is handed the same window, rank r linearizes a contiguous landmark range balanced on
 * observation count (IMU factors and prior on rank 0), per trust-region pass the ranks sum-all-reduce
 * [H_pp | g_p | Schur sums | scalars], every rank solves the identical reduced system (no broadcast, identical
 * decisions), and the marginalization costs one more all-reduce; solution and prior come out identical on every rank,
 * sol->inv_depth complete.  Same error convention as the single-context calls; lfvio_group_last_error() has the text. */
#define LFVIO_UNIQUE_ID_BYTES 128
typedef struct lfvio_group lfvio_group;
lfvio_group *lfvio_group_create(unsigned device_mask);
int lfvio_group_unique_id(char id[LFVIO_UNIQUE_ID_BYTES]);
lfvio_group *lfvio_group_create_rank(int device, int rank, int world, const char id[LFVIO_UNIQUE_ID_BYTES]);
lfvio_group *lfvio_group_create_local(int device, int shards);
void lfvio_group_destroy(lfvio_group *g);
const char *lfvio_group_last_error(const lfvio_group *g);
int lfvio_group_size(const lfvio_group *g);     /* ranks of the group                       */
int lfvio_group_local(const lfvio_group *g);    /* contexts (ranks) held by this process    */
int lfvio_group_rank(const lfvio_group *g);     /* rank of the first local context          */
lfvio_ctx *lfvio_group_ctx(lfvio_group *g, int i); /* i-th local context (any single-context call may be made on it) */
const char *lfvio_group_backend(const lfvio_group *g); /* path of the RCCL library in use, or "local" */
/* optimization() of one window, landmark-sharded: upload + optimize + download, or the three steps on their own (the
 * window stays resident: lfvio_group_optimize() may be repeated, which is what bench.py times).  marg_flag < 0: the
 * trust-region solve only. */
/* One process per GPU (lfvio_group_create_rank): these calls contain collectives, and a collective completes only when every
 * rank of the group has entered it.  All ranks must therefore make the same sequence of lfvio_group_* calls with the same
 * window, the same marg_flag and the same choice of sol == NULL / != NULL (sol->inv_depth may be NULL on some ranks and not
 * on others: the gather of the inverse depths runs either way).
 * Collectives of lfvio_group_optimize(): per pass of the trust-region loop TWO sum-all-reduces — the reduced system (only what
 * shards over landmarks: the camera part of H_pp and of g_p, the Schur sums, 16 scalars and 256 partial sums: lfvio_group_payload_doubles() = 6 886
 * doubles, 55 KB; the speed / bias rows come from the IMU factors and the prior, which every rank evaluates for itself) and the
 * 16 scalars behind the candidate (its cost and the model terms; the landmark parts of the Gauss-Newton step's norms, which the dogleg
 * needs in between, every rank forms itself from the reduced Schur sums) — and one more
 * of the reduced system for the marginalization: 2 x passes + 1 (round 4: 3 x passes + 1 of the whole 151 KB buffer).
 * A rank that fails locally inside lfvio_group_optimize() (an enqueue refused, a HIP error) does NOT leave the others waiting: it
 * enqueues no more work but keeps issuing every collective of the sequence with an error word raised in its scalars; every rank
 * reads that word behind the pass's last reduction, ends its loop in the same pass and returns LFVIO_ERR_DEVICE (the failed rank:
 * its own error).  The group stays usable.  What this cannot cover: a rank whose device or RCCL communicator is gone (its
 * collectives cannot be issued at all), and errors before the first collective that are not common to all ranks (a malformed
 * window is refused by every rank alike; an allocation failure on one rank is not) — there the caller must still treat the
 * error as fatal for the group on every rank: RCCL has no timeout of its own. */
int lfvio_group_solve(lfvio_group *g, const LfvioWindow *in, int marg_flag, LfvioSolution *sol, LfvioPrior *prior);
int lfvio_group_upload(lfvio_group *g, const LfvioWindow *in);
int lfvio_group_optimize(lfvio_group *g, int marg_flag);
int lfvio_group_download(lfvio_group *g, LfvioSolution *sol, LfvioPrior *prior);
int lfvio_group_range(const lfvio_group *g, int rank, int *lm_begin, int *lm_end); /* landmark range of a rank */
int lfvio_group_last_passes(const lfvio_group *g);      /* passes / collectives of the last lfvio_group_optimize() */
int lfvio_group_last_collectives(const lfvio_group *g);
int lfvio_group_payload_doubles(void);                  /* doubles per rank in the all-reduce of a pass's reduced system: the camera part
                                                          * of H_pp and g_p, the Schur sums, 16 scalars, 256 partial sums (55 KB; the speed / bias rows do not travel) */
/* independent resident windows split over the devices of this process (BASELINE "512 independent windows"): slot s
 * lives on local context s % lfvio_group_local(); no data-path collective */
int lfvio_group_batch_reserve(lfvio_group *g, int batch, int max_landmarks, int max_observations);
int lfvio_group_batch_upload(lfvio_group *g, int slot, const LfvioWindow *in);
int lfvio_group_batch_optimize(lfvio_group *g, int count, int marg_flag);
int lfvio_group_batch_download(lfvio_group *g, int slot, LfvioSolution *sol, LfvioPrior *prior);

#ifdef __cplusplus
}
#endif
#endif /* LFVIO_H */
