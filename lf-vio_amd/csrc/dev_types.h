// dev_types.h — HBM layout of one resident sliding window ("slot") and the
// trust-region state that lives on the device.  Shared by host packing code and kernels.
//
// One slot = one contiguous device blob:  [Slot header | input arrays | work arrays].
// The host builds header + inputs in pinned memory and uploads them with ONE copy;
// every pointer inside the header is a device address into the same blob.
//
// Landmark order on the device is NOT the caller's: landmarks are bucket-sorted by
// (start_frame, track length) so that (a) landmarks anchored at frame 0 — the ones
// marginalization drops — form a prefix, (b) neighbouring lanes take the same trip
// count in the per-landmark loops, (c) the MFMA Schur SYRK sees runs of rows with the
// same zero pattern.  `lm_perm` maps device order -> caller order.
// Observations are SoA, landmark-major in device order (coalesced 8-byte lanes); a second
// index list orders the non-anchor observations by frame pair (i, j) for the Gram sweep.
#pragma once
#include "../../include/lfvio.h"

constexpr int KC = 73;     // camera-side tangent dim: 11 poses * 6 + ex 6 + td 1
constexpr int KP = 172;    // + 11 speed/bias * 9
constexpr int WLD = 80;    // width of a W row in the LDS tile (5 MFMA column tiles of 16); in HBM a row is stored over its
                           // non-zero span: [6 cnt entries of frames start .. start + cnt - 1 | ex 6 | td | b_l | b_l kappa_l]
__host__ __device__ inline int w_row_len(int cnt) { return 6 * cnt + 9; }
constexpr int COL_B = 73;  // W pad column holding b_l
constexpr int COL_K = 74;  // W pad column holding b_l * kappa_l (Cauchy-point cross term)
constexpr int NQ = 105;    // 14x14 upper triangle (basis Gram)
constexpr int NG = 210;    // 20x20 upper triangle (expanded pair Gram incl. residual column)
constexpr int NGP = 212;   // + cost, padded
constexpr int NPAIR = 121; // pair slot = i * 11 + j
constexpr int NT = 15;     // upper tiles of the 5x5 tiling of the 80x80 Schur accumulator
constexpr int SCHUR_LEN = NT * 256;
constexpr int PACKED = KP * (KP + 1) / 2;
constexpr int JMAX_SWEEPS = 20;
constexpr int PRE_CHUNK_LIMIT = 2 * 121;  // windows with more Gram chunks reduce them per frame pair first (k_presum)
constexpr int SUM_ITEMS_CAP = PRE_CHUNK_LIMIT * 209 + 64;
// Entries of H_pp / g_p that have a visual (Gram) part, i.e. a gather list: rows < KC — the packed prefix and g_p[0, KC).
// The list bounds are stored by this compact index (0 .. SUM_VIS), not by the packed index: 11 KB per upload, not 60.
constexpr int SUM_VIS_PACKED = 73 * 74 / 2, SUM_VIS = SUM_VIS_PACKED + 73;
constexpr int HPP_CAP = 16384;
// exchange buffer of the landmark-sharded mode (one contiguous sum-all-reduce):
//   [ H_pp packed | g_p | Schur sums (80x80 upper tiles) | 16 scalars ]
constexpr int XOFF_H = 0;
constexpr int XOFF_G = 14880;                 // PACKED rounded up to even
constexpr int XOFF_S = XOFF_G + 176;          // KP rounded up
constexpr int XOFF_C = XOFF_S + 15 * 256;
constexpr int XCH_LEN = XOFF_C + 16;
// behind the scalars (lfvio_group only; not part of lfvio_shard_exchange_len()): XP_WGS partial sums of [sum c_l b_l^2, clamp count] — one pair
// per workgroup of k_lm_cb2 (kernels_solve.h); they ride in the all-reduce of the reduced system and every rank adds them up in its solve
constexpr int XP_WGS = 128, XOFF_P = XCH_LEN, XCH_ALLOC = XCH_LEN + 2 * XP_WGS;
enum { XS_COST = 0, XS_G2, XS_ASV2, XS_LAM2, XS_BMAX, XS_GN2, XS_GGN, XS_CCOST, XS_MLIN, XS_MQUAD, XS_DN, XS_XN, XS_N0, XS_ERR };  // XS_ERR: a rank of an lfvio_group that failed locally raises it in every collective it still issues (group.inc)   // packed H_pp (14878), reused as dense scratch by the marginalization
constexpr int LM_BLOCK = 64;     // landmarks per workgroup (one wave) in the landmark sweep
constexpr int CHUNK_LANES = 64;
constexpr int CHUNK_MAX = 64;    // observations per Gram chunk: one wave pass; a workgroup of k_lin takes 4 chunks
constexpr int SCHUR_LM = 64;      // landmarks per Schur SYRK part = one landmark block of k_lin
constexpr int LMS = 16;          // per-block landmark scalar partials
constexpr int IMU_RAW = 450 + 16;  // 15 x 30 Jacobian of (pose_i, sb_i, pose_j, sb_j) tangents, 15 residuals (+1 pad)
constexpr int IMU_OUT = 900 + 30 + 2;

__host__ __device__ inline int off_pose(int f) { return 6 * f; }
__host__ __device__ inline int off_ex() { return 66; }
__host__ __device__ inline int off_td() { return 72; }
__host__ __device__ inline int off_sb(int f) { return 73 + 9 * f; }
__host__ __device__ inline int pidx(int i, int j) { return i >= j ? i * (i + 1) / 2 + j : j * (j + 1) / 2 + i; }

struct FrameState {
  double pose[LFVIO_NUM_FRAMES][7];
  double sb[LFVIO_NUM_FRAMES][9];
  double ex[7];
  double td;
};

// Quantities that are uniform per frame / per frame pair at one linearization point.
//
// Two flavours of the back-rotation.  The reference rotates back with Eigen's Quaternion::inverse() = conjugate / |q|^2 in the
// RESIDUAL chain (`Qj.inverse() * (pts_w - Pj)`, `qic.inverse() * (pts_imu_j - tic)`: projection_td_factor.cpp:59-60,
// projection_factor.cpp:38-39) and with the transposed rotation matrices in the JACOBIANS (:101-146).  For a unit quaternion the two
// are one matrix; the start point of a call can hold a quaternion off the unit sphere (the newest frame is IMU-propagated with
// unnormalised delta quaternions, estimator.cpp:107-116; an extrinsic rotation typed into a config file), and there
// R(q^-1) = (1 - s^2) I + s^2 R^T, s = 1 / |q|, is not R^T.  `T` and `c` are the residual chain's: for a pair (i, j) whose frame j
// or extrinsic quaternion is off the sphere (bit j / bit 11 of the mask below) they are formed from R(q^-1), for every other pair
// from the transposes as the Jacobian tables are — the same arithmetic as before the distinction existed.  What the Jacobians of
// an off-sphere pair need beyond M1 / M2 lives in entries of the pair tables that no pair uses (only i < j is a pair):
//   c[j * 11 + i]   the Jacobian flavour of c for the pair (i, j):  ric^T (Rj^T (Ri tic + Pi - Pj) - tic)   (:131)
//   T[0]            ricF = R(qic^-1)^-1: pts_imu_j of the residual chain from pts_camera_j (= ric for a unit qic)
//   c[0][0]         the mask, as a double (0.0: every quaternion of this point is on the sphere — the only word a kernel reads then)
struct Tab {
  double R[11][9], P[11][3];
  double ric[9], ricT[9], tic[3];
  double M1[11][9];     // ric^T Rj^T
  double M2[NPAIR][9];  // ric^T Rj^T Ri
  double T[NPAIR][9];   // residual chain: X_cj = T X_ci + c;  ric^T Rj^T Ri ric on the sphere
  double c[NPAIR][3];   // ric^T (Rj^T (Ri tic + Pi - Pj) - tic) on the sphere
};
constexpr int TAB_EX_BIT = 11;                      // mask bit of the extrinsic quaternion
constexpr double TAB_OFF_SPHERE = 1e-14;            // | |q|^2 - 1 | above this is "off": a normalised quaternion is within 1e-15, and a
                                                    // defect e moves the linearization by <= ~500 e (depth / baseline), bar 1e-10
__host__ __device__ inline int tab_cj(int pair) { return (pair % 11) * 11 + pair / 11; }  // where the pair's Jacobian-flavour c lives

// indices into TRState::q (pose-side scalars produced by k_solve / k_dogleg)
enum {
  Q_GG = 0,    // G^T H_pp G   (G = unscaled Cauchy direction, pose side)
  Q_GN,        // G^T H_pp N   (N = unscaled Gauss-Newton direction)
  Q_NN,        // N^T H_pp N
  Q_gG,        // g_p . G
  Q_gN,        // g_p . N
  Q_GRAD_SQ,   // ||gradient_||^2, pose side (D-scaled space)
  Q_GN_SQ,     // ||gauss_newton_step_||^2, pose side
  Q_GRAD_GN,   // gradient_ . gauss_newton_step_, pose side
  Q_ZG,        // z-cross terms: sum_l G_l (w_l . G_c) is carried per landmark (d1/d2), unused slot
  Q_LGN,       // lfvio_group: landmark part of ||gauss_newton_step_||^2 from the reduced Schur sums (solve_body) ...
  Q_LGG,       // ... and of gradient_ . gauss_newton_step_
  Q_LEX,       // 1: the two above were formed in this solve
  Q_COUNT = 16
};

// The scalar part of the trust-region state.  TRHead and TRState share it as a common initial sequence, so the single-lane
// bookkeeping kernels can copy it to registers in one batch of loads (reinterpret_cast<TRHead *>) while Slot stays a
// standard-layout type for offsetof on the host.
// Speculative trust-region candidates (small windows): the radii Ceres tries after rejected steps are known in advance
// (radius / 2 each time), so one pass can evaluate the step for radius, radius / 2, radius / 4 and k_decide walks through
// them in Ceres' order.  Candidate 0 lives in x[cur ^ 1] / tab[cur ^ 1] / lam[cur ^ 1] as always, candidates 1, 2 in
// the *E slots and are copied over when one of them is the accepted step.
#ifndef LFVIO_SPEC_EXTRA
#define LFVIO_SPEC_EXTRA 3
#endif
constexpr int SPEC_EXTRA = LFVIO_SPEC_EXTRA, SPEC_MAX_LM = 320;
constexpr int WT_PAIRS = (KC + 1) / 2;
constexpr int SPEC_OWN = 72;  // accepted steps of a call + 1 (max_num_iterations is capped at LFVIO_MAX_TRACE = 64)
enum { SPEC_FREE = 0, SPEC_SIDE = 1, SPEC_MAIN = 2, SPEC_COMMIT = 3, SPEC_ABANDON = 4 };
enum { FIN_OPEN = 0, FIN_CLOSING = 1, FIN_MAIN = 2, FIN_SIDE = 3 };  // low bits of SpecCtl::fin; the final count of accepted steps sits above them (<< 2)
constexpr int MAIL_MAX_LM = 8192;  // landmarks a window may have for its solution to travel through the mailbox (Slot::mail)
#define TR_HEAD_FIELDS \
  double radius, mu, x_cost, x_norm, cand_cost, model_cost_change, dogleg_step_norm, alpha; \
  double cg, cn; \
  double gn_sq_total, grad_sq_total, grad_gn_total; \
  double step_sq_pose; \
  double xn2_pose_cand; \
  double gmax_pose, lm_bmax; \
  double initial_cost; \
  double function_tolerance; \
  double q[Q_COUNT]; \
  int iteration, cur, do_lin, do_schur, done, termination, chol_fail, scaled; \
  int num_succ, num_unsucc, consec_invalid, trace_len, step_valid, skip_step, error, new_point; \
  int spec_n, gn_unconfirmed; /* lfvio_group: this pass's candidate is the Gauss-Newton step and k_decide has yet to check its norm (dogleg_body) */ \
  double cgE[SPEC_EXTRA], cnE[SPEC_EXTRA], snE[SPEC_EXTRA], step_sqE[SPEC_EXTRA], xn2E[SPEC_EXTRA];
struct TRHead {
  TR_HEAD_FIELDS
};
struct TRState {
  TR_HEAD_FIELDS
  LfvioIterationSummary it;
  LfvioIterationSummary trace[LFVIO_MAX_TRACE];
};
// layout of the mailbox (Slot::mail), the same order as the pinned download block of the host side: the flag word, both
// state slots (only x[cur] is written), the trust-region header with the trace, both inverse-depth buffers (only [cur])
// Behind them the prior of the gated marginalization (an LfvioPrior, of which the header, n x n Jacobian entries and n
// residuals are written), announced by the second flag word; words 2 and 3 carry Slot::passes_used and the iteration count of the call.
constexpr size_t MAIL_X = 64, MAIL_TR = MAIL_X + 2 * sizeof(FrameState), MAIL_LAM = (MAIL_TR + sizeof(TRState) + 63) / 64 * 64,
                 MAIL_LAM_STRIDE = (size_t)MAIL_MAX_LM * 8, MAIL_PRIOR = MAIL_LAM + 2 * MAIL_LAM_STRIDE, MAIL_BYTES = MAIL_PRIOR + sizeof(LfvioPrior);

// What the trust-region bookkeeping (k_decide) changes in the header.  In the passes of a graph that follow another pass the
// bookkeeping rides in the prologue of k_lin (every workgroup repeats it, none of them may write the header the others
// are reading): workgroup 0 leaves the outcome here with dec_pending set, k_sum / k_presum take their flags from it, and
// k_solve — one workgroup per slot — moves it into the header (kernels_lin.h MODE_DECIDE, kernels_solve.h commit_decision).
struct TRDecision {
  double radius, mu, x_cost, x_norm, cand_cost, model_cost_change;
  int iteration, cur, do_lin, do_schur, done, termination, chol_fail, num_succ, num_unsucc, consec_invalid, trace_len, skip_step;
  int acc_z, pad_;  // accepted candidate slot (0: the regular one)
};

// the eight int flags at the head of the int block of TRHead, fetched with ONE load: the guards at the top of every
// kernel of the loop test two or three of them, and each separate (dependent, short-circuited) load is a memory round trip
struct __attribute__((aligned(8))) TRFlags {
  int iteration, cur, do_lin, do_schur, done, termination, chol_fail, scaled;
};
__device__ __forceinline__ TRFlags tr_flags(const TRState *tr) { return *reinterpret_cast<const TRFlags *>(&tr->iteration); }

// Window-resident linearization of a resident batch (kernels_linw.h): one workgroup owns a window.  The landmarks of one
// start frame (contiguous in device order) are cut into strips of at most 64; a wave takes whole strips — lane = landmark,
// step o = the strip's observations in frame start + o, which all belong to ONE frame pair — so every pose-dependent
// quantity of a step is wave-uniform and the step's <= 64 observations are a complete SYRK operand.
constexpr int LINW_MAX_STRIPS = 16, LINW_WAVES = 4;
struct LinwPlan {
  int ok;         // the window fits the plan (N <= SPEC_MAX_LM, strips <= LINW_MAX_STRIPS) and its extra arrays are uploaded
  int n_strips;
  int wave_first[LINW_WAVES + 1];     // wave w takes strips [wave_first[w], wave_first[w + 1]); all strips of a start on one wave
  short lm0[LINW_MAX_STRIPS], nlm[LINW_MAX_STRIPS], start[LINW_MAX_STRIPS], kmax[LINW_MAX_STRIPS];
  int pair_obs0[NPAIR + 1];           // first pair-major observation of every frame pair (prefix sums)
  int firstl[LFVIO_NUM_FRAMES][12];   // [start][o]: first landmark (device order) of that start frame with more than o observations
  // A large single window (k_linb, kernels_linw.h): the same strips, four of ONE start frame to a workgroup ("group"), as many
  // groups as the window has; strips, waves and lm0 .. kmax above are not used then
  int big;        // 1: the plan is a group list (Slot::linb_lm0 / linb_ns), ng groups; Wt rows are wt_ld apart
  int ng, wt_ld, pad_;
};

struct MargPlan {  // structure of the marginalization, computed on the host at upload
  int valid;       // 0: nothing to do (MARGIN_SECOND_NEW without a prior touching Pose[9])
  int m15;         // dropped pose-side dims (15 for MARGIN_OLD, 6 for SECOND_NEW)
  int n;           // kept dims
  int nb;          // kept blocks
  int N0;          // landmarks to eliminate (device-order prefix)
  int nChunks0;    // Gram chunks to sweep (pairs (0, j))
  int use_imu0, use_visual;
  int col[KP];     // tangent column -> column of A ([0,m15) dropped, [m15, m15+n) kept, -1 absent)
  int kind[LFVIO_MAX_PRIOR_BLOCKS], frame[LFVIO_MAX_PRIOR_BLOCKS], shifted_frame[LFVIO_MAX_PRIOR_BLOCKS];
  int idx[LFVIO_MAX_PRIOR_BLOCKS];
};

// Device arrays of the slot blob, addressed from inside the blob.  A pointer loaded from memory is "generic" to the
// compiler, which then emits flat_load / flat_store: those count on both the vector-memory and the LDS counters, so
// every LDS wait also waits for the HBM traffic in flight.  GP<T> therefore stores the distance from itself to the
// array: the address is formed from the Slot pointer, which descends from a kernel argument, so the address-space
// inference selects global_* instructions.  (Self-relative: never copy a GP out of its Slot.)
template <class T>
struct GP {
  long long off;
  __device__ __forceinline__ operator T *() const { return (T *)((char *)this + off); }
  // host: slot = start of the (staging copy of the) Slot this member lives in, target = byte offset inside the blob
  void set(const void *slot, size_t target) { off = (long long)target - (long long)((const char *)this - (const char *)slot); }
};

struct Slot {
  // ---------------- header: sizes, flags, constants
  int N, M, NV, nLmBlocks, nChunks, nSchurParts, est_ex, est_td;
  int max_iter, prior_valid, prior_n, prior_nb;
  int tail_state, passes_used, iters_done, chain_err;  // chain_err: the prior this window was to take over on the device (k_prior_chain) was not there  // passes_used: passes of the loop that began with this slot still open (k_lin)  // gated gauge fix + marginalization of this call: 0 not run, 2 finished (kernels_lin.h, MODE_GATED)
  int ex_fixed_off;              // the extrinsic quaternion is off the unit sphere and not estimated (k_setup): every table of the call carries its bit (struct Tab)
  int lm_half;                   // the landmark role of k_lin runs 8 lanes per track, 32 landmarks per workgroup (windows of at most SPEC_MAX_LM landmarks)
  int schur_lm, sharded;         // sharded: this slot holds only a landmark range of the window (multi-GPU)
  int pose_side, pre_gram;       // sharded: this rank adds the IMU + prior factors; pre_gram: gather lists index pairG
  int spec_on, wt_clean;         // wt_clean: Slot::Wt is zero outside the landmarks' spans (k_setup cleared it, k_linw has swept since; 0 after every upload)
                                 // 1: this is slot 0 of a context that keeps a shadow slot behind its last one — the marginalization may be run ahead
                                 // of the loop's end on a second stream (kernels_spec.h); set by the upload
  int dec_pending, mail_seq;     // dec holds a decision k_solve has not moved into the header yet; mail_seq: what the mailbox flags are set to (the upload's sequence number, never 0)
  // Early hand-over of the solution (lfvio_batch_optimize_begin): host memory the device writes directly — 0, or the
  // mailbox [flag | x[2] | TRState | lam[0] | lam[1]] (MAIL_* below) of the context.  The gated gauge fix ends by copying
  // the state it has just re-anchored there and raising the flag, so the caller has its poses while the marginalization
  // of the same graph is still running.
  long long mail;
  TRDecision dec;
  double g[3], tr_over_row, half_row, sqrt_info;
  double init_radius;  // Solver::Options::initial_trust_region_radius: Ceres' default 1e4 unless lfvio_debug_set_initial_radius() changed it
  double fn_tol;  // Solver::Options::function_tolerance of this window: Ceres' default 1e-6 unless lfvio_debug_set_function_tolerance() changed it
  FrameState x0;
  LfvioPreintegration imu[LFVIO_WINDOW_SIZE];
  int imu_active[LFVIO_WINDOW_SIZE];
  int prior_kind[LFVIO_MAX_PRIOR_BLOCKS], prior_frame[LFVIO_MAX_PRIOR_BLOCKS], prior_idx[LFVIO_MAX_PRIOR_BLOCKS];
  double prior_x0[LFVIO_MAX_PRIOR_BLOCKS][9];
  int prior_cmap[KP];            // prior column -> tangent column
  int prior_inv[KP];             // tangent column -> prior column (-1: not in the prior)
  int pair_chunk0[NPAIR + 1];    // chunk range per pair slot
  MargPlan marg[2];              // [MARGIN_OLD, MARGIN_SECOND_NEW]
  LinwPlan linw;
  // ---------------- input arrays (device pointers into the blob)
  GP<int> lm_start, lm_cnt, lm_obs0, lm_perm;
  GP<int> lm_woff;                 // [N + 1] offset of landmark l's row in W: rows are stored over their non-zero span only
  GP<double> lam0;
  GP<double> obs[8];                // px py pz vx vy vz cur_td uv_y, each [M]
  GP<int> pm_obs, pm_lm;           // [NV] pair-major: observation index, landmark index
  GP<double> anc[8];                // [N] the anchor observation of every landmark, SoA in device order (k_linw)
  GP<double> pmo[8];                // [NV] the non-anchor observations, SoA in pair-major order (k_linw: a strip step reads 64 consecutive ones)
  GP<unsigned char> pm_pair;        // [NV] frame pair (i * 11 + j) of every pair-major observation (k_stepw: one lane per observation)
  GP<int> linb_lm0, linb_ns;        // [ng] k_linb's groups: first landmark; landmarks | start frame << 16 (longest first)
  GP<int> chunk_pair, chunk_begin, chunk_end;
  GP<double> prior_J, prior_r;     // n*n, n
  GP<int> sum_off, sum_end_marg, sum_items;  // gather lists of k_sum: per H_pp / g_p entry, offsets into gram_part (or pairG)
  // ---------------- work arrays
  FrameState x[2];
  TRState tr;                    // directly after x[]: one small D2H copy fetches state + trace
  Tab tab[2];
  GP<double> lam[2];
  FrameState xE[SPEC_EXTRA];     // speculative candidates 1, 2 (see SPEC_EXTRA)
  Tab tabE[SPEC_EXTRA];
  GP<double> lamE[SPEC_EXTRA];      // SPEC_MAX_LM each
  GP<double> cost_partE;            // SPEC_EXTRA * (SPEC_MAX_LM / 64) * LMS
  double pose_costE[SPEC_EXTRA][16];
  double imu_sqrt[LFVIO_WINDOW_SIZE][225];
  GP<double> prior_A;               // n*n  (J0^T J0)
  double prior_b0[KP];           // J0^T r0
  GP<double> a, b, W, scale_l, grad_l, gn_l, diag_l, einv_l, d1, d2;
  GP<double> Wt;  // [WT_PAIRS][SPEC_MAX_LM][2]: windows of at most SPEC_MAX_LM landmarks — the rows of W once more, dense and
                  // transposed (columns 2p, 2p + 1 of every landmark side by side), for the back-substitution inside
                  // k_dogleg / k_step: one landmark per lane reads its row as WT_PAIRS coalesced 16-byte loads instead of KC
                  // lines per lane
  GP<double> gram_part, pairG, schur_part, schur_sum;   // schur_sum = xch + XOFF_S
  GP<double> xch, gp;              // exchange buffer; gp = xch + XOFF_G, Hpp = xch + XOFF_H
  GP<double> lm_part;               // nLmBlocks * LMS
  double lm_sum[LMS];
  GP<double> cost_part;             // nLmBlocks * LMS (candidate sweep)
  GP<double> imu_out;               // 10 * IMU_OUT
  GP<double> imu_raw;               // 10 * IMU_RAW: unweighted residual and Jacobian per factor (k_imu_raw, resident batches)
  double prior_g[KP + 4];        // prior gradient (tangent cols) + cost
  double pose_cost[16];          // candidate costs of imu[0..9], prior [10]
  GP<double> Hpp;                   // packed lower KP (assembled by k_sum) = xch + XOFF_H
  GP<double> mscr;                  // dense scratch of the marginalization (HPP_CAP)
  GP<double> eig_aux;               // eigenvalues[96] | sorted b'[96] | (int) diagonal-sort permutation[96]
  double scale_p[KP], diag_p[KP], grad_p[KP], gn_p[KP], step_p[KP];
  double uc_grad[WLD], uc_gn[WLD];  // camera-side Cauchy / Gauss-Newton directions (unscaled), zero-padded to WLD
  // The marginalization run ahead of the loop's end (kernels_spec.h).  `spec`: the words the loop (stream 0) and the workers on
  // the second stream meet at, in the slot being solved; `shadow`: a worker's own record, in the shadow slot it computes in.
  struct SpecCtl {
    int ep;          // call counter of this slot (k_setup), the tag in the upper halves of `word` and `fin`
    int word;        // ep << 16 | accepted steps << 1 | cur: the newest accepted state that is complete in x[cur] / lam[cur] (k_sum); 0: none yet
    int fin;         // ep << 16 | state: 0 loop open, 1 closing (the gauge fix is about to rewrite x[cur] in place), FIN_* the marginalization's owner
    int ticket;      // the caller's ticket of this call (mailbox word 6, read by k_setup): the worker that delivers the prior echoes it into word 7
    int own[SPEC_OWN];  // per accepted-step count: who forms the prior of that state (SPEC_FREE / _SIDE / _MAIN / _COMMIT / _ABANDON)
  } spec;
  struct SpecShadow {
    int hdr_ep;      // the header of the window was copied for this call
    int word;        // what the worker of this round took (0: nothing: its kernels return)
    int last_word;   // the state the last round worked on (finished, or given up for a newer one)
    int ticket;      // SpecCtl::ticket of the call the round works for
  } shadow;
  long long dbg[32];
  double jtrace[32];             // second half of lfvio_debug_read_clocks' record (bring-up instrumentation)
  // marginalization outputs
  LfvioPrior prior_out;
};
