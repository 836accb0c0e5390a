/*
 * lfvio_debug.h — parity / inspection hooks of liblfvio_hip.so used by tests/ only.
 * Not part of the drop-in boundary (include/lfvio.h).
 */
#ifndef LFVIO_DEBUG_H
#define LFVIO_DEBUG_H
#include "lfvio.h"
#ifdef __cplusplus
extern "C" {
#endif
/* Gauss-Newton blocks at the window's state, caller landmark order:
 * Hpp 172x172 row-major, gp 172, a/b N, W N x 73, cost. */
int lfvio_debug_linearize(lfvio_ctx *ctx, const LfvioWindow *in, double *Hpp, double *gp, double *a, double *b,
                          double *W, double *cost);
/* The Schur sums of a solve repeated with a new mu on the stored linearization (do_schur without do_lin) against a
 * full re-linearization at the same mu: largest absolute difference (expected 0). */
int lfvio_debug_schur_repeat(lfvio_ctx *ctx, const LfvioWindow *in, double mu, double *max_abs_diff);
/* Post-Schur system (A' n x n, b' n) of the last marginalization run on slot 0. */
int lfvio_debug_marg_system(lfvio_ctx *ctx, int n, double *A, double *b);
/* shader-clock stamps written by the last k_solve of slot 0 (bring-up instrumentation) */
int lfvio_debug_read_clocks(lfvio_ctx *ctx, long long *out32);
/* Average ms of `reps` launches of one pipeline kernel over slots [0,count) (HIP events on the context stream).
 * which: 0 k_lin (residual/Jacobian sweep + Schur SYRK of the landmark blocks), 2 k_sum (+ k_presum), 3 k_solve_dense;
 * 4 .. 7 k_setup by role, 8 .. 10 k_lin by role; a resident batch on the strip sweep: 12 k_linw, 13 k_solve_dense<true>, 14 k_stepw;
 * a large window group by group: 15 k_linb, 16 k_sumb, 17 k_backsub_wt (an error where the launch does not take that path). */
int lfvio_debug_time_kernel(lfvio_ctx *ctx, int which, int count, int reps, double *avg_ms);
/* Which kernel linearizes a launch over the resident slots [0, count): 0 k_lin (+ k_sum), 1 k_linw (a resident batch), 2 k_linb
 * (+ k_sumb: a large single window); < 0 on error.  bench.py asks before it times a sweep kernel. */
int lfvio_debug_sweep_kernel(lfvio_ctx *ctx, int count);
/* 0: launch kernels directly, 1: replay the captured hipGraph (default). */
int lfvio_debug_set_graph(lfvio_ctx *ctx, int on);
/* The launches a pass of few small windows saves by fusion.  on = 1 (default): the trust-region bookkeeping of a pass rides
 * in the prologue of the next pass's k_lin, and the dogleg step and the cost of its candidates are one launch (k_step).
 * on = 2: only the first of the two.  on = 0: neither — k_decide, k_dogleg and k_cost each as its own launch.  All three
 * routes must give bit-identical results. */
int lfvio_debug_set_decide_merge(lfvio_ctx *ctx, int on);
/* on != 0: the pseudo-inverse of the dropped block always comes from its eigen-decomposition (marginalization_factor.cpp:267-272);
   default: from a Cholesky factorization when every eigenvalue is provably far above eps, from the eigen-decomposition otherwise. */
int lfvio_debug_force_eig(lfvio_ctx *ctx, int on);
/* graph launches the last synchronous solve loop needed (1: every window was done within the first chunk of passes,
   and gauge fix + marginalization ran in the same graph) */
/* microseconds of the last upload: host packing | collecting a chained prior (after lfvio_batch_upload_chained_device: the graph launch of the
   lfvio_batch_optimize_begin that followed, which goes out behind work still running) | prior + copies enqueued | final synchronization */
int lfvio_debug_upload_times(lfvio_ctx *ctx, double *out4);
/* n > 0: every first graph of the synchronous entry points carries n passes instead of the most any of the last four calls needed; 0: adaptive */
int lfvio_debug_set_first_passes(lfvio_ctx *ctx, int n);
/* Solver::Options::function_tolerance of the windows uploaded from now on (Ceres' default 1e-6; estimator.cpp:810-822 leaves it
 * alone).  0 makes the loop run to its iteration cap or another criterion: the diagnostic of tests/tools/fuzz_parity.py, which
 * asks whether two solvers that disagree in the 6th digit of an inverse depth stopped early in a flat valley. */
int lfvio_debug_set_function_tolerance(lfvio_ctx *ctx, double tol);
/* Solver::Options::initial_trust_region_radius of the windows uploaded from now on (Ceres' default 1e4, which estimator.cpp:810-822
 * leaves alone; <= 0 restores it).  A small radius takes the dogleg through its Cauchy-point and interpolation cases from the first
 * iteration on — with the default they are only reached after a dozen rejected steps. */
int lfvio_debug_set_initial_radius(lfvio_ctx *ctx, double r);
/* 1 (default): the landmark role of k_lin runs eight lanes per track, 32 landmarks per workgroup, for windows of at most 320 landmarks
 * uploaded from now on; 0: four lanes, 64 landmarks, the form larger windows take.  Same sums in a different association.
 * Environment: LFVIO_LM_HALF. */
int lfvio_debug_set_lm_half(lfvio_ctx *ctx, int on);
/* The next lfvio_batch_upload_chained_device promises the device a prior of one row more than the marginalization in flight will leave:
 * k_prior_chain refuses it, the window runs without a prior, and the lfvio_batch_optimize_begin that follows returns LFVIO_ERR_DEVICE —
 * the path a failed marginalization takes, for tests. */
int lfvio_debug_break_next_chain(lfvio_ctx *ctx);
/* Where the strip sweep (kernels_linw.h) replaces the role-by-role one (k_lin + k_sum): 1 (default) for a resident batch whose
 * windows all carry a plan (k_linw: one workgroup per window, no partial sums through HBM) and for a single window — or a rank's
 * share of a sharded one — of at least 40 960 landmarks (k_linb + k_sumb: one workgroup per group of strips); 0 never; 2 for every
 * launch, however few or small the windows (<= 320 landmarks: k_linw, more: k_linb; tests).  Applies to windows uploaded
 * afterwards.  Environment: LFVIO_LINW. */
int lfvio_debug_set_linw(lfvio_ctx *ctx, int mode);
/* 1: the reduced pose system is solved along its block structure — the speed/bias chain eliminated block by block, a
 * dense 73-wide camera block left (k_solve_block, 78 KB of LDS: two windows of a batch per CU) — where every window of the launch
 * has that structure (a prior with no SpeedBias block but frame 0's: what the reference's marginalization produces); 0 (default:
 * the block form measured slower on MI355X): always the dense 172 x 172 solve (k_solve_dense).  Same semantics, a different elimination order.  Environment: LFVIO_BLOCK_SOLVE. */
int lfvio_debug_set_block_solve(lfvio_ctx *ctx, int on);
/* 1: a launch over the resident slots [0, count) takes k_solve_block, 0: k_solve_dense; < 0 on error */
int lfvio_debug_solve_kernel(lfvio_ctx *ctx, int count);
/* One linearization + dense solve of the resident slots [0, count) by the path the launch takes; then, of slot `slot`: g_p[172],
 * the Schur sums (15 x 256, tile layout), lm_sum[5], a[N], b[N] (device landmark order), the pose-side Gauss-Newton step [172], the
 * dogleg model's quadratic forms [16], the cost.  Any output may be NULL.  Returns 1 if k_linw ran, 2 if k_linb + k_sumb, 0 if k_lin + k_sum,
 * < 0 on error. */
int lfvio_debug_resident_pass(lfvio_ctx *ctx, int count, int slot, double *gp, double *schur, double *lm_sum, double *a, double *b, double *gn_p, double *q,
                              double *x_cost);
int lfvio_debug_last_chunks(lfvio_ctx *ctx);
/* passes of the trust-region loop the slowest window of the last synchronous call used */
int lfvio_debug_last_passes(lfvio_ctx *ctx);
/* out3 = {passes, iterations} of the last synchronous call on one window and the number of speculative candidates per pass the next one
 * will prepare (3, or 4 where a pass of the previous call covered two iterations or more) */
int lfvio_debug_speculation(lfvio_ctx *ctx, int *out3);
/* The marginalization run ahead of the loop's end (csrc/kernels_spec.h).  on = 0: the windows uploaded from now on end with the serial
 * tail (gauge fix, frame-0 sweep, k_marg_solve behind the last pass); 1 (default): a one-window context of at most 320 landmarks starts
 * the marginalization of every newly accepted state on a second stream.  Either way the prior is the same bits.
 * out2 (may be NULL) = {calls that started workers, priors a worker delivered} since the context was created. */
int lfvio_debug_marg_ahead(lfvio_ctx *ctx, int on, long long *out2);
/* tests: the local context `local_ctx` of the group reports a failure when it enqueues phase `phase` (0 the sweep, 1 solve + back-
 * substitution, 2 step + candidate cost, 3 bookkeeping) of pass `pass` of the next lfvio_group_optimize(); local_ctx < 0 clears it.
 * The call must still issue every collective of its sequence (the peers of a real group are waiting in them), end the loops of all
 * ranks in the same pass and return the error. */
int lfvio_debug_group_inject_failure(lfvio_group *g, int local_ctx, int pass, int phase);
#ifdef __cplusplus
}
#endif
#endif
