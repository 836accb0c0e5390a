/*
 * lfvio_debug.h — parity / inspection hooks of liblfvio_hip.so used by tests/, tools/ and bench.py only.
 * Not part of the drop-in boundary (include/lfvio.h).  Nine entry points: two that take a key, seven that look inside.
 */
#ifndef LFVIO_DEBUG_H
#define LFVIO_DEBUG_H
#include "lfvio.h"
#ifdef __cplusplus
extern "C" {
#endif
/* Every switch of the library in one call.  LFVIO_ERR_ARG for an unknown key or a value out of range.
 *   "graph"              0: launch kernels directly, 1 (default): replay the captured hipGraphs
 *   "first_passes"       n > 0: every first graph of the synchronous entry points carries n passes instead of the most any of the last
 *                        four calls needed; 0: adaptive again
 *   "spec_count"         n > 0: a fixed number of speculative candidates per pass (1 .. 4); 0: from the call before (3, or 4)
 *   "function_tolerance" Solver::Options::function_tolerance of the windows uploaded from now on (Ceres' default 1e-6, which
 *                        estimator.cpp:810-822 leaves alone); 0 makes the loop run to its iteration cap or another criterion
 *   "initial_radius"     Solver::Options::initial_trust_region_radius of the windows uploaded from now on (Ceres' default 1e4; <= 0
 *                        restores it).  A small radius takes the dogleg through its Cauchy-point and interpolation cases at once
 *   "linw"               where the strip sweep (kernels_linw.h) replaces the role-by-role one (k_lin + k_sum): 1 (default) for a resident
 *                        batch whose windows all carry a plan (k_linw) and for a single window — or a rank's share — of at least 40 960
 *                        landmarks (k_linb + k_sumb); 0 never; 2 for every launch, however few or small the windows (tests).  With the
 *                        next upload
 *   "lm_half"            1 (default): the landmark role of k_lin runs eight lanes per track, 32 landmarks per workgroup, for windows of
 *                        at most 320 landmarks uploaded from now on; 0: four lanes, 64 landmarks, the form larger windows take
 *   "force_eig"          != 0: the pseudo-inverse of the marginalization's dropped block always from its eigen-decomposition
 *                        (marginalization_factor.cpp:267-272); default: from a Cholesky factorization when every eigenvalue is provably
 *                        far above eps
 *   "marg_ahead"         0: the windows uploaded from now on end with the serial tail (gauge fix, frame-0 sweep, k_marg_solve behind the
 *                        last pass); 1 (default): a one-window context of at most 320 landmarks starts the marginalization of every
 *                        newly accepted state on worker streams (csrc/kernels_spec.h).  Either way the prior is the same bits
 *   "break_next_chain"   1: the next lfvio_batch_upload_chained_device promises the device a prior of one row more than the
 *                        marginalization in flight will leave — the path a failed marginalization takes, for tests
 *   "env"                apply LFVIO_DEBUG="key=value,key=value" from the environment (the product entry points read none) */
int lfvio_debug_configure(lfvio_ctx *ctx, const char *key, double value);
/* What the last calls left, as doubles into out[0 .. n).
 *   "last_call"     {passes of the trust-region loop the slowest window of the last synchronous call used, iterations they covered,
 *                   graph launches of that call (1: everything ran in the first graph), speculative candidates the next call prepares}
 *   "marg_ahead"    {calls that started workers, priors a worker delivered} since the context was created
 *   "upload_times"  microseconds of the last upload: host packing | collecting a chained prior | prior + copies enqueued | final wait
 *   "sweep_kernel"  in out[0]: a count of resident slots; out[0]: the kernel that linearizes a launch over them — 0 k_lin (+ k_sum),
 *                   1 k_linw (a resident batch), 2 k_linb (+ k_sumb: a large single window) */
int lfvio_debug_query(lfvio_ctx *ctx, const char *key, double *out, int n);
/* Gauss-Newton blocks at the window's state, caller landmark order:
 * Hpp 172x172 row-major, gp 172, a/b N, W N x 73, cost. */
int lfvio_debug_linearize(lfvio_ctx *ctx, const LfvioWindow *in, double *Hpp, double *gp, double *a, double *b,
                          double *W, double *cost);
/* The Schur sums of a solve repeated with a new mu on the stored linearization (do_schur without do_lin) against a
 * full re-linearization at the same mu: largest absolute difference (expected 0). */
int lfvio_debug_schur_repeat(lfvio_ctx *ctx, const LfvioWindow *in, double mu, double *max_abs_diff);
/* Post-Schur system (A' n x n, b' n) of the last marginalization run on slot 0 by the loop's own stream. */
int lfvio_debug_marg_system(lfvio_ctx *ctx, int n, double *A, double *b);
/* shader-clock stamps written by the profiling builds of the kernels of slot 0 (bring-up instrumentation) */
int lfvio_debug_read_clocks(lfvio_ctx *ctx, long long *out32);
/* Average ms of `reps` launches of one pipeline kernel over slots [0,count) (HIP events on the context stream).
 * which: 0 k_lin (residual/Jacobian sweep + Schur SYRK of the landmark blocks), 2 k_sum (+ k_presum), 3 k_solve_dense;
 * 4 .. 7 k_setup by role, 8 .. 10 k_lin by role; a resident batch on the strip sweep: 12 k_linw, 13 k_solve_dense<true>, 14 k_stepw;
 * a large window group by group: 15 k_linb, 16 k_sumb, 17 k_backsub_wt (an error where the launch does not take that path). */
int lfvio_debug_time_kernel(lfvio_ctx *ctx, int which, int count, int reps, double *avg_ms);
/* One linearization + dense solve of the resident slots [0, count) by the path the launch takes; then, of slot `slot`: g_p[172],
 * the Schur sums (15 x 256, tile layout), lm_sum[5], a[N], b[N] (device landmark order), the pose-side Gauss-Newton step [172], the
 * dogleg model's quadratic forms [16], the cost.  Any output may be NULL.  Returns 1 if k_linw ran, 2 if k_linb + k_sumb, 0 if k_lin + k_sum,
 * < 0 on error. */
int lfvio_debug_resident_pass(lfvio_ctx *ctx, int count, int slot, double *gp, double *schur, double *lm_sum, double *a, double *b, double *gn_p, double *q,
                              double *x_cost);
/* tests: the local context `local_ctx` of the group reports a failure when it enqueues phase `phase` (0 the sweep, 4 solve ..
 * candidate cost, 3 bookkeeping) of pass `pass` of the next lfvio_group_optimize(); local_ctx < 0 clears it.
 * The call must still issue every collective of its sequence (the peers of a real group are waiting in them), end the loops of all
 * ranks in the same pass and return the error. */
int lfvio_debug_group_inject_failure(lfvio_group *g, int local_ctx, int pass, int phase);
#ifdef __cplusplus
}
#endif
#endif
