// Trust-region bookkeeping (TrustRegionMinimizer::Minimize's accept / reject / terminate logic), shared by k_decide
// (kernels_solve.h) and by the prologue of k_lin in the passes where the decision rides there (kernels_lin.h, MODE_DECIDE).
#pragma once
#include "dev_math.h"
#include "dev_types.h"

// candidate z's number: slot 0 is the regular field, slots 1.. the *E arrays (compare-and-select: a run-time index into
// these small arrays would put them, and the header copy around them, in scratch memory)
DEV double pick_cand(double first, const double (&e)[SPEC_EXTRA], int z) {
  double v = first;
#pragma unroll
  for (int j = 0; j < SPEC_EXTRA; j++) v = z == j + 1 ? e[j] : v;
  return v;
}
DEV double pick_slot(const double (&a)[1 + SPEC_EXTRA], int z) {
  double v = a[0];
#pragma unroll
  for (int j = 1; j < 1 + SPEC_EXTRA; j++) v = z == j ? a[j] : v;
  return v;
}

// the loop flags as a decision that has not reached the header yet leaves them (both are requested together)
DEV TRFlags tr_flags_decided(const Slot *S) {
  TRFlags f = tr_flags(&S->tr);
  const int pending = S->dec_pending;
  const int it = S->dec.iteration, cur = S->dec.cur, dl = S->dec.do_lin, ds = S->dec.do_schur, dn = S->dec.done, te = S->dec.termination, cf = S->dec.chol_fail;
  if (pending) f.iteration = it, f.cur = cur, f.do_lin = dl, f.do_schur = ds, f.done = dn, f.termination = te, f.chol_fail = cf;
  return f;
}

struct DecideSums {
  double cost[1 + SPEC_EXTRA], mlin[1 + SPEC_EXTRA], mquad[1 + SPEC_EXTRA], dn[1 + SPEC_EXTRA], xn[1 + SPEC_EXTRA];
};
DEV int decide_candidates(const TRHead &t) { return t.chol_fail ? 1 : (t.spec_n < 1 ? 1 : (t.spec_n > 1 + SPEC_EXTRA ? 1 + SPEC_EXTRA : t.spec_n)); }

// The 64 lanes of one wave: cost and model terms of every candidate slot from the per-block partials.  The partial sums
// of EVERY slot are requested at once, whatever K says (a slot that was not evaluated holds stale numbers, which are
// dropped): behind `z < K` each slot's loads would be a memory round trip of their own.
DEV void decide_sums(const Slot *S, const TRHead &t, int K, int sharded, int nLmBlocks, int lane, DecideSums &o) {
  const double *cp0 = S->cost_part, *cpE = S->cost_partE, *pc0 = S->pose_cost;
  double c[1 + SPEC_EXTRA], l[1 + SPEC_EXTRA], q[1 + SPEC_EXTRA], d[1 + SPEC_EXTRA], x[1 + SPEC_EXTRA];
#pragma unroll
  for (int z = 0; z < 1 + SPEC_EXTRA; z++) {
    const double *cp = z == 0 ? cp0 : cpE + (size_t)(z > 0 ? z - 1 : 0) * (SPEC_MAX_LM / 64) * LMS;
    const double *pcz = z == 0 ? pc0 : (const double *)S->pose_costE[z > 0 ? z - 1 : 0];
    const int nbz = z == 0 ? nLmBlocks : min(nLmBlocks, SPEC_MAX_LM / 64);
    c[z] = l[z] = q[z] = d[z] = x[z] = 0.0;
#pragma unroll 8
    for (int k = lane; k < nbz; k += 64) {  // (unrolled: the partials of eight blocks are requested together)
      const double *p = cp + (size_t)k * LMS;
      c[z] += p[0], l[z] += p[1], q[z] += p[2], d[z] += p[3], x[z] += p[4];
    }
    if (lane < 11) c[z] += pcz[lane];
  }
#pragma unroll
  for (int z = 0; z < 1 + SPEC_EXTRA; z++) {
    const bool use = !t.chol_fail && z < K && !sharded;
    o.cost[z] = use ? wave_sum(c[z]) : 0.0, o.mlin[z] = use ? wave_sum(l[z]) : 0.0, o.mquad[z] = use ? wave_sum(q[z]) : 0.0;
    o.dn[z] = use ? wave_sum(d[z]) : 0.0, o.xn[z] = use ? wave_sum(x[z]) : 0.0;
  }
  if (sharded && !t.chol_fail) {  // all-reduced by the caller after k_xpack 3 (one candidate)
    const double *sc = S->xch + XOFF_C;
    o.cost[0] = sc[XS_CCOST], o.mlin[0] = sc[XS_MLIN], o.mquad[0] = sc[XS_MQUAD], o.dn[0] = sc[XS_DN], o.xn[0] = sc[XS_XN];
  }
}

// ONE lane: one iteration per evaluated candidate.  The pass holds K of them (the steps for radius, radius / 2, radius / 4),
// and they are taken in that order exactly as Ceres would meet them — a rejected step halves the radius and the next
// candidate is the step for that radius; the walk stops at the first accepted, invalid or terminating one.  t is updated in
// place; the iteration summaries go to trace_dst (nullptr: the caller only wants the outcome).  Returns the accepted
// candidate slot (0 also when nothing was accepted).
DEV int decide_walk(TRHead &t, const DecideSums &sm, int K, int sharded, int max_iter, TRState *trace_dst) {
  int acc_z = 0;
  if (t.chol_fail && t.mu < 1.0) {
    // retry the Gauss-Newton solve with the larger mu (k_solve raised it); LINEAR_SOLVER_FAILURE once mu >= max_mu
    t.do_lin = sharded ? 1 : 0;
    t.do_schur = 1;
    t.chol_fail = 0;
    t.skip_step = 1;
    return 0;
  }
  const double function_tolerance = t.function_tolerance, gradient_tolerance = 1e-10, parameter_tolerance = 1e-8;  // (1e-6: k_setup)
  const double min_relative_decrease = 1e-3, min_trust_region_radius = 1e-32;
  (void)gradient_tolerance;
  // (the per-candidate numbers are picked by compare-and-select: a run-time index into these small arrays would put them —
  // and the whole header copy — in scratch memory, and every access of this single-lane chain would be a memory round trip)
  for (int z = 0; z < K; z++) {
    const double cgz = pick_cand(t.cg, t.cgE, z), cnz = pick_cand(t.cn, t.cnE, z), snz = pick_cand(t.dogleg_step_norm, t.snE, z);
    const double step_sq = pick_cand(t.step_sq_pose, t.step_sqE, z), xn2c = pick_cand(t.xn2_pose_cand, t.xn2E, z);
    const double cost_z = pick_slot(sm.cost, z), mlin_z = pick_slot(sm.mlin, z), mquad_z = pick_slot(sm.mquad, z), dn_z = pick_slot(sm.dn, z), xn_z = pick_slot(sm.xn, z);
    LfvioIterationSummary it;
    it.cost = t.x_cost, it.cost_change = 0, it.gradient_max_norm = 0, it.step_norm = 0, it.relative_decrease = 0;
    it.step_is_valid = 0, it.step_is_successful = 0;
    bool finished = false, go_on = false;
    bool step_valid = false;
    double model_cost_change = 0;
    if (!t.chol_fail) {
      // model_cost_change = -(J step)^T (r + J step / 2) = -delta.g - 1/2 delta^T H delta
      // unscaled pose direction delta_p = cg' G + cn' N where gradient_/diagonal_*scale = G, gn/diag*scale = N
      const double lin = cgz * t.q[Q_gG] + cnz * t.q[Q_gN] + mlin_z;
      const double quad = cgz * cgz * t.q[Q_GG] + 2.0 * cgz * cnz * t.q[Q_GN] + cnz * cnz * t.q[Q_NN] + mquad_z;
      model_cost_change = -lin - 0.5 * quad;
      step_valid = model_cost_change > 0.0;
    }
    t.model_cost_change = model_cost_change;
    it.step_is_valid = step_valid ? 1 : 0;
    if (!step_valid) {
      // HandleInvalidStep
      if (++t.consec_invalid >= 5) {
        t.termination = LFVIO_FAILURE;
        t.done = 1;
        finished = true;
      } else {
        t.mu *= 10.0;  // StepIsInvalid
        t.chol_fail = 0;
        t.do_lin = sharded ? 1 : 0;  // sharded: the exchange buffers were reduced in place, rebuild them
        t.do_schur = 1;
      }
    } else {
      t.consec_invalid = 0;
      const double candidate_cost = isfinite(cost_z) ? cost_z : 1.79769313486231570815e+308;
      t.cand_cost = candidate_cost;
      it.step_norm = sqrt(step_sq + dn_z);
      if (it.step_norm <= parameter_tolerance * (t.x_norm + parameter_tolerance)) {
        t.termination = LFVIO_CONVERGENCE;
        t.done = 1;
        finished = true;
      } else {
        it.cost_change = t.x_cost - candidate_cost;
        if (fabs(it.cost_change) <= function_tolerance * t.x_cost) {
          t.termination = LFVIO_CONVERGENCE;
          t.done = 1;
          finished = true;
        } else {
          it.relative_decrease = it.cost_change / model_cost_change;
          if (it.relative_decrease > min_relative_decrease) {
            // HandleSuccessfulStep: x <- candidate; the next k_lin re-evaluates cost/gradient there
            t.cur ^= 1;
            acc_z = z;
            t.x_norm = sqrt(xn2c + xn_z);
            it.step_is_successful = 1;
            it.cost = candidate_cost;  // replaced by the re-evaluated x_cost when the trace is read
            // Ceres evaluates the gradient at the accepted point before it looks at the iteration cap (HandleSuccessfulStep ->
            // EvaluateGradientAndJacobian); here that is the next pass's linearization, and k_dogleg writes its max-norm into
            // this entry.  A loop that ends with this step never linearizes there: the entry keeps NaN = not evaluated
            // (include/lfvio.h): a consumer that takes a minimum or a maximum over the trace cannot mistake it for a norm.
            it.gradient_max_norm = __builtin_nan("");
            if (it.relative_decrease < 0.25) t.radius *= 0.5;
            if (it.relative_decrease > 0.75) t.radius = fmax(t.radius, 3.0 * snz);
            t.mu = fmax(1e-8, 2.0 * t.mu / 10.0);
            t.do_lin = 1;
            t.do_schur = 1;
            t.x_cost = candidate_cost;
          } else {
            // HandleUnsuccessfulStep / StepRejected: the next candidate is the step for the halved radius
            t.radius *= 0.5;
            t.do_lin = 0;
            t.do_schur = 0;
            it.cost = candidate_cost;
            go_on = true;
          }
        }
      }
    }
    if (finished) break;  // the converged iteration is not pushed (Minimize() returns before Finalize)
    // FinalizeIterationAndCheckIfMinimizerCanContinue
    if (it.step_is_successful)
      t.num_succ++;
    else
      t.num_unsucc++;
    it.trust_region_radius = t.radius;
    if (t.trace_len < LFVIO_MAX_TRACE) {
      if (trace_dst) {  // field by field: a struct copy would stage `it` in scratch memory
        LfvioIterationSummary *dst = &trace_dst->trace[t.trace_len];
        dst->cost = it.cost, dst->cost_change = it.cost_change, dst->gradient_max_norm = it.gradient_max_norm;
        dst->step_norm = it.step_norm, dst->relative_decrease = it.relative_decrease, dst->trust_region_radius = it.trust_region_radius;
        dst->step_is_valid = it.step_is_valid, dst->step_is_successful = it.step_is_successful;
      }
      t.trace_len++;
    }
    if (t.iteration >= max_iter) {
      t.termination = LFVIO_NO_CONVERGENCE;
      t.done = 1;
    } else if (t.radius <= min_trust_region_radius) {
      t.termination = LFVIO_CONVERGENCE;
      t.done = 1;
    }
    t.iteration++;
    if (!go_on || t.done) break;
  }
  return acc_z;
}

DEV void decision_from(TRDecision &d, const TRHead &t, int acc_z) {
  d.radius = t.radius, d.mu = t.mu, d.x_cost = t.x_cost, d.x_norm = t.x_norm, d.cand_cost = t.cand_cost, d.model_cost_change = t.model_cost_change;
  d.iteration = t.iteration, d.cur = t.cur, d.do_lin = t.do_lin, d.do_schur = t.do_schur, d.done = t.done, d.termination = t.termination;
  d.chol_fail = t.chol_fail, d.num_succ = t.num_succ, d.num_unsucc = t.num_unsucc, d.consec_invalid = t.consec_invalid;
  d.trace_len = t.trace_len, d.skip_step = t.skip_step, d.acc_z = acc_z, d.pad_ = 0;
}
// only what the bookkeeping changes goes back into the header
DEV void decision_to_header(TRState *tr, const TRDecision &d) {
  tr->radius = d.radius, tr->mu = d.mu, tr->x_cost = d.x_cost, tr->x_norm = d.x_norm, tr->cand_cost = d.cand_cost;
  tr->model_cost_change = d.model_cost_change;
  tr->iteration = d.iteration, tr->cur = d.cur, tr->do_lin = d.do_lin, tr->do_schur = d.do_schur, tr->done = d.done;
  tr->termination = d.termination, tr->chol_fail = d.chol_fail, tr->skip_step = d.skip_step;
  tr->num_succ = d.num_succ, tr->num_unsucc = d.num_unsucc, tr->consec_invalid = d.consec_invalid, tr->trace_len = d.trace_len;
}
// the accepted step is a speculative candidate: bring it into the slots x / tab / lam [cur] stand for (cur: after the flip)
DEV void copy_accepted(Slot *S, int az, int cur, int nlm, int tid, int nthr) {
  const double *xs = (const double *)&S->xE[az - 1], *ts = (const double *)&S->tabE[az - 1], *ls = S->lamE[az - 1];
  double *xd = (double *)&S->x[cur], *td = (double *)&S->tab[cur], *ld = S->lam[cur];
  // rounds of eight loads per thread in front of their stores (a copy loop is a memory round trip per trip: fourteen of them for the
  // 22 KB table, the state and 300 inverse depths on 256 threads — and the workers of kernels_spec.h wait for this copy)
  auto copy = [&](double *d, const double *s, int n) {
    for (int k0 = tid; k0 < n; k0 += 8 * nthr) {
      double v[8];
#pragma unroll
      for (int q = 0; q < 8; q++) v[q] = s[k0 + q * nthr < n ? k0 + q * nthr : 0];
#pragma unroll
      for (int q = 0; q < 8; q++)
        if (k0 + q * nthr < n) d[k0 + q * nthr] = v[q];
    }
  };
  copy(xd, xs, (int)(sizeof(FrameState) / 8));
  copy(td, ts, (int)(sizeof(Tab) / 8));
  copy(ld, ls, nlm);
}
