// kernels_marg.h — gauge fix of double2vector() and MarginalizationInfo::marginalize() on
// the device.
//
//   k_gauge      : double2vector() + vector2double() (estimator.cpp:532-600, 488-530)
//   k_marg_solve : assemble A, b over the pose-side blocks with the frame-0 landmarks already
//                  eliminated (they are a diagonal block of A_mm: each landmark couples only to
//                  pose-side columns), eigen-decompose the remaining dropped block (pose 0 +
//                  speed/bias 0, or pose 9 for MARGIN_SECOND_NEW), Schur, eigen-decompose the
//                  kept system and factor it into linearized_jacobians / linearized_residuals
//                  (marginalization_factor.cpp:264-291).
// The reference eigen-decomposes the whole m x m block (m = 15 + #landmarks) and thresholds its
// eigenvalues at eps = 1e-8; here the landmark part is inverted entry-wise with the same threshold
// and the dense 15 x 15 remainder by eigen-decomposition — identical when A_mm is positive definite
// (block elimination is exact), and the only formulation that scales past ~10^3 landmarks.
#pragma once
#include "kernels_solve.h"

constexpr int MARG_THREADS = 768;   // 12 waves: the gather, the Schur step and the back-transformation (eight lanes per eigenvector) use them all
constexpr int MARG_MAXD = 96;  // 15 dropped + 76 kept, padded
constexpr int LDN = 80;              // LDS row stride of the n x n (n <= 76) eigen-problem
constexpr double JACOBI_TOL = 1e-19;  // off-diagonal mass / diagonal mass at which the Jacobi of the dropped block stops (jacobi_small)
constexpr size_t MARG_LDS = (size_t)19072 * sizeof(double);  // >= LPACK + KP + 64 and >= the eigen-phase carve below
constexpr int BT_GROUP = 4;          // reflectors per step of the back-transformation (bt_apply)

// setDepth / getDepthVector round trip (feature_manager.cpp:148,191), landmarks [l0, l0 + 128)
DEV void gauge_landmarks(Slot *S, int l0) {
  const int l = l0 + (int)threadIdx.x;
  double *lam = S->lam[S->tr.cur];
  if (l < S->N) lam[l] = 1.0 / (1.0 / lam[l]);
}
// dst[0, n) = src[0, n) by the calling workgroup, eight loads of a thread in front of their stores (a copy loop with a run-time trip count
// is a memory round trip per trip: the stores may alias the next load as far as the compiler knows)
template <class T>
DEV void copy_rounds(T *dst, const T *src, int n, int tid, int nthr) {
  for (int k0 = tid; k0 < n; k0 += 8 * nthr) {
    T v[8];
#pragma unroll
    for (int q = 0; q < 8; q++) v[q] = src[k0 + q * nthr < n ? k0 + q * nthr : 0];
#pragma unroll
    for (int q = 0; q < 8; q++)
      if (k0 + q * nthr < n) dst[k0 + q * nthr] = v[q];
  }
}
DEV void gauge_poses(Slot *S, int gated, bool publish);
// The finished state into the caller's mailbox (Slot::mail, dev_types.h): x[cur], the trust-region header with as much of
// the trace as there is, lam[cur] — plain stores to host memory, a system-scope fence, then the flag.  Called by every
// thread of ONE workgroup after the stores of the gauge fix are visible to it.
DEV void publish_solution(Slot *S) {
  char *m = (char *)S->mail;
  if (!m) return;
  const int tid = threadIdx.x, nthr = blockDim.x;
  const TRState *ts = &S->tr;
  const int cur = ts->cur, N = S->N;
  // (every load of the three copies in front of the first store to the host: a copy loop is a device-memory round trip per trip)
  {
    constexpr int XW = (int)(sizeof(FrameState) / 8), XT = (XW + 63) / 64, LT = (SPEC_MAX_LM + 63) / 64, TT = 8;
    const double *xs = (const double *)&S->x[cur];
    double *xd = (double *)(m + MAIL_X + (size_t)cur * sizeof(FrameState));
    const long long *tsrc = (const long long *)ts;
    long long *tdst = (long long *)(m + MAIL_TR);
    const int tl = ts->trace_len < LFVIO_MAX_TRACE ? ts->trace_len : LFVIO_MAX_TRACE;
    const int words = (int)((offsetof(TRState, trace) + (size_t)(tl > 0 ? tl : 0) * sizeof(LfvioIterationSummary)) / 8);
    const double *ls = S->lam[cur];
    double *ld = (double *)(m + MAIL_LAM + (size_t)cur * MAIL_LAM_STRIDE);
    if (nthr >= 64 && XW <= XT * nthr && N <= LT * nthr && words <= TT * nthr) {
      double xv[XT], lv[LT];
      long long tv[TT];
#pragma unroll
      for (int k = 0; k < XT; k++) xv[k] = xs[tid + nthr * k < XW ? tid + nthr * k : 0];
#pragma unroll
      for (int k = 0; k < LT; k++) lv[k] = ls[tid + nthr * k < N ? tid + nthr * k : 0];
#pragma unroll
      for (int k = 0; k < TT; k++) tv[k] = tsrc[tid + nthr * k < words ? tid + nthr * k : 0];
#pragma unroll
      for (int k = 0; k < XT; k++)
        if (tid + nthr * k < XW) xd[tid + nthr * k] = xv[k];
#pragma unroll
      for (int k = 0; k < TT; k++)
        if (tid + nthr * k < words) tdst[tid + nthr * k] = tv[k];
#pragma unroll
      for (int k = 0; k < LT; k++)
        if (tid + nthr * k < N) ld[tid + nthr * k] = lv[k];
    } else {
      for (int k = tid; k < XW; k += nthr) xd[k] = xs[k];
      for (int k = tid; k < words; k += nthr) tdst[k] = tsrc[k];
      for (int k = tid; k < N; k += nthr) ld[k] = ls[k];
    }
  }
  if (tid == 0) ((int *)m)[4] = S->passes_used, ((int *)m)[5] = S->chain_err;  // (final: the loop is closed; k_prior_chain's verdict on the prior this window ran with)
  __threadfence_system();
  __syncthreads();
  // (the flags carry the upload's sequence number, not 1: the marginalization of the window BEFORE may still be publishing its
  // prior when the host is already waiting for this window's — lfvio_batch_upload_chained_device)
  if (tid == 0) __hip_atomic_store((int *)m, S->mail_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
// The prior the gated marginalization has just written (Slot::prior_out) into the mailbox, then the second flag.  Called by
// every thread of k_marg_solve's workgroup at its end, behind a barrier that follows the last store to prior_out.
DEV void publish_prior(Slot *S) {
  char *m = (char *)S->mail;
  if (!m) return;
  const int tid = threadIdx.x, nthr = blockDim.x;
  const LfvioPrior *src = &S->prior_out;
  LfvioPrior *dst = (LfvioPrior *)(m + MAIL_PRIOR);
  const int n = src->valid == 1 ? src->n : 0;
  copy_rounds((long long *)dst, (const long long *)src, (int)(offsetof(LfvioPrior, linearized_jacobians) / 8), tid, nthr);
  copy_rounds(dst->linearized_jacobians, src->linearized_jacobians, n * n, tid, nthr);
  copy_rounds(dst->linearized_residuals, src->linearized_residuals, n, tid, nthr);
  if (tid == 0) ((int *)m)[2] = S->passes_used, ((int *)m)[3] = S->tr.iteration;  // (two words: max_num_iterations is the caller's, either count may pass 255)
  __threadfence_system();
  __syncthreads();
  if (tid == 0) __hip_atomic_store((int *)m + 1, S->mail_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
// behind a k_gauge of several workgroups (windows too large for k_decide_gauge): the same gate, then the mailbox
__global__ __launch_bounds__(256) void k_publish(char *base, size_t stride) {
  Slot *S = SLOT(base, stride);
  if (!(S->tr.done && (S->tail_state == 0 || S->tail_state == 3))) return;  // (3: the prior of this state is a worker's — kernels_spec.h; the state is the loop's)
  publish_solution(S);
}
// k_prior_chain: grid 1 x 256, behind the upload of the NEXT window of the same estimator into a slot whose marginalization has
// just run (or is the kernel in front of this one on the stream): the prior that marginalization left in Slot::prior_out becomes
// the window's input prior where it lies — values only (J0, r0, the blocks' linearization points); the block structure is the
// host's, which planned that marginalization itself (lfvio_batch_upload_chained_device).  46 KB that neither go down nor come up
// again, and no host in between: the upload is enqueued while the marginalization is still running.
__global__ __launch_bounds__(256) void k_prior_chain(char *base, size_t stride) {
  Slot *S = SLOT(base, stride);
  const LfvioPrior *src = &S->prior_out;
  const int tid = threadIdx.x, n = S->prior_n, nb = S->prior_nb;
  // the marginalization in front of this window was a worker's on the second stream (kernels_spec.h): it ends by moving the prior
  // here and setting tail_state to 2 — bounded wait (a worker the loop has committed holds a finished prior)
  __shared__ int late;
  if (tid == 0) {
    late = 0;
    const long long t0 = wall_clock64();
    while (spec_peek(&S->tail_state) == 3) {
      if (wall_clock64() - t0 > 10 * SPEC_WAIT_TICKS) {
        late = 1;
        break;
      }
      __builtin_amdgcn_s_sleep(8);
    }
    spec_acquire();
  }
  __syncthreads();
  // the block structure the host promised (its own plan of that marginalization) against what the marginalization wrote: same blocks in
  // the same order at the same columns, not just as many of them
  // (a thread per block: one after the other behind `same &&` every block's six loads were a memory round trip of their own)
  __shared__ int differs;
  if (tid == 0) differs = 0;
  __syncthreads();
  const bool head = !late && src->valid == 1 && src->n == n && src->num_blocks == nb;
  if (head && tid < nb && !(src->blocks[tid].kind == S->prior_kind[tid] && src->blocks[tid].frame == S->prior_frame[tid] && src->block_idx[tid] == S->prior_idx[tid]))
    differs = 1;
  for (int i = 256 + tid; head && i < nb; i += 256)  // (more blocks than threads: not with LFVIO_MAX_PRIOR_BLOCKS, kept for the bound's sake)
    if (!(src->blocks[i].kind == S->prior_kind[i] && src->blocks[i].frame == S->prior_frame[i] && src->block_idx[i] == S->prior_idx[i])) differs = 1;
  __syncthreads();
  const bool same = head && !differs;
  if (!same) {
    // no prior where one was promised (the marginalization failed or produced another structure): the window runs without one and says so
    if (tid == 0) S->prior_valid = 0, S->chain_err = 1;
    return;
  }
  double *J = S->prior_J, *r = S->prior_r;
  copy_rounds(J, src->linearized_jacobians, n * n, tid, 256);
  copy_rounds(r, src->linearized_residuals, n, tid, 256);
  copy_rounds(&S->prior_x0[0][0], &src->block_x0[0][0], nb * 9, tid, 256);  // (both [blocks][9])
}
__global__ __launch_bounds__(128) void k_gauge(char *base, size_t stride, int gated) {
  Slot *S = SLOT(base, stride);
  if (gated && !tail_gate(S, S->tr.done)) return;
  if (blockIdx.x > 0) {
    gauge_landmarks(S, (blockIdx.x - 1) * 128);
    return;
  }
  gauge_poses(S, gated, false);
  // (the kernel in front of this one — k_decide, k_force_done — has raised FIN_CLOSING: several workgroups rewrite the state here)
  if (gated && S->spec_on && threadIdx.x == 0) spec_settle(S);
}
// k_decide_gauge: grid (1, batch) x 128 — behind the last pass of a graph of few small windows: k_decide and, for the slots
// that are done then, the gated k_gauge in ONE launch (one workgroup per slot: nobody else reads the header it rewrites).
__global__ __launch_bounds__(128) void k_decide_gauge(char *base, size_t stride, int publish) {
  Slot *S = SLOT(base, stride);
  decide_body(S);
  __syncthreads();  // (the header and the accepted candidate, written by wave 0, are read by all from here on)
  if (!tail_gate(S, S->tr.done)) return;
  const int spec = S->spec_on;
  if (spec) {  // a worker that has not claimed this state by now never will (kernels_spec.h): x[cur] is rewritten in place below
    if (threadIdx.x == 0) spec_closing(S);
    __syncthreads();
  }
  {  // gauge_landmarks for every landmark of the window (at most MAIL_MAX_LM with this kernel), the loads of all trips in one round
    constexpr int LT = (SPEC_MAX_LM + 127) / 128;
    double *lam = S->lam[S->tr.cur];
    const int N = S->N, tid = threadIdx.x;
    double lv[LT];
#pragma unroll
    for (int k = 0; k < LT; k++) lv[k] = lam[tid + 128 * k < N ? tid + 128 * k : 0];
#pragma unroll
    for (int k = 0; k < LT; k++)
      if (tid + 128 * k < N) lam[tid + 128 * k] = 1.0 / (1.0 / lv[k]);
    for (int l0 = 128 * LT; l0 < N; l0 += 128) gauge_landmarks(S, l0);
  }
  gauge_poses(S, 1, publish != 0);  // publish: this call hands its state over early (lfvio_batch_optimize_begin)
  if (spec && threadIdx.x == 0) spec_settle(S);  // (behind the state's way out: the prior's owner is the marginalization's business)
}
DEV void gauge_poses(Slot *S, int gated, bool publish) {
  TRState *ts = &S->tr;
  const int tid = threadIdx.x;
  __shared__ double rot[9], P0[3], oP0[3];
  FrameState *x = &S->x[ts->cur];
  if (tid == 0) {
    // rot_diff from the yaw difference of frame 0 before / after the solve (estimator.cpp:534-560).  The reference goes
    // through Utility::R2ypr (three atan2 per matrix, degrees) and ypr2R; only the yaw difference and the test "pitch
    // within one degree of +-90" are used, and both follow from the first column n of the rotation without any
    // trigonometric function: (cos y, sin y) = (n0, n1) / hypot(n0, n1), cos(pitch) = hypot(n0, n1).
    const m33 Rs0 = q2R(q_from_pose(S->x0.pose[0]));
    const m33 R00 = q2R(q_from_pose(x->pose[0]));
    const double h0 = sqrt(Rs0.a[0] * Rs0.a[0] + Rs0.a[3] * Rs0.a[3]), h00 = sqrt(R00.a[0] * R00.a[0] + R00.a[3] * R00.a[3]);
    const double c0 = Rs0.a[0] / h0, s0 = Rs0.a[3] / h0, c00 = R00.a[0] / h00, s00 = R00.a[3] / h00;
    const double cyd = c0 * c00 + s0 * s00, syd = s0 * c00 - c0 * s00;  // y_diff = yaw(Rs0) - yaw(R00)
    m33 rd;  // ypr2R(y_diff, 0, 0) = Rz(y) Ry(0) Rx(0)
    rd.a[0] = cyd, rd.a[1] = -syd, rd.a[2] = 0;
    rd.a[3] = syd, rd.a[4] = cyd, rd.a[5] = 0;
    rd.a[6] = 0, rd.a[7] = 0, rd.a[8] = 1;
    const double sin1deg = 0.017452406437283512;  // |pitch| > 89 degrees  <=>  cos(pitch) < sin(1 degree)
    if (h0 < sin1deg || h00 < sin1deg) rd = mm(Rs0, tr(R00));
    for (int k = 0; k < 9; k++) rot[k] = rd.a[k];
    for (int k = 0; k < 3; k++) P0[k] = x->pose[0][k], oP0[k] = S->x0.pose[0][k];
  }
  __syncthreads();
  const m33 rd = ldm(rot);
  d3 Psi = mk3(0, 0, 0), Vsi = Psi;
  q4 qn = q4{1, 0, 0, 0};
  if (tid < 11) {
    const m33 Rsi = mm(rd, q2R(qnormalized(q_from_pose(x->pose[tid]))));  // :565
    Psi = mul(rd, ld3(x->pose[tid]) - ld3(P0)) + ld3(oP0);               // :567-570
    Vsi = mul(rd, ld3(x->sb[tid]));                                        // :572-574
    qn = R2q(Rsi);                                                         // vector2double :494
  } else if (tid == 11) {
    qn = R2q(q2R(q_from_pose(x->ex)));  // ric = Quaterniond(para_Ex_Pose).toRotationMatrix(); Quaterniond{ric}
  }
  __syncthreads();
  __shared__ double bt[84 + TAB_SCRATCH];  // the re-anchored poses for build_tab, and its scratch
  if (tid < 11) {
    const double pn[7] = {Psi.x, Psi.y, Psi.z, qn.x, qn.y, qn.z, qn.w};
#pragma unroll
    for (int k = 0; k < 7; k++) x->pose[tid][k] = pn[k], bt[7 * tid + k] = pn[k];
    x->sb[tid][0] = Vsi.x, x->sb[tid][1] = Vsi.y, x->sb[tid][2] = Vsi.z;
  } else if (tid == 11) {
    x->ex[3] = qn.x, x->ex[4] = qn.y, x->ex[5] = qn.z, x->ex[6] = qn.w;
#pragma unroll
    for (int k = 0; k < 3; k++) bt[77 + k] = x->ex[k];
    bt[80] = qn.x, bt[81] = qn.y, bt[82] = qn.z, bt[83] = qn.w;
  }
  __syncthreads();
  if (publish) publish_solution(S);  // (ends with a barrier: the header is read before thread 0 changes its flags below)
  if (tid == 0) {
    if (!gated) ts->done = 0;  // gated: `done` stays, the gated sweep that follows tests it like this kernel did
    ts->do_lin = 1;
    ts->do_schur = 1;
    ts->chol_fail = 0;
  }
  build_tab(bt, &S->tab[ts->cur], tid, bt + 84);
}

// k_spec_wait: grid 1 x 64 on stream 0, in front of an upload that goes out behind a call still in flight
// (lfvio_batch_upload_chained_device): the worker that owns that call's prior reads slot 0's input arrays until it has delivered.
__global__ __launch_bounds__(64) void k_spec_wait(char *base) {
  Slot *S = (Slot *)base;
  if (threadIdx.x != 0) return;
  const long long t0 = wall_clock64();
  while (spec_peek(&S->tail_state) == 3 && wall_clock64() - t0 < 10 * SPEC_WAIT_TICKS) __builtin_amdgcn_s_sleep(8);
}
// k_spec_begin: grid 1 x 128 on a worker's stream — first launch of a round (kernels_spec.h).  base: the worker's SHADOW slot; back: its
// distance from the slot being solved.  Waits for an accepted state newer than the last one it looked at, copies it, claims
// its prior (the other worker may be faster: then it goes back to waiting) and re-anchors the copy like the gated gauge fix
// re-anchors the original (the same gauge_poses on the same numbers).
// Leaves tr.done = 0 in the shadow when there is nothing to do: the round's three gated launches return.
__global__ __launch_bounds__(128) void k_spec_begin(char *base, size_t back) {
  Slot *S = (Slot *)base;
  Slot *S0 = (Slot *)(base - back);
  const int tid = threadIdx.x;
  __shared__ int sh_word, sh_go;
  const long long t0 = wall_clock64();
  for (;;) {
    if (tid == 0) {
      int go = 0, w = 0;
      const int last = S->shadow.last_word;
      for (;;) {
        const int f = spec_peek(&S0->spec.fin);  // (fin first, then word: k_setup withdraws the word before it re-opens fin)
        w = spec_peek(&S0->spec.word);
        if (w != 0) {
          const int st = spec_ep(f) == spec_ep(w) ? spec_fin_state(f) : FIN_OPEN;
          if (st != FIN_OPEN) break;  // the loop is closing or closed: whatever was not claimed is its own
          if (w != last) {
            go = 1;
            break;
          }
        }
        if (wall_clock64() - t0 > SPEC_WAIT_TICKS) break;
        __builtin_amdgcn_s_sleep(8);
      }
      spec_acquire();  // (what the word announces — x[cur], lam[cur] — is read behind this)
      sh_word = w, sh_go = go;
    }
    __syncthreads();
    // (onto the scalar side explicitly: with the LDS word in a vector register the compiler of ROCm 7.2 selected `go ? w : 0`
    // by s_cselect on a condition code no scalar compare had set)
    int go = __builtin_amdgcn_readfirstlane(sh_go);
    const int w = __builtin_amdgcn_readfirstlane(sh_word);
    if (!go) {
      if (tid == 0) S->shadow.word = 0, S->tr.done = 0;
      return;
    }
    const int ep = spec_ep(w), a = spec_acc(w), cur = w & 1;
    if (S->shadow.hdr_ep != ep) {
      // first round of this call: the window's header (sizes, plans, x0, IMU factors, prior structure) and the IMU information
      // roots; the input pointers of the copy are moved back by the distance so that they lead into slot 0's arrays
      const long long *src = (const long long *)S0;
      long long *dst = (long long *)S;
      constexpr int W0 = (int)(offsetof(Slot, lm_start) / 8), W1 = (int)(offsetof(Slot, x) / 8);
      static_assert(offsetof(Slot, lm_start) % 8 == 0 && offsetof(Slot, x) % 8 == 0, "header words");
      for (int k = tid; k < W1; k += 128) dst[k] = src[k] - (k >= W0 ? (long long)back : 0ll);
      const double *is = &S0->imu_sqrt[0][0];
      double *id = &S->imu_sqrt[0][0];
      for (int k = tid; k < LFVIO_WINDOW_SIZE * 225; k += 128) id[k] = is[k];
    }
    __syncthreads();
    {  // the accepted state in ONE round of loads (a copy loop is a memory round trip per trip, and this kernel is on the call's critical
      // path); the inverse depths take the setDepth / getDepthVector round trip of the gauge fix (gauge_landmarks) on the way
      constexpr int XW = (int)(sizeof(FrameState) / 8), TW = (int)(sizeof(TRHead) / 8), XT = (XW + 127) / 128, LT = (SPEC_MAX_LM + 127) / 128;
      static_assert(TW <= 128, "the trust-region header in one trip");
      const double *xs = (const double *)&S0->x[cur];
      double *xd = (double *)&S->x[0];
      const double *ls = S0->lam[cur];
      double *ld = S->lam[0];
      const int N = S0->N;
      const long long *ts = (const long long *)&S0->tr;
      long long *td = (long long *)&S->tr;
      double xv[XT], lv[LT];
#pragma unroll
      for (int k = 0; k < XT; k++) xv[k] = xs[tid + 128 * k < XW ? tid + 128 * k : 0];
#pragma unroll
      for (int k = 0; k < LT; k++) lv[k] = ls[tid + 128 * k < N ? tid + 128 * k : 0];
      const long long tv = ts[tid < TW ? tid : 0];
#pragma unroll
      for (int k = 0; k < XT; k++)
        if (tid + 128 * k < XW) xd[tid + 128 * k] = xv[k];
#pragma unroll
      for (int k = 0; k < LT; k++)
        if (tid + 128 * k < N) ld[tid + 128 * k] = 1.0 / (1.0 / lv[k]);
      if (tid < TW) td[tid] = tv;
      for (int k = tid + 128 * LT; k < N; k += 128) ld[k] = 1.0 / (1.0 / ls[k]);  // (no window with a shadow slot is that large)
    }
    __syncthreads();  // (every load of the copy has returned)
    if (tid == 0) {
      // still the newest state, and the loop has not begun to close (its gauge fix rewrites x[cur] in place behind FIN_CLOSING)
      const int f = spec_ld(&S0->spec.fin), w2 = spec_ld(&S0->spec.word);
      const bool open = spec_ep(f) != ep || spec_fin_state(f) == FIN_OPEN;
      sh_go = (w2 == w && open && spec_cas(&S0->spec.own[a], SPEC_FREE, SPEC_SIDE) == SPEC_FREE) ? 1 : 0;
      S->shadow.hdr_ep = ep;
      S->shadow.last_word = w;  // (claimed, or somebody else's: not to be looked at again)
    }
    __syncthreads();
    go = __builtin_amdgcn_readfirstlane(sh_go);
    if (!go) {
      __syncthreads();  // (sh_go is rewritten at the top)
      continue;         // the other worker has it, or it is no longer the newest: wait for the next one
    }
    if (tid == 0) {
      S->shadow.word = w, S->shadow.ticket = S0->spec.ticket;
      S->spec_on = 0, S->dec_pending = 0, S->tail_state = 0, S->chain_err = 0;
      S->tr.cur = 0, S->tr.done = 1;
    }
    __syncthreads();
    break;
  }
  gauge_poses(S, 1, false);  // (the landmarks' part of the gauge fix went with the copy)
}

// Inverse of a symmetric positive-definite m x m (m <= 15) matrix by ONE wave, everything in registers: lane i holds
// row i (identity beyond m), Cholesky column by column (pivot and column entries broadcast with v_readlane), then
// L^-1 one column per lane by forward substitution and X = L^-T L^-1.  Returns true when the factorization went through
// AND every eigenvalue of the matrix is provably above `floor` (lambda_min = 1 / lambda_max(X) >= 1 / ||X||_F): only then
// is X the pseudo-inverse the reference forms from the eigen-decomposition (marginalization_factor.cpp:267-272), and the
// caller may skip that decomposition.  Called by the 64 lanes of one wave; Xout (m x m) is written either way.
DEV bool spd_inverse_wave(const double *Am, double *Xout, int m, int lane, double floor) {
  double a[15];
#pragma unroll
  for (int j = 0; j < 15; j++) a[j] = (lane < m && j < m) ? Am[lane * m + j] : (lane == j ? 1.0 : 0.0);
  bool ok = true;
  double dinv = 1.0;
#pragma unroll
  for (int j = 0; j < 15; j++) {
    const double d = readlane_f64(a[j], j);
    ok = ok && (d > 0.0);
    const double rs = fast_rsqrt(d);  // (one reciprocal square root per column serves the diagonal, the column and 1 / L_jj)
    const double l = lane == j ? d * rs : a[j] * rs;
    a[j] = l;
    dinv = lane == j ? rs : dinv;
#pragma unroll
    for (int c = j + 1; c < 15; c++) a[c] = fma(-l, readlane_f64(l, c), a[c]);
  }
  double y[15];  // y[i] = (L^-1)[i][lane]
#pragma unroll
  for (int i = 0; i < 15; i++) {
    double t = lane == i ? 1.0 : 0.0;
#pragma unroll
    for (int k = 0; k < i; k++) t = fma(-readlane_f64(a[k], i), y[k], t);
    y[i] = t * readlane_f64(dinv, i);
  }
  double fro = 0;
#pragma unroll
  for (int r = 0; r < 15; r++) {
    double x = 0;
#pragma unroll
    for (int i = r; i < 15; i++) x = fma(readlane_f64(y[i], r), y[i], x);
    if (lane < m && r < m) Xout[r * m + lane] = x, fro += x * x;
  }
  fro = wave_sum(fro);
  return ok && fro * floor * floor < 1.0;  // NaN compares false
}

// Parallel cyclic Jacobi eigen-decomposition of the symmetric n x n matrix A (LDS, ld = n).
// On exit diag(A) holds the eigenvalues and the columns of V (LDS, ld = n) the eigenvectors.
// Round-robin ordering: np/2 disjoint rotations per step, np-1 steps per sweep; every 2x2 block
// A[{p_a,q_a}][{p_b,q_b}] is touched by exactly one thread (J_a^T B J_b), so a step needs two
// barriers and no atomics.  Threads are mapped 16 x 16 over (pair a, pair b): no integer division.
// Stops when the off-diagonal mass is below JACOBI_TOL = 1e-19 of the diagonal mass (off/||A|| < 3e-10: the conditioning
// error of A' itself is ~1e-10, A_mm carries IMU information ~1e10) or stagnates.  What the sweeps beyond that point do
// is diagonalize the block of noise eigenvalues (|lambda| ~ 1e-6 against ||A|| ~ 2e6: 1e-23 of the mass) — three more
// sweeps at linear, not quadratic, rate; per-call parity (tests/tools/fuzz_parity.py) and the chained-window deviation
// (tests/test_flow.py) are the same with 1e-24 and with 1e-19, with 1e-17 the chains start to differ.
// round-robin pairing of step `step`: slot k rotates (step+k, step-k) mod (np-1), slot 0 pairs the fixed index np-1.
// q = -1 marks the dummy index of an odd n.  Consecutive slots give consecutive rows / columns (no p<q reordering),
// which keeps a half-wave's LDS accesses in distinct banks for the row stride LDN.
DEV void rr_pair(int k, int step, int np, int n, int &p, int &q) {
  if (k == 0) {
    p = step, q = np - 1;  // for an odd n the dummy index np-1 = n never leaves this place
  } else {
    p = step + k;
    if (p >= np - 1) p -= np - 1;
    q = step - k;
    if (q < 0) q += np - 1;
  }
  if (q >= n) q = -1;
}

// Jacobi rotation that annihilates a_pq: t = tan(phi) is the smaller root of t^2 + 2 theta t - 1 = 0,
// theta = (a_qq - a_pp) / (2 a_pq), written without the division by a_pq:  t = sgn(d) e / (|d| + sqrt(d^2 + e^2)),
// d = a_qq - a_pp, e = 2 a_pq.  (cos, sin) = (1, t) / sqrt(1 + t^2); cos^2 + sin^2 = 1 to rounding whatever the error of t.
DEV void jacobi_angle(double app, double aqq, double apq, double &c, double &s) {
  c = 1.0, s = 0.0;
  // rotations that cannot change either diagonal entry in FP64 are skipped
  if (apq != 0.0 && fabs(apq) > 1e-18 * (fabs(app) + fabs(aqq))) {
#if defined(JAC_TRIVIAL)
    c = 0.8, s = 0.6;
#elif defined(JAC_IEEE)
    const double theta = (aqq - app) / (2.0 * apq);
    const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
    c = rsqrt(t * t + 1.0);
    s = t * c;
#else
    const double d = aqq - app, e = 2.0 * apq;
    const double h2 = fma(d, d, e * e);
    const double h = h2 * fast_rsqrt(h2);
    const double t = (d >= 0 ? e : -e) * fast_rcp(fabs(d) + h);
    c = fast_rsqrt(fma(t, t, 1.0));
    s = t * c;
#endif
  }
}

// Eigen-decomposition of a small symmetric matrix (n <= 16: the dropped pose-side block of the marginalization).
// Round-robin tournament (rr_pair), one thread per matrix ELEMENT: thread (u, v) of the first np^2 threads forms
// element (u, v) of J^T A J and of V J from the 2x2 blocks it sits in and stores it at its place after the ring move
// (double-buffered position space: one LDS round trip and two barriers per step, a few dozen instructions per thread).
// On return Am holds the eigenvalues on its diagonal and Vm the eigenvectors in its columns (row stride n).
// buf: 4 * 256 doubles.
DEV int jacobi_small(double *Am, double *Vm, int n, int tid, int nthreads, double *buf, double2 *cs, double *scratch) {
  const int np = n + (n & 1), half = np / 2;
  double *Ac = buf, *An = buf + 256, *Vc = buf + 512, *Vn = buf + 768;
  const bool act = tid < np * np;
  const int u = act ? tid / np : 0, v = act ? tid % np : 0;
  const int ku = u >> 1, bu = u & 1, kv = v >> 1, bv = v & 1;
  auto idx0 = [&](int p) { return (p & 1) ? np - 1 - (p >> 1) : (p >> 1); };  // position -> index at step 0
  auto sigma = [&](int p) {                                                   // position after the ring move
    const int k = p >> 1;
    if (!(p & 1)) return k >= 1 ? 2 * (k - 1) : 3;
    if (k == 0) return 1;
    return k <= half - 2 ? 2 * (k + 1) + 1 : 2 * (half - 1);
  };
  if (act) {
    const int iu = idx0(u), iv = idx0(v);
    Ac[tid] = (iu < n && iv < n) ? Am[iu * n + iv] : 0.0;
    Vc[tid] = (u == iv) ? 1.0 : 0.0;  // V rows keep their index, columns live in position space
  }
  const int dst_a = sigma(u) * np + sigma(v), dst_v = u * np + sigma(v);
  const int blk = (2 * ku) * np + 2 * kv;
  __syncthreads();
  int sweeps = 0, g = 0;
  double prev_off = 1e300;
  for (int sweep = 0; sweep < JMAX_SWEEPS; sweep++) {
    double off = 0, dia = 0;
    if (act) {
      const double a = Ac[tid];
      if (u == v) dia = a * a;
      else off = a * a;
    }
    off = wave_sum(off), dia = wave_sum(dia);
    if ((tid & 63) == 0) scratch[tid >> 6] = off, scratch[16 + (tid >> 6)] = dia;
    __syncthreads();
    double so = 0, sd = 0;
    for (int w = 0; w < nthreads / 64; w++) so += scratch[w], sd += scratch[16 + w];
    __syncthreads();
    if (so <= 1e-24 * sd || so == 0.0) break;
    if (sweep >= 4 && so > 0.25 * prev_off) break;  // rounding floor reached
    prev_off = so;
    sweeps++;
    for (int step = 0; step < np - 1; step++, g++) {
      if (tid < half) {
        const int p = 2 * tid, q = p + 1;
        double c = 1.0, s = 0.0;
        if (tid > 0 || np == n) jacobi_angle(Ac[p * np + p], Ac[q * np + q], Ac[q * np + p], c, s);  // bot 0 of an odd n is the dummy
        cs[tid] = make_double2(c, s);
      }
      __syncthreads();
      if (act) {
        const double2 ra = cs[ku], rb = cs[kv];
        const double ru0 = bu ? ra.y : ra.x, ru1 = bu ? ra.x : -ra.y;  // (new_p, new_q) = (c x_p - s x_q, s x_p + c x_q)
        const double cv0 = bv ? rb.y : rb.x, cv1 = bv ? rb.x : -rb.y;
        const double a00 = Ac[blk], a01 = Ac[blk + 1], a10 = Ac[blk + np], a11 = Ac[blk + np + 1];
        An[dst_a] = ru0 * fma(a00, cv0, a01 * cv1) + ru1 * fma(a10, cv0, a11 * cv1);
        Vn[dst_v] = fma(Vc[u * np + 2 * kv], cv0, Vc[u * np + 2 * kv + 1] * cv1);
      }
      __syncthreads();
      double *t = Ac;
      Ac = An, An = t, t = Vc, Vc = Vn, Vn = t;
    }
  }
  if (act) {
    int p, q;
    rr_pair(kv, g % (np - 1), np, n, p, q);
    const int iv = bv ? q : p;  // index held by position v now (-1: dummy)
    if (u == v && iv >= 0) Am[iv * n + iv] = Ac[tid];
    if (u < n && iv >= 0) Vm[u * n + iv] = Vc[tid];
  }
  __syncthreads();
  return sweeps;
}

// ---------------------------------------------------------------------------------------------------------------
// The n x n (n <= 76) symmetric eigen-problem of marginalization_factor.cpp:283-291 the way Eigen's SelfAdjointEigenSolver
// and the oracle's tred2 + tql2 pose it — tridiagonalize, solve the tridiagonal problem, transform back — with the serial
// QL iteration replaced by two steps that are parallel over the eigenvalues:
//   1. Householder tridiagonalization  A = Q T Q^T  (dsytd2's recurrences), the matrix in the registers of four waves, two
//      barriers per column (see the step itself in eig_tridiag).
//   2. eigenvalues of T by multisection of the Sturm count: one shared sweep over an even grid of 256 shifts brackets
//      every eigenvalue to 1/257 of the Gershgorin interval, then three lanes per eigenvalue quarter their bracket 23 times,
//      down to the rounding level of T.
//   3. eigenvectors of T by the twisted factorization (Parlett & Dhillon; LAPACK dlar1v without the relatively robust
//      representation): one thread per eigenvalue runs the stationary, another the progressive quotient-difference
//      recurrence; they twist where |gamma| is smallest and each multiplies out on its side of the twist.  Eigenvalues
//      closer than the rounding level give parallel vectors; that only happens in the block of noise eigenvalues
//      (|lambda| ~ 1e-6 against ||A'|| ~ 1e6), whose rows contribute lambda v v^T ~ 1e-6 to J0^T J0 whatever v is.
//   4. V = Q Z: eight lanes per eigenvector push their column through the reflectors, last to first (four reflectors per
//      step, only the rows they touch), and write
//      J0 = sqrt(S) V^T,  r0 = sqrt(1/S) V^T b'  for the eigenvalues above eps, zero rows for the others.
// Rows of J0 come out in ascending eigenvalue order, as Eigen returns them.
// ---------------------------------------------------------------------------------------------------------------
// Sturm count of T - sigma I (number of eigenvalues below sigma) by the determinant recurrence
// p_i = (d_i - sigma) p_{i-1} - e_{i-1}^2 p_{i-2}: two dependent flops per row instead of a division.  T is read through
// the scalar cache (uniform addresses -> s_load, no LDS or vector-memory traffic: every lane of ten waves walks the same
// 152 numbers 18 times), the signs are shifted into a mask and counted at the end, and the pair is brought back to unit
// scale every eight rows (one row grows it by at most ||T||^2).
typedef const __attribute__((address_space(4))) double sdouble;
#define STURM_TOUCH8(v) asm volatile("" ::"s"(v[0]), "s"(v[1]), "s"(v[2]), "s"(v[3]), "s"(v[4]), "s"(v[5]), "s"(v[6]), "s"(v[7]))
DEV int sturm_count(sdouble *Tg /* d[96] | e^2[96]; beyond n: d above every shift, e^2 = 0 */, int n, double sigma) {
  // five instructions per row: d - sigma, e^2 p_{i-2}, the fma, and the sign change shifted into a word that is
  // counted once per 16 rows.  Rows beyond n keep the sign (d - sigma > 0, e^2 = 0), so they need no mask; the pair is
  // brought back to unit scale every 16 rows (one row grows it by at most 2 ||T|| + 1).
  // The scalar loads are software-pipelined in blocks of eight rows: scalar loads return out of order, so a wait for one
  // block is a wait for everything outstanding — the block in hand is therefore waited for (STURM_TOUCH8) BEFORE the next
  // one is requested, and the eight rows of arithmetic then run beside that request.
  double pm = 1.0, p = Tg[0] - sigma;
  unsigned cnt = (unsigned)__double2hiint(p) >> 31;
  // the sign of every p_i is shifted into ONE word that is never cleared (newest at bit 0): after a block of 16 rows its
  // lower half holds the block's signs and bit 16 the sign before the block, so the sign changes are the set bits of
  // (bits ^ (bits >> 1)) in the lower half — one integer instruction per row instead of two
  unsigned bits = (unsigned)__double2hiint(p) >> 31;
  double d0[8], e0[8], d1[8], e1[8];
#pragma unroll
  for (int u = 0; u < 8; u++) d0[u] = Tg[1 + u], e0[u] = Tg[96 + u];
  for (int i0 = 1; i0 < n; i0 += 16) {
    STURM_TOUCH8(d0);
    STURM_TOUCH8(e0);
#pragma unroll
    for (int u = 0; u < 8; u++) d1[u] = Tg[i0 + 8 + u], e1[u] = Tg[96 + i0 + 8 + u - 1];
#pragma unroll
    for (int u = 0; u < 8; u++) {
      const double pn = fma(d0[u] - sigma, p, -e0[u] * pm);
      bits = __builtin_amdgcn_alignbit(bits, (unsigned)__double2hiint(pn), 31);
      pm = p, p = pn;
    }
    STURM_TOUCH8(d1);
    STURM_TOUCH8(e1);
#pragma unroll
    for (int u = 0; u < 8; u++) d0[u] = Tg[i0 + 16 + u], e0[u] = Tg[96 + i0 + 16 + u - 1];  // (up to row 88 < 96)
#pragma unroll
    for (int u = 0; u < 8; u++) {
      const double pn = fma(d1[u] - sigma, p, -e1[u] * pm);
      bits = __builtin_amdgcn_alignbit(bits, (unsigned)__double2hiint(pn), 31);
      pm = p, p = pn;
    }
    cnt += __builtin_popcount((bits ^ (bits >> 1)) & 0xffffu);
    const int ex = ilogb(fmax(fmax(fabs(p), fabs(pm)), 1e-300));
    p = ldexp(p, -ex), pm = ldexp(pm, -ex);
  }
  return (int)cnt;
}
// ---- Householder tridiagonalization steps of eig_tridiag (see there); P = leading live tile of the 5 x 5 register tiles
struct TriCtx {
  double *dd, *ee, *tau, *vb0, *vb1, *pp, *RV;
  int n, r0, c0;
};
// d[kk] and the reflector of column kk (or the last e) from row kk, which is final.  Every row group of 16 lanes runs the
// arithmetic on ITS row of tile P (so that the chain sum -> rsqrt -> reciprocal is straight-line code the scheduler can put
// beside the rank-2 update of the other tile rows); only the group that owns row kk (own) stores anything.
template <int P>
DEV void tri_finish_row(const TriCtx &tc, int kk, const double (&row)[5], bool own) {
  const int n = tc.n, c0 = tc.c0;
  double *vb = (kk & 1) ? tc.vb1 : tc.vb0;
  double a_part = 0.0, ss_part = 0.0, dk = 0.0;
  bool has_d = false;
#pragma unroll
  for (int q = P; q < 5; q++) {
    const int C = c0 + 16 * q;
    if (C == kk) dk = row[q], has_d = true;
    if (C == kk + 1) a_part = row[q];
    if (C > kk + 1 && C < n) ss_part = fma(row[q], row[q], ss_part);
  }
  if (own && has_d) tc.dd[kk] = dk;
  if (kk + 2 >= n) {
    if (own && kk == n - 2 && c0 == ((n - 1) & 15)) tc.ee[kk] = a_part;
    return;
  }
  a_part = sum8(a_part), ss_part = sum8(ss_part);
  const double alpha = a_part + dpp_f64<0x140>(a_part), ss = ss_part + dpp_f64<0x140>(ss_part);
  const double nn = fma(alpha, alpha, ss);
  const bool live = ss > 0.0;
  const double rs = fast_rsqrt(live ? nn : 1.0);
  const double beta = live ? -copysign(nn * rs, alpha) : alpha;
  const double t = live ? (beta - alpha) * -copysign(rs, alpha) : 0.0;  // 1 / beta = -sign(alpha) / sqrt(nn)
  const double scal = live ? fast_rcp(alpha - beta) : 0.0;
  if (own) {
#pragma unroll
    for (int q = P; q < 5; q++) {
      const int C = c0 + 16 * q;
      const double v = C == kk + 1 ? 1.0 : ((C > kk + 1 && C < n) ? row[q] * scal : 0.0);
      vb[C] = v;
      if (C > kk && C < n) tc.RV[kk * 76 + C - kk - 1] = v;
    }
    if (c0 == 0) tc.tau[kk] = t, tc.ee[kk] = beta;
  }
}
// p = tau A v (zero for the rows that are done)
template <int P>
DEV void tri_matvec(const TriCtx &tc, int k, const double (&ar)[5][5], double (&vj)[5]) {
  const double *vb = (k & 1) ? tc.vb1 : tc.vb0;
  const double t = tc.tau[k];
#pragma unroll
  for (int q = P; q < 5; q++) vj[q] = vb[tc.c0 + 16 * q];
  double sacc[5];
#pragma unroll
  for (int i = P; i < 5; i++) {
    sacc[i] = 0.0;
#pragma unroll
    for (int q = P; q < 5; q++) sacc[i] = fma(ar[i][q], vj[q], sacc[i]);
  }
#pragma unroll
  for (int i = P; i < 5; i++) sacc[i] += dpp_f64<0xB1>(sacc[i]);
#pragma unroll
  for (int i = P; i < 5; i++) sacc[i] += dpp_f64<0x4E>(sacc[i]);
#pragma unroll
  for (int i = P; i < 5; i++) sacc[i] += dpp_f64<0x141>(sacc[i]);
#pragma unroll
  for (int i = P; i < 5; i++) sacc[i] += dpp_f64<0x140>(sacc[i]);
  if (tc.c0 == 0) {
#pragma unroll
    for (int i = P; i < 5; i++) tc.pp[tc.r0 + 16 * i] = tc.r0 + 16 * i > k ? t * sacc[i] : 0.0;
  }
}
// A -= v w^T + w v^T with w = p - K v, K = (tau/2)(p.v), written as A -= v p^T + p v^T - 2 K v v^T: the first two terms
// do not wait for the reduction of p.v; v and p are zero outside k+1 .. n-1.  The tile row that holds row k + 1 goes first
// and the next reflector is formed from it while the other tile rows take their update.
template <int P>
DEV void tri_update(const TriCtx &tc, int k, double (&ar)[5][5], const double (&vj)[5]) {
  const double *vb = (k & 1) ? tc.vb1 : tc.vb0;
  const double t = tc.tau[k];
  double pj[5], vi[5], pi[5], pvs = 0.0;
#pragma unroll
  for (int q = P; q < 5; q++) pj[q] = tc.pp[tc.c0 + 16 * q];
#pragma unroll
  for (int i = P; i < 5; i++) vi[i] = vb[tc.r0 + 16 * i], pi[i] = tc.pp[tc.r0 + 16 * i];
#pragma unroll
  for (int q = P; q < 5; q++) pvs = fma(pj[q], vj[q], pvs);
  pvs = sum8(pvs);
  pvs += dpp_f64<0x140>(pvs);
  const double K2 = t * pvs;
#pragma unroll
  for (int i = P; i < 5; i++) {
#pragma unroll
    for (int q = P; q < 5; q++) ar[i][q] = fma(-pi[i], vj[q], fma(-vi[i], pj[q], ar[i][q]));
  }
#pragma unroll
  for (int i = P; i < 5; i++) {
    const double kv = K2 * vi[i];
#pragma unroll
    for (int q = P; q < 5; q++) ar[i][q] = fma(kv, vj[q], ar[i][q]);
    if (i == P) tri_finish_row<P>(tc, k + 1, ar[P], tc.r0 == ((k + 1) & 15));
  }
}
// z <- (I - tau v v^T) z for the eight lanes that share an eigenvector (rows l8 + 8 q), for the NR reflectors k, k - 1, ...
// of one tile QM = k / 8 (the first tile with live rows): their loads leave together, so the LDS round trip is paid once
// per group.  Rows beyond n read the zero tail of the reflector's RV row — except for k < 3, where that tail is shorter
// than the tile, so the first instantiation keeps the upper mask.
template <int QM, int NR>
DEV void bt_apply(const double *RV, const double *tau, int k, int n, int l8, double (&zz)[10]) {
  double vq[NR][10], tk[NR];
#pragma unroll
  for (int r = 0; r < NR; r++) {
    const int kr = k - r;
    const double *v = RV + kr * 76 - (kr + 1);  // v[i] for rows i >= kr + 1
    tk[r] = tau[kr];
#pragma unroll
    for (int q = QM; q < 10; q++) {
      const int i = l8 + 8 * q;
      if (QM == 0) vq[r][q] = (i > kr && i < n) ? v[i] : 0.0;
      else vq[r][q] = (q > QM || i > kr) ? v[i] : 0.0;
    }
  }
#pragma unroll
  for (int r = 0; r < NR; r++) {
    double s0 = 0.0, s1 = 0.0;
#pragma unroll
    for (int q = QM; q < 10; q++) {
      if ((q - QM) & 1) s1 = fma(vq[r][q], zz[q], s1);
      else s0 = fma(vq[r][q], zz[q], s0);
    }
    const double s = sum8(s0 + s1) * tk[r];
#pragma unroll
    for (int q = QM; q < 10; q++) zz[q] = fma(-s, vq[r][q], zz[q]);
  }
}
template <int NR>
DEV void bt_dispatch(const double *RV, const double *tau, int k, int n, int l8, double (&zz)[10]) {
  switch (k >> 3) {
    case 0: bt_apply<0, NR>(RV, tau, k, n, l8, zz); break;
    case 1: bt_apply<1, NR>(RV, tau, k, n, l8, zz); break;
    case 2: bt_apply<2, NR>(RV, tau, k, n, l8, zz); break;
    case 3: bt_apply<3, NR>(RV, tau, k, n, l8, zz); break;
    case 4: bt_apply<4, NR>(RV, tau, k, n, l8, zz); break;
    case 5: bt_apply<5, NR>(RV, tau, k, n, l8, zz); break;
    case 6: bt_apply<6, NR>(RV, tau, k, n, l8, zz); break;
    case 7: bt_apply<7, NR>(RV, tau, k, n, l8, zz); break;
    case 8: bt_apply<8, NR>(RV, tau, k, n, l8, zz); break;
    default: bt_apply<9, NR>(RV, tau, k, n, l8, zz); break;
  }
}
// |q| < pivmin -> +-pivmin (the sign of q): the quotient-difference recurrences divide by q.  max + sign insertion instead of
// compare + select: a v_cmp_f64 -> SGPR mask -> v_cndmask round trip costs the chain several times what the arithmetic does.
DEV double guard_pivot(double q, double pivmin) { return copysign(fmax(fabs(q), pivmin), q); }
// A: n x n, row stride LDN, full symmetric (destroyed; receives Z, component i of vector m at [i * LDN + m]); b: n;
// RV: >= n * 76, DM: >= n * LDN, vec: >= 7 * 96, nn2: >= 2 * 96 doubles of LDS.  Writes out->linearized_jacobians / _residuals.
// sp (a worker's launch, kernels_spec.h): thread MARG_THREADS - 1 — idle in phases 1 to 3 — polls whether the state this prior belongs
// to has been overtaken; returns true (nothing written) when it has.
template <bool SPEC>
DEV bool eig_tridiag(double *A, const double *b, int n, int tid, double *RV, double *DM, double *vec, double *scr, double *nn2, double *Tglob, LfvioPrior *out, double eps, long long *dbg,
                     SpecPoll *sp_) {
  SpecPoll *const sp = SPEC ? sp_ : nullptr;  // (the plain instantiation carries none of it)
  const bool poller = SPEC && tid == MARG_THREADS - 1;
#define ESTAMP(k) do { if (tid == 0) dbg[k] = (long long)__builtin_readcyclecounter(); } while (0)
  ESTAMP(26);
  const int wave = tid >> 6, lane = tid & 63;
  double *dd = vec, *ee = vec + 96, *ee2 = vec + 192, *tau = vec + 288, *lam = vec + 384, *pp = vec + 480, *nrm = vec + 576;
  // ---- 1. tridiagonalization, the matrix in registers of the first four waves (one per SIMD): thread (r0, c0) of a
  // 16 x 16 grid owns A[r0 + 16 i][c0 + 16 q] (i, q < 5; both triangles).  LDS carries only vectors: the reflector v of the
  // current column (normalized and zero up to the diagonal, so nothing downstream needs a mask), p = tau A v, the
  // reflectors for the back-transformation.  A row sits in one DPP row of 16 lanes: A v, p.v and the norm of the next
  // reflector are reduced with four DPP steps.  The step is a latency chain (two barriers and four LDS round trips per
  // column) with the vector pipe underneath it, so the instruction count is what is kept small: four waves instead of
  // twelve carry the per-thread overhead once per SIMD, and column k only touches the 16 x 16 register tiles that still
  // hold live rows and columns (tri_matvec / tri_update are instantiated per leading tile P = (k + 1) / 16: 25, 16, 9, 4, 1
  // tiles).  The 16 lanes that own row k + 1 form the NEXT reflector (dlarfg) right after they have updated that row,
  // inside the update phase.  Only the lower triangle of the input is read, like Eigen's solver (and tred2) do.
  const bool tri = tid < 256;
  const int r0 = (tid >> 4) & 15, c0 = tid & 15;
  double ar[5][5];
#pragma unroll
  for (int i = 0; i < 5; i++)
#pragma unroll
    for (int q = 0; q < 5; q++) {
      const int R = r0 + 16 * i, C = c0 + 16 * q;
      ar[i][q] = (tri && R < n && C < n) ? A[max(R, C) * LDN + min(R, C)] : 0.0;
    }
  double *vb0 = lam, *vb1 = nrm;  // free until the eigenvalue search
  if (tid < 96) dd[tid] = 0.0, ee[tid] = 0.0, ee2[tid] = 0.0, tau[tid] = 0.0, vb0[tid] = 0.0, vb1[tid] = 0.0;  // v stays zero beyond n
  for (int e = tid; e < 76 * 76; e += MARG_THREADS) RV[e] = 0.0;  // the back-transformation reads the tail of a row as rows >= n
  __syncthreads();
  TriCtx tc = {dd, ee, tau, vb0, vb1, pp, RV, n, r0, c0};
  if (tri) tri_finish_row<0>(tc, 0, ar[0], r0 == 0);
  __syncthreads();
  for (int k = 0; k + 2 < n; k++) {
    const int P = (k + 1) >> 4;
    double vj[5];
    if (poller) {  // (six columns between the loads and their use: the poller's wave must not reach a barrier waiting for them)
      if ((k & 7) == 0) spec_poll_issue(*sp);
      else if ((k & 7) == 6) spec_poll_take(*sp);
    }
#ifdef LFVIO_TRI_PROFILE
#define TSTAMP(j) do { if (tid == 0 && k == LFVIO_TRI_PROFILE) dbg[3 + j] = (long long)__builtin_readcyclecounter(); } while (0)
#else
#define TSTAMP(j) do { } while (0)
#endif
    TSTAMP(0);
    if (tri) {
      switch (P) {
        case 0: tri_matvec<0>(tc, k, ar, vj); break;
        case 1: tri_matvec<1>(tc, k, ar, vj); break;
        case 2: tri_matvec<2>(tc, k, ar, vj); break;
        case 3: tri_matvec<3>(tc, k, ar, vj); break;
        default: tri_matvec<4>(tc, k, ar, vj); break;
      }
    }
    TSTAMP(1);
    __syncthreads();
    TSTAMP(2);
    if (tri) {
      switch (P) {
        case 0: tri_update<0>(tc, k, ar, vj); break;
        case 1: tri_update<1>(tc, k, ar, vj); break;
        case 2: tri_update<2>(tc, k, ar, vj); break;
        case 3: tri_update<3>(tc, k, ar, vj); break;
        default: tri_update<4>(tc, k, ar, vj); break;
      }
    }
    TSTAMP(3);
    __syncthreads();
    TSTAMP(4);
    if (sp && (k & 7) == 6 && *sp->flag) return true;  // (uniform: written in front of this step's first barrier)
  }
  if (tri) {
#pragma unroll
    for (int i = 0; i < 5; i++)
#pragma unroll
      for (int q = 0; q < 5; q++)
        if (r0 + 16 * i == n - 1 && c0 + 16 * q == n - 1) dd[n - 1] = ar[i][q];
  }
  if (poller) spec_poll_issue(*sp);
  __syncthreads();
  ESTAMP(27);
  // ---- 2. eigenvalues
  if (tid < n) ee2[tid] = ee[tid] * ee[tid];
  {
    double lo = 1e300, hi = -1e300;
    if (tid < n) {
      const double r = (tid > 0 ? fabs(ee[tid - 1]) : 0.0) + (tid + 1 < n ? fabs(ee[tid]) : 0.0);
      lo = dd[tid] - r, hi = dd[tid] + r;
    }
    lo = -wave_max(-lo), hi = wave_max(hi);
    if (lane == 0) scr[wave] = lo, scr[16 + wave] = hi;
  }
  __syncthreads();
  double gl = fmin(scr[0], scr[1]), gu = fmax(scr[16], scr[17]);  // n <= 76: the first two waves hold everything
  {
    const double tn = fmax(fabs(gl), fabs(gu));
    gl -= 2.2e-16 * tn * n + 1e-300, gu += 2.2e-16 * tn * n + 1e-300;
  }
  // T for the scalar loads; the rows that pad n to the unroll length sit above every shift and are decoupled
  if (tid < 96) Tglob[tid] = tid < n ? dd[tid] : gu + 1.0, Tglob[96 + tid] = tid + 1 < n ? ee[tid] * ee[tid] : 0.0;
  __threadfence();
  __syncthreads();
  __builtin_amdgcn_s_dcache_inv();
  const unsigned long long ta = (unsigned long long)Tglob;
  unsigned tlo = __builtin_amdgcn_readfirstlane((unsigned)ta), thi = __builtin_amdgcn_readfirstlane((unsigned)(ta >> 32));
  asm volatile("" : "+s"(tlo), "+s"(thi));  // the scalar loads stay behind the barrier above
  sdouble *Tg = (sdouble *)(((unsigned long long)thi << 32) | tlo);
  {
    const double pivmin = 1e-290;
    // three lanes per eigenvalue (quadrisection, 27 rounds), 21 eigenvalues per wave: four waves, one per SIMD — the
    // recurrence is bound by instruction issue, so the work is kept small rather than the round count (eight lanes per
    // eigenvalue need 18 rounds but ten waves)
    const int g3 = lane == 63 ? 20 : lane / 3, l8 = lane == 63 ? 3 : lane - 3 * g3;  // lane 63 rides along with group 20
    const int m = wave * 21 + g3;
    const int mm = m < n ? m : n - 1;
    // round 0 is shared: the 256 threads put an even grid over the Gershgorin interval, and the counts at the grid points
    // bracket every eigenvalue to 1/257 of it with one sweep (four rounds of the per-eigenvalue quadrisection)
    {
      int *cntg = (int *)pp;             // 256 counts (pp | nrm are free until the eigenvectors)
      double *blo = nn2, *bhi = nn2 + 96;  // bracket of eigenvalue m
      const double step = (gu - gl) * (1.0 / 257.0);
      if (tid < 96) blo[tid] = gl, bhi[tid] = gu;  // (whatever the counts say, every eigenvalue has a bracket)
      if (tid < 256) cntg[tid] = sturm_count(Tg, n, gl + step * (double)(tid + 1));
      __syncthreads();
      if (tid <= 256) {  // thread j owns the eigenvalues between grid points j - 1 and j
        const int c0 = tid > 0 ? cntg[tid - 1] : 0, c1 = tid < 256 ? cntg[tid] : n;
        const double a = tid > 0 ? gl + step * (double)tid : gl, b = tid < 256 ? gl + step * (double)(tid + 1) : gu;
        for (int e = max(c0, 0); e < min(c1, n); e++) blo[e] = a, bhi[e] = b;
      }
      __syncthreads();
    }
    double lo = nn2[mm], hi = nn2[96 + mm];
    if (tid < 256)
      for (int round = 0; round < 23; round++) {
        const double w = hi - lo;
        const double sigma = lo + w * (double)(l8 < 3 ? l8 + 1 : 1) * 0.25;
        const int cnt = sturm_count(Tg, n, sigma);
        const int f1 = cnt <= mm ? 1 : 0;  // eigenvalue mm (0-based, ascending) is >= sigma
        // how many of the group's three shifts lie at or below the eigenvalue: three bits of the wave's ballot (a shuffle
        // per lane of the group would be three LDS-crossbar round trips per round)
        const unsigned long long bal = __ballot(f1);
        const int f = __builtin_popcount((unsigned)(bal >> (3 * g3)) & 7u);
        const double nlo = f == 0 ? lo : lo + w * (double)f * 0.25, nhi = f == 3 ? hi : lo + w * (double)(f + 1) * 0.25;
        lo = nlo, hi = nhi;
      }
    if (tid < 256 && m < n && l8 == 0) lam[m] = 0.5 * (lo + hi);
    if (poller) spec_poll_take(*sp), spec_poll_issue(*sp);
    __syncthreads();
    ESTAMP(28);
    if (sp && *sp->flag) return true;
    // ---- 3. eigenvectors of T: thread me (set 0) runs the stationary recurrence (top down) into A, thread 128 + me (set 1)
    //         the progressive one (bottom up) into DM; each set then looks for the twist in its half of the rows and
    //         multiplies out on its side of it.  Every loop has a uniform trip count (rows on the wrong side of the twist are
    //         predicated off), so the loads of several rows leave together.
    const int set = tid >> 7, me = tid & 127;
    const bool work = set < 2 && me < n;
    const int mec = work ? me : 0;
    const double l = lam[mec];
    // Every loop below runs in blocks of eight rows: the LDS reads of a block leave together, then the chain runs on
    // registers, then the block is stored — a read issued inside the chain would cost it a round trip per row (LDS
    // operations complete in order, so it would also wait for the store of the row before).
    if (work && set == 0) {
      double q = dd[0] - l;
      A[mec] = q;
      for (int i0 = 1; i0 < n; i0 += 8) {
        double dv[8], ev[8], qs[8];
#pragma unroll
        for (int u = 0; u < 8; u++) dv[u] = dd[i0 + u], ev[u] = ee2[i0 + u - 1];  // (rows up to n + 7 < 96 exist and are zero)
#pragma unroll
        for (int u = 0; u < 8; u++) {
          q = fma(-ev[u], fast_rcp(guard_pivot(q, pivmin)), dv[u] - l);
          qs[u] = q;
        }
#pragma unroll
        for (int u = 0; u < 8; u++)
          if (i0 + u < n) A[(i0 + u) * LDN + mec] = qs[u];
      }
    } else if (work) {
      double q = dd[n - 1] - l;
      DM[(n - 1) * LDN + mec] = q;
      for (int i0 = n - 2; i0 >= 0; i0 -= 8) {
        double dv[8], ev[8], qs[8];
#pragma unroll
        for (int u = 0; u < 8; u++) dv[u] = dd[max(i0 - u, 0)], ev[u] = ee2[max(i0 - u, 0)];
#pragma unroll
        for (int u = 0; u < 8; u++) {
          q = fma(-ev[u], fast_rcp(guard_pivot(q, pivmin)), dv[u] - l);
          qs[u] = q;
        }
#pragma unroll
        for (int u = 0; u < 8; u++)
          if (i0 - u >= 0) DM[(i0 - u) * LDN + mec] = qs[u];
      }
    }
    __syncthreads();
    ESTAMP(16);
    {  // |gamma_i| = |s_i + p_i - (d_i - lambda)| is smallest at the twist: each set scans half of the rows
      const int half = (n + 1) >> 1, i0 = set == 0 ? 0 : half, i1 = set == 0 ? half : n;
      double gmin = 1e300;
      int kt = i0;
      if (work) {
#pragma unroll 8
        for (int i = i0; i < i1; i++) {
          const double g = fabs(A[i * LDN + mec] + DM[i * LDN + mec] - (dd[i] - l));
          if (g < gmin) gmin = g, kt = i;
        }
      }
      __syncthreads();  // (every thread has read e^2 by now)
      if (work) {
        pp[set * 96 + mec] = gmin;
        ((int *)ee2)[set * 96 + mec] = kt;  // (pp | nrm: 192 doubles, free by now)
      }
    }
    if (poller) spec_poll_take(*sp);
    __syncthreads();
    ESTAMP(17);
    if (sp && *sp->flag) return true;
    if (work) {
      const double g0 = pp[mec], g1 = pp[96 + mec];
      const int kt = g1 < g0 ? ((int *)ee2)[96 + mec] : ((int *)ee2)[mec];  // the first minimum, as one scan would find it
      double z = 1.0, nn = 0.0;
      if (set == 0) {
        A[kt * LDN + mec] = 1.0;
        for (int i0 = n - 2; i0 >= 0; i0 -= 8) {
          double c[8], zs[8];
#pragma unroll
          for (int u = 0; u < 8; u++) {
            const int i = max(i0 - u, 0);
            c[u] = -ee[i] * fast_rcp(guard_pivot(A[i * LDN + mec], pivmin));  // (not on the chain of z: one multiplication per row)
          }
#pragma unroll
          for (int u = 0; u < 8; u++) {
            const bool on = i0 - u < kt && i0 - u >= 0;
            z = on ? c[u] * z : 1.0;
            zs[u] = z;
            nn = on ? fma(z, z, nn) : nn;
          }
#pragma unroll
          for (int u = 0; u < 8; u++)
            if (i0 - u < kt && i0 - u >= 0) A[(i0 - u) * LDN + mec] = zs[u];
        }
      } else {
        for (int i0 = 1; i0 < n; i0 += 8) {
          double c[8], zs[8];
#pragma unroll
          for (int u = 0; u < 8; u++) {
            const int i = min(i0 + u, n - 1);
            c[u] = -ee[i - 1] * fast_rcp(guard_pivot(DM[i * LDN + mec], pivmin));
          }
#pragma unroll
          for (int u = 0; u < 8; u++) {
            const bool on = i0 + u > kt && i0 + u < n;
            z = on ? c[u] * z : 1.0;
            zs[u] = z;
            nn = on ? fma(z, z, nn) : nn;
          }
#pragma unroll
          for (int u = 0; u < 8; u++)
            if (i0 + u > kt && i0 + u < n) A[(i0 + u) * LDN + mec] = zs[u];
        }
      }
      nn2[set * 96 + mec] = nn;
    }
  }
  __syncthreads();
  ESTAMP(29);
  // ---- 4. V = Q Z and the outputs
  {
    const int m = tid >> 3, l8 = tid & 7;
    const bool live = m < n;
    const int mm = live ? m : 0;
    double zz[10];
    const double sc = fast_rsqrt(1.0 + nn2[mm] + nn2[96 + mm]);
#pragma unroll
    for (int q = 0; q < 10; q++) {
      const int i = l8 + 8 * q;
      zz[q] = i < n ? A[i * LDN + mm] * sc : 0.0;
    }
    // reflector k only touches rows i > k: the loop runs per leading live tile QM = k / 8 (bt_apply<QM>), the tiles below
    // it are skipped and only the boundary tile is masked (the tail of every RV row is zero, see the tridiagonalization)
    int k = n - 3;
    for (; k >= 0 && (k & (BT_GROUP - 1)) != BT_GROUP - 1; k--) bt_dispatch<1>(RV, tau, k, n, l8, zz);
    for (; k >= 0; k -= BT_GROUP) bt_dispatch<BT_GROUP>(RV, tau, k, n, l8, zz);
    const double l = lam[mm];
    double rb = 0.0;
#pragma unroll
    for (int q = 0; q < 10; q++) {
      const int i = l8 + 8 * q;
      if (i < n) rb = fma(zz[q], b[i], rb);
    }
    rb = sum8(rb);
    if (live) {
      const bool keep = l > eps;
      const double sq = keep ? sqrt(l) : 0.0;
#pragma unroll
      for (int q = 0; q < 10; q++) {
        const int i = l8 + 8 * q;
        if (i < n) out->linearized_jacobians[m * n + i] = sq * zz[q];
      }
      if (l8 == 0) out->linearized_residuals[m] = keep ? rb / sq : 0.0;
    }
  }
  return false;
}

// End of a worker's k_marg_solve (kernels_spec.h): the prior of the state `my_word` lies finished in the shadow slot S.  Wait for
// the loop of S0 to close (or for a newer accepted state); if this state is the final one the loop has committed the prior to us:
// it moves into S0->prior_out, tail_state 3 -> 2, the caller's ticket is echoed (and the mailbox filled when the call hands its
// results over early).  Every thread of the workgroup comes here, behind a barrier that follows the last store to S->prior_out.
DEV void spec_deliver(Slot *S, Slot *S0, int my_word, bool publish) {
  const int tid = threadIdx.x, nthr = blockDim.x;
  __shared__ int give;
  if (tid == 0) {
    const int ep = spec_ep(my_word), a = spec_acc(my_word);
    long long t0 = wall_clock64();
    int g = 0;
    for (;;) {
      const int f = spec_peek(&S0->spec.fin), w = spec_peek(&S0->spec.word);
      if (spec_ep(w) != ep) break;  // (the slot has gone on to another call: cannot happen to a committed worker)
      const int st = spec_ep(f) == ep ? spec_fin_state(f) : FIN_OPEN;
      if (st >= FIN_MAIN) {
        g = st == FIN_SIDE && spec_fin_acc(f) == a;
        break;
      }
      if (w != my_word) break;  // a newer accepted state: the next round's
      if (wall_clock64() - t0 > SPEC_WAIT_TICKS) {
        if (spec_cas(&S0->spec.own[a], SPEC_SIDE, SPEC_ABANDON) == SPEC_SIDE) break;
        t0 = wall_clock64();  // the loop has committed this prior: its FIN_SIDE is on the way
      }
      __builtin_amdgcn_s_sleep(8);
    }
    give = g;
  }
  __syncthreads();
  if (!give) return;
  const LfvioPrior *src = &S->prior_out;
  LfvioPrior *dst = &S0->prior_out;
  const int n = src->valid == 1 ? src->n : 0;
  // (the worker's last step, on the call's critical path: rounds of loads)
  copy_rounds((long long *)dst, (const long long *)src, (int)(offsetof(LfvioPrior, linearized_jacobians) / 8), tid, nthr);
  copy_rounds(dst->linearized_jacobians, src->linearized_jacobians, n * n, tid, nthr);
  copy_rounds(dst->linearized_residuals, src->linearized_residuals, n, tid, nthr);
  // ONE release for both readers — the next kernel of stream 0 that looks at tail_state (device) and the host behind the echo
  // (system) — by ONE thread, behind a barrier that has waited for every thread's stores: 768 write-backs of the same cache, one after
  // the other, were a tenth of this kernel
  __syncthreads();
  if (publish) publish_prior(S0);  // (ends behind a system-scope fence and a barrier)
  char *m = (char *)S0->mail;
  if (tid == 0) {
    __threadfence_system();
    __hip_atomic_store(&S0->tail_state, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (m) __hip_atomic_store((int *)m + 7, S->shadow.ticket, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

// grid (1, batch) x 256, dynamic LDS = MARG_LDS
// WORKER: the instantiation a worker of kernels_spec.h launches (polling, hand-over); the plain one is compiled without any of it.
template <bool WORKER>
__global__ __launch_bounds__(MARG_THREADS) void k_marg_solve(char *base, size_t stride, int flag_bits) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int flag = flag_bits & 255, force_eig = (flag_bits >> 8) & 1, gated = (flag_bits >> 9) & 1;  // bit 8: debug, see lfvio_debug_force_eig; bit 9: MODE_GATED
  const int publish = (flag_bits >> 10) & 1;  // bit 10: the prior goes into the caller's mailbox as well (lfvio_batch_optimize_begin)
  // WORKER: a worker's launch on the second stream (kernels_spec.h) — base is the shadow slot, `stride` its distance from the slot
  // being solved; the prior is handed over at the end if the state it belongs to turns out to be the final one
  constexpr bool worker = WORKER;
  Slot *S = worker ? (Slot *)base : SLOT(base, stride);
  Slot *S0 = worker ? (Slot *)(base - stride) : S;
  __shared__ int spec_flag;
  SpecPoll sp{S0, worker ? S->shadow.word : 0, &spec_flag, 0, 0};
  if (threadIdx.x == 0) spec_flag = 0;
  const bool poller = worker && threadIdx.x == MARG_THREADS - 1;
  TRState *tr = &S->tr;
  const MargPlan *mp = &S->marg[flag];
  const int tid = threadIdx.x;
  LfvioPrior *out = &S->prior_out;
  if (gated && !tail_gate(S, tr->done)) return;
  if (worker && sp.my_word == 0) return;
  if (!mp->valid) {
    // MARGIN_SECOND_NEW with no prior touching Pose[WINDOW_SIZE-1]: the prior is left as it is
    if (tid == 0) out->valid = -1;  // host copies the input prior through
    if (worker) {
      __syncthreads();
      spec_deliver(S, S0, sp.my_word, publish != 0);
      return;
    }
    if (gated && tid == 0) S->iters_done = tr->iteration, S->tail_state = 2;
    if (gated && publish) {
      __syncthreads();
      publish_prior(S);
    }
    return;
  }
  if (poller) spec_poll_issue(sp);
  STAMP(S, 10);
  // ---- the dense system over the present blocks, D = m15 + n, straight from the packed pose-side Hessian: entry
  // (a, b) of A is H(r, c) of the tangent columns the plan maps there, minus the frame-0 landmarks' Schur sums
  // (H -= sum c_l w_l w_l^T, g -= sum c_l b_l w_l) where both columns are pose / extrinsic / td columns
  const double *Sc = S->schur_sum;
  const bool sub = mp->N0 > 0 || S->sharded;  // sharded: the all-reduced sums hold the other ranks' landmarks
  const int m15 = mp->m15, n = mp->n, D = m15 + n;
  double *Ag = S->mscr;  // global scratch: A', b' are kept there for parity checks
  // LDS re-use (16.7k doubles): A dies once A' is formed, so the second eigenvector matrix aliases it
  double *A = smem;                        // D x D           (<= 92*92 = 8464)
  double *bv = A + 92 * 92;                // D
  double *Am = bv + 96;                    // m15 x m15 (then its eigenvalues on the diagonal)
  double *Vm = Am + 256;                   // m15 x m15
  double *Ainv = Vm + 256;                 // m15 x m15
  double *Tm = Ainv + 256;                 // n x m15
  double *Ar = smem + 12544;               // n x n, row stride LDN (above everything the tridiagonal path puts below it)
  double *br = Ar + 76 * LDN;              // n
  double2 *cs = (double2 *)(br + 80);      // [2][48] (cos, sin) of the Jacobi rotations, double-buffered
  int *perm = (int *)(cs + 96);            // n ints; first the inverse of the plan's column map
  double *scr = (double *)(perm + 96);     // 64
  // tridiagonal path: reflectors, the progressive recurrence and the small vectors live in the dead A / Tm area
  double *RV = smem, *vec8 = smem + 5776, *DMs = smem + 5776 + 672;  // 76 x 76, 7 x 96, 76 x LDN: ends at 12528 < 12544
  for (int c = tid; c < KP; c += MARG_THREADS) {
    const int a = mp->col[c];
    if (a >= 0) perm[a] = c;
  }
  __syncthreads();
  {  // every entry of this thread in ONE round of loads (a loop of load - subtract - store is a memory round trip per entry: eleven of them)
    constexpr int GR = (92 * 92 + MARG_THREADS - 1) / MARG_THREADS;
    const double *Hp = &S->Hpp[0];
    double hv[GR], sv[GR];
#pragma unroll
    for (int k = 0; k < GR; k++) {
      const int e = tid + MARG_THREADS * k, ec = e < D * D ? e : 0;
      const int r = perm[ec / D], c = perm[ec % D];
      const int hi = max(r, c), lo = min(r, c);
      hv[k] = Hp[hi * (hi + 1) / 2 + lo];
      sv[k] = (sub && hi < KC) ? Sc[schur_index(lo, hi)] : 0.0;
    }
#pragma unroll
    for (int k = 0; k < GR; k++) {
      const int e = tid + MARG_THREADS * k;
      if (e < D * D) A[e] = hv[k] - sv[k];
    }
  }
  for (int a = tid; a < D; a += MARG_THREADS) {
    const int c = perm[a];
    double v = S->gp[c];
    if (sub && c < KC) v -= Sc[schur_index(c, COL_B)];
    bv[a] = v;
  }
  if (poller) spec_poll_take(sp), spec_poll_issue(sp);
  __syncthreads();
  STAMP(S, 11);
  if (worker && spec_flag) return;  // (overtaken: the next round takes the newer state)
  // ---- A_mm pseudo-inverse by eigen-decomposition (marginalization_factor.cpp:267-272)
  for (int e = tid; e < m15 * m15; e += MARG_THREADS) {
    const int r = e / m15, c = e % m15;
    Am[e] = 0.5 * (A[r * D + c] + A[c * D + r]);
  }
  __syncthreads();
  // Positive definite with every eigenvalue far above eps (the usual case): the pseudo-inverse IS the inverse, one wave
  // forms it by Cholesky in a few microseconds; otherwise the eigen-decomposition decides which directions survive.
  const double eps = 1e-8;
  __shared__ int spd_fast;
  if (tid < 64) {
    const bool fast = m15 <= 15 && !force_eig && spd_inverse_wave(Am, Ainv, m15, tid, 1e3 * eps);
    if (tid == 0) spd_fast = fast ? 1 : 0;
  }
  __syncthreads();
  int sw1 = 0;
  if (!spd_fast) {
    sw1 = jacobi_small(Am, Vm, m15, tid, MARG_THREADS, Tm, cs, scr);  // Tm (n x m15) is free until the Schur step
    __syncthreads();
    for (int e = tid; e < m15 * m15; e += MARG_THREADS) {
      const int r = e / m15, c = e % m15;
      double s = 0;
      for (int k = 0; k < m15; k++) {
        const double ev = Am[k * m15 + k];
        if (ev > eps) s += Vm[r * m15 + k] * (1.0 / ev) * Vm[c * m15 + k];
      }
      Ainv[e] = s;
    }
    __syncthreads();
  }
  STAMP(S, 12);
  // ---- A' = Arr - Arm Amm^+ Amr, b' = brr - Arm Amm^+ bmm  (:275-281)
  for (int e = tid; e < n * m15; e += MARG_THREADS) {
    const int r = e / m15, c = e % m15;
    double s = 0;
    for (int k = 0; k < m15; k++) s = fma(A[(m15 + r) * D + k], Ainv[k * m15 + c], s);
    Tm[e] = s;
  }
  __syncthreads();
  {  // 4 x 4 outputs per thread (19 x 19 blocks): a quarter of the LDS reads of one output per thread
    const int nb4 = (n + 3) >> 2;
    for (int blk = tid; blk < nb4 * nb4; blk += MARG_THREADS) {
      const int r0 = 4 * (blk / nb4), c0 = 4 * (blk % nb4);
      double acc[4][4];
#pragma unroll
      for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) acc[i][j] = 0.0;
      for (int k = 0; k < m15; k++) {
        double tv[4], av[4];
#pragma unroll
        for (int i = 0; i < 4; i++) tv[i] = Tm[min(r0 + i, n - 1) * m15 + k], av[i] = A[k * D + m15 + min(c0 + i, n - 1)];
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
          for (int j = 0; j < 4; j++) acc[i][j] = fma(tv[i], av[j], acc[i][j]);
      }
#pragma unroll
      for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++)
          if (r0 + i < n && c0 + j < n) Ar[(r0 + i) * LDN + c0 + j] = A[(m15 + r0 + i) * D + m15 + c0 + j] - acc[i][j];
    }
  }
  for (int r = tid; r < n; r += MARG_THREADS) {
    double s = 0;
    for (int k = 0; k < m15; k++) s = fma(Tm[r * m15 + k], bv[k], s);
    br[r] = bv[m15 + r] - s;
  }
  __syncthreads();
  // keep A', b' for parity checks (global scratch after the gathered system)
  double *Aout = Ag + 92 * 92 + 96;
  for (int e = tid; e < n * n; e += MARG_THREADS) Aout[e] = Ar[(e / n) * LDN + e % n];
  for (int r = tid; r < n; r += MARG_THREADS) Aout[n * n + r] = br[r];
  if (poller) spec_poll_take(sp);
  __syncthreads();
  if (worker && spec_flag) return;
  // ---- second eigen-decomposition -> J0 = sqrt(S) V^T, r0 = sqrt(1/S) V^T b'  (:283-291)
  // The matrix is strongly graded (eigenvalues 1e-6 .. 1e6): cyclic Jacobi converges markedly faster when the
  // diagonal is sorted in decreasing order first (de Rijk), which is a permutation similarity.
  STAMP(S, 13);
  if (eig_tridiag<WORKER>(Ar, br, n, tid, RV, DMs, vec8, scr, (double *)cs, S->eig_aux, out, 1e-8, S->dbg, &sp)) return;
  if (tid == 0) S->dbg[24] = sw1;
  STAMP(S, 14);
  // ---- getParameterBlocks + addr_shift
  const FrameState *x = &S->x[tr->cur];
  if (tid < mp->nb) {
    const int kind = mp->kind[tid], frame = mp->frame[tid];
    out->blocks[tid].kind = kind;
    out->blocks[tid].frame = mp->shifted_frame[tid];
    out->block_idx[tid] = mp->idx[tid];
    const double *xb = kind == LFVIO_BLOCK_POSE ? x->pose[frame]
                       : kind == LFVIO_BLOCK_SPEEDBIAS ? x->sb[frame]
                       : kind == LFVIO_BLOCK_EX_POSE ? x->ex
                                                     : &x->td;
    const int gs = (kind == LFVIO_BLOCK_POSE || kind == LFVIO_BLOCK_EX_POSE) ? 7 : (kind == LFVIO_BLOCK_SPEEDBIAS ? 9 : 1);
    for (int k = 0; k < 9; k++) out->block_x0[tid][k] = k < gs ? xb[k] : 0.0;
  }
  STAMP(S, 15);
  if (tid == 0) {
    out->valid = 1;
    out->m = m15 + (S->sharded ? (int)(S->xch[XOFF_C + XS_N0] + 0.5) : mp->N0);
    out->n = n;
    out->num_blocks = mp->nb;
    if (gated && !worker) S->iters_done = tr->iteration, S->tail_state = 2;
  }
  if (worker) {
    __syncthreads();
    spec_deliver(S, S0, sp.my_word, publish != 0);
    return;
  }
  if (gated && publish) {
    __syncthreads();
    publish_prior(S);
  }
}
