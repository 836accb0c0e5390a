// kernels_stepw.h — the second half of a pass of a resident batch, window-resident like the first (kernels_linw.h):
//
//   k_stepw: ONE workgroup per window = k_dogleg (landmark back-substitution inline, dogleg step, candidate state and pair
//   table) + the cost of the candidate at every factor + the trust-region bookkeeping (k_decide), three launches of the
//   role-by-role path (four with k_cost_imu) as one.  The dogleg and the bookkeeping are the bodies those kernels run
//   (dogleg_body, decide_body: kernels_solve.h, tr_decide.h); the cost evaluation is laid out for one workgroup:
//     visual factors   one LANE per landmark (the window has at most 320), the track's residuals in a loop — ten deep at most
//     IMU factors      one lane per factor (IntegrationBase::evaluate is a serial chain), on the lanes of the last wave that
//                      hold no landmark where there are such, then the whitening
//     prior            four lanes per row of J0 (waves 0 .. 3)
//   The sums land where k_decide's decide_sums() looks for them (the window's totals in block 0 of cost_part, zeros behind).
#pragma once
#include "kernels_solve.h"

constexpr int STEPW_THREADS = DOGLEG_INLINE_THREADS;  // 320: one landmark per thread
static_assert(STEPW_THREADS == 320 && SPEC_MAX_LM == 320, "one landmark per thread");

// ws: >= 5 * 8 + KP + SPEC_MAX_LM doubles of LDS.  320 threads: the visual factors on waves 0 .. 3, the IMU factors on wave 4;
// 256 threads (k_stepw as launched): the IMU factors on wave 0, the visual ones on waves 1 .. 3.
constexpr int STEPW_WS = 5 * 8 + KP + 4 + SPEC_MAX_LM;
DEV void stepw_cost(Slot *S, int cur, double cg, double cn, double *ws) {
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, nthr = blockDim.x, nwv = nthr >> 6;
  const int imu_wave = nthr > 256 ? 4 : 0, vis0 = nthr > 256 ? 0 : 64, vis_n = nthr > 256 ? 256 : 192;
  const int nxt = cur ^ 1;
  const Tab *T = &S->tab[nxt];
  const FrameState *x = &S->x[nxt];
  double *lam_out = S->lam[nxt];
  double(*red)[8] = (double(*)[8])ws;
  double *dxs = ws + 40, *lcs = ws + 40 + KP + 4;
  const int N = S->N, NV = S->NV, est_td = S->est_td;
  const int n = S->prior_valid ? S->prior_n : 0;
  if (n > 0 && tid < S->prior_nb) prior_block_dx(S, x, tid, dxs);
  // ---- the landmark part of the candidate and of the model: one lane per landmark
  double cost = 0, mlin = 0, mquad = 0, dn = 0, xn = 0;
  // (at most two trips — SPEC_MAX_LM landmarks over >= 256 threads — written as two guarded ones with every load of both in front
  // of the first store: behind a store the compiler fetches the arrays' offsets again, and every fetch is a memory round trip)
  {
    const double *scale_l = &S->scale_l[0], *grad_l = &S->grad_l[0], *gn_l = &S->gn_l[0], *diag_l = &S->diag_l[0], *lam_cur = &S->lam[cur][0];
    const double *d1a = &S->d1[0], *d2a = &S->d2[0], *ba = &S->b[0], *aa = &S->a[0];
    double in[2][9];
#pragma unroll
    for (int t = 0; t < 2; t++) {
      const int l = tid + nthr * t, lc = l < N ? l : 0;
      in[t][0] = scale_l[lc], in[t][1] = grad_l[lc], in[t][2] = gn_l[lc], in[t][3] = diag_l[lc], in[t][4] = lam_cur[lc];
      in[t][5] = d1a[lc], in[t][6] = d2a[lc], in[t][7] = ba[lc], in[t][8] = aa[lc];
    }
#pragma unroll
    for (int t = 0; t < 2; t++) {
      const int l = tid + nthr * t;
      if (l < N) {
        const double dl = (cg * in[t][1] + cn * in[t][2]) / in[t][3] * in[t][0];
        const double lc = in[t][4] + dl;
        lam_out[l] = lc, lcs[l] = lc;
        dn += dl * dl, xn += lc * lc;
        // model: -(delta.g) - 1/2 delta^T H delta, landmark rows / cols
        const double wd = cg * in[t][5] + cn * in[t][6];  // w_l . delta_c
        mlin += dl * in[t][7];
        mquad += 2.0 * dl * wd + in[t][8] * dl * dl;
      }
    }
    for (int l = tid + 2 * nthr; l < N; l += nthr) {  // (not with the launches this kernel has)
      const double dl = (cg * grad_l[l] + cn * gn_l[l]) / diag_l[l] * scale_l[l];
      const double lc = lam_cur[l] + dl;
      lam_out[l] = lc, lcs[l] = lc;
      dn += dl * dl, xn += lc * lc;
      const double wd = cg * d1a[l] + cn * d2a[l];
      mlin += dl * ba[l];
      mquad += 2.0 * dl * wd + aa[l] * dl * dl;
    }
  }
  __syncthreads();
  if (wv != imu_wave) {
    // ---- visual factors: one lane per OBSERVATION (pair-major: neighbours share the pair's table and sit on neighbouring landmarks)
    const double td = x->td, tor = S->tr_over_row, hr = S->half_row, si = S->sqrt_info;
    // VB observations of a lane at a time: their indices in one round of loads, then everything the indices lead to in a second one
    // (one observation per trip was two dependent memory round trips per trip, 7 to 8 trips a lane); summed in the same order
    constexpr int VB = 2;
    const int *pm_lm = &S->pm_lm[0];
    const unsigned char *pm_pair = &S->pm_pair[0];
    const double *anc[8], *pmo[8];
#pragma unroll
    for (int k = 0; k < 8; k++) anc[k] = &S->anc[k][0], pmo[k] = &S->pmo[k][0];
    for (int q0 = tid - vis0; q0 < NV; q0 += VB * vis_n) {
      int lq[VB], pq[VB];
#pragma unroll
      for (int b = 0; b < VB; b++) {
        const int q = q0 + b * vis_n, qc = q < NV ? q : q0;
        lq[b] = pm_lm[qc], pq[b] = pm_pair[qc];
      }
      ObsPair ob[VB];
      m33 Tq[VB];
      d3 cq[VB];
      double lamq[VB];
#pragma unroll
      for (int b = 0; b < VB; b++) {
        const int q = q0 + b * vis_n, qc = q < NV ? q : q0, l = lq[b];
        ob[b].pi = mk3(anc[0][l], anc[1][l], anc[2][l]), ob[b].vi = mk3(anc[3][l], anc[4][l], anc[5][l]);
        ob[b].tdi = anc[6][l], ob[b].rowi = anc[7][l];
        ob[b].pj = mk3(pmo[0][qc], pmo[1][qc], pmo[2][qc]), ob[b].vj = mk3(pmo[3][qc], pmo[4][qc], pmo[5][qc]);
        ob[b].tdj = pmo[6][qc], ob[b].rowj = pmo[7][qc];
        Tq[b] = ldm(T->T[pq[b]]), cq[b] = ld3(T->c[pq[b]]);
        lamq[b] = lcs[l];
      }
#pragma unroll
      for (int b = 0; b < VB; b++)
        if (q0 + b * vis_n < NV) cost += 0.5 * visual_cost(ob[b], lamq[b], td, est_td, tor, hr, si, Tq[b], cq[b]);
    }
  } else if (lane < LFVIO_WINDOW_SIZE) {
    // ---- IMU factors: one lane per factor (IntegrationBase::evaluate is a serial chain), on the fifth wave beside the visual ones
    const int f = lane;
    double c = 0.0;
    if (S->imu_active[f]) {
      double rr[15];
      imu_raw_residual(&S->imu[f], S->g, x->pose[f], x->sb[f], x->pose[f + 1], x->sb[f + 1], rr);
      const double *Sq = S->imu_sqrt[f];
#pragma unroll
      for (int r = 0; r < 15; r++) {
        double v = 0;
#pragma unroll
        for (int k = r; k < 15; k++) v = fma(Sq[r * 15 + k], rr[k], v);
        c += v * v;
      }
      c *= 0.5;
    }
    S->pose_cost[f] = c;
  }
  // ---- prior: r = r0 + J0 dx, four lanes per row (rows tid / 4 and that + 64: the reference's prior has at most 76 rows and
  //      columns; a wider one takes the plain loop)
  const double *J = S->prior_J;
  const bool fast = n <= 76, prow = tid < 256;
  double c = 0.0;
  if (n > 0 && fast && prow) {
    const int pq = tid & 3, c0 = pq * 19, r0 = tid >> 2;
    double jv[2][19], pr[2];
#pragma unroll
    for (int rr = 0; rr < 2; rr++) {
      const int row = r0 + 64 * rr;
      pr[rr] = (pq == 0 && row < n) ? S->prior_r[row] : 0.0;
#pragma unroll
      for (int k = 0; k < 19; k++) jv[rr][k] = (row < n && c0 + k < n) ? J[row * n + c0 + k] : 0.0;
    }
#pragma unroll
    for (int rr = 0; rr < 2; rr++) {
      const int row = r0 + 64 * rr;
      double sum = 0.0;
#pragma unroll
      for (int k = 0; k < 19; k++)
        if (c0 + k < n) sum = fma(jv[rr][k], dxs[c0 + k], sum);
      sum = quad_sum(sum);
      if (pq == 0 && row < n) {
        sum += pr[rr];
        c += sum * sum;
      }
    }
  } else if (n > 0 && !fast) {
    for (int row = tid; row < n; row += nthr) {
      double sum = S->prior_r[row];
      for (int cc = 0; cc < n; cc++) sum = fma(J[row * n + cc], dxs[cc], sum);
      c += sum * sum;
    }
  }
  cost = wave_sum(cost), mlin = wave_sum(mlin), mquad = wave_sum(mquad), dn = wave_sum(dn), xn = wave_sum(xn), c = wave_sum(c);
  if (lane == 0) red[wv][0] = cost, red[wv][1] = mlin, red[wv][2] = mquad, red[wv][3] = dn, red[wv][4] = xn, red[wv][5] = c;
  __syncthreads();
  if (tid < 6) {
    double v = red[0][tid];
    for (int w = 1; w < nwv; w++) v += red[w][tid];
    if (tid < 5) {
      double *cp = S->cost_part;
      cp[tid] = v;
      for (int b = 1; b < S->nLmBlocks; b++) cp[(size_t)b * LMS + tid] = 0.0;  // (decide_sums adds the blocks: the window's total sits in block 0)
    } else {
      S->pose_cost[10] = 0.5 * v;
    }
  }
  __syncthreads();
}

#ifdef LFVIO_LINW_PROFILE
#define WPST(k) do { if (blockIdx.y == 0 && threadIdx.x == 0) S->dbg[k] = (long long)__builtin_readcyclecounter(); } while (0)
#else
#define WPST(k) do { } while (0)
#endif
template <int STEPW_WT_PARTS>
DEV void stepw_body(Slot *S, double *ws) {
  __shared__ double sh2[2 * (1 + SPEC_EXTRA)];
  const int cur = tr_flags(&S->tr).cur;
  // (dogleg_body: false when the slot takes no step in this pass — finished, or a failed factorization, which the bookkeeping
  // below turns into a retry with a larger mu)
  WPST(16);
  const bool step = dogleg_body<true, true, true, STEPW_WT_PARTS>(S, 0, 1, true, 1, sh2);
  WPST(17);
  if (step) stepw_cost(S, cur, sh2[0], sh2[1], ws);
  __syncthreads();
  WPST(18);
  decide_body(S);
  WPST(19);
}
// 256 threads, not the 320 of one landmark per thread: four waves sit one per SIMD, and at two waves per SIMD (256 registers) TWO
// workgroups share a CU; five-wave workgroups did not pair up on a CU at any register count (measured: the time of a launch
// doubles from 256 to 257 windows either way), and 512 windows were two rounds.
constexpr int STEPW_LAUNCH_THREADS = 256;
__global__ __launch_bounds__(STEPW_LAUNCH_THREADS, 2) void k_stepw(char *base, size_t stride) {
  __shared__ double ws[STEPW_WS];
  stepw_body<1>(SLOT(base, stride), ws);
}
