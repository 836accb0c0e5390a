// kernels_linw.h — window-resident linearization of a resident batch (BASELINE configs[4]: 512 independent windows).
//
// k_lin + k_sum were designed for the latency of ONE window: roles across the grid, every partial sum through HBM, a
// separate reduction kernel, every observation evaluated twice (landmark-major for the rows of W, pair-major for the Gram
// of the pose Hessian).  With hundreds of windows resident there are as many windows as compute units, and nothing of a
// window's partial sums has to leave the CU:
//
//   k_linw: ONE workgroup (4 waves) per window.
//     phase 1  every observation is evaluated ONCE.  A wave owns whole strips (<= 64 landmarks of one start frame, lane =
//              landmark); in step o all active lanes hold the observation of their landmark in frame start + o — one frame
//              pair: the per-pair tables come through the scalar cache, and the <= 64 basis rows are a complete operand of
//              the Gram SYRK (v_mfma_f64_16x16x4_f64, staged through 4 KB of LDS).  The lane keeps the landmark's sums over
//              its track in registers (a_l, b_l, the anchor-pose / extrinsic / td parts of its row of W); the pose-j part of
//              the row leaves at once.  The 14 x 14 basis Gram of the step is expanded to the 20 x 20 factor block by the
//              structure of E (three of its columns are 3-vectors, the rest unit vectors) and added into LDS accumulators
//              of the camera-side Hessian: pose-pose off-diagonal blocks have ONE writer (all strips of a start frame are
//              on one wave), everything else goes to a private copy per wave — no atomics, fixed order.
//     phase 3  the private copies are summed in wave order into the packed camera part of H_pp (2 701 entries) and g_p.
//     phase 2  the Schur SYRK sum_l c_l w_l w_l^T over all landmarks, block by block from the transposed rows (Slot::Wt)
//              through the LDS tile, accumulators in registers across the blocks: one result per window, no partials.
//   No gram_part, no per-block Schur partials, no k_sum.  The IMU and prior parts of H_pp are added where H_pp is consumed
//   (k_solve_dense<true> assembles them on load), so the packed 119 KB matrix is never written or read back.
//
// Same arithmetic per observation as k_lin (visual_basis, dev_factors.h); the sums are associated differently (a track
// is summed by one lane in frame order instead of four lanes + quad sum), so the results agree to rounding, not bit for
// bit: tests/test_linw.py holds both paths against each other and against the oracle.
#pragma once
#include "kernels_lin.h"

typedef const __attribute__((address_space(4))) double cdouble;  // uniform addresses: loads through the scalar cache

constexpr int LW_THREADS = 256;
constexpr int LW_STAGE = 32 * 17;                  // 32 basis rows x (16 + 1 pad)
constexpr int LW_WAVE = LW_STAGE + 256 + 64;       // stage | Qf 16 x 16 | Ec 20 x 3 (+4)
constexpr int LW_D = 0, LW_FX = 11 * 21, LW_XX = LW_FX + 11 * 42, LW_G = LW_XX + 28, LW_DUMMY = LW_G + 76;
constexpr int LW_PRIV = 800;                       // D 11 x 21 | FX 11 x 42 | XX 28 | G 76 | dummy
static_assert(LW_DUMMY < LW_PRIV, "private accumulator layout");
constexpr int LW_NPAIR = 55, LW_OFF = LW_NPAIR * 36;
constexpr int LW_PRIV0 = LINW_WAVES * LW_WAVE, LW_OFF0 = LW_PRIV0 + LINW_WAVES * LW_PRIV, LW_RED0 = LW_OFF0 + LW_OFF;
constexpr int LW_LDS_P1 = LW_RED0 + LINW_WAVES * 8;
constexpr int LW_LDS_P2 = LM_BLOCK * (WLD + 1) + 2 * LM_BLOCK;
constexpr int LW_HC = KC * (KC + 1) / 2 + KC + 3;  // packed camera part + its gradient, aliased over the stage areas
static_assert(LW_HC <= LW_PRIV0, "Hc fits the stage areas");
constexpr size_t LW_LDS_BYTES = (size_t)(LW_LDS_P1 > LW_LDS_P2 ? LW_LDS_P1 : LW_LDS_P2) * 8;

DEV int lw_tri(int n, int a, int b) { return a * n - (a * (a - 1)) / 2 + (b - a); }  // upper index, a <= b < n
DEV int lw_pidx(int i, int j) { return (i * (21 - i)) / 2 + (j - i - 1); }           // i < j <= 10 -> 0 .. 54
DEV m33 ldm_s(cdouble *p) {
  m33 r;
#pragma unroll
  for (int i = 0; i < 9; i++) r.a[i] = p[i];
  return r;
}
DEV d3 ld3_s(cdouble *p) { return d3{p[0], p[1], p[2]}; }
DEV int rfl(int v) { return __builtin_amdgcn_readfirstlane(v); }

// basis row (0 .. 13) that local column p (0 .. 19: Pi th_i Pj th_j tic th_ic td r) of the factor block starts at: the three
// translation blocks are [M1 | -M1 | M3]^T times the `red` rows 0 .. 2, every other column is one basis row
DEV int lw_base_row(int p) { return p < 3 ? 0 : p < 6 ? p : p < 9 ? 0 : p < 12 ? p - 3 : p < 15 ? 0 : p < 18 ? p - 6 : p - 6; }

// One expanded entry (p, c), p <= c, of a step's 20 x 20 block lands at  A + Bi i + Bj j (+ 36 pidx(i, j) for the
// pose-pose off-diagonal block) of the workgroup's LDS: see the header comment.
struct LwEntry {
  int rp, rc, p, c, A, Bi, Bj, off;
};
DEV LwEntry lw_entry(int e, int priv, int off0) {
  LwEntry d;
  const bool in = e < NG;
  int p = 0, rem = in ? e : 0;
  while (rem >= 20 - p) rem -= 20 - p, p++;
  const int c = p + rem;
  d.p = p, d.c = c, d.rp = lw_base_row(p), d.rc = lw_base_row(c);
  const int bp = p < 6 ? 0 : p < 12 ? 1 : p < 19 ? 2 : 3, bc = c < 6 ? 0 : c < 12 ? 1 : c < 19 ? 2 : 3;
  d.Bi = d.Bj = d.off = 0;
  d.A = priv + LW_DUMMY;
  if (!in) return d;
  if (bp == 0 && bc == 0) d.A = priv + LW_D + lw_tri(6, p, c), d.Bi = 21;
  else if (bp == 0 && bc == 1) d.A = off0 + p * 6 + (c - 6), d.off = 1;
  else if (bp == 0 && bc == 2) d.A = priv + LW_FX + p * 7 + (c - 12), d.Bi = 42;
  else if (bp == 0 && bc == 3) d.A = priv + LW_G + p, d.Bi = 6;
  else if (bp == 1 && bc == 1) d.A = priv + LW_D + lw_tri(6, p - 6, c - 6), d.Bj = 21;
  else if (bp == 1 && bc == 2) d.A = priv + LW_FX + (p - 6) * 7 + (c - 12), d.Bj = 42;
  else if (bp == 1 && bc == 3) d.A = priv + LW_G + (p - 6), d.Bj = 6;
  else if (bp == 2 && bc == 2) d.A = priv + LW_XX + lw_tri(7, p - 12, c - 12);
  else if (bp == 2 && bc == 3) d.A = priv + LW_G + 66 + (p - 12);
  return d;
}

// ---------------------------------------------------------------------------
// phase 1: the strips of this wave
// ---------------------------------------------------------------------------
DEV void linw_strips(Slot *S, const LinView &lv, int scaled, double *lw, double part[5]) {
  const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63;
  double *my = lw + wv * LW_WAVE;
  double(*stage)[17] = (double(*)[17]) my;
  double(*Qf)[16] = (double(*)[16])(my + LW_STAGE);
  double(*Ec)[3] = (double(*)[3])(my + LW_STAGE + 256);
  const LinwPlan *P = &S->linw;
  const int est_td = S->est_td, est_ex = S->est_ex;
  const double td = lv.x->td, tr_over_row = S->tr_over_row, half_row = S->half_row, sqrt_info = S->sqrt_info;
  // (the address built from scalars: the tables of this linearization point are read-only here and wave-uniform)
  const unsigned long long ta = (unsigned long long)(const void *)lv.tab;
  cdouble *TT = (cdouble *)(((unsigned long long)(unsigned)rfl((int)(ta >> 32)) << 32) | (unsigned)rfl((int)ta));
  constexpr int O_M1 = offsetof(Tab, M1) / 8, O_M2 = offsetof(Tab, M2) / 8, O_T = offsetof(Tab, T) / 8, O_C = offsetof(Tab, c) / 8;
  constexpr int O_RIC = offsetof(Tab, ric) / 8, O_RICT = offsetof(Tab, ricT) / 8, O_TIC = offsetof(Tab, tic) / 8;
  const double *tabv = (const double *)lv.tab;  // (per-lane reads of the same tables: the columns of E)
  // this lane's entries of the expanded block
  LwEntry en[4];
#pragma unroll
  for (int m = 0; m < 4; m++) en[m] = lw_entry(lane + 64 * m, LW_PRIV0 + wv * LW_PRIV, LW_OFF0);
  // column `lane` of E (lanes 0 .. 19): which of M1 / -M1 / M3 it is a column of, or a unit vector
  const int ekind = lane < 3 ? 1 : (lane >= 6 && lane < 9) ? 2 : (lane >= 12 && lane < 15) ? 3 : 0, eq = lane % 3;
  const double2 *wt0 = (const double2 *)(const double *)S->Wt;
  double cost_s = 0, g2_s = 0, asv2_s = 0, lam2_s = 0, bmax_s = 0;
  const int t0 = rfl(P->wave_first[wv]), t1 = rfl(P->wave_first[wv + 1]);
  for (int t = t0; t < t1; t++) {
    const int lm0 = rfl(P->lm0[t]), nlm = rfl(P->nlm[t]), s = rfl(P->start[t]), kmax = rfl(P->kmax[t]);
    const int l = lm0 + lane;
    const bool valid = lane < nlm;
    const int lc = valid ? l : lm0;
    ObsPair ob;
    ob.pi = mk3(S->anc[0][lc], S->anc[1][lc], S->anc[2][lc]);
    ob.vi = mk3(S->anc[3][lc], S->anc[4][lc], S->anc[5][lc]);
    ob.tdi = S->anc[6][lc], ob.rowi = S->anc[7][lc];
    const double lam = lv.lam[lc];
    double a = 0, b = 0, cost = 0, wtd = 0;
    d3 wPi = mk3(0, 0, 0), wTi = wPi, wTic = wPi, wTx = wPi;
    double2 *wt = (double2 *)wt0 + lc;
    for (int o = 1; o < kmax; o++) {
      const int j = s + o, pair = s * 11 + j;
      const int first = rfl(P->firstl[s][o]);
      const bool act = valid && l >= first;
      const int idx = rfl(P->pair_obs0[pair]) + (act ? l - first : 0);
      ob.pj = mk3(S->pmo[0][idx], S->pmo[1][idx], S->pmo[2][idx]);
      ob.vj = mk3(S->pmo[3][idx], S->pmo[4][idx], S->pmo[5][idx]);
      ob.tdj = S->pmo[6][idx], ob.rowj = S->pmo[7][idx];
      // the column of E this lane will put into LDS (requested here, used behind the SYRK)
      double ecol[3] = {1.0, 0.0, 0.0};
      if (ekind) {
#pragma unroll
        for (int k = 0; k < 3; k++) {
          const double m1 = tabv[O_M1 + j * 9 + 3 * k + eq];
          const double m3 = tabv[O_M2 + pair * 9 + 3 * k + eq] - tabv[O_RICT + 3 * k + eq];
          ecol[k] = ekind == 1 ? m1 : ekind == 2 ? -m1 : m3;
        }
      }
      PairU u;
      u.M2 = ldm_s(TT + O_M2 + pair * 9), u.T = ldm_s(TT + O_T + pair * 9), u.ric = ldm_s(TT + O_RIC), u.ricT = ldm_s(TT + O_RICT);
      u.c = ld3_s(TT + O_C + pair * 3), u.tic = ld3_s(TT + O_TIC);
      const m33 M1 = ldm_s(TT + O_M1 + j * 9);
      Basis B;
      visual_basis(ob, lam, td, est_td, tr_over_row, half_row, sqrt_info, u, B);
      {
        const double jl0 = B.jl[0], jl1 = B.jl[1];
        const d3 eR = jl0 * B.red[0] + jl1 * B.red[1];
        const d3 wp = vmul(eR, M1);
        const d3 wtj = jl0 * B.jtj[0] + jl1 * B.jtj[1];
        m33 M3 = u.M2;
#pragma unroll
        for (int e = 0; e < 9; e++) M3.a[e] -= u.ricT.a[e];
        if (act) {
          wPi = wPi + wp;
          wTi = wTi + (jl0 * B.jti[0] + jl1 * B.jti[1]);
          wTic = wTic + vmul(eR, M3);
          wTx = wTx + (jl0 * B.jtx[0] + jl1 * B.jtx[1]);
          wtd += jl0 * B.jtd[0] + jl1 * B.jtd[1];
          a += jl0 * jl0 + jl1 * jl1;
          b += jl0 * B.r[0] + jl1 * B.r[1];
          cost += 0.5 * B.rho0;
          // the pose-j part of the landmark's row: columns 6 j .. 6 j + 5 = pairs 3 j .. 3 j + 2 of the transposed copy
          wt[(size_t)(3 * j) * SPEC_MAX_LM] = make_double2(-wp.x, -wp.y);
          wt[(size_t)(3 * j + 1) * SPEC_MAX_LM] = make_double2(-wp.z, wtj.x);
          wt[(size_t)(3 * j + 2) * SPEC_MAX_LM] = make_double2(wtj.y, wtj.z);
        }
      }
      // ---- Gram of the step: sum over its observations of the two 14-wide basis rows (SYRK on the matrix pipe)
      double c0[14], c1[14];
      c0[0] = B.red[0].x, c0[1] = B.red[0].y, c0[2] = B.red[0].z, c1[0] = B.red[1].x, c1[1] = B.red[1].y, c1[2] = B.red[1].z;
      c0[3] = B.jti[0].x, c0[4] = B.jti[0].y, c0[5] = B.jti[0].z, c1[3] = B.jti[1].x, c1[4] = B.jti[1].y, c1[5] = B.jti[1].z;
      c0[6] = B.jtj[0].x, c0[7] = B.jtj[0].y, c0[8] = B.jtj[0].z, c1[6] = B.jtj[1].x, c1[7] = B.jtj[1].y, c1[8] = B.jtj[1].z;
      c0[9] = B.jtx[0].x, c0[10] = B.jtx[0].y, c0[11] = B.jtx[0].z, c1[9] = B.jtx[1].x, c1[10] = B.jtx[1].y, c1[11] = B.jtx[1].z;
      c0[12] = B.jtd[0], c1[12] = B.jtd[1], c0[13] = B.r[0], c1[13] = B.r[1];
#pragma unroll
      for (int e = 0; e < 14; e++) c0[e] = act ? c0[e] : 0.0, c1[e] = act ? c1[e] : 0.0;
      double4_t acc = double4_t{0, 0, 0, 0};
      const int g_lo = (first > lm0 ? first - lm0 : 0) >> 4, g_hi = (nlm - 1) >> 4;  // lane groups of 16 that hold active lanes
      for (int r = g_lo; r <= g_hi; r++) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        if ((lane >> 4) == r) {
          const int row = 2 * (lane & 15);
#pragma unroll
          for (int e = 0; e < 14; e++) stage[row][e] = c0[e], stage[row + 1][e] = c1[e];
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
#pragma unroll
        for (int g = 0; g < 8; g++) {
          const double v = stage[4 * g + (lane >> 4)][lane & 15];
          acc = __builtin_amdgcn_mfma_f64_16x16x4f64(v, v, acc, 0, 0, 0);
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
#pragma unroll
      for (int r = 0; r < 4; r++) Qf[(lane >> 4) + 4 * r][lane & 15] = acc[r];
      if (lane < 20) Ec[lane][0] = ecol[0], Ec[lane][1] = ecol[1], Ec[lane][2] = ecol[2];
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      // ---- E^T Q E by the structure of E, this lane's (up to) four entries, added to the accumulators
      const int offp = 36 * lw_pidx(s, j);
#pragma unroll
      for (int m = 0; m < 4; m++) {
        const LwEntry &d = en[m];
        double tl[3];
#pragma unroll
        for (int q = 0; q < 3; q++) {
          tl[q] = Ec[d.p][0] * Qf[d.rp][d.rc + q];
          tl[q] = fma(Ec[d.p][1], Qf[d.rp + 1][d.rc + q], tl[q]);
          tl[q] = fma(Ec[d.p][2], Qf[d.rp + 2][d.rc + q], tl[q]);
        }
        const double val = fma(Ec[d.c][2], tl[2], fma(Ec[d.c][1], tl[1], Ec[d.c][0] * tl[0]));
        double *dst = lw + (d.A + d.Bi * s + d.Bj * j + (d.off ? offp : 0));
        *dst += val;
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    }
    // ---- the landmark's own sums: anchor pose, extrinsic, td parts of its row; scalars of the trust region
    if (valid) {
      wt[(size_t)(3 * s) * SPEC_MAX_LM] = make_double2(wPi.x, wPi.y);
      wt[(size_t)(3 * s + 1) * SPEC_MAX_LM] = make_double2(wPi.z, wTi.x);
      wt[(size_t)(3 * s + 2) * SPEC_MAX_LM] = make_double2(wTi.y, wTi.z);
      if (est_ex) {
        wt[(size_t)33 * SPEC_MAX_LM] = make_double2(wTic.x, wTic.y);
        wt[(size_t)34 * SPEC_MAX_LM] = make_double2(wTic.z, wTx.x);
        wt[(size_t)35 * SPEC_MAX_LM] = make_double2(wTx.y, wTx.z);
      }
      wt[(size_t)36 * SPEC_MAX_LM] = make_double2(est_td ? wtd : 0.0, 0.0);
      double sc;
      if (!scaled) {
        sc = 1.0 / (1.0 + sqrt(a));  // jacobi_scaling, fixed at iteration 0
        S->scale_l[l] = sc;
      } else {
        sc = S->scale_l[l];
      }
      const double s2a = sc * sc * a;
      const double D2 = fmin(fmax(s2a, 1e-6), 1e32);  // min/max_lm_diagonal
      const double dg = sqrt(D2);
      const double gr = sc * b / dg;  // DoglegStrategy::ComputeGradient
      S->diag_l[l] = dg;
      S->grad_l[l] = gr;
      const double v = gr / dg;
      const double eb = s2a + lv.mu * D2;
      S->einv_l[l] = 1.0 / eb;
      S->a[l] = a;
      S->b[l] = b;
      cost_s += cost, g2_s += gr * gr, asv2_s += s2a * v * v, lam2_s += lam * lam, bmax_s = fmax(bmax_s, fabs(b));
    }
  }
  part[0] = wave_sum(cost_s), part[1] = wave_sum(g2_s), part[2] = wave_sum(asv2_s), part[3] = wave_sum(lam2_s), part[4] = wave_max(bmax_s);
}

// ---------------------------------------------------------------------------
// k_linw: grid (1, batch) x 256, dynamic LDS = LW_LDS_BYTES
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(LW_THREADS, 2) void k_linw(char *base, size_t stride, long long imu_off) {
  extern __shared__ __attribute__((aligned(16))) double lw[];
  Slot *S = SLOT(base, stride);
  TRState *tr = &S->tr;
  const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63;
  const TRFlags fl = tr_flags(tr);
  const double mu = tr->mu;
  const int N = S->N, est_ex = S->est_ex, est_td = S->est_td;
  // a pass that starts with the loop still open is a pass this slot needs (k_lin's count)
  if (!fl.done && tid == 0) S->passes_used++;
  if (fl.done | (!fl.do_lin & !fl.do_schur)) return;
  LinView lv;
  lv.x = &S->x[fl.cur], lv.tab = &S->tab[fl.cur], lv.lam = S->lam[fl.cur], lv.mu = mu;
  if (fl.do_lin) {
    for (int e = tid; e < LW_LDS_P1; e += LW_THREADS) lw[e] = 0.0;
    __syncthreads();
    double part[5];
    linw_strips(S, lv, fl.scaled, lw, part);
    if (lane == 0) {
#pragma unroll
      for (int k = 0; k < 5; k++) lw[LW_RED0 + 8 * wv + k] = part[k];
    }
    __syncthreads();
    // ---- phase 3: the camera part of H_pp (visual terms only) and the whole of g_p.  Private copies in wave order.
    double hv[11];
    const bool act_all = true;
    (void)act_all;
#pragma unroll
    for (int q = 0; q < 11; q++) {
      const int e = tid + LW_THREADS * q;  // packed entry (r, c), r >= c, then the gradient
      double v = 0.0;
      if (e < SUM_VIS) {
        int r, c;
        if (e < SUM_VIS_PACKED) {
          r = (int)((sqrt(8.0 * e + 1.0) - 1.0) * 0.5);
          while ((r + 1) * (r + 2) / 2 <= e) r++;
          while (r * (r + 1) / 2 > e) r--;
          c = e - r * (r + 1) / 2;
        } else {
          r = c = e - SUM_VIS_PACKED;
        }
        const int fr = r < 66 ? r / 6 : 11, fc = c < 66 ? c / 6 : 11, lr = r < 66 ? r - 6 * fr : r - 66, lc = c < 66 ? c - 6 * fc : c - 66;
        int at, shared = 0;
        if (e >= SUM_VIS_PACKED) at = LW_G + c;
        else if (fr == fc && fr < 11) at = LW_D + 21 * fr + lw_tri(6, lc, lr);
        else if (fr < 11) at = LW_OFF0 + 36 * lw_pidx(fc, fr) + lc * 6 + lr, shared = 1;
        else if (fc < 11) at = LW_FX + 42 * fc + lc * 7 + lr;
        else at = LW_XX + lw_tri(7, lc, lr);
        if (shared) v = lw[at];
        else {
          const double *p0 = lw + LW_PRIV0 + at;
          v = (p0[0] + p0[LW_PRIV]) + (p0[2 * LW_PRIV] + p0[3 * LW_PRIV]);
        }
        const bool act_r = !((!est_ex && r >= off_ex() && r < off_ex() + 6) || (!est_td && r == off_td()));
        const bool act_c = !((!est_ex && c >= off_ex() && c < off_ex() + 6) || (!est_td && c == off_td()));
        if (!(act_r && act_c)) v = 0.0;
      }
      hv[q] = v;
    }
    __syncthreads();  // (the accumulators are read: their LDS is free)
#pragma unroll
    for (int q = 0; q < 11; q++) {
      const int e = tid + LW_THREADS * q;
      if (e < SUM_VIS_PACKED) S->Hpp[e] = hv[q];
      else if (e < SUM_VIS) lw[e - SUM_VIS_PACKED] = hv[q];  // visual gradient, camera side
    }
    __syncthreads();
    if (tid < KP) {
      // g_p entry: visual part, the (at most two) IMU factors, the prior — k_sum's order
      const int r = tid;
      double val = r < KC ? lw[r] : 0.0;
      const int f0 = col_frame(r);
      const double *imu_out = (const double *)((const char *)S + imu_off);
      if (f0 >= 0) {
#pragma unroll
        for (int u = 0; u < 2; u++) {
          const int f = f0 - 1 + u;
          if (f >= 0 && f < LFVIO_WINDOW_SIZE) {
            const int pl = imu_local(r, f);
            if (pl >= 0) val += imu_out[(size_t)f * IMU_OUT + 900 + pl];
          }
        }
      }
      val += S->prior_g[r];
      const bool act_r = !((!est_ex && r >= off_ex() && r < off_ex() + 6) || (!est_td && r == off_td()));
      S->gp[r] = act_r ? val : 0.0;
    }
    if (tid < 5) {
      const double *rd = lw + LW_RED0 + tid;
      S->lm_sum[tid] = tid < 4 ? ((rd[0] + rd[8]) + (rd[16] + rd[24])) : fmax(fmax(rd[0], rd[8]), fmax(rd[16], rd[24]));
    }
    __syncthreads();
  }
  if (!fl.do_schur) return;
  // ---- phase 2: Schur SYRK over all landmarks, the tile refilled block by block from the transposed rows
  double(*tile)[WLD + 1] = (double(*)[WLD + 1]) lw;
  double *lcoef = lw + LM_BLOCK * (WLD + 1), *le = lcoef + LM_BLOCK;
  const int kk = lane >> 4, cc = lane & 15;
  double4_t acc[4];
  int ct[4], cu[4];
  bool scale_k[4], own[4];
#pragma unroll
  for (int j = 0; j < 4; j++) {
    acc[j] = double4_t{0, 0, 0, 0};
    const int ti = wv + 4 * j;
    const int t = ti < 5 ? 0 : ti < 9 ? 1 : ti < 12 ? 2 : ti < 14 ? 3 : 4;
    const int u = ti - (t * 5 - (t * (t - 1)) / 2) + t;
    own[j] = ti < NT;
    ct[j] = own[j] ? 16 * t + cc : 0, cu[j] = own[j] ? 16 * u + cc : 0;
    scale_k[j] = cu[j] == COL_K;
  }
  const double2 *wt0 = (const double2 *)(const double *)S->Wt;
  const int nblk = (N + LM_BLOCK - 1) / LM_BLOCK;
  for (int e = tid; e < LM_BLOCK * 6; e += LW_THREADS) tile[e / 6][75 + e % 6] = 0.0;  // pad columns 75 .. 80
  for (int blk = 0; blk < nblk; blk++) {
    __syncthreads();
    {
      const int l = blk * LM_BLOCK + lane;
#pragma unroll
      for (int k = 0; k < (WT_PAIRS + 3) / 4; k++) {
        const int cp = wv + 4 * k;
        if (cp < WT_PAIRS) {
          const double2 v = l < N ? wt0[(size_t)cp * SPEC_MAX_LM + l] : make_double2(0.0, 0.0);
          tile[lane][2 * cp] = v.x;
          if (2 * cp + 1 < KC) tile[lane][2 * cp + 1] = v.y;
        }
      }
      if (tid < LM_BLOCK) {
        double cf = 0.0, eb = 0.0, bl = 0.0, kap = 0.0;
        if (l < N) {
          const double sc = S->scale_l[l], s2a = sc * sc * S->a[l];
          const double D2 = fmin(fmax(s2a, 1e-6), 1e32);
          eb = s2a + mu * D2;  // e-block + lm_diagonal^2
          const double einv = 1.0 / eb;
          cf = sc * sc * einv;
          if (!fl.do_lin) S->einv_l[l] = einv;  // (a solve repeated with a larger mu: only the weights change)
          bl = S->b[l], kap = bl / D2;
        }
        lcoef[tid] = cf, le[tid] = eb;
        tile[tid][COL_B] = bl, tile[tid][COL_K] = kap;
      }
    }
    __syncthreads();
    int rows = N - blk * LM_BLOCK;
    rows = rows > LM_BLOCK ? LM_BLOCK : rows;
    for (int s4 = 0; 4 * s4 < rows; s4++) {
      const int row = 4 * s4 + kk;
      const double coef = lcoef[row], eb = le[row];
#pragma unroll
      for (int j = 0; j < 4; j++) {
        if (wv + 4 * j >= NT) continue;  // wave-uniform
        const double xa = tile[row][ct[j]];
        double xb = tile[row][cu[j]];
        if (scale_k[j]) xb *= eb;  // b / D2 * e  -> z2 column
        acc[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(coef * xa, xb, acc[j], 0, 0, 0);
      }
    }
  }
  double *ss = S->schur_sum;
#pragma unroll
  for (int j = 0; j < 4; j++) {
    if (wv + 4 * j >= NT) continue;
#pragma unroll
    for (int r = 0; r < 4; r++) ss[(wv + 4 * j) * 256 + r * 64 + lane] = acc[j][r];
  }
}
