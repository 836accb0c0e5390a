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

#ifdef LFVIO_LINW_PROFILE  // cycle stamps of window 0, wave 0 (tools/linw_clocks.py, a -DLFVIO_LINW_PROFILE build under variants/)
#ifndef LFVIO_LINB_GROUP
#define LFVIO_LINB_GROUP 0
#endif
#define WSTAMP(k) do { if (blockIdx.y == 0 && threadIdx.x == 0) S->dbg[k] = (long long)__builtin_readcyclecounter(); } while (0)
#define WACC(k, t0) do { wacc[(k) - 24] += (long long)__builtin_readcyclecounter() - (t0); } while (0)
#define WNOW() ((long long)__builtin_readcyclecounter())
#else
#define WSTAMP(k) do { } while (0)
#define WACC(k, t0) do { } while (0)
#define WNOW() 0ll
#endif

constexpr int LW_THREADS = 256;
constexpr int LW_STAGE = 32 * 17;                  // 32 basis rows x (16 + 1 pad)
constexpr int LW_WAVE = LW_STAGE + 256 + 64;       // stage | Qf 16 x 16 (rows 0 .. 2 used) | spare
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
// k_linb (a large single window: the four strips of a workgroup share their start frame s): the pose-pose off-diagonal blocks are
// the ten (s, j) ones, private per wave like everything else
constexpr int LB_OFF = 10 * 36, LB_PRIVSZ = LW_PRIV + LB_OFF, LB_RED0 = LW_PRIV0 + LINW_WAVES * LB_PRIVSZ;
static_assert(LB_RED0 + LINW_WAVES * 8 <= LW_LDS_P1, "k_linb's accumulators fit k_linw's workspace");

__host__ __device__ inline int lw_tri(int n, int a, int b) { return a * n - (a * (a - 1)) / 2 + (b - a); }  // upper index, a <= b < n
__host__ __device__ inline int lw_pidx(int i, int j) { return (i * (21 - i)) / 2 + (j - i - 1); }           // i < j <= 10 -> 0 .. 54
DEV m33 ldm_s(cdouble *p) {
  m33 r;
#pragma unroll
  for (int i = 0; i < 9; i++) r.a[i] = p[i];
  return r;
}
DEV d3 ld3_s(cdouble *p) { return d3{p[0], p[1], p[2]}; }

// basis row (0 .. 13) that local column p (0 .. 19: Pi th_i Pj th_j tic th_ic td r) of the factor block starts at: the three
// translation blocks are [M1 | -M1 | M3]^T times the `red` rows 0 .. 2, every other column is one basis row
DEV int lw_base_row(int p) { return p < 3 ? 0 : p < 6 ? p : p < 9 ? 0 : p < 12 ? p - 3 : p < 15 ? 0 : p < 18 ? p - 6 : p - 6; }

// Where the entry (p, c), p <= c, of a step's 20 x 20 factor block [Pi th_i Pj th_j tic th_ic td r] lands in the workgroup's
// LDS accumulators:  A + Bi i + Bj j (+ 36 pidx(i, j) for the pose-pose off-diagonal block) — see the header comment.
// One word:  A | Bi << 16 | Bj << 22 | off << 28 | negate << 29 | valid << 30.
constexpr int LWD_NEG = 1 << 29, LWD_VALID = 1 << 30;
DEV int lw_desc(int p, int c, bool neg, int priv, int off0) {
  if (p > c) {
    const int t = p;
    p = c, c = t;
  }
  const int bp = p < 6 ? 0 : p < 12 ? 1 : p < 19 ? 2 : 3, bc = c < 6 ? 0 : c < 12 ? 1 : c < 19 ? 2 : 3;
  int A = priv + LW_DUMMY, Bi = 0, Bj = 0, off = 0;
  if (bp == 0 && bc == 0) A = priv + LW_D + lw_tri(6, p, c), Bi = 21;
  else if (bp == 0 && bc == 1) A = off0 + p * 6 + (c - 6), off = 1;
  else if (bp == 0 && bc == 2) A = priv + LW_FX + p * 7 + (c - 12), Bi = 42;
  else if (bp == 0 && bc == 3) A = priv + LW_G + p, Bi = 6;
  else if (bp == 1 && bc == 1) A = priv + LW_D + lw_tri(6, p - 6, c - 6), Bj = 21;
  else if (bp == 1 && bc == 2) A = priv + LW_FX + (p - 6) * 7 + (c - 12), Bj = 42;
  else if (bp == 1 && bc == 3) A = priv + LW_G + (p - 6), Bj = 6;
  else if (bp == 2 && bc == 2) A = priv + LW_XX + lw_tri(7, p - 12, c - 12);
  else if (bp == 2 && bc == 3) A = priv + LW_G + 66 + (p - 12);
  return A | (Bi << 16) | (Bj << 22) | (off << 28) | (neg ? LWD_NEG : 0) | LWD_VALID;
}
// basis row b (0 .. 13: jP th_i th_j th_ic td r) -> the local column it is (jP -> Pi; its Pj copy is handled apart)
DEV int lw_pcol(int b) { return b < 6 ? b : b < 9 ? b + 3 : b < 12 ? b + 6 : b + 6; }

// Byte offsets, inside a slot blob, of the arrays k_linw touches (the same for every slot of a context), passed BY VALUE: a
// GP<> member of the slot would be fetched from memory before every use the compiler cannot prove unchanged — one more
// dependent round trip per step of the sweep.
struct LinwArgs {
  long long anc0, anc_stride, pmo0, pmo_stride;  // first channel and distance between the 8 channels of the two observation copies
  long long Wt, lam[2], a, b, scale_l, diag_l, grad_l, einv_l, imu_out, Hpp, gp, schur_sum;
  long long part;  // k_linb: the groups' partial sums (LINB_LEN doubles each; the role-by-role path's Schur partials live there)
  int wt_ld;       // k_linb: distance between the column pairs of Slot::Wt (landmarks, padded)
  const int *asm_tab;  // static: where every packed camera entry of H_pp is found in the LDS accumulators (build_linw_table)
};
template <class T>
DEV T *lw_at(Slot *S, long long off) { return (T *)((char *)S + off); }
DEV double lw_sel3(int q, double x0, double x1, double x2) { return q == 0 ? x0 : q == 1 ? x1 : x2; }

// ---------------------------------------------------------------------------
// phase 1: the strips of this wave
// ---------------------------------------------------------------------------
// BIG (k_linb): the wave's strips come from the caller (t0 .. t1, descriptors in the lanes' d_lm0 / d_nsk), Wt rows are A.wt_ld
// apart and every entry of a strip's span is WRITTEN (nothing zero-fills a large window's Wt), the pose-pose blocks are private.
// o1 / ostep (BIG): the wave takes the steps o1, o1 + ostep, ... of its strip — ostep waves share a strip whose tracks are long
// (the steps of a strip are a serial chain; a workgroup with one or two strips splits them) and the landmark's sums meet in LDS.
// OFFS: see k_lin (kernels_lin.h)
template <bool BIG, bool OFFS>
DEV void linw_strips(Slot *S, const LinView &lv, const LinwArgs &A, int cur, int scaled, int mode, double *lw, double part[5], int t0, int t1, int d_lm0,
                     int d_nsk, int o1 = 1, int ostep = 1) {
  const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63;
  double *my = lw + wv * LW_WAVE;
  double(*stage)[17] = (double(*)[17]) my;
  double(*Qf)[16] = (double(*)[16])(my + LW_STAGE);
  const LinwPlan *P = &S->linw;
  const bool marg = is_marg(mode);
  const int marg_n0 = marg ? rfl(marg_plan(S, mode)->N0) : 0;  // landmarks the marginalization eliminates (0: MARGIN_SECOND_NEW)
  const int est_td = S->est_td, est_ex = marg ? 1 : S->est_ex;  // (ResidualBlockInfo::Evaluate asks for every Jacobian)
  const double td = lv.x->td, tr_over_row = S->tr_over_row, half_row = S->half_row, sqrt_info = S->sqrt_info;
  // (the address built from scalars: the tables of this linearization point are read-only here and wave-uniform)
  cdouble *TT = uniform_cptr(lv.tab);
  constexpr int O_M1 = offsetof(Tab, M1) / 8, O_M2 = offsetof(Tab, M2) / 8, O_T = offsetof(Tab, T) / 8, O_C = offsetof(Tab, c) / 8;
  constexpr int O_RIC = offsetof(Tab, ric) / 8, O_RICT = offsetof(Tab, ricT) / 8, O_TIC = offsetof(Tab, tic) / 8;
  const unsigned offm = OFFS ? (unsigned)TT[O_C] : 0u;  // quaternions of this point off the unit sphere (struct Tab)
  // What this lane does with the step's basis Gram Q (its accumulator registers hold Q[kq + 4 r][ii]).  The basis rows are
  // [jP th_i th_j th_ic td r] with jP = M1^T red, so every column of the factor block but `tic` is a basis row (Pj = -jP):
  //   prim[r]   where Q[kq + 4 r][ii] itself goes (upper triangle, < 14);
  //   extra[.]  for r = 0 and a jP row: the Pj copies (negated against other columns; (Pj, Pj) and the two (Pi, Pj) entries when
  //             the column is a jP column too);
  //   the tic columns are Kt jP, Kt = M3^T M1 (M1 is orthogonal): lane e < 57 forms ONE entry (tic_a, u) = sum_k Kt[a][k] Q[k][u]
  //             — or (tic_a, tic_b) — from rows 0 .. 2 of Q, which go through LDS for that.
  const int kq = lane >> 4, ii = lane & 15, privb = LW_PRIV0 + wv * (BIG ? LB_PRIVSZ : LW_PRIV), off0 = BIG ? privb + LW_PRIV : LW_OFF0;
  const size_t wld = BIG ? (size_t)A.wt_ld : (size_t)SPEC_MAX_LM;
  const int nodesc = privb + LW_DUMMY;  // (no valid bit: the slot adds zero to the wave's dummy word)
  int prim[4], extra[3] = {nodesc, nodesc, nodesc};
#pragma unroll
  for (int r = 0; r < 4; r++) {
    const int R = kq + 4 * r;
    prim[r] = (R <= ii && ii < 14) ? lw_desc(lw_pcol(R), lw_pcol(ii), false, privb, off0) : nodesc;
  }
  if (kq < 3 && kq <= ii && ii < 14) {
    if (ii >= 3) extra[0] = lw_desc(6 + kq, lw_pcol(ii), true, privb, off0);
    else {
      extra[0] = lw_desc(6 + kq, 6 + ii, false, privb, off0);
      extra[1] = lw_desc(kq, 6 + ii, true, privb, off0);
      if (kq != ii) extra[2] = lw_desc(ii, 6 + kq, true, privb, off0);
    }
  }
  // the tic entry of this lane: column a of Kt twice (tic-tic) or once; source basis column bu, sign
  int tic_desc = nodesc, tic_a = 0, tic_b = 0, tic_u = 0;  // tic_b >= 0: (tic_a, tic_b); else (tic_a, column tic_u of Q)
  bool tic_neg = false;
  if (lane < 51) {
    tic_a = lane / 17;
    const int ui = lane % 17;  // Pi 0-2 | th_i 3-5 | Pj 6-8 | th_j 9-11 | th_ic 12-14 | td 15 | r 16
    const int lu = ui < 12 ? ui : ui < 15 ? ui + 3 : ui + 3;  // local column: 0 .. 11, 15 .. 17, 18, 19
    tic_u = ui < 6 ? ui : ui < 9 ? ui - 6 : ui < 12 ? ui - 3 : ui < 15 ? ui - 3 : ui - 3;  // basis row: Pj -> jP rows 0 .. 2
    tic_neg = ui >= 6 && ui < 9;
    tic_b = -1;
    tic_desc = lw_desc(12 + tic_a, lu, false, privb, off0);
  } else if (lane < 57) {
    const int e = lane - 51;  // (0,0) (0,1) (0,2) (1,1) (1,2) (2,2)
    tic_a = e < 3 ? 0 : e < 5 ? 1 : 2, tic_b = e < 3 ? e : e < 5 ? e - 2 : 2;
    tic_desc = lw_desc(12 + tic_a, 12 + tic_b, false, privb, off0);
  }
  const double *anc = lw_at<const double>(S, A.anc0), *pmo = lw_at<const double>(S, A.pmo0);
  const size_t ancs = (size_t)A.anc_stride / 8, pmos = (size_t)A.pmo_stride / 8;
  double2 *wt0 = lw_at<double2>(S, A.Wt);
  const double *lamv = lw_at<const double>(S, A.lam[cur]);
  double cost_s = 0, g2_s = 0, asv2_s = 0, lam2_s = 0, bmax_s = 0;
#ifdef LFVIO_LINW_PROFILE
  long long wacc[5] = {0, 0, 0, 0, 0};
#endif
  // (the strip descriptors, one per lane: a strip reads its own by v_readlane)
  for (int t = t0; t < t1; t++) {
    const long long tp0 = WNOW();
    (void)tp0;
    const int lm0 = __builtin_amdgcn_readlane(d_lm0, t), ds = __builtin_amdgcn_readlane(d_nsk, t);
    const int nlm = ds & 0xff, s = (ds >> 8) & 0xff, kmax = ds >> 16;
    // the marginalization's sweep: the landmarks anchored at frame 0 — and none at all where the plan has no visual part
    // (MARGIN_SECOND_NEW drops a pose no landmark is anchored at: estimator.cpp:942-953 adds the prior factor only)
    if (marg && (s != 0 || marg_n0 == 0)) continue;
    // per step o (lane o holds it): the first landmark of this start frame that has an observation o, and the pair-major index of its observation
    const int oc = lane < 12 && s + lane < LFVIO_NUM_FRAMES ? lane : 0;
    const int r_first = P->firstl[s][oc], r_idx0 = P->pair_obs0[s * 11 + s + oc];
    const int l = lm0 + lane;
    const bool valid = lane < nlm;
    const int lc = valid ? l : lm0;
    ObsPair ob;
    ob.pi = mk3(anc[lc], anc[ancs + lc], anc[2 * ancs + lc]);
    ob.vi = mk3(anc[3 * ancs + lc], anc[4 * ancs + lc], anc[5 * ancs + lc]);
    ob.tdi = anc[6 * ancs + lc], ob.rowi = anc[7 * ancs + lc];
    const double lam = lamv[lc];
    double a = 0, b = 0, cost = 0, wtd = 0;
    d3 wPi = mk3(0, 0, 0), wTi = wPi, wTic = wPi, wTx = wPi;
    double2 *wt = wt0 + lc;
    // the observation of step 1 (every later one is requested a step ahead)
    double nx[8];
    int first = __builtin_amdgcn_readlane(r_first, BIG ? min(o1, 11) : 1);
    {
      const int idx = __builtin_amdgcn_readlane(r_idx0, BIG ? min(o1, 11) : 1) + (valid && l >= first ? l - first : 0);
#pragma unroll
      for (int k = 0; k < 8; k++) nx[k] = kmax > (BIG ? o1 : 1) ? pmo[k * pmos + idx] : 0.0;
    }
    WACC(27, tp0);
    for (int o = BIG ? o1 : 1; o < kmax; o += BIG ? ostep : 1) {
      const long long ts0 = WNOW();
      (void)ts0;
      const int j = s + o, pair = s * 11 + j;
      const bool act = valid && l >= first;
      const int first_now = first;
      ob.pj = mk3(nx[0], nx[1], nx[2]), ob.vj = mk3(nx[3], nx[4], nx[5]), ob.tdj = nx[6], ob.rowj = nx[7];
      const int on = o + (BIG ? ostep : 1);
      if (on < kmax) {
        first = __builtin_amdgcn_readlane(r_first, on);
        const int idx = __builtin_amdgcn_readlane(r_idx0, on) + (valid && l >= first ? l - first : 0);
#pragma unroll
        for (int k = 0; k < 8; k++) nx[k] = pmo[k * pmos + idx];
      }
      PairU u;
      u.M2 = ldm_s(TT + O_M2 + pair * 9), u.T = ldm_s(TT + O_T + pair * 9), u.ric = ldm_s(TT + O_RIC), u.ricT = ldm_s(TT + O_RICT);
      u.c = ld3_s(TT + O_C + pair * 3), u.tic = ld3_s(TT + O_TIC);
      u.offc = u.offr = nullptr;
      if (OFFS && offm && pair_is_off(offm, j)) u.offc = (const double *)TT + O_C + tab_cj(pair) * 3, u.offr = (const double *)TT + O_T;  // (wave-uniform)
      const m33 M1 = ldm_s(TT + O_M1 + j * 9);
      m33 M3 = u.M2;
#pragma unroll
      for (int e = 0; e < 9; e++) M3.a[e] -= u.ricT.a[e];
      // A lane without an observation in this step (a shorter track, a lane past the strip) runs on clamped, finite inputs
      // with a zero information factor: every basis entry, the residual and log(1 + s) are exact zeros — nothing to mask
      // in the sums or in the SYRK operand.
      Basis B;
      visual_basis<OFFS>(ob, lam, td, est_td, tr_over_row, half_row, act ? sqrt_info : 0.0, u, B);
      {
        const double jl0 = B.jl[0], jl1 = B.jl[1];
        const d3 eR = jl0 * B.red[0] + jl1 * B.red[1];
        const d3 wp = vmul(eR, M1);
        const d3 wtj = jl0 * B.jtj[0] + jl1 * B.jtj[1];
        wPi = wPi + wp;
        wTi = wTi + (jl0 * B.jti[0] + jl1 * B.jti[1]);
        wTic = wTic + vmul(eR, M3);
        wTx = wTx + (jl0 * B.jtx[0] + jl1 * B.jtx[1]);
        wtd += jl0 * B.jtd[0] + jl1 * B.jtd[1];
        a += jl0 * jl0 + jl1 * jl1;
        b += jl0 * B.r[0] + jl1 * B.r[1];
        cost += 0.5 * B.rho0;
        if (BIG ? valid : act) {
          // the pose-j part of the landmark's row: columns 6 j .. 6 j + 5 = pairs 3 j .. 3 j + 2 of the transposed copy
          wt[(size_t)(3 * j) * wld] = make_double2(-wp.x, -wp.y);
          wt[(size_t)(3 * j + 1) * wld] = make_double2(-wp.z, wtj.x);
          wt[(size_t)(3 * j + 2) * wld] = make_double2(wtj.y, wtj.z);
        }
      }
      WACC(24, ts0);
      const long long ts1 = WNOW();
      (void)ts1;
      // ---- Gram of the step: sum over its observations of the two 14-wide basis rows [jP th_i th_j th_ic td r] (SYRK on
      //      the matrix pipe); the active lanes — a suffix of the strip — are packed from row 0, sixteen observations a round
      const d3 jP0 = vmul(B.red[0], M1), jP1 = vmul(B.red[1], M1);
      double4_t acc = double4_t{0, 0, 0, 0};
      const int a0 = first_now > lm0 ? first_now - lm0 : 0, rounds = (nlm - a0 + 15) >> 4;
      const int pk = (lane - a0) & 63;  // packed position: the active lanes first, then the lanes past the strip, then the shorter tracks — all of those carry zero rows
      for (int r = 0; r < rounds; r++) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        if ((pk >> 4) == r) {
          double *r0 = stage[2 * (pk & 15)], *r1 = r0 + 17;
          r0[0] = jP0.x, r0[1] = jP0.y, r0[2] = jP0.z, r1[0] = jP1.x, r1[1] = jP1.y, r1[2] = jP1.z;
          r0[3] = B.jti[0].x, r0[4] = B.jti[0].y, r0[5] = B.jti[0].z, r1[3] = B.jti[1].x, r1[4] = B.jti[1].y, r1[5] = B.jti[1].z;
          r0[6] = B.jtj[0].x, r0[7] = B.jtj[0].y, r0[8] = B.jtj[0].z, r1[6] = B.jtj[1].x, r1[7] = B.jtj[1].y, r1[8] = B.jtj[1].z;
          r0[9] = B.jtx[0].x, r0[10] = B.jtx[0].y, r0[11] = B.jtx[0].z, r1[9] = B.jtx[1].x, r1[10] = B.jtx[1].y, r1[11] = B.jtx[1].z;
          r0[12] = B.jtd[0], r1[12] = B.jtd[1], r0[13] = B.r[0], r1[13] = B.r[1];
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
#pragma unroll
        for (int g = 0; g < 8; g++) {
          const double v = stage[4 * g + (lane >> 4)][lane & 15];
          acc = __builtin_amdgcn_mfma_f64_16x16x4f64(v, v, acc, 0, 0, 0);
        }
      }
      WACC(25, ts1);
      const long long ts2 = WNOW();
      (void)ts2;
      // ---- the step's 20 x 20 block into the accumulators.  Rows 0 .. 2 of Q through LDS for the tic entries; everything else
      //      straight from the accumulator registers.
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      if (kq < 3) Qf[kq][ii] = acc[0];
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      const int offp = BIG ? 36 * (o - 1) : 36 * lw_pidx(s, j);
      auto at = [&](int d) { return lw + ((d & 0xffff) + ((d >> 16) & 63) * s + ((d >> 22) & 63) * j + ((d & (1 << 28)) ? offp : 0)); };
      double tval = 0.0;
      if (tic_desc & LWD_VALID) {
        // Kt = M3^T M1: Kt[a][k] = sum_m M3[m][a] M1[m][k] (uniform; this lane needs row a — and row b for a tic-tic entry)
        double ka[3], kb[3];
#pragma unroll
        for (int k = 0; k < 3; k++) {
          double ra[3];
#pragma unroll
          for (int a2 = 0; a2 < 3; a2++) ra[a2] = fma(M3.a[6 + a2], M1.a[6 + k], fma(M3.a[3 + a2], M1.a[3 + k], M3.a[a2] * M1.a[k]));
          ka[k] = lw_sel3(tic_a, ra[0], ra[1], ra[2]);
          kb[k] = lw_sel3(tic_b < 0 ? 0 : tic_b, ra[0], ra[1], ra[2]);
        }
        if (tic_b < 0) {
          tval = fma(ka[2], Qf[2][tic_u], fma(ka[1], Qf[1][tic_u], ka[0] * Qf[0][tic_u]));
          tval = tic_neg ? -tval : tval;
        } else {
          double tl[3];
#pragma unroll
          for (int q = 0; q < 3; q++) tl[q] = fma(ka[2], Qf[2][q], fma(ka[1], Qf[1][q], ka[0] * Qf[0][q]));
          tval = fma(kb[2], tl[2], fma(kb[1], tl[1], kb[0] * tl[0]));
        }
      }
      // (all eight read first, then added and stored: a store between two of them would hold the next read back.  An
      // unused slot points at the wave's dummy word and adds zero.)
      double *dst[8], old[8], add[8];
#pragma unroll
      for (int r = 0; r < 4; r++) dst[r] = at(prim[r]), add[r] = (prim[r] & LWD_VALID) ? acc[r] : 0.0;
#pragma unroll
      for (int e = 0; e < 3; e++) dst[4 + e] = at(extra[e]), add[4 + e] = !(extra[e] & LWD_VALID) ? 0.0 : (extra[e] & LWD_NEG) ? -acc[0] : acc[0];
      dst[7] = at(tic_desc), add[7] = (tic_desc & LWD_VALID) ? tval : 0.0;
#pragma unroll
      for (int e = 0; e < 8; e++) old[e] = *dst[e];
#pragma unroll
      for (int e = 0; e < 8; e++) *dst[e] = old[e] + add[e];
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      WACC(26, ts2);
    }
    const long long te0 = WNOW();
    (void)te0;
    bool owner = true;
    if (BIG && ostep > 1) {
      // the waves that shared the strip's steps: their sums over the track meet in the first one, in wave order (two rounds of
      // eight values through the stage areas, which are free now; workgroup-uniform: every wave of such a group has a strip)
      owner = (wv % ostep) == 0;
      double v[16] = {a, b, cost, wtd, wPi.x, wPi.y, wPi.z, wTi.x, wTi.y, wTi.z, wTic.x, wTic.y, wTic.z, wTx.x, wTx.y, wTx.z};
#pragma unroll
      for (int h = 0; h < 2; h++) {
        __syncthreads();
        if (!owner) {
#pragma unroll
          for (int k = 0; k < 8; k++) my[k * 64 + lane] = v[8 * h + k];
        }
        __syncthreads();
        if (owner) {
          for (int pw = 1; pw < ostep; pw++) {
            const double *src = lw + (wv + pw) * LW_WAVE;
#pragma unroll
            for (int k = 0; k < 8; k++) v[8 * h + k] += src[k * 64 + lane];
          }
        }
      }
      a = v[0], b = v[1], cost = v[2], wtd = v[3];
      wPi = mk3(v[4], v[5], v[6]), wTi = mk3(v[7], v[8], v[9]), wTic = mk3(v[10], v[11], v[12]), wTx = mk3(v[13], v[14], v[15]);
    }
    // ---- the landmark's own sums: anchor pose, extrinsic, td parts of its row; scalars of the trust region
    if (valid && owner) {
      wt[(size_t)(3 * s) * wld] = make_double2(wPi.x, wPi.y);
      wt[(size_t)(3 * s + 1) * wld] = make_double2(wPi.z, wTi.x);
      wt[(size_t)(3 * s + 2) * wld] = make_double2(wTi.y, wTi.z);
      if (est_ex) {
        wt[(size_t)33 * wld] = make_double2(wTic.x, wTic.y);
        wt[(size_t)34 * wld] = make_double2(wTic.z, wTx.x);
        wt[(size_t)35 * wld] = make_double2(wTx.y, wTx.z);
      } else if (BIG) {
        wt[(size_t)33 * wld] = wt[(size_t)34 * wld] = wt[(size_t)35 * wld] = make_double2(0.0, 0.0);
      }
      wt[(size_t)36 * wld] = make_double2(est_td ? wtd : 0.0, 0.0);
      if (marg) {
        // eps of marginalization_factor.h:70 on the diagonal block; no scaling, no trust-region scalars
        lw_at<double>(S, A.einv_l)[l] = (a > 1e-8) ? 1.0 / a : 0.0;
        lw_at<double>(S, A.a)[l] = a;
        lw_at<double>(S, A.b)[l] = b;
        cost_s += cost, lam2_s += lam * lam, bmax_s = fmax(bmax_s, fabs(b));
      } else {
        double *scale_l = lw_at<double>(S, A.scale_l);
        double sc;
        if (!scaled) {
          sc = 1.0 / (1.0 + sqrt(a));  // jacobi_scaling, fixed at iteration 0
          scale_l[l] = sc;
        } else {
          sc = scale_l[l];
        }
        const double s2a = sc * sc * a;
        const double D2 = fmin(fmax(s2a, 1e-6), 1e32);  // min/max_lm_diagonal
        const double dg = sqrt(D2);
        const double gr = sc * b / dg;  // DoglegStrategy::ComputeGradient
        lw_at<double>(S, A.diag_l)[l] = dg;
        lw_at<double>(S, A.grad_l)[l] = gr;
        const double v = gr / dg;
        const double eb = s2a + lv.mu * D2;
        lw_at<double>(S, A.einv_l)[l] = 1.0 / eb;
        lw_at<double>(S, A.a)[l] = a;
        lw_at<double>(S, A.b)[l] = b;
        cost_s += cost, g2_s += gr * gr, asv2_s += s2a * v * v, lam2_s += lam * lam, bmax_s = fmax(bmax_s, fabs(b));
      }
    }
    WACC(28, te0);
  }
#ifdef LFVIO_LINW_PROFILE
  if (blockIdx.y == 0 && threadIdx.x == 0 && (!BIG || blockIdx.x == LFVIO_LINB_GROUP))
    for (int k = 0; k < 5; k++) S->dbg[24 + k] = wacc[k];
#endif
  part[0] = wave_sum(cost_s), part[1] = wave_sum(g2_s), part[2] = wave_sum(asv2_s), part[3] = wave_sum(lam2_s), part[4] = wave_max(bmax_s);
}

// ---------------------------------------------------------------------------
// phase 0: the pose-side factors.  IMU factor f = wave + 4 q is evaluated by lane q < 3 of the wave (IMUFactor::Evaluate's two
// serial jobs — residual and Jacobian of integration_base.h:160-186 / imu_factor.h:88-196 — straight into LDS), then the wave
// weights it with sqrt_info and forms J^T J, J^T r, the cost (what lin_imu_role does with a workgroup per factor).  No
// workgroup barrier in here: a wave only reads what it wrote itself.
// ---------------------------------------------------------------------------
constexpr int LW_JLD = 33;                                   // 16 rows x (32 + 1 pad): columns 0 .. 29 the Jacobian, 30 the residual, row 15 zero
constexpr int LW_IMU_WAVE = 3 * 16 * LW_JLD + 16 * LW_JLD;   // Jr of the wave's three factors | Jw
static_assert(LINW_WAVES * LW_IMU_WAVE <= LW_LDS_P1, "phase 0 fits the phase-1 workspace");
DEV void linw_imu(Slot *S, const LinView &lv, double *lw, long long imu_off, int mode) {
  const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63;
  double *my = lw + wv * LW_IMU_WAVE;
  double *Jr3 = my;
  double(*Jw)[LW_JLD] = (double(*)[LW_JLD])(my + 3 * 16 * LW_JLD);
  const FrameState *x = lv.x;
  const bool pose_rank = !S->sharded || S->pose_side;
  // (the marginalization's sweep takes the factor between frames 0 and 1 only, and only if the plan says so)
  auto factor_on = [&](int f) { return S->imu_active[f] && pose_rank && (mode == MODE_SOLVE || (f == 0 && marg_plan(S, mode)->use_imu0)); };
  for (int e = lane; e < 3 * 16 * LW_JLD; e += 64) my[e] = 0.0;
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  if (lane < 3) {
    const int f = wv + 4 * lane;
    if (f < LFVIO_WINDOW_SIZE && factor_on(f)) {
      double rr[15];
      imu_raw_residual(&S->imu[f], S->g, x->pose[f], x->sb[f], x->pose[f + 1], x->sb[f + 1], rr);
      double *Jr = Jr3 + 16 * LW_JLD * lane;
#pragma unroll
      for (int k = 0; k < 15; k++) Jr[k * LW_JLD + 30] = rr[k];
      imu_raw_jacobian(&S->imu[f], S->g, x->pose[f], x->sb[f], x->pose[f + 1], x->sb[f + 1], Jr, LW_JLD);
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  WSTAMP(29);
  // Both products on the FP64 matrix pipe.  Lane (kq, ii) = (lane >> 4, lane & 15) feeds A[ii][4 s + kq] and B[4 s + kq][ii]
  // of step s and holds D[kq + 4 r][ii], r = 0 .. 3, of a 16 x 16 result tile.
  //   Jw = sqrt_info [J | r]      (16 x 16 times 16 x 32: 2 tiles x 4 steps; the whitened residual rides as column 30)
  //   G  = Jw^T Jw                (32 x 32: tiles (0,0), (0,1), (1,1) x 4 steps) = [J^T J, J^T r; . , r^T r]
  const int kq = lane >> 4, ii = lane & 15;
  double *imu_out = (double *)((char *)S + imu_off);
  // the weights of the wave's three factors in ONE round of loads (k_setup's sqrt_info: a memory round trip per factor otherwise)
  double sa3[3][4];
#pragma unroll
  for (int q = 0; q < 3; q++) {
    const int f = wv + 4 * q < LFVIO_WINDOW_SIZE ? wv + 4 * q : 0;
    const double *Sq = S->imu_sqrt[f];
#pragma unroll
    for (int s4 = 0; s4 < 4; s4++) {
      const int k = 4 * s4 + kq;
      sa3[q][s4] = (ii < 15 && k < 15) ? Sq[ii * 15 + k] : 0.0;
    }
  }
#pragma unroll
  for (int q = 0; q < 3; q++) {
    const int f = wv + 4 * q;
    if (f >= LFVIO_WINDOW_SIZE) break;
    double *out = imu_out + (size_t)f * IMU_OUT;
    if (!factor_on(f)) {
      for (int e = lane; e < IMU_OUT; e += 64) out[e] = 0.0;
      continue;
    }
    const double(*Jr)[LW_JLD] = (const double(*)[LW_JLD])(Jr3 + 16 * LW_JLD * q);
    const double(&sa)[4] = sa3[q];
    double4_t w0 = double4_t{0, 0, 0, 0}, w1 = w0;
#pragma unroll
    for (int s4 = 0; s4 < 4; s4++) {
      w0 = __builtin_amdgcn_mfma_f64_16x16x4f64(sa[s4], Jr[4 * s4 + kq][ii], w0, 0, 0, 0);
      w1 = __builtin_amdgcn_mfma_f64_16x16x4f64(sa[s4], Jr[4 * s4 + kq][16 + ii], w1, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 4; r++) Jw[kq + 4 * r][ii] = w0[r], Jw[kq + 4 * r][16 + ii] = w1[r];
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    double4_t g00 = double4_t{0, 0, 0, 0}, g01 = g00, g11 = g00;
#pragma unroll
    for (int s4 = 0; s4 < 4; s4++) {
      const double v0 = Jw[4 * s4 + kq][ii], v1 = Jw[4 * s4 + kq][16 + ii];
      g00 = __builtin_amdgcn_mfma_f64_16x16x4f64(v0, v0, g00, 0, 0, 0);
      g01 = __builtin_amdgcn_mfma_f64_16x16x4f64(v0, v1, g01, 0, 0, 0);
      g11 = __builtin_amdgcn_mfma_f64_16x16x4f64(v1, v1, g11, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const int row = kq + 4 * r, c1 = 16 + ii;  // entries G[row][ii], G[row][c1], G[16 + row][c1]
      // Of the symmetric 30 x 30 block only the entries (p, q) whose tangent columns satisfy col(p) >= col(q) are stored — the
      // ones the packed lower H_pp takes (k_solve_dense<true>'s table, the marginalization's assembly): 465 of 900.  Local
      // order is pose_f, sb_f, pose_f+1, sb_f+1; tangent order pose_f, pose_f+1, sb_f, sb_f+1.
      auto ord = [](int p) { return p < 6 ? p : p < 15 ? p + 6 : p < 21 ? p - 9 : p; };
      const int r2 = 16 + row;
      if (ord(row) >= ord(ii)) out[row * 30 + ii] = g00[r];
      if (c1 < 30) {
        if (ord(row) >= ord(c1)) out[row * 30 + c1] = g01[r];
        else out[c1 * 30 + row] = g01[r];
      } else if (c1 == 30) out[900 + row] = g01[r];  // J^T r, rows 0 .. 15
      if (r2 < 30) {
        if (c1 < 30) {
          if (ord(r2) >= ord(c1)) out[r2 * 30 + c1] = g11[r];
        } else if (c1 == 30) out[900 + r2] = g11[r];
      } else if (r2 == 30 && c1 == 30) out[930] = 0.5 * g11[r];
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  }
}

// The prior factor of a pass (MarginalizationFactor::Evaluate, marginalization_factor.cpp:333-381; lin_prior_role of
// kernels_lin.h with J0 staged in LDS: one batch of coalesced loads instead of a dependent load per term):
//   r = r0 + J0 dx,  prior_g = J0^T r (tangent columns), cost.  Whole workgroup; ends on a barrier.
constexpr int LW_PRIOR_MAXN = 88;  // J0 of up to 88 x 88 fits the workspace (the reference's is 76 x 76); larger ones take the global path
static_assert(LW_PRIOR_MAXN * LW_PRIOR_MAXN + 4 * KP <= LW_LDS_P1, "prior stage fits");
DEV void linw_prior(Slot *S, const LinView &lv, double *lw) {
  const int tid = threadIdx.x;
  const int n = S->prior_n;
  if (!S->prior_valid || (S->sharded && !S->pose_side) || n > LW_PRIOR_MAXN) {
    lin_prior_role<false>(S, lv, MODE_SOLVE, lw);  // (zeroes prior_g when there is no prior)
    __syncthreads();
    return;
  }
  double *Js = lw, *dx = lw + LW_PRIOR_MAXN * LW_PRIOR_MAXN, *r = dx + KP, *part = r + KP;  // part: [2][KP]
  double *g = S->prior_g;
  const double *J = S->prior_J;
  // (what the phases below take from memory one value at a time, requested with the first round of J0)
  const double r0_row = (tid >> 1) < n ? S->prior_r[tid >> 1] : 0.0;
  const int cmap_c = tid < n ? S->prior_cmap[tid] : 0;
  {  // J0 into LDS in rounds of eight loads per thread (a loop of load - wait - store is a memory round trip per element: 23 of them;
     // asking for it ahead of the IMU factors puts their own loads behind it in the queue: measured, no gain)
    constexpr int PJ = 8, ROUNDS = (LW_PRIOR_MAXN * LW_PRIOR_MAXN + PJ * LW_THREADS - 1) / (PJ * LW_THREADS);
    const int nn = n * n;
#pragma unroll 1
    for (int b = 0; b < ROUNDS && b * PJ * LW_THREADS < nn; b++) {
      double v[PJ];
#pragma unroll
      for (int k = 0; k < PJ; k++) {
        const int e = tid + LW_THREADS * (PJ * b + k);
        v[k] = J[e < nn ? e : 0];
      }
#pragma unroll
      for (int k = 0; k < PJ; k++) {
        const int e = tid + LW_THREADS * (PJ * b + k);
        if (e < nn) Js[e] = v[k];
      }
    }
  }
  for (int c = tid; c < KP + 4; c += LW_THREADS) g[c] = 0.0;
  if (tid < S->prior_nb) prior_block_dx(S, lv.x, tid, dx);
  __syncthreads();
  {  // r = r0 + J0 dx: two lanes per row (n <= 128), four independent partial sums each so that the LDS reads overlap
    const int row = tid >> 1, h = tid & 1;
    double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
    if (row < n) {
      const double *jr = Js + row * n;
      int c = h;
      for (; c + 6 < n; c += 8) s0 = fma(jr[c], dx[c], s0), s1 = fma(jr[c + 2], dx[c + 2], s1), s2 = fma(jr[c + 4], dx[c + 4], s2), s3 = fma(jr[c + 6], dx[c + 6], s3);
      for (; c < n; c += 2) s0 = fma(jr[c], dx[c], s0);
    }
    double sum = (s0 + s1) + (s2 + s3);
    sum += dpp_f64<0xB1>(sum);  // quad_perm [1,0,3,2]: the other half of the row
    if (row < n && h == 0) r[row] = r0_row + sum;
  }
  __syncthreads();
  {  // g = J0^T r: column c, two halves of the rows
    const int c = tid & 127, h = tid >> 7, k0 = h ? n / 2 : 0, k1 = h ? n : n / 2;
    if (c < n) {
      double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
      int k = k0;
      for (; k + 3 < k1; k += 4)
        s0 = fma(Js[k * n + c], r[k], s0), s1 = fma(Js[(k + 1) * n + c], r[k + 1], s1), s2 = fma(Js[(k + 2) * n + c], r[k + 2], s2), s3 = fma(Js[(k + 3) * n + c], r[k + 3], s3);
      for (; k < k1; k++) s0 = fma(Js[k * n + c], r[k], s0);
      part[h * KP + c] = (s0 + s1) + (s2 + s3);
    }
  }
  __syncthreads();
  static_assert(LW_PRIOR_MAXN <= LW_THREADS, "one column per thread");
  if (tid < n) g[cmap_c] = part[tid] + part[KP + tid];
  if (tid < 64) {
    double cs = 0;
    for (int row = tid; row < n; row += 64) cs += r[row] * r[row];
    cs = wave_sum(cs);
    if (tid == 0) g[KP] = 0.5 * cs;
  }
  __syncthreads();
}

// Static table of phase 3 (built once per context, lfvio_hip.hip): for the packed camera entry e of H_pp (e < SUM_VIS_PACKED)
// and the camera-side gradient entries behind them, where the value sits in the LDS accumulators —
//   bits 0..15 the offset (inside a wave's private block, or absolute for the single-writer pose-pose blocks), bit 16 "absolute",
//   bit 17 the entry touches an extrinsic column, bit 18 it touches the td column; bits 20..23 / 24..27 the frames (i, j) of a pose-pose block.
constexpr int LWT_ABS = 1 << 16, LWT_EX = 1 << 17, LWT_TD = 1 << 18;

// ---------------------------------------------------------------------------
// k_linw: grid (1, batch) x 256, dynamic LDS = LW_LDS_BYTES
// ---------------------------------------------------------------------------
template <bool OFFS>
DEV void linw_body(Slot *S, double *lw, const LinwArgs &A, int mode_bits) {
  TRState *tr = &S->tr;
  const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63;
  TRFlags fl = tr_flags(tr);
  const double mu = tr->mu;
  // mode_bits: MODE_SOLVE, or MODE_MARG + flag (| MODE_GATED behind the solve passes of the same graph) — the marginalization's
  // sweep: the landmarks anchored at frame 0, the IMU factor between frames 0 and 1, the prior; every column active; the
  // complete packed H_pp written (k_marg_solve reads it), the Schur sums over those landmarks with weights 1 / a_l.
  const int mode = mode_bits & (MODE_GATED - 1);
  const bool marg = is_marg(mode);
  const int N = marg ? marg_plan(S, mode)->N0 : S->N, est_ex = marg ? 1 : S->est_ex, est_td = marg ? 1 : S->est_td;
  if (marg) {
    if ((mode_bits & MODE_GATED) && !tail_gate(S, fl.done)) return;
    fl.do_lin = fl.do_schur = 1;
  } else {
    // a pass that starts with the loop still open is a pass this slot needs (k_lin's count)
    if (!fl.done && tid == 0) S->passes_used++;
    if (fl.done | (!fl.do_lin & !fl.do_schur)) return;
  }
  LinView lv;
  lv.x = &S->x[fl.cur], lv.tab = &S->tab[fl.cur], lv.lam = lw_at<const double>(S, A.lam[fl.cur]), lv.mu = mu;
  WSTAMP(8);
  // Phase 2 wants the first start frame of every block of 64 landmarks: two dependent loads (the array's offset, the entry) that a
  // block would wait for between its barriers.  Thread b < 5 asks for block b's here and parks it in LDS behind the strips.
  static_assert(SPEC_MAX_LM <= 5 * LM_BLOCK, "five blocks of landmarks per window");
  __shared__ int blk_start[8];
  const int my_start = tid < 5 ? S->lm_start[tid * LM_BLOCK < N ? tid * LM_BLOCK : 0] : 0;
  if (!fl.do_lin) {
    if (tid < 5) blk_start[tid] = my_start;
    __syncthreads();
  }
  if (fl.do_lin) {
    constexpr int P3_E = (SUM_VIS + LW_THREADS - 1) / LW_THREADS;
    int p3[P3_E];  // (phase 3's table entries: requested here, used a hundred microseconds later)
#pragma unroll
    for (int q = 0; q < P3_E; q++) p3[q] = A.asm_tab[tid + LW_THREADS * q < SUM_VIS ? tid + LW_THREADS * q : 0];
    linw_imu(S, lv, lw, A.imu_out, mode);
    WSTAMP(30);
    __syncthreads();
    linw_prior(S, lv, lw);
    WSTAMP(15);
    for (int e = tid; e < LW_LDS_P1; e += LW_THREADS) lw[e] = 0.0;
    __syncthreads();
    WSTAMP(9);
    double part[5];
    {
      const LinwPlan *P = &S->linw;
      const int tl = lane < LINW_MAX_STRIPS ? lane : 0;
      linw_strips<false, OFFS>(S, lv, A, fl.cur, fl.scaled, mode, lw, part, rfl(P->wave_first[wv]), rfl(P->wave_first[wv + 1]), P->lm0[tl],
                               P->nlm[tl] | (P->start[tl] << 8) | (P->kmax[tl] << 16));
    }
    WSTAMP(10);
    if (tid < 5) blk_start[tid] = my_start;
    if (lane == 0) {
#pragma unroll
      for (int k = 0; k < 5; k++) lw[LW_RED0 + 8 * wv + k] = part[k];
    }
    __syncthreads();
    WSTAMP(11);
    // ---- phase 3: the camera part of H_pp (visual terms only) and the whole of g_p.  Private copies in wave order.
    double hv[P3_E];
#pragma unroll
    for (int q = 0; q < P3_E; q++) {
      const int d = p3[q], at = d & 0xffff;
      double v;
      if (d & LWT_ABS) v = lw[at];
      else {
        const double *p0 = lw + LW_PRIV0 + at;
        v = (p0[0] + p0[LW_PRIV]) + (p0[2 * LW_PRIV] + p0[3 * LW_PRIV]);
      }
      if (((d & LWT_EX) && !est_ex) || ((d & LWT_TD) && !est_td)) v = 0.0;
      hv[q] = v;
    }
    __syncthreads();  // (the accumulators are read: their LDS is free)
    double *Hpp = lw_at<double>(S, A.Hpp);
    if (!marg) {
#pragma unroll
      for (int q = 0; q < P3_E; q++) {
        const int e = tid + LW_THREADS * q;
        if (e < SUM_VIS_PACKED) Hpp[e] = hv[q];
        else if (e < SUM_VIS) lw[e - SUM_VIS_PACKED] = hv[q];  // visual gradient, camera side
      }
      __syncthreads();
    } else {
      // the marginalization reads the complete packed matrix: visual terms (camera part, from LDS), the IMU factor, the prior
      // — k_sum's sum per entry, in its order; once per call
#pragma unroll
      for (int q = 0; q < P3_E; q++) {
        const int e = tid + LW_THREADS * q;
        if (e < SUM_VIS) lw[LW_HC + (e < SUM_VIS_PACKED ? KC + e : e - SUM_VIS_PACKED)] = hv[q];  // [gradient KC | packed camera part]
      }
      __syncthreads();
      const double *imu_out = lw_at<const double>(S, A.imu_out);
      const double *prior_A = S->prior_A;
      const int prior_ok = S->prior_valid && (!S->sharded || S->pose_side), prior_n = S->prior_n;
      // eight entries of a thread per round: their indices, then the prior's column map, then the IMU and prior terms — every round of
      // loads in front of the round's first store (one entry per trip was three dependent memory round trips per trip, 58 trips)
      constexpr int MU = 8;
      for (int e0 = tid; e0 < PACKED; e0 += MU * LW_THREADS) {
        int rr[MU], cc[MU], pri[MU], pci[MU];
#pragma unroll
        for (int u = 0; u < MU; u++) {
          const int e = e0 + u * LW_THREADS < PACKED ? e0 + u * LW_THREADS : PACKED - 1;
          int r = (int)((sqrt(8.0 * e + 1.0) - 1.0) * 0.5);
          while ((r + 1) * (r + 2) / 2 <= e) r++;
          while (r * (r + 1) / 2 > e) r--;
          rr[u] = r, cc[u] = e - r * (r + 1) / 2;
          pri[u] = prior_ok ? S->prior_inv[r] : -1, pci[u] = prior_ok ? S->prior_inv[cc[u]] : -1;
        }
        double val[MU], im[MU][2], pa[MU];
        bool imok[MU][2];
#pragma unroll
        for (int u = 0; u < MU; u++) {
          const int e = e0 + u * LW_THREADS < PACKED ? e0 + u * LW_THREADS : PACKED - 1, r = rr[u], c = cc[u];
          val[u] = r < KC ? lw[LW_HC + KC + e] : 0.0;
          const int f0 = col_frame(r);
#pragma unroll
          for (int k = 0; k < 2; k++) {
            const int f = f0 - 1 + k;
            const bool fin = f0 >= 0 && f >= 0 && f < LFVIO_WINDOW_SIZE;
            const int pl = fin ? imu_local(r, f) : -1, ql = fin ? imu_local(c, f) : -1;
            imok[u][k] = pl >= 0 && ql >= 0;
            im[u][k] = imok[u][k] ? imu_out[(size_t)f * IMU_OUT + pl * 30 + ql] : 0.0;
          }
          pa[u] = (pri[u] >= 0 && pci[u] >= 0) ? prior_A[pri[u] * prior_n + pci[u]] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < MU; u++) {
          const int e = e0 + u * LW_THREADS;
          if (e < PACKED) {
            double v = val[u];
            if (imok[u][0]) v += im[u][0];
            if (imok[u][1]) v += im[u][1];
            if (pri[u] >= 0 && pci[u] >= 0) v += pa[u];
            Hpp[e] = v;
          }
        }
      }
      if (tid < KC) lw[tid] = lw[LW_HC + tid];
      __syncthreads();
    }
    if (tid < KP) {
      // g_p entry: visual part, the (at most two) IMU factors, the prior — k_sum's order
      const int r = tid;
      double val = r < KC ? lw[r] : 0.0;
      const int f0 = col_frame(r);
      const double *imu_out = lw_at<const double>(S, A.imu_out);
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
      lw_at<double>(S, A.gp)[r] = act_r ? val : 0.0;
    }
    if (tid < 5) {
      const double *rd = lw + LW_RED0 + tid;
      S->lm_sum[tid] = tid < 4 ? ((rd[0] + rd[8]) + (rd[16] + rd[24])) : fmax(fmax(rd[0], rd[8]), fmax(rd[16], rd[24]));
    }
    __syncthreads();
  }
  WSTAMP(12);
  if (!fl.do_schur) return;
  // ---- phase 2: Schur SYRK over all landmarks, the tile refilled block by block from the transposed rows; the rows of
  //      block k + 1 are requested before the matrix pipe starts on block k
  double(*tile)[WLD + 1] = (double(*)[WLD + 1]) lw;
  double *lcoef = lw + LM_BLOCK * (WLD + 1), *le = lcoef + LM_BLOCK;
  const int kk = lane >> 4, cc = lane & 15;
  double4_t acc[4];
  int ct[4], cu[4];
  bool scale_k[4];
#pragma unroll
  for (int j = 0; j < 4; j++) {
    acc[j] = double4_t{0, 0, 0, 0};
    const int ti = wv + 4 * j;
    const int t = ti < 5 ? 0 : ti < 9 ? 1 : ti < 12 ? 2 : ti < 14 ? 3 : 4;
    const int u = ti - (t * 5 - (t * (t - 1)) / 2) + t;
    ct[j] = ti < NT ? 16 * t + cc : 0, cu[j] = ti < NT ? 16 * u + cc : 0;
    scale_k[j] = !marg && cu[j] == COL_K;
  }
  const double2 *wt0 = lw_at<const double2>(S, A.Wt);
  const double *scale_l = lw_at<const double>(S, A.scale_l), *av = lw_at<const double>(S, A.a), *bv = lw_at<const double>(S, A.b);
  double *einv_l = lw_at<double>(S, A.einv_l);
  const int nblk = (N + LM_BLOCK - 1) / LM_BLOCK;
  constexpr int WPT = (WT_PAIRS + 3) / 4;  // column pairs per thread
  double2 wreg[WPT];
  double w_sc = 0, w_a = 0, w_b = 0;
  auto request = [&](int blk) {
    const int l = blk * LM_BLOCK + lane;
    // (landmarks are sorted by start frame: no row of the block has an entry in the columns of frames before its first
    // landmark's start — those pairs are zeros without a load)
    const int pmin = 3 * rfl(blk_start[blk]);
#pragma unroll
    for (int k = 0; k < WPT; k++) {
      const int cp = wv + 4 * k;
      wreg[k] = (cp < WT_PAIRS && cp >= pmin && l < N) ? wt0[(size_t)cp * SPEC_MAX_LM + l] : make_double2(0.0, 0.0);
    }
    if (tid < LM_BLOCK && l < N) w_sc = scale_l[l], w_a = av[l], w_b = bv[l];
  };
  for (int e = tid; e < LM_BLOCK * 6; e += LW_THREADS) tile[e / 6][75 + e % 6] = 0.0;  // pad columns 75 .. 80
  if (nblk > 0) request(0);
  for (int blk = 0; blk < nblk; blk++) {
    __syncthreads();  // (the matrix pipe is done with the previous block's tile)
    {
      const int l = blk * LM_BLOCK + lane;
#pragma unroll
      for (int k = 0; k < WPT; k++) {
        const int cp = wv + 4 * k;
        if (cp < WT_PAIRS) {
          tile[lane][2 * cp] = wreg[k].x;
          if (2 * cp + 1 < KC) tile[lane][2 * cp + 1] = wreg[k].y;
        }
      }
      if (tid < LM_BLOCK) {
        double cf = 0.0, eb = 0.0, bl = 0.0, kap = 0.0;
        if (l < N && marg) {
          cf = (w_a > 1e-8) ? 1.0 / w_a : 0.0, eb = w_a, bl = w_b;
        } else if (l < N) {
          const double s2a = w_sc * w_sc * w_a;
          const double D2 = fmin(fmax(s2a, 1e-6), 1e32);
          eb = s2a + mu * D2;  // e-block + lm_diagonal^2
          const double einv = 1.0 / eb;
          cf = w_sc * w_sc * einv;
          if (!fl.do_lin) einv_l[l] = einv;  // (a solve repeated with a larger mu: only the weights change)
          bl = w_b, kap = bl / D2;
        }
        lcoef[tid] = cf, le[tid] = eb;
        tile[tid][COL_B] = bl, tile[tid][COL_K] = kap;
      }
    }
    if (blk + 1 < nblk) request(blk + 1);
    __syncthreads();
    // all sixteen steps of the block, straight-line (rows past the block's last landmark are zeros with weight zero): the
    // operand reads of a step are not held behind the previous step's MFMAs
    auto sweep = [&](auto ntl) {
      constexpr int NTL4 = decltype(ntl)::value;
#pragma unroll
      for (int s4 = 0; s4 < LM_BLOCK / 4; s4++) {
        const int row = 4 * s4 + kk;
        const double coef = lcoef[row], eb = le[row];
#pragma unroll
        for (int j = 0; j < NTL4; j++) {
          const double xa = tile[row][ct[j]];
          double xb = tile[row][cu[j]];
          if (scale_k[j]) xb *= eb;  // b / D2 * e  -> z2 column
          acc[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(coef * xa, xb, acc[j], 0, 0, 0);
        }
      }
    };
    // (skipping the tiles in front of a block's first start frame — zeros, the landmarks are sorted — was measured in round 6: the
    // phase is not bound by the matrix pipe, 62 k -> 65 k cycles)
    if (__builtin_amdgcn_readfirstlane(wv) + 12 < NT) sweep(std::integral_constant<int, 4>{});
    else sweep(std::integral_constant<int, 3>{});
  }
  WSTAMP(13);
  double *ss = lw_at<double>(S, A.schur_sum);
#pragma unroll
  for (int j = 0; j < 4; j++) {
    if (wv + 4 * j >= NT) continue;
#pragma unroll
    for (int r = 0; r < 4; r++) ss[(wv + 4 * j) * 256 + r * 64 + lane] = acc[j][r];
  }
  // every landmark's span of the transposed rows has been written: k_setup need not clear the copy again while this window is resident
  if (tid == 0 && !marg && fl.do_lin) S->wt_clean = 1;
  WSTAMP(14);
}
template <bool OFFS = true>  // (see k_lin, kernels_lin.h)
__global__ __launch_bounds__(LW_THREADS, 2) void k_linw(char *base, size_t stride, const LinwArgs A, int mode_bits) {
  extern __shared__ __attribute__((aligned(16))) double lw[];
  linw_body<OFFS>(SLOT(base, stride), lw, A, mode_bits);
}

// ---------------------------------------------------------------------------
// k_linb: grid (groups + 1, batch) x 256, dynamic LDS = LW_LDS_BYTES — the strip sweep of k_linw for ONE LARGE window (thousands
// of landmarks; LinwPlan::big).  A start frame of such a window has hundreds of strips, all of them full: workgroup g takes a
// GROUP — one to eight consecutive strips of one start frame s: a strip (or two) per wave where the tracks are short, two or four
// waves sharing the steps of a strip where they are long — and does with its <= 512 landmarks what k_linw
// does with a window: every observation once (residual, Jacobian basis, rows of W into Slot::Wt, Gram SYRK per step into LDS
// accumulators), then the camera part of H_pp and of g_p summed over its waves, then the Schur SYRK over its landmarks from the
// transposed rows — and leaves ONE partial of LINB_LEN doubles ([camera H_pp packed | camera gradient | Schur tiles | landmark
// scalars]); k_sumb adds the partials up in group order.  Against the role-by-role sweep (k_lin): an observation is evaluated once
// instead of twice, the per-pair tables come through the scalar cache, a quarter of the Schur partials and no Gram partials per
// chunk.  The last workgroup of the grid takes the pose side (IMU factors, prior) like phase 0 of k_linw.
// Differences to k_linw inside a group: all four waves share s, so the pose-pose blocks (s, j) are private per wave like the
// rest (LB_OFF); nothing zero-fills a large window's Wt, so a strip writes every entry of its span (a shorter track's zeros
// too) and the Schur phase takes from a block only the pairs its longest track reaches.
// ---------------------------------------------------------------------------
constexpr int LINB_LEN = SUM_VIS + SCHUR_LEN + 8;
// k_linb_gather: grid (ceil(max(N, NV) / 256), batch) x 256, once per upload of a large window — the copies of the observations in
// the orders the strips read them (anchors by landmark, the others pair-major with their frame pair), made on the device from the
// arrays the upload carries anyway: 36 MB less over PCIe and no host loop over half a million observations at 100 000 landmarks.
__global__ __launch_bounds__(256) void k_linb_gather(char *base, size_t stride) {
  Slot *S = SLOT(base, stride);
  if (!S->linw.big) return;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < S->N) {
    const int o = S->lm_obs0[i];
#pragma unroll
    for (int k = 0; k < 8; k++) S->anc[k][i] = S->obs[k][o];
  }
  if (i < S->NV) {
    const int o = S->pm_obs[i];
    double v[8];
#pragma unroll
    for (int k = 0; k < 8; k++) v[k] = S->obs[k][o];
#pragma unroll
    for (int k = 0; k < 8; k++) S->pmo[k][i] = v[k];
    // the frame pair whose range [pair_obs0[p], pair_obs0[p + 1]) holds i
    int lo = 0, hi = NPAIR;
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (S->linw.pair_obs0[mid] <= i) lo = mid;
      else hi = mid;
    }
    S->pm_pair[i] = (unsigned char)lo;
  }
}
#ifdef LFVIO_LINW_PROFILE  // cycle stamps of group LFVIO_LINB_GROUP (default 0: the most expensive one), tools/linb_clocks.py
#ifndef LFVIO_LINB_GROUP
#define LFVIO_LINB_GROUP 0
#endif
#define BSTAMP(k) do { if (blockIdx.x == LFVIO_LINB_GROUP && blockIdx.y == 0 && threadIdx.x == 0) S->dbg[k] = (long long)__builtin_readcyclecounter(); } while (0)
#else
#define BSTAMP(k) do { } while (0)
#endif
template <bool OFFS>
DEV void linb_body(Slot *S, double *lw, const LinwArgs &A) {
  TRState *tr = &S->tr;
  const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63;
  const TRFlags fl = tr_flags(tr);
  const double mu = tr->mu;
  const LinwPlan *P = &S->linw;
  const int ng = P->ng, g = blockIdx.x;
  if (g > ng) return;
  // a pass that starts with the loop still open is a pass this slot needs (k_lin's count)
  if (g == 0 && tid == 0 && !fl.done) S->passes_used++;
  if (fl.done | (!fl.do_lin & !fl.do_schur)) return;
  LinView lv;
  lv.x = &S->x[fl.cur], lv.tab = &S->tab[fl.cur], lv.lam = lw_at<const double>(S, A.lam[fl.cur]), lv.mu = mu;
  if (g == ng) {  // the pose side
    if (!fl.do_lin) return;
    linw_imu(S, lv, lw, A.imu_out, MODE_SOLVE);
    __syncthreads();
    linw_prior(S, lv, lw);
    return;
  }
  const int L0 = rfl(S->linb_lm0[g]), ns = rfl(S->linb_ns[g]), s = ns >> 16, L1 = L0 + (ns & 0xffff);
  const int est_ex = S->est_ex, est_td = S->est_td;
  double *out = lw_at<double>(S, A.part) + (size_t)g * LINB_LEN;
  if (fl.do_lin) {
    constexpr int P3_E = (SUM_VIS + LW_THREADS - 1) / LW_THREADS;
    int p3[P3_E];
#pragma unroll
    for (int q = 0; q < P3_E; q++) p3[q] = A.asm_tab[tid + LW_THREADS * q < SUM_VIS ? tid + LW_THREADS * q : 0];
    // one, two, (three,) four strips: four, two, one wave(s) per strip, sharing its steps; five to eight: two strips per wave
    // (wave w takes strips w and w + 4; lane t of the wave holds the descriptor of its t-th strip)
    const int nstr = (L1 - L0 + LM_BLOCK - 1) / LM_BLOCK, split = nstr <= 1 ? 4 : nstr <= 2 ? 2 : 1;
    const int mine = wv / split + (lane == 1 ? 4 : 0);
    const int lm0w = L0 + LM_BLOCK * mine, nw = min(max(L1 - lm0w, 0), LM_BLOCK);
    const int kmaxw = nw > 0 ? S->lm_cnt[lm0w + nw - 1] : 0;  // (ascending track length inside a start frame: the last one is the longest)
    const int nmine = (wv / split < nstr ? 1 : 0) + (wv / split + 4 < nstr ? 1 : 0);
    BSTAMP(8);
    for (int e = tid; e < LB_RED0 + LINW_WAVES * 8; e += LW_THREADS) lw[e] = 0.0;
    __syncthreads();
    BSTAMP(9);
    double part[5];
    linw_strips<true, OFFS>(S, lv, A, fl.cur, fl.scaled, MODE_SOLVE, lw, part, 0, nmine, lm0w, nw | (s << 8) | (kmaxw << 16), 1 + wv % split, split);
    if (lane == 0) {
#pragma unroll
      for (int k = 0; k < 5; k++) lw[LB_RED0 + 8 * wv + k] = part[k];
    }
    __syncthreads();
    BSTAMP(10);
    // ---- the group's camera part of H_pp and of the gradient: the waves' private copies in wave order
#pragma unroll
    for (int q = 0; q < P3_E; q++) {
      const int e = tid + LW_THREADS * q, d = p3[q];
      int at = d & 0xffff;
      bool live = true;
      if (d & LWT_ABS) {
        // a pose-pose block (i, j): this group has the ten (s, j) ones, at 36 (j - s - 1) behind every wave's private block
        const int i = (d >> 20) & 15, j = (d >> 24) & 15;  // (the block's frames ride in the table word)
        live = i == s;
        at = LW_PRIV + 36 * (j - s - 1) + (at - LW_OFF0 - 36 * lw_pidx(i, j));
      }
      const double *p0 = lw + LW_PRIV0 + (live ? at : 0);
      double v = (p0[0] + p0[LB_PRIVSZ]) + (p0[2 * LB_PRIVSZ] + p0[3 * LB_PRIVSZ]);
      if (!live || ((d & LWT_EX) && !est_ex) || ((d & LWT_TD) && !est_td)) v = 0.0;
      if (e < SUM_VIS) out[e] = v;
    }
    if (tid < 5) {
      const double *rd = lw + LB_RED0 + tid;
      out[SUM_VIS + SCHUR_LEN + tid] = tid < 4 ? ((rd[0] + rd[8]) + (rd[16] + rd[24])) : fmax(fmax(rd[0], rd[8]), fmax(rd[16], rd[24]));
    }
    __syncthreads();  // (the accumulators are read: their LDS is free)
    BSTAMP(11);
  }
  if (!fl.do_schur) return;
  // ---- the Schur SYRK over the group's landmarks (phase 2 of k_linw over [L0, L1): a block is a strip)
  double(*tile)[WLD + 1] = (double(*)[WLD + 1]) lw;
  double *lcoef = lw + LM_BLOCK * (WLD + 1), *le = lcoef + LM_BLOCK;
  const int kk = lane >> 4, cc = lane & 15;
  double4_t acc[4];
  int ct[4], cu[4];
  bool scale_k[4];
#pragma unroll
  for (int j = 0; j < 4; j++) {
    acc[j] = double4_t{0, 0, 0, 0};
    const int ti = wv + 4 * j;
    const int t = ti < 5 ? 0 : ti < 9 ? 1 : ti < 12 ? 2 : ti < 14 ? 3 : 4;
    const int u = ti - (t * 5 - (t * (t - 1)) / 2) + t;
    ct[j] = ti < NT ? 16 * t + cc : 0, cu[j] = ti < NT ? 16 * u + cc : 0;
    scale_k[j] = cu[j] == COL_K;
  }
  const double2 *wt0 = lw_at<const double2>(S, A.Wt);
  const size_t wld = (size_t)A.wt_ld;
  const double *scale_l = lw_at<const double>(S, A.scale_l), *av = lw_at<const double>(S, A.a), *bv = lw_at<const double>(S, A.b);
  double *einv_l = lw_at<double>(S, A.einv_l);
  const int nblk = (L1 - L0 + LM_BLOCK - 1) / LM_BLOCK;
  constexpr int WPT = (WT_PAIRS + 3) / 4;  // column pairs per thread
  double2 wreg[WPT];
  double w_sc = 0, w_a = 0, w_b = 0;
  int pmax_next = 0;
  auto request = [&](int blk) {
    const int l = L0 + blk * LM_BLOCK + lane;
    // the pairs the block's rows reach: frames s .. s + (longest track of the block) - 1, then extrinsic and td
    const int last = min(L0 + blk * LM_BLOCK + LM_BLOCK, L1) - 1;
    const int pmin = 3 * s, pmax = 3 * (s + rfl(S->lm_cnt[last]));
    pmax_next = pmax;
#pragma unroll
    for (int k = 0; k < WPT; k++) {
      const int cp = wv + 4 * k;
      wreg[k] = (cp < WT_PAIRS && ((cp >= pmin && cp < pmax) || cp >= 33) && l < L1) ? wt0[(size_t)cp * wld + l] : make_double2(0.0, 0.0);
    }
    if (tid < LM_BLOCK && l < L1) w_sc = scale_l[l], w_a = av[l], w_b = bv[l];
  };
  for (int e = tid; e < LM_BLOCK * 6; e += LW_THREADS) tile[e / 6][75 + e % 6] = 0.0;  // pad columns 75 .. 80
  if (nblk > 0) request(0);
  for (int blk = 0; blk < nblk; blk++) {
    __syncthreads();  // (the matrix pipe is done with the previous block's tile)
    // Every row of the block is zero outside the columns of frames s .. s + (its longest track) - 1 and of extrinsic / td / b / kappa
    // (64 .. 74: column block 4): a 16 x 16 tile of the SYRK whose column blocks are not among those is zero and is skipped.
    const int cb_lo = (6 * s) >> 4, cb_hi = (2 * pmax_next - 1) >> 4;  // column blocks of the pose part, from the pairs just requested for this block
    {
      const int l = L0 + blk * LM_BLOCK + lane;
#pragma unroll
      for (int k = 0; k < WPT; k++) {
        const int cp = wv + 4 * k;
        if (cp < WT_PAIRS) {
          tile[lane][2 * cp] = wreg[k].x;
          if (2 * cp + 1 < KC) tile[lane][2 * cp + 1] = wreg[k].y;
        }
      }
      if (tid < LM_BLOCK) {
        double cf = 0.0, eb = 0.0, bl = 0.0, kap = 0.0;
        if (l < L1) {
          const double s2a = w_sc * w_sc * w_a;
          const double D2 = fmin(fmax(s2a, 1e-6), 1e32);
          eb = s2a + mu * D2;  // e-block + lm_diagonal^2
          const double einv = 1.0 / eb;
          cf = w_sc * w_sc * einv;
          if (!fl.do_lin) einv_l[l] = einv;  // (a solve repeated with a larger mu: only the weights change)
          bl = w_b, kap = bl / D2;
        }
        lcoef[tid] = cf, le[tid] = eb;
        tile[tid][COL_B] = bl, tile[tid][COL_K] = kap;
      }
    }
    if (blk + 1 < nblk) request(blk + 1);
    __syncthreads();
    auto block_live = [&](int cb) { return cb == 4 || (cb >= cb_lo && cb <= cb_hi); };
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const int ti = __builtin_amdgcn_readfirstlane(wv) + 4 * j;
      if (ti >= NT) continue;
      const int t = ti < 5 ? 0 : ti < 9 ? 1 : ti < 12 ? 2 : ti < 14 ? 3 : 4, u = ti - (t * 5 - (t * (t - 1)) / 2) + t;
      if (!(block_live(t) && block_live(u))) continue;  // (wave-uniform)
#pragma unroll
      for (int s4 = 0; s4 < LM_BLOCK / 4; s4++) {
        const int row = 4 * s4 + kk;
        const double xa = tile[row][ct[j]];
        double xb = tile[row][cu[j]];
        if (scale_k[j]) xb *= le[row];  // b / D2 * e  -> z2 column
        acc[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(lcoef[row] * xa, xb, acc[j], 0, 0, 0);
      }
    }
  }
  BSTAMP(12);
  double *ss = out + SUM_VIS;
#pragma unroll
  for (int j = 0; j < 4; j++) {
    if (wv + 4 * j >= NT) continue;
#pragma unroll
    for (int r = 0; r < 4; r++) ss[(wv + 4 * j) * 256 + r * 64 + lane] = acc[j][r];
  }
  BSTAMP(13);
}
template <bool OFFS = true>
__global__ __launch_bounds__(LW_THREADS, 2) void k_linb(char *base, size_t stride, const LinwArgs A) {
  extern __shared__ __attribute__((aligned(16))) double lw[];
  linb_body<OFFS>(SLOT(base, stride), lw, A);
}

// ---------------------------------------------------------------------------
// k_sumb: grid (LINB_SUM_GRID, batch) x 1024 — the groups' partials added up in a fixed order (sixteen waves per workgroup: lane e
// of wave q adds entry e of the groups q, q + 16, ..., eight loads in flight; then the sixteen sums in wave order), into the places
// k_sum leaves a window's sums: the COMPLETE packed H_pp (camera part + IMU factors + prior; the rows below the camera part by
// workgroups of their own), the Schur sums, g_p (camera gradient + IMU factors + prior, k_sum's order), the landmark scalars.
// ---------------------------------------------------------------------------
constexpr int LINB_SUM_WGS = (SUM_VIS_PACKED + SCHUR_LEN + 63) / 64, LINB_SUM_THREADS = 1024, LINB_SUM_Q = LINB_SUM_THREADS / 64;
constexpr int LINB_SUM_GRID = LINB_SUM_WGS + 1 + (PACKED - SUM_VIS_PACKED + LINB_SUM_THREADS - 1) / LINB_SUM_THREADS;  // sums | gradient + scalars | the speed / bias rows of H_pp
__global__ __launch_bounds__(LINB_SUM_THREADS) void k_sumb(char *base, size_t stride, const LinwArgs A) {
  Slot *S = SLOT(base, stride);
  const TRFlags fl = tr_flags(&S->tr);
  if (fl.done | (!fl.do_lin & !fl.do_schur)) return;
  __shared__ double red[LINB_SUM_Q][64];
  const int tid = threadIdx.x, q = tid >> 6, lane = tid & 63, ng = S->linw.ng;
  const double *part = lw_at<const double>(S, A.part);
  // entry idx of the partials over the groups q, q + 16, ...: eight loads in flight; a fixed order whatever the timing
  auto some = [&](int idx, bool on, bool is_max) {
    double acc = 0.0;
    if (on) {
      int g = q;
      for (; g + 7 * LINB_SUM_Q < ng; g += 8 * LINB_SUM_Q) {
        double v[8];
#pragma unroll
        for (int k = 0; k < 8; k++) v[k] = part[(size_t)(g + k * LINB_SUM_Q) * LINB_LEN + idx];
#pragma unroll
        for (int k = 0; k < 8; k++) acc = is_max ? fmax(acc, v[k]) : acc + v[k];
      }
      for (; g < ng; g += LINB_SUM_Q) {
        const double v0 = part[(size_t)g * LINB_LEN + idx];
        acc = is_max ? fmax(acc, v0) : acc + v0;
      }
    }
    return acc;
  };
  auto all = [&](bool is_max) {  // (lanes of wave 0, after the barrier)
    double t = red[0][lane];
#pragma unroll
    for (int k = 1; k < LINB_SUM_Q; k++) t = is_max ? fmax(t, red[k][lane]) : t + red[k][lane];
    return t;
  };
  // what the IMU factors and the prior add to packed entry e of H_pp (k_sum's terms in its order; zero in an inactive column): the
  // matrix leaves COMPLETE, for the plain dense solve — assembling it on load costs a single large window 10 us per pass
  const double *imu_out = lw_at<const double>(S, A.imu_out);
  auto pose_terms = [&](int e) {
    int r = (int)((sqrt(8.0 * e + 1.0) - 1.0) * 0.5);
    while ((r + 1) * (r + 2) / 2 <= e) r++;
    while (r * (r + 1) / 2 > e) r--;
    const int c = e - r * (r + 1) / 2;
    const bool ex_off = !S->est_ex, td_off = !S->est_td;
    if ((ex_off && ((r >= off_ex() && r < off_ex() + 6) || (c >= off_ex() && c < off_ex() + 6))) || (td_off && (r == off_td() || c == off_td()))) return 0.0;
    double val = 0.0;
    if (!pose_terms_here(S, r)) return 0.0;  // (a landmark-sharded window: the camera part gets the pose side on ONE rank)
    const int f0 = col_frame(r);
    if (f0 >= 0) {
#pragma unroll
      for (int u = 0; u < 2; u++) {
        const int f = f0 - 1 + u;
        if (f >= 0 && f < LFVIO_WINDOW_SIZE) {
          const int pl = imu_local(r, f), ql = imu_local(c, f);
          if (pl >= 0 && ql >= 0) val += imu_out[(size_t)f * IMU_OUT + pl * 30 + ql];
        }
      }
    }
    if (S->prior_valid) {
      const int pr = S->prior_inv[r], pc = S->prior_inv[c];
      if (pr >= 0 && pc >= 0) val += S->prior_A[pr * S->prior_n + pc];
    }
    return val;
  };
  if ((int)blockIdx.x < LINB_SUM_WGS) {
    // entry e of [packed camera H_pp | Schur tiles]
    const int e = blockIdx.x * 64 + lane;
    const bool on = e < SUM_VIS_PACKED + SCHUR_LEN && (e < SUM_VIS_PACKED ? fl.do_lin : fl.do_schur);
    const int idx = e < SUM_VIS_PACKED ? e : SUM_VIS + (e - SUM_VIS_PACKED);
    red[q][lane] = some(on ? idx : 0, on, false);
    __syncthreads();
    if (q == 0 && on) {
      const double v = all(false);
      if (e < SUM_VIS_PACKED) lw_at<double>(S, A.Hpp)[e] = v + pose_terms(e);  // (the camera part is masked where it is formed)
      else lw_at<double>(S, A.schur_sum)[e - SUM_VIS_PACKED] = v;
    }
    return;
  }
  if (!fl.do_lin) return;
  if ((int)blockIdx.x > LINB_SUM_WGS) {
    // the rows of H_pp below the camera part (speed / bias): IMU factors and prior only
    const int e = SUM_VIS_PACKED + ((int)blockIdx.x - LINB_SUM_WGS - 1) * LINB_SUM_THREADS + tid;
    if (e < PACKED) lw_at<double>(S, A.Hpp)[e] = pose_terms(e);
    return;
  }
  // ---- the last workgroup: camera gradient (KC entries, lanes 0 .. 72 in two trips) and the five landmark scalars
  __shared__ double gv[KC + 8];
  for (int trip = 0; trip < 2; trip++) {
    const int c = trip * 64 + lane;
    const bool on = c < KC + 5;
    const int idx = c < KC ? SUM_VIS_PACKED + c : SUM_VIS + SCHUR_LEN + (c - KC);
    red[q][lane] = some(on ? idx : 0, on, c == KC + 4);
    __syncthreads();
    if (q == 0 && on) gv[c] = all(c == KC + 4);
    __syncthreads();
  }
  const int est_ex = S->est_ex, est_td = S->est_td;
  if (tid < KP) {
    // g_p entry: visual part, the (at most two) IMU factors, the prior — k_sum's order
    const int r = tid;
    double val = r < KC ? gv[r] : 0.0;
    const int f0 = col_frame(r);
    const bool pterms = pose_terms_here(S, r);
    if (f0 >= 0 && pterms) {
#pragma unroll
      for (int u = 0; u < 2; u++) {
        const int f = f0 - 1 + u;
        if (f >= 0 && f < LFVIO_WINDOW_SIZE) {
          const int pl = imu_local(r, f);
          if (pl >= 0) val += imu_out[(size_t)f * IMU_OUT + 900 + pl];
        }
      }
    }
    if (pterms) val += S->prior_g[r];
    const bool act_r = !((!est_ex && r >= off_ex() && r < off_ex() + 6) || (!est_td && r == off_td()));
    lw_at<double>(S, A.gp)[r] = act_r ? val : 0.0;
  }
  if (tid < 5) S->lm_sum[tid] = gv[KC + tid];
  if (S->sharded) {
    // exchange scalars of phase A (k_sum's): local cost (pose-side factors on the owning rank only), gradient norms, Cauchy
    // landmark term, ||lambda||^2; the max is sent as a sum (upper bound, only feeds the 1e-10 gradient tolerance)
    __syncthreads();
    double *sc = S->xch + XOFF_C;
    if (tid < 16) sc[tid] = 0.0;
    __syncthreads();
    if (tid == 0) {
      double cost = gv[KC];
      if (S->pose_side == 1) {
        cost += S->prior_g[KP];
        for (int f = 0; f < LFVIO_WINDOW_SIZE; f++) cost += imu_out[(size_t)f * IMU_OUT + 930];
      }
      sc[XS_COST] = cost;
      sc[XS_G2] = gv[KC + 1], sc[XS_ASV2] = gv[KC + 2], sc[XS_LAM2] = gv[KC + 3], sc[XS_BMAX] = gv[KC + 4];
    }
  }
}
