// kernels_solveb.h — the reduced pose system solved ALONG ITS BLOCK STRUCTURE (round 5).
//
// What Ceres' DENSE_SCHUR hands to its dense Cholesky (estimator.cpp:810-825) is, behind the landmark elimination, a 172 x 172
// system whose speed/bias part is a chain: the IMU factor between frames f and f + 1 couples Pose f, SpeedBias f, Pose f + 1,
// SpeedBias f + 1 and nothing else (estimator.cpp:717-724), the landmarks only touch the 73 camera columns (poses, extrinsic,
// td: estimator.cpp:755,768), and the prior of the reference's own marginalization carries SpeedBias 0 only (estimator.cpp:
// 833-939).  So with the unknowns ordered  SpeedBias 10, 9, ..., 0, camera  the factorization has
//   * a block-bidiagonal 99 x 99 corner: eleven 9 x 9 Cholesky factors L_f and ten 9 x 9 couplings Z_f = L_f^-1 A[sb_f, sb_f-1],
//   * rows W_f = L_f^-1 [C~_f | b~_f] (9 x 74) that are only needed once — for the update of the next block,
//     [C~_f-1 | b~_f-1] = [C_f-1 | b_f-1] - Z_f^T W_f, and for the camera block, A_cc -= W_f^T W_f (a SYRK on the matrix pipe) —
//   * and a dense camera block of 73 (+ 3 unit pivots that keep the 4-pivot panels whole; rhs row 76): five block columns of
//     tile_cholesky instead of eleven.
// The solution of the chain comes from its own factor afterwards, y_S = A_SS^-1 (b_S - A_SC y_c): W is never stored.
//
// LDS: 15 tiles + 22 blocks of 9 x 10 + two W panels + the vectors = 78 KB instead of 156 — TWO windows of a resident batch share
// a CU in this phase as they do in k_linw and k_stepw.  A window whose prior carries a SpeedBias block of another frame (legal
// input of the C-ABI, not something the reference's marginalization produces) does not have this structure: the host sends
// it to k_solve_dense (lfvio_hip.hip, SlotHostInfo::sb_chain).
//
// MEASURED (round 5, one MI355X; tools/scratch/cmp_solveb.py, bench.py): correct — the Gauss-Newton step and the quadratic forms
// agree with the dense solve to 2e-8 / 6e-9 of their scale, whole calls follow the dense solve and the oracle step for step — and
// SLOWER: 70 us per window against 50 (cycles: loads 31 k, scaling + build 11 k, the chain 86 k, camera Cholesky 28 k, substitutions
// 30 k, forms 11 k), 138 us per launch of 512 windows against 125.  The chain is eleven serial steps of 9 x 9 work: wave 0 needs
// 4 000 cycles a block (660 instructions of one row per lane — a wave64 FP64 instruction is four issue cycles whether nine lanes
// matter or sixty-four), the column threads 5 000 (a load - wait - fma pattern over 126 LDS broadcasts), and two workgroups on a
// CU share its SIMDs, so the second resident window buys little where every wave is issue-bound.  OFF by default
// (lfvio_debug_set_block_solve / LFVIO_BLOCK_SOLVE=1); kept as the statement of the structure and as the 78 KB form.
//
// Same semantics as solve_body (kernels_solve.h): Jacobi scaling, diagonal_, gradient_, Cauchy point, mu retry on a pivot that is
// not positive, Gauss-Newton step, the quadratic forms of the dogleg model; a different elimination ORDER, i.e. different
// rounding (parity bars: tests/test_solveb.py).
#pragma once
#include "kernels_solve.h"

constexpr int CAM_TN = 5, CAM_TK = 76;                // camera block: 73 unknowns + 3 unit pivots, rhs = row 76
constexpr int CAM_TILES = CAM_TN * (CAM_TN + 1) / 2;  // 15
constexpr int SBF = LFVIO_NUM_FRAMES;                 // 11 speed/bias blocks
constexpr int SBR = 10, SBB = 9 * SBR;                // a 9 x 9 block in LDS: row stride 10
constexpr int WB_LD = 81, WB_ROWS = 12;               // a W panel: 9 rows (+ 3 zero rows: K of the SYRK in steps of 4) x 80 camera columns
constexpr int DE_D = 45 * SBF, DE_LEN = DE_D + 81 * (SBF - 1), DE_SLOTS = (DE_LEN + SOLVE_THREADS - 1) / SOLVE_THREADS;  // 1305 entries, 6 per thread
constexpr int COLT0 = 64, NCOLT = KC + 1;             // the column threads: tid 64 .. 137 own camera column tid - 64 (73: the rhs) of every W
constexpr int SBV = 176;
constexpr int SBO_HS = 0;
constexpr int SBO_LM = SBO_HS + CAM_TILES * TSZ;      // D~_f, then L_f
constexpr int SBO_ZM = SBO_LM + SBF * SBB;            // A[sb_f, sb_f-1], then Z_f (f = 1 .. 10)
constexpr int SBO_LI = SBO_ZM + SBF * SBB;            // 1 / L_f[k][k]
constexpr int SBO_WB = SBO_LI + SBF * SBR;
constexpr int SBO_VEC = SBO_WB + 2 * WB_ROWS * WB_LD;
constexpr int SBO_END = SBO_VEC + 7 * SBV + 80 + 320;
constexpr size_t SOLVEB_LDS = (size_t)SBO_END * sizeof(double);
static_assert(SOLVEB_LDS <= 80 * 1024, "two workgroups of the block solve share the 160 KB of a CU");
constexpr int RS_LD = 20, RS0 = SBF * 9 * RS_LD;      // the partial products of b_S - A_SC y_c (in the dead tiles): [f][r][18 (+2)], then [r][80] of block 0

// entry e of the thread-distributed part of the chain: the lower triangles of the eleven diagonal blocks, then the ten couplings
DEV void de_decode(int e, int &f, int &r, int &c, bool &coupling) {
  if (e < DE_D) {
    f = e / 45;
    const int t = e - 45 * f;
    r = t < 1 ? 0 : t < 3 ? 1 : t < 6 ? 2 : t < 10 ? 3 : t < 15 ? 4 : t < 21 ? 5 : t < 28 ? 6 : t < 36 ? 7 : 8;
    c = t - r * (r + 1) / 2;
    coupling = false;
  } else {
    const int q = e - DE_D;
    f = 1 + q / 81;
    const int t = q - 81 * (f - 1);
    r = t / 9;
    c = t - 9 * r;
    coupling = true;
  }
}

// Entry (i, j), i >= j (tangent columns), of the pose-side Gauss-Newton Hessian.  ASSEMBLE (a batch linearized by k_linw): the
// exchange buffer holds the visual terms of the camera part only; the (at most two) IMU factors and the prior are added here, in
// k_sum's order.  Every load is independent of the others but the prior's (its column map first).
struct HSrc {
  const double *Hg, *imu_out, *prior_A;
  const int *prior_inv;
  int prior_n;  // 0: no prior to add
};
template <bool ASSEMBLE>
DEV double h_entry(const HSrc &src, int i, int j, bool valid) {
  if (!valid) return 0.0;
  if (!ASSEMBLE) return src.Hg[i * (i + 1) / 2 + j];
  double v = i < KC ? src.Hg[i * (i + 1) / 2 + j] : 0.0;
  const int f0 = col_frame(i);
  if (f0 >= 0) {
#pragma unroll
    for (int u = 0; u < 2; u++) {
      const int f = f0 - 1 + u;
      if (f >= 0 && f < LFVIO_WINDOW_SIZE) {
        const int pl = imu_local(i, f), ql = imu_local(j, f);
        if (pl >= 0 && ql >= 0) v += src.imu_out[(size_t)f * IMU_OUT + pl * 30 + ql];
      }
    }
  }
  if (src.prior_n > 0) {
    const int pr = src.prior_inv[i], pc = src.prior_inv[j];
    if (pr >= 0 && pc >= 0) v += src.prior_A[pr * src.prior_n + pc];
  }
  return v;
}

template <bool ASSEMBLE>
DEV void solveb_body(Slot *S, double *smem, long long xch_off, long long imu_off, long long prior_A_off) {
  TRState *tr = &S->tr;
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int er = tid >> 4, ek = tid & 15, esw = tsw(er, ek);  // this thread's entry of every camera tile
  const double *xch = (const double *)((const char *)S + xch_off);
  if (ASSEMBLE) {
    const TRFlags f0 = tr_flags_decided(S);
    if ((f0.done | !f0.do_schur) && !S->dec_pending) return;
  }
  const bool est_ex = S->est_ex != 0, est_td = S->est_td != 0;
  auto active = [&](int c) { return (est_ex || c < off_ex() || c >= off_ex() + 6) && (est_td || c != off_td()); };
  // ---- everything this thread will ever read of H_pp, once, into registers: its entry of the 15 camera tiles, its share of the
  //      chain's diagonal and coupling blocks, and — the column threads — the speed/bias rows of their camera column
  HSrc src;
  src.Hg = xch + XOFF_H;
  src.imu_out = (const double *)((const char *)S + imu_off);
  src.prior_A = (const double *)((const char *)S + prior_A_off);
  src.prior_inv = S->prior_inv;
  src.prior_n = (ASSEMBLE && S->prior_valid && (!S->sharded || S->pose_side)) ? S->prior_n : 0;
  const double *Sg = xch + XOFF_S;
  double hreg[CAM_TILES], sreg[CAM_TILES], srhs[CAM_TN], scross = 0.0, gval = 0.0, cp = 0.0;
#pragma unroll
  for (int t = 0; t < CAM_TILES; t++) {
    const int i = 16 * tile_a(t) + er, j = 16 * tile_b(t) + ek;
    hreg[t] = h_entry<ASSEMBLE>(src, i, j, i < KC && j <= i && active(i) && active(j));
    sreg[t] = Sg[schur_index(min(i, j), max(i, j))];
  }
#pragma unroll
  for (int b = 0; b < CAM_TN; b++) srhs[b] = Sg[schur_index(16 * b + ek, COL_B)];
  if (tid < KC) scross = Sg[schur_index(tid, COL_K)];
  if (tid < KP) gval = xch[XOFF_G + tid];
  double de[DE_SLOTS];
#pragma unroll
  for (int s = 0; s < DE_SLOTS; s++) {
    const int e = tid + SOLVE_THREADS * s;
    int f, r, c;
    bool cpl;
    de_decode(e < DE_LEN ? e : 0, f, r, c, cpl);
    de[s] = h_entry<ASSEMBLE>(src, off_sb(f) + r, off_sb(cpl ? f - 1 : f) + c, e < DE_LEN);
  }
  // column thread cc < 73: rows of SpeedBias f of its column, f = p - 1, p, p + 1 (p: the column's pose; slots 0 .. 2) and,
  // where that does not cover it, f = 0 (slot 3: the prior couples SpeedBias 0 to every camera column)
  const bool colt = tid >= COLT0 && tid < COLT0 + NCOLT;
  const int cc = colt ? tid - COLT0 : 0;
  const bool is_rhs = cc == KC;
  const int cpose = cc < 66 ? cc / 6 : 100;
  // (36 entries per column thread: they go to the slot's scratch — the marginalization's, idle during the passes — and come back
  // nine at a time, a block ahead of the chain, and whole for the two sums at the end: 18 registers instead of 72 across the loop)
  double *crg = S->mscr;
  auto cr_at = [&](int s, int r) { return crg + (s * 9 + r) * 80 + cc; };
#pragma unroll
  for (int s = 0; s < 4; s++) {
    const int f = s < 3 ? cpose - 1 + s : 0;
    const bool ok = colt && !is_rhs && (s < 3 ? (f >= 0 && f < SBF) : cpose >= 2) && active(cc);
#pragma unroll
    for (int r = 0; r < 9; r++) {
      const double v = h_entry<ASSEMBLE>(src, off_sb(ok ? f : 0) + r, cc, ok);
      if (colt && !is_rhs) *cr_at(s, r) = v;
    }
  }
  const int sharded = S->sharded;
  const double *ls = sharded ? xch + XOFF_C : S->lm_sum;
  if (tid < 12) {
    if (tid == 0) cp = ls[0];
    else if (!sharded) cp = tid == 1 ? S->prior_g[KP] : src.imu_out[(size_t)(tid - 2) * IMU_OUT + 930];
  }
  const TRFlags fl = tr_flags_decided(S);
  const int dec_pending = S->dec_pending;
  const double mu_decided = S->dec.mu;
  const double mu_header = tr->mu;
#pragma unroll
  for (int t = 0; t < CAM_TILES; t++) {
    SOLVE_KEEP(hreg[t]);
    SOLVE_KEEP(sreg[t]);
  }
#pragma unroll
  for (int s = 0; s < DE_SLOTS; s++) SOLVE_KEEP(de[s]);
  SOLVE_KEEP(scross);
  SOLVE_KEEP(gval);
  SOLVE_KEEP(cp);
  SOLVE_KEEP(mu_header);
  if (threadIdx.x == 0 && dec_pending) {
    decision_to_header(tr, S->dec);
    S->dec_pending = 0;
  }
  if (fl.done | !fl.do_schur) return;
  double *Hs = smem + SBO_HS, *Lm = smem + SBO_LM, *Zm = smem + SBO_ZM, *Li = smem + SBO_LI, *Wb = smem + SBO_WB;
  double *g = smem + SBO_VEC, *sc = g + SBV, *dg = sc + SBV, *gr = dg + SBV, *Gd = gr + SBV, *yv = Gd + SBV, *hv = yv + SBV;
  double *invd = hv + SBV, *scratch = invd + 80;
  STAMP(S, 0);
  // ---- the diagonal of H_pp (for the scaling), g, the pieces of the cost
#pragma unroll
  for (int a = 0; a < CAM_TN; a++)
    if (er == ek && 16 * a + er < KC) hv[16 * a + er] = hreg[tile_id(a, a)];
#pragma unroll
  for (int s = 0; s < DE_SLOTS; s++) {
    const int e = tid + SOLVE_THREADS * s;
    int f, r, c;
    bool cpl;
    de_decode(e < DE_LEN ? e : 0, f, r, c, cpl);
    if (e < DE_D && r == c) hv[off_sb(f) + r] = de[s];
  }
  if (tid < KP) g[tid] = gval;
  if (tid < 12) scratch[tid] = cp;
  for (int e = tid; e < 2 * WB_ROWS * WB_LD; e += SOLVE_THREADS) Wb[e] = 0.0;  // (rows 9 .. 11 and the columns nobody owns stay zero)
  __syncthreads();
  if (fl.do_lin && tid == 0) {
    double cost = scratch[0];
    if (!sharded) {
      cost += scratch[1];
      for (int f = 0; f < LFVIO_WINDOW_SIZE; f++) cost += scratch[2 + f];
    }
    tr->x_cost = cost;
  }
  __syncthreads();
  STAMP(S, 1);
  // ---- Jacobi scaling (iteration 0 only), diagonal_, gradient_  (dogleg_strategy.cc ComputeStep)
  const double mu = dec_pending ? mu_decided : mu_header;
  if (tid < KP) {
    const int i = tid;
    const double hii = hv[i];
    double s;
    if (!tr->scaled) {
      s = 1.0 / (1.0 + sqrt(hii));
      S->scale_p[i] = s;
    } else {
      s = S->scale_p[i];
    }
    const double d = sqrt(fmin(fmax(s * s * hii, 1e-6), 1e32));
    const double gi = active(i) ? s * g[i] / d : 0.0;
    sc[i] = s, dg[i] = d, gr[i] = gi;
    Gd[i] = s * gi / d;
    S->diag_p[i] = d;
    S->grad_p[i] = gi;
  }
  __syncthreads();
  STAMP(S, 2);
  // ---- the scaled, damped system in place — camera tiles, chain blocks — and the Cauchy point's G^T H G on the way
  double qgg_part = 0;
  {
    double Gi[CAM_TN], Gj[CAM_TN], si[CAM_TN], sj[CAM_TN];
    bool ai[CAM_TN], aj[CAM_TN];
#pragma unroll
    for (int a = 0; a < CAM_TN; a++) {
      const int i = 16 * a + er, j = 16 * a + ek;
      Gi[a] = i < KC ? Gd[i] : 0.0, si[a] = i < KC ? sc[i] : 0.0, ai[a] = i < KC && active(i);
      Gj[a] = j < KC ? Gd[j] : 0.0, sj[a] = j < KC ? sc[j] : 0.0, aj[a] = j < KC && active(j);
    }
#pragma unroll
    for (int t = 0; t < CAM_TILES; t++) {
      const int a = tile_a(t), b = tile_b(t);
      const int i = 16 * a + er, j = 16 * b + ek;
      double v = 0.0;
      if (i < KC && j <= i) {
        double h = hreg[t];
        qgg_part = fma(h * Gi[a], (i == j) ? Gj[b] : 2.0 * Gj[b], qgg_part);
        if (ai[a] && aj[b]) {
          h -= sreg[t];
          v = si[a] * sj[b] * h;
          if (i == j) v += mu * dg[i] * dg[i];
        } else {
          v = (i == j) ? 1.0 : 0.0;
        }
      } else if (i < CAM_TK && i == j) {
        v = 1.0;  // the three unit pivots
      } else if (i == CAM_TK && j < KC) {
        if (aj[b]) v = sj[b] * (g[j] - srhs[b]);  // z1
      }
      Hs[t * TSZ + esw] = v;
    }
#pragma unroll
    for (int s = 0; s < DE_SLOTS; s++) {
      const int e = tid + SOLVE_THREADS * s;
      int f, r, c;
      bool cpl;
      de_decode(e < DE_LEN ? e : 0, f, r, c, cpl);
      const int i = off_sb(f) + r, j = off_sb(cpl ? f - 1 : f) + c;
      if (e < DE_LEN) {
        const double h = de[s];
        qgg_part = fma(h * Gd[i], (i == j) ? Gd[j] : 2.0 * Gd[j], qgg_part);
        double v = sc[i] * sc[j] * h;
        if (i == j) v += mu * dg[i] * dg[i];
        (cpl ? Zm : Lm)[f * SBB + r * SBR + c] = v;
      }
    }
  }
  // (the Cauchy point's sums are taken at the end, with the other quadratic forms: nothing in this kernel needs alpha)
  __syncthreads();
  STAMP(S, 3);

  // ---- the chain, SpeedBias 10 .. 0, as a pipeline of three stages one block apart (one barrier per block):
  //   wave 0             L_g = chol(D~_g), Z_g = L_g^-1 A[sb_g, sb_g-1], D~_g-1 -= Z_g^T Z_g   — the only serial part: nothing in
  //                      it depends on the camera columns, so it runs ahead;
  //   column threads     (waves 1, 2; one camera column — or the rhs — each, the column of W in registers from block to block)
  //                      t = [C_g | b_g] - Z_g+1^T w_g+1,  w_g = L_g^-1 t  (L, Z as LDS broadcasts), w_g into the panel;
  //   waves 1 .. 3       A_cc -= W_g^T W_g: tiles of v_mfma_f64_16x16x4_f64 accumulated in registers over all blocks (a column
  //                      of tiles per wave: tile columns left of 6 (g - 1) / 16 are zero and skipped).
  bool bad = !(mu < 1.0);  // ComputeGaussNewtonStep: `while (mu_ < max_mu_)` — no attempt at mu >= 1
  constexpr int SY_T = 5;
  solve_d4 acc[SY_T];
#pragma unroll
  for (int t = 0; t < SY_T; t++) acc[t] = solve_d4{0.0, 0.0, 0.0, 0.0};
  // this wave's five tiles of the SYRK — wave 3 (no column of W to compute) the ones every block touches
  auto my_tile = [&](int t, int &a, int &b) {
    if (wave == 3) {  // (4,4) (4,3) (3,3) (4,2) (3,2)
      a = t == 2 || t == 4 ? 3 : 4, b = t == 0 ? 4 : t < 3 ? 3 : 2;
      return true;
    }
    if (wave == 1) {  // (2,2) (4,1) (3,1) (2,1) (1,1)
      a = t == 0 ? 2 : 5 - t, b = t == 0 ? 2 : 1;
      return true;
    }
    a = t, b = 0;  // wave 2: tile column 0
    return wave == 2;
  };
  double wcol[9];
#pragma unroll
  for (int r = 0; r < 9; r++) wcol[r] = 0.0;
  const double my_sc = (colt && !is_rhs) ? sc[cc] : 0.0;  // (a column that is not estimated holds zeros)
  // the rows of SpeedBias g of this thread's column: slot g - p + 1 (block 0 of the columns its neighbours do not cover: slot 3)
  auto col_fetch = [&](int gq, double (&cv)[9]) {
    int sel = cc < 66 ? gq - cpose + 1 : -1;
    if (sel > 2) sel = -1;
    if (gq == 0 && cpose >= 2) sel = 3;
    const bool ok = colt && !is_rhs && gq >= 0 && sel >= 0;
#pragma unroll
    for (int r = 0; r < 9; r++) cv[r] = ok ? *cr_at(sel, r) : 0.0;
  };
  double cnext[9];
  col_fetch(SBF - 1, cnext);
  for (int it = 0; it < SBF + 2; it++) {
    const int g0 = SBF - 1 - it, g1 = SBF - it, g2 = SBF + 1 - it;
    if (wave == 0 && g0 >= 0) {
      const int r9 = lane < 9 ? lane : 8;
      double *Lg = Lm + g0 * SBB;
      double a[9], l[9], rs[9];
#pragma unroll
      for (int k = 0; k < 9; k++) a[k] = Lg[r9 * SBR + k];
#pragma unroll
      for (int k = 0; k < 9; k++) {
        const double d = readlane_f64(a[k], k);
        if (!(d > 0.0)) bad = true;
        rs[k] = fast_rsqrt(d);
        l[k] = a[k] * rs[k];
#pragma unroll
        for (int j = k + 1; j < 9; j++) a[j] = fma(-l[k], readlane_f64(l[k], j), a[j]);
      }
      if (lane < 9) {
#pragma unroll
        for (int k = 0; k < 9; k++) Lg[lane * SBR + k] = k <= lane ? l[k] : 0.0;
      }
#pragma unroll
      for (int k = 0; k < 9; k++)
        if (lane == k) Li[g0 * SBR + k] = rs[k];
      if (g0 > 0) {
        double *Zg = Zm + g0 * SBB;
        double z[9];
#pragma unroll
        for (int k = 0; k < 9; k++) z[k] = Zg[k * SBR + r9];  // column r9 of A[sb_g, sb_g-1]
#pragma unroll
        for (int k = 0; k < 9; k++) {
          double s = z[k];
#pragma unroll
          for (int j = 0; j < k; j++) s = fma(-readlane_f64(l[j], k), z[j], s);  // L[k][j] is lane k's l[j]
          z[k] = s * rs[k];
        }
        if (lane < 9) {
#pragma unroll
          for (int k = 0; k < 9; k++) Zg[k * SBR + lane] = z[k];
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // (the LDS operations of one wave complete in program order)
        // D~_g-1 -= Z^T Z, the lower triangle: one entry per lane
        const int t = lane < 45 ? lane : 44;
        const int rr = t < 1 ? 0 : t < 3 ? 1 : t < 6 ? 2 : t < 10 ? 3 : t < 15 ? 4 : t < 21 ? 5 : t < 28 ? 6 : t < 36 ? 7 : 8, c2 = t - rr * (rr + 1) / 2;
        double s = 0.0;
#pragma unroll
        for (int k = 0; k < 9; k++) s = fma(Zg[k * SBR + rr], Zg[k * SBR + c2], s);
        if (lane < 45) Lm[(g0 - 1) * SBB + rr * SBR + c2] -= s;
      }
    }
    if (wave != 0 && colt && g1 >= 0 && g1 < SBF) {
      double t[9];
      if (!is_rhs) {
#pragma unroll
        for (int r = 0; r < 9; r++) t[r] = sc[off_sb(g1) + r] * cnext[r] * my_sc;
      } else {
#pragma unroll
        for (int r = 0; r < 9; r++) t[r] = sc[off_sb(g1) + r] * g[off_sb(g1) + r];
      }
      col_fetch(g1 - 1, cnext);  // (the next block's entries: a block of time to arrive)
      if (g1 < SBF - 1) {
        const double *Zn = Zm + (g1 + 1) * SBB;
#pragma unroll
        for (int k = 0; k < 9; k++)
#pragma unroll
          for (int r = 0; r < 9; r++) t[r] = fma(-Zn[k * SBR + r], wcol[k], t[r]);
      }
      const double *Lg = Lm + g1 * SBB, *Lig = Li + g1 * SBR;
#pragma unroll
      for (int r = 0; r < 9; r++) {
        double s = t[r];
#pragma unroll
        for (int j = 0; j < r; j++) s = fma(-Lg[r * SBR + j], wcol[j], s);
        wcol[r] = s * Lig[r];
      }
      double *Wg = Wb + (g1 & 1) * WB_ROWS * WB_LD + (is_rhs ? CAM_TK : cc);
#pragma unroll
      for (int r = 0; r < 9; r++) Wg[r * WB_LD] = wcol[r];
    }
    if (wave != 0 && g2 >= 0 && g2 < SBF) {
      const double *Wg = Wb + (g2 & 1) * WB_ROWS * WB_LD;
      const int bmin = g2 >= 2 ? (6 * (g2 - 1)) >> 4 : 0;
      const int c = lane & 15, gq = lane >> 4;
#pragma unroll
      for (int t = 0; t < SY_T; t++) {
        int a, b;
        if (my_tile(t, a, b) && b >= bmin) {  // wave-uniform
#pragma unroll
          for (int q = 0; q < 3; q++) {
            const double xa = Wg[(gq + 4 * q) * WB_LD + 16 * a + c], xb = Wg[(gq + 4 * q) * WB_LD + 16 * b + c];
            acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(xa, xb, acc[t], 0, 0, 0);
          }
        }
      }
    }
    __syncthreads();
  }
  {  // the sums leave the registers: every camera tile has one owner
    const int c = lane & 15, gq = lane >> 4;
#pragma unroll
    for (int t = 0; t < SY_T; t++) {
      int a, b;
      if (my_tile(t, a, b)) {
        double *Tc = Hs + tile_id(a, b) * TSZ + gq * TLD + c;
#pragma unroll
        for (int r = 0; r < 4; r++) Tc[4 * TLD * r] -= acc[t][r];
      }
    }
  }
  __syncthreads();
  STAMP(S, 4);
  // ---- the camera block: five block columns of the tiled Cholesky, the rhs row riding along
  bad = tile_cholesky<CAM_TN, CAM_TK>(Hs, invd, tid, bad);
  {
    double f = bad ? 1.0 : 0.0;
    f = block_max(f, scratch, tid);
    bad = f > 0.0;
  }
  STAMP(S, 5);
  tile_backsub<CAM_TN, CAM_TK>(Hs, yv, invd, tid);  // y_c in yv[0, 73) (the unit pivots' entries are zero)
  // ---- the chain's unknowns from its own factor:  y_S = A_SS^-1 (b_S - A_SC y_c).  The products of the column threads'
  //      entries go through the (dead) tiles, one slot each; then r_S, then the two sweeps of the block-bidiagonal factor.
  double *red = Hs;
  double cr[4][9];  // (from here to the end: the column threads' entries once more)
#pragma unroll
  for (int s = 0; s < 4; s++)
#pragma unroll
    for (int r = 0; r < 9; r++) cr[s][r] = (colt && !is_rhs) ? *cr_at(s, r) : 0.0;
  if (colt && !is_rhs) {
    const double yc = my_sc * yv[cc];
#pragma unroll
    for (int s = 0; s < 4; s++) {
      const int f = s < 3 ? cpose - 1 + s : 0;
      if ((s < 3 && f >= 0 && f < SBF) || (s == 3 && cpose >= 2)) {
        double *dst = f == 0 ? red + RS0 + cc : red + (f * 9) * RS_LD + (cc - 6 * (f - 1));
        const int ld = f == 0 ? 80 : RS_LD;
#pragma unroll
        for (int r = 0; r < 9; r++) dst[r * ld] = sc[off_sb(f) + r] * cr[s][r] * yc;
      }
    }
  }
  __syncthreads();
  if (tid < 9 * SBF) {
    const int f = tid / 9, r = tid - 9 * f, i = off_sb(f) + r;
    double s = 0.0;
    if (f == 0) {
      for (int c = 0; c < KC; c++) s += red[RS0 + r * 80 + c];
    } else {
      const int nc = f == SBF - 1 ? 12 : 18;
      for (int c = 0; c < nc; c++) s += red[(f * 9 + r) * RS_LD + c];
    }
    hv[i] = sc[i] * g[i] - s;
  }
  __syncthreads();
  if (wave == 0) {
    const int r9 = lane < 9 ? lane : 8;
    // forward: t_f = L_f^-1 (r_f - Z_f+1^T t_f+1), f = 10 .. 0; lane k holds entry k
    double prev = 0.0;
    for (int f = SBF - 1; f >= 0; f--) {
      double s = hv[off_sb(f) + r9];
      double lrow[9];
#pragma unroll
      for (int k = 0; k < 9; k++) lrow[k] = Lm[f * SBB + r9 * SBR + k];
      const double inv = Li[f * SBR + r9];
      if (f < SBF - 1) {
#pragma unroll
        for (int k = 0; k < 9; k++) s = fma(-Zm[(f + 1) * SBB + k * SBR + r9], readlane_f64(prev, k), s);
      }
      double tv = 0.0;
#pragma unroll
      for (int k = 0; k < 9; k++) {
        const double tk = readlane_f64(s * inv, k);
        tv = lane == k ? tk : tv;
        s = fma(-lrow[k], tk, s);  // rows below k (L[r][k] is zero above the diagonal; row k itself is done)
      }
      if (lane < 9) hv[off_sb(f) + lane] = tv;
      prev = tv;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    // backward: y_f = L_f^-T (t_f - Z_f y_f-1), f = 0 .. 10
    prev = 0.0;
    for (int f = 0; f < SBF; f++) {
      double s = hv[off_sb(f) + r9];
      double lcol[9];
#pragma unroll
      for (int k = 0; k < 9; k++) lcol[k] = Lm[f * SBB + k * SBR + r9];  // L[k][r]: zero for k < r
      const double inv = Li[f * SBR + r9];
      if (f > 0) {
#pragma unroll
        for (int k = 0; k < 9; k++) s = fma(-Zm[f * SBB + r9 * SBR + k], readlane_f64(prev, k), s);
      }
      double yk_mine = 0.0;
#pragma unroll
      for (int k = 8; k >= 0; k--) {
        const double yk = readlane_f64(s * inv, k);
        yk_mine = lane == k ? yk : yk_mine;
        s = fma(-lcol[k], yk, s);
      }
      if (lane < 9) yv[off_sb(f) + lane] = yk_mine;
      prev = yk_mine;
    }
  }
  __syncthreads();
  STAMP(S, 6);
  {
    double f = 0.0;
    if (tid < KP && !isfinite(yv[tid])) f = 1.0;
    f = block_max(f, scratch, tid);
    if (f > 0.0) bad = true;
  }
  if (bad) {
    // LINEAR_SOLVER_FAILURE inside ComputeGaussNewtonStep: mu *= 10 and retry (same Jacobian)
    if (tid == 0) {
      tr->chol_fail = 1;
      if (mu < 1.0) tr->mu = mu * 10.0;
    }
    return;
  }
  // ---- Gauss-Newton step, directions and pose-side quadratic forms
  if (tid < KP) {
    const double y = yv[tid];
    const double gn = -dg[tid] * y;  // gauss_newton_step_ = -diagonal_ * y
    S->gn_p[tid] = gn;
    const double Nd = -sc[tid] * y;  // unscaled GN direction
    yv[tid] = Nd;
    hv[tid] = gn;
    if (tid < KC) {
      S->uc_grad[tid] = Gd[tid];
      S->uc_gn[tid] = Nd;
    }
  }
  if (tid >= KC && tid < WLD) S->uc_grad[tid] = S->uc_gn[tid] = 0.0;
  __syncthreads();
  {
    // G^T H N and N^T H N from the entries of H_pp this thread has held in registers since the start
    double qgn = 0, qnn = 0;
    {
      double Gi[CAM_TN], Ni[CAM_TN], Gj[CAM_TN], Nj[CAM_TN];
#pragma unroll
      for (int a = 0; a < CAM_TN; a++) {
        const int i = 16 * a + er, j = 16 * a + ek;
        Gi[a] = i < KC ? Gd[i] : 0.0, Ni[a] = i < KC ? yv[i] : 0.0;
        Gj[a] = j < KC ? Gd[j] : 0.0, Nj[a] = j < KC ? yv[j] : 0.0;
      }
#pragma unroll
      for (int t = 0; t < CAM_TILES; t++) {
        const int a = tile_a(t), b = tile_b(t);
        const int i = 16 * a + er, j = 16 * b + ek;
        if (i < KC && j <= i) {
          const double h = hreg[t];
          if (i == j) {
            qgn = fma(h, Gi[a] * Nj[b], qgn);
            qnn = fma(h, Ni[a] * Nj[b], qnn);
          } else {
            qgn = fma(h, Gi[a] * Nj[b] + Ni[a] * Gj[b], qgn);
            qnn = fma(h, 2.0 * Ni[a] * Nj[b], qnn);
          }
        }
      }
    }
#pragma unroll
    for (int s = 0; s < DE_SLOTS; s++) {
      const int e = tid + SOLVE_THREADS * s;
      int f, r, c;
      bool cpl;
      de_decode(e < DE_LEN ? e : 0, f, r, c, cpl);
      const int i = off_sb(f) + r, j = off_sb(cpl ? f - 1 : f) + c;
      if (e < DE_LEN) {
        const double h = de[s];
        if (i == j) {
          qgn = fma(h, Gd[i] * yv[j], qgn);
          qnn = fma(h, yv[i] * yv[j], qnn);
        } else {
          qgn = fma(h, Gd[i] * yv[j] + yv[i] * Gd[j], qgn);
          qnn = fma(h, 2.0 * yv[i] * yv[j], qnn);
        }
      }
    }
    if (colt && !is_rhs) {
      const double Gc = Gd[cc], Nc = yv[cc];
#pragma unroll
      for (int s = 0; s < 4; s++) {
        const int f = s < 3 ? cpose - 1 + s : 0;
        const int fc = (f >= 0 && f < SBF) ? f : 0;
#pragma unroll
        for (int r = 0; r < 9; r++) {
          const double h = cr[s][r];  // (an absent slot holds zeros)
          const int i = off_sb(fc) + r;
          qgg_part = fma(h * Gd[i], 2.0 * Gc, qgg_part);
          qgn = fma(h, Gd[i] * Nc + yv[i] * Gc, qgn);
          qnn = fma(h, 2.0 * yv[i] * Nc, qnn);
        }
      }
    }
    double gn2 = 0, ggn = 0, gG = 0, gN = 0, gs = 0, cross = 0;
    if (tid < KP) {
      const double gn = hv[tid];
      gn2 = gn * gn;
      ggn = gr[tid] * gn;
      gG = g[tid] * Gd[tid];
      gN = g[tid] * yv[tid];
      gs = gr[tid] * gr[tid];
      if (tid < KC) cross = scross * Gd[tid];  // z2 . G_c
    }
    double sums[9] = {gn2, ggn, gG, gN, qgn, qnn, qgg_part, gs, cross};
    block_sum_n(sums, scratch, tid);
    gn2 = sums[0], ggn = sums[1], gG = sums[2], gN = sums[3], qgn = sums[4], qnn = sums[5];
    STAMP(S, 7);
    if (tid == 0) {
      // Cauchy point: alpha = ||gradient_||^2 / ||J (gradient_/diagonal_)||^2
      const double q_gg = sums[6], gsq = sums[7], crs = sums[8];
      const double Jg2 = q_gg + 2.0 * crs + ls[2];
      const double gtot = gsq + ls[1];
      tr->alpha = gtot / Jg2;
      tr->grad_sq_total = gtot;
      tr->q[Q_GG] = q_gg;
      tr->q[Q_GRAD_SQ] = gsq;
    }
    if (tid == 0) solve_epilogue(S, tr, ls, gn2, ggn, gG, gN, qgn, qnn);
  }
}
// grid (1, batch) x 256, dynamic LDS = SOLVEB_LDS; two workgroups per CU
template <bool ASSEMBLE>
__global__ __launch_bounds__(SOLVE_THREADS, 2) void k_solve_block(char *base, size_t stride, long long xch_off, long long imu_off, long long prior_A_off) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  solveb_body<ASSEMBLE>(SLOT(base, stride), smem, xch_off, imu_off, prior_A_off);
}
