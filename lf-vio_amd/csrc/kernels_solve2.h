// kernels_solve2.h — k_solve_sparse: the reduced pose system solved along its block structure (solve_plan.h) instead of as
// one dense 172 x 172 Cholesky (k_solve_dense, kernels_solve.h — kept for windows whose prior does not have the structure
// the plan assumes, and as the A/B partner).  Same inputs, same outputs, same Ceres semantics (dogleg_strategy.cc
// ComputeGaussNewtonStep: Jacobi scaling, mu D^2 regularisation, LINEAR_SOLVER_FAILURE -> mu *= 10), another elimination
// order: nine of the eleven speed/bias blocks by cyclic reduction over the IMU chain — one wave per block, 9 pivots per
// round, three rounds — then the blocked MFMA Cholesky of kernels_solve.h on the remaining 91 unknowns (6 tiles of 16).
#pragma once
#include "kernels_solve.h"
#include "solve_plan.h"
#include <type_traits>

// Host-built table (lfvio_create): where each speed/bias entry of H_pp a thread loads goes in the solve storage.  Built
// with the functions of solve_plan.h, i.e. the ones the CPU check exercises; the update segments of the fronts are
// compile-time constants from the same header.
// The speed/bias entries of H_pp are loaded as "combos" of nine consecutive columns of one row — 36 per block (9 rows x
// [own block | previous block of the chain | poses f - 1 .. f + 1 in two halves]) and 81 for the camera-side row of sb_0
// (the prior couples it to everything): 477 combos, two per thread.  One row index per combo, consecutive columns: the
// vectors every entry is scaled with (S_p, G, N) are read once per row and in runs.
constexpr int S2_NCOMBO = 11 * 36 + 81;
constexpr int S2_SB_SLOTS = 18;  // entries per thread: 2 combos x 9
struct S2DevTables {
  int scatter[S2_SB_SLOTS * 256];  // [2 q + k ... ]: slot (combo q of the thread, entry k): dst | (mirror + 1) << 16, or -1: no entry
};
// row and first column of combo `cb`; false: the combo does not exist (sb_0 has no previous block ...)
PLAN_HD bool s2_combo(int cb, int *i, int *j0) {
  if (cb < 11 * 36) {
    const int f = cb / 36, rem = cb % 36, r = rem / 4, ch = rem % 4;
    *i = S2_KC + 9 * f + r;
    if (ch == 0) {
      *j0 = S2_KC + 9 * f;
      return true;
    }
    if (f == 0) return false;  // its camera-side row comes whole through the combos below
    *j0 = ch == 1 ? S2_KC + 9 * (f - 1) : 6 * (f - 1) + 9 * (ch - 2);
    return true;
  }
  if (cb >= S2_NCOMBO) return false;
  const int e = cb - 11 * 36;
  *i = S2_KC + e / 9, *j0 = 9 * (e % 9);
  return true;
}
// entry k of the combo: a structural non-zero of the lower triangle?
PLAN_HD bool s2_combo_entry(int cb, int k, int *i, int *j) {
  int j0 = 0;
  if (!s2_combo(cb, i, &j0)) return false;
  *j = j0 + k;
  if (cb < 11 * 36) {
    const int ch = cb % 4;
    if (ch == 0) return *j <= *i;
    if (ch == 1) return true;
    return *j < 66;
  }
  return *j < S2_KC;
}

constexpr int S2_THREADS = 256;
constexpr int S2_VEC = S2_STORE_LEN;                       // vectors behind the storage
constexpr size_t SOLVE2_LDS = (size_t)(S2_STORE_LEN + 7 * KP + 96 + 96 + 96 + 256 + 64 + 8) * sizeof(double);
#define S2_FI(n) std::integral_constant<int, n>{}
// the fine-grained stamps perturb what they measure (a stamp is a global store of thread 0, and the next barrier waits for
// it): only in -DLFVIO_SOLVE_PROFILE builds (tests/tools/solve_clocks.py)
#ifdef LFVIO_SOLVE_PROFILE
#define XSTAMP(S, k) STAMP(S, k)
#else
#define XSTAMP(S, k) do { } while (0)
#endif
// tasks of the update segments of a front (+ its rhs x camera row), and rounds of 256 threads they take
PLAN_HD int s2_seg_tasks(int fi) {
  const S2SegList L = s2_segment_list(fi);
  int n = s2_c1(fi) - s2_c0(fi);
  for (int k = 0; k < L.n; k++) n += L.s[k].rows * L.s[k].cols;
  return n;
}
constexpr int S2_MAX_SEG_ROUNDS = 4;

__global__ __launch_bounds__(S2_THREADS) void k_solve_sparse(char *base, size_t stride, long long xch_off, long long imu_off, const S2DevTables *T) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  Slot *S = SLOT(base, stride);
  TRState *tr = &S->tr;
  const int tid = threadIdx.x;
  const int er = tid >> 4, ek = tid & 15, esw = tsw(er, ek);
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  XSTAMP(S, 15);
  // ---- everything this kernel reads, requested in one round before the first branch
  double hc[15], sreg[15], hsb[2][9];
  int sdst[2][9];
  double z1 = 0.0, scross = 0.0, gval = 0.0, cp = 0.0;
  const double *xch = (const double *)((const char *)S + xch_off);
  const double *Hg = xch + XOFF_H, *Sg = xch + XOFF_S;
  // (explicit tile loops: with tile_a(t) / tile_b(t) in the body the unroller gives up, and a runtime index into the
  // register arrays is a waterfall of v_readlane — 18 000 cycles for fifteen entries)
#pragma unroll
  for (int a = 0; a < 5; a++)
#pragma unroll
    for (int b = 0; b <= a; b++) {
      const int t = a * (a + 1) / 2 + b, i = 16 * a + er, j = 16 * b + ek;
      hc[t] = (i < KC && j <= i) ? Hg[i * (i + 1) / 2 + j] : 0.0;
      sreg[t] = Sg[schur_index(min(i, j), max(i, j))];
    }
  // the speed/bias entries: two combos per thread (nine consecutive columns of one row each, see S2_NCOMBO)
  int ci[2] = {0, 0}, cj0[2] = {0, 0};
  bool cok[2];
#pragma unroll
  for (int q = 0; q < 2; q++) {
    cok[q] = s2_combo(tid + 256 * q, &ci[q], &cj0[q]);
    const double *row = Hg + ci[q] * (ci[q] + 1) / 2 + cj0[q];
#pragma unroll
    for (int k = 0; k < 9; k++) {
      sdst[q][k] = T->scatter[(9 * q + k) * 256 + tid];
      hsb[q][k] = cok[q] ? row[k] : 0.0;
    }
  }
  if (tid < KP) gval = xch[XOFF_G + tid];
  if (tid < KC) z1 = Sg[schur_index(tid, COL_B)], scross = Sg[schur_index(tid, COL_K)];
  const int sharded = S->sharded;
  const double *ls = sharded ? xch + XOFF_C : S->lm_sum;
  if (tid < 12) {
    if (tid == 0) cp = ls[0];
    else if (!sharded) cp = tid == 1 ? S->prior_g[KP] : ((const double *)((const char *)S + imu_off))[(size_t)(tid - 2) * IMU_OUT + 930];
  }
  const bool est_ex = S->est_ex != 0, est_td = S->est_td != 0;
  const TRFlags fl = tr_flags_decided(S);
  const int dec_pending = S->dec_pending;
  const double mu_decided = S->dec.mu;
  const double mu_header = tr->mu;
  double *st = smem;  // [remainder tiles | fronts]
  // zero the storage while the loads are in flight (entries the structure leaves empty, padding)
  for (int k = tid; k < S2_STORE_LEN; k += S2_THREADS) st[k] = 0.0;
#pragma unroll
  for (int t = 0; t < 15; t++) {
    SOLVE_KEEP(hc[t]);
    SOLVE_KEEP(sreg[t]);
  }
#pragma unroll
  for (int q = 0; q < 2; q++) {
#pragma unroll
    for (int k = 0; k < 9; k++) SOLVE_KEEP(hsb[q][k]);
  }
#pragma unroll
  for (int q = 0; q < 2; q++) {
#pragma unroll
    for (int k = 0; k < 9; k++) hsb[q][k] = sdst[q][k] >= 0 ? hsb[q][k] : 0.0;  // (a slot without an entry read its neighbour's)
  }
  SOLVE_KEEP(z1);
  SOLVE_KEEP(scross);
  SOLVE_KEEP(gval);
  SOLVE_KEEP(cp);
  SOLVE_KEEP(mu_header);
  if (threadIdx.x == 0 && dec_pending) {
    decision_to_header(tr, S->dec);
    S->dec_pending = 0;
  }
  if (fl.done | !fl.do_schur) return;
  double *g = st + S2_VEC;      // KP
  double *sc = g + KP;          // scale
  double *dg = sc + KP;         // diagonal_
  double *gr = dg + KP;         // gradient_
  double *Gd = gr + KP;         // unscaled Cauchy direction
  double *yv = Gd + KP;         // y (by tangent column), then the unscaled Gauss-Newton direction
  double *hv = yv + KP;         // diagonal of H_pp, then gauss_newton_step_
  double *invd = hv + KP;       // 96: 1 / L_ii of the remainder
  double *finv = invd + 96;     // 96: 1 / L_kk of the fronts (9 per front)
  double *yr = finv + 96;       // 96: rhs / solution of the remainder, by remainder index
  double *scratch = yr + 96;    // 256 (+ 64 pad)
  auto active = [&](int c) { return (est_ex || c < off_ex() || c >= off_ex() + 6) && (est_td || c != off_td()); };
  STAMP(S, 0);
  // ---- the diagonal of H_pp, g, cost pieces
#pragma unroll
  for (int a = 0; a < 5; a++)
    if (er == ek && 16 * a + er < KC) hv[16 * a + er] = hc[tile_id(a, a)];
#pragma unroll
  for (int q = 0; q < 2; q++) {
    const int cb = tid + 256 * q;
    if (cb < 11 * 36 && (cb & 3) == 0) {  // own-block combo of row r: its diagonal entry is entry r
      const int r = (cb % 36) >> 2;
#pragma unroll
      for (int k = 0; k < 9; k++)
        if (k == r) hv[ci[q]] = hsb[q][k];
    }
  }
  if (tid < KP) g[tid] = gval;
  if (tid < 12) scratch[tid] = cp;
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
  // ---- the reduced system  S_p (H_pp - Schur) S_p + mu D^2  and its right-hand side, each entry at its storage address;
  //      the Cauchy-point quadratic form G^T H G from the same entries on the way
  double qgg_part = 0;
  {
    double Gi[5], Gj[5], si[5], sj[5];
#pragma unroll
    for (int a = 0; a < 5; a++) {
      const int i = 16 * a + er, j = 16 * a + ek;
      Gi[a] = i < KC ? Gd[i] : 0.0, si[a] = i < KC ? sc[i] : 0.0;
      Gj[a] = j < KC ? Gd[j] : 0.0, sj[a] = j < KC ? sc[j] : 0.0;
    }
#pragma unroll
    for (int a = 0; a < 5; a++)
#pragma unroll
    for (int b = 0; b <= a; b++) {
      const int t = a * (a + 1) / 2 + b;
      const int i = 16 * a + er, j = 16 * b + ek;
      if (i < KC && j <= i) {
        double h = hc[t];
        qgg_part = fma(h * Gi[a], (i == j) ? Gj[b] : 2.0 * Gj[b], qgg_part);
        double v;
        if (active(i) && active(j)) {
          h -= sreg[t];
          v = si[a] * sj[b] * h;
          if (i == j) v += mu * dg[i] * dg[i];
        } else {
          v = (i == j) ? 1.0 : 0.0;
        }
        st[t * TSZ + esw] = v;  // camera columns are the first 73 remainder indices: tile (a, b) of H is tile (a, b) here
      }
    }
  }
#pragma unroll
  for (int q = 0; q < 2; q++)
    if (cok[q]) {
      const int i = ci[q];
      const double Gi = Gd[i], si = sc[i], mud = mu * dg[i] * dg[i];
      double Gj[9], sj[9];
#pragma unroll
      for (int k = 0; k < 9; k++) Gj[k] = Gd[cj0[q] + k], sj[k] = sc[cj0[q] + k];
#pragma unroll
      for (int k = 0; k < 9; k++) {
        const int d = sdst[q][k];
        if (d >= 0) {
          const int j = cj0[q] + k;
          const double h = hsb[q][k];
          qgg_part = fma(h * Gi, (i == j) ? Gj[k] : 2.0 * Gj[k], qgg_part);
          double v = 0.0;
          if (active(j)) v = si * sj[k] * h + ((i == j) ? mud : 0.0);  // (speed/bias columns are always active)
          st[d & 0xffff] = v;
          const int mirror = (d >> 16) - 1;
          if (mirror >= 0) st[mirror] = v;
        }
      }
    }
  if (tid < KP) {  // right-hand side:  S_p (g_p - z1)
    double r = 0.0;
    if (active(tid)) r = sc[tid] * (tid < KC ? g[tid] - z1 : g[tid]);
    st[s2_store(KP, tid, nullptr)] = r;
  }
  // ---- Cauchy point: alpha = ||gradient_||^2 / ||J (gradient_/diagonal_)||^2
  {
    double gs = 0, cross = 0;
    if (tid < KP) {
      gs = gr[tid] * gr[tid];
      if (tid < KC) cross = scross * Gd[tid];
    }
    double sums[3] = {qgg_part, gs, cross};
    block_sum_n(sums, scratch, tid);
    const double q_gg = sums[0], gsq = sums[1], cr = sums[2];
    if (tid == 0) {
      const double Jg2 = q_gg + 2.0 * cr + ls[2];
      const double gtot = gsq + ls[1];
      tr->alpha = gtot / Jg2;
      tr->grad_sq_total = gtot;
      tr->q[Q_GG] = q_gg;
      tr->q[Q_GRAD_SQ] = gsq;
    }
  }
  __syncthreads();
  STAMP(S, 3);
  bool bad = !(mu < 1.0);  // ComputeGaussNewtonStep: `while (mu_ < max_mu_)` — no attempt at mu >= 1

  // ---- speed/bias blocks by cyclic reduction.  A front is 9 rows x [own 9 | nbA 9 | nbB 9 | camera 73 | rhs]; one wave
  //      eliminates it row-wise with the rows in registers, two columns per lane: pivot and multipliers travel by
  //      v_readlane (column k of the own block sits in lane k).  Afterwards the rows hold [L^T | L^-1 A_fN | L^-1 b_f].
  auto factor_front = [&](int fi) {
    double *X = st + s2_front_base(fi);
    double r0[9], r1[9];
    const int l1 = lane + 64 < S2_LDX ? lane + 64 : S2_LDX - 1;  // (clamped: the loads stay unconditional)
#pragma unroll
    for (int i = 0; i < 9; i++) {
      r0[i] = X[i * S2_LDX + lane];
      const double v = X[i * S2_LDX + l1];
      r1[i] = lane + 64 < S2_LDX ? v : 0.0;
    }
    double rsv = 0.0;
#pragma unroll
    for (int k = 0; k < 9; k++) {
      const double d = readlane_f64(r0[k], k);
      if (!(d > 0.0)) bad = true;
      double m[9];
#pragma unroll
      for (int i = k + 1; i < 9; i++) m[i] = readlane_f64(r0[i], k);
      const double rs = fast_rsqrt(d);
      if (lane == k) rsv = rs;
      r0[k] *= rs, r1[k] *= rs;
#pragma unroll
      for (int i = k + 1; i < 9; i++) {
        const double li = m[i] * rs;
        r0[i] = fma(-li, r0[k], r0[i]);
        r1[i] = fma(-li, r1[k], r1[i]);
      }
    }
#pragma unroll
    for (int i = 0; i < 9; i++) {
      X[i * S2_LDX + lane] = r0[i];
      if (lane + 64 < S2_LDX) X[i * S2_LDX + 64 + lane] = r1[i];
    }
    if (lane < 9) finv[9 * fi + lane] = rsv;
  };
  // What a factored front takes out of the rest:  target -= sum_k X[k][p] X[k][q].
  // Only what the NEXT fronts need is on the critical path of the rounds — the updates of blocks that are fronts
  // themselves (solve_plan.h: segments of kind 0).  Everything that lands in the dense remainder (rows of sb_6 / sb_8, the
  // rhs row, camera x camera) waits until all nine fronts are factored and is then taken in ONE pass, every target summed
  // over the fronts that touch it by the thread (or wave) that owns it.
  // (1) critical segments: every thread first forms the sums of ALL its tasks (loads only), then subtracts them — a
  //     read-modify-write per task inside the loop would put every task's loads behind the previous task's store.
  auto seg_collect = [&](auto FI, double *acc, int *addr) {
    constexpr int fi = decltype(FI)::value;
    constexpr S2SegList L = s2_segment_list(fi);
    const double *X = st + s2_front_base(fi);
    int total = 0;
#pragma unroll
    for (int s = 0; s < L.n; s++)
      if (L.s[s].kind == 0) total += L.s[s].rows * L.s[s].cols;
#pragma unroll
    for (int it = 0; it < S2_MAX_SEG_ROUNDS; it++) {
      acc[it] = 0.0, addr[it] = -1;
      const int e = tid + S2_THREADS * it;
      if (e < total) {
        int p = -1, q = -1, a = -1, lo = 0;
#pragma unroll
        for (int s = 0; s < L.n; s++) {
          const S2Seg gsg = L.s[s];
          if (gsg.kind != 0) continue;  // compile-time
          const int n = gsg.rows * gsg.cols;
          if (e >= lo && e < lo + n) {
            const int le = e - lo, r = le / gsg.cols, c = le - r * gsg.cols;
            p = gsg.src_r + r, q = gsg.src_c + c;
            a = gsg.base + r * gsg.sr + c * gsg.sc;
          }
          lo += n;
        }
        {
          const double *xp = X + (a >= 0 ? p : 0), *xq = X + (a >= 0 ? q : 0);  // (unconditional loads)
          double v = 0.0;
#pragma unroll
          for (int k = 0; k < 9; k++) v = fma(xp[k * S2_LDX], xq[k * S2_LDX], v);
          acc[it] = v, addr[it] = a;
        }
      }
    }
  };
  auto seg_commit = [&](const double *acc, const int *addr) {
    double cur[S2_MAX_SEG_ROUNDS];
#pragma unroll
    for (int it = 0; it < S2_MAX_SEG_ROUNDS; it++) cur[it] = addr[it] >= 0 ? st[addr[it]] : 0.0;
#pragma unroll
    for (int it = 0; it < S2_MAX_SEG_ROUNDS; it++)
      if (addr[it] >= 0) st[addr[it]] = cur[it] - acc[it];
  };
  auto apply_two = [&](auto FA, auto FB) {  // two fronts with disjoint targets
    double accA[S2_MAX_SEG_ROUNDS], accB[S2_MAX_SEG_ROUNDS];
    int adA[S2_MAX_SEG_ROUNDS], adB[S2_MAX_SEG_ROUNDS];
    seg_collect(FA, accA, adA);
    seg_collect(FB, accB, adB);
    seg_commit(accA, adA);
    seg_commit(accB, adB);
  };
  auto apply_one = [&](auto FA) {
    double accA[S2_MAX_SEG_ROUNDS];
    int adA[S2_MAX_SEG_ROUNDS];
    seg_collect(FA, accA, adA);
    seg_commit(accA, adA);
  };
  // (2) deferred, rows of a speed/bias block that stays (rem = its remainder index): tasks (r, cc), cc < 9 the block itself
  //     (lower triangle), cc - 9 < 73 a camera column, cc = 82 the rhs; slot = where the block sits in the front's layout
  auto defer_rows = [&](int rem, double *acc, int *addr, auto... FS) {
#pragma unroll
    for (int it = 0; it < 3; it++) {
      const int e = tid + S2_THREADS * it;
      acc[it] = 0.0, addr[it] = -1;
      const int r = e / 83, cc = e - 83 * r;
      const bool live = e < 9 * 83 && !(cc < 9 && cc > r);
      const int rr = live ? r : 0, cq = live ? cc : 0;  // (clamped: unconditional loads, the sum is dropped)
      double sum = 0.0;
      auto contrib = [&](auto FS1) {
        constexpr int fi = decltype(FS1)::value >> 8, slot = decltype(FS1)::value & 255;
        constexpr int c0 = s2_c0(fi), c1 = s2_c1(fi);
        const double *X = st + s2_front_base(fi);
        const bool in = cq < 9 || cq == 82 || (cq - 9 >= c0 && cq - 9 < c1);
        const int q = cq < 9 ? slot + cq : cq < 82 ? (in ? S2_COL_CAM + cq - 9 : S2_COL_CAM + c0) : S2_COL_RHS;
        const double *xp = X + slot + rr, *xq = X + q;
        double v = 0.0;
#pragma unroll
        for (int k = 0; k < 9; k++) v = fma(xp[k * S2_LDX], xq[k * S2_LDX], v);
        sum += in ? v : 0.0;
      };
      (contrib(FS), ...);
      if (live) acc[it] = sum, addr[it] = cq < 9 ? s2_lidx(rem + rr, rem + cq) : cq < 82 ? s2_lidx(rem + rr, cq - 9) : s2_lidx(S2_NR, rem + rr);
    }
  };
#define S2_FS(fi, slot) std::integral_constant<int, ((fi) << 8) | (slot)>{}
  // (3) deferred, camera x camera on the matrix pipe:  C(a, b) -= sum over the fronts of X_a^T X_b, the 16 x 16 tiles dealt
  //     round-robin to the waves, three v_mfma_f64_16x16x4_f64 per front and tile (K = 9 rows of X, padded to 12); lane
  //     (c, g) holds X[4 s + g][16 a + c] — the operand of column tile a as A and of column tile b as B alike.  Row 91
  //     (rhs x camera) is the same sum with the rhs column as the left factor, one thread per camera column.
  auto cam_update_all = [&]() {
    const int c = lane & 15, gq = lane >> 4;
    int cnt = 0;
#pragma unroll
    for (int a = 0; a < 5; a++)
#pragma unroll
      for (int b = 0; b <= a; b++) {
        if ((cnt++ & 3) == wave) {
          double *Tc = st + tile_id(a, b) * TSZ + gq * TLD + c;
          solve_d4 cv;
#pragma unroll
          for (int r = 0; r < 4; r++) cv[r] = Tc[4 * TLD * r];
#pragma unroll
          for (int fi = 0; fi < S2_NF; fi++) {
            const int ta0 = s2_c0(fi) / 16, ta1 = (s2_c1(fi) - 1) / 16;
            if (a < ta0 || a > ta1 || b < ta0) continue;  // compile-time
            const double *X = st + s2_front_base(fi) + S2_COL_CAM;
            double xa[3], xb[3];
#pragma unroll
            for (int s3 = 0; s3 < 3; s3++) {
              // (load first, select afterwards: a load inside a conditional arm is a branch with its own wait, and six of
              // those per front and tile were most of this pass)
              const int k = 4 * s3 + gq, kk = k < 9 ? k : 8;
              const double va = X[kk * S2_LDX + 16 * a + c], vb = X[kk * S2_LDX + 16 * b + c];  // column 16 a + c <= 79: inside the row
              xa[s3] = (k < 9 && 16 * a + c < KC) ? va : 0.0;
              xb[s3] = (k < 9 && 16 * b + c < KC) ? vb : 0.0;
            }
#pragma unroll
            for (int s3 = 0; s3 < 3; s3++) cv = __builtin_amdgcn_mfma_f64_16x16x4f64(-xa[s3], xb[s3], cv, 0, 0, 0);
          }
#pragma unroll
          for (int r = 0; r < 4; r++) Tc[4 * TLD * r] = cv[r];
        }
      }
    if (tid < KC) {
      double acc = 0.0;
#pragma unroll
      for (int fi = 0; fi < S2_NF; fi++)
        if (tid >= s2_c0(fi) && tid < s2_c1(fi)) {
          const double *X = st + s2_front_base(fi);
          double v = 0.0;
#pragma unroll
          for (int k = 0; k < 9; k++) v = fma(X[k * S2_LDX + S2_COL_RHS], X[k * S2_LDX + S2_COL_CAM + tid], v);
          acc += v;
        }
      st[s2_lidx(S2_NR, tid)] -= acc;
    }
  };
  {
    // round 1: fronts 0..3 (sb 1 3 5 7), round 2: 4..6 (sb 9 0 4), round 3: 7, 8 (sb 2 10); between the rounds only the
    // updates of later FRONTS: {sb_1 -> sb_0, sb_2; sb_5 -> sb_4} then {sb_3 -> sb_2, sb_4}; {sb_9 -> sb_10; sb_0 -> sb_2} then
    // {sb_4 -> sb_2}
    factor_front(wave);
    __syncthreads();
    XSTAMP(S, 8);
    apply_two(S2_FI(0), S2_FI(2));
    __syncthreads();
    apply_one(S2_FI(1));
    __syncthreads();
    XSTAMP(S, 9);
    if (wave < 3) factor_front(4 + wave);
    __syncthreads();
    XSTAMP(S, 10);
    apply_two(S2_FI(4), S2_FI(5));
    __syncthreads();
    apply_one(S2_FI(6));
    __syncthreads();
    XSTAMP(S, 11);
    if (wave < 2) factor_front(7 + wave);
    __syncthreads();
    XSTAMP(S, 12);
    // what goes into the remainder: rows of sb_6 (from sb_5, sb_7, sb_4, sb_2), rows of sb_8 (from sb_7, sb_9, sb_10), their
    // coupling (from sb_7 alone: a segment of kind 1 of front 3), the rhs row and the camera tiles
    {
      double acc6[3], acc8[3], acc68 = 0.0;
      int ad6[3], ad8[3], ad68 = -1;
      defer_rows(S2_REM_SB6, acc6, ad6, S2_FS(2, S2_COL_B), S2_FS(3, S2_COL_A), S2_FS(6, S2_COL_B), S2_FS(7, S2_COL_A));
      defer_rows(S2_REM_SB8, acc8, ad8, S2_FS(3, S2_COL_B), S2_FS(4, S2_COL_A), S2_FS(8, S2_COL_A));
      {  // (sb_8, sb_6) from front 3 (sb_7): rows = sb_6 index (nbA), columns = sb_8 index (nbB)
        const int e = tid < 81 ? tid : 0, r = e / 9, c = e - 9 * r;
        const double *X = st + s2_front_base(3);
        double v = 0.0;
#pragma unroll
        for (int k = 0; k < 9; k++) v = fma(X[k * S2_LDX + S2_COL_A + r], X[k * S2_LDX + S2_COL_B + c], v);
        if (tid < 81) acc68 = v, ad68 = s2_lidx(S2_REM_SB8 + c, S2_REM_SB6 + r);
      }
      double cur[7];
#pragma unroll
      for (int it = 0; it < 3; it++) cur[it] = st[ad6[it] >= 0 ? ad6[it] : 0], cur[3 + it] = st[ad8[it] >= 0 ? ad8[it] : 0];
      cur[6] = st[ad68 >= 0 ? ad68 : 0];
#pragma unroll
      for (int it = 0; it < 3; it++) {
        if (ad6[it] >= 0) st[ad6[it]] = cur[it] - acc6[it];
        if (ad8[it] >= 0) st[ad8[it]] = cur[3 + it] - acc8[it];
      }
      if (ad68 >= 0) st[ad68] = cur[6] - acc68;
    }
    XSTAMP(S, 13);
    cam_update_all();
    __syncthreads();
  }
  STAMP(S, 4);

  // ---- the dense remainder [camera 73 | sb_6 | sb_8], rhs as row 91: blocked right-looking Cholesky on 6 tiles of 16 —
  //      the scheme of k_solve_dense (kernels_solve.h: in-wave factorization of the diagonal tile with the rank-4 panel
  //      update on the matrix pipe, one thread per row below, MFMA trailing update with look-ahead)
  constexpr int NT2 = S2_NT, NRR = S2_NR;
  double *Hs = st;
  auto factor = [&](int kb) {
    const int nb = kb < NT2 - 1 ? 16 : NRR - 16 * (NT2 - 1);  // pivots in this block column (11 in the last)
    double *Td = Hs + tile_id(kb, kb) * TSZ;
    const int c = lane & 15, gq = lane >> 4;
    solve_d4 a;
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const int col = gq + 4 * i;
      a[i] = col <= c ? Td[tsw(c, col)] : Td[tsw(col, c)];
    }
    double dsave[4] = {1.0, 1.0, 1.0, 1.0};
#pragma unroll
    for (int p = 0; p < 4; p++) {
      double bop = 0.0;
#pragma unroll
      for (int t = 0; t < 4; t++) {
        const int k = 4 * p + t;
        if (k < nb) {  // wave-uniform
          double colk = 0.0, u = 0.0;
          if (t < 3) {
            colk = quarter_bcast(a[p], t);
            u = row_bcast_k(a[p], k);
          }
          const double d = readlane_f64(a[p], 16 * t + k);
          if (!(d > 0.0)) bad = true;
          const double rc = fast_rcp(d);
          if (gq == t) dsave[p] = d, bop = a[p] * rc;
          if (t < 3) {
            const double upd = fma(c > k ? -(colk * rc) : 0.0, u, a[p]);
            if (gq > t) a[p] = upd;
          }
        }
      }
      if (p < 3 && 4 * p < nb) {
        solve_d4 cv = __builtin_amdgcn_mfma_f64_16x16x4f64(-a[p], bop, a, 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 4; i++)
          if (i > p) a[i] = cv[i];
      }
    }
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const int col = gq + 4 * i;
      const double rs = fast_rsqrt(dsave[i]);
      double v = 0.0;
      if (col < nb) v = c > col ? a[i] * rs : (c == col ? dsave[i] * rs : 0.0);
      Td[tsw(c, col)] = v;
      if (c == col && col < nb) invd[16 * kb + col] = rs;
    }
  };
  auto update = [&](int kb, int first, int step, int last) {
    const int c = lane & 15, gq = lane >> 4, offA = c * TLD + gq, offC = gq * TLD + c;
    int ti = kb + 1, tj = kb + 1, u = 0;
    auto advance = [&](int n) {
      for (int q = 0; q < n; q++, u++)
        if (++tj > ti) ti++, tj = kb + 1;
    };
    advance(first);
    while (u < last) {
      double av[4][4], bv[4][4];
      solve_d4 cv[4];
      double *Tc[4];
      int nt = 0;
#pragma unroll
      for (int t = 0; t < 4; t++) {
        Tc[t] = nullptr;
        if (u < last) {
          const double *Ta = Hs + tile_id(ti, kb) * TSZ + offA, *Tb = Hs + tile_id(tj, kb) * TSZ + offA;
          Tc[t] = Hs + tile_id(ti, tj) * TSZ + offC;
#pragma unroll
          for (int q = 0; q < 4; q++) av[t][q] = Ta[4 * q], bv[t][q] = Tb[4 * q], cv[t][q] = Tc[t][4 * TLD * q];
          nt = t + 1;
          advance(step);
        }
      }
#pragma unroll
      for (int t = 0; t < 4; t++)
        if (t < nt) {
#pragma unroll
          for (int q = 0; q < 4; q++) cv[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(-av[t][q], bv[t][q], cv[t], 0, 0, 0);
        }
#pragma unroll
      for (int t = 0; t < 4; t++)
        if (t < nt) {
#pragma unroll
          for (int r = 0; r < 4; r++) Tc[t][4 * TLD * r] = cv[t][r];
        }
    }
  };
  if (wave == 0) factor(0);
  __syncthreads();
  for (int kb = 0; kb < NT2 - 1; kb++) {
    {  // P: the rows below solve x L_kk^T = a, one thread per row
      const int ta = kb + 1 + (tid >> 4), r = tid & 15;
      if (ta < NT2) {
        double *Tp = Hs + tile_id(ta, kb) * TSZ;
        const double *Tk = Hs + tile_id(kb, kb) * TSZ;
        double x[16], dv[16];
#pragma unroll
        for (int j = 0; j < 16; j++) x[j] = Tp[tsw(r, j)], dv[j] = invd[16 * kb + j];
        double lc[16], ln[16];
#pragma unroll
        for (int t = 1; t < 16; t++) lc[t] = Tk[tsw(t, 0)];
#pragma unroll
        for (int j = 0; j < 16; j++) {
#pragma unroll
          for (int t = j + 2; t < 16; t++) ln[t] = Tk[tsw(t, j + 1)];
          x[j] *= dv[j];
#pragma unroll
          for (int t = j + 1; t < 16; t++) x[t] = fma(-x[j], lc[t], x[t]);
#pragma unroll
          for (int t = j + 2; t < 16; t++) lc[t] = ln[t];
        }
#pragma unroll
        for (int j = 0; j < 16; j++) Tp[tsw(r, j)] = x[j];
      }
    }
    __syncthreads();
    {  // U with look-ahead: wave 0 updates the next diagonal tile first and factors it while waves 1..3 update the rest
      const int m = NT2 - 1 - kb, ntiles = m * (m + 1) / 2;
      if (wave == 0) {
        update(kb, 0, 1, 1);
        factor(kb + 1);
      } else {
        update(kb, wave, 3, ntiles);
      }
    }
    __syncthreads();
  }
  STAMP(S, 5);
  {
    double f = bad ? 1.0 : 0.0;
    f = block_max(f, scratch, tid);
    bad = f > 0.0;
  }
  // ---- back-substitution of the remainder  L^T y = z  (the diagonal tiles inverted first, all at once: kernels_solve.h)
  if (tid < 96) yr[tid] = tid < NRR ? Hs[lidx(NRR, tid)] : 0.0;
  __syncthreads();
  {
    const int tsel = 4 * wave + (lane >> 4), c = lane & 15;
    if (tsel < NT2) {
      const int nb = tsel < NT2 - 1 ? 16 : NRR - 16 * (NT2 - 1);
      double *Tk = Hs + tile_id(tsel, tsel) * TSZ;
      double x[16], dv[16], lr[16], ln[16];
#pragma unroll
      for (int r = 0; r < 16; r++) dv[r] = invd[16 * tsel + (r < nb ? r : 0)];
      lr[0] = Tk[tsw(1, 0)];
      x[0] = (0 >= c && 0 < nb) ? dv[0] : 0.0;
#pragma unroll
      for (int r = 1; r < 16; r++) {
        if (r + 1 < 16) {
#pragma unroll
          for (int j = 0; j <= r; j++) ln[j] = Tk[tsw(r + 1, j)];
        }
        double acc = r == c ? 1.0 : 0.0, acc1 = 0.0;
#pragma unroll
        for (int j = 0; j + 1 < r; j += 2) acc = fma(-lr[j], x[j], acc), acc1 = fma(-lr[j + 1], x[j + 1], acc1);
        if (r & 1) acc = fma(-lr[r - 1], x[r - 1], acc);
        x[r] = (r >= c && r < nb) ? (acc + acc1) * dv[r] : 0.0;
        if (r + 1 < 16) {
#pragma unroll
          for (int j = 0; j <= r; j++) lr[j] = ln[j];
        }
      }
#pragma unroll
      for (int r = 0; r < 16; r++) Tk[tsw(r, c)] = x[r];
    }
  }
  __syncthreads();
  auto tile_solve = [&](int blk) {
    const int o = 16 * blk, nb = (NRR - o) < 16 ? (NRR - o) : 16, c = lane & 15;
    const double *Tk = Hs + tile_id(blk, blk) * TSZ;
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int r = 0; r < 16; r++) acc[r & 3] = fma(Tk[tsw(r, c)], r < nb ? yr[o + r] : 0.0, acc[r & 3]);
    if (c < nb) yr[o + c] = (acc[0] + acc[1]) + (acc[2] + acc[3]);
  };
  auto apply = [&](int blk, int j, int c) {
    const int o = 16 * blk, nb = (NRR - o) < 16 ? (NRR - o) : 16;
    const double *Tj = Hs + tile_id(blk, j) * TSZ;
    double acc[4] = {yr[16 * j + c], 0.0, 0.0, 0.0};
#pragma unroll
    for (int r = 0; r < 16; r++)
      if (r < nb) acc[r & 3] = fma(-Tj[tsw(r, c)], yr[o + r], acc[r & 3]);
    yr[16 * j + c] = (acc[0] + acc[1]) + (acc[2] + acc[3]);
  };
  if (wave == 0 && lane < 16) tile_solve(NT2 - 1);
  __syncthreads();
  for (int blk = NT2 - 1; blk >= 1; blk--) {
    if (wave == 0) {
      if (lane < 16) {
        apply(blk, blk - 1, lane);
        tile_solve(blk - 1);
      }
    } else {
      const int j = (tid - 64) >> 4;
      if (j < blk - 1) apply(blk, j, tid & 15);
    }
    __syncthreads();
  }
  XSTAMP(S, 14);
  // remainder index -> tangent column
  if (tid < NRR) yv[tid < KC ? tid : tid < S2_REM_SB8 ? off_sb(6) + (tid - S2_REM_SB6) : off_sb(8) + (tid - S2_REM_SB8)] = yr[tid];
  __syncthreads();
  // ---- the fronts, last round first:  L^T y_f = L^-1 b_f - X y_neighbours  (one wave per front; every LDS read of the
  //      substitution chain is requested before the chain starts)
  auto backsub_front = [&](int fi) {
    const double *X = st + s2_front_base(fi);
    const int f = s2_block(fi), A = s2_nb(fi, 0), B = s2_nb(fi, 1);
    auto ycol = [&](int col) -> double {
      if (col < S2_COL_A || col >= S2_COL_RHS) return 0.0;
      if (col < S2_COL_B) return yv[off_sb(A) + (col - S2_COL_A)];
      if (col < S2_COL_CAM) return B >= 0 ? yv[off_sb(B) + (col - S2_COL_B)] : 0.0;
      return yv[col - S2_COL_CAM];
    };
    const double y0 = ycol(lane), y1 = ycol(lane + 64);
    double x0[9], x1[9], lt[9][9], zr[9], iv[9];
#pragma unroll
    for (int k = 0; k < 9; k++) {
      x0[k] = X[k * S2_LDX + lane];
      {
        const double v = X[k * S2_LDX + (lane + 64 < S2_LDX ? lane + 64 : S2_LDX - 1)];
        x1[k] = lane + 64 < S2_LDX ? v : 0.0;
      }
      zr[k] = X[k * S2_LDX + S2_COL_RHS], iv[k] = finv[9 * fi + k];
#pragma unroll
      for (int j = k + 1; j < 9; j++) lt[k][j] = X[k * S2_LDX + j];
    }
    double t[9];
#pragma unroll
    for (int k = 0; k < 9; k++) t[k] = wave_sum(fma(x0[k], y0, x1[k] * y1));
    double ys[9];
#pragma unroll
    for (int k = 8; k >= 0; k--) {
      double a = zr[k] - t[k];
#pragma unroll
      for (int j = k + 1; j < 9; j++) a = fma(-lt[k][j], ys[j], a);
      ys[k] = a * iv[k];
    }
    if (lane == 0) {
#pragma unroll
      for (int k = 0; k < 9; k++) yv[off_sb(f) + k] = ys[k];
    }
  };
  if (wave < 2) backsub_front(7 + wave);
  __syncthreads();
  if (wave < 3) backsub_front(4 + wave);
  __syncthreads();
  backsub_front(wave);
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
  XSTAMP(S, 20);
  // ---- Gauss-Newton step, directions and pose-side quadratic forms
  if (tid < KP) {
    const double y = yv[tid];
    const double gn = -dg[tid] * y;
    S->gn_p[tid] = gn;
    const double Nd = -sc[tid] * y;
    yv[tid] = Nd;
    hv[tid] = gn;
    if (tid < KC) {
      S->uc_grad[tid] = Gd[tid];
      S->uc_gn[tid] = Nd;
    }
  }
  if (tid >= KC && tid < WLD) S->uc_grad[tid] = S->uc_gn[tid] = 0.0;
  __syncthreads();
  XSTAMP(S, 21);
  {
    // G^T H N and N^T H N from the entries of H_pp this thread has held in registers since the start
    double qgn = 0, qnn = 0;
    {
      double Gi[5], Ni[5], Gj[5], Nj[5];
#pragma unroll
      for (int a = 0; a < 5; a++) {
        const int i = 16 * a + er, j = 16 * a + ek;
        Gi[a] = i < KC ? Gd[i] : 0.0, Ni[a] = i < KC ? yv[i] : 0.0;
        Gj[a] = j < KC ? Gd[j] : 0.0, Nj[a] = j < KC ? yv[j] : 0.0;
      }
#pragma unroll
      for (int a = 0; a < 5; a++)
#pragma unroll
      for (int b = 0; b <= a; b++) {
        const int t = a * (a + 1) / 2 + b;
        const int i = 16 * a + er, j = 16 * b + ek;
        if (i < KC && j <= i) {
          const double h = hc[t];
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
    XSTAMP(S, 22);
#pragma unroll
    for (int q = 0; q < 2; q++)
      if (cok[q]) {
        const int i = ci[q];
        const double Gi = Gd[i], Ni = yv[i];
        double Gj[9], Nj[9];
#pragma unroll
        for (int k = 0; k < 9; k++) Gj[k] = Gd[cj0[q] + k], Nj[k] = yv[cj0[q] + k];
#pragma unroll
        for (int k = 0; k < 9; k++) {
          const double h = hsb[q][k], w = (i == cj0[q] + k) ? 0.5 : 1.0;  // (0 where the slot holds no entry)
          qgn = fma(h * w, Gi * Nj[k] + Ni * Gj[k], qgn);
          qnn = fma(h * w, 2.0 * Ni * Nj[k], qnn);
        }
      }
    XSTAMP(S, 23);
    double gn2 = 0, ggn = 0, gG = 0, gN = 0;
    if (tid < KP) {
      const double gn = hv[tid];
      gn2 = gn * gn;
      ggn = gr[tid] * gn;
      gG = g[tid] * Gd[tid];
      gN = g[tid] * yv[tid];
    }
    double sums[6] = {gn2, ggn, gG, gN, qgn, qnn};
    block_sum_n(sums, scratch, tid);
    gn2 = sums[0], ggn = sums[1], gG = sums[2], gN = sums[3], qgn = sums[4], qnn = sums[5];
    STAMP(S, 7);
    if (tid == 0) solve_epilogue(S, tr, ls, gn2, ggn, gG, gN, qgn, qnn);
  }
}
