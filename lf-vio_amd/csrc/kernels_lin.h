// kernels_lin.h — linearization sweep, MFMA Schur SYRK, partial reductions.
//
//   k_setup   : state reset, per-pair tables, IMU sqrt-information, prior A' = J0^T J0
//   k_lin     : ONE launch, four workgroup roles selected by blockIdx.x
//                 [landmark blocks (+ Schur SYRK of the block on v_mfma_f64_16x16x4_f64) | Gram chunks | IMU factors | prior]
//   k_sum     : deterministic fixed-order reduction of the per-workgroup partials
#pragma once
#include "dev_factors.h"
#include "kernels_spec.h"
#include "tr_decide.h"
#include <type_traits>

#define SLOT(base, stride) ((Slot *)((char *)(base) + (size_t)blockIdx.y * (stride)))

constexpr int MODE_SOLVE = 0;
constexpr int MODE_MARG = 1;        // MODE_MARG + flag: 1 = MARGIN_OLD, 2 = MARGIN_SECOND_NEW
// The synchronous entry points put gauge fix + marginalization into the same graph as the first chunk of solve passes.
// Those launches carry MODE_GATED and act on a slot only once its solve is done and its marginalization has not run yet
// (Slot::tail_state: 0 not run, 2 finished); they leave `done` set, so later solve passes skip the slot.  Nothing a
// gated kernel tests is written before the last of them (k_marg_solve) ends.
constexpr int MODE_GATED = 16;
constexpr int MODE_DECIDE = 64;   // the trust-region bookkeeping of the pass before rides in the prologue (see k_lin)
constexpr int MODE_NOCOUNT = 32;  // k_lin launched role by role: only the first of the launches counts the pass
DEV bool tail_gate(const Slot *S, int done) { return done && S->tail_state == 0; }
DEV bool is_marg(int mode) { return mode >= MODE_MARG; }
DEV const MargPlan *marg_plan(const Slot *S, int mode) { return &S->marg[mode - MODE_MARG]; }

DEV int gidx20(int p, int q) { return p * 20 - (p * (p - 1)) / 2 + (q - p); }  // upper index, p <= q

// ---------------------------------------------------------------------------
// k_setup: grid (SETUP_WGS + ceil(N / 256), batch) x 256:
//   [0] state, [1] IMU information roots, [2..17] prior normal matrix, [18..] inverse depths
// ---------------------------------------------------------------------------
constexpr int SETUP_PRIOR_WGS = 16;
constexpr int SETUP_WGS = 2 + SETUP_PRIOR_WGS;  // [0] state, [1] the ten IMU information roots, [2 ..) prior normal matrix
constexpr int SETUP_TILED_MAXN = 88;          // a prior of up to 88 rows is one workgroup's in the compact grid of a batch (bit 2 of k_setup's last argument)
__global__ __launch_bounds__(256, 4) void k_setup(char *base, size_t stride, int mode, int zero_wt) {  // (four waves per SIMD: the IMU role holds three 15-vectors per lane)
  Slot *S = SLOT(base, stride);
  const int tid = threadIdx.x;
  // one workspace for the three roles that need one (the state's table, the IMU factors' transposition tiles, the prior's slab of J0):
  // 19 KB, five workgroups of a resident batch per CU
  constexpr int SETUP_PRIOR_SLAB = 2048, SETUP_LDS = LFVIO_WINDOW_SIZE * 15 * 16;
  static_assert(SETUP_LDS >= SETUP_PRIOR_SLAB + LFVIO_MAX_PRIOR_DIM && SETUP_LDS >= 84 + TAB_SCRATCH, "k_setup's workspace");
  __shared__ double sh_setup[SETUP_LDS];
  const int setup_wgs = (zero_wt & 4) ? 3 : SETUP_WGS;  // compact grid of a resident batch: [0] state [1] IMU roots [2] prior, then the inverse depths
  if (blockIdx.x == 0) {
    // state + trust-region init (TrustRegionMinimizer::Init, DoglegStrategy ctor)
    const double *src = (const double *)&S->x0;
    double *dst = (double *)&S->x[0];
    for (int k = tid; k < (int)(sizeof(FrameState) / 8); k += 256) dst[k] = src[k];
    if (tid == 0) {
      TRState *t = &S->tr;
      t->radius = S->init_radius;
      t->function_tolerance = S->fn_tol;
      t->mu = 1e-8;
      t->x_cost = t->cand_cost = t->model_cost_change = t->dogleg_step_norm = t->alpha = 0.0;
      t->iteration = 0;
      t->cur = 0;
      t->do_lin = 1;
      t->do_schur = 1;
      t->done = 0;
      t->termination = LFVIO_NO_CONVERGENCE;
      t->chol_fail = 0;
      t->scaled = 0;
      t->num_succ = t->num_unsucc = t->consec_invalid = t->trace_len = 0;
      t->step_valid = 0;
      t->skip_step = 0;
      t->error = 0;
      t->new_point = 0;
      t->spec_n = 1;
      S->tail_state = 0;
      S->passes_used = 0;
      S->dec_pending = 0;
      if (mode >= MODE_MARG) t->mu = 0.0;
    }
    double *bt = sh_setup;
    if (tid < 77) bt[tid] = (&S->x0.pose[0][0])[tid];
    else if (tid < 84) bt[tid] = S->x0.ex[tid - 77];
    __syncthreads();
    const unsigned offm = build_tab(bt, &S->tab[0], tid, bt + 84);
    if (tid == 0) S->ex_fixed_off = ((offm >> TAB_EX_BIT) & 1u) && !S->est_ex;  // (the candidates' tables inherit it: build_tab<false>)
    if (tid == 0 && S->spec_on) spec_arm(S);
  } else if (blockIdx.x == 1) {
    // sqrt_info = LLT(cov^-1).matrixL()^T (imu_factor.h:64), hoisted out of the iteration loop: the reference recomputes it in every
    // Evaluate().  All ten factors of the window in this workgroup, SIXTEEN LANES PER FACTOR (round 6; a workgroup of 256 per factor ran
    // the elimination with one useful lane in a hundred: 92 k wave-instructions per window).  Gauss-Jordan with partial pivoting on
    // [cov | I], 15 x 30, a lane holding COLUMNS l and l + 16 in registers: the pivot search of step k is lane k's own column, pivot and
    // multiplier column reach the other lanes of the factor by DPP row broadcasts (the step is unrolled: the source lane is an
    // immediate), the row swap is a select per register.  The arithmetic per entry is what it was — mk = M[p][c] / piv, entry =
    // M[sr][c] - M[sr][k] mk on the swapped rows — and so are the bits.  Then the column Cholesky of the inverse, a lane per row.
    if (tid >= 16 * ((LFVIO_WINDOW_SIZE + 3) & ~3)) return;  // (the fourth wave holds no factor: it would issue the whole elimination for nothing; the barrier below counts the waves that are left)
    const int grp = tid >> 4, l = tid & 15;
    const int f = grp < LFVIO_WINDOW_SIZE ? grp : 0;  // (lanes beyond the tenth factor run along on the first one's numbers and store nothing)
    const bool mine = grp < LFVIO_WINDOW_SIZE && S->imu_active[f];
    const double *cov = S->imu[f].covariance;
    double m0[15], m1[15];
#pragma unroll
    for (int r = 0; r < 15; r++) {
      m0[r] = l < 15 ? cov[r * 15 + l] : (r == 0 ? 1.0 : 0.0);  // column l of the covariance; lane 15: column 15 = the identity's first
      m1[r] = r == l + 1 ? 1.0 : 0.0;                            // column 16 + l: the identity's column l + 1 (lanes 14, 15: nothing)
    }
#pragma unroll
    for (int k = 0; k < 15; k++) {
      // lane k's search of its column k (a chain of fifteen compares: a tournament over arrays of candidates was tried and spilled —
      // 34 us against 14 for the kernel)
      int p = k;
      double best = -1.0, piv = 0.0;
#pragma unroll
      for (int r = 0; r < 15; r++) {
        const double v = fabs(m0[r]);
        if (r >= k && v > best) best = v, p = r, piv = m0[r];
      }
      p = row_bcast_ik(p, k), piv = row_bcast_k(piv, k);
      double csr[15];                                      // M[src(r)][k]: column k with rows k and p exchanged
#pragma unroll
      for (int r = 0; r < 15; r++) csr[r] = row_bcast_k(m0[r], k);
      const double ck = csr[k];
#pragma unroll
      for (int r = 0; r < 15; r++) csr[r] = r == p ? ck : csr[r];
      auto step = [&](double(&m)[15]) {
        double Mp = m[0];
#pragma unroll
        for (int r = 1; r < 15; r++) Mp = p == r ? m[r] : Mp;
        const double Mk = m[k], mk = Mp / piv;
#pragma unroll
        for (int r = 0; r < 15; r++) {
          const double Msr = r == p ? Mk : m[r];
          m[r] = r == k ? mk : Msr - csr[r] * mk;
        }
      };
      step(m0), step(m1);
    }
    // the inverse is columns 15 .. 29: column 15 + j sits in lane 15 (j = 0) or lane j - 1 (its second column); through LDS it becomes
    // a row per lane, which is what the column Cholesky (and Eigen's LLT: the lower triangle) reads
    double *T = sh_setup + f * (15 * 16);  // (the spare lanes write the first factor's numbers once more)
#pragma unroll
    for (int r = 0; r < 15; r++) {
      if (l == 15) T[r * 16] = m0[r];
      else if (l < 14) T[r * 16 + l + 1] = m1[r];
    }
    __syncthreads();
    // Column Cholesky, lane i: row i.  Same operations in the same order: t = m_ij - l_i0 l_j0 - l_i1 l_j1 ..., d = sqrt(t_jj), l_ij = t / d.
    const int li = l < 15 ? l : 0;
    double lrow[15], mrow[15];
#pragma unroll
    for (int j = 0; j < 15; j++) mrow[j] = T[li * 16 + j], lrow[j] = 0.0;
    bool ok = true;
#pragma unroll
    for (int j = 0; j < 15; j++) {
      double t = mrow[j];
#pragma unroll
      for (int k = 0; k < j; k++) t -= lrow[k] * row_bcast_k(lrow[k], j);
      const double sj = row_bcast_k(t, j);
      if (!(sj > 0.0)) ok = false;
      const double d = sqrt(sj);
      lrow[j] = l == j ? d : t / d;
    }
    if (mine && l < 15) {
#pragma unroll
      for (int i = 0; i < 15; i++) S->imu_sqrt[f][i * 15 + l] = (l >= i && ok) ? lrow[i] : 0.0;  // sqrt_info[i][j] = L[j][i], j >= i
    }
    if (mine && !ok && l == 0) S->imu_active[f] = 0;
  } else if (blockIdx.x >= setup_wgs) {
    // inverse depths of the window: 256 landmarks per workgroup
    const int l = (blockIdx.x - setup_wgs) * 256 + tid;
    if (l < S->N) S->lam[0][l] = S->lam0[l];
    if ((zero_wt & 1) && !S->wt_clean) {
      // k_linw writes the transposed rows (Slot::Wt) over the landmarks' own spans only: what lies outside is zero from here
      // on (the spans do not change while the window is resident: k_linw marks the copy clean behind its first sweep, the next
      // upload of the slot clears the mark — 189 KB per window that a batch re-solved where it lies does not write again)
      double2 *wt = (double2 *)(double *)S->Wt;
      const int nw = gridDim.x - setup_wgs, w = blockIdx.x - setup_wgs;
      for (int e = w * 256 + tid; e < WT_PAIRS * SPEC_MAX_LM; e += nw * 256) wt[e] = make_double2(0.0, 0.0);
    }
  } else {
    // prior: A' = J0^T J0, b0 = J0^T r0 (constant over the solve), entries spread over SETUP_PRIOR_WGS workgroups
    if (!S->prior_valid) return;
    const int n = S->prior_n;
    const int part = blockIdx.x - 2;
    // (a resident batch: as few of the SETUP_PRIOR_WGS workgroups as the entries need at PRIOR_EPT per thread — three for the usual
    // n = 76 —, because every one of them stages all of J0 and the batch pays that for every window; few windows: all of them,
    // 1.4 entries per thread — the latency of the call)
    constexpr int PRIOR_EPT = 8;
    const int nb4 = (n + 3) >> 2, tiles = nb4 * (nb4 + 1) / 2;  // 4 x 4 blocks of the upper triangle of A': 190 for n = 76
    static_assert(((SETUP_TILED_MAXN + 3) / 4) * ((SETUP_TILED_MAXN + 3) / 4 + 1) / 2 <= 256, "a thread per 4 x 4 block of the upper triangle");
    if (zero_wt & 4) {  // (the host has looked at every resident prior: at most SETUP_TILED_MAXN rows, setup_launch)
      // A resident batch: ONE workgroup per window, a thread per 4 x 4 block of the upper triangle (and its mirror image): eight LDS
      // reads per sixteen products instead of two per product — the entry-per-thread form below is bound by exactly that traffic
      // (7.5 MB of LDS reads per window, three workgroups staging all of J0 each).  Every entry is the same chain of fma over the
      // rows in ascending order as below, and (c, r) the same products as (r, c): same bits.
      if (part != 0) return;
      constexpr int PRIOR_SLAB = SETUP_PRIOR_SLAB;
      double *Js = sh_setup;
      const double *J = S->prior_J;
      const int rows_per = PRIOR_SLAB / n;
      int br = 0, rem = tid < tiles ? tid : 0;
      while (rem >= nb4 - br) rem -= nb4 - br, br++;
      const int r0 = 4 * br, c0 = 4 * (br + rem);
      double acc[4][4], accb = 0;
#pragma unroll
      for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) acc[i][j] = 0.0;
      for (int k0 = 0; k0 < n; k0 += rows_per) {
        const int nk = min(rows_per, n - k0);
        __syncthreads();
        {
          double v[8];
          const double *Jk = J + k0 * n;
#pragma unroll
          for (int q = 0; q < 8; q++) v[q] = Jk[tid + 256 * q < nk * n ? tid + 256 * q : 0];
          const double rv = S->prior_r[k0 + (tid < nk ? tid : 0)];
#pragma unroll
          for (int q = 0; q < 8; q++)
            if (tid + 256 * q < nk * n) Js[tid + 256 * q] = v[q];
          if (tid < nk) Js[PRIOR_SLAB + tid] = rv;
        }
        __syncthreads();
        if (tid < tiles) {
          // (a row's last block may reach up to three entries past the row: the next row's numbers — or, behind the slab, r0's —, which
          // are multiplied and never stored)
#pragma unroll 2
          for (int k = 0; k < nk; k++) {
            double a[4], b[4];
#pragma unroll
            for (int i = 0; i < 4; i++) a[i] = Js[k * n + r0 + i], b[i] = Js[k * n + c0 + i];
#pragma unroll
            for (int i = 0; i < 4; i++)
#pragma unroll
              for (int j = 0; j < 4; j++) acc[i][j] = fma(a[i], b[j], acc[i][j]);
          }
        }
        if (tid < n)
          for (int k = 0; k < nk; k++) accb = fma(Js[k * n + tid], Js[PRIOR_SLAB + k], accb);
      }
      if (tid < tiles) {
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
          for (int j = 0; j < 4; j++) {
            const int r = r0 + i, c = c0 + j;
            if (r < n && c < n) {
              S->prior_A[r * n + c] = acc[i][j];
              if (c0 != r0) S->prior_A[c * n + r] = acc[i][j];
            }
          }
      }
      if (tid < n) S->prior_b0[tid] = accb;
      return;
    }
    const int np = gridDim.y >= 8 ? min(SETUP_PRIOR_WGS, (n * n + 256 * PRIOR_EPT - 1) / (256 * PRIOR_EPT)) : SETUP_PRIOR_WGS;  // (a batch with a prior beyond SETUP_TILED_MAXN rows)
    if (part >= np) return;
    // J0 goes through LDS in slabs of rows (all of it for the usual n = 76): one batch of independent loads instead of
    // a dependent load per term.  A thread owns up to PRIOR_EPT entries of A' (n <= 172: 29 584 entries over 4 096 threads).
    constexpr int PRIOR_SLAB = SETUP_PRIOR_SLAB;  // 16 KB
    double *Js = sh_setup;
    const double *J = S->prior_J;
    const int rows_per = PRIOR_SLAB / n;
    double acc[PRIOR_EPT], accb = 0;
#pragma unroll
    for (int q = 0; q < PRIOR_EPT; q++) acc[q] = 0;
    for (int k0 = 0; k0 < n; k0 += rows_per) {
      const int nk = min(rows_per, n - k0);
      __syncthreads();
      {  // the slab in ONE round of loads (a loop of load - wait - store is a memory round trip per element: eight per slab)
        static_assert(PRIOR_SLAB <= 8 * 256, "eight elements of a slab per thread");
        double v[8];
        const double *Jk = J + k0 * n;
#pragma unroll
        for (int q = 0; q < 8; q++) v[q] = Jk[tid + 256 * q < nk * n ? tid + 256 * q : 0];
        const double rv = S->prior_r[k0 + (tid < nk ? tid : 0)];
#pragma unroll
        for (int q = 0; q < 8; q++)
          if (tid + 256 * q < nk * n) Js[tid + 256 * q] = v[q];
        if (tid < nk) Js[PRIOR_SLAB + tid] = rv;  // (nk <= PRIOR_SLAB / n < 256 rows)
      }
      __syncthreads();
#pragma unroll
      for (int q = 0; q < PRIOR_EPT; q++) {
        const int e = part * 256 + tid + q * 256 * np;
        if (e < n * n) {
          const int r = e / n, c = e % n;
          double s = acc[q];
          for (int k = 0; k < nk; k++) s = fma(Js[k * n + r], Js[k * n + c], s);
          acc[q] = s;
        }
      }
      if (part == 0 && tid < n)
        for (int k = 0; k < nk; k++) accb = fma(Js[k * n + tid], Js[PRIOR_SLAB + k], accb);
    }
#pragma unroll
    for (int q = 0; q < PRIOR_EPT; q++) {
      const int e = part * 256 + tid + q * 256 * np;
      if (e < n * n) S->prior_A[e] = acc[q];
    }
    if (part == 0 && tid < n) S->prior_b0[tid] = accb;
  }
}

// ---------------------------------------------------------------------------
// k_lin: grid (nLmBlocks + ceil(nChunks / 4) + 10 + 1, batch) x 256
// ---------------------------------------------------------------------------
DEV void load_pair_uniform(const Tab *T, int pair, PairU &u) {
  u.M2 = ldm(T->M2[pair]);
  u.T = ldm(T->T[pair]);
  u.ric = ldm(T->ric);
  u.ricT = ldm(T->ricT);
  u.c = ld3(T->c[pair]);
  u.tic = ld3(T->tic);
  u.offc = u.offr = nullptr;
}
// the same for a table with a quaternion off the unit sphere (offm = tab_offmask(T) != 0, read once per workgroup)
DEV void load_pair_uniform(const Tab *T, int pair, unsigned offm, PairU &u) {
  load_pair_uniform(T, pair, u);
  if (offm && pair_is_off(offm, pair % 11)) u.offc = T->c[tab_cj(pair)], u.offr = T->T[0];
}

typedef double double4_t __attribute__((ext_vector_type(4)));

constexpr int LIN_THREADS = 256;
constexpr int LIN_LDS_LM = LM_BLOCK * (WLD + 1) + 64 + 2 * LM_BLOCK;  // doubles: W tile of the landmark role, reduction scratch, Schur weights
constexpr int LIN_PRIOR_N = 76;  // a prior of up to 76 rows (the reference's own: 10 poses, one speed/bias, extrinsic, td) is staged in LDS by the prior role
constexpr int LIN_LDS_PRIOR = LIN_PRIOR_N * LIN_PRIOR_N + 4 * KP;
constexpr int LIN_LDS = LIN_LDS_LM > LIN_LDS_PRIOR ? LIN_LDS_LM : LIN_LDS_PRIOR;

DEV double quad_sum(double v) {  // sum over the 4 lanes of a quad, same value and same order in every lane
  v += dpp_f64<0xB1>(v);  // quad_perm [1,0,3,2]  (DPP moves: a __shfl_xor is a ds_bpermute round trip per dword)
  v += dpp_f64<0x4E>(v);  // quad_perm [2,3,0,1]
  return v;
}
DEV d3 quad_sum3(d3 v) { return mk3(quad_sum(v.x), quad_sum(v.y), quad_sum(v.z)); }
// the lanes of one track: 4 (a quad) or 8
template <int LPT>
DEV double track_sum(double v) { return LPT == 8 ? sum8(v) : quad_sum(v); }
template <int LPT>
DEV d3 track_sum3(d3 v) { return mk3(track_sum<LPT>(v.x), track_sum<LPT>(v.y), track_sum<LPT>(v.z)); }

// Schur SYRK of one landmark block from the LDS tile: part `blk` of the 15 upper 16x16 tiles of sum_l c_l w_l w_l^T
// (cols 73/74 carry b_l and the Cauchy cross-term column) on the FP64 matrix pipe.  Wave w owns tiles w, w+4, w+8,
// w+12; four landmarks per MFMA: lane (k, c) feeds A[i = c][k] = c_l w_l[16 t + c], B[k][j = c] = w_l[16 u + c]; the
// accumulator holds D[row = (lane >> 4) + 4 reg][col = lane & 15].
// TIGHT: the form for launches whose landmark role is a latency chain (k_lin with all roles in one grid: single windows);
// a resident batch (the landmark role as a launch of its own, two waves per SIMD) measured 0.4 % better with the plain loop.
template <bool TIGHT>
DEV void schur_block(int lm0, int lmb, int mode, int Nlim, double (*tile)[WLD + 1], const double *lcoef, const double *le, double *part) {
  const int tid = threadIdx.x;
  const int wv = tid >> 6, lane = tid & 63, kk = lane >> 4, cc = lane & 15;
  double4_t acc[4];
  int ct[4], cu[4];
  bool scale_k[4];
#pragma unroll
  for (int j = 0; j < 4; j++) {
    acc[j] = double4_t{0, 0, 0, 0};
    const int ti = wv + 4 * j;  // upper tile index: (0,0..4) (1,1..4) (2,2..4) (3,3..4) (4,4)
    const int t = ti < 5 ? 0 : ti < 9 ? 1 : ti < 12 ? 2 : ti < 14 ? 3 : 4;
    const int u = ti - (t * 5 - (t * (t - 1)) / 2) + t;
    ct[j] = ti < NT ? 16 * t + cc : 0, cu[j] = ti < NT ? 16 * u + cc : 0;
    scale_k[j] = mode == MODE_SOLVE && cu[j] == COL_K;  // b/D2 * e  -> z2 column
  }
  int rows = Nlim - lm0;  // (lm0: the workgroup's first landmark; lmb: how many it has at most — 64, or 32 in the 8-lanes-per-track form)
  rows = rows > lmb ? lmb : rows;
  // No branch inside the loop: a tile skipped there (the sixteenth slot, wave 3) was two branches of the wave per tile and
  // step — 670 cycles per step, now 550 — so wave 3 runs its own loop over three tiles.  (Reading the operands of step s + 1
  // while the matrix pipe works on step s, kept from being undone with opaque copies, measured no better.)  Rows past the
  // block's last landmark are zeros in the tile and carry weight zero.
  auto sweep = [&](auto ntl) {
    constexpr int NTL = decltype(ntl)::value;
    for (int s4 = 0; 4 * s4 < rows; s4++) {
      const int row = 4 * s4 + kk;
      const double coef = lcoef[row], eb = le[row];
#pragma unroll
      for (int j = 0; j < NTL; j++)
        acc[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(coef * tile[row][ct[j]], tile[row][cu[j]] * (scale_k[j] ? eb : 1.0), acc[j], 0, 0, 0);
    }
  };
  if (TIGHT) {
    if (__builtin_amdgcn_readfirstlane(wv) + 12 < NT) sweep(std::integral_constant<int, 4>{});
    else sweep(std::integral_constant<int, 3>{});
  } else {
    for (int s4 = 0; 4 * s4 < rows; s4++) {
      const int row = 4 * s4 + kk;
      const double coef = lcoef[row], eb = le[row];
#pragma unroll
      for (int j = 0; j < 4; j++) {
        if (wv + 4 * j >= NT) continue;  // wave-uniform
        const double xa = tile[row][ct[j]];
        double xb = tile[row][cu[j]];
        if (mode == MODE_SOLVE && cu[j] == COL_K) xb *= eb;  // b/D2 * e  -> z2 column
        acc[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(coef * xa, xb, acc[j], 0, 0, 0);
      }
    }
  }
#pragma unroll
  for (int j = 0; j < 4; j++) {
    if (wv + 4 * j >= NT) continue;
#pragma unroll
    for (int r = 0; r < 4; r++) part[(wv + 4 * j) * 256 + r * 64 + lane] = acc[j][r];
  }
}
// tangent column of entry ci of a stored W row (see w_row_len)
DEV int w_col(int ci, int start, int cnt) { return ci < 6 * cnt ? 6 * start + ci : 66 + (ci - 6 * cnt); }

// The state k_lin linearizes at.  Normally the header's: x / tab / lam [cur] and mu.  In a MODE_DECIDE pass it is what the
// bookkeeping in the prologue decided, which has not reached the header yet — and when the accepted step is a speculative
// candidate, its *E slots (one workgroup copies them into the [cur] slots for the kernels that follow, meanwhile).
struct LinView {
  const FrameState *x;
  const Tab *tab;
  const double *lam;
  double mu;
};

// A solve repeated with a new mu on an unchanged linearization (do_schur without do_lin): only the Schur weights
// change.  The block's W rows come back from HBM into the tile and the SYRK is redone.
template <int LPT>
DEV void lin_schur_only_role(Slot *S, const LinView &lv, int wg, double *lds, double *part) {
  constexpr int LMB = LIN_THREADS / LPT;  // landmarks of the workgroup
  double(*tile)[WLD + 1] = (double(*)[WLD + 1]) lds;
  double *lcoef = lds + LM_BLOCK * (WLD + 1) + 64, *le = lcoef + LM_BLOCK;
  const int tid = threadIdx.x;
  for (int e = tid; e < LM_BLOCK * (WLD + 1); e += LIN_THREADS) lds[e] = 0.0;
  __syncthreads();
  {  // the stored rows back into the 80-wide tile: LPT lanes per landmark
    const int lml = tid / LPT, q = tid % LPT, l = wg * LMB + lml;
    if (l < S->N) {
      const int st = S->lm_start[l], cnt = S->lm_cnt[l];
      const double *w = S->W + S->lm_woff[l];
      for (int ci = q; ci < w_row_len(cnt); ci += LPT) tile[lml][w_col(ci, st, cnt)] = w[ci];
    }
  }
  if (tid < LM_BLOCK) {
    const int l = wg * LMB + tid;
    double cf = 0.0, eb = 0.0;
    if (tid < LMB && l < S->N) {
      const double sc = S->scale_l[l], s2a = sc * sc * S->a[l];
      const double D2 = fmin(fmax(s2a, 1e-6), 1e32);
      eb = s2a + lv.mu * D2;  // e-block + lm_diagonal^2
      const double einv = 1.0 / eb;
      cf = sc * sc * einv;
      S->einv_l[l] = einv;
    }
    lcoef[tid] = cf, le[tid] = eb;
  }
  __syncthreads();
  schur_block<false>(wg * LMB, LMB, MODE_SOLVE, S->N, tile, lcoef, le, part);
}

// Landmark role: 64 landmarks per workgroup, 4 lanes per landmark (lane q takes the observations 1+q, 5+q, 9+q of the
// track), so the dependent chain per lane is a quarter of the track.  The 80-wide row w_l is built in an LDS tile and
// leaves as whole 512-byte lines.
// phase stamps of workgroup 0 (tests/tools/lin_clocks.py)
#ifdef LFVIO_LIN_PROFILE
#define LSTAMP(k) do { if (blockIdx.x == 0 && mode == MODE_SOLVE && threadIdx.x == 0) S->dbg[k] = (long long)__builtin_readcyclecounter(); } while (0)
#else
#define LSTAMP(k) do { } while (0)
#endif
// LPT: lanes per track.  4: 64 landmarks per workgroup (every window).  8 (round 5; windows of at most SPEC_MAX_LM landmarks,
// Slot::lm_half): 32 landmarks per workgroup, twice the workgroups — a lane takes the observations 1 + q, 9 + q of its track
// instead of 1 + q, 5 + q, 9 + q, and the workgroup's Schur SYRK is 8 steps instead of 16: the role is the latency chain of a single
// window's k_lin (36 k cycles of which the observations 14 k and the SYRK 8 k), and a 300-landmark window has 246 CUs to spare.
// OFFS: see k_lin.
template <bool TIGHT, int LPT, bool OFFS>
DEV void lin_landmark_role(Slot *S, const LinView &lv, int wg, int mode, double *lds, double *part) {
  constexpr int LMB = LIN_THREADS / LPT;  // landmarks of the workgroup
  double(*tile)[WLD + 1] = (double(*)[WLD + 1]) lds;
  double *red = lds + LM_BLOCK * (WLD + 1);
  double *lcoef = red + 64, *le = lcoef + LM_BLOCK;  // Schur weight c_l and e-block of the block's landmarks
  const int tid = threadIdx.x, lml = tid / LPT, q = tid % LPT;
  const TRState *tr = &S->tr;
  const Tab *T = lv.tab;
  const int Nlim = is_marg(mode) ? marg_plan(S, mode)->N0 : S->N;
  const int l = wg * LMB + lml;
  const bool valid = l < Nlim;
  const int est_td = S->est_td;
  const int est_ex = is_marg(mode) ? 1 : S->est_ex;  // ResidualBlockInfo::Evaluate asks for every Jacobian
  const double td = lv.x->td;
  LSTAMP(22);
  for (int e = tid; e < LM_BLOCK * (WLD + 1); e += LIN_THREADS) lds[e] = 0.0;
  if (tid < 2 * LM_BLOCK) lcoef[tid] = 0.0;
  __syncthreads();
  LSTAMP(8);
  double a = 0, b = 0, cost = 0, lam = 1.0, wtd = 0;
  d3 wPi = mk3(0, 0, 0), wTi = wPi, wTic = wPi, wTx = wPi;
  int i = 0, cnt_l = 0, woff_l = 0;
  if (valid) {
    i = S->lm_start[l];
    const int k = S->lm_cnt[l], o0 = S->lm_obs0[l];
    cnt_l = k, woff_l = S->lm_woff[l];
    lam = lv.lam[l];
    ObsPair ob;
    load_obs(S, o0, ob.pi, ob.vi, ob.tdi, ob.rowi);
    const m33 ricT = ldm(T->ricT);
    const unsigned offm = OFFS ? tab_offmask(T) : 0u;
    for (int o = 1 + q; o < k; o += LPT) {
      const int j = i + o, pair = i * 11 + j;
      load_obs(S, o0 + o, ob.pj, ob.vj, ob.tdj, ob.rowj);
      PairU u;
      if (OFFS) load_pair_uniform(T, pair, offm, u);
      else load_pair_uniform(T, pair, u);
      Basis B;
      visual_basis<OFFS>(ob, lam, td, est_td, S->tr_over_row, S->half_row, S->sqrt_info, u, B);
      const double jl0 = B.jl[0], jl1 = B.jl[1];
      d3 eR = jl0 * B.red[0] + jl1 * B.red[1];
      const m33 M1 = ldm(T->M1[j]);
      d3 wp = vmul(eR, M1);  // M1^T eR
      wPi = wPi + wp;
      wTi = wTi + (jl0 * B.jti[0] + jl1 * B.jti[1]);
      d3 wtj = jl0 * B.jtj[0] + jl1 * B.jtj[1];
      tile[lml][6 * j + 0] = -wp.x, tile[lml][6 * j + 1] = -wp.y, tile[lml][6 * j + 2] = -wp.z;
      tile[lml][6 * j + 3] = wtj.x, tile[lml][6 * j + 4] = wtj.y, tile[lml][6 * j + 5] = wtj.z;
      m33 M3 = u.M2;
#pragma unroll
      for (int e = 0; e < 9; e++) M3.a[e] -= ricT.a[e];
      wTic = wTic + vmul(eR, M3);
      wTx = wTx + (jl0 * B.jtx[0] + jl1 * B.jtx[1]);
      wtd += jl0 * B.jtd[0] + jl1 * B.jtd[1];
      a += jl0 * jl0 + jl1 * jl1;
      b += jl0 * B.r[0] + jl1 * B.r[1];
      cost += 0.5 * B.rho0;
    }
  }
  LSTAMP(9);
  // the track's sums over its lanes (fixed order)
  wPi = track_sum3<LPT>(wPi), wTi = track_sum3<LPT>(wTi), wTic = track_sum3<LPT>(wTic), wTx = track_sum3<LPT>(wTx);
  wtd = track_sum<LPT>(wtd), a = track_sum<LPT>(a), b = track_sum<LPT>(b), cost = track_sum<LPT>(cost);
  const bool lead = valid && q == 0;
  if (lead) {
    tile[lml][6 * i + 0] = wPi.x, tile[lml][6 * i + 1] = wPi.y, tile[lml][6 * i + 2] = wPi.z;
    tile[lml][6 * i + 3] = wTi.x, tile[lml][6 * i + 4] = wTi.y, tile[lml][6 * i + 5] = wTi.z;
    if (est_ex) {
      tile[lml][66] = wTic.x, tile[lml][67] = wTic.y, tile[lml][68] = wTic.z;
      tile[lml][69] = wTx.x, tile[lml][70] = wTx.y, tile[lml][71] = wTx.z;
    }
    if (est_td) tile[lml][72] = wtd;
  }
  // per-landmark scalars
  double g2 = 0, asv2 = 0, lam2 = 0, bmax = 0;
  if (lead) {
    double s = 1.0, D2 = a;
    if (mode == MODE_SOLVE) {
      if (!tr->scaled) {
        s = 1.0 / (1.0 + sqrt(a));  // jacobi_scaling, fixed at iteration 0
        S->scale_l[l] = s;
      } else {
        s = S->scale_l[l];
      }
      D2 = fmin(fmax(s * s * a, 1e-6), 1e32);  // min/max_lm_diagonal
      const double dg = sqrt(D2);
      const double gr = s * b / dg;  // DoglegStrategy::ComputeGradient
      S->diag_l[l] = dg;
      S->grad_l[l] = gr;
      g2 = gr * gr;
      const double v = gr / dg;
      asv2 = s * s * a * v * v;
      tile[lml][COL_K] = b / D2;
      // Schur weight: e-block + lm_diagonal^2 (k_schur restates this when only mu changes)
      const double s2a = s * s * a;
      const double eb = s2a + lv.mu * D2, einv = 1.0 / eb;
      le[lml] = eb, lcoef[lml] = s * s * einv;
      S->einv_l[l] = einv;
    } else {
      const double cf = (a > 1e-8) ? 1.0 / a : 0.0;  // eps of marginalization_factor.h:70 on the diagonal block
      le[lml] = a, lcoef[lml] = cf;
      S->einv_l[l] = cf;
    }
    S->a[l] = a;
    S->b[l] = b;
    tile[lml][COL_B] = b;
    lam2 = lam * lam;
    bmax = fabs(b);
  } else {
    cost = 0.0;  // counted once per track
  }
  LSTAMP(18);
  cost = wave_sum(cost);
  g2 = wave_sum(g2);
  asv2 = wave_sum(asv2);
  lam2 = wave_sum(lam2);
  bmax = wave_max(bmax);
  if ((tid & 63) == 0) {
    double *r = red + 8 * (tid >> 6);
    r[0] = cost, r[1] = g2, r[2] = asv2, r[3] = lam2, r[4] = bmax;
  }
  lds_barrier();  // (the per-landmark scalars stored above are read by later kernels only: their write latency is not waited for)
  if (tid < 5) {
    // (the scalar partials are kept per block of 64 landmarks — k_backsub's and k_cost's unit: the second half-block of the
    // 8-lanes form writes slots 10 .. 14 of the same record, k_sum adds both)
    double *p = S->lm_part + (size_t)(LPT == 8 ? wg >> 1 : wg) * LMS + (LPT == 8 && (wg & 1) ? 10 : 0);
    p[tid] = tid < 4 ? ((red[tid] + red[8 + tid]) + (red[16 + tid] + red[24 + tid]))
                     : fmax(fmax(red[4], red[12]), fmax(red[20], red[28]));
  }
  LSTAMP(19);
  // the rows leave over their non-zero span only (the block's rows are one contiguous stretch of W: offsets are prefix sums)
  if (valid) {
    double *w = S->W + woff_l;
    for (int ci = q; ci < w_row_len(cnt_l); ci += LPT) w[ci] = tile[lml][w_col(ci, i, cnt_l)];
  }
  if (TIGHT && mode == MODE_SOLVE && S->N <= SPEC_MAX_LM) {  // (TIGHT: not in the role-by-role launches of a resident batch, whose k_dogleg keeps the compact rows)
    // small windows: the block's columns once more, transposed (Slot::Wt) — 512-byte lines, LDS stride 81: no bank conflict
    // (columns in pairs — [pair][landmark][2] — so that the reader's row is WT_PAIRS 16-byte loads: a wave has 63 loads in flight at most)
    constexpr int WGRP = LIN_THREADS / LMB;  // thread groups of LMB lanes: group g takes the column pairs g, g + WGRP, ...
    const int lr = tid % LMB, wg4 = tid / LMB;
    double2 *wt = (double2 *)(double *)S->Wt + wg * LMB + lr;
    constexpr int WPT = (WT_PAIRS + WGRP - 1) / WGRP;  // pairs per thread: all read from the tile, then all stored
    double2 v[WPT];
#pragma unroll
    for (int k = 0; k < WPT; k++) {
      const int cp = wg4 + WGRP * k, c = cp < WT_PAIRS ? 2 * cp : 0;
      v[k] = make_double2(tile[lr][c], c + 1 < KC ? tile[lr][c + 1] : 0.0);
    }
#pragma unroll
    for (int k = 0; k < WPT; k++) {
      const int cp = wg4 + WGRP * k;
      if (cp < WT_PAIRS) wt[(size_t)cp * SPEC_MAX_LM] = v[k];
    }
  }
  LSTAMP(23);
  schur_block<TIGHT>(wg * LMB, LMB, mode, Nlim, tile, lcoef, le, part);
  LSTAMP(20);
}

// Gram role: one chunk (<= 64 observations of one frame pair) per WAVE, four chunks per workgroup; the waves never
// meet (no workgroup barrier in this role).  Every lane turns its observation into the two 14-wide basis rows c0, c1;
// the 14x14 Gram sum of c c^T over the chunk is a SYRK and runs on the FP64 matrix pipe: 16 lanes at a time put their
// rows in LDS ([32 rows][16 + 1]), then eight v_mfma_f64_16x16x4_f64 take four rows each (lane (k, col) feeds row k,
// column col to both operands: D += C^T C).  The accumulator is 4 doubles per lane instead of 105 and no cross-lane
// reduction of the Gram entries is left.  LDS operations of one wave execute in program order, so between the phases
// only the compiler has to be kept from moving them (wavefront-scope fences).
template <bool OFFS>
DEV void lin_gram_role(Slot *S, const LinView &lv, int wg, int mode, double *lds) {
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int chunk = 4 * wg + wv;
  if (chunk >= S->nChunks) return;
  if (is_marg(mode) && chunk >= marg_plan(S, mode)->nChunks0) return;
  double *my = lds + wv * (32 * 17 + 256 + 14 * 20);
  double(*stage)[17] = (double(*)[17]) my;              // 32 rows x (16 + 1 pad); later T1 (14 x 20)
  double(*Qf)[16] = (double(*)[16])(my + 32 * 17);      // 16 x 16
  double(*E)[20] = (double(*)[20])(my + 32 * 17 + 256);  // 14 x 20
  double(*T1)[20] = (double(*)[20]) my;
  const Tab *T = lv.tab;
  const int pair = S->chunk_pair[chunk];
  const int j = pair % 11;
  const int begin = S->chunk_begin[chunk], end = S->chunk_end[chunk];
  const int est_td = S->est_td;
  const double td = lv.x->td;
  PairU u;
  if (OFFS) load_pair_uniform(T, pair, tab_offmask(T), u);
  else load_pair_uniform(T, pair, u);
  // E: basis(14) -> factor columns(20) = [Pi th_i Pj th_j tic th_ic td r]
  for (int e = lane; e < 14 * 20; e += 64) E[0][e] = 0.0;
  for (int e = lane; e < 32 * 17; e += 64) stage[0][e] = 0.0;  // columns 14, 15 stay zero
  const int idx = begin + lane;
  double c0[14], c1[14];
#pragma unroll
  for (int e = 0; e < 14; e++) c0[e] = c1[e] = 0.0;
  if (idx < end) {
    const int oj = S->pm_obs[idx], l = S->pm_lm[idx];
    const int oi = S->lm_obs0[l];
    const double lam = lv.lam[l];
    ObsPair ob;
    load_obs(S, oi, ob.pi, ob.vi, ob.tdi, ob.rowi);
    load_obs(S, oj, ob.pj, ob.vj, ob.tdj, ob.rowj);
    Basis B;
    visual_basis<OFFS>(ob, lam, td, est_td, S->tr_over_row, S->half_row, S->sqrt_info, u, B);
    c0[0] = B.red[0].x, c0[1] = B.red[0].y, c0[2] = B.red[0].z;
    c1[0] = B.red[1].x, c1[1] = B.red[1].y, c1[2] = B.red[1].z;
    c0[3] = B.jti[0].x, c0[4] = B.jti[0].y, c0[5] = B.jti[0].z;
    c1[3] = B.jti[1].x, c1[4] = B.jti[1].y, c1[5] = B.jti[1].z;
    c0[6] = B.jtj[0].x, c0[7] = B.jtj[0].y, c0[8] = B.jtj[0].z;
    c1[6] = B.jtj[1].x, c1[7] = B.jtj[1].y, c1[8] = B.jtj[1].z;
    c0[9] = B.jtx[0].x, c0[10] = B.jtx[0].y, c0[11] = B.jtx[0].z;
    c1[9] = B.jtx[1].x, c1[10] = B.jtx[1].y, c1[11] = B.jtx[1].z;
    c0[12] = B.jtd[0], c1[12] = B.jtd[1];
    c0[13] = B.r[0], c1[13] = B.r[1];
  }
  double4_t acc = double4_t{0, 0, 0, 0};
  const int nrounds = (end - begin + 15) >> 4;  // rounds of 16 observations that hold any
  for (int r = 0; r < nrounds; r++) {
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
  // accumulator element (row = (lane >> 4) + 4 reg, col = lane & 15)
#pragma unroll
  for (int r = 0; r < 4; r++) Qf[(lane >> 4) + 4 * r][lane & 15] = acc[r];
  if (lane < 9) {
    const int r = lane / 3, c = lane % 3;
    const double m1 = T->M1[j][lane];
    E[r][c] = m1;                                       // dr/dPi  = red M1
    E[r][6 + c] = -m1;                                  // dr/dPj  = -red M1
    E[r][12 + c] = T->M2[pair][lane] - T->ricT[lane];   // dr/dtic = red (M2 - ric^T)
  } else if (lane < 12) {
    const int k = lane - 9;
    E[3 + k][3 + k] = 1.0;
    E[6 + k][9 + k] = 1.0;
    E[9 + k][15 + k] = 1.0;
  } else if (lane == 12) {
    E[12][18] = 1.0;
    E[13][19] = 1.0;
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  for (int e = lane; e < 14 * 20; e += 64) {
    const int r = e / 20, c = e % 20;
    double s = 0;
#pragma unroll
    for (int k = 0; k < 14; k++) s = fma(Qf[r][k], E[k][c], s);
    T1[r][c] = s;
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  double *out = S->gram_part + (size_t)chunk * NGP;
  for (int e = lane; e < 400; e += 64) {
    const int p = e / 20, c = e % 20;
    if (p > c) continue;
    double s = 0;
#pragma unroll
    for (int k = 0; k < 14; k++) s = fma(E[k][p], T1[k][c], s);
    out[gidx20(p, c)] = s;
  }
}

template <bool raw_ready>
DEV void lin_imu_role(Slot *S, const LinView &lv, int f, int mode, double *lds) {
  double(*Jr)[30] = (double(*)[30]) lds;
  double(*Jw)[31] = (double(*)[31])(lds + 450);
  double *rr = lds + 450 + 465, *rw = rr + 16;
  const int tid = threadIdx.x;
  double *out = S->imu_out + (size_t)f * IMU_OUT;
  const bool active = S->imu_active[f] && (mode == MODE_SOLVE || (f == 0 && marg_plan(S, mode)->use_imu0)) &&
                      (!S->sharded || S->pose_side);
  if (!active) {
    for (int e = tid; e < IMU_OUT; e += LIN_THREADS) out[e] = 0.0;
    return;
  }
  const FrameState *x = lv.x;
  if (raw_ready) {
    // a resident batch: k_imu_raw has evaluated the factor (one lane per factor, 64 factors per wave) — a workgroup per factor
    // running the two serial jobs below on one lane each spends whole waves of issue slots on them
    const double *raw = S->imu_raw + (size_t)f * IMU_RAW;
    for (int e = tid; e < 450; e += LIN_THREADS) Jr[e / 30][e % 30] = raw[e];
    if (tid < 15) rr[tid] = raw[450 + tid];
  } else {
    for (int e = tid; e < 450; e += LIN_THREADS) Jr[e / 30][e % 30] = 0.0;
    __syncthreads();
    // two single-lane jobs on two waves: residual | Jacobian
    if (tid == 0) imu_raw_residual(&S->imu[f], S->g, x->pose[f], x->sb[f], x->pose[f + 1], x->sb[f + 1], rr);
    if (tid == 64) imu_raw_jacobian(&S->imu[f], S->g, x->pose[f], x->sb[f], x->pose[f + 1], x->sb[f + 1], &Jr[0][0]);
  }
  __syncthreads();
  const double *Sq = S->imu_sqrt[f];
  if (tid < 15) {
    double s = 0;
    for (int k = 0; k < 15; k++) s = fma(Sq[tid * 15 + k], rr[k], s);
    rw[tid] = s;
  }
  for (int e = tid; e < 450; e += LIN_THREADS) {
    const int r = e / 30, c = e % 30;
    double s = 0;
    for (int k = r; k < 15; k++) s = fma(Sq[r * 15 + k], Jr[k][c], s);  // sqrt_info is upper triangular
    Jw[r][c] = s;
  }
  __syncthreads();
  for (int e = tid; e < 900; e += LIN_THREADS) {
    const int p = e / 30, c = e % 30;
    double s = 0;
#pragma unroll
    for (int k = 0; k < 15; k++) s = fma(Jw[k][p], Jw[k][c], s);
    out[e] = s;
  }
  if (tid < 30) {
    double s = 0;
#pragma unroll
    for (int k = 0; k < 15; k++) s = fma(Jw[k][tid], rw[k], s);
    out[900 + tid] = s;
  }
  if (tid == 32) {
    double s = 0;
    for (int k = 0; k < 15; k++) s = fma(rw[k], rw[k], s);
    out[930] = 0.5 * s;
  }
}

// STAGED (the workspace holds LIN_LDS_PRIOR doubles: the all-roles launch of a single window): J0 comes into LDS with one batch of
// coalesced loads while the block differences are formed — the two products then read LDS; from global memory every term of them was
// a load the loop had to wait for, and the prior's workgroup was the longest of the launch (12.3 us against 8.9 for an IMU factor).
template <bool STAGED>
DEV void lin_prior_role(Slot *S, const LinView &lv, int mode, double *lds) {
  const int n = S->prior_n;
  const bool staged = STAGED && n <= LIN_PRIOR_N;
  double *Js = lds, *dx = lds + (STAGED ? LIN_PRIOR_N * LIN_PRIOR_N : 0), *r = dx + KP, *part = r + KP;  // part: [2][KP]
  const int tid = threadIdx.x;
  double *g = S->prior_g;
  for (int c = tid; c < KP + 4; c += LIN_THREADS) g[c] = 0.0;
  if (!S->prior_valid || (S->sharded && !S->pose_side)) return;
  const FrameState *x = lv.x;
  const double *Jg = S->prior_J;
  if (staged) {
    constexpr int PER = (LIN_PRIOR_N * LIN_PRIOR_N + LIN_THREADS - 1) / LIN_THREADS;
    double v[PER];
#pragma unroll
    for (int k = 0; k < PER; k++) v[k] = tid + LIN_THREADS * k < n * n ? Jg[tid + LIN_THREADS * k] : 0.0;
    if (tid < S->prior_nb) prior_block_dx(S, x, tid, dx);
#pragma unroll
    for (int k = 0; k < PER; k++)
      if (tid + LIN_THREADS * k < n * n) Js[tid + LIN_THREADS * k] = v[k];
  } else if (tid < S->prior_nb) {
    prior_block_dx(S, x, tid, dx);
  }
  __syncthreads();
  const double *J = staged ? Js : Jg;
  // r = r0 + J0 dx: 4 lanes per row
  for (int row = tid >> 2; row < n; row += LIN_THREADS / 4) {
    double s = 0;
    for (int c = tid & 3; c < n; c += 4) s = fma(J[row * n + c], dx[c], s);
    s = quad_sum(s);
    if ((tid & 3) == 0) r[row] = S->prior_r[row] + s;
  }
  __syncthreads();
  // g = J0^T r: column c, two halves of the rows
  {
    const int c = tid & 127, h = tid >> 7, k0 = h ? n / 2 : 0, k1 = h ? n : n / 2;
    for (int cc = c; cc < n; cc += 128) {
      double s = 0;
      for (int k = k0; k < k1; k++) s = fma(J[k * n + cc], r[k], s);
      part[h * KP + cc] = s;
    }
  }
  __syncthreads();
  for (int c = tid; c < n; c += LIN_THREADS) g[S->prior_cmap[c]] = part[c] + part[KP + c];
  if (tid < 64) {
    double cs = 0;
    for (int row = tid; row < n; row += 64) cs += r[row] * r[row];
    cs = wave_sum(cs);
    if (tid == 0) g[KP] = 0.5 * cs;
  }
}

// mode_bits: the mode, plus MODE_GATED for the marginalization sweep that rides behind the solve passes in the same graph
// (see tail_gate)
// k_imu_raw: grid ceil(10 * batch / 64) x 64 — resident batches only.  Unweighted residual and Jacobian of the IMU factors of
// the pass, ONE LANE PER FACTOR (64 factors per wave, slot = index / 10): the two evaluations are a few thousand serial
// instructions each, and a workgroup per factor running them on one lane spends a whole wave's issue slots per factor —
// at 512 windows that was a quarter of the sweep.  Written to S->imu_raw for the IMU role of k_lin<LIN_ROLE_POSE_RAW>, which
// keeps the part that is parallel inside a factor (sqrt_info weighting, J^T J, J^T r).
__global__ __launch_bounds__(64) void k_imu_raw(char *base, size_t stride, int count) {
  const int g = blockIdx.x * 64 + threadIdx.x;
  if (g >= count * LFVIO_WINDOW_SIZE) return;
  const int slot = g / LFVIO_WINDOW_SIZE, f = g - slot * LFVIO_WINDOW_SIZE;
  Slot *S = reinterpret_cast<Slot *>(base + stride * (size_t)slot);
  const TRFlags fl = tr_flags(&S->tr);
  if (fl.done | !fl.do_lin) return;
  if (!S->imu_active[f] || (S->sharded && !S->pose_side)) return;
  const FrameState *x = &S->x[fl.cur];
  double *raw = S->imu_raw + (size_t)f * IMU_RAW;
  for (int e = 0; e < 450; e++) raw[e] = 0.0;  // (imu_raw_jacobian fills the non-zero blocks)
  imu_raw_residual(&S->imu[f], S->g, x->pose[f], x->sb[f], x->pose[f + 1], x->sb[f + 1], raw + 450);
  imu_raw_jacobian(&S->imu[f], S->g, x->pose[f], x->sb[f], x->pose[f + 1], x->sb[f + 1], raw);
}

// ROLES: which of the roles the instantiation contains.  The launch of a single window carries all of them; a resident
// batch goes out role by role (lfvio_hip.hip launch_lin), and each of those kernels is compiled for its role alone: a
// third of the code in the instruction cache, and the Gram role — 14-wide basis rows and a 4-double accumulator — runs
// three waves per SIMD where the landmark role needs the registers of two.
constexpr int LIN_ROLE_LM = 1, LIN_ROLE_GRAM = 2, LIN_ROLE_POSE = 4, LIN_ROLE_ALL = 7;
constexpr int LIN_ROLE_POSE_RAW = 8;  // the pose-side roles with the IMU factors evaluated beforehand by k_imu_raw
constexpr int LIN_LDS_POSE = 1024;     // doubles: what the IMU (947) and the prior (688) roles carve out of the workspace
// OFFS: the instantiation whose visual roles know the reference's two back-rotations of a quaternion off the unit sphere (struct Tab,
// dev_types.h; it reads the table's mask and serves every table).  The host launches it where the point may hold such a quaternion —
// the first pass of a call whose uploaded state has one, every pass of a window whose FIXED extrinsic has one (SlotHostInfo::offs_*)
// — and the instantiation without that flavour everywhere else.
template <int ROLES, bool OFFS = true>
__global__ __launch_bounds__(LIN_THREADS, ROLES == LIN_ROLE_GRAM ? 3 : (ROLES == LIN_ROLE_POSE_RAW ? 6 : 2)) void k_lin(char *base, size_t stride, int mode_bits, int gLw, int gCh) {
  Slot *S = SLOT(base, stride);
  TRState *tr = &S->tr;
  const int mode = mode_bits & (MODE_GATED - 1);
  LSTAMP(21);
  const TRFlags fl = tr_flags(tr);
  int do_lin = fl.do_lin, do_schur = fl.do_schur, cur = fl.cur, acc_z = 0;
  double mu = tr->mu;
  int num_succ = tr->num_succ;
  if (mode_bits & MODE_GATED) {
    if (!tail_gate(S, fl.done)) return;
  } else {
    int done = fl.done;
    const bool owner = blockIdx.x == 0 && !(mode_bits & MODE_NOCOUNT);
    if ((mode_bits & MODE_DECIDE) && !done) {
      // The bookkeeping of the pass before (what k_decide does when it is launched on its own), repeated by every
      // workgroup: none of them may write the header the others are still reading, so workgroup 0 leaves the outcome in
      // S->dec for k_sum and k_solve (which moves it into the header) and each workgroup goes on with its own copy.
      __shared__ TRDecision dsh;
      if (threadIdx.x < 64) {
        TRHead t = *reinterpret_cast<const TRHead *>(tr);
        const int K = decide_candidates(t);
        DecideSums sm;
        decide_sums(S, t, K, 0, S->nLmBlocks, threadIdx.x, sm);
        if (threadIdx.x == 0) {
          const int az = decide_walk(t, sm, K, 0, S->max_iter, owner ? tr : nullptr);
          decision_from(dsh, t, az);
          if (owner) S->dec = dsh, S->dec_pending = 1;
        }
      }
      __syncthreads();
      // (wave-uniform: kept on the scalar side, the view's pointers included)
      do_lin = __builtin_amdgcn_readfirstlane(dsh.do_lin), do_schur = __builtin_amdgcn_readfirstlane(dsh.do_schur);
      cur = __builtin_amdgcn_readfirstlane(dsh.cur), acc_z = __builtin_amdgcn_readfirstlane(dsh.acc_z), done = __builtin_amdgcn_readfirstlane(dsh.done);
      num_succ = __builtin_amdgcn_readfirstlane(dsh.num_succ);
      mu = __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(dsh.mu)), __builtin_amdgcn_readfirstlane(__double2loint(dsh.mu)));
      if (acc_z > 0 && blockIdx.x == gridDim.x - 1 && !(mode_bits & MODE_NOCOUNT)) {
        copy_accepted(S, acc_z, cur, S->N, threadIdx.x, LIN_THREADS);
        if (S->spec_on) {  // the accepted state is complete in x[cur] / lam[cur] once this copy is: a worker may take it (kernels_spec.h)
          __threadfence();
          __syncthreads();
          if (threadIdx.x == 0) spec_publish(S, num_succ, cur);
        }
      } else if (acc_z == 0 && owner && threadIdx.x == 0 && S->spec_on) spec_publish(S, num_succ, cur);  // (candidate 0 was written where it lies by the pass before)
    } else if (owner && threadIdx.x == 0 && mode == MODE_SOLVE && S->spec_on) spec_publish(S, num_succ, cur);  // (the header's own state: the first pass of a graph)
    // a pass that starts with the loop still open is a pass this slot needs (the synchronous drivers size the first
    // graph of the next call from this count)
    if (mode == MODE_SOLVE && owner && !done && threadIdx.x == 0) S->passes_used++;
    if (done | (!do_lin & !do_schur)) return;
  }
  LinView lv;
  lv.x = acc_z > 0 ? &S->xE[acc_z > 0 ? acc_z - 1 : 0] : &S->x[cur];
  lv.tab = acc_z > 0 ? &S->tabE[acc_z > 0 ? acc_z - 1 : 0] : &S->tab[cur];
  lv.lam = acc_z > 0 ? (const double *)S->lamE[acc_z > 0 ? acc_z - 1 : 0] : (const double *)S->lam[cur];
  lv.mu = mu;
  __shared__ __attribute__((aligned(16))) double lds[(ROLES & (LIN_ROLE_LM | LIN_ROLE_GRAM)) ? LIN_LDS : LIN_LDS_POSE];  // one workspace, aliased per role
  // the grid is sized for the largest resident window (gLw, gCh); each slot uses its own counts
  int b = blockIdx.x;
  if (b < gLw) {
    // One landmark block per workgroup, one Schur partial per block.  (Several blocks per workgroup with the partial summed in
    // LDS was built and measured at 100 000 landmarks: k_presum + k_sum 33.5 -> 25.3 us, but the loop around the sweep — nothing
    // is carried through it — costs it 50 spilled registers and the landmark role 64 -> 97 us.)
    if (ROLES & LIN_ROLE_LM) {
      const int half = S->lm_half, lmb = half ? LM_BLOCK / 2 : LM_BLOCK;
      const int nblk = ((is_marg(mode) ? marg_plan(S, mode)->N0 : S->N) + lmb - 1) / lmb;
      if (b >= nblk) return;
      double *part = S->schur_part + (size_t)b * SCHUR_LEN;
      if (half) {  // (wave-uniform)
        if (do_lin) lin_landmark_role<ROLES == LIN_ROLE_ALL, 8, OFFS>(S, lv, b, mode, lds, part);
        else lin_schur_only_role<8>(S, lv, b, lds, part);
      } else {
        if (do_lin) lin_landmark_role<ROLES == LIN_ROLE_ALL, 4, OFFS>(S, lv, b, mode, lds, part);
        else lin_schur_only_role<4>(S, lv, b, lds, part);
      }
    }
    return;
  }
  if (!do_lin) return;
  b -= gLw;
  if (b < gCh) {  // gCh workgroups of 4 chunks
    if (ROLES & LIN_ROLE_GRAM) lin_gram_role<OFFS>(S, lv, b, mode, lds);
    return;
  }
  b -= gCh;
  if (ROLES & (LIN_ROLE_POSE | LIN_ROLE_POSE_RAW)) {
    if (b < LFVIO_WINDOW_SIZE) {
      if (ROLES & LIN_ROLE_POSE_RAW) lin_imu_role<true>(S, lv, b, mode, lds);
      else lin_imu_role<false>(S, lv, b, mode, lds);
      return;
    }
    lin_prior_role<ROLES == LIN_ROLE_ALL>(S, lv, mode, lds);
  }
}

// element (R, Cc) of the reduced 80x80 accumulator, R <= Cc
DEV int schur_index(int R, int Cc) {
  const int t = R >> 4, u = Cc >> 4;
  const int tile = t * 5 - (t * (t - 1)) / 2 + (u - t);
  const int row = R & 15, col = Cc & 15;
  return tile * 256 + (row >> 2) * 64 + ((row & 3) << 4) + col;
}

// ---------------------------------------------------------------------------
// k_sum: grid (HPP_BLOCKS + SCHUR_LEN/256 + 1, batch) x 256   (pre = 1: second level, after k_presum)
//   role 1: every packed entry of the pose-side Gauss-Newton Hessian H_pp (and of g_p) is
//           produced by exactly ONE thread that adds its contributions in a fixed order —
//           Gram chunks of the frame pairs touching it, the (at most two) IMU factors, the
//           prior — so the result is deterministic and needs no atomics
//   role 2: fixed-order sum of the Schur SYRK partials
//   role 3: landmark scalar partials
// ---------------------------------------------------------------------------
constexpr int HPP_ITEMS = PACKED + KP;
constexpr int HPP_BLOCKS = (HPP_ITEMS + 255) / 256;

// IMU factor f sees tangent column c as local column (0..29) or -1
DEV int imu_local(int c, int f) {
  if (c < 66) {
    const int fr = c / 6, l = c - 6 * fr;
    if (fr == f) return l;
    if (fr == f + 1) return 15 + l;
    return -1;
  }
  if (c < KC) return -1;
  const int fr = (c - KC) / 9, l = (c - KC) - 9 * fr;
  if (fr == f) return 6 + l;
  if (fr == f + 1) return 21 + l;
  return -1;
}
DEV int col_frame(int c) { return c < 66 ? c / 6 : (c < KC ? -10 : (c - KC) / 9); }
// Does this rank add the pose-side factors (IMU, prior) to an entry of H_pp / g_p whose (larger) tangent index is r?
//   Slot::pose_side 1  the pose-side rank of a landmark-sharded window (and every unsharded window): everywhere;
//   Slot::pose_side 2  the other ranks of an lfvio_group: they evaluate the same factors — replicated, a few microseconds — and add
//                      them OUTSIDE the camera part only (rows >= KC: speed / bias), which is then complete on every rank and
//                      stays out of the all-reduce (54 KB instead of 151: group.inc); the camera part gets them once, on rank 0;
//   Slot::pose_side 0  a rank of a caller that all-reduces the whole exchange buffer itself (lfvio_shard_*): nowhere.
DEV bool pose_terms_here(const Slot *S, int r) { return !S->sharded || S->pose_side == 1 || (S->pose_side == 2 && r >= KC); }

DEV bool col_active(const Slot *S, int c, int mode) {
  if (mode >= MODE_MARG) return true;
  if (!S->est_ex && c >= off_ex() && c < off_ex() + 6) return false;
  if (!S->est_td && c == off_td()) return false;
  return true;
}

// ---------------------------------------------------------------------------
// k_presum: grid (NPAIR + 15 * groups + 1, batch) x 256, groups = ceil(parts / PRE_GROUP) — only launched for large windows, where one thread of
// k_sum would otherwise walk thousands of partials.  First level of the fixed-order reductions:
//   [0, NPAIR)            Gram chunks of one frame pair -> pairG[pair]
//   [NPAIR, +25*G)        Schur SYRK partials in groups of PRE_GROUP, in place (the sum lands in the group's first part)
//   last                  landmark scalar partials -> lm_sum
// ---------------------------------------------------------------------------
constexpr int PRE_GROUP = 32;
__global__ __launch_bounds__(256) void k_presum(char *base, size_t stride, int mode_bits, int groups) {
  const int PRE_SCHUR_BLOCKS = (SCHUR_LEN / 256) * groups;
  Slot *S = SLOT(base, stride);
  const TRState *tr = &S->tr;
  const int mode = mode_bits & (MODE_GATED - 1);
  const TRFlags fl = tr_flags_decided(S);
  if (mode_bits & MODE_GATED) {
    if (!tail_gate(S, fl.done)) return;
  } else if (fl.done | (!fl.do_lin & !fl.do_schur)) return;  // nothing was re-linearized in this pass (rejected step): the sums stand
  const int tid = threadIdx.x;
  int b = blockIdx.x;
  if (b < NPAIR) {
    if (!fl.do_lin) return;
    const int chunk_limit = is_marg(mode) ? marg_plan(S, mode)->nChunks0 : S->nChunks;
    const int c0 = S->pair_chunk0[b];
    int c1 = S->pair_chunk0[b + 1];
    if (c1 > chunk_limit) c1 = chunk_limit;
    if (tid < NGP) {
      const double *gp = S->gram_part + tid;
      double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
      int c = c0;
      for (; c + 4 <= c1; c += 4) {
        s0 += gp[(size_t)c * NGP], s1 += gp[(size_t)(c + 1) * NGP];
        s2 += gp[(size_t)(c + 2) * NGP], s3 += gp[(size_t)(c + 3) * NGP];
      }
      for (; c < c1; c++) s0 += gp[(size_t)c * NGP];
      S->pairG[(size_t)b * NGP + tid] = (s0 + s1) + (s2 + s3);
    }
    return;
  }
  b -= NPAIR;
  if (b < PRE_SCHUR_BLOCKS) {
    if (!fl.do_schur) return;
    const int grp = b / (SCHUR_LEN / 256), e = (b % (SCHUR_LEN / 256)) * 256 + tid;
    int parts = S->nSchurParts;
    if (is_marg(mode)) parts = (marg_plan(S, mode)->N0 + S->schur_lm - 1) / S->schur_lm;
    const int p0 = grp * PRE_GROUP;
    if (p0 >= parts) return;
    const int p1 = min(parts, p0 + PRE_GROUP);
    double *sp = S->schur_part + e;
    double a0 = 0, a1 = 0, a2 = 0, a3 = 0;
    int p = p0;
    for (; p + 4 <= p1; p += 4) {
      a0 += sp[(size_t)p * SCHUR_LEN], a1 += sp[(size_t)(p + 1) * SCHUR_LEN];
      a2 += sp[(size_t)(p + 2) * SCHUR_LEN], a3 += sp[(size_t)(p + 3) * SCHUR_LEN];
    }
    for (; p < p1; p++) a0 += sp[(size_t)p * SCHUR_LEN];
    sp[(size_t)p0 * SCHUR_LEN] = (a0 + a1) + (a2 + a3);
    return;
  }
  if (!fl.do_lin) return;
  // landmark scalars: 4 sums + 1 max over the blocks; lane-strided partials, then a fixed tree
  __shared__ double red[5][256];
  const int blocks = S->nLmBlocks;
  double v[5] = {0, 0, 0, 0, 0};
  for (int k = tid; k < blocks; k += 256) {
    const double *q = S->lm_part + (size_t)k * LMS;
    v[0] += q[0], v[1] += q[1], v[2] += q[2], v[3] += q[3], v[4] = fmax(v[4], q[4]);
  }
#pragma unroll
  for (int k = 0; k < 5; k++) red[k][tid] = v[k];
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if (tid < w) {
#pragma unroll
      for (int k = 0; k < 4; k++) red[k][tid] += red[k][tid + w];
      red[4][tid] = fmax(red[4][tid], red[4][tid + w]);
    }
    __syncthreads();
  }
  if (tid < 5) S->lm_sum[tid] = red[tid][0];
}

// Byte offsets, inside a slot blob, of the arrays k_sum's H_pp role reads first (the same for every slot of a context):
// passed BY VALUE, so that the first round of loads needs nothing from memory but the block and thread index — a GP<>
// member would have to be fetched first, and every dependent fetch is a memory round trip (~1.5 us) of this short kernel.
struct SumArgs {
  long long sum_off, sum_end_marg, sum_items, gram_part, pairG, imu_out, prior_A;
};
template <class T>
DEV T *blob_at(const Slot *S, long long off) { return (T *)((char *)S + off); }
#define SUM_KEEP(v) asm volatile("" ::"v"(v))

// One entry e of the packed H_pp (e < PACKED) or of g_p: Gram partials of the frame pairs touching it, the (at most two)
// IMU factors, the prior — added in that fixed order.  Three rounds of loads: (1) flags, list bounds, IMU / prior operands
// and indices, all addressed from the arguments; (2) the list of partial offsets, the prior's normal-matrix entry;
// (3) the partials.
DEV void sum_hpp_entry(Slot *S, const SumArgs &o, int mode_bits, int e) {
  const int mode = mode_bits & (MODE_GATED - 1);
  const bool in = e < HPP_ITEMS;
  const int ec = in ? e : 0;  // (the threads past the end go through the loads with entry 0)
  const bool packed = ec < PACKED;
  int r, c;
  if (packed) {
    r = (int)((sqrt(8.0 * ec + 1.0) - 1.0) * 0.5);
    while ((r + 1) * (r + 2) / 2 <= ec) r++;
    while (r * (r + 1) / 2 > ec) r--;
    c = ec - r * (r + 1) / 2;
  } else {
    r = c = ec - PACKED;
  }
  // ---- round 1
  const TRFlags fl = tr_flags_decided(S);
  const int tail_state = S->tail_state, est_ex = S->est_ex, est_td = S->est_td, prior_valid = S->prior_valid, sharded = S->sharded;
  const int pose_side = S->pose_side, pre_gram = S->pre_gram, prior_n = S->prior_n;
  const int marg_chunks = is_marg(mode) ? marg_plan(S, mode)->nChunks0 : 1;
  // (list bounds by the compact index of the entries that have a visual part; the others read entry 0 and drop it)
  const int vis = r < KC ? (packed ? ec : SUM_VIS_PACKED + r) : 0;
  const int beg = blob_at<int>(S, o.sum_off)[vis];
  int end = blob_at<int>(S, o.sum_off)[vis + 1];
  const int endm = blob_at<int>(S, o.sum_end_marg)[vis];
  const int pr = S->prior_inv[r], pc = S->prior_inv[c];
  const double pg = S->prior_g[r];
  double imu_v[2] = {0.0, 0.0};
  const int f0 = col_frame(r);
  const double *imu_out = blob_at<double>(S, o.imu_out);
#pragma unroll
  for (int u = 0; u < 2; u++) {
    const int f = f0 - 1 + u;
    if (f0 >= 0 && f >= 0 && f < LFVIO_WINDOW_SIZE) {
      const int pl = imu_local(r, f), ql = imu_local(c, f);
      if (packed) {
        if (pl >= 0 && ql >= 0) imu_v[u] = imu_out[(size_t)f * IMU_OUT + pl * 30 + ql];
      } else if (pl >= 0) {
        imu_v[u] = imu_out[(size_t)f * IMU_OUT + 900 + pl];
      }
    }
  }
  SUM_KEEP(beg);
  SUM_KEEP(end);
  SUM_KEEP(endm);
  SUM_KEEP(pr);
  SUM_KEEP(pc);
  SUM_KEEP(pg);
  SUM_KEEP(imu_v[0]);
  SUM_KEEP(imu_v[1]);
  if (mode_bits & MODE_GATED) {
    if (!(fl.done && tail_state == 0)) return;
  } else if (fl.done | !fl.do_lin) return;
  if (!in) return;
  const bool act_r = is_marg(mode) || !((!est_ex && r >= off_ex() && r < off_ex() + 6) || (!est_td && r == off_td()));
  const bool act_c = is_marg(mode) || !((!est_ex && c >= off_ex() && c < off_ex() + 6) || (!est_td && c == off_td()));
  double val = 0.0;
  if (act_r && act_c) {
    // ---- round 2
    if (is_marg(mode)) end = marg_chunks > 0 ? endm : beg;
    const bool pterms = !sharded || pose_side == 1 || (pose_side == 2 && r >= KC);  // pose_terms_here(S, r)
    const bool use_prior = packed && prior_valid && pterms && pr >= 0 && pc >= 0;
    double pa = 0.0;
    if (use_prior) pa = blob_at<double>(S, o.prior_A)[pr * prior_n + pc];
    if (r < KC) {
      // visual: four accumulators, fixed association => deterministic
      const int *it = blob_at<int>(S, o.sum_items);
      const double *gp = blob_at<double>(S, pre_gram ? o.pairG : o.gram_part);
      double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
      int k = beg;
      for (; k + 4 <= end; k += 4) {
        const int o0 = it[k], o1 = it[k + 1], o2 = it[k + 2], o3 = it[k + 3];
        s0 += gp[o0], s1 += gp[o1], s2 += gp[o2], s3 += gp[o3];
      }
      for (; k < end; k++) s0 += gp[it[k]];
      val += (s0 + s1) + (s2 + s3);
    }
    if (packed) {
      if (f0 >= 0 && pterms) {
#pragma unroll
        for (int u = 0; u < 2; u++) {
          const int f = f0 - 1 + u;
          if (f >= 0 && f < LFVIO_WINDOW_SIZE && imu_local(r, f) >= 0 && imu_local(c, f) >= 0) val += imu_v[u];
        }
      }
      if (use_prior) val += pa;  // prior: A' = J0^T J0
    } else {
      if (f0 >= 0 && pterms) {
#pragma unroll
        for (int u = 0; u < 2; u++) {
          const int f = f0 - 1 + u;
          if (f >= 0 && f < LFVIO_WINDOW_SIZE && imu_local(r, f) >= 0) val += imu_v[u];
        }
      }
      if (pterms) val += pg;
    }
  }
  if (packed) S->Hpp[e] = val;
  else S->gp[r] = val;
}

__global__ __launch_bounds__(256) void k_sum(char *base, size_t stride, int mode_bits, int pre, const SumArgs o) {
  Slot *S = SLOT(base, stride);
  const int mode = mode_bits & (MODE_GATED - 1);
  const int tid = threadIdx.x;
  int b = blockIdx.x;
  if (S->sharded && mode == MODE_SOLVE && !(mode_bits & MODE_GATED)) {
    // Landmark-sharded window, stream-ordered driver: the sum-all-reduce behind this kernel goes out in EVERY pass.  In a
    // pass that re-linearized nothing (rejected step, finished loop) the exchange buffer of the pose-side rank still holds
    // the reduced sums of the last linearization; every other rank zeroes its copy, so the collective reproduces them, bit
    // for bit.  (Round 2 had a kernel of its own for this, k_xstage.)
    const TRFlags f0 = tr_flags(&S->tr);
    if (f0.done | (!f0.do_lin & !f0.do_schur)) {
      if (S->pose_side != 1) {
        // (a rank of an lfvio_group, pose_side 2: only the camera part travels — its speed / bias rows are its own complete copy)
        const bool cam_only = S->pose_side == 2;
        const int e = b * 256 + tid;  // blocks: H_pp | g_p entries, then the Schur sums, then the scalars
        double *x = S->xch;
        if (b < HPP_BLOCKS) {
          if (e < (cam_only ? SUM_VIS_PACKED : PACKED)) x[XOFF_H + e] = 0.0;
          else if (e >= PACKED && e < PACKED + (cam_only ? KC : KP)) x[XOFF_G + (e - PACKED)] = 0.0;
        } else if (b < HPP_BLOCKS + SCHUR_LEN / 256) {
          x[XOFF_S + (e - HPP_BLOCKS * 256)] = 0.0;
        } else if (tid < 16) {
          x[XOFF_C + tid] = 0.0;
        }
      }
      return;
    }
  }
  if (b < HPP_BLOCKS) {
    sum_hpp_entry(S, o, mode_bits, b * 256 + tid);
    return;
  }
  const TRFlags fl = tr_flags_decided(S);
  if (mode_bits & MODE_GATED) {
    if (!tail_gate(S, fl.done)) return;
  } else if (fl.done | (!fl.do_lin & !fl.do_schur)) return;  // nothing was re-linearized in this pass (rejected step): the sums stand
  b -= HPP_BLOCKS;
  if (b < SCHUR_LEN / 256) {
    if (!fl.do_schur) return;
    const int e = b * 256 + tid;
    int parts = S->nSchurParts;
    if (is_marg(mode)) parts = (marg_plan(S, mode)->N0 + S->schur_lm - 1) / S->schur_lm;
    // fixed association (8 interleaved accumulators, then a fixed tree): deterministic, 8 loads in flight
    const size_t ps = pre ? (size_t)PRE_GROUP * SCHUR_LEN : (size_t)SCHUR_LEN;  // pre: one partial per group
    if (pre) parts = (parts + PRE_GROUP - 1) / PRE_GROUP;
    const double *sp = S->schur_part + e;
    double a0 = 0, a1 = 0, a2 = 0, a3 = 0, a4 = 0, a5 = 0, a6 = 0, a7 = 0;
    int p = 0;
    for (; p + 8 <= parts; p += 8) {
      a0 += sp[p * ps], a1 += sp[(p + 1) * ps], a2 += sp[(p + 2) * ps], a3 += sp[(p + 3) * ps];
      a4 += sp[(p + 4) * ps], a5 += sp[(p + 5) * ps], a6 += sp[(p + 6) * ps], a7 += sp[(p + 7) * ps];
    }
    for (; p < parts; p++) a0 += sp[p * ps];
    S->schur_sum[e] = ((a0 + a1) + (a2 + a3)) + ((a4 + a5) + (a6 + a7));
    return;
  }
  if (!fl.do_lin) return;
  int blocks = pre ? 0 : S->nLmBlocks;  // pre: lm_sum comes from k_presum
  if (pre) {
  } else if (tid < 4) {
    // (Slot::lm_half: a block's record holds two partials — the second where the block has more than 32 landmarks)
    const int half = S->lm_half, N = S->N;
    double s = 0;
    for (int k = 0; k < blocks; k++) {
      s += S->lm_part[(size_t)k * LMS + tid];
      if (half && k * LM_BLOCK + LM_BLOCK / 2 < N) s += S->lm_part[(size_t)k * LMS + 10 + tid];
    }
    S->lm_sum[tid] = s;
  } else if (tid == 4) {
    const int half = S->lm_half, N = S->N;
    double m = 0;
    for (int k = 0; k < blocks; k++) {
      m = fmax(m, S->lm_part[(size_t)k * LMS + 4]);
      if (half && k * LM_BLOCK + LM_BLOCK / 2 < N) m = fmax(m, S->lm_part[(size_t)k * LMS + 14]);
    }
    S->lm_sum[4] = m;
  }
  if (S->sharded) {
    // exchange scalars of phase A: local cost (pose-side factors on the owning rank only), gradient
    // norms, Cauchy landmark term, ||lambda||^2; the max is sent as a sum (upper bound, only feeds the
    // 1e-10 gradient tolerance)
    __syncthreads();
    double *sc = S->xch + XOFF_C;
    if (tid < 16) sc[tid] = 0.0;
    __syncthreads();
    if (tid == 0) {
      double cost = S->lm_sum[0];
      if (S->pose_side == 1) {
        cost += S->prior_g[KP];
        for (int f = 0; f < LFVIO_WINDOW_SIZE; f++) cost += S->imu_out[(size_t)f * IMU_OUT + 930];
      }
      sc[XS_COST] = cost;
      sc[XS_G2] = S->lm_sum[1], sc[XS_ASV2] = S->lm_sum[2], sc[XS_LAM2] = S->lm_sum[3], sc[XS_BMAX] = S->lm_sum[4];
      if (is_marg(mode)) sc[XS_N0] = (double)marg_plan(S, mode)->N0;  // landmarks this rank eliminates
    }
  }
}
