// kernels_lin.h — linearization sweep, MFMA Schur SYRK, partial reductions.
//
//   k_setup   : state reset, per-pair tables, IMU sqrt-information, prior A' = J0^T J0
//   k_lin     : ONE launch, four workgroup roles selected by blockIdx.x
//                 [landmark blocks | Gram chunks | IMU factors | prior]
//   k_schur   : sum_l c_l w_l w_l^T (80x80) on v_mfma_f64_16x16x4_f64
//   k_sum     : deterministic fixed-order reduction of the per-workgroup partials
#pragma once
#include "dev_factors.h"

#define SLOT(base, stride) ((Slot *)((char *)(base) + (size_t)blockIdx.y * (stride)))

constexpr int MODE_SOLVE = 0;
constexpr int MODE_MARG = 1;        // MODE_MARG + flag: 1 = MARGIN_OLD, 2 = MARGIN_SECOND_NEW
DEV bool is_marg(int mode) { return mode >= MODE_MARG; }
DEV const MargPlan *marg_plan(const Slot *S, int mode) { return &S->marg[mode - MODE_MARG]; }

DEV int gidx20(int p, int q) { return p * 20 - (p * (p - 1)) / 2 + (q - p); }  // upper index, p <= q

// ---------------------------------------------------------------------------
// k_setup: grid (3, batch) x 256
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_setup(char *base, size_t stride, int mode) {
  Slot *S = SLOT(base, stride);
  const int tid = threadIdx.x;
  if (blockIdx.x == 0) {
    // state + trust-region init (TrustRegionMinimizer::Init, DoglegStrategy ctor)
    const double *src = (const double *)&S->x0;
    double *dst = (double *)&S->x[0];
    for (int k = tid; k < (int)(sizeof(FrameState) / 8); k += 256) dst[k] = src[k];
    for (int l = tid; l < S->N; l += 256) S->lam[0][l] = S->lam0[l];
    if (tid == 0) {
      TRState *t = &S->tr;
      t->radius = 1e4;
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
      if (mode >= MODE_MARG) t->mu = 0.0;
    }
    __syncthreads();
    build_tab(&S->x0, &S->tab[0], tid);
  } else if (blockIdx.x == 1) {
    // sqrt_info = LLT(cov^-1).matrixL()^T (imu_factor.h:64), hoisted out of the iteration loop:
    // the reference recomputes it in every Evaluate().  One factor per wave: Gauss-Jordan with
    // partial pivoting on [cov | I] in LDS, then a 15x15 Cholesky of the inverse.
    __shared__ double A[4][15][31];
    const int wv = tid >> 6, lane = tid & 63;
    for (int it = 0; it < 3; it++) {
      const int f = wv + 4 * it;
      const bool act = f < LFVIO_WINDOW_SIZE && S->imu_active[f];
      double(*M)[31] = A[wv];
      if (act)
        for (int e = lane; e < 225; e += 64) {
          int r = e / 15, c = e % 15;
          M[r][c] = S->imu[f].covariance[e];
          M[r][15 + c] = (r == c) ? 1.0 : 0.0;
        }
      __syncthreads();
      for (int k = 0; k < 15; k++) {
        int p = k;
        if (act) {
          double best = fabs(M[k][k]);
          for (int r = k + 1; r < 15; r++) {
            double v = fabs(M[r][k]);
            if (v > best) best = v, p = r;
          }
        }
        __syncthreads();
        if (act && p != k && lane < 30) {
          double t = M[k][lane];
          M[k][lane] = M[p][lane];
          M[p][lane] = t;
        }
        __syncthreads();
        double piv = act ? M[k][k] : 1.0;
        __syncthreads();
        if (act && lane < 30) M[k][lane] /= piv;
        __syncthreads();
        double fac[15];
#pragma unroll
        for (int r = 0; r < 15; r++) fac[r] = act ? M[r][k] : 0.0;
        __syncthreads();
        if (act && lane < 30) {
          double mk = M[k][lane];
#pragma unroll
          for (int r = 0; r < 15; r++)
            if (r != k) M[r][lane] -= fac[r] * mk;
        }
        __syncthreads();
      }
      if (act && lane == 0) {
        double L[15][15];
        bool ok = true;
        for (int j = 0; j < 15; j++) {
          double s = M[j][15 + j];
          for (int k = 0; k < j; k++) s -= L[j][k] * L[j][k];
          if (!(s > 0.0)) ok = false;
          double d = sqrt(s);
          L[j][j] = d;
          for (int i = j + 1; i < 15; i++) {
            double t = M[i][15 + j];
            for (int k = 0; k < j; k++) t -= L[i][k] * L[j][k];
            L[i][j] = t / d;
          }
        }
        for (int i = 0; i < 15; i++)
          for (int j = 0; j < 15; j++) S->imu_sqrt[f][i * 15 + j] = (j >= i && ok) ? L[j][i] : 0.0;
        if (!ok) S->imu_active[f] = 0;
      }
      __syncthreads();
    }
  } else {
    // prior: A' = J0^T J0, b0 = J0^T r0 (constant over the solve)
    if (!S->prior_valid) return;
    const int n = S->prior_n;
    const double *J = S->prior_J;
    for (int e = tid; e < n * n; e += 256) {
      int r = e / n, c = e % n;
      double s = 0;
      for (int k = 0; k < n; k++) s = fma(J[k * n + r], J[k * n + c], s);
      S->prior_A[e] = s;
    }
    for (int c = tid; c < n; c += 256) {
      double s = 0;
      for (int k = 0; k < n; k++) s = fma(J[k * n + c], S->prior_r[k], s);
      S->prior_b0[c] = s;
    }
  }
}

// ---------------------------------------------------------------------------
// k_lin: grid (nLmBlocks + nChunks + 10 + 1, batch) x 64
// ---------------------------------------------------------------------------
DEV void load_pair_uniform(const Tab *T, int pair, PairU &u) {
  u.M2 = ldm(T->M2[pair]);
  u.T = ldm(T->T[pair]);
  u.ric = ldm(T->ric);
  u.ricT = ldm(T->ricT);
  u.c = ld3(T->c[pair]);
  u.tic = ld3(T->tic);
}

DEV void lin_landmark_role(Slot *S, int blk, int mode, double *lds) {
  double(*tile)[WLD + 1] = (double(*)[WLD + 1]) lds;
  const int lane = threadIdx.x;
  const TRState *tr = &S->tr;
  const int cur = tr->cur;
  const Tab *T = &S->tab[cur];
  const int Nlim = is_marg(mode) ? marg_plan(S, mode)->N0 : S->N;
  const int l = blk * LM_BLOCK + lane;
  const bool valid = l < Nlim;
  const int est_td = S->est_td;
  const int est_ex = is_marg(mode) ? 1 : S->est_ex;  // ResidualBlockInfo::Evaluate asks for every Jacobian
  const double td = S->x[cur].td;
#pragma unroll 8
  for (int c = 0; c < WLD; c++) tile[lane][c] = 0.0;
  double a = 0, b = 0, cost = 0, lam = 1.0;
  if (valid) {
    const int i = S->lm_start[l], k = S->lm_cnt[l], o0 = S->lm_obs0[l];
    lam = S->lam[cur][l];
    ObsPair ob;
    load_obs(S, o0, ob.pi, ob.vi, ob.tdi, ob.rowi);
    d3 wPi = mk3(0, 0, 0), wTi = wPi, wTic = wPi, wTx = wPi;
    double wtd = 0;
    const m33 ricT = ldm(T->ricT);
    for (int o = 1; o < k; o++) {
      const int j = i + o, pair = i * 11 + j;
      load_obs(S, o0 + o, ob.pj, ob.vj, ob.tdj, ob.rowj);
      PairU u;
      load_pair_uniform(T, pair, u);
      Basis B;
      visual_basis(ob, lam, td, est_td, S->tr_over_row, S->half_row, S->sqrt_info, u, B);
      const double jl0 = B.jl[0], jl1 = B.jl[1];
      d3 eR = jl0 * B.red[0] + jl1 * B.red[1];
      const m33 M1 = ldm(T->M1[j]);
      d3 wp = vmul(eR, M1);  // M1^T eR
      wPi = wPi + wp;
      wTi = wTi + (jl0 * B.jti[0] + jl1 * B.jti[1]);
      d3 wtj = jl0 * B.jtj[0] + jl1 * B.jtj[1];
      tile[lane][6 * j + 0] = -wp.x, tile[lane][6 * j + 1] = -wp.y, tile[lane][6 * j + 2] = -wp.z;
      tile[lane][6 * j + 3] = wtj.x, tile[lane][6 * j + 4] = wtj.y, tile[lane][6 * j + 5] = wtj.z;
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
    tile[lane][6 * i + 0] = wPi.x, tile[lane][6 * i + 1] = wPi.y, tile[lane][6 * i + 2] = wPi.z;
    tile[lane][6 * i + 3] = wTi.x, tile[lane][6 * i + 4] = wTi.y, tile[lane][6 * i + 5] = wTi.z;
    if (est_ex) {
      tile[lane][66] = wTic.x, tile[lane][67] = wTic.y, tile[lane][68] = wTic.z;
      tile[lane][69] = wTx.x, tile[lane][70] = wTx.y, tile[lane][71] = wTx.z;
    }
    if (est_td) tile[lane][72] = wtd;
  }
  // per-landmark scalars
  double g2 = 0, asv2 = 0, lam2 = 0, bmax = 0;
  if (valid) {
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
      tile[lane][COL_K] = b / D2;
    }
    S->a[l] = a;
    S->b[l] = b;
    tile[lane][COL_B] = b;
    lam2 = lam * lam;
    bmax = fabs(b);
  }
  cost = wave_sum(cost);
  g2 = wave_sum(g2);
  asv2 = wave_sum(asv2);
  lam2 = wave_sum(lam2);
  bmax = wave_max(bmax);
  if (lane == 0) {
    double *p = S->lm_part + (size_t)blk * LMS;
    p[0] = cost, p[1] = g2, p[2] = asv2, p[3] = lam2, p[4] = bmax;
  }
  __syncthreads();
  double *Wb = S->W + (size_t)blk * LM_BLOCK * WLD;
  for (int e = lane; e < LM_BLOCK * WLD; e += 64) Wb[e] = tile[e / WLD][e % WLD];
}

DEV void lin_gram_role(Slot *S, int chunk, int mode, double *lds) {
  double(*Qf)[15] = (double(*)[15]) lds;
  double(*E)[20] = (double(*)[20])(lds + 14 * 15);
  double(*T1)[20] = (double(*)[20])(lds + 14 * 15 + 14 * 20);
  const int lane = threadIdx.x;
  const TRState *tr = &S->tr;
  const int cur = tr->cur;
  const Tab *T = &S->tab[cur];
  const int pair = S->chunk_pair[chunk];
  const int j = pair % 11;
  const int begin = S->chunk_begin[chunk], end = S->chunk_end[chunk];
  const int est_td = S->est_td;
  const double td = S->x[cur].td;
  PairU u;
  load_pair_uniform(T, pair, u);
  double q[NQ];
#pragma unroll
  for (int e = 0; e < NQ; e++) q[e] = 0.0;
  for (int idx = begin + lane; idx < end; idx += 64) {
    const int oj = S->pm_obs[idx], l = S->pm_lm[idx];
    const int oi = S->lm_obs0[l];
    const double lam = S->lam[cur][l];
    ObsPair ob;
    load_obs(S, oi, ob.pi, ob.vi, ob.tdi, ob.rowi);
    load_obs(S, oj, ob.pj, ob.vj, ob.tdj, ob.rowj);
    Basis B;
    visual_basis(ob, lam, td, est_td, S->tr_over_row, S->half_row, S->sqrt_info, u, B);
    double c0[14], c1[14];
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
    int e = 0;
#pragma unroll
    for (int p = 0; p < 14; p++)
#pragma unroll
      for (int r = p; r < 14; r++) {
        q[e] = fma(c0[p], c0[r], fma(c1[p], c1[r], q[e]));
        e++;
      }
  }
  // wave reduction (fixed butterfly order => deterministic)
#pragma unroll
  for (int e = 0; e < NQ; e++) q[e] = wave_sum(q[e]);
  if (lane == 0) {
    int e = 0;
#pragma unroll
    for (int p = 0; p < 14; p++)
#pragma unroll
      for (int r = p; r < 14; r++) {
        Qf[p][r] = q[e];
        Qf[r][p] = q[e];
        e++;
      }
  }
  // E: basis(14) -> factor columns(20) = [Pi th_i Pj th_j tic th_ic td r]
  for (int e = lane; e < 14 * 20; e += 64) E[e / 20][e % 20] = 0.0;
  __syncthreads();
  if (lane < 9) {
    const int r = lane / 3, c = lane % 3;
    const double m1 = T->M1[j][lane];
    E[r][c] = m1;                                  // dr/dPi  = red M1
    E[r][6 + c] = -m1;                             // dr/dPj  = -red M1
    E[r][12 + c] = T->M2[pair][lane] - T->ricT[lane];  // dr/dtic = red (M2 - ric^T)
  } else if (lane < 12) {
    const int k = lane - 9;
    E[3 + k][3 + k] = 1.0;
    E[6 + k][9 + k] = 1.0;
    E[9 + k][15 + k] = 1.0;
  } else if (lane == 12) {
    E[12][18] = 1.0;
    E[13][19] = 1.0;
  }
  __syncthreads();
  for (int e = lane; e < 14 * 20; e += 64) {
    const int r = e / 20, c = e % 20;
    double s = 0;
#pragma unroll
    for (int k = 0; k < 14; k++) s = fma(Qf[r][k], E[k][c], s);
    T1[r][c] = s;
  }
  __syncthreads();
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

DEV void lin_imu_role(Slot *S, int f, int mode, double *lds) {
  double(*Jr)[30] = (double(*)[30]) lds;
  double(*Jw)[31] = (double(*)[31])(lds + 450);
  double *rr = lds + 450 + 465, *rw = rr + 16;
  const int lane = threadIdx.x;
  double *out = S->imu_out + (size_t)f * IMU_OUT;
  const bool active = S->imu_active[f] && (mode == MODE_SOLVE || (f == 0 && marg_plan(S, mode)->use_imu0)) &&
                      (!S->sharded || S->pose_side);
  if (!active) {
    for (int e = lane; e < IMU_OUT; e += 64) out[e] = 0.0;
    return;
  }
  const FrameState *x = &S->x[S->tr.cur];
  for (int e = lane; e < 450; e += 64) Jr[e / 30][e % 30] = 0.0;
  __syncthreads();
  if (lane == 0) {
    imu_raw_residual(&S->imu[f], S->g, x->pose[f], x->sb[f], x->pose[f + 1], x->sb[f + 1], rr);
    imu_raw_jacobian(&S->imu[f], S->g, x->pose[f], x->sb[f], x->pose[f + 1], x->sb[f + 1], &Jr[0][0]);
  }
  __syncthreads();
  const double *Sq = S->imu_sqrt[f];
  if (lane < 15) {
    double s = 0;
    for (int k = 0; k < 15; k++) s = fma(Sq[lane * 15 + k], rr[k], s);
    rw[lane] = s;
  }
  for (int e = lane; e < 450; e += 64) {
    const int r = e / 30, c = e % 30;
    double s = 0;
    for (int k = r; k < 15; k++) s = fma(Sq[r * 15 + k], Jr[k][c], s);  // sqrt_info is upper triangular
    Jw[r][c] = s;
  }
  __syncthreads();
  for (int e = lane; e < 900; e += 64) {
    const int p = e / 30, c = e % 30;
    double s = 0;
#pragma unroll
    for (int k = 0; k < 15; k++) s = fma(Jw[k][p], Jw[k][c], s);
    out[e] = s;
  }
  if (lane < 30) {
    double s = 0;
#pragma unroll
    for (int k = 0; k < 15; k++) s = fma(Jw[k][lane], rw[k], s);
    out[900 + lane] = s;
  }
  if (lane == 32) {
    double s = 0;
    for (int k = 0; k < 15; k++) s = fma(rw[k], rw[k], s);
    out[930] = 0.5 * s;
  }
}

DEV void lin_prior_role(Slot *S, int mode, double *lds) {
  double *dx = lds, *r = lds + KP;
  const int lane = threadIdx.x;
  double *g = S->prior_g;
  for (int c = lane; c < KP + 4; c += 64) g[c] = 0.0;
  if (!S->prior_valid || (S->sharded && !S->pose_side)) return;
  const int n = S->prior_n;
  const FrameState *x = &S->x[S->tr.cur];
  if (lane < S->prior_nb) prior_block_dx(S, x, lane, dx);
  __syncthreads();
  const double *J = S->prior_J;
  for (int row = lane; row < n; row += 64) {
    double s = S->prior_r[row];
    for (int c = 0; c < n; c++) s = fma(J[row * n + c], dx[c], s);
    r[row] = s;
  }
  __syncthreads();
  for (int c = lane; c < n; c += 64) {
    double s = 0;
    for (int k = 0; k < n; k++) s = fma(J[k * n + c], r[k], s);
    g[S->prior_cmap[c]] = s;
  }
  double cs = 0;
  for (int row = lane; row < n; row += 64) cs += r[row] * r[row];
  cs = wave_sum(cs);
  if (lane == 0) g[KP] = 0.5 * cs;
}

__global__ __launch_bounds__(64) void k_lin(char *base, size_t stride, int mode, int gLm, int gCh) {
  Slot *S = SLOT(base, stride);
  const TRState *tr = &S->tr;
  if (tr->done || !tr->do_lin) return;
  __shared__ __attribute__((aligned(16))) double lds[LM_BLOCK * (WLD + 1)];  // one workspace, aliased per role
  // the grid is sized for the largest resident window (gLm, gCh); each slot uses its own counts
  int b = blockIdx.x;
  if (b < gLm) {
    if (b >= S->nLmBlocks) return;
    if (is_marg(mode) && b * LM_BLOCK >= marg_plan(S, mode)->N0) return;
    lin_landmark_role(S, b, mode, lds);
    return;
  }
  b -= gLm;
  if (b < gCh) {
    if (b >= S->nChunks) return;
    if (is_marg(mode) && b >= marg_plan(S, mode)->nChunks0) return;
    lin_gram_role(S, b, mode, lds);
    return;
  }
  b -= gCh;
  if (b < LFVIO_WINDOW_SIZE) {
    lin_imu_role(S, b, mode, lds);
    return;
  }
  lin_prior_role(S, mode, lds);
}

// ---------------------------------------------------------------------------
// k_schur: grid (nSchurParts, batch) x 64.  Sc = sum_l c_l w_l w_l^T over the 80-wide rows
// (cols 73/74 carry b_l and the Cauchy cross-term column), upper 15 tiles of 16x16, on the
// FP64 matrix pipe.  Lane l feeds A[i = l&15][k = l>>4] and B[k = l>>4][j = l&15]; the
// accumulator holds D[row = (l>>4) + 4*reg][col = l&15].
// ---------------------------------------------------------------------------
typedef double double4_t __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(64) void k_schur(char *base, size_t stride, int mode) {
  Slot *S = SLOT(base, stride);
  const TRState *tr = &S->tr;
  if (tr->done || !tr->do_schur) return;
  const int lane = threadIdx.x, kk = lane >> 4, cc = lane & 15;
  const int part = blockIdx.x;
  const int Nlim = is_marg(mode) ? marg_plan(S, mode)->N0 : S->N;
  const double mu = tr->mu;
  double4_t acc[NT];
#pragma unroll
  for (int t = 0; t < NT; t++) acc[t] = double4_t{0, 0, 0, 0};
  const int lm = S->schur_lm;
  const int l0 = part * lm;
  for (int s = 0; s < lm / 4; s++) {
    if (l0 + 4 * s >= Nlim) break;
    const int l = l0 + 4 * s + kk;
    double coef = 0.0, e = 0.0;
    double x[5] = {0, 0, 0, 0, 0};
    if (l < Nlim) {
      const double a = S->a[l];
      if (is_marg(mode)) {
        e = a;
        coef = (a > 1e-8) ? 1.0 / a : 0.0;  // eps of marginalization_factor.h:70 on the diagonal block
        if (cc == 0) S->einv_l[l] = coef;
      } else {
        const double sc = S->scale_l[l];
        const double s2a = sc * sc * a;
        const double D2 = fmin(fmax(s2a, 1e-6), 1e32);
        e = s2a + mu * D2;  // e-block + lm_diagonal^2
        const double einv = 1.0 / e;
        coef = sc * sc * einv;
        if (cc == 0) S->einv_l[l] = einv;
      }
      const double *row = S->W + (size_t)l * WLD + cc;
#pragma unroll
      for (int t = 0; t < 5; t++) x[t] = row[16 * t];
    }
    double bx4 = x[4];
    if (mode == MODE_SOLVE && cc == (COL_K - 64)) bx4 = x[4] * e;  // b/D2 * e  -> z2 column
    int ti = 0;
#pragma unroll
    for (int t = 0; t < 5; t++) {
      const double at = coef * x[t];
#pragma unroll
      for (int u = t; u < 5; u++) {
        const double bu = (u == 4) ? bx4 : x[u];
        acc[ti] = __builtin_amdgcn_mfma_f64_16x16x4f64(at, bu, acc[ti], 0, 0, 0);
        ti++;
      }
    }
  }
  double *out = S->schur_part + (size_t)part * SCHUR_LEN;
#pragma unroll
  for (int t = 0; t < NT; t++)
#pragma unroll
    for (int r = 0; r < 4; r++) out[t * 256 + r * 64 + lane] = acc[t][r];
}

// element (R, Cc) of the reduced 80x80 accumulator, R <= Cc
DEV int schur_index(int R, int Cc) {
  const int t = R >> 4, u = Cc >> 4;
  const int tile = t * 5 - (t * (t - 1)) / 2 + (u - t);
  const int row = R & 15, col = Cc & 15;
  return tile * 256 + (row >> 2) * 64 + ((row & 3) << 4) + col;
}
DEV double schur_get(const double *Sc, int r, int c) { return r <= c ? Sc[schur_index(r, c)] : Sc[schur_index(c, r)]; }

// ---------------------------------------------------------------------------
// k_sum: grid (HPP_BLOCKS + SCHUR_LEN/256 + 1, batch) x 256
//   role 1: every packed entry of the pose-side Gauss-Newton Hessian H_pp (and of g_p) is
//           produced by exactly ONE thread that adds its contributions in a fixed order —
//           Gram chunks of the frame pairs touching it, the (at most two) IMU factors, the
//           prior — so the result is deterministic and needs no atomics
//   role 2: fixed-order sum of the Schur SYRK partials
//   role 3: landmark scalar partials
// ---------------------------------------------------------------------------
constexpr int HPP_ITEMS = PACKED + KP;
constexpr int HPP_BLOCKS = (HPP_ITEMS + 255) / 256;

// frame block of a camera-side column: 0..10 pose, 11 ex, 12 td; lc = index inside the block
DEV void cam_block(int c, int &f, int &lc) {
  if (c < 66) {
    f = c / 6;
    lc = c - 6 * f;
  } else if (c < 72) {
    f = 11;
    lc = c - 66;
  } else {
    f = 12;
    lc = 0;
  }
}
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

DEV bool col_active(const Slot *S, int c, int mode) {
  if (mode >= MODE_MARG) return true;
  if (!S->est_ex && c >= off_ex() && c < off_ex() + 6) return false;
  if (!S->est_td && c == off_td()) return false;
  return true;
}

DEV double gram_pair_sum(const Slot *S, int i, int j, int idx, int chunk_limit) {
  const int p = i * 11 + j;
  const int c0 = S->pair_chunk0[p];
  int c1 = S->pair_chunk0[p + 1];
  if (c1 > chunk_limit) c1 = chunk_limit;
  const double *gp = S->gram_part + idx;
  double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
  int c = c0;
  for (; c + 4 <= c1; c += 4) {
    s0 += gp[(size_t)c * NGP], s1 += gp[(size_t)(c + 1) * NGP];
    s2 += gp[(size_t)(c + 2) * NGP], s3 += gp[(size_t)(c + 3) * NGP];
  }
  for (; c < c1; c++) s0 += gp[(size_t)c * NGP];
  return (s0 + s1) + (s2 + s3);
}

// the ex/td entries receive a term from EVERY chunk: chunks are contiguous in pair order, so this is a plain
// strided sweep with independent loads (4 accumulators, fixed association => still deterministic)
DEV double gram_all_pairs(const Slot *S, int idx, int chunk_limit) {
  const double *gp = S->gram_part + idx;
  double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
  int c = 0;
  for (; c + 4 <= chunk_limit; c += 4) {
    s0 += gp[(size_t)c * NGP];
    s1 += gp[(size_t)(c + 1) * NGP];
    s2 += gp[(size_t)(c + 2) * NGP];
    s3 += gp[(size_t)(c + 3) * NGP];
  }
  for (; c < chunk_limit; c++) s0 += gp[(size_t)c * NGP];
  return (s0 + s1) + (s2 + s3);
}

__global__ __launch_bounds__(256) void k_sum(char *base, size_t stride, int mode) {
  Slot *S = SLOT(base, stride);
  const TRState *tr = &S->tr;
  if (tr->done) return;
  const int tid = threadIdx.x;
  int b = blockIdx.x;
  if (b < HPP_BLOCKS) {
    if (!tr->do_lin) return;
    const int e = b * 256 + tid;
    if (e >= HPP_ITEMS) return;
    const int chunk_limit = is_marg(mode) ? marg_plan(S, mode)->nChunks0 : S->nChunks;
    double val = 0.0;
    if (e < PACKED) {
      int r = (int)((sqrt(8.0 * e + 1.0) - 1.0) * 0.5);
      while ((r + 1) * (r + 2) / 2 <= e) r++;
      while (r * (r + 1) / 2 > e) r--;
      const int c = e - r * (r + 1) / 2;
      if (col_active(S, r, mode) && col_active(S, c, mode)) {
        // ---- visual: Gram blocks of the frame pairs that contain both columns
        if (r < KC) {
          int fr, lr, fc, lc;
          cam_block(r, fr, lr);
          cam_block(c, fc, lc);
          if (fr < 11) {  // both pose blocks (c <= r => fc <= fr)
            if (fr == fc) {
              for (int j = fr + 1; j < 11; j++) val += gram_pair_sum(S, fr, j, gidx20(lc, lr), chunk_limit);
              for (int i = 0; i < fr; i++) val += gram_pair_sum(S, i, fr, gidx20(6 + lc, 6 + lr), chunk_limit);
            } else {
              val += gram_pair_sum(S, fc, fr, gidx20(lc, 6 + lr), chunk_limit);
            }
          } else {
            const int hi = fr == 11 ? 12 + lr : 18;
            if (fc < 11) {
              for (int j = fc + 1; j < 11; j++) val += gram_pair_sum(S, fc, j, gidx20(lc, hi), chunk_limit);
              for (int i = 0; i < fc; i++) val += gram_pair_sum(S, i, fc, gidx20(6 + lc, hi), chunk_limit);
            } else {
              const int lo = fc == 11 ? 12 + lc : 18;
              val += gram_all_pairs(S, gidx20(lo, hi), chunk_limit);
            }
          }
        }
        // ---- IMU factors covering both columns (at most two)
        const int f0 = col_frame(r);
        if (f0 >= 0) {
          for (int f = f0 - 1; f <= f0; f++) {
            if (f < 0 || f >= LFVIO_WINDOW_SIZE) continue;
            const int p = imu_local(r, f), q = imu_local(c, f);
            if (p >= 0 && q >= 0) val += S->imu_out[(size_t)f * IMU_OUT + p * 30 + q];
          }
        }
        // ---- prior: A' = J0^T J0
        if (S->prior_valid && (!S->sharded || S->pose_side)) {
          const int pr = S->prior_inv[r], pc = S->prior_inv[c];
          if (pr >= 0 && pc >= 0) val += S->prior_A[pr * S->prior_n + pc];
        }
      }
      S->Hpp[e] = val;
    } else {
      const int r = e - PACKED;
      if (col_active(S, r, mode)) {
        if (r < KC) {
          int fr, lr;
          cam_block(r, fr, lr);
          if (fr < 11) {
            for (int j = fr + 1; j < 11; j++) val += gram_pair_sum(S, fr, j, gidx20(lr, 19), chunk_limit);
            for (int i = 0; i < fr; i++) val += gram_pair_sum(S, i, fr, gidx20(6 + lr, 19), chunk_limit);
          } else {
            const int lo = fr == 11 ? 12 + lr : 18;
            val += gram_all_pairs(S, gidx20(lo, 19), chunk_limit);
          }
        }
        const int f0 = col_frame(r);
        if (f0 >= 0) {
          for (int f = f0 - 1; f <= f0; f++) {
            if (f < 0 || f >= LFVIO_WINDOW_SIZE) continue;
            const int p = imu_local(r, f);
            if (p >= 0) val += S->imu_out[(size_t)f * IMU_OUT + 900 + p];
          }
        }
        val += S->prior_g[r];
      }
      S->gp[r] = val;
    }
    return;
  }
  b -= HPP_BLOCKS;
  if (b < SCHUR_LEN / 256) {
    if (!tr->do_schur) return;
    const int e = b * 256 + tid;
    int parts = S->nSchurParts;
    if (is_marg(mode)) parts = (marg_plan(S, mode)->N0 + S->schur_lm - 1) / S->schur_lm;
    // fixed association (8 interleaved accumulators, then a fixed tree): deterministic, 8 loads in flight
    const double *sp = S->schur_part + e;
    double a0 = 0, a1 = 0, a2 = 0, a3 = 0, a4 = 0, a5 = 0, a6 = 0, a7 = 0;
    int p = 0;
    for (; p + 8 <= parts; p += 8) {
      a0 += sp[(size_t)p * SCHUR_LEN], a1 += sp[(size_t)(p + 1) * SCHUR_LEN];
      a2 += sp[(size_t)(p + 2) * SCHUR_LEN], a3 += sp[(size_t)(p + 3) * SCHUR_LEN];
      a4 += sp[(size_t)(p + 4) * SCHUR_LEN], a5 += sp[(size_t)(p + 5) * SCHUR_LEN];
      a6 += sp[(size_t)(p + 6) * SCHUR_LEN], a7 += sp[(size_t)(p + 7) * SCHUR_LEN];
    }
    for (; p < parts; p++) a0 += sp[(size_t)p * SCHUR_LEN];
    S->schur_sum[e] = ((a0 + a1) + (a2 + a3)) + ((a4 + a5) + (a6 + a7));
    return;
  }
  if (!tr->do_lin) return;
  int blocks = S->nLmBlocks;
  if (tid < 4) {
    double s = 0;
    for (int k = 0; k < blocks; k++) s += S->lm_part[(size_t)k * LMS + tid];
    S->lm_sum[tid] = s;
  } else if (tid == 4) {
    double m = 0;
    for (int k = 0; k < blocks; k++) m = fmax(m, S->lm_part[(size_t)k * LMS + 4]);
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
      if (S->pose_side) {
        cost += S->prior_g[KP];
        for (int f = 0; f < LFVIO_WINDOW_SIZE; f++) cost += S->imu_out[(size_t)f * IMU_OUT + 930];
      }
      sc[XS_COST] = cost;
      sc[XS_G2] = S->lm_sum[1], sc[XS_ASV2] = S->lm_sum[2], sc[XS_LAM2] = S->lm_sum[3], sc[XS_BMAX] = S->lm_sum[4];
    }
  }
}
