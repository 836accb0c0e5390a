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

constexpr int MARG_THREADS = 1024;  // 4 waves / SIMD: the Jacobi steps are LDS-latency bound
constexpr int MARG_MAXD = 96;  // 15 dropped + 76 kept, padded
constexpr size_t MARG_LDS = (size_t)17408 * sizeof(double);  // >= LPACK + KP + 64 and >= the eigen-phase carve below

__global__ __launch_bounds__(128) void k_gauge(char *base, size_t stride) {
  Slot *S = SLOT(base, stride);
  TRState *ts = &S->tr;
  const int tid = threadIdx.x;
  __shared__ double rot[9], P0[3], oP0[3];
  FrameState *x = &S->x[ts->cur];
  if (tid == 0) {
    // rot_diff from the yaw difference of frame 0 before / after the solve (estimator.cpp:534-560)
    auto R2ypr = [](const m33 &R, double *ypr) {
      const double n0 = R.a[0], n1 = R.a[3], n2 = R.a[6];
      const double o0 = R.a[1], o1 = R.a[4];
      const double a0 = R.a[2], a1 = R.a[5];
      const double y = atan2(n1, n0);
      const double p = atan2(-n2, n0 * cos(y) + n1 * sin(y));
      const double r = atan2(a0 * sin(y) - a1 * cos(y), -o0 * sin(y) + o1 * cos(y));
      ypr[0] = y / M_PI * 180.0, ypr[1] = p / M_PI * 180.0, ypr[2] = r / M_PI * 180.0;
    };
    const m33 Rs0 = q2R(q_from_pose(S->x0.pose[0]));
    const m33 R00 = q2R(q_from_pose(x->pose[0]));
    double o0[3], o00[3];
    R2ypr(Rs0, o0);
    R2ypr(R00, o00);
    const double yd = (o0[0] - o00[0]) / 180.0 * M_PI;
    m33 rd;  // ypr2R(y_diff, 0, 0) = Rz(y) Ry(0) Rx(0)
    rd.a[0] = cos(yd), rd.a[1] = -sin(yd), rd.a[2] = 0;
    rd.a[3] = sin(yd), rd.a[4] = cos(yd), rd.a[5] = 0;
    rd.a[6] = 0, rd.a[7] = 0, rd.a[8] = 1;
    if (fabs(fabs(o0[1]) - 90) < 1.0 || fabs(fabs(o00[1]) - 90) < 1.0) rd = mm(Rs0, tr(R00));
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
  if (tid < 11) {
    x->pose[tid][0] = Psi.x, x->pose[tid][1] = Psi.y, x->pose[tid][2] = Psi.z;
    x->pose[tid][3] = qn.x, x->pose[tid][4] = qn.y, x->pose[tid][5] = qn.z, x->pose[tid][6] = qn.w;
    x->sb[tid][0] = Vsi.x, x->sb[tid][1] = Vsi.y, x->sb[tid][2] = Vsi.z;
  } else if (tid == 11) {
    x->ex[3] = qn.x, x->ex[4] = qn.y, x->ex[5] = qn.z, x->ex[6] = qn.w;
  }
  // setDepth / getDepthVector round trip (feature_manager.cpp:148,191)
  double *lam = S->lam[ts->cur];
  for (int l = tid; l < S->N; l += 128) lam[l] = 1.0 / (1.0 / lam[l]);
  __syncthreads();
  if (tid == 0) {
    ts->done = 0;
    ts->do_lin = 1;
    ts->do_schur = 1;
    ts->chol_fail = 0;
  }
  build_tab(x, &S->tab[ts->cur], tid);
}

// Parallel cyclic Jacobi eigen-decomposition of the symmetric n x n matrix A (LDS, ld = n).
// On exit diag(A) holds the eigenvalues and the columns of V (LDS, ld = n) the eigenvectors.
// Round-robin ordering: np/2 disjoint rotations per step, np-1 steps per sweep; every 2x2 block
// A[{p_a,q_a}][{p_b,q_b}] is touched by exactly one thread (J_a^T B J_b), so a step needs two
// barriers and no atomics.  Threads are mapped 16 x 16 over (pair a, pair b): no integer division.
// Stops when the off-diagonal mass is below 1e-24 of the diagonal mass (off/||A|| < 1e-12, below the
// conditioning error of A' itself, which is ~1e-10: A_mm carries IMU information ~1e10) or stagnates.
DEV int jacobi_eig(double *A, double *V, int n, int tid, int nthreads, double *rotc, double *rots, int *rp, int *rq,
                   double *scratch) {
  for (int e = tid; e < n * n; e += nthreads) V[e] = 0.0;
  __syncthreads();
  for (int e = tid; e < n; e += nthreads) V[e * n + e] = 1.0;
  __syncthreads();
  const int np = n + (n & 1);
  const int half = np / 2;
  if (n < 2) return 0;
  int sweeps = 0;
  const int tx = tid & 15, ty = tid >> 4, nty = nthreads >> 4;
  double prev_off = 1e300;
  for (int sweep = 0; sweep < 24; sweep++) {
    double off = 0, dia = 0;
    for (int r = ty; r < n; r += nty)
      for (int c = tx; c < n; c += 16) {
        const double v = A[r * n + c];
        if (r == c) dia += v * v;
        else off += v * v;
      }
    off = wave_sum(off), dia = wave_sum(dia);
    __syncthreads();
    if ((tid & 63) == 0) scratch[tid >> 6] = off, scratch[16 + (tid >> 6)] = dia;
    __syncthreads();
    double so = 0, sd = 0;
    for (int w = 0; w < nthreads / 64; w++) so += scratch[w], sd += scratch[16 + w];
    __syncthreads();
    if (so <= 1e-24 * sd || so == 0.0) break;
    if (sweep >= 4 && so > 0.25 * prev_off) break;  // rounding floor reached
    prev_off = so;
    sweeps++;
    for (int step = 0; step < np - 1; step++) {
      if (tid < half) {
        int p, q;
        if (tid == 0) {
          p = np - 1, q = step;
        } else {
          p = step + tid;
          if (p >= np - 1) p -= np - 1;
          q = step - tid;
          if (q < 0) q += np - 1;
        }
        if (p > q) {
          int t = p;
          p = q, q = t;
        }
        double c = 1.0, s = 0.0;
        if (q < n) {
          const double apq = A[p * n + q];
          if (apq != 0.0) {
            const double theta = (A[q * n + q] - A[p * n + p]) / (2.0 * apq);
            const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
            c = rsqrt(t * t + 1.0);
            s = t * c;
          }
        }
        rp[tid] = p, rq[tid] = (q < n) ? q : -1;
        rotc[tid] = c, rots[tid] = s;
      }
      __syncthreads();
      // A <- J^T A J on disjoint 2x2 blocks (q = -1: the dummy index of an odd n)
      for (int ka = ty; ka < half; ka += nty) {
        const int pa = rp[ka], qa = rq[ka];
        const double ca = rotc[ka], sa = rots[ka];
        double *rowp = A + pa * n;
        double *rowq = A + (qa >= 0 ? qa : pa) * n;
        for (int kb = tx; kb < half; kb += 16) {
          const int pb = rp[kb], qb = rq[kb];
          const double cb = rotc[kb], sb = rots[kb];
          const bool va = qa >= 0, vb = qb >= 0;
          const double b00 = rowp[pb];
          const double b01 = vb ? rowp[qb] : 0.0;
          const double b10 = va ? rowq[pb] : 0.0;
          const double b11 = (va && vb) ? rowq[qb] : 0.0;
          const double t00 = cb * b00 - sb * b01, t01 = sb * b00 + cb * b01;
          const double t10 = cb * b10 - sb * b11, t11 = sb * b10 + cb * b11;
          rowp[pb] = ca * t00 - sa * t10;
          if (vb) rowp[qb] = ca * t01 - sa * t11;
          if (va) rowq[pb] = sa * t00 + ca * t10;
          if (va && vb) rowq[qb] = sa * t01 + ca * t11;
        }
      }
      // V <- V J
      for (int r = ty; r < n; r += nty) {
        double *vr = V + r * n;
        for (int kb = tx; kb < half; kb += 16) {
          const int pb = rp[kb], qb = rq[kb];
          if (qb < 0) continue;
          const double cb = rotc[kb], sb = rots[kb];
          const double v0 = vr[pb], v1 = vr[qb];
          vr[pb] = cb * v0 - sb * v1;
          vr[qb] = sb * v0 + cb * v1;
        }
      }
      __syncthreads();
    }
  }
  return sweeps;
}

// grid (1, batch) x 256, dynamic LDS = MARG_LDS
__global__ __launch_bounds__(MARG_THREADS) void k_marg_solve(char *base, size_t stride, int flag) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  Slot *S = SLOT(base, stride);
  TRState *tr = &S->tr;
  const MargPlan *mp = &S->marg[flag];
  const int tid = threadIdx.x;
  LfvioPrior *out = &S->prior_out;
  if (!mp->valid) {
    // MARGIN_SECOND_NEW with no prior touching Pose[WINDOW_SIZE-1]: the prior is left as it is
    if (tid == 0) out->valid = -1;  // host copies the input prior through
    return;
  }
  double *Hs = smem;            // PACKED, then reused: A (D x D), V (n x n), ...
  double *g = Hs + LPACK;       // KP
  double *scratch = g + KP;     // 64
  for (int e = tid; e < PACKED; e += MARG_THREADS) Hs[e] = S->Hpp[e];
  for (int c = tid; c < KP; c += MARG_THREADS) g[c] = S->gp[c];
  __syncthreads();
  // eliminate the frame-0 landmarks: H -= sum c_l w_l w_l^T, g -= sum c_l b_l w_l
  const double *Sc = S->schur_sum;
  if (mp->N0 > 0) {
    for (int e = tid; e < KC * (KC + 1) / 2; e += MARG_THREADS) {
      int r = (int)((sqrt(8.0 * e + 1.0) - 1.0) * 0.5);
      while ((r + 1) * (r + 2) / 2 <= e) r++;
      while (r * (r + 1) / 2 > e) r--;
      const int c = e - r * (r + 1) / 2;
      Hs[e] -= Sc[schur_index(c, r)];
    }
    for (int c = tid; c < KC; c += MARG_THREADS) g[c] -= Sc[schur_index(c, COL_B)];
  }
  __syncthreads();
  // ---- gather the dense system over the present blocks: D = m15 + n
  const int m15 = mp->m15, n = mp->n, D = m15 + n;
  double *Ag = S->mscr;  // global scratch, D x D + D
  for (int e = tid; e < KP * KP; e += MARG_THREADS) {
    const int r = e / KP, c = e % KP;
    const int ar = mp->col[r], ac = mp->col[c];
    if (ar >= 0 && ac >= 0) Ag[ar * D + ac] = Hs[pidx(r, c)];
  }
  for (int c = tid; c < KP; c += MARG_THREADS)
    if (mp->col[c] >= 0) Ag[D * D + mp->col[c]] = g[c];
  __syncthreads();
  // LDS re-use (16.7k doubles): A dies once A' is formed, so the second eigenvector matrix aliases it
  double *A = smem;                        // D x D           (<= 92*92 = 8464)
  double *V2 = smem;                       // n x n, aliases A after the Schur step
  double *bv = A + 92 * 92;                // D
  double *Am = bv + 96;                    // m15 x m15 (then its eigenvalues on the diagonal)
  double *Vm = Am + 256;                   // m15 x m15
  double *Ainv = Vm + 256;                 // m15 x m15
  double *Tm = Ainv + 256;                 // n x m15
  double *Ar = Tm + 80 * 16;               // n x n
  double *br = Ar + 76 * 76;               // n
  double *rotc = br + 80, *rots = rotc + 48;
  int *rp = (int *)(rots + 48), *rq = rp + 48;
  double *scr = (double *)(rq + 48);
  for (int e = tid; e < D * D; e += MARG_THREADS) A[e] = Ag[e];
  for (int c = tid; c < D; c += MARG_THREADS) bv[c] = Ag[D * D + c];
  __syncthreads();
  // ---- A_mm pseudo-inverse by eigen-decomposition (marginalization_factor.cpp:267-272)
  for (int e = tid; e < m15 * m15; e += MARG_THREADS) {
    const int r = e / m15, c = e % m15;
    Am[e] = 0.5 * (A[r * D + c] + A[c * D + r]);
  }
  __syncthreads();
  const int sw1 = jacobi_eig(Am, Vm, m15, tid, MARG_THREADS, rotc, rots, rp, rq, scr);
  __syncthreads();
  const double eps = 1e-8;
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
  // ---- A' = Arr - Arm Amm^+ Amr, b' = brr - Arm Amm^+ bmm  (:275-281)
  for (int e = tid; e < n * m15; e += MARG_THREADS) {
    const int r = e / m15, c = e % m15;
    double s = 0;
    for (int k = 0; k < m15; k++) s = fma(A[(m15 + r) * D + k], Ainv[k * m15 + c], s);
    Tm[e] = s;
  }
  __syncthreads();
  for (int e = tid; e < n * n; e += MARG_THREADS) {
    const int r = e / n, c = e % n;
    double s = 0;
    for (int k = 0; k < m15; k++) s = fma(Tm[r * m15 + k], A[k * D + m15 + c], s);
    Ar[e] = A[(m15 + r) * D + m15 + c] - s;
  }
  for (int r = tid; r < n; r += MARG_THREADS) {
    double s = 0;
    for (int k = 0; k < m15; k++) s = fma(Tm[r * m15 + k], bv[k], s);
    br[r] = bv[m15 + r] - s;
  }
  __syncthreads();
  // keep A', b' for parity checks (global scratch after the gathered system)
  double *Aout = Ag + 92 * 92 + 96;
  for (int e = tid; e < n * n; e += MARG_THREADS) Aout[e] = Ar[e];
  for (int r = tid; r < n; r += MARG_THREADS) Aout[n * n + r] = br[r];
  __syncthreads();
  // ---- second eigen-decomposition -> J0 = sqrt(S) V^T, r0 = sqrt(1/S) V^T b'  (:283-291)
  const int sw2 = jacobi_eig(Ar, V2, n, tid, MARG_THREADS, rotc, rots, rp, rq, scr);
  if (tid == 0) S->dbg[24] = sw1, S->dbg[25] = sw2;
  __syncthreads();
  for (int e = tid; e < n * n; e += MARG_THREADS) {
    const int k = e / n, i = e % n;
    const double ev = Ar[k * n + k];
    out->linearized_jacobians[e] = (ev > eps) ? sqrt(ev) * V2[i * n + k] : 0.0;
  }
  for (int k = tid; k < n; k += MARG_THREADS) {
    const double ev = Ar[k * n + k];
    double vb = 0;
    for (int i = 0; i < n; i++) vb = fma(V2[i * n + k], br[i], vb);
    out->linearized_residuals[k] = (ev > eps) ? sqrt(1.0 / ev) * vb : 0.0;
  }
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
  if (tid == 0) {
    out->valid = 1;
    out->m = m15 + mp->N0;
    out->n = n;
    out->num_blocks = mp->nb;
  }
}
