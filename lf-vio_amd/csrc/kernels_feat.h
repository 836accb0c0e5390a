// kernels_feat.h — producers and consumers either side of optimization().
//
//   k_triangulate : FeatureManager::triangulate            feature_manager.cpp:199-253
//   k_shift_depth : FeatureManager::removeBackShiftDepth   feature_manager.cpp:271-310 (the depth arithmetic)
//   k_preintegrate: IntegrationBase::push_back / propagate / midPointIntegration   factor/integration_base.h:29-158
// One thread per landmark.
#pragma once
#include "dev_math.h"

struct FeatFrames {  // frame poses + extrinsics of one call
  double Ps[LFVIO_NUM_FRAMES][3];
  double Rs[LFVIO_NUM_FRAMES][9];
  double tic[3], ric[9];
  double init_depth;
};

constexpr int TRI_THREADS = 64;
constexpr int TRI_ROWS = 2 * LFVIO_NUM_FRAMES;  // 22

DEV m33 mT(const m33 &a) {
  m33 r;
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++) r.a[3 * i + j] = a.a[3 * j + i];
  return r;
}

// grid ceil(N / 64) x 64.  The 2k x 4 system (k <= 11) stays in registers: every loop over rows / observations is fully
// unrolled to the maximum and predicated on the landmark's own count, so there is no indexed private array.
__global__ __launch_bounds__(TRI_THREADS) void k_triangulate(const FeatFrames *F, int N, const int *start_frame, const int *obs_offset,
                                                            const double *obs_point, double *depth) {
  const int l = blockIdx.x * TRI_THREADS + threadIdx.x;
  if (l >= N) return;
  if (depth[l] > 0.0) return;  // :207
  const int imu_i = start_frame[l], o0 = obs_offset[l], k = obs_offset[l + 1] - o0;
  const m33 ric = ldm(F->ric);
  const d3 tic = ld3(F->tic);
  const d3 t0 = ld3(F->Ps[imu_i]) + mul(ldm(F->Rs[imu_i]), tic);  // :216
  const m33 R0T = mT(mm(ldm(F->Rs[imu_i]), ric));
  double A[TRI_ROWS][4];
#pragma unroll
  for (int o = 0; o < LFVIO_NUM_FRAMES; o++) {
#pragma unroll
    for (int c = 0; c < 4; c++) A[2 * o][c] = A[2 * o + 1][c] = 0.0;
    if (o < k) {
      const int imu_j = imu_i + o;
      const d3 t1 = ld3(F->Ps[imu_j]) + mul(ldm(F->Rs[imu_j]), tic);
      const m33 R1 = mm(ldm(F->Rs[imu_j]), ric);
      const d3 t = mul(R0T, t1 - t0);
      const m33 Rt = mT(mm(R0T, R1));  // P = [R^T | -R^T t]
      const d3 mt = -1.0 * mul(Rt, t);
      const d3 p = ld3(obs_point + 3 * (size_t)(o0 + o));
      const double pn = sqrt(p.x * p.x + p.y * p.y + p.z * p.z);
      const d3 f = mk3(p.x / pn, p.y / pn, p.z / pn);  // :235  normalized()
      const double P0[4] = {Rt.a[0], Rt.a[1], Rt.a[2], mt.x}, P1[4] = {Rt.a[3], Rt.a[4], Rt.a[5], mt.y},
                   P2[4] = {Rt.a[6], Rt.a[7], Rt.a[8], mt.z};
#pragma unroll
      for (int c = 0; c < 4; c++) {
        A[2 * o][c] = f.x * P2[c] - f.z * P0[c];      // :236
        A[2 * o + 1][c] = f.y * P2[c] - f.z * P1[c];  // :237
      }
    }
  }
  // one-sided Jacobi SVD of the 2k x 4 matrix: orthogonalize the columns, accumulate V.  Rows beyond 2k are zero and
  // add exact zeros to the sums, so the arithmetic is the one of a 2k-row loop.
  double V[4][4];
#pragma unroll
  for (int i = 0; i < 4; i++)
#pragma unroll
    for (int j = 0; j < 4; j++) V[i][j] = i == j ? 1.0 : 0.0;
  for (int sweep = 0; sweep < 30; sweep++) {
    bool rotated = false;
#pragma unroll
    for (int p = 0; p < 3; p++)
#pragma unroll
      for (int q = p + 1; q < 4; q++) {
        double al = 0, be = 0, ga = 0;
#pragma unroll
        for (int r = 0; r < TRI_ROWS; r++) {
          const double x = A[r][p], y = A[r][q];
          al += x * x, be += y * y, ga += x * y;
        }
        if (!(ga == 0.0 || fabs(ga) <= 1e-15 * sqrt(al * be))) {
          rotated = true;
          const double zeta = (be - al) / (2.0 * ga);
          const double t = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
          const double c = 1.0 / sqrt(1.0 + t * t), s = c * t;
#pragma unroll
          for (int r = 0; r < TRI_ROWS; r++) {
            const double x = A[r][p], y = A[r][q];
            A[r][p] = c * x - s * y;
            A[r][q] = s * x + c * y;
          }
#pragma unroll
          for (int r = 0; r < 4; r++) {
            const double x = V[r][p], y = V[r][q];
            V[r][p] = c * x - s * y;
            V[r][q] = s * x + c * y;
          }
        }
      }
    if (!rotated) break;
  }
  // the column with the smallest norm belongs to the smallest singular value (Eigen: matrixV().rightCols<1>())
  double v0 = 0, v1 = 0, v2 = 0, v3 = 0, bn = 0;
#pragma unroll
  for (int c = 0; c < 4; c++) {
    double n2 = 0;
#pragma unroll
    for (int r = 0; r < TRI_ROWS; r++) n2 += A[r][c] * A[r][c];
    if (c == 0 || n2 < bn) bn = n2, v0 = V[0][c], v1 = V[1][c], v2 = V[2][c], v3 = V[3][c];
  }
  const d3 X = mk3(v0 / v3, v1 / v3, v2 / v3);         // :246
  double d = dot(X, ld3(obs_point + 3 * (size_t)o0));  // :247
  if (d < 0) d = F->init_depth;                         // :249-252
  depth[l] = d;
}

// grid ceil(n / 256) x 256; T = [marg_R(9) marg_P(3) new_R(9) new_P(3) init_depth]
__global__ __launch_bounds__(256) void k_shift_depth(int n, const double *uv_i, const double *T, double *depth) {
  const int l = blockIdx.x * 256 + threadIdx.x;
  if (l >= n) return;
  const m33 mR = ldm(T), nRT = mT(ldm(T + 12));
  const d3 pts_i = depth[l] * ld3(uv_i + 3 * (size_t)l);     // :292
  const d3 w = mul(mR, pts_i) + ld3(T + 9);
  const d3 pj = mul(nRT, w - ld3(T + 21));
  const double dep = sqrt(dot(pj, pj));  // :296, the range ("changed by wz")
  depth[l] = dep > 0 ? dep : T[24];
}

// ---------------------------------------------------------------------------------------------------------------
// IMU pre-integration: one workgroup per keyframe interval, the samples of an interval are a serial
// chain (20 at 200 Hz / 10 Hz); inside a step lane 0 advances the mid-point state and fills F (15x15) and V (15x18),
// then 225 threads form  jacobian <- F jacobian  and  covariance <- F cov F^T + V noise V^T  (integration_base.h:124-125)
// one output entry each.  Sums run k = 0, 1, ... like the restatement in oracle/ (no fused multiply-add: -ffp-contract=off).
// ---------------------------------------------------------------------------------------------------------------
struct ImuJob {
  int n, off;  // samples [off, off + n) of the packed dt / acc / gyr arrays
  double acc_0[3], gyr_0[3], ba[3], bg[3];
};

DEV m33 msc(double s, const m33 &a) {
  m33 r;
#pragma unroll
  for (int e = 0; e < 9; e++) r.a[e] = s * a.a[e];
  return r;
}
DEV m33 madd(const m33 &a, const m33 &b) {
  m33 r;
#pragma unroll
  for (int e = 0; e < 9; e++) r.a[e] = a.a[e] + b.a[e];
  return r;
}
DEV m33 msub(const m33 &a, const m33 &b) {
  m33 r;
#pragma unroll
  for (int e = 0; e < 9; e++) r.a[e] = a.a[e] - b.a[e];
  return r;
}
DEV m33 mneg(const m33 &a) { return msc(-1.0, a); }
DEV m33 mmul_plain(const m33 &a, const m33 &b) {  // a b without fma, like the restatement
  m33 r;
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++) r.a[3 * i + j] = a.a[3 * i] * b.a[j] + a.a[3 * i + 1] * b.a[3 + j] + a.a[3 * i + 2] * b.a[6 + j];
  return r;
}
DEV m33 meye() {
  m33 r = skewm(mk3(0, 0, 0));
  r.a[0] = r.a[4] = r.a[8] = 1.0;
  return r;
}
DEV void put_blk(double *M, int ld, int r0, int c0, const m33 &B) {
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++) M[(r0 + i) * ld + c0 + j] = B.a[3 * i + j];
}

// grid (intervals) x 256
__global__ __launch_bounds__(256) void k_preintegrate(const ImuJob *jobs, const double *dts, const double *accs, const double *gyrs,
                                                     const double *noise4, LfvioPreintegration *out) {
  __shared__ double J[225], P[225], F[225], V[270], FJ[225], FC[225], nd[18];
  __shared__ double st[16];  // delta_p(3) delta_q(w x y z) delta_v(3) acc_0(3) gyr_0(3)
  const int tid = threadIdx.x;
  const ImuJob *jb = &jobs[blockIdx.x];
  if (tid < 225) J[tid] = (tid / 15 == tid % 15) ? 1.0 : 0.0, P[tid] = 0.0;
  if (tid < 18) {
    const int b = tid / 3;
    const double an = noise4[0], gn = noise4[1], aw = noise4[2], gw = noise4[3];
    nd[tid] = b == 0 || b == 2 ? an * an : b == 1 || b == 3 ? gn * gn : b == 4 ? aw * aw : gw * gw;  // integration_base.h:21-27
  }
  if (tid == 0) {
    st[0] = st[1] = st[2] = 0.0, st[3] = 1.0, st[4] = st[5] = st[6] = 0.0, st[7] = st[8] = st[9] = 0.0;
    for (int k = 0; k < 3; k++) st[10 + k] = jb->acc_0[k], st[13 + k] = jb->gyr_0[k];
  }
  double sum_dt = 0.0;
  __syncthreads();
  for (int sidx = 0; sidx < jb->n; sidx++) {
    for (int e = tid; e < 225; e += 256) F[e] = 0.0;
    for (int e = tid; e < 270; e += 256) V[e] = 0.0;
    __syncthreads();
    const double dt = dts[jb->off + sidx];
    if (tid == 0) {
      // midPointIntegration, integration_base.h:54-128
      const d3 acc0 = ld3(st + 10), gyr0 = ld3(st + 13), acc1 = ld3(accs + 3 * (size_t)(jb->off + sidx)),
               gyr1 = ld3(gyrs + 3 * (size_t)(jb->off + sidx));
      const d3 ba = ld3(jb->ba), bg = ld3(jb->bg);
      const q4 dq = q4{st[3], st[4], st[5], st[6]};
      const d3 dp = ld3(st), dv = ld3(st + 7);
      const d3 un_acc_0 = qrot(dq, acc0 - ba);                                                   // :63
      const d3 un_gyr = 0.5 * (gyr0 + gyr1) - bg;                                                // :64
      const q4 rdq = qmul(dq, q4{1.0, un_gyr.x * dt / 2, un_gyr.y * dt / 2, un_gyr.z * dt / 2});  // :65 (unnormalized)
      const d3 un_acc_1 = qrot(rdq, acc1 - ba);                                                  // :66
      const d3 un_acc = 0.5 * (un_acc_0 + un_acc_1);                                             // :67
      const d3 rdp = dp + dv * dt + 0.5 * un_acc * dt * dt;                                      // :68
      const d3 rdv = dv + un_acc * dt;                                                           // :69
      // F and V, :73-120
      const d3 w_x = 0.5 * (gyr0 + gyr1) - bg, a_0_x = acc0 - ba, a_1_x = acc1 - ba;
      const m33 R_w_x = skewm(w_x), R_a_0_x = skewm(a_0_x), R_a_1_x = skewm(a_1_x);
      const m33 Rdq = q2R(dq), Rrdq = q2R(rdq), I = meye();
      const m33 ImW = msub(I, msc(dt, R_w_x));
      const m33 RA1 = mmul_plain(Rrdq, R_a_1_x), RA0 = mmul_plain(Rdq, R_a_0_x);
      put_blk(F, 15, 0, 0, I);
      put_blk(F, 15, 0, 3, madd(msc(dt, msc(dt, msc(-0.25, RA0))), msc(dt, msc(dt, mmul_plain(msc(-0.25, RA1), ImW)))));
      put_blk(F, 15, 0, 6, msc(dt, I));
      put_blk(F, 15, 0, 9, msc(dt, msc(dt, msc(-0.25, madd(Rdq, Rrdq)))));
      put_blk(F, 15, 0, 12, msc(-dt, msc(dt, msc(dt, msc(-0.25, RA1)))));
      put_blk(F, 15, 3, 3, ImW);
      put_blk(F, 15, 3, 12, msc(dt, msc(-1.0, I)));
      put_blk(F, 15, 6, 3, madd(msc(dt, msc(-0.5, RA0)), msc(dt, mmul_plain(msc(-0.5, RA1), ImW))));
      put_blk(F, 15, 6, 6, I);
      put_blk(F, 15, 6, 9, msc(dt, msc(-0.5, madd(Rdq, Rrdq))));
      put_blk(F, 15, 6, 12, msc(-dt, msc(dt, msc(-0.5, RA1))));
      put_blk(F, 15, 9, 9, I);
      put_blk(F, 15, 12, 12, I);
      const m33 nRA1 = mmul_plain(mneg(Rrdq), R_a_1_x);
      const m33 V03 = msc(dt, msc(0.5, msc(dt, msc(dt, msc(0.25, nRA1)))));
      const m33 V63 = msc(dt, msc(0.5, msc(dt, msc(0.5, nRA1))));
      put_blk(V, 18, 0, 0, msc(dt, msc(dt, msc(0.25, Rdq))));
      put_blk(V, 18, 0, 3, V03);
      put_blk(V, 18, 0, 6, msc(dt, msc(dt, msc(0.25, Rrdq))));
      put_blk(V, 18, 0, 9, V03);
      put_blk(V, 18, 3, 3, msc(dt, msc(0.5, I)));
      put_blk(V, 18, 3, 9, msc(dt, msc(0.5, I)));
      put_blk(V, 18, 6, 0, msc(dt, msc(0.5, Rdq)));
      put_blk(V, 18, 6, 3, V63);
      put_blk(V, 18, 6, 6, msc(dt, msc(0.5, Rrdq)));
      put_blk(V, 18, 6, 9, V63);
      put_blk(V, 18, 9, 12, msc(dt, I));
      put_blk(V, 18, 12, 15, msc(dt, I));
      // propagate, :130-158
      const q4 nq = qnormalized(rdq);
      st[0] = rdp.x, st[1] = rdp.y, st[2] = rdp.z;
      st[3] = nq.w, st[4] = nq.x, st[5] = nq.y, st[6] = nq.z;
      st[7] = rdv.x, st[8] = rdv.y, st[9] = rdv.z;
      st[10] = acc1.x, st[11] = acc1.y, st[12] = acc1.z, st[13] = gyr1.x, st[14] = gyr1.y, st[15] = gyr1.z;
    }
    sum_dt += dt;
    __syncthreads();
    if (tid < 225) {
      const int i = tid / 15, j = tid % 15;
      double a = 0, b = 0;
      for (int k = 0; k < 15; k++) a += F[i * 15 + k] * J[k * 15 + j], b += F[i * 15 + k] * P[k * 15 + j];
      FJ[tid] = a, FC[tid] = b;
    }
    __syncthreads();
    if (tid < 225) {
      const int i = tid / 15, j = tid % 15;
      double a = 0, b = 0;
      for (int k = 0; k < 15; k++) a += FC[i * 15 + k] * F[j * 15 + k];
      for (int k = 0; k < 18; k++) b += (V[i * 18 + k] * nd[k]) * V[j * 18 + k];
      J[tid] = FJ[tid], P[tid] = a + b;
    }
    __syncthreads();
  }
  LfvioPreintegration *o = &out[blockIdx.x];
  if (tid < 225) o->jacobian[tid] = J[tid], o->covariance[tid] = P[tid];
  if (tid == 0) {
    o->sum_dt = sum_dt;
    for (int k = 0; k < 3; k++) o->delta_p[k] = st[k], o->delta_v[k] = st[7 + k], o->linearized_ba[k] = jb->ba[k], o->linearized_bg[k] = jb->bg[k];
    o->delta_q[0] = st[4], o->delta_q[1] = st[5], o->delta_q[2] = st[6], o->delta_q[3] = st[3];
  }
}
