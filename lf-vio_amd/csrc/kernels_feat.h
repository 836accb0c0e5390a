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
// IMU pre-integration: one workgroup of eight waves per keyframe interval.  The samples of an interval
// form a serial chain, but only three small things are truly serial: the quaternion chain (F and V of step k depend on
// delta_q before and after the step, never on delta_p / delta_v or on the matrices), the delta_p / delta_v running sums,
// and the two matrix recursions  jacobian <- F_k jacobian,  covariance <- F_k covariance F_k^T + V_k noise V_k^T
// (integration_base.h:124-125).  Samples are taken PRE_TILE at a time, their dt / acc / gyr travel global -> registers (one
// tile ahead) -> 16-entry LDS rings (entry 8 t is acc_0 / gyr_0 of tile t, so nothing is carried over by hand), and an
// iteration has two phases, two barriers:
//
//   phase 1   waves 2 and 3, lane s = sample s of tile t: the quaternion chain as a three-step prefix scan of quaternion
//             products over the lanes (DPP row shifts; |a (x) b| = |a| |b|, so the normalizations of :134 are taken
//             afterwards, per lane), then the 3x3 ingredients — wave 2 the delta_q side (R(dq), R(dq)[a0]x, dq a0),
//             wave 3 the result_delta_q side (R(rdq), R(rdq)[a1]x, I - [w]x dt, their product, R(dq) + R(rdq), rdq a1) (:63-67, :76-86)
//             WHILE waves 0 and 1 run the matrix recursions of tile t-1 and one lane of wave 4 adds up its delta_p / delta_v (:68-69)
//   phase 2   wave s forms Q_s = V_s noise V_s^T of sample s of tile t on the FP64 matrix pipe (five v_mfma_f64_16x16x4_f64)
//
// Every entry of F_k and V_k is  c_a dt^pa A[e] + c_b dt^pb B[e]  for two ingredients A, B picked by its 3x3 block from a
// 55-row table (:88-120); a lane resolves the table ONCE for the entries it feeds to the matrix pipe (F[c][g + 4 kb] for
// lane 16 g + c) and then only reads ingredients: F and V are never stored.  The recursions live in accumulator registers
// and never touch LDS or a barrier: the MFMA result layout (row g + 4 r, column c) IS the B-operand layout of the next
// product, so with A = F (the lane's four entries)
//     jacobian' = F jacobian                  4 MFMAs   (wave 1)
//     G = covariance F^T                      4 MFMAs   (A = covariance read through its symmetry, B = F^T = the same four entries)
//     covariance' = F G + Q                   4 MFMAs   (wave 0, accumulator preloaded with Q)
// v_mfma_f64_16x16x4_f64 is pipe-bound on gfx950 (64 cycles each, dependent or not: tools/micro/mfma64.hip), which makes
// the covariance recursion (8 per sample) the critical path: ~0.4 us per sample, 17 us for a 10 x 20-sample window.
// Results agree with the restatement in oracle/ to ~1e-15 relative (fused multiply-adds and a reciprocal square root in
// the quaternion normalization differ from it in the last bits); tests/test_preintegration.py holds 1e-12.
// ---------------------------------------------------------------------------------------------------------------
struct ImuJob {
  int n, off;  // samples [off, off + n) of the packed dt / acc / gyr arrays
  double acc_0[3], gyr_0[3], ba[3], bg[3];
};

constexpr int PRE_TILE = 8, PRE_THREADS = 64 * PRE_TILE;
enum { PM_I, PM_RDQ, PM_RRDQ, PM_RA0, PM_RA1, PM_RA1W, PM_IMW, PM_RSUM, PM_COUNT };
struct PreBlk {  // one 3x3 block of F (5x5 blocks) or V (5x6 blocks): ca dt^pa A + cb dt^pb B; a < 0: zero block
  int a, pa, b, pb;
  double ca, cb;
};
#define PZ {-1, 0, -1, 0, 0.0, 0.0}
#define P1(A, C, P) {A, P, -1, 0, C, 0.0}
__device__ const PreBlk g_pre_tbl[55] = {
    // F, integration_base.h:88-104 (row blocks O_P O_R O_V O_BA O_BG)
    P1(PM_I, 1.0, 0), {PM_RA0, 2, PM_RA1W, 2, -0.25, -0.25}, P1(PM_I, 1.0, 1), P1(PM_RSUM, -0.25, 2), P1(PM_RA1, 0.25, 3),
    PZ, P1(PM_IMW, 1.0, 0), PZ, PZ, P1(PM_I, -1.0, 1),
    PZ, {PM_RA0, 1, PM_RA1W, 1, -0.5, -0.5}, P1(PM_I, 1.0, 0), P1(PM_RSUM, -0.5, 1), P1(PM_RA1, 0.5, 2),
    PZ, PZ, PZ, P1(PM_I, 1.0, 0), PZ,
    PZ, PZ, PZ, PZ, P1(PM_I, 1.0, 0),
    // V, :107-120 (column blocks acc_n gyr_n acc_n gyr_n acc_w gyr_w)
    P1(PM_RDQ, 0.25, 2), P1(PM_RA1, -0.125, 3), P1(PM_RRDQ, 0.25, 2), P1(PM_RA1, -0.125, 3), PZ, PZ,
    PZ, P1(PM_I, 0.5, 1), PZ, P1(PM_I, 0.5, 1), PZ, PZ,
    P1(PM_RDQ, 0.5, 1), P1(PM_RA1, -0.25, 2), P1(PM_RRDQ, 0.5, 1), P1(PM_RA1, -0.25, 2), PZ, PZ,
    PZ, PZ, PZ, PZ, P1(PM_I, 1.0, 1), PZ,
    PZ, PZ, PZ, PZ, PZ, P1(PM_I, 1.0, 1)};
#undef PZ
#undef P1

struct PreEl {  // one matrix entry, table resolved: ca dt^pa M[oa] + cb dt^pb M[ob]   (ca = cb = 0: structural zero / padding)
  int oa, ob, pa, pb;
  double ca, cb;
};
DEV PreEl pre_el(int blk0, int nbc, int i, int j, int ni, int nj) {
  PreEl d{0, 0, 0, 0, 0.0, 0.0};
  if (i < ni && j < nj) {
    const PreBlk b = g_pre_tbl[blk0 + (i / 3) * nbc + j / 3];
    const int e = (i % 3) * 3 + j % 3;
    if (b.a >= 0) d.oa = b.a * 9 + e, d.ca = b.ca, d.pa = b.pa;
    if (b.b >= 0) d.ob = b.b * 9 + e, d.cb = b.cb, d.pb = b.pb;
  }
  return d;
}
DEV double pre_val(const PreEl &d, const double *M, const double *dtp /* 1, dt, dt^2, dt^3 */) {
  return (d.ca * dtp[d.pa]) * M[d.oa] + (d.cb * dtp[d.pb]) * M[d.ob];
}
DEV m33 mul_skew(const m33 &R, d3 v) {  // R [v]x
  m33 o;
#pragma unroll
  for (int i = 0; i < 3; i++) {
    const double r0 = R.a[3 * i], r1 = R.a[3 * i + 1], r2 = R.a[3 * i + 2];
    o.a[3 * i] = r1 * v.z - r2 * v.y, o.a[3 * i + 1] = r2 * v.x - r0 * v.z, o.a[3 * i + 2] = r0 * v.y - r1 * v.x;
  }
  return o;
}
typedef double pre_d4 __attribute__((ext_vector_type(4)));

// lane i <- lane i - D inside its row of 16 lanes (DPP row_shr); lanes without a source keep `none`
template <int D>
DEV double row_shr(double v, double none) {
  const int lo = __builtin_amdgcn_update_dpp(__double2loint(none), __double2loint(v), 0x110 | D, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(__double2hiint(none), __double2hiint(v), 0x110 | D, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
template <int D>
DEV q4 q_row_shr(q4 x, q4 none) {
  return q4{row_shr<D>(x.w, none.w), row_shr<D>(x.x, none.x), row_shr<D>(x.y, none.y), row_shr<D>(x.z, none.z)};
}
DEV double lane_bcast(double v, int src) {  // src wave-uniform
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), src), __builtin_amdgcn_readlane(__double2loint(v), src));
}
// The quaternion chain of a tile as a scan, lane s = sample s (whole wave active; lanes >= nt pass the identity):
// delta_q_k = normalized(delta_q_{k-1} (x) d_k) (:65, :134) is p_k / |p_k| with p_k = d_1 (x) ... (x) d_k, the norm being
// multiplicative — so the products are formed by a three-step prefix scan and the norms taken afterwards, per lane.
// carry: normalized product of everything before the tile (wave-uniform); out: delta_q before step s and result_delta_q.
DEV void pre_quat_scan(q4 &carry, q4 d, int nt, q4 &dq, q4 &rdq) {
  const q4 one = q4{1.0, 0.0, 0.0, 0.0};
  q4 x = d;
  x = qmul(q_row_shr<1>(x, one), x);
  x = qmul(q_row_shr<2>(x, one), x);
  x = qmul(q_row_shr<4>(x, one), x);
  const q4 p = qmul(carry, x), pm = q_row_shr<1>(p, carry);
  const double rn = fast_rsqrt(pm.w * pm.w + pm.x * pm.x + pm.y * pm.y + pm.z * pm.z);
  dq = q4{pm.w * rn, pm.x * rn, pm.y * rn, pm.z * rn};
  rdq = q4{p.w * rn, p.x * rn, p.y * rn, p.z * rn};  // unnormalized, as midPointIntegration uses it (:65-66, :86)
  const q4 l = q4{lane_bcast(p.w, nt - 1), lane_bcast(p.x, nt - 1), lane_bcast(p.y, nt - 1), lane_bcast(p.z, nt - 1)};
  const double rl = fast_rsqrt(l.w * l.w + l.x * l.x + l.y * l.y + l.z * l.z);
  carry = q4{l.w * rl, l.x * rl, l.y * rl, l.z * rl};
}

// grid (intervals) x PRE_THREADS
__global__ __launch_bounds__(PRE_THREADS) void k_preintegrate(const ImuJob *jobs, const double *dts, const double *accs,
                                                             const double *gyrs, const double *noise4, LfvioPreintegration *out) {
  __shared__ double Mt[2][PRE_TILE][PM_COUNT * 9];        // the 3x3 ingredients of the samples of tiles t (being made) and t - 1
  __shared__ double Qd[PRE_TILE][256];                    // Q_s in accumulator layout [r][lane]
  __shared__ double u0[2][PRE_TILE][3], u1[2][PRE_TILE][3];  // dq (acc_0 - ba), result_dq (acc_1 - ba)
  __shared__ double sdt[16][4], sacc[16][3], sgyr[16][3];  // rings: sample s of tile t at (8 t + 1 + s) & 15; dt powers 0..3 at (8 t + s) & 15
  __shared__ double fin[10];                              // delta_p, delta_q, delta_v at the end
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, g = lane >> 4, c = lane & 15;
  const ImuJob *jb = &jobs[blockIdx.x];
  const int n = jb->n, T = (n + PRE_TILE - 1) / PRE_TILE;
  // the entries this lane feeds to the matrix pipe: F[c][g + 4 kb] (kb < 4), V[c][g + 4 kb] (kb < 5) and the noise diagonal
  PreEl fd[4], vd[5];
  double ndl[5];
  {
    const double an = noise4[0], gn = noise4[1], aw = noise4[2], gw = noise4[3];
#pragma unroll
    for (int kb = 0; kb < 5; kb++) {
      const int k = g + 4 * kb, b = k / 3;
      if (kb < 4) fd[kb] = pre_el(0, 5, c, k, 15, 15);
      vd[kb] = pre_el(25, 6, c, k, 15, 18);
      ndl[kb] = k >= 18 ? 0.0 : b == 0 || b == 2 ? an * an : b == 1 || b == 3 ? gn * gn : b == 4 ? aw * aw : gw * gw;  // :21-27
    }
  }
  if (tid < 2 * PRE_TILE * 9) (&Mt[0][0][0])[(tid / 9) * PM_COUNT * 9 + PM_I * 9 + tid % 9] = (tid % 9) % 4 == 0 ? 1.0 : 0.0;
  if (tid >= 64 && tid < 67) sacc[0][tid - 64] = jb->acc_0[tid - 64], sgyr[0][tid - 64] = jb->gyr_0[tid - 64];
  const d3 ba = ld3(jb->ba), bg = ld3(jb->bg);
  q4 carry = q4{1.0, 0.0, 0.0, 0.0};        // waves 2, 3: delta_q entering the tile
  d3 dp = mk3(0, 0, 0), dv = mk3(0, 0, 0);  // wave 4 lane 0: delta_p, delta_v, sum_dt
  double sum_dt = 0.0;
  pre_d4 acc = {0, 0, 0, 0};                // wave 0: covariance, wave 1: jacobian, rows g + 4 r, column c
  if (wave == 1)
#pragma unroll
    for (int r = 0; r < 4; r++) acc[r] = (g + 4 * r == c && c < 15) ? 1.0 : 0.0;
  // a tile's samples travel global -> registers (one tile ahead) -> rings
  double pf_dt = 0, pf_a = 0, pf_g = 0;
  auto prefetch = [&](int t) {
    const int nt = t < T ? min(PRE_TILE, n - PRE_TILE * t) : 0;
    if (tid < nt) pf_dt = dts[jb->off + 8 * t + tid];
    if (tid >= 64 && tid < 64 + 3 * nt) pf_a = accs[3 * (size_t)(jb->off + 8 * t) + tid - 64], pf_g = gyrs[3 * (size_t)(jb->off + 8 * t) + tid - 64];
  };
  auto deposit = [&](int t) {
    const int nt = t < T ? min(PRE_TILE, n - PRE_TILE * t) : 0;
    if (tid < nt) {
      double *o = sdt[(8 * t + tid) & 15];
      o[0] = 1.0, o[1] = pf_dt, o[2] = pf_dt * pf_dt, o[3] = pf_dt * pf_dt * pf_dt;
    }
    if (tid >= 64 && tid < 64 + 3 * nt) {
      const int k = tid - 64;
      sacc[(8 * t + 1 + k / 3) & 15][k % 3] = pf_a, sgyr[(8 * t + 1 + k / 3) & 15][k % 3] = pf_g;
    }
  };
  prefetch(0), deposit(0), prefetch(1);
  __syncthreads();
  for (int t = 0; t <= T; t++) {
    const int nt = t < T ? min(PRE_TILE, n - PRE_TILE * t) : 0, np = t > 0 ? min(PRE_TILE, n - PRE_TILE * (t - 1)) : 0;
    // ---- phase 1
    if (wave < 2) {  // recursions of tile t - 1; F of the next sample is gathered while the matrix pipe works on this one
      const double(*M)[PM_COUNT * 9] = Mt[(t - 1) & 1];
      double fa[4], fb[4];
#pragma unroll
      for (int kb = 0; kb < 4; kb++) fa[kb] = np > 0 ? pre_val(fd[kb], M[0], sdt[(8 * (t - 1)) & 15]) : 0.0;
      for (int s = 0; s < np; s++) {
        const int s1 = min(s + 1, np - 1);
#pragma unroll
        for (int kb = 0; kb < 4; kb++) fb[kb] = pre_val(fd[kb], M[s1], sdt[(8 * (t - 1) + s1) & 15]);
        if (wave == 0) {
          // two accumulators per product: a dependent v_mfma_f64 waits out the whole pipe, an independent one does not
          pre_d4 Ga = {0, 0, 0, 0}, Gb = {0, 0, 0, 0}, Pa, Pb = {0, 0, 0, 0};
#pragma unroll
          for (int r = 0; r < 4; r++) Pa[r] = Qd[s][64 * r + lane];
          Ga = __builtin_amdgcn_mfma_f64_16x16x4f64(acc[0], fa[0], Ga, 0, 0, 0);
          Gb = __builtin_amdgcn_mfma_f64_16x16x4f64(acc[2], fa[2], Gb, 0, 0, 0);
          Ga = __builtin_amdgcn_mfma_f64_16x16x4f64(acc[1], fa[1], Ga, 0, 0, 0);
          Gb = __builtin_amdgcn_mfma_f64_16x16x4f64(acc[3], fa[3], Gb, 0, 0, 0);
          const pre_d4 G = Ga + Gb;
          Pa = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[0], G[0], Pa, 0, 0, 0);
          Pb = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[2], G[2], Pb, 0, 0, 0);
          Pa = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[1], G[1], Pa, 0, 0, 0);
          Pb = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[3], G[3], Pb, 0, 0, 0);
          acc = Pa + Pb;
        } else {
          pre_d4 Ja = {0, 0, 0, 0}, Jb = {0, 0, 0, 0};
          Ja = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[0], acc[0], Ja, 0, 0, 0);
          Jb = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[2], acc[2], Jb, 0, 0, 0);
          Ja = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[1], acc[1], Ja, 0, 0, 0);
          Jb = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[3], acc[3], Jb, 0, 0, 0);
          acc = Ja + Jb;
        }
#pragma unroll
        for (int kb = 0; kb < 4; kb++) fa[kb] = fb[kb];
      }
    } else if (wave < 4 && nt > 0) {  // quaternion chain and 3x3 ingredients of tile t, lane s = sample s
      const int s = lane & 7;
      const bool live = lane < nt;
      const double dt = sdt[(8 * t + s) & 15][1];
      const d3 w_x = 0.5 * (ld3(sgyr[(8 * t + s) & 15]) + ld3(sgyr[(8 * t + s + 1) & 15])) - bg;  // :64, :76
      q4 dq, rdq;
      pre_quat_scan(carry, live ? q4{1.0, w_x.x * dt / 2, w_x.y * dt / 2, w_x.z * dt / 2} : q4{1.0, 0.0, 0.0, 0.0}, nt, dq, rdq);
      double *M = Mt[t & 1][s];
      if (wave == 2 && live) {
        const d3 a_0_x = ld3(sacc[(8 * t + s) & 15]) - ba;  // :77
        const m33 Rdq = q2R(dq);
        stm(M + PM_RDQ * 9, Rdq), stm(M + PM_RA0 * 9, mul_skew(Rdq, a_0_x));
        const d3 u = qrot(dq, a_0_x);  // :63
        u0[t & 1][s][0] = u.x, u0[t & 1][s][1] = u.y, u0[t & 1][s][2] = u.z;
      } else if (wave == 3 && live) {
        const d3 a_1_x = ld3(sacc[(8 * t + s + 1) & 15]) - ba;  // :78
        const m33 Rdq = q2R(dq), Rrdq = q2R(rdq), R_w_x = skewm(w_x), RA1 = mul_skew(Rrdq, a_1_x);
        m33 ImW, Rsum;
#pragma unroll
        for (int e = 0; e < 9; e++) ImW.a[e] = (e % 4 == 0 ? 1.0 : 0.0) - R_w_x.a[e] * dt, Rsum.a[e] = Rdq.a[e] + Rrdq.a[e];
        stm(M + PM_RRDQ * 9, Rrdq), stm(M + PM_RA1 * 9, RA1), stm(M + PM_RA1W * 9, mm(RA1, ImW));
        stm(M + PM_IMW * 9, ImW), stm(M + PM_RSUM * 9, Rsum);
        const d3 u = qrot(rdq, a_1_x);  // :66
        u1[t & 1][s][0] = u.x, u1[t & 1][s][1] = u.y, u1[t & 1][s][2] = u.z;
      }
    } else if (tid == 256) {  // delta_p, delta_v, sum_dt over tile t - 1
      for (int s = 0; s < np; s++) {
        const double dt = sdt[(8 * (t - 1) + s) & 15][1];
        const d3 ua = 0.5 * (ld3(u0[(t - 1) & 1][s]) + ld3(u1[(t - 1) & 1][s]));  // :67
        dp = dp + dv * dt + 0.5 * ua * dt * dt;                                    // :68
        dv = dv + ua * dt;                                                         // :69
        sum_dt += dt;
      }
    }
    __syncthreads();
    // ---- phase 2: Q_s = V_s noise V_s^T, wave s; then the next tile's samples into the rings
    if (wave < nt) {
      const int s = wave;
      pre_d4 Q = {0, 0, 0, 0};
#pragma unroll
      for (int kb = 0; kb < 5; kb++) {
        const double v = pre_val(vd[kb], Mt[t & 1][s], sdt[(8 * t + s) & 15]);
        Q = __builtin_amdgcn_mfma_f64_16x16x4f64(v * ndl[kb], v, Q, 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < 4; r++) Qd[s][64 * r + lane] = Q[r];
    }
    deposit(t + 1), prefetch(t + 2);
    __syncthreads();
  }
  if (tid == 128) fin[3] = carry.w, fin[4] = carry.x, fin[5] = carry.y, fin[6] = carry.z;
  if (tid == 256) fin[0] = dp.x, fin[1] = dp.y, fin[2] = dp.z, fin[7] = dv.x, fin[8] = dv.y, fin[9] = dv.z, out[blockIdx.x].sum_dt = sum_dt;
  __syncthreads();
  LfvioPreintegration *o = &out[blockIdx.x];
  if (wave < 2 && c < 15) {
    double *dst = wave == 0 ? o->covariance : o->jacobian;
#pragma unroll
    for (int r = 0; r < 4; r++)
      if (g + 4 * r < 15) dst[(g + 4 * r) * 15 + c] = acc[r];
  }
  if (tid >= 320 && tid < 323) {
    const int k = tid - 320;
    o->delta_p[k] = fin[k], o->delta_v[k] = fin[7 + k], o->linearized_ba[k] = jb->ba[k], o->linearized_bg[k] = jb->bg[k];
    o->delta_q[k] = fin[4 + k];
    if (k == 0) o->delta_q[3] = fin[3];
  }
}
