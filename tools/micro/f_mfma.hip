// In-wave Cholesky of one 16 x 16 tile, two ways (cycles per tile and the error against a host factorization):
//   A  row per lane, pivot row by v_readlane (what k_solve used): every lane carries its whole row, 120 broadcast+fma pairs
//   B  lane (row c, quarter g) owns columns g, g+4, g+8, g+12 of its row of the FULL symmetric tile; the four pivots of a
//      panel are eliminated on the vector pipe inside the panel only, the rest of the tile takes the rank-4 update
//      C -= P diag(1/d) P^T in ONE v_mfma_f64_16x16x4_f64 whose A / B / C operands are exactly the registers the lanes
//      already hold (the accumulator layout D[g + 4 r][c] is the transpose of the ownership, and the tile is symmetric)
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#define DEV __device__ __forceinline__
typedef double d4 __attribute__((ext_vector_type(4)));
DEV double fast_rcp(double x) { double y = __builtin_amdgcn_rcp(x); y = fma(y, fma(-x, y, 1.0), y); y = fma(y, fma(-x, y, 1.0), y); return y; }
DEV double fast_rcp1(double x) { double y = __builtin_amdgcn_rcp(x); y = fma(y, fma(-x, y, 1.0), y); return y; }
DEV double fast_rsqrt(double x) { double y = __builtin_amdgcn_rsq(x); y = y * fma(-0.5 * x * y, y, 1.5); y = y * fma(-0.5 * x * y, y, 1.5); return y; }
DEV double readlane_f64(double v, int src) { return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), src), __builtin_amdgcn_readlane(__double2loint(v), src)); }
template <int J> DEV double row_bcast(double v) { return __builtin_amdgcn_update_dpp(v, v, 0x150 + J, 0xf, 0xf, false); }
// every lane <- the lane with the same row in quarter T (lanes 16 T .. 16 T + 15)
template <int T> DEV int quarter_bcast32(int v) {
  auto r = __builtin_amdgcn_permlane16_swap(v, v, false, false);
  const int w = (T & 1) ? r[1] : r[0];
  auto q = __builtin_amdgcn_permlane32_swap(w, w, false, false);
  return (T & 2) ? q[1] : q[0];
}
template <int T> DEV double quarter_bcast(double x) { return __hiloint2double(quarter_bcast32<T>(__double2hiint(x)), quarter_bcast32<T>(__double2loint(x))); }

constexpr int TLD = 17;
DEV int tsw(int r, int k) { return r * TLD + k; }

template <int NB> DEV bool factor_rows(double *Td, double *invd, int lane) {  // variant A (k_solve's)
  const int row = lane & 15;
  bool bad = false;
  double a[16];
#pragma unroll
  for (int j = 0; j < 16; j++) a[j] = j <= row ? Td[tsw(row, j)] : 0.0;
  double mydiag = 1.0;
#pragma unroll
  for (int k = 0; k < 16; k++) {
    if (k < NB) {
      const double d = readlane_f64(a[k], k);
      if (!(d > 0.0)) bad = true;
      if (row == k) mydiag = d;
      const double f = row > k ? a[k] * fast_rcp(d) : 0.0;
#pragma unroll
      for (int j = k + 1; j < 16; j++) a[j] = fma(-f, readlane_f64(a[k], j), a[j]);
    }
  }
  const double myrs = fast_rsqrt(mydiag);
  if (lane < 16) {
#pragma unroll
    for (int j = 0; j < 16; j++) {
      const double rsj = readlane_f64(myrs, j);
      double v = 0.0;
      if (j < NB) v = j < row ? a[j] * rsj : (j == row ? mydiag * myrs : 0.0);
      Td[tsw(row, j)] = v;
    }
    if (lane < NB) invd[lane] = myrs;
  }
  return bad;
}

#ifndef ABL
#define ABL 0
#endif
#ifndef RCP
#define RCP fast_rcp
#endif
template <int NB> DEV bool factor_mfma(double *Td, double *invd, int lane) {  // variant B
  const int c = lane & 15, g = lane >> 4;
  bool bad = false;
  d4 a;  // a[i] = A[c][g + 4 i], both triangles
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const int col = g + 4 * i;
    a[i] = col <= c ? Td[tsw(c, col)] : Td[tsw(col, c)];
  }
  double dsave[4] = {1.0, 1.0, 1.0, 1.0};
#pragma unroll
  for (int p = 0; p < 4; p++) {
    double bop = 0.0;  // this lane's B operand: its panel column times 1/d of that column's pivot
#pragma unroll
    for (int t = 0; t < 4; t++) {
      const int k = 4 * p + t;
      if (k >= NB) continue;
      // everything that does not need 1/d is requested first, so that it runs beside the reciprocal: the pivot column
      // for every quarter (v_permlane swaps) and row k of this lane's own panel column (DPP)
      double colk = 0.0, u = 0.0;
      if (t < 3) {
#if ABL & 2
        colk = a[p];
#else
        if (t == 0) colk = quarter_bcast<0>(a[p]);
        else if (t == 1) colk = quarter_bcast<1>(a[p]);
        else colk = quarter_bcast<2>(a[p]);
#endif
#if ABL & 4
        u = a[p];
        if (0)
#endif
        switch (k) {
#define RB(K) case K: u = row_bcast<K>(a[p]); break;
          RB(0) RB(1) RB(2) RB(3) RB(4) RB(5) RB(6) RB(7) RB(8) RB(9) RB(10) RB(11) RB(12) RB(13) RB(14) default: u = row_bcast<15>(a[p]);
#undef RB
        }
      }
#if ABL & 8
      const double d = a[p] + 3.0;
#else
      const double d = readlane_f64(a[p], 16 * t + k);  // pivot: row k in quarter t
#endif
      if (!(d > 0.0)) bad = true;
#if ABL & 1
      const double rc = __builtin_amdgcn_rcp(d);
#else
      const double rc = RCP(d);
#endif
      if (g == t) dsave[p] = d, bop = a[p] * rc;  // B operand of the panel's rank-4 update: a[c][k] / d
      if (t < 3) {
        // the rows below the pivot, the panel columns right of it (owned by the quarters above t)
        const double upd = fma(c > k ? -(colk * rc) : 0.0, u, a[p]);
        if (g > t) a[p] = upd;
      }
    }
    if (p < 3 && 4 * p < NB) {
      // rank-4 update of the columns behind the panel: operands and accumulator are the registers as they stand
      d4 cv = __builtin_amdgcn_mfma_f64_16x16x4f64(-a[p], bop, a, 0, 0, 0);
#pragma unroll
      for (int i = 0; i < 4; i++)
        if (i > p) a[i] = cv[i];
    }
  }
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const int col = g + 4 * i;
    const double rs = fast_rsqrt(dsave[i]);
    double v = 0.0;
    if (col < NB) v = c > col ? a[i] * rs : (c == col ? dsave[i] * rs : 0.0);
    Td[tsw(c, col)] = v;
    if (c == col && col < NB) invd[col] = rs;
  }
  return bad;
}

template <int MODE, int NB>
__global__ void k(long long *out, double *res, const double *in) {
  __shared__ double Td[16 * TLD], invd[16];
  const int lane = threadIdx.x;
  long long total = 0;
  bool bad = false;
  for (int rep = 0; rep < 8; rep++) {
    for (int e = lane; e < 256; e += 64) Td[tsw(e >> 4, e & 15)] = in[e];
    __syncthreads();
    const long long t0 = __builtin_readcyclecounter();
    bad |= MODE == 0 ? factor_rows<NB>(Td, invd, lane) : factor_mfma<NB>(Td, invd, lane);
    const long long t1 = __builtin_readcyclecounter();
    total += t1 - t0;
    __syncthreads();
  }
  if (lane == 0) out[MODE] = total / 8 + (bad ? 1000000 : 0);
  for (int e = lane; e < 256; e += 64) res[MODE * 512 + e] = Td[tsw(e >> 4, e & 15)];
  if (lane < 16) res[MODE * 512 + 256 + lane] = invd[lane];
}

__global__ void k_perm(int *o) {
  const int v = threadIdx.x;
  o[threadIdx.x] = quarter_bcast32<0>(v), o[64 + threadIdx.x] = quarter_bcast32<1>(v), o[128 + threadIdx.x] = quarter_bcast32<2>(v),
  o[192 + threadIdx.x] = quarter_bcast32<3>(v);
}

template <int NB> static void run(const char *name) {
  long long *out; double *res, *in; double h[256], L[256] = {0};
  // SPD with a spread of scales; row 12.. of the NB = 12 case is the rhs row / padding (never a pivot)
  for (int i = 0; i < 16; i++) for (int j = 0; j < 16; j++) h[i * 16 + j] = (i == j ? 3.0 + i : 0.0) + 1.0 / (1 + i + j) + 0.01 * ((i * 7 + j * 7) % 5);
  for (int j = 0; j < NB; j++) {
    double s = h[j * 16 + j];
    for (int k = 0; k < j; k++) s -= L[j * 16 + k] * L[j * 16 + k];
    L[j * 16 + j] = sqrt(s);
    for (int i = j + 1; i < 16; i++) {
      double t = h[i * 16 + j];
      for (int k = 0; k < j; k++) t -= L[i * 16 + k] * L[j * 16 + k];
      L[i * 16 + j] = t / L[j * 16 + j];
    }
  }
  hipMalloc(&out, 64); hipMalloc(&res, 8 * 1024); hipMalloc(&in, 8 * 256); hipMemcpy(in, h, 2048, hipMemcpyHostToDevice);
  for (int rep = 0; rep < 2; rep++) {
    hipLaunchKernelGGL((k<0, NB>), dim3(1), dim3(64), 0, 0, out, res, in);
    hipLaunchKernelGGL((k<1, NB>), dim3(1), dim3(64), 0, 0, out, res, in);
    long long r[2]; double hr[1024];
    hipMemcpy(r, out, 16, hipMemcpyDeviceToHost); hipMemcpy(hr, res, 8192, hipMemcpyDeviceToHost);
    double err[2] = {0, 0}, erri[2] = {0, 0};
    for (int m = 0; m < 2; m++) {
      for (int i = 0; i < 16; i++) for (int j = 0; j < NB; j++) if (j <= i) err[m] = fmax(err[m], fabs(hr[m * 512 + i * 16 + j] - L[i * 16 + j]));
      for (int j = 0; j < NB; j++) erri[m] = fmax(erri[m], fabs(hr[m * 512 + 256 + j] - 1.0 / L[j * 16 + j]));
    }
    printf("%s: cycles per tile rows+readlane %lld (max |dL| %.1e, |d 1/Lii| %.1e) | quarters+mfma %lld (max |dL| %.1e, |d 1/Lii| %.1e)\n", name, r[0], err[0], erri[0], r[1],
           err[1], erri[1]);
  }
}

int main() {
  int *o; int h[256];
  hipMalloc(&o, 1024);
  hipLaunchKernelGGL(k_perm, dim3(1), dim3(64), 0, 0, o);
  hipMemcpy(h, o, 1024, hipMemcpyDeviceToHost);
  bool ok = true;
  for (int t = 0; t < 4; t++) for (int l = 0; l < 64; l++) ok &= h[64 * t + l] == 16 * t + (l & 15);
  printf("quarter broadcast by v_permlane16_swap + v_permlane32_swap: %s (lane 37 gets %d %d %d %d)\n", ok ? "ok" : "WRONG", h[37], h[64 + 37], h[128 + 37], h[192 + 37]);
  run<16>("16 pivots");
  run<12>("12 pivots (last tile)");
  return 0;
}
