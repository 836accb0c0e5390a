// the in-wave factorization of one 16 x 16 diagonal tile (row per lane), variants of the broadcast / update schedule
#include <hip/hip_runtime.h>
#include <cstdio>
#define DEV __device__ __forceinline__
DEV double fast_rcp(double x) { double y = __builtin_amdgcn_rcp(x); y = fma(y, fma(-x, y, 1.0), y); y = fma(y, fma(-x, y, 1.0), y); return y; }
DEV double fast_rcp1(double x) { double y = __builtin_amdgcn_rcp(x); y = fma(y, fma(-x, y, 1.0), y); return y; }
template <int J> DEV double row_bcast(double v) { return __builtin_amdgcn_update_dpp(v, v, 0x150 + J, 0xf, 0xf, false); }
DEV double readlane_f64(double v, int src) { return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), src), __builtin_amdgcn_readlane(__double2loint(v), src)); }
#define FOR_J(K, OP) OP(K, 1) OP(K, 2) OP(K, 3) OP(K, 4) OP(K, 5) OP(K, 6) OP(K, 7) OP(K, 8) OP(K, 9) OP(K, 10) OP(K, 11) OP(K, 12) OP(K, 13) OP(K, 14) OP(K, 15)
#define ALLK(P) P(0) P(1) P(2) P(3) P(4) P(5) P(6) P(7) P(8) P(9) P(10) P(11) P(12) P(13) P(14) P(15)
template <int MODE>
__global__ void k(long long *out, double *sink, const double *in) {
  const int lane = threadIdx.x & 63, row = lane & 15;
  double a[16];
  for (int j = 0; j < 16; j++) a[j] = in[row * 16 + j];
  long long t0 = __builtin_readcyclecounter();
  for (int rep = 0; rep < 8; rep++) {
    if (MODE == 0) {  // products first, reciprocal beside them
#define BC(K, J) if (J > K) pr[J] = a[K] * row_bcast<J>(a[K]);
#define UPD(K, J) if (J > K) a[J] = fma(pr[J], ndinv, a[J]);
#define PIV(K) { const double d = row_bcast<K>(a[K]); double pr[16]; FOR_J(K, BC) const double ndinv = row > K ? -fast_rcp(d) : 0.0; FOR_J(K, UPD) }
      ALLK(PIV)
#undef PIV
#undef UPD
#undef BC
    } else if (MODE == 1) {  // f = a/d, then fma with the broadcast
#define UPD(K, J) if (J > K) a[J] = fma(-f, row_bcast<J>(a[K]), a[J]);
#define PIV(K) { const double d = row_bcast<K>(a[K]); const double f = row > K ? a[K] * fast_rcp(d) : 0.0; FOR_J(K, UPD) }
      ALLK(PIV)
#undef PIV
#undef UPD
    } else if (MODE == 2) {  // as 1 with one Newton step
#define UPD(K, J) if (J > K) a[J] = fma(-f, row_bcast<J>(a[K]), a[J]);
#define PIV(K) { const double d = row_bcast<K>(a[K]); const double f = row > K ? a[K] * fast_rcp1(d) : 0.0; FOR_J(K, UPD) }
      ALLK(PIV)
#undef PIV
#undef UPD
    } else {  // readlane version
#define UPD(K, J) if (J > K) a[J] = fma(-f, readlane_f64(a[K], J), a[J]);
#define PIV(K) { const double d = readlane_f64(a[K], K); const double f = row > K ? a[K] * fast_rcp(d) : 0.0; FOR_J(K, UPD) }
      ALLK(PIV)
#undef PIV
#undef UPD
    }
    for (int j = 0; j < 16; j++) a[j] = a[j] * 1e-3 + in[row * 16 + j];  // keep it well-conditioned for the next repetition
  }
  long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) out[MODE] = (t1 - t0) / 8;
  double s = 0; for (int j = 0; j < 16; j++) s += a[j];
  sink[threadIdx.x] = s;
}
int main() {
  long long *out; double *sink, *in; double h[256];
  for (int i = 0; i < 16; i++) for (int j = 0; j < 16; j++) h[i * 16 + j] = (i == j ? 20.0 : 0.0) + 1.0 / (1 + i + j);
  hipMalloc(&out, 64); hipMalloc(&sink, 8 * 256); hipMalloc(&in, 8 * 256); hipMemcpy(in, h, 2048, hipMemcpyHostToDevice);
  for (int rep = 0; rep < 2; rep++) {
    hipLaunchKernelGGL(k<0>, dim3(1), dim3(64), 0, 0, out, sink, in); hipLaunchKernelGGL(k<1>, dim3(1), dim3(64), 0, 0, out, sink, in);
    hipLaunchKernelGGL(k<2>, dim3(1), dim3(64), 0, 0, out, sink, in); hipLaunchKernelGGL(k<3>, dim3(1), dim3(64), 0, 0, out, sink, in);
    long long r[4]; hipMemcpy(r, out, 32, hipMemcpyDeviceToHost);
    printf("16x16 in-wave factor, cycles per tile: products-first %lld, f*bcast %lld, f*bcast 1 Newton %lld, readlane %lld\n", r[0], r[1], r[2], r[3]);
  }
  return 0;
}
