// Ticks per step of the quotient-difference recurrence q <- d - l - e2 / q (v_rcp_f64 + two Newton steps), the chain of
// the twisted factorization in k_marg_solve, with the two pivot guards.
#include <hip/hip_runtime.h>
#include <cstdio>
#define DEV __device__ __forceinline__
DEV double fast_rcp(double x) { double y = __builtin_amdgcn_rcp(x); y = fma(y, fma(-x, y, 1.0), y); y = fma(y, fma(-x, y, 1.0), y); return y; }
DEV double fast_rcp1(double x) { double y = __builtin_amdgcn_rcp(x); y = fma(y, fma(-x, y, 1.0), y); return y; }
DEV long long stamp(double &x) { long long t; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t), "+v"(x)::"memory"); return t; }
template <int MODE> __global__ void k(long long *out, double *sink, const double *T, double l0) {
  __shared__ double dd[96], ee2[96];
  for (int i = threadIdx.x; i < 96; i += 64) dd[i] = T[i], ee2[i] = T[96 + i];
  __syncthreads();
  const double l = l0 + 1e-3 * threadIdx.x, piv = 1e-290;
  double q = dd[0] - l;
  const long long t0 = stamp(q);
  for (int i0 = 1; i0 < 73; i0 += 8) {
    double dv[8], ev[8];
#pragma unroll
    for (int u = 0; u < 8; u++) dv[u] = dd[i0 + u], ev[u] = ee2[i0 + u - 1];
#pragma unroll
    for (int u = 0; u < 8; u++) {
      if (MODE == 0) { if (fabs(q) < piv) q = -piv; q = fma(-ev[u], fast_rcp(q), dv[u] - l); }
      if (MODE == 1) q = fma(-ev[u], fast_rcp(copysign(fmax(fabs(q), piv), q)), dv[u] - l);
      if (MODE == 2) q = fma(-ev[u], fast_rcp(q), dv[u] - l);
      if (MODE == 3) q = fma(-ev[u], fast_rcp1(q), dv[u] - l);
      if (MODE == 4) q = fma(-ev[u], __builtin_amdgcn_rcp(q), dv[u] - l);
    }
  }
  const long long t1 = stamp(q);
  if (threadIdx.x == 0) out[MODE] = t1 - t0;
  sink[MODE * 64 + threadIdx.x] = q;
}
int main() {
  double h[192], *T, *sink; long long *out, r[8];
  for (int i = 0; i < 96; i++) h[i] = 2.0 + 0.01 * i, h[96 + i] = 0.3 + 0.001 * i;
  hipMalloc(&T, sizeof h), hipMalloc(&sink, 8 * 512), hipMalloc(&out, 64);
  hipMemcpy(T, h, sizeof h, hipMemcpyHostToDevice);
  for (int rep = 0; rep < 2; rep++) {
    hipLaunchKernelGGL(k<0>, dim3(1), dim3(64), 0, 0, out, sink, T, 0.7);
    hipLaunchKernelGGL(k<1>, dim3(1), dim3(64), 0, 0, out, sink, T, 0.7);
    hipLaunchKernelGGL(k<2>, dim3(1), dim3(64), 0, 0, out, sink, T, 0.7);
    hipLaunchKernelGGL(k<3>, dim3(1), dim3(64), 0, 0, out, sink, T, 0.7);
    hipLaunchKernelGGL(k<4>, dim3(1), dim3(64), 0, 0, out, sink, T, 0.7);
    hipMemcpy(r, out, 40, hipMemcpyDeviceToHost);
    printf("ticks per row (72 rows): cmp+select guard %.1f | max+copysign guard %.1f | no guard %.1f | one Newton step %.1f | raw v_rcp_f64 %.1f\n", r[0] / 72.0, r[1] / 72.0, r[2] / 72.0, r[3] / 72.0, r[4] / 72.0);
  }
  return 0;
}
