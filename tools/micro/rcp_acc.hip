// Relative error of v_rcp_f64 / v_rsq_f64 raw and after one and two Newton steps (against long double on the host),
// and the latency of a dependent chain of v_fma_f64 / v_mov_dpp + v_add_f64 with one wave on the SIMD.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
__global__ void k(const double *x, double *o, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double v = x[i];
  double y = __builtin_amdgcn_rcp(v);
  o[i] = y;
  y = fma(y, fma(-v, y, 1.0), y);
  o[n + i] = y;
  y = fma(y, fma(-v, y, 1.0), y);
  o[2 * n + i] = y;
  double r = __builtin_amdgcn_rsq(v);
  o[3 * n + i] = r;
  r = r * fma(-0.5 * v * r, r, 1.5);
  o[4 * n + i] = r;
  r = r * fma(-0.5 * v * r, r, 1.5);
  o[5 * n + i] = r;
}
__device__ __forceinline__ long long stamp(double &x) {  // the counter is read once everything x depends on has been issued
  long long t;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t), "+v"(x)::"memory");
  return t;
}
__global__ void lat(long long *out, double *sink, double a) {
  double x = a + threadIdx.x, y = 1.0 + 1e-9 * threadIdx.x;
  long long t0 = stamp(x);
#pragma unroll
  for (int i = 0; i < 256; i++) x = fma(x, y, a);
  long long t1 = stamp(x);
#pragma unroll
  for (int i = 0; i < 256; i++) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), 0xB1, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), 0xB1, 0xf, 0xf, true);
    x += __hiloint2double(hi, lo);
  }
  long long t2 = stamp(x);
#pragma unroll
  for (int i = 0; i < 64; i++) x = __builtin_amdgcn_rcp(x);
  long long t3 = stamp(x);
  double z0 = x, z1 = x + 1, z2 = x + 2, z3 = x + 3;
#pragma unroll
  for (int i = 0; i < 64; i++) z0 = fma(z0, y, a), z1 = fma(z1, y, a), z2 = fma(z2, y, a), z3 = fma(z3, y, a);
  z0 += z1 + z2 + z3;
  long long t4 = stamp(z0);
  if (threadIdx.x == 0) out[0] = t1 - t0, out[1] = t2 - t1, out[2] = t3 - t2, out[3] = t4 - t3;
  sink[threadIdx.x] = x + z0 + z1 + z2 + z3;
}
int main() {
  const int n = 1 << 20;
  std::vector<double> h(n), o(6 * n);
  unsigned long long s = 88172645463325252ull;
  for (int i = 0; i < n; i++) {
    s ^= s << 13, s ^= s >> 7, s ^= s << 17;
    const double u = (double)(s >> 11) / 9007199254740992.0;
    h[i] = ldexp(1.0 + u, (int)(s % 41) - 20);
  }
  double *dx, *dout;
  hipMalloc(&dx, 8 * n), hipMalloc(&dout, 48 * n);
  hipMemcpy(dx, h.data(), 8 * n, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, dout, n);
  hipMemcpy(o.data(), dout, 48 * n, hipMemcpyDeviceToHost);
  const char *nm[6] = {"rcp raw", "rcp 1 NR", "rcp 2 NR", "rsq raw", "rsq 1 NR", "rsq 2 NR"};
  for (int m = 0; m < 6; m++) {
    long double worst = 0;
    for (int i = 0; i < n; i++) {
      const long double ex = m < 3 ? 1.0L / (long double)h[i] : 1.0L / sqrtl((long double)h[i]);
      const long double e = fabsl(((long double)o[(size_t)m * n + i] - ex) / ex);
      if (e > worst) worst = e;
    }
    printf("%s: max relative error %.3Le (%.2Lf ulp)\n", nm[m], worst, worst / 1.1102230246251565e-16L);
  }
  long long *dl, hl[4];
  hipMalloc(&dl, 64);
  for (int rep = 0; rep < 2; rep++) {
    hipLaunchKernelGGL(lat, dim3(1), dim3(64), 0, 0, dl, dout, 0.5);
    hipMemcpy(hl, dl, 32, hipMemcpyDeviceToHost);
    printf("one wave: dependent v_fma_f64 %.1f cycles each | dpp pair + v_add_f64 %.1f per step | dependent v_rcp_f64 %.1f | 4 independent fma chains %.1f per fma\n", hl[0] / 256.0,
           hl[1] / 256.0, hl[2] / 64.0, hl[3] / 256.0);
  }
  return 0;
}
