// skeleton of the k_solve pivot loop (no trailing update), to see what one pivot costs and why
#include <hip/hip_runtime.h>
#include <cstdio>
#define N 1000
template <int MODE>
__global__ __launch_bounds__(256) void k(long long *out, double *sink) {
  __shared__ double colbuf[2 * 176];
  __shared__ double invd[176];
  const int tid = threadIdx.x, trow = tid & 15, tcol = tid >> 4;
  double m[11];
  for (int a = 0; a < 11; a++) m[a] = 1.0 + 1e-3 * (tid + a);
  if (tid < 352) colbuf[tid] = 1e-3;
  __syncthreads();
  long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < N; it++) {
    const int kk = it & 15, k = it % 160;
    const double *cb = colbuf + (it & 1) * 176;
    double li[11], lj[11];
    if (MODE & 1) {
#pragma unroll
      for (int a = 0; a < 11; a++) li[a] = cb[trow + 16 * a], lj[a] = cb[tcol + 16 * a];
    } else {
#pragma unroll
      for (int a = 0; a < 11; a++) li[a] = 1e-3, lj[a] = 1e-3;
    }
#pragma unroll
    for (int a = 0; a < 11; a++) m[a] = fma(-li[a], lj[0], m[a]);
    if (MODE & 2) {
      const int src = ((kk & 3) << 4) | kk;
      const double dd = m[0];
      const double d = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(dd), src), __builtin_amdgcn_readlane(__double2loint(dd), src));
      if (tcol == kk) {
        double *cbw = colbuf + ((it + 1) & 1) * 176;
        double y = __builtin_amdgcn_rsq(d);
        const double hx = 0.5 * d;
        y = fma(y, fma(-hx * y, y, 0.5), y);
        y = fma(y, fma(-hx * y, y, 0.5), y);
#pragma unroll
        for (int a = 0; a < 11; a++) {
          const int i = trow + 16 * a;
          double v = 0.0;
          if (i > k && i <= 172) {
            m[a] = m[a] * y + 1.0;
            v = m[a] * 1e-3;
          } else if (i == k) {
            m[a] = d * y;
            invd[k] = y;
          }
          if (MODE & 4) cbw[i] = v;
        }
      }
    }
    if (MODE & 8) __syncthreads();
    if (MODE & 16) {  // trailing update, 55 FMAs
#pragma unroll
      for (int a = 1; a < 11; a++)
#pragma unroll
        for (int b = 1; b <= a; b++) m[a] = fma(-li[a], lj[b], m[a]);
    }
  }
  long long t1 = __builtin_readcyclecounter();
  if (tid == 0) out[0] = t1 - t0;
  double s = 0;
  for (int a = 0; a < 11; a++) s += m[a];
  sink[tid] = s;
}
template <int MODE> void run(long long *out, double *sink, const char *what) {
  hipLaunchKernelGGL(k<MODE>, dim3(1), dim3(256), 0, 0, out, sink);
  (void)hipDeviceSynchronize();
  long long h;
  (void)hipMemcpy(&h, out, 8, hipMemcpyDeviceToHost);
  printf("%-58s %7.1f cycles/pivot\n", what, h / (double)N);
}
int main() {
  long long *out; double *sink;
  (void)hipMalloc(&out, 64); (void)hipMalloc(&sink, 256 * 8);
  run<0>(out, sink, "11 FMA only");
  run<1>(out, sink, "+ 22 LDS column loads");
  run<1 | 8>(out, sink, "loads + barrier");
  run<1 | 2>(out, sink, "loads + readlane/rsqrt/scale (no store)");
  run<1 | 2 | 4>(out, sink, "loads + factor + column store");
  run<1 | 2 | 4 | 8>(out, sink, "loads + factor + store + barrier  (= skeleton)");
  run<1 | 2 | 4 | 8 | 16>(out, sink, "skeleton + 55 FMA trailing update");
  run<2 | 4 | 8>(out, sink, "factor + store + barrier, no loads");
  return 0;
}
