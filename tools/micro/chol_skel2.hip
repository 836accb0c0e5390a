// what inside the column-factor block of the pivot loop is expensive? (owner wave only does the work; barrier each pivot)
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
    double d = 2.0 + 1e-9 * it;
    if (MODE & 1) {
      const int src = ((kk & 3) << 4) | kk;
      const double dd = m[0];
      d = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(dd), src), __builtin_amdgcn_readlane(__double2loint(dd), src));
      d = fabs(d) + 1.0;
    }
    if (tcol == kk) {
      double *cbw = colbuf + ((it + 1) & 1) * 176;
      double y = 0.7;
      if (MODE & 2) {
        y = __builtin_amdgcn_rsq(d);
        const double hx = 0.5 * d;
        y = fma(y, fma(-hx * y, y, 0.5), y);
        y = fma(y, fma(-hx * y, y, 0.5), y);
      }
      if (MODE & 4) {  // branchy scale as in the kernel
#pragma unroll
        for (int a = 0; a < 11; a++) {
          const int i = trow + 16 * a;
          double v = 0.0;
          if (i > k && i <= 172) {
            m[a] *= y;
            v = m[a];
          } else if (i == k) {
            m[a] = d * y;
            invd[k] = y;
          }
          if (MODE & 16) cbw[i] = v;
        }
      }
      if (MODE & 8) {  // select-based scale
#pragma unroll
        for (int a = 0; a < 11; a++) {
          const int i = trow + 16 * a;
          const double sc = m[a] * y;
          const bool in = i > k && i <= 172;
          m[a] = in ? sc : (i == k ? d * y : m[a]);
          if (MODE & 16) cbw[i] = in ? sc : 0.0;
        }
        if (trow == (k & 15)) invd[k] = y;
      }
      m[0] += y * 1e-9;
    }
    __syncthreads();
  }
  long long t1 = __builtin_readcyclecounter();
  if (tid == 0) out[0] = t1 - t0;
  double s = 0;
  for (int a = 0; a < 11; a++) s += m[a];
  sink[tid] = s + invd[tid & 127];
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
  run<0>(out, sink, "barrier only (+owner branch)");
  run<1>(out, sink, "readlane");
  run<1 | 2>(out, sink, "readlane + rsqrt");
  run<1 | 2 | 4>(out, sink, "readlane + rsqrt + branchy scale");
  run<1 | 2 | 4 | 16>(out, sink, "readlane + rsqrt + branchy scale + store");
  run<1 | 2 | 8>(out, sink, "readlane + rsqrt + select scale");
  run<1 | 2 | 8 | 16>(out, sink, "readlane + rsqrt + select scale + store");
  run<2 | 8 | 16>(out, sink, "rsqrt + select scale + store (no readlane)");
  return 0;
}
