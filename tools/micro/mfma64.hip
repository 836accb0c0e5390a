// v_mfma_f64_16x16x4_f64 on gfx950: cycles per instruction, dependent (same accumulator) and independent (two accumulators)
#include <hip/hip_runtime.h>
#include <cstdio>
#define N 1000
typedef double d4 __attribute__((ext_vector_type(4)));
__global__ void k(long long *out, double *sink) {
  const int tid = threadIdx.x;
  double a = 1.0 + tid * 1e-9, b = 0.5 + tid * 1e-9;
  d4 c0 = {0, 0, 0, 0}, c1 = {0, 0, 0, 0}, c2 = {0, 0, 0, 0}, c3 = {0, 0, 0, 0};
  long long t0, t1;
  t0 = __builtin_readcyclecounter();
  for (int i = 0; i < N; i++) c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
  t1 = __builtin_readcyclecounter();
  if (tid == 0) out[0] = t1 - t0;
  t0 = __builtin_readcyclecounter();
  for (int i = 0; i < N / 2; i++) {
    c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
  }
  t1 = __builtin_readcyclecounter();
  if (tid == 0) out[1] = t1 - t0;
  t0 = __builtin_readcyclecounter();
  for (int i = 0; i < N / 4; i++) {
    c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0);
    c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0);
  }
  t1 = __builtin_readcyclecounter();
  if (tid == 0) out[2] = t1 - t0;
  // result of one feeding the B operand of the next (the recursion's pattern)
  t0 = __builtin_readcyclecounter();
  for (int i = 0; i < N; i++) {
    d4 z = {0, 0, 0, 0};
    c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, c0[0], z, 0, 0, 0);
  }
  t1 = __builtin_readcyclecounter();
  if (tid == 0) out[3] = t1 - t0;
  // dependent FP64 FMA chain for scale
  double x = a;
  t0 = __builtin_readcyclecounter();
  for (int i = 0; i < N; i++) x = fma(x, 0.999999, b);
  t1 = __builtin_readcyclecounter();
  if (tid == 0) out[4] = t1 - t0;
  sink[tid] = c0[0] + c1[1] + c2[2] + c3[3] + x;
}
int main() {
  long long *out; double *sink;
  hipMalloc(&out, 64); hipMalloc(&sink, 8 * 256);
  for (int threads : {64, 256}) {
    hipLaunchKernelGGL(k, dim3(1), dim3(threads), 0, 0, out, sink);
    long long h[5]; hipMemcpy(h, out, 40, hipMemcpyDeviceToHost);
    printf("threads %d: cycles per v_mfma_f64_16x16x4: dependent %.1f, 2 accumulators %.1f, 4 accumulators %.1f, result->operand %.1f; dependent v_fma_f64 %.1f\n",
           threads, h[0] / (double)N, h[1] / (double)N, h[2] / (double)N, h[3] / (double)N, h[4] / (double)N);
  }
  return 0;
}
