// micro-benchmarks of the primitives the latency-bound kernels (k_solve, k_marg_solve) are built from
#include <hip/hip_runtime.h>
#include <cstdio>
#define N 2000
__global__ void k(long long *out, double *sink, int threads_mode) {
  __shared__ double lds[4096];
  const int tid = threadIdx.x;
  double x = 1.0 + tid * 1e-9, y = 0.5;
  long long t0, t1;
  // 1. dependent FMA chain
  t0 = __builtin_readcyclecounter();
  for (int i = 0; i < N; i++) x = fma(x, 0.999999, y);
  t1 = __builtin_readcyclecounter();
  if (tid == 0) out[0] = (t1 - t0);
  // 2. barrier only
  __syncthreads();
  t0 = __builtin_readcyclecounter();
  for (int i = 0; i < N; i++) __syncthreads();
  t1 = __builtin_readcyclecounter();
  if (tid == 0) out[1] = (t1 - t0);
  // 3. LDS write -> barrier -> LDS read (other thread's value) -> dependent
  lds[tid] = x;
  __syncthreads();
  t0 = __builtin_readcyclecounter();
  for (int i = 0; i < N; i++) {
    lds[tid] = x;
    __syncthreads();
    x = lds[(tid + 17) & (blockDim.x - 1)] * 0.5 + 0.25;
    __syncthreads();
  }
  t1 = __builtin_readcyclecounter();
  if (tid == 0) out[2] = (t1 - t0);
  // 4. LDS read latency (dependent pointer chase)
  for (int i = tid; i < 4096; i += blockDim.x) lds[i] = (double)((i * 7 + 1) & 4095);
  __syncthreads();
  int p = tid;
  t0 = __builtin_readcyclecounter();
  for (int i = 0; i < N; i++) p = (int)lds[p];
  t1 = __builtin_readcyclecounter();
  if (tid == 0) out[3] = (t1 - t0);
  // 5. fast rsqrt chain (v_rsq_f64 + 2 Newton)
  t0 = __builtin_readcyclecounter();
  for (int i = 0; i < N; i++) {
    double r = __builtin_amdgcn_rsq(x);
    const double hx = 0.5 * x;
    r = fma(r, fma(-hx * r, r, 0.5), r);
    r = fma(r, fma(-hx * r, r, 0.5), r);
    x = r + 1.0;
  }
  t1 = __builtin_readcyclecounter();
  if (tid == 0) out[4] = (t1 - t0);
  // 6. readlane -> dependent
  t0 = __builtin_readcyclecounter();
  for (int i = 0; i < N; i++) {
    int lo = __builtin_amdgcn_readlane(__double2loint(x), 5), hi = __builtin_amdgcn_readlane(__double2hiint(x), 5);
    x = __hiloint2double(hi, lo) * 0.999 + 0.001 * tid;
  }
  t1 = __builtin_readcyclecounter();
  if (tid == 0) out[5] = (t1 - t0);
  // 7. independent FMAs (throughput): 16 accumulators
  double a[16];
  for (int j = 0; j < 16; j++) a[j] = x + j;
  t0 = __builtin_readcyclecounter();
  for (int i = 0; i < N; i++)
#pragma unroll
    for (int j = 0; j < 16; j++) a[j] = fma(a[j], 0.999999, y);
  t1 = __builtin_readcyclecounter();
  if (tid == 0) out[6] = (t1 - t0);
  for (int j = 0; j < 16; j++) x += a[j];
  // 8. IEEE divide + sqrt chain
  t0 = __builtin_readcyclecounter();
  for (int i = 0; i < N; i++) x = sqrt(x) / (x + 1.5) + 1.0;
  t1 = __builtin_readcyclecounter();
  if (tid == 0) out[7] = (t1 - t0);
  sink[tid] = x + p;
}
int main() {
  long long *out; double *sink;
  hipMalloc(&out, 64 * 8); hipMalloc(&sink, 1024 * 8);
  for (int th : {64, 256, 768, 1024}) {
    hipLaunchKernelGGL(k, dim3(1), dim3(th), 0, 0, out, sink, 0);
    hipDeviceSynchronize();
    long long h[8];
    hipMemcpy(h, out, sizeof h, hipMemcpyDeviceToHost);
    printf("threads %4d: cycles per op: fma-chain %.1f | barrier %.1f | lds write+bar+read+bar %.1f | lds read chase %.1f | fast rsqrt chain %.1f | readlane chain %.1f | 16 indep fma %.1f | sqrt+div chain %.1f\n",
           th, h[0] / (double)N, h[1] / (double)N, h[2] / (double)N, h[3] / (double)N, h[4] / (double)N, h[5] / (double)N, h[6] / (double)N, h[7] / (double)N);
  }
  // clock rate of the cycle counter
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0); for (int i = 0; i < 20; i++) hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, out, sink, 0); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  long long h[8]; hipMemcpy(h, out, sizeof h, hipMemcpyDeviceToHost);
  long long tot = 0; for (int i = 0; i < 8; i++) tot += h[i];
  printf("20 launches %.3f ms; counted cycles per launch %lld -> counter ~ %.0f MHz (lower bound)\n", ms, tot, tot / (ms / 20 * 1e3));
  return 0;
}
