// what an early-returning kernel costs as a function of its resource footprint (back-to-back launches on one stream)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k_small(const int *flag) { if (*flag) return; }
__global__ void k_lds(const int *flag, double *out) {
  extern __shared__ double l[];
  if (*flag) return;
  l[threadIdx.x] = 1.0;
  __syncthreads();
  out[threadIdx.x] = l[(threadIdx.x + 1) & 63];
}
__global__ __launch_bounds__(256) void k_regs(const int *flag, double *out) {
  if (*flag) return;
  double a[120];
  for (int i = 0; i < 120; i++) a[i] = out[i + threadIdx.x];
  double s = 0;
  for (int r = 0; r < 4; r++)
    for (int i = 0; i < 120; i++) s = fma(a[i], a[(i * 7 + r) % 120], s), a[i] += s;
  out[threadIdx.x] = s;
}
template <class F> static double run(F f, int n) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 50; i++) f();
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int i = 0; i < n; i++) f();
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e3 / n;
}
int main() {
  int *flag; double *out; int one = 1;
  hipMalloc(&flag, 4); hipMalloc(&out, 8 * 4096); hipMemcpy(flag, &one, 4, hipMemcpyHostToDevice);
  hipFuncSetAttribute((const void *)k_lds, hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024);
  const int n = 2000;
  printf("us per early-returning launch, back to back: 1 x 64 threads %.2f | 27 x 256 threads %.2f | 1 x 256 threads + 159 KB LDS %.2f | 27 x 256 threads + 43 KB LDS %.2f | 27 x 256 threads, ~250 VGPRs %.2f\n",
         run([&] { hipLaunchKernelGGL(k_small, dim3(1), dim3(64), 0, 0, flag); }, n),
         run([&] { hipLaunchKernelGGL(k_small, dim3(27), dim3(256), 0, 0, flag); }, n),
         run([&] { hipLaunchKernelGGL(k_lds, dim3(1), dim3(256), 159 * 1024, 0, flag, out); }, n),
         run([&] { hipLaunchKernelGGL(k_lds, dim3(27), dim3(256), 43 * 1024, 0, flag, out); }, n),
         run([&] { hipLaunchKernelGGL(k_regs, dim3(27), dim3(256), 0, 0, flag, out); }, n));
  return 0;
}
