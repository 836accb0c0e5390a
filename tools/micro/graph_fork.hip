// Does a captured hipGraph run a forked branch BESIDE the main chain on this runtime?  (round 6: the speculative marginalization)
//   main:  A(T) -> [fork] -> C(T) -> C(T) -> [join] -> D(short)
//   side:            B(2T)
// serial: 5T; concurrent: 3T.  Also: a chain of many short kernels on the main branch while one long kernel runs on the side.
// hipcc --offload-arch=gfx950 -O2 graph_fork.hip -o bin/graph_fork
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#define CK(x) do { hipError_t err_ = (x); if (err_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(err_)); return 1; } } while (0)
__global__ void spin(long long cycles, int *out) {
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < cycles) { }
  if (out) out[0] = 1;
}
int main() {
  hipStream_t s0, s1, s2;
  CK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
  hipEvent_t ef, ej, ef2, ej2;
  CK(hipEventCreateWithFlags(&ef, hipEventDisableTiming));
  CK(hipEventCreateWithFlags(&ej, hipEventDisableTiming));
  CK(hipEventCreateWithFlags(&ef2, hipEventDisableTiming));
  CK(hipEventCreateWithFlags(&ej2, hipEventDisableTiming));
  int *d;
  CK(hipMalloc(&d, 64));
  const long long T = 10000;  // wall_clock64 ticks at 100 MHz: 100 us
  for (int variant = 0; variant < 7; variant++) {
    hipGraph_t g;
    hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s0, hipStreamCaptureModeThreadLocal));
    if (variant == 0) {
      hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s0, T, d);
      CK(hipEventRecord(ef, s0));
      CK(hipStreamWaitEvent(s1, ef, 0));
      hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s1, 2 * T, d);
      CK(hipEventRecord(ej, s1));
      hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s0, T, d);
      hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s0, T, d);
      CK(hipStreamWaitEvent(s0, ej, 0));
      hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s0, 10, d);
    } else if (variant == 1) {
      // 40 short kernels (5 us) on the main branch beside one 200 us kernel: 200 us if concurrent, 400 if serial
      hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s0, 10, d);
      CK(hipEventRecord(ef, s0));
      CK(hipStreamWaitEvent(s1, ef, 0));
      hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s1, 2 * T, d);
      CK(hipEventRecord(ej, s1));
      for (int k = 0; k < 40; k++) hipLaunchKernelGGL(spin, dim3(40), dim3(256), 0, s0, T / 20, d);
      CK(hipStreamWaitEvent(s0, ej, 0));
      hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s0, 10, d);
    } else if (variant >= 3) {
      // 3: 40 kernels of 5 us alone.  4: the same with a fork after kernels 0, 10, 20, 30 — three trivial kernels each, on alternating
      // side streams, all joined at the end.  5: all forks on ONE side stream.  6: like 4, and the main chain waits for the first
      // kernel of branch k before its kernel 10 k + 5 (the snapshot dependency)
      hipEvent_t e[8], f[4];
      for (int k = 0; k < 8; k++) CK(hipEventCreateWithFlags(&e[k], hipEventDisableTiming));
      for (int k = 0; k < 4; k++) CK(hipEventCreateWithFlags(&f[k], hipEventDisableTiming));
      for (int k = 0; k < 40; k++) {
        if (variant == 6 && k % 10 == 5) CK(hipStreamWaitEvent(s0, f[k / 10], 0));
        hipLaunchKernelGGL(spin, dim3(40), dim3(256), 0, s0, T / 20, d);
        if (variant >= 4 && k % 10 == 0) {
          hipStream_t ss = variant == 5 ? s1 : (k / 10) % 2 ? s2 : s1;
          CK(hipEventRecord(e[k / 10], s0));
          CK(hipStreamWaitEvent(ss, e[k / 10], 0));
          hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, ss, 10, d);
          if (variant == 6) CK(hipEventRecord(f[k / 10], ss));
          for (int q = 0; q < 2; q++) hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, ss, 10, d);
          CK(hipEventRecord(e[4 + k / 10], ss));
        }
      }
      if (variant >= 4) for (int k = 0; k < 4; k++) CK(hipStreamWaitEvent(s0, e[4 + k], 0));
    } else {
      // two forks in flight at once (branch k started after main kernel k), each 2T long; main 4 x T: 4T (+) if all concurrent
      hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s0, T, d);
      CK(hipEventRecord(ef, s0));
      CK(hipStreamWaitEvent(s1, ef, 0));
      hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s1, 2 * T, d);
      CK(hipEventRecord(ej, s1));
      hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s0, T, d);
      CK(hipEventRecord(ef2, s0));
      CK(hipStreamWaitEvent(s2, ef2, 0));
      hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s2, 2 * T, d);
      CK(hipEventRecord(ej2, s2));
      hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s0, T, d);
      hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s0, T, d);
      CK(hipStreamWaitEvent(s0, ej, 0));
      CK(hipStreamWaitEvent(s0, ej2, 0));
      hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s0, 10, d);
    }
    CK(hipStreamEndCapture(s0, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int rep = 0; rep < 3; rep++) {
      CK(hipStreamSynchronize(s0));
      const auto t0 = std::chrono::steady_clock::now();
      CK(hipGraphLaunch(ge, s0));
      CK(hipStreamSynchronize(s0));
      const double us = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() * 1e6;
      printf("variant %d rep %d: %.1f us  (%s)\n", variant, rep, us,
             variant == 0 ? "serial 500, concurrent 300" : variant == 1 ? "serial 400+, concurrent ~200" : variant == 2 ? "serial 800, concurrent 400" : "40 x 5 us");
    }
    CK(hipGraphExecDestroy(ge));
    CK(hipGraphDestroy(g));
  }
  return 0;
}
