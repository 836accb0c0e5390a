// cost of a grid-wide barrier between a few dozen persistent workgroups (atomic counter in device memory, agent-scope
// fences), with and without a data hand-off across it; optionally with the stream confined to one XCD by a CU mask
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__device__ __forceinline__ void grid_barrier(unsigned *counter, unsigned nblocks, unsigned &target) {
  __syncthreads();
  if (threadIdx.x == 0) {
    __threadfence();
    target += nblocks;
    atomicAdd(counter, 1u);
    while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
    __threadfence();
  }
  __syncthreads();
}
__global__ __launch_bounds__(256) void k_bar(unsigned *counter, int rounds, double *buf, int handoff, long long *clk) {
  unsigned target = 0;
  const int nb = gridDim.x, b = blockIdx.x;
  double acc = 0;
  long long t0 = __builtin_readcyclecounter();
  for (int r = 0; r < rounds; r++) {
    if (handoff) buf[(size_t)b * 256 + threadIdx.x] = r + acc;       // 2 KB per block
    grid_barrier(counter, nb, target);
    if (handoff) acc += buf[(size_t)((b + 1) % nb) * 256 + threadIdx.x];  // the neighbour's, written before the barrier
  }
  long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) clk[b] = t1 - t0;
  if (handoff && acc == -1.0) buf[0] = acc;
}
static double run(hipStream_t s, int nb, int rounds, int handoff, unsigned *counter, double *buf, long long *clk) {
  hipMemsetAsync(counter, 0, 4, s);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0, s);
  hipLaunchKernelGGL(k_bar, dim3(nb), dim3(256), 0, s, counter, rounds, buf, handoff, clk);
  hipEventRecord(e1, s); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e3 / rounds;
}
int main() {
  unsigned *counter; double *buf; long long *clk;
  hipMalloc(&counter, 256); hipMalloc(&buf, 8 * 256 * 256); hipMalloc(&clk, 8 * 256);
  hipStream_t s0; hipStreamCreate(&s0);
  // one XCD: CU mask with 32 bits set -- the numbering of CUs over XCDs is probed by trying both a contiguous and a strided mask
  hipStream_t s1, s2;
  std::vector<uint32_t> contiguous(8, 0), strided(8, 0);
  contiguous[0] = 0xffffffffu;
  for (int i = 0; i < 256; i += 8) strided[i / 32] |= 1u << (i % 32);
  hipExtStreamCreateWithCUMask(&s1, 8, contiguous.data());
  hipExtStreamCreateWithCUMask(&s2, 8, strided.data());
  const int rounds = 2000;
  for (int nb : {8, 16, 24, 32}) {
    for (int handoff : {0, 1}) {
      run(s0, nb, 50, handoff, counter, buf, clk);
      printf("blocks %2d handoff %d: us per barrier  all CUs %.2f | CU mask 0..31 %.2f | CU mask every 8th %.2f\n", nb, handoff,
             run(s0, nb, rounds, handoff, counter, buf, clk), run(s1, nb, rounds, handoff, counter, buf, clk),
             run(s2, nb, rounds, handoff, counter, buf, clk));
    }
  }
  return 0;
}
