// the trailing update of the blocked Cholesky in isolation: 55 tiles (kb = 0), four waves, LDS-resident tiles
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
constexpr int TLD = 17, TSZ = 16 * TLD, NT = 66;
__device__ __forceinline__ int tile_id(int a, int b) { return a * (a + 1) / 2 + b; }
template <int MODE>
__global__ void k(long long *out, double *sink) {
  extern __shared__ double Hs[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  for (int e = tid; e < NT * TSZ; e += 256) Hs[e] = 1.0 + 1e-3 * (e % 97);
  __syncthreads();
  const int c = lane & 15, gq = lane >> 4, offA = c * TLD + gq, offC = gq * TLD + c;
  long long t0 = __builtin_readcyclecounter();
  const int kb = 0;
  int u = 0;
  for (int ti = kb + 1; ti < 11; ti++)
    for (int tj = kb + 1; tj <= ti; tj++, u++) {
      if ((u & 3) != wave) continue;
      const double *Ta = Hs + tile_id(ti, kb) * TSZ + offA, *Tb = Hs + tile_id(tj, kb) * TSZ + offA;
      double *Tc = Hs + tile_id(ti, tj) * TSZ + offC;
      d4 acc;
      for (int r = 0; r < 4; r++) acc[r] = Tc[4 * TLD * r];
      if (MODE == 0) {
        for (int q = 0; q < 4; q++) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(-Ta[4 * q], Tb[4 * q], acc, 0, 0, 0);
      } else {
        for (int q = 0; q < 4; q++) acc[q] = fma(-Ta[4 * q], Tb[4 * q], acc[q]);
      }
      for (int r = 0; r < 4; r++) Tc[4 * TLD * r] = acc[r];
    }
  __syncthreads();
  long long t1 = __builtin_readcyclecounter();
  if (tid == 0) out[MODE] = t1 - t0;
  // batched: four tiles in flight per wave
  __syncthreads();
  t0 = __builtin_readcyclecounter();
  {
    int ti = kb + 1, tj = kb + 1, uu = 0;
    const int last = 55;
    auto advance = [&](int n) { for (int q = 0; q < n; q++, uu++) if (++tj > ti) ti++, tj = kb + 1; };
    advance(wave);
    while (uu < last) {
      double av[4][4], bv[4][4]; d4 cv[4]; double *Tc[4]; int nt = 0;
#pragma unroll
      for (int t = 0; t < 4; t++) {
        Tc[t] = nullptr;
        if (uu < last) {
          const double *Ta = Hs + tile_id(ti, kb) * TSZ + offA, *Tb = Hs + tile_id(tj, kb) * TSZ + offA;
          Tc[t] = Hs + tile_id(ti, tj) * TSZ + offC;
#pragma unroll
          for (int q = 0; q < 4; q++) av[t][q] = Ta[4 * q], bv[t][q] = Tb[4 * q], cv[t][q] = Tc[t][4 * TLD * q];
          nt = t + 1;
          advance(4);
        }
      }
#pragma unroll
      for (int t = 0; t < 4; t++) if (t < nt) {
#pragma unroll
        for (int q = 0; q < 4; q++) cv[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(-av[t][q], bv[t][q], cv[t], 0, 0, 0);
      }
#pragma unroll
      for (int t = 0; t < 4; t++) if (t < nt) {
#pragma unroll
        for (int r = 0; r < 4; r++) Tc[t][4 * TLD * r] = cv[t][r];
      }
    }
  }
  __syncthreads();
  t1 = __builtin_readcyclecounter();
  if (tid == 0) out[2 + MODE] = t1 - t0;
  sink[tid] = Hs[tid];
}
int main() {
  long long *out; double *sink;
  hipMalloc(&out, 64); hipMalloc(&sink, 8 * 256);
  hipFuncSetAttribute((const void *)k<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 150000);
  hipFuncSetAttribute((const void *)k<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 150000);
  for (int rep = 0; rep < 2; rep++) {
    hipLaunchKernelGGL(k<0>, dim3(1), dim3(256), 150000, 0, out, sink);
    hipLaunchKernelGGL(k<1>, dim3(1), dim3(256), 150000, 0, out, sink);
    long long h[4]; hipMemcpy(h, out, 32, hipMemcpyDeviceToHost);
    printf("U(kb=0), 55 tiles on 4 waves: with MFMA %lld cycles (%.0f per tile-round of 14), loads/stores + 4 FMAs only %lld\n", h[0], h[0] / 14.0, h[1]);
    printf("   four tiles in flight: %lld cycles (%.0f per tile-round)\n", h[2], h[2] / 14.0);
  }
  return 0;
}
