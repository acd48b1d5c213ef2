// What the bf16 matrix pipe of an MI355X sustains, and at which clock: every CU runs WPS waves per SIMD of back-to-back
// v_mfma_f32_32x32x16_bf16 on NACC independent accumulators and random operands for `iters` MFMAs per wave; the kernel stamps
// s_memtime (shader cycles) and s_memrealtime (100 MHz) around the loop.  Prints cycles per MFMA per SIMD, the in-kernel clock and the
// chip-wide dense bf16 TFLOP/s.  (The x3 GEMMs of csrc/pointmlp.hip are priced against this, not against 2.5 PFLOP/s at 2.4 GHz.)
//   hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_rate.hip -o /tmp/mfma_rate && /tmp/mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int NACC>
__global__ __launch_bounds__(1024) void k_mfma(const unsigned* __restrict__ seed, float* __restrict__ out, unsigned long long* __restrict__ stamps,
                                               int iters) {
  const int tid = threadIdx.x;
  unsigned s0 = seed[(blockIdx.x * blockDim.x + tid) & 65535];
  bf16x8 a[2], b[2];
  for (int j = 0; j < 2; ++j)
    for (int e = 0; e < 8; ++e) {
      s0 = s0 * 1664525u + 1013904223u;
      a[j][e] = (__bf16)((float)(int)(s0 >> 8) * (1.f / 8388608.f) - 1.f);
      s0 = s0 * 1664525u + 1013904223u;
      b[j][e] = (__bf16)((float)(int)(s0 >> 8) * (1.f / 8388608.f) - 1.f);
    }
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i)
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  __syncthreads();
  const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; it += NACC * 2) {
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[j], b[(i + j) & 1], acc[i], 0, 0, 0);
  }
  const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0.f;
  for (int i = 0; i < NACC; ++i)
    for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * blockDim.x + tid] = s;
  if ((tid & 63) == 0) {
    stamps[(blockIdx.x * (blockDim.x / 64) + tid / 64) * 2 + 0] = c1 - c0;
    stamps[(blockIdx.x * (blockDim.x / 64) + tid / 64) * 2 + 1] = r1 - r0;
  }
}

int main() {
  const int iters = 1 << 15, cus = 256;
  unsigned* seed; float* out; unsigned long long* st;
  hipMalloc(&seed, 65536 * 4); hipMalloc(&out, cus * 1024 * 4); hipMalloc(&st, cus * 16 * 16);
  std::vector<unsigned> h(65536);
  for (auto& v : h) v = rand();
  hipMemcpy(seed, h.data(), 65536 * 4, hipMemcpyHostToDevice);
  for (int wps : {1, 2, 4}) {
    for (int nacc : {1, 2, 4}) {
      const int threads = 256 * wps;
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      auto launch = [&]() {
        if (nacc == 1) hipLaunchKernelGGL(k_mfma<1>, dim3(cus), dim3(threads), 0, 0, seed, out, st, iters);
        else if (nacc == 2) hipLaunchKernelGGL(k_mfma<2>, dim3(cus), dim3(threads), 0, 0, seed, out, st, iters);
        else hipLaunchKernelGGL(k_mfma<4>, dim3(cus), dim3(threads), 0, 0, seed, out, st, iters);
      };
      for (int w = 0; w < 30; ++w) launch();      // warm: the clock settles under load
      hipEventRecord(e0);
      const int reps = 20;
      for (int r = 0; r < reps; ++r) launch();
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      std::vector<unsigned long long> hs(cus * wps * 4 * 2);
      hipMemcpy(hs.data(), st, hs.size() * 8, hipMemcpyDeviceToHost);
      std::vector<double> cyc, clk;
      for (size_t i = 0; i < hs.size() / 2; ++i) { cyc.push_back((double)hs[2 * i] / iters); clk.push_back((double)hs[2 * i] / (double)hs[2 * i + 1] * 0.1); }
      std::sort(cyc.begin(), cyc.end()); std::sort(clk.begin(), clk.end());
      const double flops = 2.0 * 32 * 32 * 16 * (double)iters * cus * wps * 4 * reps;
      printf("waves/SIMD %d  accumulators %d: %.1f cycles per MFMA per wave (median), in-kernel clock %.2f GHz (median; min %.2f max %.2f), %.0f TFLOP/s dense bf16 chip-wide\n",
             wps, nacc, cyc[cyc.size() / 2], clk[clk.size() / 2], clk.front(), clk.back(), flops / (ms * 1e-3) / 1e12);
    }
  }
  return 0;
}
