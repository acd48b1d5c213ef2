// How the two waves of a SIMD share it when each runs [G matrix instructions, V vector instructions] in a loop -- the shape of the x3
// GEMM main loop of csrc/pointmlp.hip (six groups of four MFMAs with ~22 staging instructions between them) without memory, LDS or
// barriers.  Prints cycles per iteration per wave for: MFMAs alone, vector work alone, both in ONE wave, and both with 1 / 2 waves per
// SIMD, with the second wave optionally delayed by half an iteration.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_valu_overlap.hip -o /tmp/ovl && /tmp/ovl
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int G, int V, bool DO_M, bool DO_V>
__global__ __launch_bounds__(512) void k(const float* __restrict__ in, float* __restrict__ out, unsigned long long* __restrict__ stamps, int iters,
                                         int delay_second) {
  const int tid = threadIdx.x;
  bf16x8 a, b;
  for (int e = 0; e < 8; ++e) { a[e] = (__bf16)in[(tid + e) & 1023]; b[e] = (__bf16)in[(tid * 3 + e) & 1023]; }
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float v[8];
  for (int e = 0; e < 8; ++e) v[e] = in[(tid + 17 * e) & 1023];
  const float c0 = in[5], c1 = in[6];
  if (delay_second && tid >= 256) {      // the second wave of every SIMD starts later
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    while (__builtin_amdgcn_s_memtime() - t0 < (unsigned long long)delay_second) {}
  }
  const unsigned long long s0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    if (DO_M) {
#pragma unroll
      for (int g = 0; g < G; ++g) acc[g & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[g & 3], 0, 0, 0);
    }
    if (DO_V) {
#pragma unroll
      for (int j = 0; j < V; ++j) v[j & 7] = __builtin_fmaf(v[j & 7], c0, c1);      // eight independent chains
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  const unsigned long long s1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  for (int e = 0; e < 8; ++e) s += v[e];
  out[blockIdx.x * blockDim.x + tid] = s;
  if ((tid & 63) == 0) stamps[blockIdx.x * 8 + tid / 64] = s1 - s0;
}

// the same work with the vector instructions INTERLEAVED between the matrix instructions: [MFMA, V / G vector instructions] x G
template <int G, int V>
__global__ __launch_bounds__(512) void k_il(const float* __restrict__ in, float* __restrict__ out, unsigned long long* __restrict__ stamps, int iters) {
  const int tid = threadIdx.x;
  bf16x8 a, b;
  for (int e = 0; e < 8; ++e) { a[e] = (__bf16)in[(tid + e) & 1023]; b[e] = (__bf16)in[(tid * 3 + e) & 1023]; }
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float v[8];
  for (int e = 0; e < 8; ++e) v[e] = in[(tid + 17 * e) & 1023];
  const float c0 = in[5], c1 = in[6];
  __syncthreads();
  const unsigned long long s0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int g = 0; g < G; ++g) {
      acc[g & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[g & 3], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < V / G; ++j) v[(g * (V / G) + j) & 7] = __builtin_fmaf(v[(g * (V / G) + j) & 7], c0, c1);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  const unsigned long long s1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  for (int e = 0; e < 8; ++e) s += v[e];
  out[blockIdx.x * blockDim.x + tid] = s;
  if ((tid & 63) == 0) stamps[blockIdx.x * 8 + tid / 64] = s1 - s0;
}
// ... with the accumulators in AGPRs (inline asm, constraint "a"): does the matrix instruction then leave the vector ports alone?
template <int G, int V>
__global__ __launch_bounds__(512) void k_il_agpr(const float* __restrict__ in, float* __restrict__ out, unsigned long long* __restrict__ stamps, int iters) {
  const int tid = threadIdx.x;
  bf16x8 a, b;
  for (int e = 0; e < 8; ++e) { a[e] = (__bf16)in[(tid + e) & 1023]; b[e] = (__bf16)in[(tid * 3 + e) & 1023]; }
  f32x16 acc0, acc1, acc2, acc3;
  for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; acc2[r] = 0.f; acc3[r] = 0.f; }
  float v[8];
  for (int e = 0; e < 8; ++e) v[e] = in[(tid + 17 * e) & 1023];
  const float c0 = in[5], c1 = in[6];
  __syncthreads();
  const unsigned long long s0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int g = 0; g < G; ++g) {
      if ((g & 3) == 0) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc0) : "v"(a), "v"(b));
      else if ((g & 3) == 1) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc1) : "v"(a), "v"(b));
      else if ((g & 3) == 2) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc2) : "v"(a), "v"(b));
      else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc3) : "v"(a), "v"(b));
#pragma unroll
      for (int j = 0; j < V / G; ++j) {
        float& x = v[(g * (V / G) + j) & 7];
        asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(c0), "v"(c1));
      }
    }
  }
  const unsigned long long s1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
  for (int r = 0; r < 16; ++r) s += acc0[r] + acc1[r] + acc2[r] + acc3[r];
  for (int e = 0; e < 8; ++e) s += v[e];
  out[blockIdx.x * blockDim.x + tid] = s;
  if ((tid & 63) == 0) stamps[blockIdx.x * 8 + tid / 64] = s1 - s0;
}
// ... and with the vector work of the x3 staging pass (one pair of split3: cvt_pk, shift, mask, two subtractions per term) instead of
// independent fmas: six instructions per MFMA, AGPR or VGPR accumulators
template <int G, bool AGPR>
__global__ __launch_bounds__(512) void k_il_split(const float* __restrict__ in, float* __restrict__ out, unsigned long long* __restrict__ stamps, int iters) {
  const int tid = threadIdx.x;
  bf16x8 a, b;
  for (int e = 0; e < 8; ++e) { a[e] = (__bf16)in[(tid + e) & 1023]; b[e] = (__bf16)in[(tid * 3 + e) & 1023]; }
  f32x16 acc0, acc1, acc2, acc3;
  for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; acc2[r] = 0.f; acc3[r] = 0.f; }
  float x0 = in[tid & 1023], x1 = in[(tid + 64) & 1023];
  unsigned sink = 0;
  __syncthreads();
  const unsigned long long s0 = __builtin_amdgcn_s_memtime();
#define T3D_M_(ACC)                                                                                         \
  do {                                                                                                      \
    if (AGPR) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(ACC) : "v"(a), "v"(b));         \
    else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(ACC) : "v"(a), "v"(b));              \
  } while (0)
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int g = 0; g < G; ++g) {
      if ((g & 3) == 0) T3D_M_(acc0); else if ((g & 3) == 1) T3D_M_(acc1); else if ((g & 3) == 2) T3D_M_(acc2); else T3D_M_(acc3);
      unsigned hp; float r0, r1, h0, h1;
      asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(hp) : "v"(x0), "v"(x1));
      asm volatile("v_lshlrev_b32 %0, 16, %1" : "=v"(h0) : "v"(hp));
      asm volatile("v_and_b32 %0, 0xffff0000, %1" : "=v"(h1) : "v"(hp));
      asm volatile("v_sub_f32 %0, %1, %2" : "=v"(r0) : "v"(x0), "v"(h0));
      asm volatile("v_sub_f32 %0, %1, %2" : "=v"(r1) : "v"(x1), "v"(h1));
      asm volatile("v_xor_b32 %0, %0, %1" : "+v"(sink) : "v"(hp));
      x0 = r0 + 1.0f; x1 = r1 + 1.0f;      // (two more: the next pair's inputs)
    }
  }
#undef T3D_M_
  const unsigned long long s1 = __builtin_amdgcn_s_memtime();
  float s = x0 + x1 + (float)sink;
  for (int r = 0; r < 16; ++r) s += acc0[r] + acc1[r] + acc2[r] + acc3[r];
  out[blockIdx.x * blockDim.x + tid] = s;
  if ((tid & 63) == 0) stamps[blockIdx.x * 8 + tid / 64] = s1 - s0;
}
template <int G, bool AGPR>
void run_il_split(const float* in, float* out, unsigned long long* st, int iters) {
  double r[2];
  for (int t = 0; t < 2; ++t) {
    const int threads = 256 * (t + 1);
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((k_il_split<G, AGPR>), dim3(256), dim3(threads), 0, 0, in, out, st, iters);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(256 * 8);
    hipMemcpy(h.data(), st, h.size() * 8, hipMemcpyDeviceToHost);
    std::vector<double> c;
    for (int b = 0; b < 256; ++b) for (int w = 0; w < threads / 64; ++w) c.push_back((double)h[b * 8 + w] / iters);
    std::sort(c.begin(), c.end());
    r[t] = c[c.size() / 2];
  }
  printf("%s accumulators, [MFMA, 8 instructions of a split3 pair] x %d: 1 wave/SIMD %6.1f cycles per iteration (matrix pipe alone %d), 2 waves/SIMD %6.1f (matrix pipe alone %d)\n",
         AGPR ? "AGPR" : "VGPR", G, r[0], G * 32, r[1], 2 * G * 32);
}

template <int G, int V>
void run_il_agpr(const float* in, float* out, unsigned long long* st, int iters) {
  double r[2];
  for (int t = 0; t < 2; ++t) {
    const int threads = 256 * (t + 1);
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((k_il_agpr<G, V>), dim3(256), dim3(threads), 0, 0, in, out, st, iters);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(256 * 8);
    hipMemcpy(h.data(), st, h.size() * 8, hipMemcpyDeviceToHost);
    std::vector<double> c;
    for (int b = 0; b < 256; ++b) for (int w = 0; w < threads / 64; ++w) c.push_back((double)h[b * 8 + w] / iters);
    std::sort(c.begin(), c.end());
    r[t] = c[c.size() / 2];
  }
  printf("AGPR accumulators, interleaved [MFMA, %d vector instructions] x %d: 1 wave/SIMD %6.1f cycles per iteration (matrix pipe alone %d), 2 waves/SIMD %6.1f (matrix pipe alone %d)\n",
         V / G, G, r[0], G * 32, r[1], 2 * G * 32);
}

template <int G, int V>
void run_il(const float* in, float* out, unsigned long long* st, int iters) {
  double r[2];
  for (int t = 0; t < 2; ++t) {
    const int threads = 256 * (t + 1);
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((k_il<G, V>), dim3(256), dim3(threads), 0, 0, in, out, st, iters);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(256 * 8);
    hipMemcpy(h.data(), st, h.size() * 8, hipMemcpyDeviceToHost);
    std::vector<double> c;
    for (int b = 0; b < 256; ++b) for (int w = 0; w < threads / 64; ++w) c.push_back((double)h[b * 8 + w] / iters);
    std::sort(c.begin(), c.end());
    r[t] = c[c.size() / 2];
  }
  printf("interleaved [MFMA, %d vector instructions] x %d: 1 wave/SIMD %6.1f cycles per iteration (matrix pipe alone %d), 2 waves/SIMD %6.1f (matrix pipe alone %d)\n",
         V / G, G, r[0], G * 32, r[1], 2 * G * 32);
}

// wave roles: the first wave of every SIMD (tid < 256) runs ONLY the G matrix instructions per iteration, the second ONLY the V vector
// instructions; both time their own loop (is a SIMD's vector work free beside another wave's matrix work?)
template <int G, int V>
__global__ __launch_bounds__(512) void k_roles(const float* __restrict__ in, float* __restrict__ out, unsigned long long* __restrict__ stamps, int iters,
                                               int prio_role) {
  const int tid = threadIdx.x;
  const bool mrole = tid < 256;
  if (prio_role == 1 && mrole) __builtin_amdgcn_s_setprio(2);
  if (prio_role == 2 && !mrole) __builtin_amdgcn_s_setprio(2);
  bf16x8 a, b;
  for (int e = 0; e < 8; ++e) { a[e] = (__bf16)in[(tid + e) & 1023]; b[e] = (__bf16)in[(tid * 3 + e) & 1023]; }
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float v[8];
  for (int e = 0; e < 8; ++e) v[e] = in[(tid + 17 * e) & 1023];
  const float c0 = in[5], c1 = in[6];
  __syncthreads();
  const unsigned long long s0 = __builtin_amdgcn_s_memtime();
  if (mrole) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int g = 0; g < G; ++g) acc[g & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[g & 3], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  } else {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int j = 0; j < V; ++j) v[j & 7] = __builtin_fmaf(v[j & 7], c0, c1);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  const unsigned long long s1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  for (int e = 0; e < 8; ++e) s += v[e];
  out[blockIdx.x * blockDim.x + tid] = s;
  if ((tid & 63) == 0) stamps[blockIdx.x * 8 + tid / 64] = s1 - s0;
}

template <int G, int V>
void run_roles(int prio, const float* in, float* out, unsigned long long* st, int iters) {
  for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((k_roles<G, V>), dim3(256), dim3(512), 0, 0, in, out, st, iters, prio);
  hipDeviceSynchronize();
  std::vector<unsigned long long> h(256 * 8);
  hipMemcpy(h.data(), st, h.size() * 8, hipMemcpyDeviceToHost);
  std::vector<double> cm, cv;
  for (int b = 0; b < 256; ++b) for (int w = 0; w < 8; ++w) (w < 4 ? cm : cv).push_back((double)h[b * 8 + w] / iters);
  std::sort(cm.begin(), cm.end()); std::sort(cv.begin(), cv.end());
  printf("roles G=%d V=%2d prio %d: the matrix wave %6.1f cycles per iteration (alone %d), the vector wave %6.1f (alone ~%d)\n", G, V, prio,
         cm[cm.size() / 2], G * 32, cv[cv.size() / 2], V * 4);
}

template <int G, int V, bool M, bool VV>
double run(int threads, int delay, const float* in, float* out, unsigned long long* st, int iters) {
  for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((k<G, V, M, VV>), dim3(256), dim3(threads), 0, 0, in, out, st, iters, delay);
  hipDeviceSynchronize();
  std::vector<unsigned long long> h(256 * 8);
  hipMemcpy(h.data(), st, h.size() * 8, hipMemcpyDeviceToHost);
  std::vector<double> c;
  for (int b = 0; b < 256; ++b) for (int w = 0; w < threads / 64; ++w) c.push_back((double)h[b * 8 + w] / iters);
  std::sort(c.begin(), c.end());
  return c[c.size() / 2];
}

int main() {
  float* in; float* out; unsigned long long* st;
  hipMalloc(&in, 4096); hipMalloc(&out, 256 * 512 * 4); hipMalloc(&st, 256 * 8 * 8);
  std::vector<float> h(1024);
  for (int i = 0; i < 1024; ++i) h[i] = 0.5f + 0.001f * (i % 97);
  h[5] = 0.999f; h[6] = 0.001f;
  hipMemcpy(in, h.data(), 4096, hipMemcpyHostToDevice);
  const int iters = 20000;
#define ROW(G, V)                                                                                                                  \
  printf("G=%d MFMA + V=%2d VALU per iteration | 1 wave/SIMD: mfma %6.1f  valu %6.1f  both %6.1f | 2 waves/SIMD: mfma %6.1f  valu %6.1f  both %6.1f  "  \
         "both, 2nd wave delayed by 64 / 150 / 1000 cycles %6.1f %6.1f %6.1f   (cycles per iteration per wave; ideal 2-wave both = max(2 x %d, ...))\n",       \
         G, V, run<G, V, true, false>(256, 0, in, out, st, iters), run<G, V, false, true>(256, 0, in, out, st, iters),              \
         run<G, V, true, true>(256, 0, in, out, st, iters), run<G, V, true, false>(512, 0, in, out, st, iters),                     \
         run<G, V, false, true>(512, 0, in, out, st, iters), run<G, V, true, true>(512, 0, in, out, st, iters),                     \
         run<G, V, true, true>(512, 64, in, out, st, iters), run<G, V, true, true>(512, 150, in, out, st, iters),                   \
         run<G, V, true, true>(512, 1000, in, out, st, iters), G * 32)
  run_il_split<4, true>(in, out, st, iters);
  run_il_split<4, false>(in, out, st, iters);
  run_il_agpr<4, 0>(in, out, st, iters);
  run_il_agpr<4, 8>(in, out, st, iters);
  run_il_agpr<4, 16>(in, out, st, iters);
  run_il_agpr<4, 24>(in, out, st, iters);
  run_il_agpr<4, 32>(in, out, st, iters);
  run_il_agpr<4, 48>(in, out, st, iters);
  run_il<4, 8>(in, out, st, iters);
  run_il<4, 16>(in, out, st, iters);
  run_il<4, 24>(in, out, st, iters);
  run_il<4, 32>(in, out, st, iters);
  run_il<4, 48>(in, out, st, iters);
  run_il<24, 144>(in, out, st, iters);
  run_roles<4, 8>(0, in, out, st, iters);
  run_roles<4, 16>(0, in, out, st, iters);
  run_roles<4, 22>(0, in, out, st, iters);
  run_roles<4, 32>(0, in, out, st, iters);
  run_roles<4, 48>(0, in, out, st, iters);
  run_roles<4, 22>(1, in, out, st, iters);
  run_roles<4, 22>(2, in, out, st, iters);
  run_roles<4, 32>(1, in, out, st, iters);
  run_roles<4, 32>(2, in, out, st, iters);
  ROW(4, 8);
  ROW(4, 16);
  ROW(4, 22);
  ROW(4, 32);
  ROW(4, 48);
  ROW(8, 44);
  ROW(24, 132);
  return 0;
}
