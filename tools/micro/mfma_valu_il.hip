// What one v_mfma_f32_32x32x16_bf16 gap tolerates beside it on gfx950: [MFMA, NF single-issue vector instructions (, one LDS access)] in a
// loop, EVERY instruction an `asm volatile` statement (volatile asm statements keep their order: the emitted loop IS the source, see the
// .s next to the log), one and two waves per SIMD.  Replaces the compiler-scheduled rows of mfma_valu_overlap.hip (round 5), whose
// __builtin_fmaf streams hipcc packed into v_pk_fma_f32 and did not interleave ("What's weak" 5).
//   MIX 0   v_fma_f32 on eight independent registers
//   MIX 1   the x3 staging mix on four rotating register groups (v_fma_f32, v_max_f32, v_cvt_pk_bf16_f32, v_lshlrev_b32, v_and_b32,
//           v_sub_f32 x 2: batch-norm + ReLU + one term of the three-way split), consecutive instructions in different groups
//   MIX 2   MIX 1 plus one LDS access per gap (ds_write_b64 and ds_read_b128 alternating, conflict-free addresses)
//   hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_valu_il.hip -o /tmp/mvil && /tmp/mvil
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define MFMA(ACC) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(ACC) : "v"(a), "v"(b))

// one filler instruction: number n of the stream (compile-time), MIX as above
template <int MIX, int n>
__device__ __forceinline__ void filler(float (&x)[8], unsigned (&u)[4], float (&r)[8], float c0, float c1) {
  if constexpr (MIX == 0) {
    asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[n & 7]) : "v"(c0), "v"(c1));
  } else {
    constexpr int g = n & 3, k = (n >> 2) % 7;      // group g, instruction k of the group's cycle
    if constexpr (k == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[2 * g]) : "v"(c0), "v"(c1));
    else if constexpr (k == 1) asm volatile("v_max_f32 %0, %0, %1" : "+v"(x[2 * g + 1]) : "v"(c1));
    else if constexpr (k == 2) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(u[g]) : "v"(x[2 * g]), "v"(x[2 * g + 1]));
    else if constexpr (k == 3) asm volatile("v_lshlrev_b32 %0, 16, %1" : "=v"(r[2 * g]) : "v"(u[g]));
    else if constexpr (k == 4) asm volatile("v_and_b32 %0, 0xffff0000, %1" : "=v"(r[2 * g + 1]) : "v"(u[g]));
    else if constexpr (k == 5) asm volatile("v_sub_f32 %0, %1, %0" : "+v"(r[2 * g]) : "v"(x[2 * g]));
    else asm volatile("v_sub_f32 %0, %1, %0" : "+v"(r[2 * g + 1]) : "v"(x[2 * g + 1]));
  }
}
template <int MIX, int NF, int G, int J = 0>
__device__ __forceinline__ void fillers(float (&x)[8], unsigned (&u)[4], float (&r)[8], float c0, float c1) {
  if constexpr (J < NF) {
    filler<MIX, G * NF + J>(x, u, r, c0, c1);
    fillers<MIX, NF, G, J + 1>(x, u, r, c0, c1);
  }
}

template <int MIX, int NF, int G>
__device__ __forceinline__ void gap(f32x16& acc, const bf16x8& a, const bf16x8& b, float (&x)[8], unsigned (&u)[4], float (&r)[8], float c0, float c1,
                                    unsigned lds_w, unsigned lds_r, f32x4& rd) {
  MFMA(acc);
  fillers<MIX, NF, G>(x, u, r, c0, c1);
  if constexpr (MIX == 2) {
    if constexpr (G & 1) asm volatile("ds_read_b128 %0, %1" : "=v"(rd) : "v"(lds_r));
    else asm volatile("ds_write_b64 %0, %1" : : "v"(lds_w), "v"(*reinterpret_cast<const unsigned long long*>(&x[0])) : "memory");
  }
}

template <int MIX, int NF>
__global__ __launch_bounds__(512) void k_gap(const float* __restrict__ in, float* __restrict__ out, unsigned long long* __restrict__ stamps, int iters) {
  __shared__ __attribute__((aligned(16))) float lds[512 * 4 + 512 * 2];
  const int tid = threadIdx.x;
  bf16x8 a, b;
  for (int e = 0; e < 8; ++e) { a[e] = (__bf16)in[(tid + e) & 1023]; b[e] = (__bf16)in[(tid * 3 + e) & 1023]; }
  f32x16 acc0, acc1, acc2, acc3;
  for (int i = 0; i < 16; ++i) { acc0[i] = 0.f; acc1[i] = 0.f; acc2[i] = 0.f; acc3[i] = 0.f; }
  float x[8], r[8];
  unsigned u[4] = {0, 0, 0, 0};
  for (int e = 0; e < 8; ++e) { x[e] = in[(tid + 17 * e) & 1023]; r[e] = 0.f; }
  const float c0 = in[5], c1 = in[6];
  for (int i = tid; i < 512 * 6; i += blockDim.x) lds[i] = in[i & 1023];
  const unsigned lds_r = (unsigned)(size_t)(lds) + tid * 16, lds_w = (unsigned)(size_t)(lds + 512 * 4) + tid * 8;
  f32x4 rd = {0.f, 0.f, 0.f, 0.f};
  __syncthreads();
  const unsigned long long s0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {      // eight gaps per iteration, four accumulators
    gap<MIX, NF, 0>(acc0, a, b, x, u, r, c0, c1, lds_w, lds_r, rd);
    gap<MIX, NF, 1>(acc1, a, b, x, u, r, c0, c1, lds_w, lds_r, rd);
    gap<MIX, NF, 2>(acc2, a, b, x, u, r, c0, c1, lds_w, lds_r, rd);
    gap<MIX, NF, 3>(acc3, a, b, x, u, r, c0, c1, lds_w, lds_r, rd);
    gap<MIX, NF, 4>(acc0, a, b, x, u, r, c0, c1, lds_w, lds_r, rd);
    gap<MIX, NF, 5>(acc1, a, b, x, u, r, c0, c1, lds_w, lds_r, rd);
    gap<MIX, NF, 6>(acc2, a, b, x, u, r, c0, c1, lds_w, lds_r, rd);
    gap<MIX, NF, 7>(acc3, a, b, x, u, r, c0, c1, lds_w, lds_r, rd);
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  const unsigned long long s1 = __builtin_amdgcn_s_memtime();
  float s = rd[0] + rd[1] + rd[2] + rd[3];
  for (int i = 0; i < 16; ++i) s += acc0[i] + acc1[i] + acc2[i] + acc3[i];
  for (int e = 0; e < 8; ++e) s += x[e] + r[e];
  for (int e = 0; e < 4; ++e) s += (float)u[e];
  out[blockIdx.x * blockDim.x + tid] = s;
  if ((tid & 63) == 0) stamps[blockIdx.x * 8 + tid / 64] = s1 - s0;
}

template <int MIX, int NF>
void row(const float* in, float* out, unsigned long long* st, int iters) {
  double res[2];
  for (int t = 0; t < 2; ++t) {
    const int threads = 256 * (t + 1);
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((k_gap<MIX, NF>), dim3(256), dim3(threads), 0, 0, in, out, st, iters);
    (void)hipDeviceSynchronize();
    std::vector<unsigned long long> h(256 * 8);
    (void)hipMemcpy(h.data(), st, h.size() * 8, hipMemcpyDeviceToHost);
    std::vector<double> c;
    for (int b = 0; b < 256; ++b) for (int w = 0; w < threads / 64; ++w) c.push_back((double)h[b * 8 + w] / iters / 8.0);
    std::sort(c.begin(), c.end());
    res[t] = c[c.size() / 2];
  }
  static const char* mix[3] = {"v_fma_f32 x 8 chains", "x3 staging mix", "x3 staging mix + 1 LDS access"};
  // per MFMA of the SIMD: one wave -> cycles per own MFMA; two waves -> cycles per own MFMA / 2
  printf("[MFMA, %d x %-30s] 1 wave/SIMD %5.1f cycles per MFMA   2 waves/SIMD %5.1f cycles per MFMA of the SIMD (%5.1f per wave-MFMA)   matrix pipe alone 32.0\n",
         NF, mix[MIX], res[0], res[1] / 2.0, res[1]);
}

int main() {
  float* in; float* out; unsigned long long* st;
  (void)hipMalloc(&in, 4096); (void)hipMalloc(&out, 256 * 512 * 4); (void)hipMalloc(&st, 256 * 8 * 8);
  std::vector<float> h(1024);
  for (int i = 0; i < 1024; ++i) h[i] = 0.5f + 0.001f * (i % 97);
  h[5] = 0.999f; h[6] = 0.001f;
  (void)hipMemcpy(in, h.data(), 4096, hipMemcpyHostToDevice);
  const int iters = 10000;
  row<0, 0>(in, out, st, iters); row<0, 2>(in, out, st, iters); row<0, 4>(in, out, st, iters); row<0, 5>(in, out, st, iters);
  row<0, 6>(in, out, st, iters); row<0, 7>(in, out, st, iters); row<0, 8>(in, out, st, iters); row<0, 10>(in, out, st, iters);
  row<1, 2>(in, out, st, iters); row<1, 4>(in, out, st, iters); row<1, 5>(in, out, st, iters); row<1, 6>(in, out, st, iters);
  row<1, 7>(in, out, st, iters); row<1, 8>(in, out, st, iters); row<1, 10>(in, out, st, iters);
  row<2, 3>(in, out, st, iters); row<2, 4>(in, out, st, iters); row<2, 5>(in, out, st, iters); row<2, 6>(in, out, st, iters);
  row<2, 7>(in, out, st, iters);
  return 0;
}
