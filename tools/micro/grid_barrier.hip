// Micro-benchmark (diagnosis only): a chain of dependent small stages as (a) one launch per stage, (b) one launch with a
// grid barrier between the stages.  Every stage: workgroup b reads what workgroup (b+1)%G wrote in the previous stage.
#include <hip/hip_runtime.h>
#include <stdint.h>

template <int MODE>
__device__ __forceinline__ void grid_barrier(unsigned* counter, unsigned target) {
  __syncthreads();
  if (threadIdx.x == 0) {
    if (MODE == 0) {
      __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      unsigned spins = 0;
      while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target && ++spins < (1u << 22)) __builtin_amdgcn_s_sleep(1);
    } else {
      __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      unsigned spins = 0;
      while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target && ++spins < (1u << 24)) { if (MODE == 1) __builtin_amdgcn_s_sleep(1); }
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
  }
  __syncthreads();
}

__device__ __forceinline__ void stage_body(const float* __restrict__ in, float* __restrict__ out, int G, int b, int width) {
  // width floats per workgroup
  for (int i = threadIdx.x; i < width; i += blockDim.x) {
    const float v = __builtin_nontemporal_load(in + (size_t)((b + 1) % G) * width + i);
    out[(size_t)b * width + i] = v + 1.0f;
  }
}

__global__ void k_stage(const float* in, float* out, int G, int width, int stride_blocks) {
  if (blockIdx.x % stride_blocks) return;
  stage_body(in, out, G, blockIdx.x / stride_blocks, width);
}

template <int MODE>
__global__ void k_chain(float* buf, int G, int width, int n, unsigned* counter, int stride_blocks) {
  if (blockIdx.x % stride_blocks) return;
  const int b = blockIdx.x / stride_blocks;
  const unsigned base = *reinterpret_cast<volatile unsigned*>(counter + 1);     // epoch * G, advanced by the host-side reset kernel
  for (int s = 0; s < n; ++s) {
    stage_body(buf + (size_t)(s & 1) * G * width, buf + (size_t)((s + 1) & 1) * G * width, G, b, width);
    if (s + 1 < n) { if (MODE == 0) __threadfence(); grid_barrier<MODE>(counter, base + (unsigned)(s + 1) * G); }
  }
}

__global__ void k_reset(unsigned* counter) { counter[0] = 0; counter[1] = 0; }

extern "C" int mb_stage(void* in, void* out, int G, int width, int stride_blocks, void* stream) {
  hipLaunchKernelGGL(k_stage, dim3(G * stride_blocks), dim3(512), 0, (hipStream_t)stream, (const float*)in, (float*)out, G, width, stride_blocks);
  return (int)hipGetLastError();
}
extern "C" int mb_chain(void* buf, int G, int width, int n, void* counter, int stride_blocks, void* stream, int mode) {
  hipLaunchKernelGGL(k_reset, dim3(1), dim3(1), 0, (hipStream_t)stream, (unsigned*)counter);
  if (mode == 0) hipLaunchKernelGGL(k_chain<0>, dim3(G * stride_blocks), dim3(512), 0, (hipStream_t)stream, (float*)buf, G, width, n, (unsigned*)counter, stride_blocks);
  else if (mode == 1) hipLaunchKernelGGL(k_chain<1>, dim3(G * stride_blocks), dim3(512), 0, (hipStream_t)stream, (float*)buf, G, width, n, (unsigned*)counter, stride_blocks);
  else hipLaunchKernelGGL(k_chain<2>, dim3(G * stride_blocks), dim3(512), 0, (hipStream_t)stream, (float*)buf, G, width, n, (unsigned*)counter, stride_blocks);
  return (int)hipGetLastError();
}
