// Shared device helpers for libt3d (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "t3d.h"
#include "abi_take.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// hipGetLastError() is process-wide and sticky: clear whatever an earlier (unrelated) HIP call left behind
// before launching, so that T3D_CHECK_LAUNCH reports only this launch.
#define T3D_LAUNCH(...)          \
  do {                           \
    (void)hipGetLastError();     \
    hipLaunchKernelGGL(__VA_ARGS__); \
  } while (0)

#define T3D_CHECK_LAUNCH()                                  \
  do {                                                      \
    hipError_t e__ = hipGetLastError();                     \
    if (e__ != hipSuccess) return T3D_ERR_LAUNCH;           \
  } while (0)

// XCD-aware block remap: blocks b and b+8 share an XCD (round-robin dispatch), so give each XCD a
// contiguous chunk of the logical tile order; bijective for any grid size.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
  const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + idx;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// ---------------------------------------------------------------------------------------------------------------------------
// Element types of the per-point layer tensors (t3d.h: T3D_F32 / T3D_BF16).  The ABI structs carry `float*` whatever the element
// type; device code reinterprets through these helpers.  bf16 <-> fp32 conversions are plain casts (v_cvt_pk_bf16_f32: round to
// nearest even, NaN stays NaN).
// ---------------------------------------------------------------------------------------------------------------------------
typedef __bf16 bf16_t;
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

template <class T> struct Elem;
template <> struct Elem<float> {
  static constexpr bool BF16 = false;
  typedef float4 V4;
  typedef float2 V2;
  __device__ __forceinline__ static V4 ld4(const float* base, size_t i) { return *reinterpret_cast<const float4*>(base + i); }
  __device__ __forceinline__ static float4 widen(const V4& v) { return v; }
  __device__ __forceinline__ static float2 ld2(const float* base, size_t i) { return *reinterpret_cast<const float2*>(base + i); }
  __device__ __forceinline__ static float ld1(const float* base, size_t i) { return base[i]; }
  __device__ __forceinline__ static void st1(float* base, size_t i, float v) { base[i] = v; }
  __device__ __forceinline__ static void st2(float* base, size_t i, float a, float b) { *reinterpret_cast<float2*>(base + i) = make_float2(a, b); }
  __device__ __forceinline__ static float rnd(float v) { return v; }      // value as it will be read back
};
template <> struct Elem<bf16_t> {
  static constexpr bool BF16 = true;
  typedef bf16x4 V4;
  typedef bf16x2 V2;
  __device__ __forceinline__ static V4 ld4(const float* base, size_t i) {
    return *reinterpret_cast<const bf16x4*>(reinterpret_cast<const bf16_t*>(base) + i);
  }
  __device__ __forceinline__ static float4 widen(const V4& v) { return make_float4((float)v[0], (float)v[1], (float)v[2], (float)v[3]); }
  __device__ __forceinline__ static float2 ld2(const float* base, size_t i) {
    const bf16x2 v = *reinterpret_cast<const bf16x2*>(reinterpret_cast<const bf16_t*>(base) + i);
    return make_float2((float)v[0], (float)v[1]);
  }
  __device__ __forceinline__ static float ld1(const float* base, size_t i) { return (float)reinterpret_cast<const bf16_t*>(base)[i]; }
  __device__ __forceinline__ static void st1(float* base, size_t i, float v) { reinterpret_cast<bf16_t*>(base)[i] = (bf16_t)v; }
  __device__ __forceinline__ static void st2(float* base, size_t i, float a, float b) {
    bf16x2 v = {(bf16_t)a, (bf16_t)b};
    *reinterpret_cast<bf16x2*>(reinterpret_cast<bf16_t*>(base) + i) = v;
  }
  __device__ __forceinline__ static float rnd(float v) { return (float)(bf16_t)v; }
};

// run-time (workgroup-uniform) element type, for the small kernels that are not instantiated per type
__device__ __forceinline__ float ld_elem(const float* base, size_t i, int dtype) {
  return dtype == T3D_BF16 ? Elem<bf16_t>::ld1(base, i) : base[i];
}
__device__ __forceinline__ float4 ld_elem4(const float* base, size_t i, int dtype) {
  return dtype == T3D_BF16 ? Elem<bf16_t>::widen(Elem<bf16_t>::ld4(base, i)) : *reinterpret_cast<const float4*>(base + i);
}
