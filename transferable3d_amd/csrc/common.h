// Shared device helpers for libt3d (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "t3d.h"

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
