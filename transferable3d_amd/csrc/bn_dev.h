// Device bodies of the batch-norm-backward finalizer and the per-frustum dy column sums (bn_optim.hip) -- a header so that the
// paired small-launch kernel (pair.hip) can run them beside the FC kernels in one launch.
#pragma once
#include "common.h"
#include <math.h>

namespace {

// 256 threads = 16 channels x 16 tile groups: short, unrolled, independent loads (the 64-iteration serial
// loop of the first version was pure L2 latency: 19 us per launch in profiles/r01_baseline).
// At the row counts of config 4 (2048 row tiles) a 16-group block walks 128 tiles per thread -- eight dependent memory round trips
// (13.7 us per launch, profiles/r02_bf16_v2): above T3D_FIN_BIG tiles (default 512) the launchers take the GR = 64 instantiation
// (1024 threads, two round trips; only group 0 walks the 64 LDS partials): fwd finalize 195 -> 163 us per step, bwd 124 -> 112 at
// B=128 N=2048 (same-box A/B).  Up to 512 tiles GR stays 16, so the results at the sizes of configs 1-3 do not change by a bit.
// Still at config 4: N / 16 = 4 .. 32 such blocks per launch read all 2048 x N partials -- eight CUs' worth of address processing for
// the 128-channel layers, 12 us per launch, 29 launches per step.  Above T3D_FIN_WIDE tiles (default 1024) a block takes FOUR channels
// and 256 tile groups (`CH`, 1024 threads): four times the blocks, one round trip of eight tiles per thread, and the 256 partials per
// channel meet in two LDS levels (16 x 16).
// Back to back at 2048 tiles (tools/bench_finalize.py, us per launch, 64-group -> wide): N = 64 6.9 -> 5.0, 128 7.6 -> 5.3, 256 7.2 -> 7.3,
// 512 7.2 -> 14.9, 1024 8.6 -> 23.1 -- a 64-byte line then serves four blocks, and from 64 blocks on that traffic is the bound: the
// wide form is taken for N <= 128 only.
constexpr int FC_CH = 16, FC_GR = 16, FC_GR_BIG = 64, FC_CH_WIDE = 4, FC_GR_WIDE = 256, FC_WIDE_MAX_N = 128;

// Sixteen tiles per quantity are in flight per thread (32 loads for the two-quantity reductions): at 256 tiles the whole
// reduction is ONE memory round trip instead of four.
template <int NQ, int GR = FC_GR, int U = 16>
__device__ __forceinline__ void tile_sums(const float* const (&src)[NQ], int n_tiles, int N, int c, int grp, bool ok,
                                          double (&acc)[NQ]) {
#pragma unroll
  for (int q = 0; q < NQ; ++q) acc[q] = 0.0;
  if (!ok) return;
  for (int t0 = grp; t0 < n_tiles; t0 += U * GR) {
    float v[NQ][U];
#pragma unroll
    for (int q = 0; q < NQ; ++q)
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int t = t0 + u * GR;
        const float x = src[q][(size_t)min(t, n_tiles - 1) * N + c];     // clamped: no branch around the load
        v[q][u] = t < n_tiles ? x : 0.f;
      }
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      double a = 0.0;
#pragma unroll
      for (int u = 0; u < U; u += 4) a += ((double)v[q][u] + (double)v[q][u + 1]) + ((double)v[q][u + 2] + (double)v[q][u + 3]);
      acc[q] += a;
    }
  }
}

template <int GR = FC_GR, int CH = FC_CH>
__device__ __forceinline__ double group_reduce(double v, double (*red)[CH], int grp, int cl, bool act = true) {
  __syncthreads();
  if (act) red[grp][cl] = v;
  __syncthreads();
  double s = 0.0;
  if (GR > 64) {      // two levels: groups 0..15 sum 16 partials each, group 0 sums those (fixed order: reproducible)
    static_assert(GR <= 64 || GR == 256, "two-level reduction: 16 x 16 groups");
    if (grp < 16) {
#pragma unroll
      for (int g = 0; g < 16; ++g) s += red[grp * 16 + g][cl];
    }
    __syncthreads();
    if (grp < 16) red[grp][cl] = s;
    __syncthreads();
    s = 0.0;
    if (grp == 0) {
#pragma unroll
      for (int g = 0; g < 16; ++g) s += red[g][cl];
    }
  } else if (GR == FC_GR || grp == 0) {      // only group 0 uses the sum; with 64 groups the other 1008 threads' reads are pure LDS traffic
#pragma unroll
    for (int g = 0; g < GR; ++g) s += red[g][cl];
  }
  return s;
}

// `tid` may exceed GR * FC_CH (the 512-thread paired launch, pair.hip): the extra threads take part in the barriers only
template <int GR, int CH = FC_CH>
__device__ __forceinline__ void bn_bwd_finalize_body(const t3d_bn_bwd_finalize_args& p, double (*red)[CH], const int bid, const int tid) {
  const bool act = tid < GR * CH;
  const int cl = tid & (CH - 1), grp = act ? tid / CH : 0;
  const int c = bid * CH + cl;
  const bool ok = act && c < p.N;
  const int cc = ok ? c : 0;
  const float mean_f = p.mean[cc], invstd_f = p.invstd[cc], gamma_f = p.gamma[cc];   // ahead of the reduction
  double acc[2] = {0.0, 0.0};   // sum dz, sum dz*y
  if (p.psum_dz != nullptr) {
    const float* const src[2] = {p.psum_dz, p.psum_dzy};
    tile_sums<2, GR, (GR > 64 ? 8 : 16)>(src, p.n_tiles, p.N, c, grp, ok, acc);
  } else if (ok) {
    for (int b = grp; b < p.B; b += GR) {
      const float live = p.pooled[(size_t)b * p.ld_pooled + c] > 0.f ? 1.f : 0.f;
      const float g = p.dpool_in[(size_t)b * p.ld_dpool_in + c] * live;
      p.dpool[(size_t)b * p.N + c] = g;
      acc[0] += (double)g;
      acc[1] += (double)g * (double)p.ysel[(size_t)b * p.N + c];
    }
  }
  const double s1 = group_reduce<GR, CH>(acc[0], red, grp, cl, act);
  const double s2 = group_reduce<GR, CH>(acc[1], red, grp, cl, act);
  if (grp == 0 && ok) {
    if (p.frozen) {
      p.coef[c] = p.scale[c];
      p.coef[p.N + c] = 0.f;
      p.coef[2 * p.N + c] = 0.f;
      return;
    }
    const double mean = mean_f, invstd = invstd_f, gamma = gamma_f, n = p.count;
    const double dbeta = s1;
    const double dgamma = invstd * (s2 - mean * s1);       // sum dz * xhat
    if (p.dbeta) p.dbeta[c] = (float)dbeta;
    if (p.dgamma) p.dgamma[c] = (float)dgamma;
    // dy = gamma*invstd*(dz - dbeta/n - xhat*dgamma/n), xhat = (y-mean)*invstd
    const double c1 = gamma * invstd;
    const double k3 = dgamma / n * invstd;
    p.coef[c] = (float)c1;
    p.coef[p.N + c] = (float)(-c1 * k3);
    p.coef[2 * p.N + c] = (float)(c1 * (k3 * mean - dbeta / n));
  }
}

// pooled[b,c], argidx, ysel from the per-tile max/min partials (shared by k_pool_finalize and the fused finalize)
__device__ __forceinline__ void pool_pick_impl(const float* pmax, const float* pmin, const int32_t* pamax, const int32_t* pamin,
                                               int tiles_per_frustum, int N, int b, int c, float sc, float sh, float* pooled,
                                               int ld_pooled, int32_t* argidx, float* ysel) {
  const bool use_max = sc >= 0.f;
  float best = use_max ? -INFINITY : INFINITY;
  int arg = -1;
  // only the side the sign of the scale selects is read; eight tiles' loads are in flight together (clamped, then masked)
  const float* pv = use_max ? pmax : pmin;
  const int32_t* pa = use_max ? pamax : pamin;
  for (int t0 = 0; t0 < tiles_per_frustum; t0 += 8) {
    float v[8];
    int a[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const size_t o = (size_t)(b * tiles_per_frustum + min(t0 + u, tiles_per_frustum - 1)) * N + c;
      v[u] = pv[o];
      a[u] = pa[o];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const bool take = (t0 + u < tiles_per_frustum) && a[u] >= 0 && (arg < 0 || (use_max ? v[u] > best : v[u] < best));
      best = take ? v[u] : best;
      arg = take ? a[u] : arg;
    }
  }
  float out = 0.f;
  if (arg >= 0) out = fmaxf(fmaf(best, sc, sh), 0.f);
  const bool live = out > 0.f;
  const size_t i = (size_t)b * N + c;
  pooled[(size_t)b * ld_pooled + c] = out;
  argidx[i] = live ? arg : -1;
  ysel[i] = live ? best : 0.f;
}
__device__ __forceinline__ void pool_pick(const t3d_bn_fwd_finalize_args& p, int b, int c, float sc, float sh) {
  pool_pick_impl(p.pool_pmax, p.pool_pmin, p.pool_pamax, p.pool_pamin, p.pool_tiles_per_frustum, p.N, b, c, sc, sh, p.pooled,
                 p.ld_pooled, p.argidx, p.ysel);
}

// `red`: GR x CH doubles, `s_scsh`: 2 x CH floats of LDS (static arrays of the stand-alone kernel, a piece of the host launch's dynamic
// LDS when the body runs as a rider)
template <int GR, int CH = FC_CH>
__device__ __forceinline__ void bn_fwd_finalize_body(const t3d_bn_fwd_finalize_args& p, double (*red)[CH], float* s_scsh, const int bid) {
  const int cl = threadIdx.x & (CH - 1), grp = threadIdx.x / CH;
  const int c = bid * CH + cl;
  const bool ok = c < p.N;
  if (p.is_training) {
    // parameters first: their latency hides under the tile reduction instead of following it
    const int cc = ok ? c : 0;
    const float gam = p.gamma[cc], bet = p.beta[cc], mm = p.moving_mean[cc], mv = p.moving_var[cc], dec = p.decay[0];
    const float* const src[2] = {p.psum, p.psumsq};
    double acc[2];
    tile_sums<2, GR, (GR > 64 ? 8 : 16)>(src, p.n_tiles, p.N, c, grp, ok, acc);
    const double s = group_reduce<GR, CH>(acc[0], red, grp, cl);
    const double ss = group_reduce<GR, CH>(acc[1], red, grp, cl);
    if (grp == 0 && ok) {
      const double n = (double)p.count;
      const double mean = s / n;
      double var = ss / n - mean * mean;
      if (var < 0.0) var = 0.0;
      const double invstd = 1.0 / sqrt(var + (double)p.eps);
      const double sc = (double)gam * invstd;
      p.scale[c] = (float)sc;
      p.shift[c] = (float)((double)bet - mean * sc);
      p.mean[c] = (float)mean;
      p.invstd[c] = (float)invstd;
      const double d = (double)dec;
      const double var_ema = p.unbiased_ema ? var * (n / (n > 1.0 ? n - 1.0 : 1.0)) : var;
      p.moving_mean[c] = (float)((double)mm * d + mean * (1.0 - d));
      p.moving_var[c] = (float)((double)mv * d + var_ema * (1.0 - d));
    }
  } else if (grp == 0 && ok) {
    const double invstd = 1.0 / sqrt((double)p.moving_var[c] + (double)p.eps);
    const double sc = (double)p.gamma[c] * invstd;
    p.scale[c] = (float)sc;
    p.shift[c] = (float)((double)p.beta[c] - (double)p.moving_mean[c] * sc);
    p.mean[c] = p.moving_mean[c];
    p.invstd[c] = (float)invstd;
  }
  // optional K3: the max-pool pick of the same 16 channels (scale/shift handed over through LDS, not through memory)
  if (p.pool_pmax != nullptr) {
    float* s_sc = s_scsh;
    float* s_sh = s_scsh + CH;
    __syncthreads();
    if (grp == 0 && ok) { s_sc[cl] = p.scale[c]; s_sh[cl] = p.shift[c]; }     // this thread's own stores: visible to itself
    __syncthreads();
    if (ok) {
      const float sc = s_sc[cl], sh = s_sh[cl];
      for (int b = grp; b < p.pool_B; b += GR) pool_pick(p, b, c, sc, sh);
    }
  }
}

__device__ __forceinline__ void dy_colsum_body(const t3d_dy_colsum_args& p, const int bid, const int tid) {
  const int i = bid * 256 + tid;
  if (tid >= 256 || i >= p.B * p.N) return;
  const int b = i / p.N, c = i % p.N;
  double sdz = 0.0, sy = 0.0;
  for (int t = 0; t < p.tiles_per_frustum; ++t) {
    const size_t o = (size_t)(b * p.tiles_per_frustum + t) * p.N + c;
    sdz += (double)p.psum_dz[o];
    sy += (double)p.psum_y[o];
  }
  const double v = (double)p.coef[c] * sdz + (double)p.coef[p.N + c] * sy +
                   (double)p.coef[2 * p.N + c] * (double)p.rows_per_frustum;
  p.out[i] = (float)((double)p.alpha * v);
}

// every slab region starts 16-byte aligned and numel % 4 == 0 for the engine's allocations (float4 path); anything
// else takes the scalar path.  A block = 32 float4 elements x 8 slab groups: thread (e, g) sums slabs g, g+8, ... with
// four loads in flight, the 8 group sums are combined through LDS in a fixed order.  Splitting the slab chain over
// threads is what keeps the many-slab / few-element tensors (64x64 layers with 256 slabs) from being one long latency
// chain while the rest of the chip idles.
__device__ __forceinline__ void reduce_slabs_body(const float* __restrict__ slab_base, float* __restrict__ grad_base,
                                                  const t3d_slab_desc* __restrict__ table, float4 (*part)[32], int bx, int by,
                                                  int gx) {
  const t3d_slab_desc d = table[by];
  const bool vec = ((d.slab_off | d.grad_off | (int64_t)d.numel) & 3) == 0;
  const int n4 = vec ? (d.numel >> 2) : 0;
  const int el = threadIdx.x & 31, grp = threadIdx.x >> 5;
  for (int e0 = bx * 32; e0 < n4; e0 += gx * 32) {
    const int e = e0 + el;
    float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0, a2 = a0, a3 = a0;
    if (e < n4) {
      const float4* s = reinterpret_cast<const float4*>(slab_base + d.slab_off) + e;
      int k = grp;
      for (; k + 56 < d.n_slabs; k += 64) {            // eight slabs in flight per thread
        float4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = s[(size_t)(k + 8 * u) * n4];
#pragma unroll
        for (int u = 0; u < 8; u += 4) {
          a0.x += v[u].x; a0.y += v[u].y; a0.z += v[u].z; a0.w += v[u].w;
          a1.x += v[u + 1].x; a1.y += v[u + 1].y; a1.z += v[u + 1].z; a1.w += v[u + 1].w;
          a2.x += v[u + 2].x; a2.y += v[u + 2].y; a2.z += v[u + 2].z; a2.w += v[u + 2].w;
          a3.x += v[u + 3].x; a3.y += v[u + 3].y; a3.z += v[u + 3].z; a3.w += v[u + 3].w;
        }
      }
      for (; k + 24 < d.n_slabs; k += 32) {
        const float4 v0 = s[(size_t)(k + 0) * n4], v1 = s[(size_t)(k + 8) * n4], v2 = s[(size_t)(k + 16) * n4],
                     v3 = s[(size_t)(k + 24) * n4];
        a0.x += v0.x; a0.y += v0.y; a0.z += v0.z; a0.w += v0.w;
        a1.x += v1.x; a1.y += v1.y; a1.z += v1.z; a1.w += v1.w;
        a2.x += v2.x; a2.y += v2.y; a2.z += v2.z; a2.w += v2.w;
        a3.x += v3.x; a3.y += v3.y; a3.z += v3.z; a3.w += v3.w;
      }
      for (; k < d.n_slabs; k += 8) {
        const float4 v = s[(size_t)k * n4];
        a0.x += v.x; a0.y += v.y; a0.z += v.z; a0.w += v.w;
      }
    }
    float4 r;
    r.x = (a0.x + a1.x) + (a2.x + a3.x); r.y = (a0.y + a1.y) + (a2.y + a3.y);
    r.z = (a0.z + a1.z) + (a2.z + a3.z); r.w = (a0.w + a1.w) + (a2.w + a3.w);
    part[grp][el] = r;
    __syncthreads();
    if (grp == 0 && e < n4) {
      float4 t = part[0][el];
#pragma unroll
      for (int g = 1; g < 8; ++g) {
        const float4 u = part[g][el];
        t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w;
      }
      reinterpret_cast<float4*>(grad_base + d.grad_off)[e] = t;
    }
    __syncthreads();
  }
  if (!vec) {
    for (int e = bx * blockDim.x + threadIdx.x; e < d.numel; e += gx * blockDim.x) {
      float acc = 0.f;
      for (int k = 0; k < d.n_slabs; ++k) acc += slab_base[d.slab_off + (size_t)k * d.numel + e];
      grad_base[d.grad_off + e] = acc;
    }
  }
}


}  // namespace
