// Device bodies of the batch-norm-backward finalizer and the per-frustum dy column sums (bn_optim.hip) -- a header so that the
// paired small-launch kernel (pair.hip) can run them beside the FC kernels in one launch.
#pragma once
#include "common.h"

namespace {

// 256 threads = 16 channels x 16 tile groups: short, unrolled, independent loads (the 64-iteration serial
// loop of the first version was pure L2 latency: 19 us per launch in profiles/r01_baseline).
// At the row counts of config 4 (2048 row tiles) a 16-group block walks 128 tiles per thread -- eight dependent memory round trips
// (13.7 us per launch, profiles/r02_bf16_v2): above T3D_FIN_BIG tiles (default 512) the launchers take the GR = 64 instantiation
// (1024 threads, two round trips; only group 0 walks the 64 LDS partials): fwd finalize 195 -> 163 us per step, bwd 124 -> 112 at
// B=128 N=2048 (same-box A/B).  Up to 512 tiles GR stays 16, so the results at the sizes of configs 1-3 do not change by a bit.
constexpr int FC_CH = 16, FC_GR = 16, FC_GR_BIG = 64;

// Sixteen tiles per quantity are in flight per thread (32 loads for the two-quantity reductions): at 256 tiles the whole
// reduction is ONE memory round trip instead of four.
template <int NQ, int GR = FC_GR>
__device__ __forceinline__ void tile_sums(const float* const (&src)[NQ], int n_tiles, int N, int c, int grp, bool ok,
                                          double (&acc)[NQ]) {
  constexpr int U = 16;
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

template <int GR = FC_GR>
__device__ __forceinline__ double group_reduce(double v, double (*red)[FC_CH], int grp, int cl, bool act = true) {
  __syncthreads();
  if (act) red[grp][cl] = v;
  __syncthreads();
  double s = 0.0;
  if (GR == FC_GR || grp == 0) {      // only group 0 uses the sum; with 64 groups the other 1008 threads' reads are pure LDS traffic
#pragma unroll
    for (int g = 0; g < GR; ++g) s += red[g][cl];
  }
  return s;
}

// `tid` may exceed GR * FC_CH (the 512-thread paired launch, pair.hip): the extra threads take part in the barriers only
template <int GR>
__device__ __forceinline__ void bn_bwd_finalize_body(const t3d_bn_bwd_finalize_args& p, const int bid, const int tid) {
  __shared__ double red[GR][FC_CH];
  const bool act = tid < GR * FC_CH;
  const int cl = tid & (FC_CH - 1), grp = act ? tid / FC_CH : 0;
  const int c = bid * FC_CH + cl;
  const bool ok = act && c < p.N;
  const int cc = ok ? c : 0;
  const float mean_f = p.mean[cc], invstd_f = p.invstd[cc], gamma_f = p.gamma[cc];   // ahead of the reduction
  double acc[2] = {0.0, 0.0};   // sum dz, sum dz*y
  if (p.psum_dz != nullptr) {
    const float* const src[2] = {p.psum_dz, p.psum_dzy};
    tile_sums<2, GR>(src, p.n_tiles, p.N, c, grp, ok, acc);
  } else if (ok) {
    for (int b = grp; b < p.B; b += GR) {
      const float live = p.pooled[(size_t)b * p.ld_pooled + c] > 0.f ? 1.f : 0.f;
      const float g = p.dpool_in[(size_t)b * p.ld_dpool_in + c] * live;
      p.dpool[(size_t)b * p.N + c] = g;
      acc[0] += (double)g;
      acc[1] += (double)g * (double)p.ysel[(size_t)b * p.N + c];
    }
  }
  const double s1 = group_reduce<GR>(acc[0], red, grp, cl, act);
  const double s2 = group_reduce<GR>(acc[1], red, grp, cl, act);
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

}  // namespace
