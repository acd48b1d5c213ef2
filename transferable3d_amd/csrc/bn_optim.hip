// Batch-norm finalizers, max-pool finalize, per-frustum dy column sums, slab reduction, schedules,
// TF-form Adam and the dropout mask generator.  All are short bandwidth/latency-bound kernels that
// sit between the MFMA GEMMs of pointmlp.hip; reductions combine tile partials in a fixed order
// (double accumulators), so results are run-to-run reproducible.
#include "common.h"
#include "poolbwd_dev.h"
#include "bn_dev.h"

namespace {

template <int GR, int CH = FC_CH>
__global__ __launch_bounds__(GR * CH) void k_bn_fwd_finalize(const t3d_bn_fwd_finalize_args p) {
  __shared__ double red[GR][CH];
  __shared__ float s_scsh[2 * CH];
  bn_fwd_finalize_body<GR, CH>(p, red, s_scsh, blockIdx.x);
}

// one thread per (frustum, channel)
__global__ __launch_bounds__(256) void k_pool_finalize(const t3d_pool_finalize_args p) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= p.B * p.N) return;
  const int b = i / p.N, c = i % p.N;
  pool_pick_impl(p.pmax, p.pmin, p.pamax, p.pamin, p.tiles_per_frustum, p.N, b, c, p.scale[c], p.shift[c], p.pooled, p.ld_pooled,
                 p.argidx, p.ysel);
}

template <int GR, int CH = FC_CH>
__global__ __launch_bounds__(GR * CH) void k_bn_bwd_finalize(const t3d_bn_bwd_finalize_args p) {
  __shared__ double red[GR][CH];
  bn_bwd_finalize_body<GR, CH>(p, red, blockIdx.x, threadIdx.x);
}

__global__ __launch_bounds__(256) void k_dy_colsum(const t3d_dy_colsum_args p) { dy_colsum_body(p, blockIdx.x, threadIdx.x); }

__global__ __launch_bounds__(256) void k_reduce_slabs(const float* __restrict__ slab_base, float* __restrict__ grad_base,
                                                      const t3d_slab_desc* __restrict__ table) {
  __shared__ float4 part[8][32];
  reduce_slabs_body(slab_base, grad_base, table, part, blockIdx.x, blockIdx.y, gridDim.x);
}

// slab reduction and the sparse arg-max rows of a pooled layer's backward in one launch: blocks [0, n_reduce) reduce
// (logical grid gx x n_tensors), the rest run the sparse-row tiles
__global__ __launch_bounds__(256) void k_pool_bwd_mid(const float* __restrict__ slab_base, float* __restrict__ grad_base,
                                                      const t3d_slab_desc* __restrict__ table, const int gx, const int n_reduce,
                                                      const t3d_pool_sparse_rows_args sp) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int b = blockIdx.x;
  if (b < n_reduce) {
    reduce_slabs_body(slab_base, grad_base, table, reinterpret_cast<float4(*)[32]>(smem), b % gx, b / gx, gx);
  } else {
    const int r = b - n_reduce, tiles = sp.B * sp.rows_per_frustum / 128;
    pool_sparse_rows_body(sp, smem, r % tiles, r / tiles);
  }
}

// One wave.  The four pow() of the schedule are independent dependent-chains of a few hundred instructions each: lanes 0-3 take one
// each (same function, same arguments as the one-thread form: bit-identical), lane 0 combines them.
__global__ __launch_bounds__(64) void k_schedule_step(float* hyper, const t3d_schedule s) {
  if (blockIdx.x != 0) return;
  const int lane = threadIdx.x;
  // global step BEFORE this update drives lr / bn_decay (tf: minimize() increments after use)
  const double step = (double)hyper[0] + (double)s.step_offset;
  const double seen = step * (double)s.batch_size;
  const double t = step + 1.0;   // Adam's t starts at 1
  const double base = lane == 0 ? (double)s.lr_decay_rate : lane == 1 ? (double)s.bn_decay_rate : lane == 2 ? (double)s.beta2 : (double)s.beta1;
  const double expo = lane == 0 ? floor(seen / (double)s.lr_decay_step) : lane == 1 ? floor(seen / (double)s.bn_decay_step) : t;
  const double pw = lane < 4 ? pow(base, expo) : 0.0;
  const double p_lr = __shfl(pw, 0), p_bn = __shfl(pw, 1), p_b2 = __shfl(pw, 2), p_b1 = __shfl(pw, 3);
  if (lane != 0) return;
  const double lr = (double)s.base_lr * p_lr;
  const double bnm = (double)s.bn_init_decay * p_bn;
  const double bnd = fmin((double)s.bn_decay_clip, 1.0 - bnm);
  const double lr_t = lr * sqrt(1.0 - p_b2) / (1.0 - p_b1);
  hyper[1] = (float)lr;
  hyper[2] = (float)bnd;
  hyper[3] = (float)lr_t;
  hyper[0] = (float)t;
}

__global__ __launch_bounds__(256) void k_adam_tf(float* __restrict__ w, const float* __restrict__ g, float* __restrict__ m,
                                                 float* __restrict__ v, int64_t n, const float* __restrict__ hyper,
                                                 float b1, float b2, float eps, float gscale) {
  const float lr_t = hyper[3];
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float gi = g[i] * gscale;
    const float mi = b1 * m[i] + (1.f - b1) * gi;
    const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    w[i] -= lr_t * mi / (sqrtf(vi) + eps);
  }
}

// tf.train.MomentumOptimizer (train_semisup.py:226-228; use_nesterov = False): accum = momentum * accum + g; w -= lr * accum
__global__ __launch_bounds__(256) void k_momentum_tf(float* __restrict__ w, const float* __restrict__ g, float* __restrict__ acc,
                                                     int64_t n, const float* __restrict__ hyper, float momentum, float gscale) {
  const float lr = hyper[1];
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float a = fmaf(momentum, acc[i], g[i] * gscale);
    acc[i] = a;
    w[i] -= lr * a;
  }
}

// counter-based generator: 2 rounds of a 64-bit mix over (seed, step, index)
__device__ __forceinline__ uint32_t mix_u32(uint64_t x) {
  x ^= x >> 33; x *= 0xff51afd7ed558ccdULL;
  x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL;
  x ^= x >> 33;
  return (uint32_t)(x >> 16);
}
__global__ __launch_bounds__(256) void k_dropout_mask(float* __restrict__ mask, int64_t n, float keep, uint32_t seed,
                                                      const float* __restrict__ hyper) {
  const uint64_t step = (uint64_t)hyper[0];
  const uint64_t key = ((uint64_t)seed << 32) ^ (step * 0x9E3779B97F4A7C15ULL);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const uint32_t r = mix_u32(key + (uint64_t)i * 0xD6E8FEB86659FD93ULL);
    const float u = (float)(r >> 8) * (1.0f / 16777216.0f);
    mask[i] = u < keep ? 1.f : 0.f;
  }
}

}  // namespace

extern "C" int t3d_abi_version(void) { return T3D_ABI_VERSION; }

static int fin_big_tiles() {      // T3D_FIN_BIG: tile count above which the 64-group finalizers run (0 = never)
  static const int v = []() { const char* e = getenv("T3D_FIN_BIG"); const int x = e ? atoi(e) : 512; return x > 0 ? x : (1 << 30); }();
  return v;
}

static int fin_wide_tiles() {      // T3D_FIN_WIDE: tile count from which the 4-channel x 256-group finalizers run (0 = never)
  static const int v = []() { const char* e = getenv("T3D_FIN_WIDE"); const int x = e ? atoi(e) : 1024; return x > 0 ? x : (1 << 30); }();
  return v;
}

extern "C" int t3d_bn_fwd_finalize(const t3d_bn_fwd_finalize_args* a, t3d_stream_t stream) {
  if (!a || !a->gamma || !a->beta || !a->moving_mean || !a->moving_var || !a->scale || !a->shift || !a->mean ||
      !a->invstd)
    return T3D_ERR_ARG;
  if (a->is_training && (!a->psum || !a->psumsq || !a->decay || a->count <= 0)) return T3D_ERR_ARG;
  if (a->is_training && a->n_tiles >= fin_wide_tiles() && a->N <= FC_WIDE_MAX_N)
    T3D_LAUNCH((k_bn_fwd_finalize<FC_GR_WIDE, FC_CH_WIDE>), dim3((a->N + FC_CH_WIDE - 1) / FC_CH_WIDE), dim3(FC_GR_WIDE * FC_CH_WIDE), 0,
               static_cast<hipStream_t>(stream), *a);
  else if (a->is_training && a->n_tiles > fin_big_tiles())
    T3D_LAUNCH(k_bn_fwd_finalize<FC_GR_BIG>, dim3((a->N + FC_CH - 1) / FC_CH), dim3(FC_GR_BIG * FC_CH), 0, static_cast<hipStream_t>(stream), *a);
  else
    T3D_LAUNCH(k_bn_fwd_finalize<FC_GR>, dim3((a->N + FC_CH - 1) / FC_CH), dim3(FC_GR * FC_CH), 0, static_cast<hipStream_t>(stream), *a);
  T3D_CHECK_LAUNCH();
  return T3D_OK;
}

extern "C" int t3d_pool_finalize(const t3d_pool_finalize_args* a, t3d_stream_t stream) {
  if (!a || !a->scale || !a->shift || !a->pmax || !a->pmin || !a->pamax || !a->pamin || !a->pooled || !a->argidx ||
      !a->ysel)
    return T3D_ERR_ARG;
  T3D_LAUNCH(k_pool_finalize, dim3((a->B * a->N + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream),
                     *a);
  T3D_CHECK_LAUNCH();
  return T3D_OK;
}

extern "C" int t3d_bn_bwd_finalize(const t3d_bn_bwd_finalize_args* a, t3d_stream_t stream) {
  if (!a || !a->coef) return T3D_ERR_ARG;
  if (a->psum_dz == nullptr && (!a->dpool_in || !a->pooled || !a->ysel || !a->dpool)) return T3D_ERR_ARG;
  if (a->psum_dz != nullptr && !a->psum_dzy) return T3D_ERR_ARG;
  if (a->frozen ? !a->scale : (!a->gamma || !a->mean || !a->invstd)) return T3D_ERR_ARG;
  if (a->psum_dz != nullptr && a->n_tiles >= fin_wide_tiles() && a->N <= FC_WIDE_MAX_N)
    T3D_LAUNCH((k_bn_bwd_finalize<FC_GR_WIDE, FC_CH_WIDE>), dim3((a->N + FC_CH_WIDE - 1) / FC_CH_WIDE), dim3(FC_GR_WIDE * FC_CH_WIDE), 0,
               static_cast<hipStream_t>(stream), *a);
  else if (a->psum_dz != nullptr && a->n_tiles > fin_big_tiles())
    T3D_LAUNCH(k_bn_bwd_finalize<FC_GR_BIG>, dim3((a->N + FC_CH - 1) / FC_CH), dim3(FC_GR_BIG * FC_CH), 0, static_cast<hipStream_t>(stream), *a);
  else
    T3D_LAUNCH(k_bn_bwd_finalize<FC_GR>, dim3((a->N + FC_CH - 1) / FC_CH), dim3(FC_GR * FC_CH), 0, static_cast<hipStream_t>(stream), *a);
  T3D_CHECK_LAUNCH();
  return T3D_OK;
}

extern "C" int t3d_dy_colsum(const t3d_dy_colsum_args* a, t3d_stream_t stream) {
  if (!a || !a->psum_dz || !a->psum_y || !a->coef || !a->out) return T3D_ERR_ARG;
  T3D_LAUNCH(k_dy_colsum, dim3((a->B * a->N + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream), *a);
  T3D_CHECK_LAUNCH();
  return T3D_OK;
}

extern "C" int t3d_reduce_slabs(const float* slab_base, float* grad_base, const t3d_slab_desc* table_dev, int n_tensors,
                                int max_numel, t3d_stream_t stream) {
  if (!slab_base || !grad_base || !table_dev || n_tensors <= 0) return T3D_ERR_ARG;
  int gx = (max_numel / 4 + 31) / 32;      // one block per 32 float4 elements of the largest tensor
  if (gx > 256) gx = 256;
  if (gx < 1) gx = 1;
  T3D_LAUNCH(k_reduce_slabs, dim3(gx, n_tensors), dim3(256), 0, static_cast<hipStream_t>(stream), slab_base,
                     grad_base, table_dev);
  T3D_CHECK_LAUNCH();
  return T3D_OK;
}

extern "C" int t3d_schedule_step(float* hyper, const t3d_schedule* s, t3d_stream_t stream) {
  if (!hyper || !s) return T3D_ERR_ARG;
  T3D_LAUNCH(k_schedule_step, dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream), hyper, *s);
  T3D_CHECK_LAUNCH();
  return T3D_OK;
}

extern "C" int t3d_adam_tf_step(float* params, const float* grads, float* m, float* v, int64_t n, const float* hyper,
                                float beta1, float beta2, float eps, float grad_scale, t3d_stream_t stream) {
  if (!params || !grads || !m || !v || !hyper || n <= 0) return T3D_ERR_ARG;
  int64_t blocks = (n + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  T3D_LAUNCH(k_adam_tf, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), params, grads, m,
                     v, n, hyper, beta1, beta2, eps, grad_scale);
  T3D_CHECK_LAUNCH();
  return T3D_OK;
}

extern "C" int t3d_momentum_step(float* params, const float* grads, float* accum, int64_t n, const float* hyper, float momentum,
                                 float grad_scale, t3d_stream_t stream) {
  if (!params || !grads || !accum || !hyper || n <= 0) return T3D_ERR_ARG;
  int64_t blocks = (n + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  T3D_LAUNCH(k_momentum_tf, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), params, grads, accum, n, hyper,
             momentum, grad_scale);
  T3D_CHECK_LAUNCH();
  return T3D_OK;
}

// bf16 copy of the weights for the T3D_BF16 GEMMs (8 elements per thread: one 32-byte read, one 16-byte write)
__global__ __launch_bounds__(256) void k_cast_bf16(const float* __restrict__ src, bf16_t* __restrict__ dst, int64_t n) {
  const int64_t stride = (int64_t)gridDim.x * 256 * 8;
  for (int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 8; i < n; i += stride) {
    if (i + 8 <= n) {
      const float4 a = *reinterpret_cast<const float4*>(src + i), b = *reinterpret_cast<const float4*>(src + i + 4);
      const bf16x8 h = {(bf16_t)a.x, (bf16_t)a.y, (bf16_t)a.z, (bf16_t)a.w, (bf16_t)b.x, (bf16_t)b.y, (bf16_t)b.z, (bf16_t)b.w};
      *reinterpret_cast<bf16x8*>(dst + i) = h;
    } else {
      for (int64_t j = i; j < n; ++j) dst[j] = (bf16_t)src[j];
    }
  }
}

extern "C" int t3d_cast_bf16(const float* src, void* dst, int64_t n, t3d_stream_t stream) {
  if (!src || !dst || n <= 0 || (reinterpret_cast<uintptr_t>(src) & 15) || (reinterpret_cast<uintptr_t>(dst) & 15)) return T3D_ERR_ARG;
  int64_t blocks = (n + 2047) / 2048;
  if (blocks > 2048) blocks = 2048;
  T3D_LAUNCH(k_cast_bf16, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), src, static_cast<bf16_t*>(dst), n);
  T3D_CHECK_LAUNCH();
  return T3D_OK;
}

extern "C" int t3d_dropout_mask(float* mask, int64_t n, float keep_prob, uint32_t seed, const float* hyper,
                                t3d_stream_t stream) {
  if (!mask || !hyper || n <= 0) return T3D_ERR_ARG;
  int64_t blocks = (n + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  T3D_LAUNCH(k_dropout_mask, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), mask, n,
                     keep_prob, seed, hyper);
  T3D_CHECK_LAUNCH();
  return T3D_OK;
}

extern "C" int t3d_pool_bwd_mid(const float* slab_base, float* grad_base, const t3d_slab_desc* table_dev, int n_tensors,
                                int max_numel, const t3d_pool_sparse_rows_args* a, t3d_stream_t stream) {
  if (!slab_base || !grad_base || !table_dev || n_tensors <= 0) return T3D_ERR_ARG;
  const int rc = check_sparse_rows(a);
  if (rc != T3D_OK) return rc;
  int gx = (max_numel / 4 + 31) / 32;
  if (gx > 256) gx = 256;
  if (gx < 1) gx = 1;
  const int n_reduce = gx * n_tensors;
  const long M = (long)a->B * a->rows_per_frustum;
  const int n_sparse = (int)(M / 128) * (a->K / SR_KC);
  const size_t lds = sparse_rows_lds(a->N);
  static size_t allowed = 0;
  if (lds > 64 * 1024 && lds > allowed) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_pool_bwd_mid), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    allowed = lds;
  }
  T3D_LAUNCH(k_pool_bwd_mid, dim3(n_reduce + n_sparse), dim3(256), lds, static_cast<hipStream_t>(stream), slab_base, grad_base,
             table_dev, gx, n_reduce, *a);
  T3D_CHECK_LAUNCH();
  return T3D_OK;
}
