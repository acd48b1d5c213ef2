// Per-frustum fully-connected layers (B rows): forward, backward, input gradient.
//
// Replaces tf_util.fully_connected (models/tf_util.py:1463-1499) with its batch-norm over the B rows
// (tf_util.py:1666-1677), activation and the tf_util.dropout that follows it (tf_util.py:1720-1741) at
// semisup_models.py:196-198, 253-261, 385-392 and mlps_with_dropout (44-63).  These layers are latency
// bound (M = B = 32..128 rows, weights <= 1 MB): one workgroup owns 32 output columns for ALL rows, so the
// batch-norm reductions over the batch stay inside the workgroup and a layer is a single launch.
#include "common.h"

namespace {

constexpr int KC = 64;       // reduction chunk
constexpr int CB = 32;       // columns per workgroup
constexpr int MAXRB = 4;     // B <= 128

struct RowSrc {              // [in | in2] row-concatenated input
  const float* in; int ld_in; int K;
  const float* in2; int ld_in2; int K2;
  __device__ __forceinline__ float at(int r, int k) const {
    if (k < K) return in[(size_t)r * ld_in + k];
    if (k < K + K2) return in2[(size_t)r * ld_in2 + (k - K)];
    return 0.f;
  }
};

// acc[rb][j] += sum_k src(row, k) * W(k, col) for rows rb*32 + rg*4 + j.
// WT == false: W(k,col) = w[k*ldw + c0+col]; WT == true: W(k,col) = w[(c0+col)*ldw + k].
template <bool WT>
__device__ __forceinline__ void rows_gemm(float (&acc)[MAXRB][4], const RowSrc& src, int B, int RB, int BP, const float* w,
                                          int ldw, int c0, int ncols_valid, float* in_s, float* w_s) {
  const int tid = threadIdx.x, col = tid & 31, rg = tid >> 5;
  const int Kt = src.K + src.K2;
#pragma unroll
  for (int rb = 0; rb < MAXRB; ++rb)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[rb][j] = 0.f;
  for (int k0 = 0; k0 < Kt; k0 += KC) {
    for (int idx = tid; idx < RB * 32 * KC; idx += 256) {
      const int r = idx / KC, kk = idx % KC;
      in_s[kk * BP + r] = (r < B) ? src.at(r, k0 + kk) : 0.f;
    }
    for (int idx = tid; idx < KC * CB; idx += 256) {
      float v = 0.f;
      if (!WT) {
        const int kk = idx / CB, c = idx % CB;
        if (k0 + kk < Kt && c < ncols_valid) v = w[(size_t)(k0 + kk) * ldw + c0 + c];
        w_s[kk * CB + c] = v;
      } else {
        const int c = idx / KC, kk = idx % KC;
        if (k0 + kk < Kt && c < ncols_valid) v = w[(size_t)(c0 + c) * ldw + k0 + kk];
        w_s[kk * CB + c] = v;
      }
    }
    __syncthreads();
#pragma unroll 8
    for (int kk = 0; kk < KC; ++kk) {
      const float wv = w_s[kk * CB + col];
#pragma unroll
      for (int rb = 0; rb < MAXRB; ++rb) {
        if (rb < RB) {
          const float4 x = *reinterpret_cast<const float4*>(in_s + kk * BP + rb * 32 + rg * 4);
          acc[rb][0] = fmaf(x.x, wv, acc[rb][0]);
          acc[rb][1] = fmaf(x.y, wv, acc[rb][1]);
          acc[rb][2] = fmaf(x.z, wv, acc[rb][2]);
          acc[rb][3] = fmaf(x.w, wv, acc[rb][3]);
        }
      }
    }
    __syncthreads();
  }
}

// sum over all rows of the workgroup's per-thread partial, per column; result broadcast to every thread
__device__ __forceinline__ float col_reduce(float part, float* red) {
  const int tid = threadIdx.x, col = tid & 31, rg = tid >> 5;
  __syncthreads();
  red[rg * CB + col] = part;
  __syncthreads();
  float s = 0.f;
#pragma unroll
  for (int g = 0; g < 8; ++g) s += red[g * CB + col];
  return s;
}

__device__ __forceinline__ float act_fwd(float z, int act, float alpha) {
  switch (act) {
    case T3D_ACT_RELU: return fmaxf(z, 0.f);
    case T3D_ACT_LEAKY_RELU: return z > 0.f ? z : alpha * z;
    case T3D_ACT_TANH: return tanhf(z);
    default: return z;
  }
}
__device__ __forceinline__ float act_bwd(float z, int act, float alpha) {
  switch (act) {
    case T3D_ACT_RELU: return z > 0.f ? 1.f : 0.f;
    case T3D_ACT_LEAKY_RELU: return z > 0.f ? 1.f : alpha;
    case T3D_ACT_TANH: { const float t = tanhf(z); return 1.f - t * t; }
    default: return 1.f;
  }
}

__global__ __launch_bounds__(256) void k_fc_fwd(const t3d_fc_fwd_args p) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int RB = (p.B + 31) / 32, BP = RB * 32 + 4;
  float* in_s = sm;
  float* w_s = in_s + KC * BP;
  float* red = w_s + KC * CB;
  const int tid = threadIdx.x, col = tid & 31, rg = tid >> 5;
  const int c0 = blockIdx.x * CB, c = c0 + col;
  const int nvalid = min(CB, p.N - c0);
  const bool cok = c < p.N;

  float acc[MAXRB][4];
  RowSrc src{p.in, p.ld_in, p.K, p.in2, p.ld_in2, p.K2};
  rows_gemm<false>(acc, src, p.B, RB, BP, p.w, p.N, c0, nvalid, in_s, w_s);

  const float bias = (cok && p.bias) ? p.bias[c] : 0.f;
  float part = 0.f;
#pragma unroll
  for (int rb = 0; rb < MAXRB; ++rb)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      acc[rb][j] += bias;
      const int r = rb * 32 + rg * 4 + j;
      if (rb < RB && r < p.B) part += acc[rb][j];
    }
  const bool bn = p.gamma != nullptr;
  float mean = 0.f, invstd = 1.f, g = 1.f, be = 0.f;
  if (bn) {
    if (cok) { g = p.gamma[c]; be = p.beta[c]; }
    if (p.is_training) {
      mean = col_reduce(part, red) / (float)p.B;
      float vpart = 0.f;
#pragma unroll
      for (int rb = 0; rb < MAXRB; ++rb)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int r = rb * 32 + rg * 4 + j;
          if (rb < RB && r < p.B) { const float d = acc[rb][j] - mean; vpart = fmaf(d, d, vpart); }
        }
      const float var = col_reduce(vpart, red) / (float)p.B;
      invstd = 1.0f / sqrtf(var + p.eps);
      if (cok && rg == 0) {
        const float d = p.decay[0];
        const float var_ema = p.unbiased_ema ? var * ((float)p.B / (float)max(p.B - 1, 1)) : var;
        p.moving_mean[c] = p.moving_mean[c] * d + mean * (1.f - d);
        p.moving_var[c] = p.moving_var[c] * d + var_ema * (1.f - d);
      }
    } else if (cok) {
      mean = p.moving_mean[c];
      invstd = 1.0f / sqrtf(p.moving_var[c] + p.eps);
    }
    if (cok && rg == 0) { p.mean[c] = mean; p.invstd[c] = invstd; }
  }
  if (!cok) return;
  const float inv_keep = p.drop_mask ? 1.0f / p.keep_prob : 1.f;
#pragma unroll
  for (int rb = 0; rb < MAXRB; ++rb)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int r = rb * 32 + rg * 4 + j;
      if (rb < RB && r < p.B) {
        const float y = acc[rb][j];
        if (p.y) p.y[(size_t)r * p.N + c] = y;
        float z = bn ? (y - mean) * invstd * g + be : y;
        z = act_fwd(z, p.act, p.leaky_alpha);
        if (p.drop_mask) z *= p.drop_mask[(size_t)r * p.N + c] * inv_keep;
        if (p.add_in && c < p.add_n) z += p.add_in[(size_t)r * p.ld_add + c];
        p.out[(size_t)r * p.ld_out + c] = z;
      }
    }
}

__global__ __launch_bounds__(256) void k_fc_bwd(const t3d_fc_bwd_args p) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int RB = (p.B + 31) / 32, BP = RB * 32 + 4;
  float* in_s = sm;
  float* w_s = in_s + KC * BP;
  float* red = w_s + KC * CB;
  float* dy_s = red + 8 * CB;            // [RB*32][CB]
  const int tid = threadIdx.x, col = tid & 31, rg = tid >> 5;
  const int c0 = blockIdx.x * CB, c = c0 + col;
  const int nvalid = min(CB, p.N - c0);
  const bool cok = c < p.N;

  // (a) gradient w.r.t. this layer's output
  float gout[MAXRB][4];
  if (p.dout != nullptr) {
#pragma unroll
    for (int rb = 0; rb < MAXRB; ++rb)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int r = rb * 32 + rg * 4 + j;
        gout[rb][j] = (rb < RB && r < p.B && cok) ? p.dout[(size_t)r * p.ld_dout + c] : 0.f;
      }
  } else {
    RowSrc src{p.dy_next, p.N_next, p.N_next, nullptr, 0, 0};
    rows_gemm<true>(gout, src, p.B, RB, BP, p.w_next, p.N_next, c0, nvalid, in_s, w_s);
  }

  // (b) dropout / activation backward, (c) batch-norm backward over the B rows
  const bool bn = p.gamma != nullptr;
  float mean = 0.f, invstd = 1.f, g = 1.f, be = 0.f;
  if (bn && cok) { mean = p.mean[c]; invstd = p.invstd[c]; g = p.gamma[c]; be = p.beta[c]; }
  const float inv_keep = p.drop_mask ? 1.0f / p.keep_prob : 1.f;
  float xh[MAXRB][4];
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int rb = 0; rb < MAXRB; ++rb)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int r = rb * 32 + rg * 4 + j;
      float dz = 0.f, x = 0.f;
      if (rb < RB && r < p.B && cok) {
        const float y = p.y ? p.y[(size_t)r * p.N + c] : 0.f;
        x = bn ? (y - mean) * invstd : y;
        const float z = bn ? x * g + be : y;
        dz = gout[rb][j];
        if (p.drop_mask) dz *= p.drop_mask[(size_t)r * p.N + c] * inv_keep;
        dz *= act_bwd(z, p.act, p.leaky_alpha);
        s1 += dz;
        s2 = fmaf(dz, x, s2);
      }
      gout[rb][j] = dz;
      xh[rb][j] = x;
    }
  float dbias = 0.f;
  if (bn && p.bn_training) {
    const float dbeta = col_reduce(s1, red);
    const float dgamma = col_reduce(s2, red);
    if (cok && rg == 0) {
      if (p.dbeta) p.dbeta[c] = dbeta;
      if (p.dgamma) p.dgamma[c] = dgamma;
    }
    const float invB = 1.0f / (float)p.B, c1 = g * invstd;
#pragma unroll
    for (int rb = 0; rb < MAXRB; ++rb)
#pragma unroll
      for (int j = 0; j < 4; ++j) gout[rb][j] = c1 * (gout[rb][j] - dbeta * invB - xh[rb][j] * dgamma * invB);
  } else if (bn) {
    const float c1 = g * invstd;
#pragma unroll
    for (int rb = 0; rb < MAXRB; ++rb)
#pragma unroll
      for (int j = 0; j < 4; ++j) gout[rb][j] *= c1;
  } else {
    dbias = col_reduce(s1, red);
  }
  if (cok && rg == 0 && p.dbias) p.dbias[c] = dbias;   // exactly 0 under training-mode BN

  __syncthreads();
#pragma unroll
  for (int rb = 0; rb < MAXRB; ++rb)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int r = rb * 32 + rg * 4 + j;
      if (rb < RB) {
        const float v = (r < p.B && cok) ? gout[rb][j] : 0.f;
        dy_s[r * CB + col] = v;
        if (r < p.B && cok) p.dy[(size_t)r * p.N + c] = v;
      }
    }
  __syncthreads();
  if (p.dw == nullptr) return;

  // (d) dW[k, c] = sum_r in[r,k] * dy[r,c]; thread = (col, k-group of 8)
  RowSrc src{p.in, p.ld_in, p.K, p.in2, p.ld_in2, p.K2};
  const int Kt = p.K + p.K2;
  for (int k0 = 0; k0 < Kt; k0 += KC) {
    for (int idx = tid; idx < RB * 32 * KC; idx += 256) {
      const int r = idx / KC, kk = idx % KC;
      in_s[kk * BP + r] = (r < p.B) ? src.at(r, k0 + kk) : 0.f;
    }
    __syncthreads();
    for (int kk = rg; kk < KC; kk += 8) {
      float a = 0.f;
      for (int r = 0; r < RB * 32; r += 4) {
        const float4 x = *reinterpret_cast<const float4*>(in_s + kk * BP + r);
        a = fmaf(x.x, dy_s[(r + 0) * CB + col], a);
        a = fmaf(x.y, dy_s[(r + 1) * CB + col], a);
        a = fmaf(x.z, dy_s[(r + 2) * CB + col], a);
        a = fmaf(x.w, dy_s[(r + 3) * CB + col], a);
      }
      if (cok && k0 + kk < Kt) p.dw[(size_t)(k0 + kk) * p.N + c] = a;
    }
    __syncthreads();
  }
}

__global__ __launch_bounds__(256) void k_fc_dinput(const t3d_fc_dinput_args p) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int RB = (p.B + 31) / 32, BP = RB * 32 + 4;
  float* in_s = sm;
  float* w_s = in_s + KC * BP;
  const int tid = threadIdx.x, col = tid & 31, rg = tid >> 5;
  const int c0 = blockIdx.x * CB, c = c0 + col;
  const int nvalid = min(CB, p.K - c0);
  float acc[MAXRB][4];
  RowSrc src{p.dy, p.N, p.N, nullptr, 0, 0};
  rows_gemm<true>(acc, src, p.B, RB, BP, p.w, p.N, c0, nvalid, in_s, w_s);
  if (c >= p.K) return;
#pragma unroll
  for (int rb = 0; rb < MAXRB; ++rb)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int r = rb * 32 + rg * 4 + j;
      if (rb < RB && r < p.B) {
        float v = p.alpha * acc[rb][j];
        if (p.add_in) v += p.add_in[(size_t)r * p.ld_add + c];
        p.din[(size_t)r * p.ld_din + c] = v;
      }
    }
}

size_t fc_lds_bytes(int B, bool bwd) {
  const int RB = (B + 31) / 32, BP = RB * 32 + 4;
  size_t f = (size_t)KC * BP + KC * CB + 8 * CB;
  if (bwd) f += (size_t)RB * 32 * CB;
  return f * sizeof(float);
}

}  // namespace

extern "C" int t3d_fc_fwd(const t3d_fc_fwd_args* a, t3d_stream_t stream) {
  if (!a || !a->in || !a->w || !a->out || (a->K2 > 0 && !a->in2)) return T3D_ERR_ARG;
  if (a->gamma && (!a->beta || !a->moving_mean || !a->moving_var || !a->mean || !a->invstd || !a->y)) return T3D_ERR_ARG;
  if (a->gamma && a->is_training && !a->decay) return T3D_ERR_ARG;
  if (a->B <= 0 || a->B > 32 * MAXRB || a->N <= 0 || a->K <= 0) return T3D_ERR_SHAPE;
  T3D_LAUNCH(k_fc_fwd, dim3((a->N + CB - 1) / CB), dim3(256), fc_lds_bytes(a->B, false),
                     static_cast<hipStream_t>(stream), *a);
  T3D_CHECK_LAUNCH();
  return T3D_OK;
}

extern "C" int t3d_fc_bwd(const t3d_fc_bwd_args* a, t3d_stream_t stream) {
  if (!a || !a->dy || (!a->dout && (!a->dy_next || !a->w_next))) return T3D_ERR_ARG;
  if (a->dw && (!a->in || (a->K2 > 0 && !a->in2))) return T3D_ERR_ARG;
  if (a->gamma && (!a->beta || !a->mean || !a->invstd || !a->y)) return T3D_ERR_ARG;
  if (a->act != T3D_ACT_NONE && !a->y) return T3D_ERR_ARG;
  if (a->B <= 0 || a->B > 32 * MAXRB || a->N <= 0) return T3D_ERR_SHAPE;
  T3D_LAUNCH(k_fc_bwd, dim3((a->N + CB - 1) / CB), dim3(256), fc_lds_bytes(a->B, true),
                     static_cast<hipStream_t>(stream), *a);
  T3D_CHECK_LAUNCH();
  return T3D_OK;
}

extern "C" int t3d_fc_dinput(const t3d_fc_dinput_args* a, t3d_stream_t stream) {
  if (!a || !a->dy || !a->w || !a->din) return T3D_ERR_ARG;
  if (a->B <= 0 || a->B > 32 * MAXRB || a->N <= 0 || a->K <= 0) return T3D_ERR_SHAPE;
  T3D_LAUNCH(k_fc_dinput, dim3((a->K + CB - 1) / CB), dim3(256), fc_lds_bytes(a->B, false),
                     static_cast<hipStream_t>(stream), *a);
  T3D_CHECK_LAUNCH();
  return T3D_OK;
}
