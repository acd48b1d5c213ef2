// Device bodies of the fully-connected kernels (fc.hip) -- a header so that the paired small-launch kernel (pair.hip) can run them
// beside the batch-norm finalizers in one launch.  See fc.hip for what they replace.
#pragma once
#include "common.h"

namespace {


constexpr int CB = 32;       // columns per workgroup
constexpr int MAXRB = 4;     // B <= 128
constexpr int NW = 8;        // waves per workgroup (the reduction dimension is split over them)
constexpr int NTH = NW * 64;
constexpr int RG = NTH / 32; // row groups of the epilogue thread map
constexpr int LDT = 33;      // padded row stride of the LDS tiles
// The bodies are written for NW waves / NTH threads.  As RIDERS inside a 256-thread GEMM launch (rider_dev.h) they run with
// NWP = 4 physical waves: every physical wave / thread then plays V = NW / NWP of the NW-wave form's waves / threads in turn
// (virtual wave pw + NWP*v, virtual row group rgp + (NWP*2)*v), so every sum is formed in the same order: bit-identical results.

#ifdef T3D_TRACE             // diagnostic builds: per-workgroup phase clock (tools/trace_fc.py)
__device__ unsigned long long* t3d_trace_fc_ptr = nullptr;
#define FC_MARK(slot) do { if (threadIdx.x == 0 && t3d_trace_fc_ptr) t3d_trace_fc_ptr[(size_t)bid * 8 + (slot)] = wall_clock64(); } while (0)
#else
#define FC_MARK(slot) do {} while (0)
#endif

struct RowSrc {              // [in | in2] row-concatenated input, B valid rows
  const float* in; int ld_in; int K;
  const float* in2; int ld_in2; int K2;
  int B;
  __device__ __forceinline__ float at(int r, int k) const {
    if (k < K) return in[(size_t)r * ld_in + k];
    if (k < K + K2) return in2[(size_t)r * ld_in2 + (k - K)];
    return 0.f;
  }
  // branch-free: clamped address, value masked afterwards (a load never sits behind a per-lane branch)
  __device__ __forceinline__ float at_nb(int r, int k) const {
    const bool first = k < K, valid = r < B && k < K + K2;
    const int rc = min(r, B - 1);
    const float* base = (first || in2 == nullptr) ? in : in2;
    const size_t off = (first || in2 == nullptr) ? (size_t)rc * ld_in + min(k, K - 1) : (size_t)rc * ld_in2 + min(k - K, K2 - 1);
    const float v = base[off];
    return valid ? v : 0.f;
  }
  // 4 consecutive reduction elements of one row: load4_raw only issues loads (clamped addresses), valid() says without touching
  // memory which values count; wave_gemm masks where it consumes (see there).
  __device__ __forceinline__ void load4_raw(int r, int k, float (&v)[4]) const {      // four clamped scalar loads, nothing else
    const int rc = min(r, B - 1);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int ke = k + e;
      const bool first = ke < K || in2 == nullptr;
      const float* base = first ? in : in2;
      const size_t off = first ? (size_t)rc * ld_in + min(ke, K - 1) : (size_t)rc * ld_in2 + min(ke - K, K2 - 1);
      v[e] = base[off];
    }
  }
  __device__ __forceinline__ bool valid(int r, int k) const { return r < B && k < K + K2; }
  __device__ __forceinline__ void load4(int r, int k, float (&v)[4]) const {
    load4_raw(r, k, v);
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = valid(r, k + e) ? v[e] : 0.f;
  }
};

// acc[rb] (32x32 tiles, rows rb*32.., cols c0..c0+31) += sum over this wave's k groups of A[row,k] * W(k,col)
//   WT == false: W(k,col) = w[k*ldw + c0+col]   (forward: weights [K,N])
//   WT == true : W(k,col) = w[(c0+col)*ldw + k] (input gradient: dy . W^T)
// RBT = compile-time number of 32-row blocks (1 for B <= 32: the shapes of the hot path; MAXRB otherwise).  GIF k-groups
// are loaded before the first MFMA of a batch: these layers are pure latency (operands tiny, read once per step from
// HBM/L2), so memory-level parallelism per wave is what matters -- with RBT = 1 a wave has its whole share of a
// K <= 1024 reduction in flight at once.
// (GIF_ = 0: the default batch; the 256-thread rider form asks for 8 -- it only changes how many loads are in flight, not the sums)
template <bool WT, int RBT, int GIF_ = 0>
__device__ __forceinline__ void wave_gemm(f32x16 (&acc)[RBT], const RowSrc& src, const float* __restrict__ w, int ldw,
                                          int Kred, int c0, int ncols, int wave_, int lane) {
  constexpr int GIF = GIF_ > 0 ? GIF_ : (RBT == 1 ? 16 : 4);
  const int wave = __builtin_amdgcn_readfirstlane(wave_);      // provably wave-uniform: the k-group bounds below are scalar branches
  const int l31 = lane & 31, h = lane >> 5;
#pragma unroll
  for (int rb = 0; rb < RBT; ++rb)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[rb][r] = 0.f;
  const int ngroups = (Kred + 7) >> 3;
  const bool cok = l31 < ncols;
  const int cc = c0 + min(l31, ncols - 1);
  // every wave owns a CONTIGUOUS run of k-groups: the four loads that share a 128-byte line of an input row (32 reduction
  // elements) are then issued back to back by one wave and hit the line while it is in flight; with the groups dealt out round
  // robin over the waves each line was fetched by four different waves (load + MFMA phase of the 512x512 layer 7.2 -> 6.2 us,
  // tools/trace_fc.py)
  const int gpw = (ngroups + NW - 1) / NW, gbeg = wave * gpw, gend = min(gbeg + gpw, ngroups);
  // Round 3: loads and masks apart, and the 16-byte and the scalar load forms in SEPARATE loops.  The former per-lane form (row
  // test, vector-or-scalar test, zero fill, each with its select right behind the load) compiled into exec-masked regions with an
  // `s_waitcnt vmcnt(0)` each: the "16 k-groups in flight" were 16 SERIAL round trips per wave in every FC kernel
  // (tools/kernel_branches.py: 260 branches in k_fc_fwd<1>).  A wave-uniform branch between the two forms INSIDE one loop is no
  // better: hipcc merges the arms into four dword loads on selected addresses.  So: groups that lie entirely inside the first input
  // block (and the weight rows, for the transposed form) take the vector loop -- nothing but 16-byte loads on clamped indices, then
  // masks + MFMAs; the few groups behind them (the one-hot / norm_box2D block, a ragged end) take the scalar loop.
  const bool vec_ok = (src.ld_in & 3) == 0 && (!WT || (ldw & 3) == 0);
  const int gfull = vec_ok ? ((WT ? min(src.K, Kred) : src.K) >> 3) : 0;      // groups [0, gfull) have k + 7 < K
  const int gv_end = min(gend, gfull);
  constexpr int GT = 2;                       // k-groups per batch of the scalar loop (it sees the one-hot block or a ragged end: 1-2 groups)
  auto mma_batch = [&](const int g0, const int glim, auto& a, auto& b, auto nb_tag) {
    constexpr int NB = decltype(nb_tag)::value;
#pragma unroll
    for (int u = 0; u < NB; ++u) {
      if (g0 + u < glim) {                    // scalar branch
        const int k = 8 * (g0 + u) + 4 * h;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float bv = (cok && k + i < Kred) ? b[u][i] : 0.f;
#pragma unroll
          for (int rb = 0; rb < RBT; ++rb) {
            const float av = src.valid(rb * 32 + l31, k + i) ? a[u][rb][i] : 0.f;
            acc[rb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[rb], 0, 0, 0);
          }
        }
      }
    }
  };
  for (int g0 = gbeg; g0 < gv_end; g0 += GIF) {
    float a[GIF][RBT][4], b[GIF][4];
#pragma unroll
    for (int u = 0; u < GIF; ++u) {
      if (g0 + u < gv_end) {                  // scalar branch with nothing but loads inside (a group past the run issues nothing)
        const int k = 8 * (g0 + u) + 4 * h;
#pragma unroll
        for (int rb = 0; rb < RBT; ++rb) {
          const float4 t = *reinterpret_cast<const float4*>(src.in + (size_t)min(rb * 32 + l31, src.B - 1) * src.ld_in + k);
          a[u][rb][0] = t.x; a[u][rb][1] = t.y; a[u][rb][2] = t.z; a[u][rb][3] = t.w;
        }
        if (!WT) {
#pragma unroll
          for (int i = 0; i < 4; ++i) b[u][i] = w[(size_t)(k + i) * ldw + cc];
        } else {
          const float4 t = *reinterpret_cast<const float4*>(w + (size_t)cc * ldw + k);
          b[u][0] = t.x; b[u][1] = t.y; b[u][2] = t.z; b[u][3] = t.w;
        }
      }
    }
    mma_batch(g0, gv_end, a, b, std::integral_constant<int, GIF>{});
  }
  for (int g0 = max(gbeg, gfull); g0 < gend; g0 += GT) {
    float a[GT][RBT][4], b[GT][4];
#pragma unroll
    for (int u = 0; u < GT; ++u) {
      if (g0 + u < gend) {                    // scalar branch; loads only inside
        const int k = 8 * (g0 + u) + 4 * h;
#pragma unroll
        for (int rb = 0; rb < RBT; ++rb) src.load4_raw(rb * 32 + l31, k, a[u][rb]);
#pragma unroll
        for (int i = 0; i < 4; ++i)
          b[u][i] = WT ? w[(size_t)cc * ldw + min(k + i, Kred - 1)] : w[(size_t)min(k + i, Kred - 1) * ldw + cc];
      }
    }
    mma_batch(g0, gend, a, b, std::integral_constant<int, GT>{});
  }
}

// Sum the NW waves' tiles through LDS.  Afterwards (virtual) thread t owns column (t & 31) and rows (t >> 5) + RG*j.
template <int RBT, int NWP = NW>
__device__ __forceinline__ void reduce_tiles(const f32x16 (&acc)[NW / NWP][RBT], float* red, float (&val)[NW / NWP][RBT * 32 / RG]) {
  constexpr int RB = RBT, NVAL = RBT * 32 / RG, V = NW / NWP, RGP = NWP * 2;
  const int tid = threadIdx.x, lane = tid & 63, pw = tid >> 6, l31 = lane & 31, h = lane >> 5;
  const int rows = RB * 32;
#pragma unroll
  for (int v = 0; v < V; ++v) {
    const int wave = pw + NWP * v;
#pragma unroll
    for (int rb = 0; rb < RBT; ++rb)
      {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = rb * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
          red[(wave * rows + row) * LDT + l31] = acc[v][rb][r];
        }
      }
  }
  __syncthreads();
  const int col = tid & 31, rgp = tid >> 5;
#pragma unroll
  for (int v = 0; v < V; ++v) {
    const int rg = rgp + RGP * v;
#pragma unroll
    for (int j = 0; j < NVAL; ++j) {
      const int row = rg + RG * j;
      float s = 0.f;
      if (row < rows) {
#pragma unroll
        for (int wv = 0; wv < NW; ++wv) s += red[(wv * rows + row) * LDT + col];
      }
      val[v][j] = s;
    }
  }
  __syncthreads();
}

// sum over all rows of the workgroup's per-(virtual-)thread partials, per column; result broadcast to every thread
template <int NWP = NW>
__device__ __forceinline__ float col_reduce(const float (&part)[NW / NWP], float* red) {
  constexpr int V = NW / NWP, RGP = NWP * 2;
  const int tid = threadIdx.x, col = tid & 31, rgp = tid >> 5;
  __syncthreads();
#pragma unroll
  for (int v = 0; v < V; ++v) red[(rgp + RGP * v) * CB + col] = part[v];
  __syncthreads();
  float s = 0.f;
#pragma unroll
  for (int g = 0; g < RG; ++g) s += red[g * CB + col];
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

template <int RBT, int NWP = NW>
__device__ __forceinline__ void fc_fwd_body(const t3d_fc_fwd_args& p, float* sm, const int bid, const int cb = CB) {
  constexpr int RB = RBT, NVAL = RBT * 32 / RG, V = NW / NWP, RGP = NWP * 2;
  const int tid = threadIdx.x, lane = tid & 63, pw = tid >> 6;
  const int col = tid & 31, rgp = tid >> 5;
  // cb <= CB columns per workgroup (round 3): the batch statistics are per column, so a layer may be cut into narrower column
  // runs and a wide layer spans more than N / 32 CUs (the 1024 -> 512 row-bias layer of conv6: 16 workgroups before); lanes past
  // cb idle in the MFMA and in the epilogue.
  const int c0 = bid * cb, c = c0 + col;
  const int nvalid = min(cb, p.N - c0);
  const bool cok = col < nvalid;

  FC_MARK(0);
  float y[V][NVAL];
  if (p.w != nullptr) {          // workgroup-uniform
    f32x16 acc[V][RBT];
    RowSrc src{p.in, p.ld_in, p.K, p.in2, p.ld_in2, p.K2, p.B};
#pragma unroll
    for (int v = 0; v < V; ++v) { wave_gemm<false, RBT, (NWP < NW ? 6 : 0)>(acc[v], src, p.w, p.N, p.K + p.K2, c0, nvalid, pw + NWP * v, lane); if (V > 1) __builtin_amdgcn_sched_barrier(0); }
    FC_MARK(1);
    reduce_tiles<RBT, NWP>(acc, sm, y);
  } else {                       // identity: a standalone batch-norm / dropout node on a [B,N] tensor
#pragma unroll
    for (int v = 0; v < V; ++v)
#pragma unroll
      for (int j = 0; j < NVAL; ++j) {
        const int r = rgp + RGP * v + RG * j;
        y[v][j] = (r < p.B && cok) ? p.in[(size_t)r * p.ld_in + c] : 0.f;
      }
  }
  FC_MARK(2);

  const float bias = (cok && p.bias) ? p.bias[c] : 0.f;
  float part[V];
#pragma unroll
  for (int v = 0; v < V; ++v) {
    part[v] = 0.f;
#pragma unroll
    for (int j = 0; j < NVAL; ++j) {
      y[v][j] += bias;
      if (rgp + RGP * v + RG * j < p.B) part[v] += y[v][j];
    }
  }
  const bool bn = p.gamma != nullptr;
  float mean = 0.f, invstd = 1.f, g = 1.f, be = 0.f;
  if (bn) {
    if (cok) { g = p.gamma[c]; be = p.beta[c]; }
    if (p.is_training) {
      mean = col_reduce<NWP>(part, sm) / (float)p.B;
      float vpart[V];
#pragma unroll
      for (int v = 0; v < V; ++v) {
        vpart[v] = 0.f;
#pragma unroll
        for (int j = 0; j < NVAL; ++j)
          if (rgp + RGP * v + RG * j < p.B) { const float d = y[v][j] - mean; vpart[v] = fmaf(d, d, vpart[v]); }
      }
      const float var = col_reduce<NWP>(vpart, sm) / (float)p.B;
      invstd = 1.0f / sqrtf(var + p.eps);
      if (cok && rgp == 0) {
        const float d = p.decay[0];
        const float var_ema = p.unbiased_ema ? var * ((float)p.B / (float)max(p.B - 1, 1)) : var;
        p.moving_mean[c] = p.moving_mean[c] * d + mean * (1.f - d);
        p.moving_var[c] = p.moving_var[c] * d + var_ema * (1.f - d);
      }
    } else if (cok) {
      mean = p.moving_mean[c];
      invstd = 1.0f / sqrtf(p.moving_var[c] + p.eps);
    }
    if (cok && rgp == 0) { p.mean[c] = mean; p.invstd[c] = invstd; }
  }
  FC_MARK(3);
  if (!cok) return;
  const float inv_keep = p.drop_mask ? 1.0f / p.keep_prob : 1.f;
#pragma unroll
  for (int v = 0; v < V; ++v)
#pragma unroll
    for (int j = 0; j < NVAL; ++j) {
      const int r = rgp + RGP * v + RG * j;
      if (r < p.B) {
        const float yv = y[v][j];
        if (p.y) p.y[(size_t)r * p.N + c] = yv;
        float z = bn ? (yv - mean) * invstd * g + be : yv;
        z = act_fwd(z, p.act, p.leaky_alpha);
        if (p.drop_mask) z *= p.drop_mask[(size_t)r * p.N + c] * inv_keep;
        if (p.add_in && c < p.add_n) z += p.add_in[(size_t)r * p.ld_add + c];
        p.out[(size_t)r * p.ld_out + c] = z;
      }
    }
  FC_MARK(4);
}

template <int RBT, int NWP = NW>
__device__ __forceinline__ void fc_bwd_body(const t3d_fc_bwd_args& p, float* sm, const int bid) {
  constexpr int RB = RBT, NVAL = RBT * 32 / RG, V = NW / NWP, RGP = NWP * 2;
  const int tid = threadIdx.x, lane = tid & 63, pw = tid >> 6;
  const int col = tid & 31, rgp = tid >> 5;
  const int c0 = bid * CB, c = c0 + col;
  const int nvalid = min(CB, p.N - c0);
  const bool cok = c < p.N;

  // the dW phase's input operand does not depend on anything computed here: request it first, it lands under (a)-(c)
  constexpr int XKB = RBT == 1 ? (NWP < NW ? 2 : 4) : 0;   // 32-channel blocks per wave held in registers
  float xa[XKB > 0 ? XKB : 1][16];
  const int Kt = p.K + p.K2, nkb = (Kt + 31) / 32;
  if (XKB > 0 && p.dw != nullptr) {
    RowSrc xs{p.in, p.ld_in, p.K, p.in2, p.ld_in2, p.K2, p.B};
#pragma unroll
    for (int j = 0; j < XKB; ++j) {
      const int k = (pw + NWP * j) * 32 + (lane & 31);
#pragma unroll
      for (int e = 0; e < 16; ++e) xa[j][e] = xs.at_nb(8 * (e >> 2) + 4 * (lane >> 5) + (e & 3), k);   // k >= Kt reads as 0
    }
  }

  // (a) gradient w.r.t. this layer's output: given, or dy_next . w_next^T on the fly
  float gout[V][NVAL];
  if (p.dout != nullptr) {
#pragma unroll
    for (int v = 0; v < V; ++v)
#pragma unroll
      for (int j = 0; j < NVAL; ++j) {
        const int r = rgp + RGP * v + RG * j;
        gout[v][j] = (r < p.B && cok) ? p.dout[(size_t)r * p.ld_dout + c] : 0.f;
      }
  } else {
    f32x16 acc[V][RBT];
    RowSrc src{p.dy_next, p.N_next, p.N_next, nullptr, 0, 0, p.B};
#pragma unroll
    for (int v = 0; v < V; ++v) { wave_gemm<true, RBT, (NWP < NW ? 6 : 0)>(acc[v], src, p.w_next, p.N_next, p.N_next, c0, nvalid, pw + NWP * v, lane); if (V > 1) __builtin_amdgcn_sched_barrier(0); }
    reduce_tiles<RBT, NWP>(acc, sm, gout);
  }

  // (b) dropout / activation backward, (c) batch-norm backward over the B rows
  const bool bn = p.gamma != nullptr;
  float mean = 0.f, invstd = 1.f, g = 1.f, be = 0.f;
  if (bn && cok) { mean = p.mean[c]; invstd = p.invstd[c]; g = p.gamma[c]; be = p.beta[c]; }
  const float inv_keep = p.drop_mask ? 1.0f / p.keep_prob : 1.f;
  float xh[V][NVAL];
  float s1[V], s2[V];
#pragma unroll
  for (int v = 0; v < V; ++v) {
    s1[v] = 0.f;
    s2[v] = 0.f;
#pragma unroll
    for (int j = 0; j < NVAL; ++j) {
      const int r = rgp + RGP * v + RG * j;
      float dz = 0.f, x = 0.f;
      if (r < p.B && cok) {
        const float yv = p.y ? p.y[(size_t)r * p.N + c] : 0.f;
        x = bn ? (yv - mean) * invstd : yv;
        const float z = bn ? x * g + be : yv;
        dz = gout[v][j];
        if (p.drop_mask) dz *= p.drop_mask[(size_t)r * p.N + c] * inv_keep;
        dz *= act_bwd(z, p.act, p.leaky_alpha);
        s1[v] += dz;
        s2[v] = fmaf(dz, x, s2[v]);
      }
      gout[v][j] = dz;
      xh[v][j] = x;
    }
  }
  float dbias = 0.f;
  if (bn && p.bn_training) {
    const float dbeta = col_reduce<NWP>(s1, sm);
    const float dgamma = col_reduce<NWP>(s2, sm);
    if (cok && rgp == 0) {
      if (p.dbeta) p.dbeta[c] = dbeta;
      if (p.dgamma) p.dgamma[c] = dgamma;
    }
    const float invB = 1.0f / (float)p.B, c1 = g * invstd;
#pragma unroll
    for (int v = 0; v < V; ++v)
#pragma unroll
      for (int j = 0; j < NVAL; ++j) gout[v][j] = c1 * (gout[v][j] - dbeta * invB - xh[v][j] * dgamma * invB);
  } else if (bn) {
    const float c1 = g * invstd;
#pragma unroll
    for (int v = 0; v < V; ++v)
#pragma unroll
      for (int j = 0; j < NVAL; ++j) gout[v][j] *= c1;
  } else {
    dbias = col_reduce<NWP>(s1, sm);
  }
  if (cok && rgp == 0 && p.dbias) p.dbias[c] = dbias;   // exactly 0 under training-mode BN

  // dy -> global and LDS ([rows][LDT], zero padded)
  float* dy_s = sm;
  __syncthreads();
#pragma unroll
  for (int v = 0; v < V; ++v)
#pragma unroll
    for (int j = 0; j < NVAL; ++j) {
      const int r = rgp + RGP * v + RG * j;
      if (r < RB * 32) {
        const float val = (r < p.B && cok) ? gout[v][j] : 0.f;
        dy_s[r * LDT + col] = val;
        if (r < p.B && cok) p.dy[(size_t)r * p.N + c] = val;
      }
    }
  __syncthreads();
  if (p.dw == nullptr) return;

  // (d) dW[k, c] = sum_r in[r,k] * dy[r,c]: one 32x32 MFMA tile per 32 input channels, reduction over rows (a block is computed
  // whole by one wave, in the same MFMA order whichever wave takes it and whether its operand was prefetched or not)
  RowSrc src{p.in, p.ld_in, p.K, p.in2, p.ld_in2, p.K2, p.B};
  const int l31 = lane & 31, h = lane >> 5;
  auto store_block = [&](int kb, const f32x16& acc) {
    if (l31 < nvalid) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int kk = kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (kk < Kt) p.dw[(size_t)kk * p.N + c0 + l31] = acc[r];
      }
    }
  };
#pragma unroll
  for (int j = 0; j < XKB; ++j) {                        // blocks whose operand was prefetched at kernel start
    const int kb = pw + NWP * j;
    if (kb < nkb) {
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
      for (int e = 0; e < 16; ++e)
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(xa[j][e], dy_s[(8 * (e >> 2) + 4 * h + (e & 3)) * LDT + l31], acc, 0, 0, 0);
      store_block(kb, acc);
    }
  }
  for (int kb = pw + NWP * XKB; kb < nkb; kb += NWP) {
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const int k = kb * 32 + l31;
    // the block's input operand is requested 32 loads at a time ahead of their MFMAs: with four loads per dependent round trip,
    // B = 128 rows walked 16 round trips per block (k_fc_bwd<4>: 33 us per launch, profiles/r02_bf16_v2); all 64 at once spill
    constexpr int GB = RB * 4 < 8 ? RB * 4 : 8;
#pragma unroll 1
    for (int g0 = 0; g0 < RB * 4; g0 += GB) {
      float a[GB][4];
#pragma unroll
      for (int u = 0; u < GB; ++u)
#pragma unroll
        for (int i = 0; i < 4; ++i) a[u][i] = src.at_nb(8 * (g0 + u) + 4 * h + i, k);
#pragma unroll
      for (int u = 0; u < GB; ++u)
#pragma unroll
        for (int i = 0; i < 4; ++i)
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u][i], dy_s[(8 * (g0 + u) + 4 * h + i) * LDT + l31], acc, 0, 0, 0);
    }
    store_block(kb, acc);
  }
}

template <int RBT, int NWP = NW>
__device__ __forceinline__ void fc_dinput_body(const t3d_fc_dinput_args& p, float* sm, const int bid) {
  constexpr int RB = RBT, NVAL = RBT * 32 / RG, V = NW / NWP, RGP = NWP * 2;
  const int tid = threadIdx.x, lane = tid & 63, pw = tid >> 6;
  const int col = tid & 31, rgp = tid >> 5;
  const int c0 = bid * CB, c = c0 + col;
  const int nvalid = min(CB, p.K - c0);
  f32x16 acc[V][RBT];
  RowSrc src{p.dy, p.N, p.N, nullptr, 0, 0, p.B};
#pragma unroll
  for (int v = 0; v < V; ++v) wave_gemm<true, RBT, (NWP < NW ? 6 : 0)>(acc[v], src, p.w, p.N, p.N, c0, nvalid, pw + NWP * v, lane);
  float val[V][NVAL];
  reduce_tiles<RBT, NWP>(acc, sm, val);
  const bool cok = c < p.K;
  float s1[V], s2[V];
#pragma unroll
  for (int v = 0; v < V; ++v) {
    s1[v] = 0.f;
    s2[v] = 0.f;
#pragma unroll
    for (int j = 0; j < NVAL; ++j) {
      const int r = rgp + RGP * v + RG * j;
      if (r < p.B && cok) {
        float o = p.alpha * val[v][j];
        if (p.add_in) o += p.add_in[(size_t)r * p.ld_add + c];
        p.din[(size_t)r * p.ld_din + c] = o;
        if (p.bn_coef) {            // pooled form of the batch-norm backward statistics on this column (K11c)
          const float live = p.bn_pooled[(size_t)r * p.bn_ld_pooled + c] > 0.f ? 1.f : 0.f;
          const float g = o * live;
          p.bn_dpool[(size_t)r * p.K + c] = g;
          s1[v] += g;
          s2[v] = fmaf(g, p.bn_ysel[(size_t)r * p.K + c], s2[v]);
        }
      }
    }
  }
  if (p.bn_coef == nullptr) return;          // workgroup-uniform
  const float t1 = col_reduce<NWP>(s1, sm), t2 = col_reduce<NWP>(s2, sm);
  if (cok && rgp == 0) {
    if (p.bn_frozen) {
      p.bn_coef[c] = p.bn_scale[c];
      p.bn_coef[p.K + c] = 0.f;
      p.bn_coef[2 * p.K + c] = 0.f;
      return;
    }
    const double mean = p.bn_mean[c], invstd = p.bn_invstd[c], gamma = p.bn_gamma[c], n = p.bn_count;
    const double dbeta = t1;
    const double dgamma = invstd * ((double)t2 - mean * (double)t1);
    if (p.bn_dbeta) p.bn_dbeta[c] = (float)dbeta;
    if (p.bn_dgamma) p.bn_dgamma[c] = (float)dgamma;
    const double c1 = gamma * invstd, k3 = dgamma / n * invstd;
    p.bn_coef[c] = (float)c1;
    p.bn_coef[p.K + c] = (float)(-c1 * k3);
    p.bn_coef[2 * p.K + c] = (float)(c1 * (k3 * mean - dbeta / n));
  }
}


inline size_t fc_lds_bytes(int B) {
  const int RB = (B + 31) / 32;
  return (size_t)NW * RB * 32 * LDT * sizeof(float);
}

}  // namespace
