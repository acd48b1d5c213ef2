// Per-point shared-MLP layer on fp32 MFMA: forward, data-gradient and weight-gradient GEMMs.
//
// Replaces the reference's tf_util.conv2d 1x1 call sites (models/tf_util.py:1258-1323, called at
// sunrgbd/sunrgbd_detection/semisup_models.py:76-135,172-183,224-239,354-369) and their autodiff twins.
//
// One workgroup = 4 waves (2x2) computes a 128 x BN output tile with v_mfma_f32_32x32x2_f32 (exact fp32,
// k-ordered fma chain).  Operand tiles are staged global -> registers -> LDS with the element-wise work
// fused into the staging pass: batch-norm apply + ReLU of the producing layer on activations
// (t3d_act_src), batch-norm backward on gradients (t3d_dy_src).  Two LDS tile formats:
//   type R: [lane_dim][BK+4]   reduction index contiguous; one ds_read_b128 feeds four MFMAs
//   type C: [BK][lane_dim]     lane index contiguous;     ds_read_b32 per MFMA (conflict-free)
// The MFMA k index of step i of group g on lane half h is 8g+4h+i for both operands, so either
// format can be paired with either.
#include "common.h"

namespace {

constexpr int BK = 32;
constexpr int LDR = BK + 4;
constexpr int NT = 256;

__device__ __forceinline__ float4 f4zero() { return make_float4(0.f, 0.f, 0.f, 0.f); }

// ---------------------------------------------------------------------------------------------
// loaders: fetch() issues the global loads, xform() does the fused element-wise math afterwards
// ---------------------------------------------------------------------------------------------
struct ActLoader {
  t3d_act_src s;
  int K;     // valid columns
  int rpf;   // rows per frustum
  struct Raw { float4 x; };
  struct Coef { float4 sc, sh; };
  __device__ __forceinline__ Coef fetch_coef(int col) const {
    Coef c;
    c.sc = make_float4(1.f, 1.f, 1.f, 1.f);
    c.sh = f4zero();
    if (s.scale != nullptr && col < K) {
      if (col + 3 < K) {
        c.sc = *reinterpret_cast<const float4*>(s.scale + col);
        c.sh = *reinterpret_cast<const float4*>(s.shift + col);
      } else {
        float a[4] = {1.f, 1.f, 1.f, 1.f}, b[4] = {0.f, 0.f, 0.f, 0.f};
        for (int e = 0; e < 4; ++e)
          if (col + e < K) { a[e] = s.scale[col + e]; b[e] = s.shift[col + e]; }
        c.sc = make_float4(a[0], a[1], a[2], a[3]);
        c.sh = make_float4(b[0], b[1], b[2], b[3]);
      }
    }
    return c;
  }
  __device__ __forceinline__ Raw fetch(int row, int col) const {
    Raw r;
    r.x = (col < K) ? *reinterpret_cast<const float4*>(s.x + (size_t)row * s.ldx + s.coff + col) : f4zero();
    return r;
  }
  __device__ __forceinline__ float4 xform(const Raw& r, const Coef& c, int row, int col) const {
    float v[4] = {r.x.x, r.x.y, r.x.z, r.x.w};
    const float sc[4] = {c.sc.x, c.sc.y, c.sc.z, c.sc.w};
    const float sh[4] = {c.sh.x, c.sh.y, c.sh.z, c.sh.w};
    const bool has_bn = s.scale != nullptr;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float t = v[e];
      if (has_bn) t = fmaf(t, sc[e], sh[e]);
      if (s.relu) t = fmaxf(t, 0.f);
      v[e] = (col + e < K) ? t : 0.f;
    }
    if (s.sub != nullptr) {
      const int b = row / rpf;
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (col + e < K) v[e] -= s.sub[(size_t)b * s.sub_ld + col + e];
    }
    return make_float4(v[0], v[1], v[2], v[3]);
  }
};

struct DyLoader {
  t3d_dy_src s;
  int N;
  int rpf;
  struct Raw { float4 dz, y; };
  struct Coef { float4 c0, c1, c2; };
  __device__ __forceinline__ Coef fetch_coef(int col) const {
    Coef c;
    if (col < N) {
      c.c0 = *reinterpret_cast<const float4*>(s.coef + col);
      c.c1 = *reinterpret_cast<const float4*>(s.coef + N + col);
      c.c2 = *reinterpret_cast<const float4*>(s.coef + 2 * N + col);
    } else {
      c.c0 = c.c1 = c.c2 = f4zero();
    }
    return c;
  }
  __device__ __forceinline__ Raw fetch(int row, int col) const {
    Raw r;
    if (col >= N) { r.dz = r.y = f4zero(); return r; }
    r.y = *reinterpret_cast<const float4*>(s.y + (size_t)row * N + col);
    if (s.dz != nullptr) {
      r.dz = *reinterpret_cast<const float4*>(s.dz + (size_t)row * N + col);
    } else {
      const int b = row / rpf, rin = row - b * rpf;
      const int4 a = *reinterpret_cast<const int4*>(s.argidx + (size_t)b * N + col);
      const float4 g = *reinterpret_cast<const float4*>(s.dpool + (size_t)b * N + col);
      r.dz = make_float4(a.x == rin ? g.x : 0.f, a.y == rin ? g.y : 0.f, a.z == rin ? g.z : 0.f,
                         a.w == rin ? g.w : 0.f);
    }
    return r;
  }
  __device__ __forceinline__ float4 xform(const Raw& r, const Coef& c, int, int) const {
    return make_float4(fmaf(c.c0.x, r.dz.x, fmaf(c.c1.x, r.y.x, c.c2.x)), fmaf(c.c0.y, r.dz.y, fmaf(c.c1.y, r.y.y, c.c2.y)),
                       fmaf(c.c0.z, r.dz.z, fmaf(c.c1.z, r.y.z, c.c2.z)), fmaf(c.c0.w, r.dz.w, fmaf(c.c1.w, r.y.w, c.c2.w)));
  }
};

struct WLoader {
  const float* w;
  int ld;
  int rows, cols;   // valid extent; cols % 4 == 0
  struct Raw { float4 x; };
  struct Coef {};
  __device__ __forceinline__ Coef fetch_coef(int) const { return Coef(); }
  __device__ __forceinline__ Raw fetch(int row, int col) const {
    Raw r;
    r.x = (row < rows && col < cols) ? *reinterpret_cast<const float4*>(w + (size_t)row * ld + col) : f4zero();
    return r;
  }
  __device__ __forceinline__ float4 xform(const Raw& r, const Coef&, int, int) const { return r.x; }
};

// ---------------------------------------------------------------------------------------------
// staging of one [DIM x BK] operand tile through registers into LDS
// ---------------------------------------------------------------------------------------------
template <int DIM, bool TYPE_R, class L>
struct Stager {
  static constexpr int NV = DIM * (BK / 4) / NT;
  static constexpr int LDS_FLOATS = TYPE_R ? DIM * LDR : BK * DIM;
  typename L::Raw raw[NV];
  typename L::Coef coef;
  int lane0, red0;

  __device__ __forceinline__ static void coords(int tid, int q, int& lane_i, int& red_i) {
    const int f = tid + NT * q;
    if (TYPE_R) { lane_i = f >> 3; red_i = (f & 7) * 4; }
    else { constexpr int C4 = DIM / 4; red_i = f / C4; lane_i = (f % C4) * 4; }
  }
  // TYPE_C: the thread's column chunk never changes -> fetch its coefficients once
  __device__ __forceinline__ void init(const L& l, int lane0_, int tid) {
    lane0 = lane0_;
    if (!TYPE_R) { int li, ri; coords(tid, 0, li, ri); coef = l.fetch_coef(lane0 + li); }
  }
  __device__ __forceinline__ void fetch(const L& l, int red0_, int tid) {
    red0 = red0_;
    if (TYPE_R) { int li, ri; coords(tid, 0, li, ri); coef = l.fetch_coef(red0 + ri); }
#pragma unroll
    for (int q = 0; q < NV; ++q) {
      int li, ri; coords(tid, q, li, ri);
      raw[q] = TYPE_R ? l.fetch(lane0 + li, red0 + ri) : l.fetch(red0 + ri, lane0 + li);
    }
  }
  __device__ __forceinline__ void store(const L& l, float* tile, int tid) {
#pragma unroll
    for (int q = 0; q < NV; ++q) {
      int li, ri; coords(tid, q, li, ri);
      if (TYPE_R) {
        const float4 v = l.xform(raw[q], coef, lane0 + li, red0 + ri);
        *reinterpret_cast<float4*>(tile + li * LDR + ri) = v;
      } else {
        const float4 v = l.xform(raw[q], coef, red0 + ri, lane0 + li);
        *reinterpret_cast<float4*>(tile + ri * DIM + li) = v;
      }
    }
  }
};

// one BK-deep step of the wave's TM x TN grid of 32x32 MFMA tiles
template <int TM, int TN, bool AR, int DIMA, bool BR, int DIMB>
__device__ __forceinline__ void mma_step(const float* As, const float* Bs, int a0, int b0, f32x16 (&acc)[TM][TN],
                                         int lane) {
  const int l31 = lane & 31, h = lane >> 5;
#pragma unroll
  for (int g = 0; g < BK / 8; ++g) {
    float a[TM][4], b[TN][4];
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
      if (AR) {
        const float4 v = *reinterpret_cast<const float4*>(As + (a0 + tm * 32 + l31) * LDR + 8 * g + 4 * h);
        a[tm][0] = v.x; a[tm][1] = v.y; a[tm][2] = v.z; a[tm][3] = v.w;
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) a[tm][i] = As[(8 * g + 4 * h + i) * DIMA + a0 + tm * 32 + l31];
      }
    }
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
      if (BR) {
        const float4 v = *reinterpret_cast<const float4*>(Bs + (b0 + tn * 32 + l31) * LDR + 8 * g + 4 * h);
        b[tn][0] = v.x; b[tn][1] = v.y; b[tn][2] = v.z; b[tn][3] = v.w;
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) b[tn][i] = Bs[(8 * g + 4 * h + i) * DIMB + b0 + tn * 32 + l31];
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
          acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[tm][i], b[tn][i], acc[tm][tn], 0, 0, 0);
  }
}

template <int TM, int TN>
__device__ __forceinline__ void zero_acc(f32x16 (&acc)[TM][TN]) {
#pragma unroll
  for (int tm = 0; tm < TM; ++tm)
#pragma unroll
    for (int tn = 0; tn < TN; ++tn)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[tm][tn][r] = 0.f;
}

// register-prefetched main loop over the reduction range [red_begin, red_end)
template <int TM, int TN, class SA, class SB, class LA, class LB, bool AR, int DIMA, bool BR, int DIMB>
__device__ __forceinline__ void gemm_mainloop(SA& sa, SB& sb, const LA& la, const LB& lb, float* As, float* Bs,
                                              int red_begin, int red_end, int a0, int b0, f32x16 (&acc)[TM][TN],
                                              int tid) {
  const int lane = tid & 63;
  sa.fetch(la, red_begin, tid);
  sb.fetch(lb, red_begin, tid);
  sa.store(la, As, tid);
  sb.store(lb, Bs, tid);
  __syncthreads();
  for (int red = red_begin; red < red_end; red += BK) {
    const bool more = red + BK < red_end;
    if (more) { sa.fetch(la, red + BK, tid); sb.fetch(lb, red + BK, tid); }
    mma_step<TM, TN, AR, DIMA, BR, DIMB>(As, Bs, a0, b0, acc, lane);
    __syncthreads();
    if (more) { sa.store(la, As, tid); sb.store(lb, Bs, tid); }
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------------
template <int BN>
__global__ __launch_bounds__(NT, 2) void k_pointmlp_fwd(const t3d_pointmlp_fwd_args p) {
  constexpr int BM = 128, TM = 2, TN = BN / 64;
  using SA = Stager<BM, true, ActLoader>;
  using SB = Stager<BN, false, WLoader>;
  __shared__ __attribute__((aligned(16))) float smem[SA::LDS_FLOATS + SB::LDS_FLOATS];
  float* As = smem;
  float* Bs = smem + SA::LDS_FLOATS;

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid >> 1, wn = wid & 1;
  const int tiles_n = p.N / BN;
  const int lin = xcd_remap(blockIdx.x, gridDim.x);
  const int tile_m = lin / tiles_n, tile_n = lin % tiles_n;
  const int row0 = tile_m * BM, col0 = tile_n * BN;

  ActLoader la{p.a, p.K, p.rows_per_frustum};
  WLoader lb{p.w, p.N, p.K, p.N};
  SA sa; SB sb;
  sa.init(la, row0, tid);
  sb.init(lb, col0, tid);

  f32x16 acc[TM][TN];
  zero_acc<TM, TN>(acc);
  const int kred = (p.K + BK - 1) / BK * BK;
  gemm_mainloop<TM, TN, SA, SB, ActLoader, WLoader, true, BM, false, BN>(sa, sb, la, lb, As, Bs, 0, kred, wm * 64,
                                                                        wn * (BN / 2), acc, tid);

  // epilogue: + bias (+ per-frustum row bias), store y, column statistics, optional pool partials
  const int l31 = lane & 31, h = lane >> 5;
  const int b = row0 / p.rows_per_frustum;
  const bool pool = p.pmax != nullptr;
  float* red = smem;   // [2][BN] x 6 quantities
  float csum[TN], csq[TN], cmax[TN], cmin[TN];
  int amax[TN], amin[TN];
#pragma unroll
  for (int tn = 0; tn < TN; ++tn) {
    const int col = col0 + wn * (BN / 2) + tn * 32 + l31;
    float add = p.bias ? p.bias[col] : 0.f;
    if (p.rowbias) add += p.rowbias[(size_t)b * p.N + col];
    float s = 0.f, ss = 0.f, mx = -INFINITY, mn = INFINITY;
    int ax = -1, an = -1;
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = row0 + wm * 64 + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        const float v = acc[tm][tn][r] + add;
        p.y[(size_t)row * p.N + col] = v;
        s += v;
        ss = fmaf(v, v, ss);
        if (pool) {
          const bool keep = p.rowmask ? (p.rowmask[row] != 0.f) : true;
          const int rin = row - b * p.rows_per_frustum;
          if (keep && v > mx) { mx = v; ax = rin; }
          if (keep && v < mn) { mn = v; an = rin; }
        }
      }
    }
    // combine the two lane halves (rows +4): lower row index wins ties
    s += __shfl_xor(s, 32, 64);
    ss += __shfl_xor(ss, 32, 64);
    if (pool) {
      const float omx = __shfl_xor(mx, 32, 64), omn = __shfl_xor(mn, 32, 64);
      const int oax = __shfl_xor(ax, 32, 64), oan = __shfl_xor(an, 32, 64);
      if (oax >= 0 && (omx > mx || ax < 0 || (omx == mx && oax < ax))) { mx = omx; ax = oax; }
      if (oan >= 0 && (omn < mn || an < 0 || (omn == mn && oan < an))) { mn = omn; an = oan; }
    }
    csum[tn] = s; csq[tn] = ss; cmax[tn] = mx; cmin[tn] = mn; amax[tn] = ax; amin[tn] = an;
  }
  __syncthreads();
  if (h == 0) {
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
      const int c = wn * (BN / 2) + tn * 32 + l31;
      red[(0 * 2 + wm) * BN + c] = csum[tn];
      red[(1 * 2 + wm) * BN + c] = csq[tn];
      if (pool) {
        red[(2 * 2 + wm) * BN + c] = cmax[tn];
        red[(3 * 2 + wm) * BN + c] = cmin[tn];
        reinterpret_cast<int*>(red)[(4 * 2 + wm) * BN + c] = amax[tn];
        reinterpret_cast<int*>(red)[(5 * 2 + wm) * BN + c] = amin[tn];
      }
    }
  }
  __syncthreads();
  if (tid < BN) {
    const int c = tid;
    const size_t o = (size_t)tile_m * p.N + col0 + c;
    p.psum[o] = red[(0 * 2 + 0) * BN + c] + red[(0 * 2 + 1) * BN + c];
    p.psumsq[o] = red[(1 * 2 + 0) * BN + c] + red[(1 * 2 + 1) * BN + c];
    if (pool) {
      float mx = red[(2 * 2 + 0) * BN + c], mn = red[(3 * 2 + 0) * BN + c];
      int ax = reinterpret_cast<int*>(red)[(4 * 2 + 0) * BN + c], an = reinterpret_cast<int*>(red)[(5 * 2 + 0) * BN + c];
      const float mx1 = red[(2 * 2 + 1) * BN + c], mn1 = red[(3 * 2 + 1) * BN + c];
      const int ax1 = reinterpret_cast<int*>(red)[(4 * 2 + 1) * BN + c], an1 = reinterpret_cast<int*>(red)[(5 * 2 + 1) * BN + c];
      if (ax1 >= 0 && (ax < 0 || mx1 > mx)) { mx = mx1; ax = ax1; }
      if (an1 >= 0 && (an < 0 || mn1 < mn)) { mn = mn1; an = an1; }
      p.pmax[o] = mx; p.pmin[o] = mn; p.pamax[o] = ax; p.pamin[o] = an;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// data gradient
// ---------------------------------------------------------------------------------------------
template <int BN>   // BN = tile width over the layer's INPUT channels K
__global__ __launch_bounds__(NT, 2) void k_pointmlp_dgrad(const t3d_pointmlp_dgrad_args p) {
  constexpr int BM = 128, TM = 2, TN = BN / 64;
  using SA = Stager<BM, true, DyLoader>;
  using SB = Stager<BN, true, WLoader>;
  __shared__ __attribute__((aligned(16))) float smem[SA::LDS_FLOATS + SB::LDS_FLOATS];
  float* As = smem;
  float* Bs = smem + SA::LDS_FLOATS;

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid >> 1, wn = wid & 1;
  const int tiles_n = p.K / BN;
  const int lin = xcd_remap(blockIdx.x, gridDim.x);
  const int tile_m = lin / tiles_n, tile_n = lin % tiles_n;
  const int row0 = tile_m * BM, col0 = tile_n * BN;

  DyLoader la{p.dy, p.N, p.rows_per_frustum};
  WLoader lb{p.w, p.N, p.K, p.N};
  SA sa; SB sb;
  sa.init(la, row0, tid);
  sb.init(lb, col0, tid);

  f32x16 acc[TM][TN];
  zero_acc<TM, TN>(acc);
  const int nred = (p.N + BK - 1) / BK * BK;
  gemm_mainloop<TM, TN, SA, SB, DyLoader, WLoader, true, BM, true, BN>(sa, sb, la, lb, As, Bs, 0, nred, wm * 64,
                                                                      wn * (BN / 2), acc, tid);

  const int l31 = lane & 31, h = lane >> 5;
  const bool relu_mask = p.prev_y != nullptr;
  const bool stats = p.psum_dz != nullptr;
  float* red = smem;
  float cs1[TN], cs2[TN];
#pragma unroll
  for (int tn = 0; tn < TN; ++tn) {
    const int col = col0 + wn * (BN / 2) + tn * 32 + l31;
    const float psc = relu_mask ? p.prev_scale[col] : 0.f, psh = relu_mask ? p.prev_shift[col] : 0.f;
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = row0 + wm * 64 + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        const size_t o = (size_t)row * p.K + col;
        float v = acc[tm][tn][r];
        if (p.add_in) v += p.add_in[o];
        if (relu_mask) {
          const float yp = p.prev_y[o];
          if (!(fmaf(yp, psc, psh) > 0.f)) v = 0.f;
          s1 += v;
          s2 = fmaf(v, yp, s2);
        }
        p.out[o] = v;
      }
    }
    s1 += __shfl_xor(s1, 32, 64);
    s2 += __shfl_xor(s2, 32, 64);
    cs1[tn] = s1; cs2[tn] = s2;
  }
  if (stats) {
    __syncthreads();
    if (h == 0) {
#pragma unroll
      for (int tn = 0; tn < TN; ++tn) {
        const int c = wn * (BN / 2) + tn * 32 + l31;
        red[(0 * 2 + wm) * BN + c] = cs1[tn];
        red[(1 * 2 + wm) * BN + c] = cs2[tn];
      }
    }
    __syncthreads();
    if (tid < BN) {
      const size_t o = (size_t)tile_m * p.K + col0 + tid;
      p.psum_dz[o] = red[(0 * 2 + 0) * BN + tid] + red[(0 * 2 + 1) * BN + tid];
      p.psum_dzy[o] = red[(1 * 2 + 0) * BN + tid] + red[(1 * 2 + 1) * BN + tid];
    }
  }
}

// ---------------------------------------------------------------------------------------------
// weight gradient (split over rows)
// ---------------------------------------------------------------------------------------------
template <int BMK, int BN>
__global__ __launch_bounds__(NT, 2) void k_pointmlp_wgrad(const t3d_pointmlp_wgrad_args p) {
  constexpr int TM = BMK / 64, TN = BN / 64;
  using SA = Stager<BMK, false, ActLoader>;
  using SB = Stager<BN, false, DyLoader>;
  __shared__ __attribute__((aligned(16))) float smem[SA::LDS_FLOATS + SB::LDS_FLOATS];
  float* As = smem;
  float* Bs = smem + SA::LDS_FLOATS;

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid >> 1, wn = wid & 1;
  const int tiles_k = (p.K + BMK - 1) / BMK, tiles_n = p.N / BN;
  const int lin = xcd_remap(blockIdx.x, gridDim.x);
  const int split = lin / (tiles_k * tiles_n);
  const int t = lin % (tiles_k * tiles_n);
  const int k0 = (t / tiles_n) * BMK, n0 = (t % tiles_n) * BN;

  ActLoader la{p.a, p.K, p.rows_per_frustum};
  DyLoader lb{p.dy, p.N, p.rows_per_frustum};
  SA sa; SB sb;
  sa.init(la, k0, tid);
  sb.init(lb, n0, tid);

  f32x16 acc[TM][TN];
  zero_acc<TM, TN>(acc);
  const int m_begin = split * p.rows_per_split;
  gemm_mainloop<TM, TN, SA, SB, ActLoader, DyLoader, false, BMK, false, BN>(sa, sb, la, lb, As, Bs, m_begin,
                                                                           m_begin + p.rows_per_split, wm * (BMK / 2),
                                                                           wn * (BN / 2), acc, tid);
  const int l31 = lane & 31, h = lane >> 5;
  float* slab = p.slabs + (size_t)split * p.K * p.N;
#pragma unroll
  for (int tn = 0; tn < TN; ++tn) {
    const int col = n0 + wn * (BN / 2) + tn * 32 + l31;
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int k = k0 + wm * (BMK / 2) + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (k < p.K) slab[(size_t)k * p.N + col] = acc[tm][tn][r];
      }
    }
  }
}

bool act_ok(const t3d_act_src& a, int K) {
  return a.x != nullptr && (a.ldx % 4) == 0 && (a.coff % 4) == 0 && a.coff + (K + 3) / 4 * 4 <= a.ldx &&
         (a.scale == nullptr || a.shift != nullptr);
}
bool dy_ok(const t3d_dy_src& d) {
  return d.y != nullptr && d.coef != nullptr && (d.dz != nullptr || (d.argidx != nullptr && d.dpool != nullptr));
}

}  // namespace

extern "C" int t3d_pointmlp_fwd(const t3d_pointmlp_fwd_args* a, t3d_stream_t stream) {
  if (!a || !a->w || !a->y || !a->psum || !a->psumsq || !act_ok(a->a, a->K)) return T3D_ERR_ARG;
  if (a->pmax && (!a->pmin || !a->pamax || !a->pamin)) return T3D_ERR_ARG;
  if (a->M <= 0 || a->K <= 0 || a->N <= 0 || a->M % T3D_TILE_ROWS || a->rows_per_frustum % T3D_TILE_ROWS ||
      a->M % a->rows_per_frustum || a->N % 64)
    return T3D_ERR_SHAPE;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int tiles_m = a->M / 128;
  if (a->N % 128 == 0)
    T3D_LAUNCH(k_pointmlp_fwd<128>, dim3(tiles_m * (a->N / 128)), dim3(NT), 0, s, *a);
  else
    T3D_LAUNCH(k_pointmlp_fwd<64>, dim3(tiles_m * (a->N / 64)), dim3(NT), 0, s, *a);
  T3D_CHECK_LAUNCH();
  return T3D_OK;
}

extern "C" int t3d_pointmlp_dgrad(const t3d_pointmlp_dgrad_args* a, t3d_stream_t stream) {
  if (!a || !a->w || !a->out || !dy_ok(a->dy)) return T3D_ERR_ARG;
  if (a->prev_y && (!a->prev_scale || !a->prev_shift)) return T3D_ERR_ARG;
  if (a->psum_dz && (!a->psum_dzy || !a->prev_y)) return T3D_ERR_ARG;
  if (a->M <= 0 || a->M % T3D_TILE_ROWS || a->rows_per_frustum % T3D_TILE_ROWS || a->M % a->rows_per_frustum ||
      a->K % 64 || a->N % 4)
    return T3D_ERR_SHAPE;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int tiles_m = a->M / 128;
  if (a->K % 128 == 0)
    T3D_LAUNCH(k_pointmlp_dgrad<128>, dim3(tiles_m * (a->K / 128)), dim3(NT), 0, s, *a);
  else
    T3D_LAUNCH(k_pointmlp_dgrad<64>, dim3(tiles_m * (a->K / 64)), dim3(NT), 0, s, *a);
  T3D_CHECK_LAUNCH();
  return T3D_OK;
}

// Split/tile policy of the weight gradient, shared by the host (slab allocation) and the launcher.
// Goal: >= ~512 workgroups (2 per CU) without drowning HBM in partial slabs: the split count is capped so
// that a layer's slabs stay <= 2M floats (or 64 splits), and small K x N layers drop to 64-wide tiles to
// regain parallelism instead of splitting the rows finer.
extern "C" int t3d_wgrad_plan(int M, int K, int N, int* rows_per_split, int* tile_k, int* tile_n) {
  if (M <= 0 || K <= 0 || N <= 0 || N % 64 || M % 128 || !rows_per_split || !tile_k || !tile_n) return T3D_ERR_SHAPE;
  long cap = (1L << 21) / ((long)K * N);
  if (cap < 64) cap = 64;
  if (cap > M / 128) cap = M / 128;
  if (cap < 1) cap = 1;
  const int cand[4][2] = {{128, 128}, {128, 64}, {64, 128}, {64, 64}};
  int tk = 64, tn = 64;
  long tiles = (long)((K + 63) / 64) * (N / 64);
  for (int i = 0; i < 4; ++i) {
    const int ck = cand[i][0], cn = cand[i][1];
    if ((ck == 128 && K <= 64) || N % cn) continue;
    const long t = (long)((K + ck - 1) / ck) * (N / cn);
    if (t * cap >= 512) { tk = ck; tn = cn; tiles = t; break; }
  }
  long want = (512 + tiles - 1) / tiles;
  if (want > cap) want = cap;
  int s = 1;
  while ((long)s * 2 <= want && M % (s * 2) == 0 && (M / (s * 2)) % BK == 0) s *= 2;
  *rows_per_split = M / s;
  *tile_k = tk;
  *tile_n = tn;
  return T3D_OK;
}

extern "C" int t3d_pointmlp_wgrad(const t3d_pointmlp_wgrad_args* a, t3d_stream_t stream) {
  if (!a || !a->slabs || !act_ok(a->a, a->K) || !dy_ok(a->dy)) return T3D_ERR_ARG;
  if (a->M <= 0 || a->rows_per_split <= 0 || a->rows_per_split % BK || a->M % a->rows_per_split || a->N % 64 ||
      a->rows_per_frustum % BK || a->M % a->rows_per_frustum)
    return T3D_ERR_SHAPE;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int splits = a->M / a->rows_per_split;
  // tile choice: the plan's tile if the caller used t3d_wgrad_plan's split, else by shape
  int rps = 0, tk = 0, tn = 0;
  if (a->M % 128 == 0 && t3d_wgrad_plan(a->M, a->K, a->N, &rps, &tk, &tn) == T3D_OK && rps == a->rows_per_split) {
  } else {
    tk = a->K > 64 ? 128 : 64;
    tn = a->N % 128 == 0 ? 128 : 64;
  }
  const int tiles_k = (a->K + tk - 1) / tk, tiles_n = a->N / tn;
  const dim3 grid(tiles_k * tiles_n * splits);
  if (tk == 128 && tn == 128) T3D_LAUNCH((k_pointmlp_wgrad<128, 128>), grid, dim3(NT), 0, s, *a);
  else if (tk == 128) T3D_LAUNCH((k_pointmlp_wgrad<128, 64>), grid, dim3(NT), 0, s, *a);
  else if (tn == 128) T3D_LAUNCH((k_pointmlp_wgrad<64, 128>), grid, dim3(NT), 0, s, *a);
  else T3D_LAUNCH((k_pointmlp_wgrad<64, 64>), grid, dim3(NT), 0, s, *a);
  T3D_CHECK_LAUNCH();
  return T3D_OK;
}
