// Device bodies of the small kernels of the Gram-form backward of a max-pooled per-point layer (t3d.h K11e): the K x K
// matrices, the sparse argmax rows, the activation column sums and the final weight-gradient assembly.  Shared by the
// stand-alone kernels (poolbwd.hip) and the fused stage kernels (pointmlp.hip).  Every body takes its LDS as a pointer
// into the kernel's dynamic allocation and its block coordinates as arguments.  Everything is summed in a fixed order
// (no float atomics): the step stays bit-reproducible.
#pragma once
#include "common.h"

namespace {


// ---------------------------------------------------------------------------------------------
// P = w diag(c1) w^T, rowconst = w (c1*bias + c2), wc[n,k] = c0[n] w[k,n]
// grid (K/32, K/32, ceil(N/128)): block (bi,bj,ch) adds the columns n of chunk ch into its own slab of
// P[32bi.., 32bj..] (2x2 outputs per thread); t3d_reduce_slabs sums the chunks.  Blocks with bj == 0 also write
// their rowconst slab and their piece of wc.
// ---------------------------------------------------------------------------------------------
constexpr int PCH = 128;   // columns of w per chunk

constexpr size_t PREP_LDS = (2 * 32 * (PCH + 1) + 2 * PCH) * sizeof(float);

// tid_: thread index within a 256-thread logical block (default: the hardware block IS the logical block); the 512-thread bf16 stage
// kernels run two logical blocks side by side, every barrier below then simply spans both
__device__ __forceinline__ void pool_bwd_prep_body(const t3d_pool_bwd_prep_args& p, float* smem, int bi, int bj, int ch, int tid_ = -1) {
  float (*wi)[PCH + 1] = reinterpret_cast<float (*)[PCH + 1]>(smem);
  float (*wj)[PCH + 1] = reinterpret_cast<float (*)[PCH + 1]>(smem + 32 * (PCH + 1));
  float* v = smem + 2 * 32 * (PCH + 1);
  float* c0s = v + PCH;
  const int tid = tid_ < 0 ? (int)threadIdx.x : tid_, ti = tid >> 4, tj = tid & 15;
  const int n0 = ch * PCH;
  const bool edge = bj == 0;
  const float* c0 = p.coef;
  const float* c1 = p.coef + p.N;
  const float* c2 = p.coef + 2 * p.N;
  // N % 4 == 0: a float4 never straddles the end of a row; chunks past N are clamped and zeroed
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int f = tid + 256 * q;             // 32 rows x 32 float4
    const int r = f >> 5, c = (f & 31) * 4;
    const int n = min(n0 + c, p.N - 4);
    const bool ok = n0 + c < p.N;
    const float4 a = *reinterpret_cast<const float4*>(p.w + (size_t)(32 * bi + r) * p.N + n);
    const float4 b = *reinterpret_cast<const float4*>(p.w + (size_t)(32 * bj + r) * p.N + n);
    const float4 s1 = *reinterpret_cast<const float4*>(c1 + n);
    wi[r][c + 0] = ok ? a.x : 0.f; wi[r][c + 1] = ok ? a.y : 0.f; wi[r][c + 2] = ok ? a.z : 0.f; wi[r][c + 3] = ok ? a.w : 0.f;
    wj[r][c + 0] = ok ? b.x * s1.x : 0.f; wj[r][c + 1] = ok ? b.y * s1.y : 0.f;
    wj[r][c + 2] = ok ? b.z * s1.z : 0.f; wj[r][c + 3] = ok ? b.w * s1.w : 0.f;
  }
  if (tid < PCH) {
    const int n = min(n0 + tid, p.N - 1);
    const bool ok = n0 + tid < p.N;
    v[tid] = ok ? fmaf(p.bias ? p.bias[n] : 0.f, c1[n], c2[n]) : 0.f;
    c0s[tid] = ok ? c0[n] : 0.f;
  }
  __syncthreads();
  float a00 = 0.f, a01 = 0.f, a10 = 0.f, a11 = 0.f;
#pragma unroll 8
  for (int c = 0; c < PCH; ++c) {
    const float x0 = wi[ti][c], x1 = wi[ti + 16][c], y0 = wj[tj][c], y1 = wj[tj + 16][c];
    a00 = fmaf(x0, y0, a00); a01 = fmaf(x0, y1, a01);
    a10 = fmaf(x1, y0, a10); a11 = fmaf(x1, y1, a11);
  }
  float* ps = p.p_slabs + (size_t)ch * p.K * p.K;
  ps[(size_t)(32 * bi + ti) * p.K + 32 * bj + tj] = a00;
  ps[(size_t)(32 * bi + ti) * p.K + 32 * bj + tj + 16] = a01;
  ps[(size_t)(32 * bi + ti + 16) * p.K + 32 * bj + tj] = a10;
  ps[(size_t)(32 * bi + ti + 16) * p.K + 32 * bj + tj + 16] = a11;
  if (edge) {
    // rowconst: thread = (row r, residue q of c mod 8); the 8 lanes of one row are adjacent in a wave
    const int r = tid >> 3, q = tid & 7;
    float rc = 0.f;
#pragma unroll
    for (int c = q; c < PCH; c += 8) rc = fmaf(wi[r][c], v[c], rc);
    rc += __shfl_xor(rc, 4, 64);
    rc += __shfl_xor(rc, 2, 64);
    rc += __shfl_xor(rc, 1, 64);
    if (q == 0) p.rc_slabs[(size_t)ch * p.K + 32 * bi + r] = rc;
    // transposed, c0-scaled copy: 32 consecutive k per column n
    const int k = tid & 31;
    if (p.wc != nullptr)
      for (int c = tid >> 5; c < PCH; c += 8)
        if (n0 + c < p.N) p.wc[(size_t)(n0 + c) * p.K + 32 * bi + k] = c0s[c] * wi[k][c];
  }
}

// ---------------------------------------------------------------------------------------------
// S[m,:] = sum_{n : argidx[b,n] == m - b*rpf} dpool[b,n] * wc[n,:]
// grid (M/128, K/128); one workgroup = one 128-row tile x 128 columns, accumulated in LDS.  Wave w owns the rows
// r = w (mod 4): it lists its hits in ascending n (ballot compaction), then adds them one after the other, so every
// row is summed in ascending n whatever the hit pattern; the wc rows of UNROLL hits are in flight together.
// ---------------------------------------------------------------------------------------------
constexpr int SR_KC = 128;
constexpr int SR_UNROLL = 16;
constexpr int SR_PRE = 16;     // argidx chunks (of 64) loaded together

__device__ __forceinline__ void pool_sparse_rows_body(const t3d_pool_sparse_rows_args& p, float* smem, int bx, int by) {
  float* tile = smem;                                            // [128][SR_KC]
  int* lists = reinterpret_cast<int*>(smem + 128 * SR_KC);        // [4][N]
  int* hitrow = lists + 4 * p.N;                                  // [128] row received a hit
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int row0 = bx * 128, kc0 = by * SR_KC;
  const int b = row0 / p.rows_per_frustum, rin0 = row0 - b * p.rows_per_frustum;
  for (int f = tid; f < 128 * SR_KC / 4; f += 256) reinterpret_cast<float4*>(tile)[f] = make_float4(0.f, 0.f, 0.f, 0.f);
  if (tid < 128) hitrow[tid] = 0;
  int* mine = lists + w * p.N;
  int cnt = 0;
#ifndef T3D_ABL_SR_NOSCAN      // diagnostic builds (tools/bench_sparse_rows.py): the kernel without its scan / without its add loop
  for (int nb = 0; nb < p.N; nb += 64 * SR_PRE) {
    int rr[SR_PRE];                      // all argidx loads of the block in flight before the first ballot
#pragma unroll
    for (int q = 0; q < SR_PRE; ++q) {
      const int n = nb + 64 * q + lane;
      rr[q] = p.argidx[(size_t)b * p.N + min(n, p.N - 1)];
    }
#pragma unroll
    for (int q = 0; q < SR_PRE; ++q) {
      const int n = nb + 64 * q + lane;
      const int r = rr[q] - rin0;
      const bool hit = n < p.N && r >= 0 && r < 128 && (r & 3) == w;
      const unsigned long long m = __ballot(hit);
      if (hit) mine[cnt + __popcll(m & ((1ull << lane) - 1ull))] = n | (r << 16);
      cnt += __popcll(m);
    }
  }
#endif
  __syncthreads();
#ifdef T3D_ABL_SR_NOADD
  if (cnt > 0 && lane == 0) hitrow[mine[0] >> 16] = 1;
  cnt = 0;
#endif
  for (int i0 = 0; i0 < cnt; i0 += SR_UNROLL) {
    float2 wv[SR_UNROLL];
    float g[SR_UNROLL];
    int rr[SR_UNROLL], en[SR_UNROLL];
    // Round 3: the batch's list entries are read FIRST, all of them, then the global loads go out, then the hit flags are stored.
    // With the flag store between two list reads (it may alias the list as far as the compiler knows) every entry was an LDS
    // round trip of its own in front of its loads -- 16 serial ds_read + wait per batch -- and the guarded dpool load an
    // exec-masked region per hit.  Indices past the list's end repeat its last entry (valid addresses); their g is masked to 0.
#pragma unroll
    for (int u = 0; u < SR_UNROLL; ++u) en[u] = mine[min(i0 + u, cnt - 1)];
#pragma unroll
    for (int u = 0; u < SR_UNROLL; ++u) {
      const int n = en[u] & 0xffff;
      rr[u] = en[u] >> 16;
      g[u] = p.dpool[(size_t)b * p.N + n];
      wv[u] = *reinterpret_cast<const float2*>(p.wc + (size_t)n * p.K + kc0 + 2 * lane);
    }
#pragma unroll
    for (int u = 0; u < SR_UNROLL; ++u) {
      hitrow[rr[u]] = 1;                // same value from every lane; rows of this wave only
      g[u] = (i0 + u < cnt) ? g[u] : 0.f;
    }
    // (two hits per LDS round trip -- both rows read together, a same-row pair summed in registers behind a scalar branch -- is
    // bit-identical and measured SLOWER: 26.4 / 19.6 vs 24.1 / 17.1 us for the box / seg layers' launches)
#pragma unroll
    for (int u = 0; u < SR_UNROLL; ++u) {
      float2* t = reinterpret_cast<float2*>(tile + rr[u] * SR_KC + 2 * lane);
      float2 cur = *t;
      cur.x = fmaf(g[u], wv[u].x, cur.x);
      cur.y = fmaf(g[u], wv[u].y, cur.y);
      *t = cur;
    }
  }
  __syncthreads();
  const bool gated = p.row_live != nullptr;                // rows without a hit stay unwritten: the reader consults row_live
  if (gated && by == 0 && tid < 128) p.row_live[row0 + tid] = hitrow[tid];
  for (int f = tid; f < 128 * SR_KC / 4; f += 256) {
    const int r = f / (SR_KC / 4), c4 = f % (SR_KC / 4);
    if (gated && !hitrow[r]) continue;
    *reinterpret_cast<float4*>(p.s + (size_t)(row0 + r) * p.K + kc0 + 4 * c4) = reinterpret_cast<const float4*>(tile)[f];
  }
}

// ---------------------------------------------------------------------------------------------
// part[t,k] = sum of act(a)[m,k] over the 128 rows of tile t.  One workgroup per tile.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float act_elem(const t3d_act_src& s, size_t row, int k) {
  float v = ld_elem(s.x, row * s.ldx + s.coff + k, s.dtype);
  if (s.scale) v = fmaf(v, s.scale[k], s.shift[k]);
  if (s.relu) v = fmaxf(v, 0.f);
  return v;
}

constexpr size_t COLSUM_LDS = 256 * sizeof(float4);

template <class XT = float>      // XT: element type of the activation source (fp32 path: float, the round-1 code)
__device__ __forceinline__ void act_colsum_body(const t3d_act_colsum_args& p, float* smem, int bx) {
  float4* red = reinterpret_cast<float4*>(smem);
  const int tid = threadIdx.x;
  const int row0 = bx * 128;
  // thread = (row group, float4 column chunk): K/4 chunks, 1024/K groups of K/8 rows, all loads of a thread in flight
  const int chunks = p.K / 4, c4 = tid % chunks, grp = tid / chunks, groups = 256 / chunks, rows = 128 / groups;
  float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = make_float4(0.f, 0.f, 0.f, 0.f);
  if (p.a.scale) {
    sc = *reinterpret_cast<const float4*>(p.a.scale + 4 * c4);
    sh = *reinterpret_cast<const float4*>(p.a.shift + 4 * c4);
  }
  const float floor_ = p.a.relu ? 0.f : -INFINITY;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  const size_t base = (size_t)(row0 + grp * rows) * p.a.ldx + p.a.coff + 4 * c4;       // element offset (fp32 or bf16 source)
  // Round 3: explicit batches of 16 loads (rows is 8 ... 64, a multiple of 8).  The `#pragma unroll 16` loop it replaces compiled
  // into load -> s_waitcnt vmcnt(0) -> add, sixteen times over (one destination register quad re-used): 16-32 SERIAL round trips
  // per workgroup inside every stage-1 launch.  Same sums in the same order.
  constexpr int CSB = 16;
  for (int r0 = 0; r0 < rows; r0 += CSB) {
    typename Elem<XT>::V4 raw[CSB];
#pragma unroll
    for (int u = 0; u < CSB; ++u) raw[u] = Elem<XT>::ld4(p.a.x, base + (size_t)min(r0 + u, rows - 1) * p.a.ldx);
#pragma unroll
    for (int u = 0; u < CSB; ++u) {
      if (r0 + u < rows) {                     // workgroup-uniform
        const float4 x = Elem<XT>::widen(raw[u]);
        acc.x += fmaxf(fmaf(x.x, sc.x, sh.x), floor_);
        acc.y += fmaxf(fmaf(x.y, sc.y, sh.y), floor_);
        acc.z += fmaxf(fmaf(x.z, sc.z, sh.z), floor_);
        acc.w += fmaxf(fmaf(x.w, sc.w, sh.w), floor_);
      }
    }
  }
  red[tid] = acc;
  __syncthreads();
  if (tid < chunks) {
    float4 t = red[tid];
    for (int g = 1; g < groups; ++g) {
      const float4 u = red[g * chunks + tid];
      t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w;
    }
    *reinterpret_cast<float4*>(p.part + (size_t)bx * p.K + 4 * tid) = t;
  }
}

// ---------------------------------------------------------------------------------------------
// dw[k,n] = c1[n]*(sum_j G[k,j] w[j,n] + abar[k] bias[n]) + abar[k] c2[n] + c0[n] sum_b dpool[b,n] a[b*rpf+argidx[b,n], k]
// grid (K/32, N/16): one workgroup = a 32 x 16 block of dw.
//   gather part : thread (k, q) sums, in ascending b, the argmax rows of the columns n = q and q+8 (a 128-byte segment
//                 of every gathered row is read by the 32 threads k); results cross to the GEMM mapping through LDS
//   G.w part    : G[32 rows][K] and w[K][16] staged in LDS, thread (ti,tj) owns dw[ti | ti+16][tj]
// ---------------------------------------------------------------------------------------------
constexpr int FK = 32, FN = 16, FB = 32;   // block of k, block of n, frustums per gather pass

inline size_t finish_lds(int K) {
  return ((size_t)FK * (K + 1) + (size_t)K * FN + FK * (FN + 1) + 2 * FB * FN) * sizeof(float);
}

template <class XT = float>
__device__ __forceinline__ void pool_wgrad_finish_body(const t3d_pool_wgrad_finish_args& p, float* smem, int bx, int by, int tid_ = -1) {
  float* gsm = smem;                               // [FK][K+1]
  float* wsm = gsm + FK * (p.K + 1);               // [K][FN]
  float* gat = wsm + p.K * FN;                     // [FK][FN+1]
  float* dps = gat + FK * (FN + 1);                // [FB][FN]
  int* ais = reinterpret_cast<int*>(dps + FB * FN);// [FB][FN]
  const int tid = tid_ < 0 ? (int)threadIdx.x : tid_;
  const int k0 = bx * FK, n0 = by * FN;
  const int ldg = p.K + 1;
  // stage G rows and w columns (coalesced along the fast index)
  for (int f = tid; f < FK * p.K; f += 256) gsm[(f / p.K) * ldg + f % p.K] = p.g[(size_t)(k0 + f / p.K) * p.K + f % p.K];
  for (int f = tid; f < p.K * FN; f += 256) wsm[f] = p.w[(size_t)(f / FN) * p.N + n0 + f % FN];
  // ---- gather ----
  {
    const int k = tid & 31, q = tid >> 5;
    const float sc = p.a.scale ? p.a.scale[k0 + k] : 1.f, sh = p.a.scale ? p.a.shift[k0 + k] : 0.f;
    const float floor_ = p.a.relu ? 0.f : -INFINITY;
    const size_t xk = (size_t)(p.a.coff + k0 + k);
    float g0 = 0.f, g1 = 0.f;
    for (int b0 = 0; b0 < p.B; b0 += FB) {
      __syncthreads();
      for (int f = tid; f < FB * FN; f += 256) {
        const int b = b0 + f / FN;
        const bool ok = b < p.B;
        const size_t o = (size_t)min(b, p.B - 1) * p.N + n0 + f % FN;
        const int ai = p.argidx[o];
        const float dpv = p.dpool[o];                    // unconditional (o is a valid address): in flight with argidx, not behind it
        dps[f] = (ok && ai >= 0) ? dpv : 0.f;
        ais[f] = max(ai, 0);
      }
      __syncthreads();
      float x0[FB], x1[FB];
#pragma unroll
      for (int bb = 0; bb < FB; ++bb) {              // 64 independent loads in flight
        const size_t rowbase = (size_t)min(b0 + bb, p.B - 1) * p.rows_per_frustum;
        x0[bb] = Elem<XT>::ld1(p.a.x, xk + (rowbase + ais[bb * FN + q]) * p.a.ldx);
        x1[bb] = Elem<XT>::ld1(p.a.x, xk + (rowbase + ais[bb * FN + q + 8]) * p.a.ldx);
      }
#pragma unroll
      for (int bb = 0; bb < FB; ++bb) {
        g0 = fmaf(dps[bb * FN + q], fmaxf(fmaf(x0[bb], sc, sh), floor_), g0);
        g1 = fmaf(dps[bb * FN + q + 8], fmaxf(fmaf(x1[bb], sc, sh), floor_), g1);
      }
    }
    gat[k * (FN + 1) + q] = g0;
    gat[k * (FN + 1) + q + 8] = g1;
  }
  __syncthreads();
  // ---- G.w and assembly ----
  const int ti = tid >> 4, tj = tid & 15;
  float a0 = 0.f, a1 = 0.f;
#pragma unroll 8
  for (int j = 0; j < p.K; ++j) {
    const float wv = wsm[j * FN + tj];
    a0 = fmaf(gsm[ti * ldg + j], wv, a0);
    a1 = fmaf(gsm[(ti + 16) * ldg + j], wv, a1);
  }
  const int n = n0 + tj;
  const float c0 = p.coef[n], c1 = p.coef[p.N + n], c2 = p.coef[2 * p.N + n];
  const float bias = p.bias ? p.bias[n] : 0.f;
  const float ab0 = p.abar[k0 + ti], ab1 = p.abar[k0 + ti + 16];
  p.dw[(size_t)(k0 + ti) * p.N + n] = fmaf(c0, gat[ti * (FN + 1) + tj], fmaf(c1, fmaf(ab0, bias, a0), ab0 * c2));
  p.dw[(size_t)(k0 + ti + 16) * p.N + n] = fmaf(c0, gat[(ti + 16) * (FN + 1) + tj], fmaf(c1, fmaf(ab1, bias, a1), ab1 * c2));
}



// ---- host-side argument checks, shared by the stand-alone and the fused launchers ----------------------------
inline bool pool_act_ok(const t3d_act_src& a, int K) {
  return a.x != nullptr && a.coff + K <= a.ldx && (a.scale == nullptr || a.shift != nullptr) && a.sub == nullptr;
}
inline int check_prep(const t3d_pool_bwd_prep_args* a) {
  if (!a || !a->w || !a->coef || !a->p_slabs || !a->rc_slabs) return T3D_ERR_ARG;
  if (a->K <= 0 || a->K % 32 || a->N <= 0 || a->N % 4) return T3D_ERR_SHAPE;
  return T3D_OK;
}
inline size_t sparse_rows_lds(int N) { return (size_t)128 * SR_KC * sizeof(float) + ((size_t)4 * N + 128) * sizeof(int); }
inline int check_sparse_rows(const t3d_pool_sparse_rows_args* a) {
  if (!a || !a->argidx || !a->dpool || !a->wc || !a->s) return T3D_ERR_ARG;
  if (a->B <= 0 || a->N <= 0 || a->N > 65535 || a->K % SR_KC || a->rows_per_frustum % T3D_TILE_ROWS ||
      a->rows_per_frustum > 32767 || sparse_rows_lds(a->N) > 160 * 1024)
    return T3D_ERR_SHAPE;
  return T3D_OK;
}
inline int check_colsum(const t3d_act_colsum_args* a) {
  if (!a || !a->part || !pool_act_ok(a->a, a->K)) return T3D_ERR_ARG;
  if (a->M <= 0 || a->M % T3D_TILE_ROWS || (a->K != 64 && a->K != 128 && a->K != 256) || a->a.ldx % 4 || a->a.coff % 4)
    return T3D_ERR_SHAPE;
  return T3D_OK;
}
inline int check_finish(const t3d_pool_wgrad_finish_args* a) {
  if (!a || !a->argidx || !a->dpool || !a->coef || !a->w || !a->g || !a->abar || !a->dw || !pool_act_ok(a->a, a->K))
    return T3D_ERR_ARG;
  if (a->K <= 0 || a->K % FK || a->K > 256 || a->N <= 0 || a->N % FN || a->B <= 0) return T3D_ERR_SHAPE;
  return T3D_OK;
}

}  // namespace
