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
#include "poolbwd_dev.h"
#include "rider_dev.h"

#include <cstdlib>
#include <mutex>
#include <type_traits>
#include <utility>
#include <unordered_map>

namespace {

#ifndef T3D_BK
#define T3D_BK 32
#endif
constexpr int BK = T3D_BK;      // reduction depth of one LDS stage
#ifndef T3D_WAVES
#define T3D_WAVES 2
#endif
#ifndef T3D_FORCE_TILE
#define T3D_FORCE_TILE 0       // diagnostic: 64 / 128 forces the column tile of fwd and dgrad
#endif
constexpr int LDR = BK + 4;
constexpr int NT = 256;
// prefetch distance (register slots) per kernel family.  Distance 2 was measured on MI355X (64-column kernels and
// dgrad_gram<128>, B=32 N=1024 step): no gain (1.843 ms vs 1.822 ms per step) at 30-90 more VGPRs -- the narrow kernels
// are bound by their fixed per-workgroup latency (first load, epilogue), not by the per-tile load latency -- so the
// default stays 1; the variant is kept for other shapes.
#ifndef T3D_PF_NARROW
#define T3D_PF_NARROW 1      // 64-column tiles
#endif
#ifndef T3D_PF_WIDE
#define T3D_PF_WIDE 1        // 128-column tiles of fwd / dgrad / wgrad (213-233 VGPRs already)
#endif
#ifndef T3D_PF_GRAM128
#define T3D_PF_GRAM128 1     // dgrad_gram<128>
#endif

__device__ __forceinline__ float4 f4zero() { return make_float4(0.f, 0.f, 0.f, 0.f); }

// Keep flags of a lane's 32 accumulator rows (row = base + tm * 32 + (r & 3) + 8 * (r >> 2), bit tm * 16 + r) from a per-row mask:
// eight 16-byte loads requested together, then the compares.  Written as 32 scalar loads OR-ed into one word, hipcc formed the
// same eight loads but consumed each before requesting the next (one register quad, `s_waitcnt vmcnt(0)` eight times): eight
// serial round trips in the prologue of every pooled forward (round 3, device assembly of k_pointmlp_fwd_pool).
__device__ __forceinline__ unsigned keep_bits32(const float* __restrict__ rowmask, int base) {
  float4 m[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) m[q] = *reinterpret_cast<const float4*>(rowmask + base + (q >> 2) * 32 + (q & 3) * 8);
  unsigned bits = 0u;
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const unsigned nib = (m[q].x != 0.f ? 1u : 0u) | (m[q].y != 0.f ? 2u : 0u) | (m[q].z != 0.f ? 4u : 0u) | (m[q].w != 0.f ? 8u : 0u);
    bits |= nib << ((q >> 2) * 16 + (q & 3) * 4);
  }
  return bits;
}

// Diagnostic builds (-DT3D_TRACE, tools/trace_blocks.py): every workgroup records the 100 MHz wall clock at kernel entry,
// after the main loop and at exit, plus its XCC / HW id, into a buffer installed with t3d_set_trace().
#ifdef T3D_TRACE
#ifndef T3D_TRACE_STRIDE
#define T3D_TRACE_STRIDE 4      // 8: slot 4 = the x3 main loop's prologue done (first k-tile staged), tools/trace_blocks.py T3D_TRACE_STRIDE=8
#endif
__device__ unsigned long long* t3d_trace_ptr = nullptr;
#define T3D_TRACE_MARK(slot)                                                                                  \
  do {                                                                                                        \
    if (threadIdx.x == 0 && t3d_trace_ptr) {                                                                  \
      t3d_trace_ptr[(size_t)blockIdx.x * T3D_TRACE_STRIDE + (slot)] = wall_clock64();                         \
      if ((slot) == 0)                                                                                        \
        t3d_trace_ptr[(size_t)blockIdx.x * T3D_TRACE_STRIDE + 3] =                                            \
            ((unsigned long long)__builtin_amdgcn_s_getreg((20) | (0 << 6) | ((4 - 1) << 11)) << 32) |        \
            (unsigned long long)__builtin_amdgcn_s_getreg((4) | (0 << 6) | ((32 - 1) << 11));                 \
    }                                                                                                         \
  } while (0)
// eight marks per workgroup (tools/trace_fwd_res.py): the phases of the FIRST row tile of the resident bf16 forward
#define T3D_TRACE_MARK8(slot)                                                                                 \
  do {                                                                                                        \
    if (threadIdx.x == 0 && t3d_trace_ptr) t3d_trace_ptr[(size_t)blockIdx.x * 8 + (slot)] = wall_clock64();  \
  } while (0)
#else
#define T3D_TRACE_MARK(slot) do {} while (0)
#define T3D_TRACE_MARK8(slot) do {} while (0)
#endif

// ---------------------------------------------------------------------------------------------
// loaders: fetch() issues the global loads, xform() does the fused element-wise math afterwards
// ---------------------------------------------------------------------------------------------
// All loads are UNCONDITIONAL on clamped addresses and masked afterwards with selects: a per-lane
// `cond ? load : 0` compiles to an exec-masked branch with its own s_waitcnt, which serialises the global
// latency of every k-tile (seen in the first version's ISA).  Only wave-uniform conditions branch.
// fp32 loaders: the code of round 1, textually apart from the typed loaders of the bf16 path below (routing fp32 through the
// typed templates changed the compiler's addressing / schedule of the fp32 main loops: k_pointmlp_bwd<64,64,64> 30.0 -> 35.0 us
// on one box, same-box A/B)
template <bool HAS_SUB>
struct ActLoader {
  static constexpr bool EXACT = false;
  __device__ __forceinline__ static void pin(float4& x) { asm volatile("" : "+v"(x.x), "+v"(x.y), "+v"(x.z), "+v"(x.w)); }
  t3d_act_src s;
  int K;     // valid columns
  int rpf;   // rows per frustum
  struct Raw { float4 x; };
  __device__ __forceinline__ static void pin(Raw& r) { pin(r.x); }
  struct Coef { float4 sc, sh; };
  __device__ __forceinline__ Coef fetch_coef(int col) const {
    Coef c;
    c.sc = make_float4(1.f, 1.f, 1.f, 1.f);
    c.sh = f4zero();
    if (s.scale != nullptr) {                 // uniform; scale/shift hold >= roundup4(K) floats (host contract)
      const int cc = min(col, ((K + 3) & ~3) - 4);
      c.sc = *reinterpret_cast<const float4*>(s.scale + cc);
      c.sh = *reinterpret_cast<const float4*>(s.shift + cc);
    }
    return c;
  }
  __device__ __forceinline__ Raw fetch(int row, int col) const {
    Raw r;
    const int cc = min(col, ((K + 3) & ~3) - 4);
    r.x = *reinterpret_cast<const float4*>(s.x + (size_t)row * s.ldx + s.coff + cc);
    return r;
  }
  // straight-line (no branches): identity scale/shift when there is no batch-norm, ReLU floor -inf when off
  __device__ __forceinline__ float4 xform(const Raw& r, const Coef& c, int row, int col) const {
    float v[4] = {r.x.x, r.x.y, r.x.z, r.x.w};
    const float sc[4] = {c.sc.x, c.sc.y, c.sc.z, c.sc.w};
    const float sh[4] = {c.sh.x, c.sh.y, c.sh.z, c.sh.w};
    const float floor_ = s.relu ? 0.f : -INFINITY;
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = fmaxf(fmaf(v[e], sc[e], sh[e]), floor_);
    if (HAS_SUB) {                            // raw point inputs only (K <= 4)
      const int b = row / rpf;
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] -= s.sub[(size_t)b * s.sub_ld + min(col + e, K - 1)];
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = (col + e < K) ? v[e] : 0.f;
    return make_float4(v[0], v[1], v[2], v[3]);
  }
};

template <bool POOLED>
struct DyLoader {      // N % 32 == 0: every tile column is valid
  t3d_dy_src s;
  int N;
  int rpf;
  struct Raw { float4 dz, y; };
  struct Coef { float4 c0, c1, c2; };
  __device__ __forceinline__ Coef fetch_coef(int col) const {
    Coef c;
    c.c0 = *reinterpret_cast<const float4*>(s.coef + col);
    c.c1 = *reinterpret_cast<const float4*>(s.coef + N + col);
    c.c2 = *reinterpret_cast<const float4*>(s.coef + 2 * N + col);
    return c;
  }
  __device__ __forceinline__ Raw fetch(int row, int col) const {
    Raw r;
    r.y = *reinterpret_cast<const float4*>(s.y + (size_t)row * N + col);
    if (!POOLED) {
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
  // Uniform base + 32-bit per-lane offset (x3_iter_il): the address of a load is (a scalar 64-bit pointer, advanced per k-tile with
  // scalar instructions) + (a per-lane BYTE offset fixed for the whole kernel) -- the `saddr` form of global_load, no vector
  // instruction per load; fetch() costs a 64-bit multiply-add and a 64-bit add on the vector ALU per load.
  static constexpr bool HAS_AT = !POOLED;
  __device__ __forceinline__ static void pin(Raw& r) {
    asm volatile("" : "+v"(r.dz.x), "+v"(r.dz.y), "+v"(r.dz.z), "+v"(r.dz.w), "+v"(r.y.x), "+v"(r.y.y), "+v"(r.y.z), "+v"(r.y.w));
  }
  __device__ __forceinline__ int ld_elems() const { return N; }
  __device__ __forceinline__ const float* base_ptr() const { return s.y; }
  __device__ __forceinline__ Raw fetch_at(size_t uoff, unsigned lbytes) const {
    Raw r;
    r.y = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(s.y + uoff) + lbytes);
    r.dz = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(s.dz + uoff) + lbytes);
    return r;
  }
  __device__ __forceinline__ Coef fetch_coef_at(int ucol, unsigned lbytes) const {
    Coef c;
    c.c0 = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(s.coef + ucol) + lbytes);
    c.c1 = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(s.coef + N + ucol) + lbytes);
    c.c2 = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(s.coef + 2 * N + ucol) + lbytes);
    return c;
  }
  // The per-channel coefficients of the WHOLE reduction range as a table in LDS (T3D_X3_COEF_LDS, gemm_mainloop_x3): a k-tile's
  // coefficients then cost two or three ds_read_b128 instead of as many global loads -- each of which holds the wave ~50-60 cycles
  // beside MFMAs for 128 bytes of data every lane group shares (round 6: the memory INSTRUCTIONS bound these loops).
  static constexpr bool HAS_CTAB = !POOLED;
  __device__ __forceinline__ int ctab_floats() const { return 3 * N; }
  __device__ __forceinline__ void ctab_fill(float* tab, int tid, int nthreads) const {
    for (int i = tid * 4; i < 3 * N; i += nthreads * 4) *reinterpret_cast<float4*>(tab + i) = *reinterpret_cast<const float4*>(s.coef + i);
  }
  __device__ __forceinline__ Coef ctab_coef(const float* tab, int col) const {
    Coef c;
    c.c0 = *reinterpret_cast<const float4*>(tab + col);
    c.c1 = *reinterpret_cast<const float4*>(tab + N + col);
    c.c2 = *reinterpret_cast<const float4*>(tab + 2 * N + col);
    return c;
  }
};

// typed loaders (T3D_BF16 path)
template <bool HAS_SUB, class XT>      // XT: element type of the source tensor (t3d_act_src.dtype)
struct ActLoaderT {
  t3d_act_src s;
  int K;     // valid columns
  int rpf;   // rows per frustum
  struct Raw { typename Elem<XT>::V4 x; };
  struct Coef { float4 sc, sh; };
  // a bf16 source is a layer output: K is a multiple of the 64-deep k-tile (launcher-checked), so no column is clamped or masked
  static constexpr bool EXACT = Elem<XT>::BF16;
  static constexpr bool HAS8 = Elem<XT>::BF16 && !HAS_SUB;      // 16-byte granules (StagerH8): two adjacent Raw from one load
  __device__ __forceinline__ void fetch8(int row, int col, Raw& lo, Raw& hi) const {
    const bf16x8 v = *reinterpret_cast<const bf16x8*>(reinterpret_cast<const bf16_t*>(s.x) + (size_t)row * s.ldx + s.coff + col);
    lo.x = __builtin_shufflevector(v, v, 0, 1, 2, 3);
    hi.x = __builtin_shufflevector(v, v, 4, 5, 6, 7);
  }
  __device__ __forceinline__ Coef fetch_coef(int col) const {
    Coef c;
    c.sc = make_float4(1.f, 1.f, 1.f, 1.f);
    c.sh = f4zero();
    if (s.scale != nullptr) {                 // uniform; scale/shift hold >= roundup4(K) floats (host contract)
      const int cc = EXACT ? col : min(col, ((K + 3) & ~3) - 4);
      c.sc = *reinterpret_cast<const float4*>(s.scale + cc);
      c.sh = *reinterpret_cast<const float4*>(s.shift + cc);
    }
    return c;
  }
  __device__ __forceinline__ Raw fetch(int row, int col) const {
    Raw r;
    const int cc = EXACT ? col : min(col, ((K + 3) & ~3) - 4);
    r.x = Elem<XT>::ld4(s.x, (size_t)row * s.ldx + s.coff + cc);
    return r;
  }
  // straight-line (no branches): identity scale/shift when there is no batch-norm, ReLU floor -inf when off
  __device__ __forceinline__ float4 xform(const Raw& r, const Coef& c, int row, int col) const {
    const float4 rx = Elem<XT>::widen(r.x);
    float v[4] = {rx.x, rx.y, rx.z, rx.w};
    const float sc[4] = {c.sc.x, c.sc.y, c.sc.z, c.sc.w};
    const float sh[4] = {c.sh.x, c.sh.y, c.sh.z, c.sh.w};
    const float floor_ = s.relu ? 0.f : -INFINITY;
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = fmaxf(fmaf(v[e], sc[e], sh[e]), floor_);
    if (HAS_SUB) {                            // raw point inputs only (K <= 4)
      const int b = row / rpf;
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] -= s.sub[(size_t)b * s.sub_ld + min(col + e, K - 1)];
    }
    if (!EXACT) {
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = (col + e < K) ? v[e] : 0.f;
    }
    return make_float4(v[0], v[1], v[2], v[3]);
  }
};

template <bool POOLED, class T>      // T: element type of dz and y (t3d_dy_src.dtype); the pooled-sparse form is fp32 only
struct DyLoaderT {      // N % 32 == 0: every tile column is valid
  t3d_dy_src s;
  int N;
  int rpf;
  struct Raw { typename Elem<T>::V4 dz, y; };
  struct Coef { float4 c0, c1, c2; };
  static constexpr bool HAS8 = Elem<T>::BF16 && !POOLED;
  __device__ __forceinline__ void fetch8(int row, int col, Raw& lo, Raw& hi) const {
    const size_t o = (size_t)row * N + col;
    const bf16x8 a = *reinterpret_cast<const bf16x8*>(reinterpret_cast<const bf16_t*>(s.y) + o);
    const bf16x8 b = *reinterpret_cast<const bf16x8*>(reinterpret_cast<const bf16_t*>(s.dz) + o);
    lo.y = __builtin_shufflevector(a, a, 0, 1, 2, 3); hi.y = __builtin_shufflevector(a, a, 4, 5, 6, 7);
    lo.dz = __builtin_shufflevector(b, b, 0, 1, 2, 3); hi.dz = __builtin_shufflevector(b, b, 4, 5, 6, 7);
  }
  __device__ __forceinline__ Coef fetch_coef(int col) const {
    Coef c;
    c.c0 = *reinterpret_cast<const float4*>(s.coef + col);
    c.c1 = *reinterpret_cast<const float4*>(s.coef + N + col);
    c.c2 = *reinterpret_cast<const float4*>(s.coef + 2 * N + col);
    return c;
  }
  __device__ __forceinline__ Raw fetch(int row, int col) const {
    Raw r;
    r.y = Elem<T>::ld4(s.y, (size_t)row * N + col);
    if constexpr (!POOLED) {
      r.dz = Elem<T>::ld4(s.dz, (size_t)row * N + col);
    } else {
      static_assert(!Elem<T>::BF16, "pooled-sparse dy is an fp32 form (bf16 layers take the Gram path)");
      const int b = row / rpf, rin = row - b * rpf;
      const int4 a = *reinterpret_cast<const int4*>(s.argidx + (size_t)b * N + col);
      const float4 g = *reinterpret_cast<const float4*>(s.dpool + (size_t)b * N + col);
      r.dz = make_float4(a.x == rin ? g.x : 0.f, a.y == rin ? g.y : 0.f, a.z == rin ? g.z : 0.f,
                         a.w == rin ? g.w : 0.f);
    }
    return r;
  }
  __device__ __forceinline__ float4 xform(const Raw& r, const Coef& c, int, int) const {
    const float4 dz = Elem<T>::widen(r.dz), y = Elem<T>::widen(r.y);
    return make_float4(fmaf(c.c0.x, dz.x, fmaf(c.c1.x, y.x, c.c2.x)), fmaf(c.c0.y, dz.y, fmaf(c.c1.y, y.y, c.c2.y)),
                       fmaf(c.c0.z, dz.z, fmaf(c.c1.z, y.z, c.c2.z)), fmaf(c.c0.w, dz.w, fmaf(c.c1.w, y.w, c.c2.w)));
  }
};

// WT: element type of the matrix (bf16 path: the optimiser's bf16 copy of the weights).  EXACT: every tile lies inside the matrix
// (no clamped address, no mask); a bf16 EXACT tile is copied into the LDS image as it is (PASS_BF16).
template <class WT = float, bool EXACT_ = false>
struct WLoaderT {
  const float* w;
  int ld;
  int rows, cols;   // valid extent; cols % 4 == 0
  static constexpr bool PASS_BF16 = EXACT_ && Elem<WT>::BF16;
  static constexpr bool HAS8 = EXACT_ && Elem<WT>::BF16;
  struct Raw { typename Elem<WT>::V4 x; };
  struct Coef {};
  __device__ __forceinline__ Coef fetch_coef(int) const { return Coef(); }
  __device__ __forceinline__ void fetch8(int row, int col, Raw& lo, Raw& hi) const {      // HAS8 only
    if constexpr (Elem<WT>::BF16) {
      const bf16x8 v = *reinterpret_cast<const bf16x8*>(reinterpret_cast<const bf16_t*>(w) + (size_t)row * ld + col);
      lo.x = __builtin_shufflevector(v, v, 0, 1, 2, 3);
      hi.x = __builtin_shufflevector(v, v, 4, 5, 6, 7);
    }
  }
  __device__ __forceinline__ Raw fetch(int row, int col) const {
    Raw r;
    if (EXACT_) r.x = Elem<WT>::ld4(w, (size_t)row * ld + col);
    else r.x = Elem<WT>::ld4(w, (size_t)min(row, rows - 1) * ld + min(col, cols - 4));
    return r;
  }
  __device__ __forceinline__ float4 xform(const Raw& r, const Coef&, int row, int col) const {
    const bool ok = EXACT_ || (row < rows && col < cols);      // masked AFTER the MFMA phase, never right behind the load
    const float4 x = Elem<WT>::widen(r.x);
    return make_float4(ok ? x.x : 0.f, ok ? x.y : 0.f, ok ? x.z : 0.f, ok ? x.w : 0.f);
  }
  static constexpr bool HAS_AT = EXACT_ && !Elem<WT>::BF16;      // (see DyLoader::fetch_at)
  __device__ __forceinline__ static void pin(Raw& r) {
    if constexpr (!Elem<WT>::BF16) asm volatile("" : "+v"(r.x.x), "+v"(r.x.y), "+v"(r.x.z), "+v"(r.x.w));
  }
  __device__ __forceinline__ int ld_elems() const { return ld; }
  __device__ __forceinline__ const float* base_ptr() const { return w; }
  __device__ __forceinline__ Raw fetch_at(size_t uoff, unsigned lbytes) const {
    Raw r;
    r.x = *reinterpret_cast<const typename Elem<WT>::V4*>(reinterpret_cast<const char*>(w + uoff) + lbytes);
    return r;
  }
  __device__ __forceinline__ Coef fetch_coef_at(int, unsigned) const { return Coef(); }
};
typedef WLoaderT<float> WLoader;
template <class L> struct PassBf16 { static constexpr bool value = false; };
template <class WT, bool E> struct PassBf16<WLoaderT<WT, E>> { static constexpr bool value = WLoaderT<WT, E>::PASS_BF16; };

// ---------------------------------------------------------------------------------------------
// staging of one [DIM x BK] operand tile through registers into LDS
// ---------------------------------------------------------------------------------------------
// PF = prefetch distance in k-tiles = number of register slots: with PF = 2 the loads of tile t+2 are already in flight
// while tile t+1 is transformed into LDS, so a load has about two k-tile MFMA phases to land instead of one.  The narrow
// (64-column) kernels need that: their k-tile is only 32 MFMAs per wave, shorter than the memory latency under load.
template <int DIM, bool TYPE_R, class L, int PF_ = 1>
struct Stager {
  static constexpr int PF = PF_;
  static constexpr int NV = DIM * (BK / 4) / NT;
  static constexpr int LDS_FLOATS = TYPE_R ? DIM * LDR : BK * DIM;
  typename L::Raw raw[PF][NV];
  typename L::Coef coef[PF];      // TYPE_C uses coef[0] only (the thread's column chunk never changes)
  int lane0, red0[PF];

  __device__ __forceinline__ static void coords(int tid, int q, int& lane_i, int& red_i) {
    const int f = tid + NT * q;
    if (TYPE_R) { constexpr int CH = BK / 4; lane_i = f / CH; red_i = (f % CH) * 4; }
    else { constexpr int C4 = DIM / 4; red_i = f / C4; lane_i = (f % C4) * 4; }
  }
  __device__ __forceinline__ void init(const L& l, int lane0_, int tid) {
    lane0 = lane0_;
    if (!TYPE_R) { int li, ri; coords(tid, 0, li, ri); coef[0] = l.fetch_coef(lane0 + li); }
  }
  template <int S = 0>
  __device__ __forceinline__ void fetch(const L& l, int red0_, int tid) {
    red0[S] = red0_;
    if (TYPE_R) { int li, ri; coords(tid, 0, li, ri); coef[S] = l.fetch_coef(red0_ + ri); }
#pragma unroll
    for (int q = 0; q < NV; ++q) {
      int li, ri; coords(tid, q, li, ri);
      raw[S][q] = TYPE_R ? l.fetch(lane0 + li, red0_ + ri) : l.fetch(red0_ + ri, lane0 + li);
    }
  }
  template <int S = 0>
  __device__ __forceinline__ void store(const L& l, float* tile, int tid) {
#pragma unroll
    for (int q = 0; q < NV; ++q) store_piece<S>(l, tile, tid, q);
  }
  template <int S = 0>
  __device__ __forceinline__ void store_piece(const L& l, float* tile, int tid, int q) {
    int li, ri; coords(tid, q, li, ri);
    const typename L::Coef& c = coef[TYPE_R ? S : 0];
    if (TYPE_R) {
      const float4 v = l.xform(raw[S][q], c, lane0 + li, red0[S] + ri);
      *reinterpret_cast<float4*>(tile + li * LDR + ri) = v;
    } else {
      const float4 v = l.xform(raw[S][q], c, red0[S] + ri, lane0 + li);
      *reinterpret_cast<float4*>(tile + ri * DIM + li) = v;
    }
  }
};

// one BK-deep step of the wave's TM x TN grid of 32x32 MFMA tiles; the fragments of group g+1 are read from
// LDS before the 16 MFMAs of group g are issued, so ds_read latency hides under the matrix pipe.
template <int TM, int TN, bool AR, int DIMA, bool BR, int DIMB>
struct Frags {
  float a[TM][4], b[TN][4];
  __device__ __forceinline__ void load(const float* As, const float* Bs, int a0, int b0, int g, int l31, int h) {
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
  }
};

// MFMAs of k-groups [G0, G1) of the current tile.  Fragments of group g+1 are requested before the MFMAs of
// group g.  After every cluster of TM*TN MFMAs (one k step) `filler(step)` is invoked: the main loop uses it to
// drop one staging piece (transform + ds_write of the NEXT tile) into the 64-cycle shadows of the MFMAs, in
// program order, which is what an in-order wave needs to keep the matrix pipe busy.
template <int TM, int TN, bool AR, int DIMA, bool BR, int DIMB, int G0, int G1, class F>
__device__ __forceinline__ void mma_groups(const float* As, const float* Bs, int a0, int b0, f32x16 (&acc)[TM][TN],
                                           int lane, F&& filler) {
  const int l31 = lane & 31, h = lane >> 5;
  Frags<TM, TN, AR, DIMA, BR, DIMB> f[2];
  f[G0 & 1].load(As, Bs, a0, b0, G0, l31, h);
#pragma unroll
  for (int g = G0; g < G1; ++g) {
    if (g + 1 < G1) f[(g + 1) & 1].load(As, Bs, a0, b0, g + 1, l31, h);
#ifdef T3D_PIN_FRAGS
    __builtin_amdgcn_sched_barrier(0);   // keep the next group's ds_reads ahead of this group's 16 MFMAs
#endif
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
      for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
          acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(f[g & 1].a[tm][i], f[g & 1].b[tn][i], acc[tm][tn], 0, 0, 0);
      filler((g - G0) * 4 + i);
    }
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

// Software-pipelined main loop over the reduction range [red_begin, red_end), two LDS stages, ONE barrier per
// k-tile.  Iteration t: MFMAs of tile t read stage t&1; meanwhile tile t+1 (whose global loads were issued one
// iteration earlier) is transformed and written into the other stage between the MFMAs of the second half,
// and the loads of tile t+2 are issued.  The last tile is peeled so that the steady-state body is branch-free.
//   RAW: stage (t+1)&1 is written during iteration t and read after the barrier that ends it.
//   WAR: stage t&1 is overwritten (tile t+2) during iteration t+1, after the same barrier.
// Prefetch-distance-2 variant (SA::PF == 2).  Register slot (t+1)&1 holds tile t+1 (landed), slot t&1 holds tile t+2
// (in flight).  Iteration t: MFMAs of tile t from LDS stage t&1; tile t+1 goes from its slot into the other stage
// between the MFMAs of the second half; the freed slot is refilled with tile t+3.  Unrolled by two so that the slot
// index is a compile-time constant (register arrays).
template <int S, int TM, int TN, class SA, class SB, class LA, class LB, bool AR, int DIMA, bool BR, int DIMB>
__device__ __forceinline__ void pf2_body(SA& sa, SB& sb, const LA& la, const LB& lb, float* smem, int cur, int red_fetch,
                                         bool do_fetch, int a0, int b0, f32x16 (&acc)[TM][TN], int tid) {
  constexpr int STAGE = SA::LDS_FLOATS + SB::LDS_FLOATS;
  constexpr int GH = BK / 16, GT = BK / 8;
  const int lane = tid & 63;
  const float* As = smem + cur * STAGE;
  const float* Bs = As + SA::LDS_FLOATS;
  float* An = smem + (cur ^ 1) * STAGE;
  float* Bn = An + SA::LDS_FLOATS;
  mma_groups<TM, TN, AR, DIMA, BR, DIMB, 0, GH>(As, Bs, a0, b0, acc, lane, [](int) {});
  __builtin_amdgcn_sched_barrier(0);
  mma_groups<TM, TN, AR, DIMA, BR, DIMB, GH, GT>(As, Bs, a0, b0, acc, lane, [&](int step) {
    if (step < SA::NV) sa.template store_piece<S>(la, An, tid, step);
    else if (step - SA::NV < SB::NV) sb.template store_piece<S>(lb, Bn, tid, step - SA::NV);
  });
  __builtin_amdgcn_sched_barrier(0);
  if (do_fetch) {                        // workgroup-uniform; short reductions would otherwise re-read their last tile
    sa.template fetch<S>(la, red_fetch, tid);
    sb.template fetch<S>(lb, red_fetch, tid);
  }
  __syncthreads();
}

template <int TM, int TN, class SA, class SB, class LA, class LB, bool AR, int DIMA, bool BR, int DIMB>
__device__ __forceinline__ void gemm_mainloop_pf2(SA& sa, SB& sb, const LA& la, const LB& lb, float* smem, int red_begin,
                                                  int red_end, int a0, int b0, f32x16 (&acc)[TM][TN], int tid) {
  constexpr int STAGE = SA::LDS_FLOATS + SB::LDS_FLOATS;
  constexpr int GT = BK / 8;
  static_assert(SA::NV + SB::NV <= (GT - BK / 16) * 4, "staging pieces must fit the MFMA clusters of the second half");
  const int nt = (red_end - red_begin) / BK;
  sa.template fetch<0>(la, red_begin, tid);
  sb.template fetch<0>(lb, red_begin, tid);
  if (nt > 1) { sa.template fetch<1>(la, red_begin + BK, tid); sb.template fetch<1>(lb, red_begin + BK, tid); }
  sa.template store<0>(la, smem, tid);
  sb.template store<0>(lb, smem + SA::LDS_FLOATS, tid);
  if (nt > 2) { sa.template fetch<0>(la, red_begin + 2 * BK, tid); sb.template fetch<0>(lb, red_begin + 2 * BK, tid); }
  __syncthreads();
  int cur = 0, t = 0;
  for (; t + 2 < nt; t += 2) {
    pf2_body<1, TM, TN, SA, SB, LA, LB, AR, DIMA, BR, DIMB>(sa, sb, la, lb, smem, cur, red_begin + (t + 3) * BK, t + 3 < nt, a0, b0,
                                                           acc, tid);
    cur ^= 1;
    pf2_body<0, TM, TN, SA, SB, LA, LB, AR, DIMA, BR, DIMB>(sa, sb, la, lb, smem, cur, red_begin + (t + 4) * BK, t + 4 < nt, a0, b0,
                                                           acc, tid);
    cur ^= 1;
  }
  if (t + 1 < nt) {                      // two tiles left: t is even here, tile t+1 sits in slot 1
    pf2_body<1, TM, TN, SA, SB, LA, LB, AR, DIMA, BR, DIMB>(sa, sb, la, lb, smem, cur, red_begin, false, a0, b0, acc, tid);
    cur ^= 1;
  }
  {
    const float* As = smem + cur * STAGE;
    const float* Bs = As + SA::LDS_FLOATS;
    mma_groups<TM, TN, AR, DIMA, BR, DIMB, 0, GT>(As, Bs, a0, b0, acc, tid & 63, [](int) {});
  }
  __syncthreads();
}

template <int TM, int TN, class SA, class SB, class LA, class LB, bool AR, int DIMA, bool BR, int DIMB>
__device__ __forceinline__ void gemm_mainloop(SA& sa, SB& sb, const LA& la, const LB& lb, float* smem, int red_begin,
                                              int red_end, int a0, int b0, f32x16 (&acc)[TM][TN], int tid) {
  static_assert(SA::PF == SB::PF, "both operands use the same prefetch distance");
  if constexpr (SA::PF == 2) {
    gemm_mainloop_pf2<TM, TN, SA, SB, LA, LB, AR, DIMA, BR, DIMB>(sa, sb, la, lb, smem, red_begin, red_end, a0, b0, acc, tid);
    return;
  }
  constexpr int STAGE = SA::LDS_FLOATS + SB::LDS_FLOATS;
  constexpr int GH = BK / 16, GT = BK / 8;     // k-groups in the first half / in the whole tile
  static_assert(SA::NV + SB::NV <= (GT - GH) * 4, "staging pieces must fit the MFMA clusters of the second half");
  const int lane = tid & 63;
  auto nofill = [](int) {};
  sa.fetch(la, red_begin, tid);
  sb.fetch(lb, red_begin, tid);
  sa.store(la, smem, tid);
  sb.store(lb, smem + SA::LDS_FLOATS, tid);
  if (red_begin + BK < red_end) { sa.fetch(la, red_begin + BK, tid); sb.fetch(lb, red_begin + BK, tid); }
  __syncthreads();
  int cur = 0;
  int red = red_begin;
  for (; red + BK < red_end; red += BK) {
    const float* As = smem + cur * STAGE;
    const float* Bs = As + SA::LDS_FLOATS;
    float* An = smem + (cur ^ 1) * STAGE;
    float* Bn = An + SA::LDS_FLOATS;
    mma_groups<TM, TN, AR, DIMA, BR, DIMB, 0, GH>(As, Bs, a0, b0, acc, lane, nofill);
    // nothing of the staging work may be hoisted into the first half: the next tile's global loads were
    // issued only one barrier ago and get the first half's MFMAs (>= 2048 cycles) to land
    __builtin_amdgcn_sched_barrier(0);
#ifndef T3D_ABL_NOSTAGE
    mma_groups<TM, TN, AR, DIMA, BR, DIMB, GH, GT>(As, Bs, a0, b0, acc, lane, [&](int step) {
      if (step < SA::NV) sa.store_piece(la, An, tid, step);
      else if (step - SA::NV < SB::NV) sb.store_piece(lb, Bn, tid, step - SA::NV);
    });
#else
    (void)An; (void)Bn;
    mma_groups<TM, TN, AR, DIMA, BR, DIMB, GH, GT>(As, Bs, a0, b0, acc, lane, nofill);
#endif
    // keep the new loads BEHIND every wait on the previous batch (vmcnt counts in issue order)
    __builtin_amdgcn_sched_barrier(0);
    // clamp instead of branching: the (unused) tile past the end re-reads the last one
    const int nxt = min(red + 2 * BK, red_end - BK);
#ifndef T3D_ABL_NOSTAGE
    sa.fetch(la, nxt, tid);
    sb.fetch(lb, nxt, tid);
#else
    (void)nxt;
#endif
#ifndef T3D_ABL_NOBAR
    __syncthreads();
#endif
    cur ^= 1;
  }
  {
    const float* As = smem + cur * STAGE;
    const float* Bs = As + SA::LDS_FLOATS;
    mma_groups<TM, TN, AR, DIMA, BR, DIMB, 0, GT>(As, Bs, a0, b0, acc, lane, nofill);
  }
  __syncthreads();
}

// ---------------------------------------------------------------------------------------------
// bf16 path (T3D_BF16, BASELINE configs[4]): the same kernels on v_mfma_f32_32x32x16_bf16
// ---------------------------------------------------------------------------------------------
// Operands are rounded to bf16 in the staging pass (after the fused fp32 element-wise work) and accumulated in fp32.  An MFMA step
// covers 16 reduction indices: lane l holds, for operand row / column l & 31, the 8 consecutive indices 8 * (l >> 5) .. + 7.  LDS
// images keep the layout the tile has in HBM, so global loads stay coalesced and every LDS store is an 8-byte one:
//   R image [lane_dim][BKH + 8]   reduction index contiguous (activations in fwd / dgrad, w in dgrad): fragment = one ds_read_b128;
//                                 rows are 144 B apart = 9 x 16 B, odd, so the 16 lanes a ds_read_b128 serves together hit 16
//                                 different 16-byte slots of the 256-byte bank row
//   C image [BKH][lane_dim + 32]  lane index contiguous (w in fwd, P, both operands of the weight gradient): fragment = two
//                                 ds_read_b64_tr_b16 -- the hardware transposing read hands lane i of a 16-lane group column i of a
//                                 4-row block; the 4 rows of a half-wave's two blocks are DIM/2 + 16 dwords apart, i.e. they tile
//                                 the 64 banks without overlap
constexpr int BKH = 64;
constexpr int LDRH = BKH + 8;
typedef bf16_t __attribute__((address_space(3))) lds_bf16_t;
typedef s16x4 __attribute__((address_space(3))) lds_s16x4;

template <int DIM, bool TYPE_R, class L, int PF_ = 1>
struct StagerH {
  static constexpr int PF = PF_;
  static constexpr int NV = DIM * (BKH / 4) / NT;
  static constexpr int LDC = DIM + 32;
  static constexpr int LDS_ELEMS = TYPE_R ? DIM * LDRH : BKH * LDC;
  typename L::Raw raw[PF][NV];
  typename L::Coef coef[PF];
  int lane0, red0[PF];

  __device__ __forceinline__ static void coords(int tid, int q, int& lane_i, int& red_i) {
    const int f = tid + NT * q;
    if (TYPE_R) { constexpr int CH = BKH / 4; lane_i = f / CH; red_i = (f % CH) * 4; }
    else { constexpr int C4 = DIM / 4; red_i = f / C4; lane_i = (f % C4) * 4; }
  }
  __device__ __forceinline__ void init(const L& l, int lane0_, int tid) {
    lane0 = lane0_;
    if (!TYPE_R) { int li, ri; coords(tid, 0, li, ri); coef[0] = l.fetch_coef(lane0 + li); }
  }
  template <int S>
  __device__ __forceinline__ void fetch(const L& l, int red0_, int tid) {
    red0[S] = red0_;
    if (TYPE_R) { int li, ri; coords(tid, 0, li, ri); coef[S] = l.fetch_coef(red0_ + ri); }
#pragma unroll
    for (int q = 0; q < NV; ++q) {
      int li, ri; coords(tid, q, li, ri);
      raw[S][q] = TYPE_R ? l.fetch(lane0 + li, red0_ + ri) : l.fetch(red0_ + ri, lane0 + li);
    }
  }
  // the two halves of fetch(), for callers that request the data early and the per-channel coefficients late (k_pointmlp_fwd_res)
  template <int S>
  __device__ __forceinline__ void fetch_raw(const L& l, int red0_, int tid) {
    red0[S] = red0_;
#pragma unroll
    for (int q = 0; q < NV; ++q) {
      int li, ri; coords(tid, q, li, ri);
      raw[S][q] = TYPE_R ? l.fetch(lane0 + li, red0_ + ri) : l.fetch(red0_ + ri, lane0 + li);
    }
  }
  template <int S>
  __device__ __forceinline__ void fetch_coefs(const L& l, int red0_, int tid) {
    if (TYPE_R) { int li, ri; coords(tid, 0, li, ri); coef[S] = l.fetch_coef(red0_ + ri); }
  }
  template <int S>
  __device__ __forceinline__ void store_piece(const L& l, bf16_t* tile, int tid, int q) {
    int li, ri; coords(tid, q, li, ri);
    bf16_t* dst = tile + (TYPE_R ? li * LDRH + ri : ri * LDC + li);
    if constexpr (PassBf16<L>::value) {       // bf16 weights inside the matrix: no arithmetic at all between the load and the LDS store
      *reinterpret_cast<bf16x4*>(dst) = raw[S][q].x;
    } else {
      const typename L::Coef& c = coef[TYPE_R ? S : 0];
      const float4 v = TYPE_R ? l.xform(raw[S][q], c, lane0 + li, red0[S] + ri) : l.xform(raw[S][q], c, red0[S] + ri, lane0 + li);
      const bf16x4 h = {(bf16_t)v.x, (bf16_t)v.y, (bf16_t)v.z, (bf16_t)v.w};
      *reinterpret_cast<bf16x4*>(dst) = h;
    }
  }
  template <int S>
  __device__ __forceinline__ void store(const L& l, bf16_t* tile, int tid) {
#pragma unroll
    for (int q = 0; q < NV; ++q) store_piece<S>(l, tile, tid, q);
  }
};

// The same stager with 16-byte granules (loaders with HAS8: bf16 sources inside the matrix): ONE global load and ONE LDS store per
// eight elements instead of two of each -- the arithmetic per element is unchanged, the loads, address computations and stores,
// a third of the staging pass's instructions, halve.  A granule is two adjacent Raw of the 4-wide interface (L::fetch8), so the
// loaders' xform is shared and the values are bit-identical to StagerH's.
template <int DIM, bool TYPE_R, class L, int PF_ = 1>
struct StagerH8 {
  static constexpr int PF = PF_;
  static constexpr int NV = DIM * (BKH / 8) / NT;
  static constexpr int LDC = DIM + 32;
  static constexpr int LDS_ELEMS = TYPE_R ? DIM * LDRH : BKH * LDC;
  static_assert(NV >= 1, "tile too small for 16-byte granules");
  typename L::Raw raw[PF][NV][2];
  typename L::Coef coef[PF][2];
  int lane0, red0[PF];

  __device__ __forceinline__ static void coords(int tid, int q, int& lane_i, int& red_i) {
    const int f = tid + NT * q;
    if (TYPE_R) { constexpr int CH = BKH / 8; lane_i = f / CH; red_i = (f % CH) * 8; }
    else { constexpr int C8 = DIM / 8; red_i = f / C8; lane_i = (f % C8) * 8; }
  }
  __device__ __forceinline__ void init(const L& l, int lane0_, int tid) {
    lane0 = lane0_;
    if (!TYPE_R) { int li, ri; coords(tid, 0, li, ri); coef[0][0] = l.fetch_coef(lane0 + li); coef[0][1] = l.fetch_coef(lane0 + li + 4); }
  }
  template <int S>
  __device__ __forceinline__ void fetch(const L& l, int red0_, int tid) {
    red0[S] = red0_;
    if (TYPE_R) { int li, ri; coords(tid, 0, li, ri); coef[S][0] = l.fetch_coef(red0_ + ri); coef[S][1] = l.fetch_coef(red0_ + ri + 4); }
#pragma unroll
    for (int q = 0; q < NV; ++q) {
      int li, ri; coords(tid, q, li, ri);
      if (TYPE_R) l.fetch8(lane0 + li, red0_ + ri, raw[S][q][0], raw[S][q][1]);
      else l.fetch8(red0_ + ri, lane0 + li, raw[S][q][0], raw[S][q][1]);
    }
  }
  template <int S>
  __device__ __forceinline__ void fetch_raw(const L& l, int red0_, int tid) {
    red0[S] = red0_;
#pragma unroll
    for (int q = 0; q < NV; ++q) {
      int li, ri; coords(tid, q, li, ri);
      if (TYPE_R) l.fetch8(lane0 + li, red0_ + ri, raw[S][q][0], raw[S][q][1]);
      else l.fetch8(red0_ + ri, lane0 + li, raw[S][q][0], raw[S][q][1]);
    }
  }
  template <int S>
  __device__ __forceinline__ void fetch_coefs(const L& l, int red0_, int tid) {
    if (TYPE_R) { int li, ri; coords(tid, 0, li, ri); coef[S][0] = l.fetch_coef(red0_ + ri); coef[S][1] = l.fetch_coef(red0_ + ri + 4); }
  }
  template <int S>
  __device__ __forceinline__ void store(const L& l, bf16_t* tile, int tid) {
#pragma unroll
    for (int q = 0; q < NV; ++q) {
      int li, ri; coords(tid, q, li, ri);
      bf16_t* dst = tile + (TYPE_R ? li * LDRH + ri : ri * LDC + li);
      if constexpr (PassBf16<L>::value) {
        *reinterpret_cast<bf16x8*>(dst) = __builtin_shufflevector(raw[S][q][0].x, raw[S][q][1].x, 0, 1, 2, 3, 4, 5, 6, 7);
      } else {
        const typename L::Coef& c0 = coef[TYPE_R ? S : 0][0];
        const typename L::Coef& c1 = coef[TYPE_R ? S : 0][1];
        const float4 a = TYPE_R ? l.xform(raw[S][q][0], c0, lane0 + li, red0[S] + ri) : l.xform(raw[S][q][0], c0, red0[S] + ri, lane0 + li);
        const float4 b = TYPE_R ? l.xform(raw[S][q][1], c1, lane0 + li, red0[S] + ri + 4) : l.xform(raw[S][q][1], c1, red0[S] + ri, lane0 + li + 4);
        const bf16x8 h = {(bf16_t)a.x, (bf16_t)a.y, (bf16_t)a.z, (bf16_t)a.w, (bf16_t)b.x, (bf16_t)b.y, (bf16_t)b.z, (bf16_t)b.w};
        *reinterpret_cast<bf16x8*>(dst) = h;
      }
    }
  }
};
template <class L, class = void> struct Has8 { static constexpr bool value = false; };
template <class L> struct Has8<L, typename std::enable_if<L::HAS8>::type> { static constexpr bool value = true; };
#ifndef T3D_STAGE8
#define T3D_STAGE8 1       // 0: the 8-byte granules everywhere (A/B of the 16-byte stager)
#endif
template <int DIM, bool TYPE_R, class L, int PF>
using StagerHSel = typename std::conditional<(T3D_STAGE8 != 0) && Has8<L>::value, StagerH8<DIM, TYPE_R, L, PF>, StagerH<DIM, TYPE_R, L, PF>>::type;

// fragment of MFMA step `st` (16 reduction indices) for the 32 operand rows / columns starting at `c0`
template <bool TYPE_R, int DIM>
__device__ __forceinline__ bf16x8 frag_h(const bf16_t* img, int c0, int st, int lane) {
  if (TYPE_R) {
    return *reinterpret_cast<const bf16x8*>(img + (c0 + (lane & 31)) * LDRH + 16 * st + 8 * (lane >> 5));
  } else {
    constexpr int LDC = DIM + 32;
    // 16-lane group g: block columns c0 + 16 * (g & 1) .. + 15, rows k0 .. k0 + 3 (then + 4 .. + 7); lane 4q + p of the group
    // supplies the address of row q, columns 4p .. 4p + 3 and receives column (lane & 15) of the 4 rows
    const int k0 = 16 * st + 8 * (lane >> 5);
    const bf16_t* base = img + (k0 + ((lane & 15) >> 2)) * LDC + c0 + 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(base));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(base + 4 * LDC));
    return __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
  }
}

// MFMA steps [S0, S1) of the current k-tile; filler(step) runs after each step's TM*TN MFMAs (staging pieces of the next tile)
template <int TM, int TN, bool AR, int DIMA, bool BR, int DIMB, int S0, int S1, class F>
__device__ __forceinline__ void mma_steps_h(const bf16_t* As, const bf16_t* Bs, int a0, int b0, f32x16 (&acc)[TM][TN], int lane,
                                            F&& filler) {
  bf16x8 fa[2][TM], fb[2][TN];
#pragma unroll
  for (int tm = 0; tm < TM; ++tm) fa[S0 & 1][tm] = frag_h<AR, DIMA>(As, a0 + tm * 32, S0, lane);
#pragma unroll
  for (int tn = 0; tn < TN; ++tn) fb[S0 & 1][tn] = frag_h<BR, DIMB>(Bs, b0 + tn * 32, S0, lane);
#pragma unroll
  for (int st = S0; st < S1; ++st) {
    if (st + 1 < S1) {
#pragma unroll
      for (int tm = 0; tm < TM; ++tm) fa[(st + 1) & 1][tm] = frag_h<AR, DIMA>(As, a0 + tm * 32, st + 1, lane);
#pragma unroll
      for (int tn = 0; tn < TN; ++tn) fb[(st + 1) & 1][tn] = frag_h<BR, DIMB>(Bs, b0 + tn * 32, st + 1, lane);
    }
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
      for (int tn = 0; tn < TN; ++tn)
        acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[st & 1][tm], fb[st & 1][tn], acc[tm][tn], 0, 0, 0);
    filler(st - S0);
  }
}

// Two LDS stages, one barrier per k-tile of BKH reduction indices.  The A operand (the stream from HBM) has TWO register slots,
// the B operand one.  Iteration t: tile t+1 (A: slot (t+1)&1, loaded during iterations t-2 and t-1; B: loaded during t-1) is
// transformed, rounded to bf16 and written into the other LDS stage FIRST, its registers are refilled right away with A tile
// t+3 and B tile t+2, then the MFMAs of tile t run from stage t&1.  A streamed load therefore has two full iterations to land
// (HBM under load: > 2 us; with one slot every iteration waited out that round trip -- tools/trace_blocks.py: 2.1 us per
// k-tile whatever the layer).  Both slots for both operands do not fit (350+ VGPRs in the 128-wide kernels).
template <int S, int TM, int TN, class SA, class SB, class LA, class LB, bool AR, int DIMA, bool BR, int DIMB>
__device__ __forceinline__ void h_iter(SA& sa, SB& sb, const LA& la, const LB& lb, bf16_t* smem, int cur, int red_a, int red_b,
                                       int a0, int b0, f32x16 (&acc)[TM][TN], int tid) {
  constexpr int STAGE = SA::LDS_ELEMS + SB::LDS_ELEMS;
  constexpr int ST = BKH / 16;
  const bf16_t* As = smem + cur * STAGE;
  const bf16_t* Bs = As + SA::LDS_ELEMS;
  bf16_t* An = smem + (cur ^ 1) * STAGE;
  bf16_t* Bn = An + SA::LDS_ELEMS;
  sa.template store<S>(la, An, tid);
  sb.template store<0>(lb, Bn, tid);
  __builtin_amdgcn_sched_barrier(0);
  sa.template fetch<S>(la, red_a, tid);          // clamped by the caller: past the end the last tile is re-read, never used
  sb.template fetch<0>(lb, red_b, tid);
  __builtin_amdgcn_sched_barrier(0);
  mma_steps_h<TM, TN, AR, DIMA, BR, DIMB, 0, ST>(As, Bs, a0, b0, acc, tid & 63, [](int) {});
  __syncthreads();
}

template <int TM, int TN, class SA, class SB, class LA, class LB, bool AR, int DIMA, bool BR, int DIMB>
__device__ __forceinline__ void gemm_mainloop_h(SA& sa, SB& sb, const LA& la, const LB& lb, float* smem_f, int red_begin,
                                                int red_end, int a0, int b0, f32x16 (&acc)[TM][TN], int tid) {
  static_assert((SA::PF == 2 || SA::PF == 1) && SB::PF == 1, "A: two register slots (or one), B: one");
  bf16_t* smem = reinterpret_cast<bf16_t*>(smem_f);
  constexpr int STAGE = SA::LDS_ELEMS + SB::LDS_ELEMS;
  constexpr int ST = BKH / 16;
  const int last = red_end - BKH;                              // first reduction index of the last tile
  auto clampr = [&](int r) { return min(r, last); };
  if constexpr (SA::PF == 1) {
    // one register slot per operand (the 128-column data-gradient tilings: the second A slot spilled 10-27 VGPRs to scratch):
    // iteration t stores tile t+1 -- requested during iteration t-1 -- into the other stage, then requests tile t+2
    sa.template fetch<0>(la, red_begin, tid);
    sb.template fetch<0>(lb, red_begin, tid);
    sa.template store<0>(la, smem, tid);
    sb.template store<0>(lb, smem + SA::LDS_ELEMS, tid);
    sa.template fetch<0>(la, clampr(red_begin + BKH), tid);
    sb.template fetch<0>(lb, clampr(red_begin + BKH), tid);
    __syncthreads();
    int cur1 = 0;
    for (int red1 = red_begin; red1 + BKH < red_end; red1 += BKH) {
      h_iter<0, TM, TN, SA, SB, LA, LB, AR, DIMA, BR, DIMB>(sa, sb, la, lb, smem, cur1, clampr(red1 + 2 * BKH), clampr(red1 + 2 * BKH), a0, b0,
                                                           acc, tid);
      cur1 ^= 1;
    }
    const bf16_t* As1 = smem + cur1 * STAGE;
    mma_steps_h<TM, TN, AR, DIMA, BR, DIMB, 0, ST>(As1, As1 + SA::LDS_ELEMS, a0, b0, acc, tid & 63, [](int) {});
    __syncthreads();
    return;
  }
  // prologue: A tiles 0, 1 -> slots 0, 1; tile 0 into stage 0; then A tile 2 -> slot 0, B tile 1
  sa.template fetch<0>(la, red_begin, tid);
  sb.template fetch<0>(lb, red_begin, tid);
  sa.template fetch<1>(la, clampr(red_begin + BKH), tid);
  sa.template store<0>(la, smem, tid);
  sb.template store<0>(lb, smem + SA::LDS_ELEMS, tid);
  sa.template fetch<0>(la, clampr(red_begin + 2 * BKH), tid);
  sb.template fetch<0>(lb, clampr(red_begin + BKH), tid);
  __syncthreads();
  int cur = 0, red = red_begin;
  // iteration t (red = its first index) consumes A slot (t+1)&1: unrolled by two so that the slot is a compile-time constant
  for (; red + 2 * BKH < red_end; red += 2 * BKH) {
    h_iter<1, TM, TN, SA, SB, LA, LB, AR, DIMA, BR, DIMB>(sa, sb, la, lb, smem, cur, clampr(red + 3 * BKH), clampr(red + 2 * BKH), a0, b0,
                                                         acc, tid);
    cur ^= 1;
    h_iter<0, TM, TN, SA, SB, LA, LB, AR, DIMA, BR, DIMB>(sa, sb, la, lb, smem, cur, clampr(red + 4 * BKH), clampr(red + 3 * BKH), a0, b0,
                                                         acc, tid);
    cur ^= 1;
  }
  if (red + BKH < red_end) {                     // two tiles left (workgroup-uniform)
    h_iter<1, TM, TN, SA, SB, LA, LB, AR, DIMA, BR, DIMB>(sa, sb, la, lb, smem, cur, last, last, a0, b0, acc, tid);
    cur ^= 1;
  }
  {
    const bf16_t* As = smem + cur * STAGE;
    const bf16_t* Bs = As + SA::LDS_ELEMS;
    mma_steps_h<TM, TN, AR, DIMA, BR, DIMB, 0, ST>(As, Bs, a0, b0, acc, tid & 63, [](int) {});
  }
  __syncthreads();
}

// ---------------------------------------------------------------------------------------------
// fp32 GEMMs on the bf16 matrix pipe: every operand split into three bf16 terms (T3D_X3)
// ---------------------------------------------------------------------------------------------
// gfx950 has no TF32 and its fp32 MFMA (v_mfma_f32_32x32x2_f32, 64 FLOP / cycle / SIMD) runs at 1/16 of the bf16 rate.  An fp32 value
// is the exact sum of three bf16 values, x = h + m + l with h = bf16(x), m = bf16(x - h), l = bf16(x - h - m) (8 + 8 + 8 significand
// bits; both subtractions are exact in fp32), and the product of two bf16 values is exact in fp32.  So
//     x y = hh + (hm + mh) + (hl + lh + mm) + [ml + lm + ll]
// where the bracket is below 2^-24 |x y|: SIX bf16 MFMAs with fp32 accumulation reproduce the fp32 product to fp32 rounding (the
// dropped terms are of the size of the rounding of the fp32 fma chain they replace) at 16 / 6 = 2.7x the fp32 MFMA rate.  Storage,
// loaders, fused element-wise work and epilogues are the fp32 path's; only the LDS images (three bf16 planes per operand tile, in the
// layouts of the bf16 path) and the main loop differ.  Results agree with the fp32 MFMA kernels to the last bits, not bit for bit
// (another summation order): the launchers take this path only when asked (T3D_X3, see t3d_pointmlp_fwd_r).
constexpr int BKX = 16;             // reduction depth of an LDS stage = ONE step of v_mfma_f32_32x32x16_bf16
// Where the accumulators live (an experiment of round 5; T3D_X3_AGPR=1 builds it, the default is 0).  hipcc keeps the C / D matrix of
// an MFMA builtin in ordinary VGPRs whenever a kernel fits 256 of them ("AGPRs: 0" for every kernel of this file), and in a bare loop a
// v_mfma whose accumulator is a VGPR block does not let vector instructions run beside it: tools/micro/mfma_valu_overlap.hip (MI355X,
// two waves per SIMD, [one MFMA, six v_fma_f32] x 4 per iteration) takes 356 cycles per iteration with VGPR accumulators -- the 256
// cycles of the matrix pipe PLUS the vector instructions -- against 284 with the accumulators in AGPRs, and a wave that only multiplies
// starves its SIMD partner's vector stream almost completely (which is why the producer / consumer kernels below lose).  The compiler
// takes the AGPR form of the builtins as soon as the function may use AGPRs at all -- which one (empty) inline-asm statement with an
// "a" operand tells it.  In THESE kernels it buys nothing: per launch at M = 32768, AGPR vs VGPR form, forward 512 -> 256 55.7 vs
// 56.1 us, 256 -> 128 26.2 vs 22.5, 128 -> 1024 63.6 vs 61.5, fused backward 512 -> 256 124.7 vs 112.1; the rider kernels, whose
// small-op bodies need ~240 VGPRs of their own, spill 56-60 registers beside 128 AGPRs and the step goes from 1.205 to 1.257 ms
// (tools/agpr_ab.sh, same box).  hipcc does interleave the staging pass between the MFMAs in the AGPR build (one MFMA : ~7 vector
// instructions); what the waves wait for there is their fragment reads (s_waitcnt lgkmcnt in front of every other MFMA), not the
// vector ports.
#ifndef T3D_X3_W8_DEFAULT
#define T3D_X3_W8_DEFAULT 1         // eight-wave 128 x 256 forward tiles (PathX3W): the default of T3D_X3_W8
#endif
#ifndef T3D_X3_AGPR
#define T3D_X3_AGPR 0
#endif
#if T3D_X3_AGPR
#define T3D_MFMA_IN_AGPRS() do { float agpr_hint_ = 0.f; asm volatile("; accumulators in AGPRs" : "+a"(agpr_hint_)); } while (0)
#else
#define T3D_MFMA_IN_AGPRS() do {} while (0)
#endif
// LDS budget: three planes per operand.  With the bf16 path's padded images (R rows of 16 + 8, C rows of DIM + 32) a 128 x 128 forward
// tile needs 67.6 KB for its two stages: two workgroups per CU.  Alternative: the R image UNPADDED -- rows of 16 bf16 = 32 B, the two
// 16-byte halves of row r swapped when bit 3 of r is set, so that the 16 lanes a ds_read_b128 serves together (16 consecutive rows,
// one half) hit 16 different 16-byte slots -- and the C image is padded by 16 (two-way conflicts on its transposing reads, 12 per
// k-tile): 52.2 KB, THREE workgroups per CU (T3D_X3_LDS=1).
#ifndef T3D_X3_LDS
#define T3D_X3_LDS 0                // 1: that layout.  Measured (B=32 N=1024 step, same box): 1.246 vs 1.241 ms -- most launches have
                                    // exactly two tiles per CU, a third slot stays empty; the conflict-free images stay the default
#endif
constexpr int LDRX = T3D_X3_LDS ? BKX : BKX + 8;
constexpr int LDCX_PAD = T3D_X3_LDS ? 16 : 32;
__device__ __forceinline__ int x3_r_off(int row, int red) {      // bf16 element offset of (row, reduction index red) in an R image plane
  if constexpr (T3D_X3_LDS != 0) return row * LDRX + ((((red >> 3) ^ (row >> 3)) & 1) << 3) + (red & 7);
  else return row * LDRX + red;
}

#ifndef T3D_X3_SPLIT_ASM
#define T3D_X3_SPLIT_ASM 1
#endif
__device__ __forceinline__ void split3(const float4& v, bf16x4& h, bf16x4& m, bf16x4& l) {
  const float x[4] = {v.x, v.y, v.z, v.w};
#ifdef T3D_ABL_X3_FAKESPLIT      // timing ablation (wrong results): one conversion, no residuals
#pragma unroll
  for (int e = 0; e < 4; ++e) { h[e] = (bf16_t)x[e]; m[e] = h[e]; l[e] = h[e]; }
  return;
#endif
#if T3D_X3_SPLIT_ASM
  // Two elements at a time, twelve vector instructions per pair: one packed conversion per term, the two halves widened with a shift
  // and a mask, scalar subtractions.  Written out because from the per-element casts below hipcc (a) converts every other element
  // twice -- once alone for the residual, once packed for the store -- and (b) SLP-packs the subtractions into v_pk_add_f32, which
  // holds the issue port of a SIMD several times as long as two v_sub_f32 beside MFMAs.  Same roundings (v_cvt_pk_bf16_f32: nearest even) and exact subtractions: bit-identical to the form below.
  unsigned hp[2], mp[2], lp[2];
  typedef float f32x2_ __attribute__((ext_vector_type(2)));
  auto pk = [](float a, float b) { const f32x2_ v = {a, b}; return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2)); };
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    const float x0 = x[2 * e], x1 = x[2 * e + 1];
    hp[e] = pk(x0, x1);
#if T3D_X3_SPLIT_ASM == 2      // the subtractions as instructions (hipcc then pads the hazards it cannot see with s_nop)
    float r0, r1, s0, s1;
    asm("v_sub_f32 %0, %1, %2" : "=v"(r0) : "v"(x0), "v"(hp[e] << 16));
    asm("v_sub_f32 %0, %1, %2" : "=v"(r1) : "v"(x1), "v"(hp[e] & 0xffff0000u));
    mp[e] = pk(r0, r1);
    asm("v_sub_f32 %0, %1, %2" : "=v"(s0) : "v"(r0), "v"(mp[e] << 16));
    asm("v_sub_f32 %0, %1, %2" : "=v"(s1) : "v"(r1), "v"(mp[e] & 0xffff0000u));
#else                          // plain subtractions; the empty statements keep the SLP vectoriser from pairing them
    float r0 = x0 - __uint_as_float(hp[e] << 16);
    asm("" : "+v"(r0));
    const float r1 = x1 - __uint_as_float(hp[e] & 0xffff0000u);
    mp[e] = pk(r0, r1);
    float s0 = r0 - __uint_as_float(mp[e] << 16);
    asm("" : "+v"(s0));
    const float s1 = r1 - __uint_as_float(mp[e] & 0xffff0000u);
#endif
    lp[e] = pk(s0, s1);
  }
  h = __builtin_bit_cast(bf16x4, make_uint2(hp[0], hp[1]));
  m = __builtin_bit_cast(bf16x4, make_uint2(mp[0], mp[1]));
  l = __builtin_bit_cast(bf16x4, make_uint2(lp[0], lp[1]));
#else
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const bf16_t a = (bf16_t)x[e];
    const float r1 = x[e] - (float)a;          // exact
    const bf16_t b = (bf16_t)r1;
    const float r2 = r1 - (float)b;            // exact
    h[e] = a; m[e] = b; l[e] = (bf16_t)r2;
  }
#endif
}

// A [K, N] fp32 matrix that was split into three bf16 planes beforehand (t3d_split_x3: the optimiser's weights, once per step): the
// tile is COPIED into the LDS planes -- no arithmetic between the load and the store.  Tiles lie inside the matrix (launcher-checked).
struct WLoaderX3 {
  static constexpr bool PRESPLIT = true;
  const bf16_t* p;      // plane 0
  long stride;          // elements between planes
  int ld;
  struct Raw { bf16x8 v; };      // EIGHT consecutive elements of ONE plane: a 16-byte load and a 16-byte LDS store (StagerX3W)
  struct Coef {};
  __device__ __forceinline__ Coef fetch_coef(int) const { return Coef(); }
  __device__ __forceinline__ Raw fetch(int plane, int row, int col) const {
    Raw r;
    r.v = *reinterpret_cast<const bf16x8*>(p + (size_t)plane * stride + (size_t)row * ld + col);
    return r;
  }
};
// A [K, N] fp32 weight matrix split beforehand into three bf16 planes IN MFMA-FRAGMENT ORDER (t3d_split_x3_frag, once per step behind
// the optimiser): plane p, 16-deep reduction tile rt, 32-wide block nb of the other index, then the 64 lanes' eight elements
//     frag[((p * RT + rt) * NB + nb) * 64 + lane][j] = Op[nb * 32 + (lane & 31)][rt * 16 + 8 * (lane >> 5) + j]
// (Op[n][k] = w[k][n] for the forward, Op[k][n] = w[k][n] for the data gradient): exactly the B operand of one
// v_mfma_f32_32x32x16_bf16, so a wave reads a fragment with ONE global_load_dwordx4 per lane -- 1 KB contiguous, no LDS image, no
// conversion, no ds_write, no ds_read.  Round 6's ablations of the hand-placed iteration (profiles/r06_il_ablations.log) put the cost
// of a k-tile in its LDS and global INSTRUCTIONS as much as in its vector ALU work (a ds_write_b64 or a fragment read costs the wave two
// to three vector-instruction slots, a global load six); the weight operand was a third of all three.
struct WLoaderX3F {
  static constexpr bool FRAG = true;
  const bf16_t* p;      // plane 0
  long stride;          // elements between planes (= K * N)
  int nb;               // 32-wide blocks of the lane index (N / 32 forward, K / 32 data gradient)
  struct Raw {};
  struct Coef {};
  // one dword of every 128-byte line of the 3 x TN fragments of a wave's k-tile: lane -> (plane, block, line); the lanes past the last
  // line repeat it.  A prefetch into the XCD's L2 (see x3_iter_il), not a read of the data.
  template <int TN>
  __device__ __forceinline__ unsigned touch(int c0, int red0, int lane) const {
    const int l = lane < 24 * TN ? lane : 24 * TN - 1;
    const int plane = l / (8 * TN), x = (l / 8) % TN, line = l % 8;
    const size_t uoff = ((size_t)(red0 >> 4) * (size_t)nb + (size_t)(c0 >> 5)) * 512u;      // uniform
    return *reinterpret_cast<const unsigned*>(reinterpret_cast<const char*>(p + uoff) + ((size_t)plane * (size_t)stride + (size_t)x * 512u) * 2u +
                                              (unsigned)line * 128u);
  }
  __device__ __forceinline__ bf16x8 gfrag(int plane, int c0, int red0, int lane) const {
    const size_t uoff = (size_t)plane * (size_t)stride + ((size_t)(red0 >> 4) * (size_t)nb + (size_t)(c0 >> 5)) * 512u;      // uniform
    return *reinterpret_cast<const bf16x8*>(reinterpret_cast<const char*>(p + uoff) + (unsigned)lane * 16u);
  }
};
template <class L, class = void> struct IsFrag { static constexpr bool value = false; };
template <class L> struct IsFrag<L, typename std::enable_if<L::FRAG>::type> { static constexpr bool value = true; };
template <class L, class = void> struct HasCtab { static constexpr bool value = false; };
template <class L> struct HasCtab<L, typename std::enable_if<L::HAS_CTAB>::type> { static constexpr bool value = true; };
template <class L, class = void> struct HasAt { static constexpr bool value = false; };
template <class L> struct HasAt<L, typename std::enable_if<L::HAS_AT>::type> { static constexpr bool value = true; };
template <class L, class = void> struct PreSplit { static constexpr bool value = false; };
template <class L> struct PreSplit<L, typename std::enable_if<L::PRESPLIT>::type> { static constexpr bool value = true; };

// one [DIM x BKX] operand tile: fp32 loader -> registers -> three bf16 planes in LDS (R image [DIM][LDRX] / C image [BKX][DIM + 32])
#ifndef T3D_X3_PF
#define T3D_X3_PF 1                 // register slots = k-tiles in flight per operand (see gemm_mainloop_x3)
#endif
#ifndef T3D_X3_PIECEWISE
#define T3D_X3_PIECEWISE 1          // refill a slot piece by piece, right behind each piece's store
#endif
#ifndef T3D_X3_FRAGPF
#define T3D_X3_FRAGPF 1             // 1: the next tile's fragments are read behind a mid-iteration barrier (see x3_iter_fp)
#endif
#ifndef T3D_X3_SGB
#define T3D_X3_SGB 0                // > 0: sched_group_barrier pipeline, that many VALU instructions behind each MFMA (see x3_iter)
#endif
#ifndef T3D_X3_BRING
// Two experiments of round 6 on the LATENCY of the global weight-fragment loads, both measured SLOWER and off (step 1.213 / 1.216 against
// 1.194 ms, same box; forward 512 -> 256 51.6 / 49.8 against 49.5 us): T3D_X3_BRING=1 keeps the fragments in a ring of three register
// sets loaded two tiles ahead (the 128-wide kernels then spill 10-22 VGPRs); T3D_X3_BTOUCH=n touches the fragment lines of the tile n
// tiles ahead with one dword load per lane.  What the fragment loads cost is their ISSUE (a global load holds its wave ~50-60 cycles
// beside MFMAs), not their latency: one more load per iteration only adds to it.
#define T3D_X3_BRING 0
#endif
#ifndef T3D_X3_BTOUCH
#define T3D_X3_BTOUCH 0
#endif
#ifndef T3D_X3_FAIR
#define T3D_X3_FAIR 1               // the younger workgroup of a CU pair leads the first part of its k loop at wave priority 1 (gemm_mainloop_x3)
#endif
#ifndef T3D_X3_FAIR_NUM
#define T3D_X3_FAIR_NUM 1
#define T3D_X3_FAIR_DEN 2
#endif
#ifndef T3D_X3_PRIO_WGRAD
#define T3D_X3_PRIO_WGRAD 1         // weight-gradient / Gram tiles at wave priority 1 for their whole k loop (gemm_mainloop_x3)
#endif
#define T3D_FAIR_CUS 256            // CUs of an MI355X: the first this many workgroups of a launch are the older halves of the CU pairs
#ifndef T3D_X3_COEF_LDS
#define T3D_X3_COEF_LDS 1           // fragment-weight kernels: the first operand's per-channel coefficients from a table in LDS (DyLoader::ctab_fill)
#endif
#ifndef T3D_X3_LATE_M
#define T3D_X3_LATE_M 1             // x3_iter_il: the m-plane fragments are read at the head of the iteration that multiplies them (ILSched)
#endif
#ifndef T3D_X3_IL
#define T3D_X3_IL 1                 // 1: the hand-placed iteration (x3_iter_il): ONE MFMA, then its share of the staging pass, fenced
#endif
template <int... I, class F>
__device__ __forceinline__ void static_for_seq(std::integer_sequence<int, I...>, F&& f) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, class F>      // f(integral_constant<int, 0>) ... f(integral_constant<int, N - 1>): compile-time indices (register arrays, if constexpr)
__device__ __forceinline__ void static_for(F&& f) { static_for_seq(std::make_integer_sequence<int, N>{}, static_cast<F&&>(f)); }

template <int DIM, bool TYPE_R, class L, int PF_ = T3D_X3_PF, int NTX = NT>      // NTX: threads that stage the tile (512: the eight-wave tiles)
struct StagerX3 {
  static constexpr int PF = PF_;
  static constexpr int NV = DIM * (BKX / 4) / NTX;
  static constexpr int LDC = DIM + LDCX_PAD;
  static constexpr int PLANE = TYPE_R ? DIM * LDRX : BKX * LDC;      // bf16 elements of one plane
  static constexpr int LDS_ELEMS = 3 * PLANE;
  static_assert(NV >= 1, "tile smaller than one staging pass");
  typename L::Raw raw[PF][NV];
  typename L::Coef coef[PF];      // TYPE_C uses coef[0] only (the thread's column chunk never changes)
  int lane0, red0[PF];
  static constexpr bool AT = T3D_X3_IL && HasAt<L>::value;      // uniform base + per-lane byte offset (L::fetch_at)
  unsigned lbytes[AT ? NV : 1], cbytes;                          // the lane's byte offsets: piece q of the tile; its coefficient chunk
  int ld;

  __device__ __forceinline__ static void coords(int tid, int q, int& lane_i, int& red_i) {
    const int f = tid + NTX * q;
    if (TYPE_R) { constexpr int CH = BKX / 4; lane_i = f / CH; red_i = (f % CH) * 4; }
    else { constexpr int C4 = DIM / 4; red_i = f / C4; lane_i = (f % C4) * 4; }
  }
  __device__ __forceinline__ void init(const L& l, int lane0_, int tid) {
    lane0 = lane0_;
    if (!TYPE_R) { int li, ri; coords(tid, 0, li, ri); coef[0] = l.fetch_coef(lane0 + li); }
    if constexpr (AT) {
      ld = l.ld_elems();
#pragma unroll
      for (int q = 0; q < NV; ++q) {
        int li, ri; coords(tid, q, li, ri);
        lbytes[q] = (TYPE_R ? (unsigned)li * (unsigned)ld + (unsigned)ri : (unsigned)ri * (unsigned)ld + (unsigned)li) * 4u;
        asm volatile("" : "+v"(lbytes[q]));      // (opaque: not folded back into a 64-bit product per load)
      }
      int li, ri; coords(tid, 0, li, ri);
      cbytes = (unsigned)ri * 4u;
      asm volatile("" : "+v"(cbytes));
    }
  }
  // the il_step forms of fetch_piece / fetch_head
  template <int S>
  __device__ __forceinline__ void il_fetch_piece(const L& l, int red0_, int tid, int q) {
    if constexpr (AT) {
      const size_t uoff = TYPE_R ? (size_t)lane0 * (size_t)ld + (size_t)red0_ : (size_t)red0_ * (size_t)ld + (size_t)lane0;
      raw[S][q] = l.fetch_at(uoff, lbytes[q]);
    } else {
      fetch_piece<S>(l, red0_, tid, q);
    }
  }
  template <int S, bool CT = false>      // CT: the coefficients come from the loader's table in LDS (T3D_X3_COEF_LDS; `ctab`)
  __device__ __forceinline__ void il_fetch_head(const L& l, int red0_, int tid, const float* ctab = nullptr) {
    if constexpr (AT) {
      red0[S] = red0_;
      if constexpr (CT && TYPE_R && HasCtab<L>::value) {
        int li, ri; coords(tid, 0, li, ri);
        coef[S] = l.ctab_coef(ctab, red0_ + ri);
      } else {
        if (TYPE_R) coef[S] = l.fetch_coef_at(red0_, cbytes);
      }
    } else {
      fetch_head<S>(l, red0_, tid);
    }
  }
  template <int S>
  __device__ __forceinline__ void fetch(const L& l, int red0_, int tid) {
    red0[S] = red0_;
    if (TYPE_R) { int li, ri; coords(tid, 0, li, ri); coef[S] = l.fetch_coef(red0_ + ri); }
#pragma unroll
    for (int q = 0; q < NV; ++q) {
      int li, ri; coords(tid, q, li, ri);
      raw[S][q] = TYPE_R ? l.fetch(lane0 + li, red0_ + ri) : l.fetch(red0_ + ri, lane0 + li);
    }
  }
  // one piece of fetch<S>(): requested right behind the store of the piece that held the register (the slot is refilled piece by piece,
  // so every load has a whole iteration to land instead of the tail of one)
  template <int S>
  __device__ __forceinline__ void fetch_piece(const L& l, int red0_, int tid, int q) {
    int li, ri; coords(tid, q, li, ri);
    raw[S][q] = TYPE_R ? l.fetch(lane0 + li, red0_ + ri) : l.fetch(red0_ + ri, lane0 + li);
  }
  template <int S>
  __device__ __forceinline__ void fetch_head(const L& l, int red0_, int tid) {      // behind the LAST piece's store: the slot's coordinates
    red0[S] = red0_;
    if (TYPE_R) { int li, ri; coords(tid, 0, li, ri); coef[S] = l.fetch_coef(red0_ + ri); }
  }
  template <int S>
  __device__ __forceinline__ void store_piece(const L& l, bf16_t* tile, int tid, int q) {
    int li, ri; coords(tid, q, li, ri);
    bf16x4 h, m, lo;
    {
      const typename L::Coef& c = coef[TYPE_R ? S : 0];
      const float4 v = TYPE_R ? l.xform(raw[S][q], c, lane0 + li, red0[S] + ri) : l.xform(raw[S][q], c, red0[S] + ri, lane0 + li);
      split3(v, h, m, lo);
    }
    bf16_t* dst = tile + (TYPE_R ? x3_r_off(li, ri) : ri * LDC + li);
    *reinterpret_cast<bf16x4*>(dst) = h;
#ifdef T3D_ABL_X3_WRITE1        // timing ablation (wrong results): one plane written instead of three
    if (ri == 12345) {
#endif
    *reinterpret_cast<bf16x4*>(dst + PLANE) = m;
    *reinterpret_cast<bf16x4*>(dst + 2 * PLANE) = lo;
#ifdef T3D_ABL_X3_WRITE1
    }
#endif
  }
  template <int S>
  __device__ __forceinline__ void store(const L& l, bf16_t* tile, int tid) {
#pragma unroll
    for (int q = 0; q < NV; ++q) store_piece<S>(l, tile, tid, q);
  }
  // ---- the same piece in SIX micro-steps (x3_iter_il places one or two behind each MFMA; same roundings, same exact subtractions
  // as split3: bit-identical planes).  Vector instructions per step: the loader's element-wise work (batch-norm + ReLU: 8; a weight
  // tile: none), then 5 / 5 / 6 / 5 of the split, then the last conversion with the three plane stores and the slot's refill.
  static constexpr int NU = 6;
  static constexpr int XCOST = std::is_empty<typename L::Coef>::value ? 0 : 8;
  __host__ __device__ __forceinline__ static constexpr int il_cost(int u) { return u == 0 ? XCOST : (u == 3 ? 6 : 5); }
  float ilx[4], ilr0, ilr1;
  unsigned ilh[2], ilm[2], ill[2];
  __device__ __forceinline__ static unsigned il_pk(float a, float b) {
    typedef float f32x2_ __attribute__((ext_vector_type(2)));
    const f32x2_ v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
  }
  template <int E> __device__ __forceinline__ void il_h() {      // first term of pair E and its residuals
    ilh[E] = il_pk(ilx[2 * E], ilx[2 * E + 1]);
    ilr0 = ilx[2 * E] - __uint_as_float(ilh[E] << 16);
    asm("" : "+v"(ilr0));                                        // (keeps the SLP vectoriser from pairing the subtractions: v_pk_add_f32)
    ilr1 = ilx[2 * E + 1] - __uint_as_float(ilh[E] & 0xffff0000u);
  }
  template <int E> __device__ __forceinline__ void il_m() {      // second term, second residuals
    ilm[E] = il_pk(ilr0, ilr1);
    float s0 = ilr0 - __uint_as_float(ilm[E] << 16);
    asm("" : "+v"(s0));
    ilr1 = ilr1 - __uint_as_float(ilm[E] & 0xffff0000u);
    ilr0 = s0;
  }
  template <int E> __device__ __forceinline__ void il_l() { ill[E] = il_pk(ilr0, ilr1); }
  // A micro-step ENDS with a volatile empty asm statement on its outputs (and an MFMA with one on its accumulator).  sched_barrier(0) alone
  // fences the machine schedulers only: the arithmetic between two fences is side-effect free for the IR optimisers and the selection
  // DAG, and in the fused backward kernels they moved whole pieces across ten fences at a time (ISA of the first round-6 build: ten
  // empty fence pairs in a row, then 30 vector instructions and five MFMAs in one lump).  Volatile asm statements keep their order
  // among themselves and with the fences: a value that leaves a step through one cannot be computed after it, and the next step, which
  // consumes it, not before it.  Not at the FRONT of a step as well: an asm statement that defines a VGPR costs an s_nop in front of
  // the next instruction that reads it (the compiler cannot see what wrote it), and never on a register a load is still filling (the
  // wait-count pass treats the asm as its use: s_waitcnt vmcnt(0) in the loop of the first pinned build).
#define T3D_PIN1(a) asm volatile("" : "+v"(a))
#define T3D_PIN2(a, b) asm volatile("" : "+v"(a), "+v"(b))
#define T3D_PIN3(a, b, c) asm volatile("" : "+v"(a), "+v"(b), "+v"(c))
#define T3D_PIN4(a, b, c, d) asm volatile("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d))
  template <int S, int Q, int U, bool CT = false>
  __device__ __forceinline__ void il_step(const L& l, bf16_t* tile, int tid, int red_fetch, const float* ctab = nullptr) {
    if constexpr (U == 0) {
      int li, ri; coords(tid, Q, li, ri);
      const typename L::Coef& c = coef[TYPE_R ? S : 0];
#ifdef T3D_X3_PIN_RAW
      l.pin(raw[S][Q]);                                          // (the step that consumes the loaded registers: the wait belongs here)
#endif
      const float4 v = TYPE_R ? l.xform(raw[S][Q], c, lane0 + li, red0[S] + ri) : l.xform(raw[S][Q], c, red0[S] + ri, lane0 + li);
      ilx[0] = v.x; ilx[1] = v.y; ilx[2] = v.z; ilx[3] = v.w;
      // (not when the loader computes nothing -- a weight tile: the asm would tie single registers to parts of the load's 128-bit
      // result, the allocator then copies them out at the loop's back edge, behind s_waitcnt vmcnt(0) on a load two gaps old)
      if constexpr (XCOST != 0) T3D_PIN4(ilx[0], ilx[1], ilx[2], ilx[3]);
      // The slot's coefficients (batch-norm scale / shift, dy's c0 c1 c2) are requested HERE, right behind their last use, not with
      // the last piece's refill: as the youngest loads of the iteration, a register copy of them at the loop's back edge (the
      // allocator's, for a v_fmac that accumulates into one) waited with s_waitcnt vmcnt(0) on loads two gaps old.
#ifndef T3D_ABL_IL_NOLOAD
      if constexpr (Q == NV - 1) { asm volatile("" ::: "memory"); il_fetch_head<S, CT>(l, red_fetch, tid, ctab); asm volatile("" ::: "memory"); }
#endif
    } else if constexpr (U == 1) {
      il_h<0>();
      T3D_PIN3(ilh[0], ilr0, ilr1);
    } else if constexpr (U == 2) {
      il_m<0>();
      T3D_PIN3(ilm[0], ilr0, ilr1);
    } else if constexpr (U == 3) {
      il_l<0>();
      il_h<1>();
      T3D_PIN4(ill[0], ilh[1], ilr0, ilr1);
    } else if constexpr (U == 4) {
      il_m<1>();
      T3D_PIN3(ilm[1], ilr0, ilr1);
    } else {
      asm volatile("" ::: "memory");
      il_l<1>();
      int li, ri; coords(tid, Q, li, ri);
      bf16_t* dst = tile + (TYPE_R ? x3_r_off(li, ri) : ri * LDC + li);
#ifdef T3D_ABL_IL_NOWRITE      // timing ablation (wrong results): the planes are not written (one conditional store keeps the split alive)
      if (ilh[0] == 0x12345678u && ill[1] == 0x9abcdef0u && ilm[0] == 0x1u && ilm[1] == ilh[1] && ill[0] == 7u) *reinterpret_cast<uint2*>(dst) = make_uint2(ilh[0], ilh[1]);
#else
      *reinterpret_cast<uint2*>(dst) = make_uint2(ilh[0], ilh[1]);
      *reinterpret_cast<uint2*>(dst + PLANE) = make_uint2(ilm[0], ilm[1]);
      *reinterpret_cast<uint2*>(dst + 2 * PLANE) = make_uint2(ill[0], ill[1]);
#endif
#if defined(T3D_ABL_IL_NOLOAD)       // timing ablation (wrong results): the slot is never refilled (the first tile is staged again and again)
#elif defined(T3D_ABL_IL_NOLOAD_R)   // ... only type-R operands (the [M, C] stream of forward / data gradient) are not refilled
      if constexpr (!TYPE_R) il_fetch_piece<S>(l, red_fetch, tid, Q);
#elif defined(T3D_ABL_IL_NOLOAD_C)   // ... only type-C operands
      if constexpr (TYPE_R) il_fetch_piece<S>(l, red_fetch, tid, Q);
#elif defined(T3D_ABL_IL_LOADDUMMY)  // ... the loads are issued, their results never used (issue cost without the waits)
      if constexpr (AT) {
        const size_t uoff = TYPE_R ? (size_t)lane0 * (size_t)ld + (size_t)red_fetch : (size_t)red_fetch * (size_t)ld + (size_t)lane0;
        const char* pp = reinterpret_cast<const char*>(l.base_ptr() + uoff) + lbytes[Q];
        float4 dummy;
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dummy) : "v"(pp) : "memory");
      }
#else
      il_fetch_piece<S>(l, red_fetch, tid, Q);
#endif
      asm volatile("" ::: "memory");
    }
  }
};

// ... of a weight matrix that arrives as three bf16 planes (WLoaderX3): the tile is COPIED, sixteen bytes at a time -- chunk f of the
// tile's 3 x DIM x BKX / 8 chunks is eight consecutive elements of plane f / (2 DIM); no arithmetic, one load and one ds_write_b128 per
// 8 elements (round 4's form moved 8-byte pieces, three loads and three stores per 4 elements, and lost to the in-kernel split).
template <int DIM, bool TYPE_R, class L, int PF_ = T3D_X3_PF>
struct StagerX3W {
  static constexpr int PF = PF_;
  static constexpr int LDC = DIM + LDCX_PAD;
  static constexpr int PLANE = TYPE_R ? DIM * LDRX : BKX * LDC;
  static constexpr int LDS_ELEMS = 3 * PLANE;
  static constexpr int CPP = DIM * BKX / 8;                 // chunks per plane
  static constexpr int NCH = 3 * CPP;
  static constexpr int NV = (NCH + NT - 1) / NT;            // 3 (DIM = 128), 2 (DIM = 64: the second pass on half of the waves)
  static_assert(NCH % 64 == 0, "a pass ends on a wave boundary");
  typename L::Raw raw[PF][NV];
  int lane0;
  __device__ __forceinline__ static bool coords(int tid, int q, int& plane, int& lane_i, int& red_i) {
    const int f = tid + NT * q;
    plane = f / CPP;
    const int c = f % CPP;
    if (TYPE_R) { lane_i = c / (BKX / 8); red_i = (c % (BKX / 8)) * 8; }
    else { constexpr int C8 = DIM / 8; red_i = c / C8; lane_i = (c % C8) * 8; }
    return NCH % NT == 0 || f < NCH;      // (wave-uniform)
  }
  __device__ __forceinline__ void init(const L&, int lane0_, int) { lane0 = lane0_; }
  template <int S>
  __device__ __forceinline__ void fetch_piece(const L& l, int red0_, int tid, int q) {
    int pl, li, ri;
    if (coords(tid, q, pl, li, ri)) raw[S][q] = TYPE_R ? l.fetch(pl, lane0 + li, red0_ + ri) : l.fetch(pl, red0_ + ri, lane0 + li);
  }
  template <int S>
  __device__ __forceinline__ void fetch(const L& l, int red0_, int tid) {
#pragma unroll
    for (int q = 0; q < NV; ++q) fetch_piece<S>(l, red0_, tid, q);
  }
  template <int S> __device__ __forceinline__ void fetch_head(const L&, int, int) {}
  template <int S>
  __device__ __forceinline__ void store_piece(const L&, bf16_t* tile, int tid, int q) {
    int pl, li, ri;
    if (coords(tid, q, pl, li, ri))
      *reinterpret_cast<bf16x8*>(tile + pl * PLANE + (TYPE_R ? x3_r_off(li, ri) : ri * LDC + li)) = raw[S][q].v;
  }
  template <int S>
  __device__ __forceinline__ void store(const L& l, bf16_t* tile, int tid) {
#pragma unroll
    for (int q = 0; q < NV; ++q) store_piece<S>(l, tile, tid, q);
  }
  static constexpr int NU = 1;      // x3_iter_il: a piece is one copy (ds_write_b128) and its refill
  __host__ __device__ __forceinline__ static constexpr int il_cost(int) { return 3; }
  template <int S, int Q, int U>
  __device__ __forceinline__ void il_step(const L& l, bf16_t* tile, int tid, int red_fetch) {
    asm volatile("" ::: "memory");
    store_piece<S>(l, tile, tid, Q);
    fetch_piece<S>(l, red_fetch, tid, Q);
    asm volatile("" ::: "memory");
  }
};

// ... of a weight matrix in fragment order (WLoaderX3F): nothing is staged -- the main loop reads the fragments from global memory
template <int DIM, bool TYPE_R, class L, int PF_ = T3D_X3_PF>
struct StagerX3F {
  static constexpr int PF = PF_;
  static constexpr bool FROM_GLOBAL = true;
  static constexpr int NV = 0, NU = 1, PLANE = 0, LDS_ELEMS = 0;
  __host__ __device__ __forceinline__ static constexpr int il_cost(int) { return 0; }
  int lane0;      // first column of the workgroup's tile (the fragment planes are indexed by absolute column)
  __device__ __forceinline__ void init(const L&, int lane0_, int) { lane0 = lane0_; }
  __device__ __forceinline__ bf16x8 gfrag(const L& l, int plane, int c_rel, int red0, int lane) const { return l.gfrag(plane, lane0 + c_rel, red0, lane); }
  template <int TN> __device__ __forceinline__ unsigned touch(const L& l, int c_rel, int red0, int lane) const { return l.template touch<TN>(lane0 + c_rel, red0, lane); }
  template <int S> __device__ __forceinline__ void fetch(const L&, int, int) {}
  template <int S> __device__ __forceinline__ void store(const L&, bf16_t*, int) {}
  template <int S, int Q, int U> __device__ __forceinline__ void il_step(const L&, bf16_t*, int, int) {}
};
template <class SB, class = void> struct FromGlobal { static constexpr bool value = false; };
template <class SB> struct FromGlobal<SB, typename std::enable_if<SB::FROM_GLOBAL>::type> { static constexpr bool value = true; };

// fragment of the tile's one MFMA step for the 32 operand rows / columns starting at c0 (cf. frag_h)
template <bool TYPE_R, int DIM>
__device__ __forceinline__ bf16x8 frag_x(const bf16_t* img, int c0, int lane) {
  if (TYPE_R) {
    return *reinterpret_cast<const bf16x8*>(img + x3_r_off(c0 + (lane & 31), 8 * (lane >> 5)));
  } else {
    constexpr int LDC = DIM + LDCX_PAD;
    const int k0 = 8 * (lane >> 5);
    const bf16_t* base = img + (k0 + ((lane & 15) >> 2)) * LDC + c0 + 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(base));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(base + 4 * LDC));
    return __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
  }
}

// The six products of one 16-deep step, smallest terms first; after each product's TM*TN MFMAs `filler(p)` runs (staging pieces of the
// next tile: their conversions issue in the shadows of the MFMAs).
// SYM (Gram matrices, both operands the same tensor): the result must be BITWISE symmetric -- the weight-gradient assembly reads G
// transposed (t3d_pool_wgrad_finish) -- but G[i][j] sums l_i h_j, h_i l_j, ... and G[j][i] the same values in another order.  Three
// accumulators make the order irrelevant: A takes l h and m h, B takes h l and h m, C the symmetric m m and h h; then A[j][i] is
// bit for bit B[i][j], C is symmetric, and (A + B) + C is the same number on both sides (two-term fp32 addition commutes).
template <bool SYM, int TM, int TN, bool AR, int DIMA, int PLA, bool BR, int DIMB, int PLB, class F>
__device__ __forceinline__ void mma_x3(const bf16_t* As, const bf16_t* Bs, int a0, int b0, f32x16 (&acc)[TM][TN], f32x16 (&accb)[SYM ? TM : 1][SYM ? TN : 1],
                                       f32x16 (&accc)[SYM ? TM : 1][SYM ? TN : 1], int lane, F&& filler) {
  bf16x8 fa[3][TM], fb[3][TN];
#pragma unroll
  for (int pl = 0; pl < 3; ++pl) {
#ifdef T3D_ABL_X3_READ1         // timing ablation (wrong results): the fragments of one plane read instead of three
    if (pl > 0) {
#pragma unroll
      for (int tm = 0; tm < TM; ++tm) fa[pl][tm] = fa[0][tm];
#pragma unroll
      for (int tn = 0; tn < TN; ++tn) fb[pl][tn] = fb[0][tn];
      continue;
    }
#endif
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) fa[pl][tm] = frag_x<AR, DIMA>(As + pl * PLA, a0 + tm * 32, lane);
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) fb[pl][tn] = frag_x<BR, DIMB>(Bs + pl * PLB, b0 + tn * 32, lane);
  }
  constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};      // l h, h l, m m, m h, h m, h h
  constexpr int TG[6] = {0, 1, 2, 0, 1, 2};                                  // SYM: which accumulator
#pragma unroll
  for (int p = 0; p < 6; ++p) {
#ifdef T3D_ABL_X3_1PROD      // timing ablation (wrong results): one product instead of six
    if (p == 5)
#endif
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
      for (int tn = 0; tn < TN; ++tn) {
        if constexpr (SYM) {
          if (TG[p] == 0) acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[PA[p]][tm], fb[PB[p]][tn], acc[tm][tn], 0, 0, 0);
          else if (TG[p] == 1) accb[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[PA[p]][tm], fb[PB[p]][tn], accb[tm][tn], 0, 0, 0);
          else accc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[PA[p]][tm], fb[PB[p]][tn], accc[tm][tn], 0, 0, 0);
        } else {
          acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[PA[p]][tm], fb[PB[p]][tn], acc[tm][tn], 0, 0, 0);
        }
      }
    filler(p);
  }
}

// ---- fragments carried across the barrier (T3D_X3_FRAGPF) ---------------------------------------------------------------------------
// In the loop above every iteration begins behind its barrier with twelve fragment reads whose latency nothing covers (the MFMAs need
// them).  Here the barrier sits in the MIDDLE of an iteration: the staging pieces of tile t + 1 go into the other stage behind the FIRST
// product groups, then the barrier, then the fragment reads of tile t + 1 into a second register set while the last product groups of
// tile t still run; iteration t + 1 starts with its fragments in registers.  Still two stages and one barrier per k-tile: the stage
// written in iteration t held tile t - 1, whose fragments every wave had consumed before it passed the barrier of iteration t - 1.
template <int TM, int TN> struct FragsX3 { bf16x8 a[3][TM], b[3][TN]; };

template <bool AR, int DIMA, int PLA, bool BR, int DIMB, int PLB, int TM, int TN>
__device__ __forceinline__ void load_frags_x3(const bf16_t* As, const bf16_t* Bs, int a0, int b0, int lane, FragsX3<TM, TN>& f) {
#pragma unroll
  for (int pl = 0; pl < 3; ++pl) {
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) f.a[pl][tm] = frag_x<AR, DIMA>(As + pl * PLA, a0 + tm * 32, lane);
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) f.b[pl][tn] = frag_x<BR, DIMB>(Bs + pl * PLB, b0 + tn * 32, lane);
  }
}

template <bool SYM, int TM, int TN>      // the six products of one tile from fragment arrays (the last tile of the ring form)
__device__ __forceinline__ void mma_x3_ab(const bf16x8 (&a)[3][TM], const bf16x8 (&b)[3][TN], f32x16 (&acc)[TM][TN],
                                          f32x16 (&accb)[SYM ? TM : 1][SYM ? TN : 1], f32x16 (&accc)[SYM ? TM : 1][SYM ? TN : 1]) {
  constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};      // l h, h l, m m, m h, h m, h h
  constexpr int TG[6] = {0, 1, 2, 0, 1, 2};
#pragma unroll
  for (int p = 0; p < 6; ++p)
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
      for (int tn = 0; tn < TN; ++tn) {
        if constexpr (SYM) {
          if (TG[p] == 0) acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[PA[p]][tm], b[PB[p]][tn], acc[tm][tn], 0, 0, 0);
          else if (TG[p] == 1) accb[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[PA[p]][tm], b[PB[p]][tn], accb[tm][tn], 0, 0, 0);
          else accc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[PA[p]][tm], b[PB[p]][tn], accc[tm][tn], 0, 0, 0);
        } else {
          acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[PA[p]][tm], b[PB[p]][tn], acc[tm][tn], 0, 0, 0);
        }
      }
}

template <bool SYM, int TM, int TN, class F>
__device__ __forceinline__ void mma_x3_f(const FragsX3<TM, TN>& f, f32x16 (&acc)[TM][TN], f32x16 (&accb)[SYM ? TM : 1][SYM ? TN : 1],
                                         f32x16 (&accc)[SYM ? TM : 1][SYM ? TN : 1], F&& filler) {
  constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};      // l h, h l, m m, m h, h m, h h
  constexpr int TG[6] = {0, 1, 2, 0, 1, 2};
#pragma unroll
  for (int p = 0; p < 6; ++p) {
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
      for (int tn = 0; tn < TN; ++tn) {
        if constexpr (SYM) {
          if (TG[p] == 0) acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[PA[p]][tm], f.b[PB[p]][tn], acc[tm][tn], 0, 0, 0);
          else if (TG[p] == 1) accb[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[PA[p]][tm], f.b[PB[p]][tn], accb[tm][tn], 0, 0, 0);
          else accc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[PA[p]][tm], f.b[PB[p]][tn], accc[tm][tn], 0, 0, 0);
        } else {
          acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[PA[p]][tm], f.b[PB[p]][tn], acc[tm][tn], 0, 0, 0);
        }
      }
    filler(p);
  }
}

template <int S, bool SYM, int TM, int TN, class SA, class SB, class LA, class LB, bool AR, int DIMA, bool BR, int DIMB>
__device__ __forceinline__ void x3_iter_fp(SA& sa, SB& sb, const LA& la, const LB& lb, bf16_t* smem, int cur, int red_fetch, int a0, int b0,
                                           const FragsX3<TM, TN>& fc, FragsX3<TM, TN>& fn, f32x16 (&acc)[TM][TN],
                                           f32x16 (&accb)[SYM ? TM : 1][SYM ? TN : 1], f32x16 (&accc)[SYM ? TM : 1][SYM ? TN : 1], int tid) {
  constexpr int STAGE = SA::LDS_ELEMS + SB::LDS_ELEMS;
  constexpr int NP = SA::NV + SB::NV;
  static_assert(NP <= 5, "at least one product group behind the barrier");
  bf16_t* An = smem + (cur ^ 1) * STAGE;
  bf16_t* Bn = An + SA::LDS_ELEMS;
  mma_x3_f<SYM, TM, TN>(fc, acc, accb, accc, [&](int p) {
    if (p < SA::NV) {
      sa.template store_piece<S>(la, An, tid, p);
      sa.template fetch_piece<S>(la, red_fetch, tid, p);
      if (p == SA::NV - 1) sa.template fetch_head<S>(la, red_fetch, tid);
    } else if (p < NP) {
      sb.template store_piece<S>(lb, Bn, tid, p - SA::NV);
      sb.template fetch_piece<S>(lb, red_fetch, tid, p - SA::NV);
      if (p == NP - 1) sb.template fetch_head<S>(lb, red_fetch, tid);
    }
    if (p == NP - 1) {
      __syncthreads();
      load_frags_x3<AR, DIMA, SA::PLANE, BR, DIMB, SB::PLANE>(An, Bn, a0, b0, tid & 63, fn);
    }
  });
}

// ---- the hand-placed iteration (T3D_X3_IL, round 6) ----------------------------------------------------------------------------------
// x3_iter_fp hands hipcc a product group (TM x TN MFMAs) and then a whole staging piece (~35 vector instructions); the scheduler merges
// groups further, and the loop it emits has MFMA bursts of 8-12 and runs of 30-67 vector instructions with no MFMA between them
// (tools/isa_loops.py on round 5's listing).  A wave issues in order: through a burst its vector work cannot start, through a run the
// matrix pipe has nothing of this wave's to do, and the SIMD's other wave runs the same program.  What the pipe tolerates beside an
// MFMA is <= 5-6 single-issue vector instructions per 32-cycle gap (MI355X_MICROARCH.md, 'vector-instruction ISSUE cost'; tools/micro/
// mfma_valu_il.hip: [MFMA, 6 x v_fma_f32] at two waves per SIMD runs at 35.5 cycles per MFMA).  So the iteration is written out as
//     MFMA_0 | steps of gap 0 | MFMA_1 | steps of gap 1 | ...                    (every `|` a sched_barrier(0): nothing crosses)
// where the steps are, in order: the micro-steps of every staging piece (StagerX3::il_step: <= 8 vector instructions each, the piece's
// three plane stores and its slot's refill in the last one), the workgroup barrier, and the fragment reads of the next tile in the
// order the next iteration's products need them (l h, h l, m m, ...: a2 b0 a0 b2 a1 b1).  ILSched spreads the steps over the gaps by
// their issue cost at compile time.  Same MFMAs on the same operands in the same order as x3_iter_fp: bit-identical results.
template <class SA, class SB, int TM, int TN>
struct ILSched {
  static constexpr int NM = 6 * TM * TN;                       // MFMAs = gaps of one iteration
  // T3D_X3_LATE_M: the m-plane fragments of a tile (first needed by product 2, MFMA 2 TM TN) are read at the HEAD of the iteration that
  // multiplies them, into the fragment set it is consuming, instead of with the other planes behind the barrier of the iteration before:
  // 12 instead of 18 LDS reads queue behind the barrier (the four waves of a workgroup pass it together; SQ_LDS_CMD_FIFO_FULL and
  // SQ_WAIT_INST_LDS doubled when the hand-placed iteration packed the 18 reads into four gaps), 6 are spread over the first gaps.
  static constexpr bool GB = FromGlobal<SB>::value;            // the second operand's fragments come from global memory (StagerX3F)
  static constexpr int NG = GB ? 3 * TN : 0;                   // ... 3 TN loads of the NEXT tile's fragments, first in the list (they wait for nothing)
  static constexpr int NLA = (T3D_X3_IL && T3D_X3_LATE_M) ? TM : 0, NLB = (T3D_X3_IL && T3D_X3_LATE_M && !GB) ? TN : 0;
  static constexpr int NL = NLA + NLB;                         // late reads: a1[*] then b1[*] of the CURRENT tile
  static constexpr int NSA = SA::NV * SA::NU, NSB = SB::NV * SB::NU;
  static constexpr int L0 = 0;                                 // first late read: FIRST in the list (product 2 needs them at MFMA 2 TM TN)
  static constexpr int G0 = NL;                                // first global fragment load
  static constexpr int ST0 = NG + NL;                          // first staging step
  static constexpr int BAR = ST0 + NSA + NSB;                  // index of the barrier step
  // fragment reads of the NEXT tile from LDS behind the barrier, in groups a2 b0 a0 b2 [a1 b1] (b groups only when B is staged in LDS)
  static constexpr int NGRP = (NLA ? 2 : 3) * (GB ? 1 : 2);
  __host__ __device__ __forceinline__ static constexpr int grp_is_b(int g) { return GB ? 0 : (g & 1); }
  __host__ __device__ __forceinline__ static constexpr int grp_plane(int g) {
    const int o = GB ? g : g / 2;                              // 0, 1, 2 -> the pair (a2 b0), (a0 b2), (a1 b1)
    return o == 0 ? (grp_is_b(g) ? 0 : 2) : o == 1 ? (grp_is_b(g) ? 2 : 0) : 1;
  }
  __host__ __device__ __forceinline__ static constexpr int grp_n(int g) { return grp_is_b(g) ? TN : TM; }
  __host__ __device__ __forceinline__ static constexpr int count_nf() { int n = 0; for (int g = 0; g < NGRP; ++g) n += grp_n(g); return n; }
  static constexpr int NF = count_nf();
  static constexpr int NSTEP = BAR + 1 + NF;
  __host__ __device__ __forceinline__ static constexpr int cost(int k) {
    if (k < NL) return 2;
    if (k < ST0) return 6;
    if (k < ST0 + NSA) return SA::il_cost((k - ST0) % SA::NU);
    if (k < BAR) return SB::il_cost((k - ST0 - NSA) % SB::NU);
    if (k == BAR) return 0;
    return 2;
  }
  __host__ __device__ __forceinline__ static constexpr int total() { int w = 0; for (int k = 0; k < NSTEP; ++k) w += cost(k); return w; }
  __host__ __device__ __forceinline__ static constexpr int gap_of(int k) {      // by the midpoint of the step's share of the total cost
    int before = 0;
    for (int j = 0; j < k; ++j) before += cost(j);
    const int g = (2 * before + cost(k)) * NM / (2 * total());
    return g < NM ? g : NM - 1;
  }
  __host__ __device__ __forceinline__ static constexpr int first_step(int gap) { int k = 0; while (k < NSTEP && gap_of(k) < gap) ++k; return k; }
  // fragment read j behind the barrier: its group and its 32-wide block
  __host__ __device__ __forceinline__ static constexpr int frag_group(int j) {
    int g = 0;
    while (true) { const int n = grp_n(g); if (j < n) return g; j -= n; ++g; }
  }
  __host__ __device__ __forceinline__ static constexpr int frag_index(int j) {
    int g = 0;
    while (true) { const int n = grp_n(g); if (j < n) return j; j -= n; ++g; }
  }
};

template <int S, bool SYM, int TM, int TN, class SA, class SB, class LA, class LB, bool AR, int DIMA, bool BR, int DIMB>
__device__ __forceinline__ void x3_iter_il(SA& sa, SB& sb, const LA& la, const LB& lb, bf16_t* smem, int cur, int red_next, int red_fetch, int a0, int b0,
                                           FragsX3<TM, TN>& fc, FragsX3<TM, TN>& fn, bf16x8 (&bcur)[3][TN], bf16x8 (&bload)[3][TN],
                                           unsigned& touch, int red_touch, f32x16 (&acc)[TM][TN],
                                           f32x16 (&accb)[SYM ? TM : 1][SYM ? TN : 1], f32x16 (&accc)[SYM ? TM : 1][SYM ? TN : 1], int tid) {
  // bcur: the B fragments this iteration multiplies; bload: where the B fragments it loads go.  LDS-staged B: fc.b / fn.b (the next
  // tile's).  Global fragment planes (T3D_X3_BRING): a ring of three sets, the loads run TWO tiles ahead (`red_next` is that tile's
  // offset) -- see gemm_mainloop_x3.
  using SC = ILSched<SA, SB, TM, TN>;
  // the late reads land in gaps well ahead of the first MFMA of product 2 (index 2 TM TN), which multiplies them: a schedule that put
  // them behind the global fragment loads had the eight-wave tiles multiply the m planes of the tile before last
  static_assert(SC::NL == 0 || SC::gap_of(SC::NL - 1) + 1 < 2 * TM * TN, "late m-plane reads too late for product 2");
  constexpr int STAGE = SA::LDS_ELEMS + SB::LDS_ELEMS;
  bf16_t* An = smem + (cur ^ 1) * STAGE;
  bf16_t* Bn = An + SA::LDS_ELEMS;
  const bf16_t* Ac = smem + cur * STAGE;
  const bf16_t* Bc = Ac + SA::LDS_ELEMS;
  const int lane = tid & 63;
  // T3D_X3_COEF_LDS: with the second operand read from global memory the LDS its image would take holds the first operand's coefficient
  // table (gemm_mainloop_x3 fills it); derived from `smem` here so that the reads are ds_read, not flat
  constexpr bool CTAB = T3D_X3_COEF_LDS && SC::GB && AR && HasCtab<LA>::value;
  const float* ctab = reinterpret_cast<const float*>(smem + 2 * STAGE);
#if T3D_X3_BTOUCH
  if constexpr (SC::GB) {
    // Every workgroup of an XCD asks for the same k-tile of the weight fragments at about the same time, and the planes were written by
    // another launch: the first request of a line misses the XCD's L2, and with the fragment loads one tile ahead that miss sat on the
    // critical path of every iteration (multiply-only timing builds: 48.8 us with global fragments against 37.4 with the LDS-staged
    // operand, profiles/r06_il_ablations_frag.log).  One dword load per lane, one lane per 128-byte line of the fragments of the tile
    // T3D_X3_BTOUCH tiles ahead, starts the fill early; its value is "used" one iteration later (an empty asm statement, so that the
    // wait-count pass sees a use it can place a counted wait for -- by then long satisfied).
    asm volatile("" :: "v"(touch));
    touch = sb.template touch<TN>(lb, b0, red_touch, lane);
  }
#endif
  static_for<SC::NM>([&](auto ic) {
    constexpr int i = decltype(ic)::value;
    constexpr int p = i / (TM * TN), tm = (i / TN) % TM, tn = i % TN;
    constexpr int pa = p == 0 ? 2 : (p == 2 || p == 3) ? 1 : 0;      // l h, h l, m m, m h, h m, h h
    constexpr int pb = p == 1 ? 2 : (p == 2 || p == 4) ? 1 : 0;
    // (the MFMA between two volatile asm statements on its accumulator: see StagerX3::il_step)
    if constexpr (SYM) {
      if constexpr (p % 3 == 0) {
        acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fc.a[pa][tm], bcur[pb][tn], acc[tm][tn], 0, 0, 0);
        T3D_PIN1(acc[tm][tn]);
      } else if constexpr (p % 3 == 1) {
        accb[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fc.a[pa][tm], bcur[pb][tn], accb[tm][tn], 0, 0, 0);
        T3D_PIN1(accb[tm][tn]);
      } else {
        accc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fc.a[pa][tm], bcur[pb][tn], accc[tm][tn], 0, 0, 0);
        T3D_PIN1(accc[tm][tn]);
      }
    } else {
      acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fc.a[pa][tm], bcur[pb][tn], acc[tm][tn], 0, 0, 0);
      T3D_PIN1(acc[tm][tn]);
    }
    __builtin_amdgcn_sched_barrier(0);
    constexpr int k0 = SC::first_step(i), k1 = SC::first_step(i + 1);
    static_for<k1 - k0>([&](auto jc) {
      constexpr int k = k0 + decltype(jc)::value;
      if constexpr (k < SC::NL) {                  // the current tile's m planes (T3D_X3_LATE_M)
        constexpr int q = k - SC::L0;
        asm volatile("" ::: "memory");
        if constexpr (q < TM) fc.a[1][q] = frag_x<AR, DIMA>(Ac + SA::PLANE, a0 + q * 32, lane);
        else bcur[1][q - TM] = frag_x<BR, DIMB>(Bc + SB::PLANE, b0 + (q - TM) * 32, lane);
        asm volatile("" ::: "memory");
      } else if constexpr (k < SC::ST0) {          // the NEXT tile's weight fragments, straight from global memory (WLoaderX3F)
        constexpr int kg = k - SC::G0, o = kg / TN, x = kg % TN, pl = o == 0 ? 0 : o == 1 ? 2 : 1;      // b0, b2, b1: the order the next products need them
        if constexpr (SC::GB) {
#ifdef T3D_ABL_IL_NOGB      // timing ablation (wrong results): the weight fragments are not reloaded
          bload[pl][x] = bcur[pl][x];
#else
          asm volatile("" ::: "memory");
          bload[pl][x] = sb.gfrag(lb, pl, b0 + x * 32, red_next, lane);
          asm volatile("" ::: "memory");
#endif
        }
      } else if constexpr (k < SC::ST0 + SC::NSA) {
#ifndef T3D_ABL_IL_NOSTAGE      // timing ablations (wrong results): no staging pass / no barrier / no fragment reads
        sa.template il_step<S, (k - SC::ST0) / SA::NU, (k - SC::ST0) % SA::NU, CTAB>(la, An, tid, red_fetch, ctab);
#endif
      } else if constexpr (k < SC::BAR) {
#ifndef T3D_ABL_IL_NOSTAGE
        sb.template il_step<S, (k - SC::ST0 - SC::NSA) / SB::NU, (k - SC::ST0 - SC::NSA) % SB::NU>(lb, Bn, tid, red_fetch);
#endif
      } else if constexpr (k == SC::BAR) {
#ifndef T3D_ABL_IL_NOBAR
        __syncthreads();
#endif
      } else {
        constexpr int j = k - SC::BAR - 1, g = SC::frag_group(j), x = SC::frag_index(j);
        constexpr int pl = SC::grp_plane(g);
        constexpr bool isb = SC::grp_is_b(g) != 0;
#ifdef T3D_ABL_IL_NOFRAG
        if constexpr (!isb) fn.a[pl][x] = fc.a[pl][x]; else bload[pl][x] = bcur[pl][x];
#else
        asm volatile("" ::: "memory");
        if constexpr (!isb) fn.a[pl][x] = frag_x<AR, DIMA>(An + pl * SA::PLANE, a0 + x * 32, lane);
        else bload[pl][x] = frag_x<BR, DIMB>(Bn + pl * SB::PLANE, b0 + x * 32, lane);
        asm volatile("" ::: "memory");
#endif
      }
    });
    __builtin_amdgcn_sched_barrier(0);
  });
}

// Two LDS stages, one barrier per 16-deep k-tile, PF register slots per operand.  A k-tile is only 24 MFMAs of 32 cycles per wave
// (0.3 us): with ONE tile in flight the loop ran at one k-tile per memory round trip (1.7 us per k-tile measured on 512 -> 256, the
// matrix pipe a third busy; splitting the weights beforehand or a third workgroup per CU changed nothing -- the loop was waiting for its
// loads).  Tile j travels in slot j % PF: iteration t multiplies tile t from LDS stage t & 1, stores tile t + 1 (requested PF iterations
// earlier) into the other stage between the MFMAs, and refills that slot with tile t + 1 + PF.
template <int S, bool SYM, int TM, int TN, class SA, class SB, class LA, class LB, bool AR, int DIMA, bool BR, int DIMB>
__device__ __forceinline__ void x3_iter(SA& sa, SB& sb, const LA& la, const LB& lb, bf16_t* smem, int cur, int red_fetch, int a0, int b0,
                                        f32x16 (&acc)[TM][TN], f32x16 (&accb)[SYM ? TM : 1][SYM ? TN : 1],
                                        f32x16 (&accc)[SYM ? TM : 1][SYM ? TN : 1], int tid) {
  constexpr int STAGE = SA::LDS_ELEMS + SB::LDS_ELEMS;
  const bf16_t* As = smem + cur * STAGE;
  const bf16_t* Bs = As + SA::LDS_ELEMS;
  bf16_t* An = smem + (cur ^ 1) * STAGE;
  bf16_t* Bn = An + SA::LDS_ELEMS;
#if T3D_X3_PIECEWISE
  mma_x3<SYM, TM, TN, AR, DIMA, SA::PLANE, BR, DIMB, SB::PLANE>(As, Bs, a0, b0, acc, accb, accc, tid & 63, [&](int p) {
    if (p < SA::NV) {
      sa.template store_piece<S>(la, An, tid, p);
      sa.template fetch_piece<S>(la, red_fetch, tid, p);
      if (p == SA::NV - 1) sa.template fetch_head<S>(la, red_fetch, tid);
    } else if (p - SA::NV < SB::NV) {
      sb.template store_piece<S>(lb, Bn, tid, p - SA::NV);
      sb.template fetch_piece<S>(lb, red_fetch, tid, p - SA::NV);
      if (p - SA::NV == SB::NV - 1) sb.template fetch_head<S>(lb, red_fetch, tid);
    }
  });
#if T3D_X3_SGB
  // Prescribe the interleave of this iteration's instructions: hipcc clumps the staging pass into runs of 11-13 VALU instructions
  // with no MFMA between them (the matrix pipe idles through each run), and the four MFMAs of a product group back to back (the wave
  // cannot issue the VALU work behind them until the fourth has been accepted).  One MFMA, then its share of the VALU / LDS work.
#pragma unroll
  for (int i = 0; i < 6 * TM * TN; ++i) {
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);               // one MFMA
    __builtin_amdgcn_sched_group_barrier(0x002, T3D_X3_SGB, 0);      // VALU
    __builtin_amdgcn_sched_group_barrier(0x080, 1, 0);               // one LDS access (fragment read / plane write)
  }
#endif
#else
  mma_x3<SYM, TM, TN, AR, DIMA, SA::PLANE, BR, DIMB, SB::PLANE>(As, Bs, a0, b0, acc, accb, accc, tid & 63, [&](int p) {
    if (p < SA::NV) sa.template store_piece<S>(la, An, tid, p);
    else if (p - SA::NV < SB::NV) sb.template store_piece<S>(lb, Bn, tid, p - SA::NV);
  });
  __builtin_amdgcn_sched_barrier(0);      // the new loads stay behind every wait on the older ones (vmcnt counts in issue order)
  sa.template fetch<S>(la, red_fetch, tid);
  sb.template fetch<S>(lb, red_fetch, tid);
#endif
  __syncthreads();
}

template <bool SYM, int TM, int TN, class SA, class SB, class LA, class LB, bool AR, int DIMA, bool BR, int DIMB>
__device__ __forceinline__ void gemm_mainloop_x3(SA& sa, SB& sb, const LA& la, const LB& lb, float* smem_f, int red_begin, int red_end,
                                                 int a0, int b0, f32x16 (&acc)[TM][TN], int tid) {
  static_assert(!SYM || TM * TN == 1, "symmetric accumulation: 64 x 64 tiles (three accumulator sets)");
  static_assert(SA::PF == SB::PF && SA::PF >= 1 && SA::PF <= 4, "one to four register slots");
  T3D_MFMA_IN_AGPRS();
  constexpr int PF = SA::PF;
  f32x16 accb[SYM ? TM : 1][SYM ? TN : 1], accc[SYM ? TM : 1][SYM ? TN : 1];
  if constexpr (SYM) { zero_acc<TM, TN>(accb); zero_acc<TM, TN>(accc); }
  bf16_t* smem = reinterpret_cast<bf16_t*>(smem_f);
  constexpr int STAGE = SA::LDS_ELEMS + SB::LDS_ELEMS;
  static_assert(SA::NV + SB::NV <= 6, "staging pieces must fit the six product groups");
  const int last = red_end - BKX;
#ifdef T3D_ABL_X3_SAMETILE      // timing ablation (wrong results): every k-tile re-reads the first one (cache hits: no memory latency)
  auto tile_red = [&](int j) { return min(red_begin + (j & 1) * BKX, last); };
#else
  auto tile_red = [&](int j) { return min(red_begin + j * BKX, last); };      // past the end the last tile is re-read, never used
#endif
  sa.template fetch<0>(la, tile_red(0), tid);
  sb.template fetch<0>(lb, tile_red(0), tid);
  if constexpr (PF > 1) { sa.template fetch<1>(la, tile_red(1), tid); sb.template fetch<1>(lb, tile_red(1), tid); }
  if constexpr (PF > 2) { sa.template fetch<2>(la, tile_red(2), tid); sb.template fetch<2>(lb, tile_red(2), tid); }
  if constexpr (PF > 3) { sa.template fetch<3>(la, tile_red(3), tid); sb.template fetch<3>(lb, tile_red(3), tid); }
  sa.template store<0>(la, smem, tid);
  sb.template store<0>(lb, smem + SA::LDS_ELEMS, tid);
  sa.template fetch<0>(la, tile_red(PF), tid);
  sb.template fetch<0>(lb, tile_red(PF), tid);
#if T3D_X3_IL && T3D_X3_COEF_LDS
  if constexpr (FromGlobal<SB>::value && AR && HasCtab<LA>::value) {
    // the launcher sized the dynamic LDS for an image of the second operand as well (lds_fwd_x3 / lds_dgrad_x3); the fragment form does
    // not use it: bytes behind the two stages, at least those of that image
    constexpr int FREE_BYTES = 2 * 3 * (BR ? DIMB * LDRX : BKX * (DIMB + LDCX_PAD)) * 2;
    (void)FREE_BYTES;      // >= 18 KB for every tiling; the launchers take the fragment form for tables of <= 1536 channels only (t3d_x3_fwd ...)
    la.ctab_fill(reinterpret_cast<float*>(smem + 2 * STAGE), tid, (int)blockDim.x);
  }
#endif
  __syncthreads();
#if defined(T3D_TRACE) && T3D_TRACE_STRIDE >= 8
  T3D_TRACE_MARK(4);
#endif
  const int nt = (red_end - red_begin) / BKX;
  int cur = 0;
#if T3D_X3_FRAGPF
  static_assert(PF == 1 || PF == 2, "fragments across the barrier: one or two register slots");
  {
#if T3D_X3_IL
#define T3D_X3_ITER_FP(S_, T_, FC_, FN_) \
  x3_iter_il<S_, SYM, TM, TN, SA, SB, LA, LB, AR, DIMA, BR, DIMB>(sa, sb, la, lb, smem, cur, tile_red((T_) + 1), tile_red((T_) + 1 + PF), a0, b0, FC_, FN_, (FC_).b, (FN_).b, btouch, tile_red((T_) + (T3D_X3_BTOUCH > 0 ? T3D_X3_BTOUCH : 1)), acc, accb, accc, tid)
#else
#define T3D_X3_ITER_FP(S_, T_, FC_, FN_) \
  x3_iter_fp<S_, SYM, TM, TN, SA, SB, LA, LB, AR, DIMA, BR, DIMB>(sa, sb, la, lb, smem, cur, tile_red((T_) + 1 + PF), a0, b0, FC_, FN_, acc, accb, accc, tid)
#endif
    constexpr int SODD = PF == 2 ? 1 : 0;      // the slot of tile t + 1 for even t
    FragsX3<TM, TN> f0, f1;
    unsigned btouch = 0u;                      // (T3D_X3_BTOUCH: the value of the last prefetch touch, see x3_iter_il)
#if T3D_X3_IL && T3D_X3_BRING
    if constexpr (FromGlobal<SB>::value) {
      // The weight fragments come straight from global memory (WLoaderX3F) and every workgroup of an XCD asks for the same k-tile of
      // them at the same time: the first request of a k-tile misses the XCD's L2 (the planes were written by another launch), and with
      // the loads ONE tile ahead that miss was on the critical path of every iteration -- the multiply-only timing build of the
      // fragment kernels ran 48.8 us against 37.4 for the LDS-staged form (profiles/r06_il_ablations_frag.log).  So the B fragments
      // live in a ring of THREE register sets and are loaded TWO tiles ahead (3 TN more fragment registers); the A fragments keep their
      // two sets.  Periods 2 and 3: the steady-state body is six iterations with compile-time set indices.
      static_assert(PF == 1, "the B ring is written for one register slot per staged operand");
      bf16x8 br[3][3][TN];
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) {
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) f0.a[pl][tm] = frag_x<AR, DIMA>(smem + pl * SA::PLANE, a0 + tm * 32, tid & 63);
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) {
          br[0][pl][tn] = sb.gfrag(lb, pl, b0 + tn * 32, tile_red(0), tid & 63);
          br[1][pl][tn] = sb.gfrag(lb, pl, b0 + tn * 32, tile_red(1), tid & 63);
        }
      }
      int t = 0;
#define T3D_X3_ITER_R(PH_)                                                                                                               \
  do {                                                                                                                                   \
    x3_iter_il<0, SYM, TM, TN, SA, SB, LA, LB, AR, DIMA, BR, DIMB>(sa, sb, la, lb, smem, cur, tile_red(t + (PH_) + 2), tile_red(t + (PH_) + 2), \
        a0, b0, ((PH_) & 1) ? f1 : f0, ((PH_) & 1) ? f0 : f1, br[(PH_) % 3], br[((PH_) + 2) % 3], btouch, tile_red(t + (PH_) + 4), acc, accb, accc, tid);                  \
    cur ^= 1;                                                                                                                            \
  } while (0)
      for (; t + 6 < nt; t += 6) {
        T3D_X3_ITER_R(0); T3D_X3_ITER_R(1); T3D_X3_ITER_R(2); T3D_X3_ITER_R(3); T3D_X3_ITER_R(4); T3D_X3_ITER_R(5);
      }
      // one to six tiles left: an iteration for each but the last
      if (t + 1 < nt) T3D_X3_ITER_R(0);
      if (t + 2 < nt) T3D_X3_ITER_R(1);
      if (t + 3 < nt) T3D_X3_ITER_R(2);
      if (t + 4 < nt) T3D_X3_ITER_R(3);
      if (t + 5 < nt) T3D_X3_ITER_R(4);
#undef T3D_X3_ITER_R
      const int ph = nt - 1 - t;      // phase of the last tile (0 ... 5); `cur` is its LDS stage
      auto fin = [&](FragsX3<TM, TN>& f, bf16x8 (&b)[3][TN]) {
#if T3D_X3_LATE_M
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) f.a[1][tm] = frag_x<AR, DIMA>(smem + cur * STAGE + SA::PLANE, a0 + tm * 32, tid & 63);
#endif
        mma_x3_ab<SYM, TM, TN>(f.a, b, acc, accb, accc);
      };
      if (ph == 0) fin(f0, br[0]);
      else if (ph == 1) fin(f1, br[1]);
      else if (ph == 2) fin(f0, br[2]);
      else if (ph == 3) fin(f1, br[0]);
      else if (ph == 4) fin(f0, br[1]);
      else fin(f1, br[2]);
    } else
#endif
    if constexpr (FromGlobal<SB>::value) {      // (A from the LDS image, the weight fragments from global memory)
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) {
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) f0.a[pl][tm] = frag_x<AR, DIMA>(smem + pl * SA::PLANE, a0 + tm * 32, tid & 63);
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) f0.b[pl][tn] = sb.gfrag(lb, pl, b0 + tn * 32, tile_red(0), tid & 63);
      }
    } else {
      load_frags_x3<AR, DIMA, SA::PLANE, BR, DIMB, SB::PLANE>(smem, smem + SA::LDS_ELEMS, a0, b0, tid & 63, f0);
    }
    int t = 0;
#if T3D_X3_FAIR
    // The two workgroups of a CU do not advance together: a SIMD's arbiter takes the OLDER wave whenever both are ready, so the workgroup
    // dispatched first wins every conflict (matrix pipe, vector ports, LDS, the memory pipeline), finishes its k loop early -- in 256 of
    // 256 CU pairs of the forward 512 -> 256, by 7.6 us of 46 (tools/trace_blocks.py, profiles/r06_trace_blocks_x3.log) -- and leaves the
    // other one alone on the CU, one wave per SIMD with nobody to overlap its staging pass, for the rest of the launch.  A launch's
    // first T3D_FAIR_CUS workgroups land one per CU; the later ones are the younger halves of the pairs (and, in launches of more rounds,
    // always younger than the workgroup they join).  They run the first T3D_X3_FAIR_NUM / T3D_X3_FAIR_DEN of their k loop at wave
    // priority 1 -- priority beats age -- and the rest at 0: the younger workgroup leads first, the older one catches up, both leave
    // the loop together.  Scheduling only: the instruction streams and the results are unchanged.
    // In the fused backward launches the rule is another one: the weight-gradient tiles (both operands reduced over rows: !AR && !BR) are few,
    // long and first in the launch -- one per CU for 42 ... 86 us while two to four rounds of data-gradient tiles pass through the CU's other
    // slot, and in every layer but the widest the launch ends when THEY end, alone on their CUs (tools/trace_bwd.py).  They run their whole k
    // loop at priority 1 (T3D_X3_PRIO_WGRAD); the data-gradient tiles (AR && BR) keep priority 0 and no forward rule.
    constexpr bool WG_TILE = T3D_X3_PRIO_WGRAD && !AR && !BR, FWD_TILE = AR && !BR;
    const int fair_sw = (FWD_TILE && (int)blockIdx.x >= T3D_FAIR_CUS) ? ((nt * T3D_X3_FAIR_NUM / T3D_X3_FAIR_DEN) & ~1) : 0;
    if (WG_TILE || fair_sw > 0) __builtin_amdgcn_s_setprio(1);
#endif
#if T3D_X3_IL && T3D_X3_BRING
    if constexpr (!FromGlobal<SB>::value)
#endif
#if T3D_X3_FAIR
    // (two passes over ONE loop body -- the k-tiles in front of the switch, then the rest -- so that the body stays a single basic block)
#pragma nounroll
    for (int pass = 0; pass < 2; ++pass) {
      const int lim = pass == 0 ? fair_sw : nt;
      for (; t + 2 < nt && t < lim; t += 2) {
        T3D_X3_ITER_FP(SODD, t, f0, f1);
        cur ^= 1;
        T3D_X3_ITER_FP(0, t + 1, f1, f0);
        cur ^= 1;
      }
      if constexpr (!WG_TILE) __builtin_amdgcn_s_setprio(0);
    }
#else
    for (; t + 2 < nt; t += 2) {
      T3D_X3_ITER_FP(SODD, t, f0, f1);
      cur ^= 1;
      T3D_X3_ITER_FP(0, t + 1, f1, f0);
      cur ^= 1;
    }
#endif
    // (T3D_X3_LATE_M: the m planes of the last tile are read here -- no iteration follows that would read them at its head)
    auto load_m = [&](FragsX3<TM, TN>& f, const bf16_t* st) {
#if T3D_X3_IL && T3D_X3_LATE_M
#pragma unroll
      for (int tm = 0; tm < TM; ++tm) f.a[1][tm] = frag_x<AR, DIMA>(st + SA::PLANE, a0 + tm * 32, tid & 63);
      if constexpr (!FromGlobal<SB>::value) {
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) f.b[1][tn] = frag_x<BR, DIMB>(st + SA::LDS_ELEMS + SB::PLANE, b0 + tn * 32, tid & 63);
      }
#endif
    };
#if T3D_X3_IL && T3D_X3_BRING
    if constexpr (!FromGlobal<SB>::value)
#endif
    {
      if (t + 1 < nt) {      // two tiles left
        T3D_X3_ITER_FP(SODD, t, f0, f1);
        load_m(f1, smem + (cur ^ 1) * STAGE);
        mma_x3_f<SYM, TM, TN>(f1, acc, accb, accc, [](int) {});
      } else {
        load_m(f0, smem + cur * STAGE);
        mma_x3_f<SYM, TM, TN>(f0, acc, accb, accc, [](int) {});
      }
    }
#undef T3D_X3_ITER_FP
#if T3D_X3_FAIR
    if constexpr (T3D_X3_PRIO_WGRAD && !AR && !BR) __builtin_amdgcn_s_setprio(0);
#endif
  }
#else
#define T3D_X3_ITER(S_) x3_iter<S_, SYM, TM, TN, SA, SB, LA, LB, AR, DIMA, BR, DIMB>(sa, sb, la, lb, smem, cur, tile_red(t + 1 + PF), a0, b0, acc, accb, accc, tid)
  constexpr int S1 = (PF > 1) ? 1 : 0, S2 = (PF > 2) ? 2 : 0, S3 = (PF > 3) ? 3 : 0;
  int t = 0;
  if constexpr (PF == 2) {      // slot and LDS stage have the same period: a branch-free body of two iterations
    for (; t + 2 < nt; t += 2) {
      T3D_X3_ITER(1);
      cur ^= 1;
      ++t; T3D_X3_ITER(0); --t;
      cur ^= 1;
    }
  }
  for (; t + 1 < nt; ++t) {      // (the slot is workgroup-uniform: a scalar branch per k-tile)
    const int slot = (t + 1) % PF;
    if (PF == 1 || slot == 0) T3D_X3_ITER(0);
    else if (PF == 2 || slot == 1) T3D_X3_ITER(S1);
    else if (PF == 3 || slot == 2) T3D_X3_ITER(S2);
    else T3D_X3_ITER(S3);
    cur ^= 1;
  }
#undef T3D_X3_ITER
  {
    const bf16_t* As = smem + cur * STAGE;
    mma_x3<SYM, TM, TN, AR, DIMA, SA::PLANE, BR, DIMB, SB::PLANE>(As, As + SA::LDS_ELEMS, a0, b0, acc, accb, accc, tid & 63, [](int) {});
  }
#endif
  if constexpr (SYM) {
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
      for (int tn = 0; tn < TN; ++tn)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[tm][tn][r] = (acc[tm][tn][r] + accb[tm][tn][r]) + accc[tm][tn][r];
  }
  __syncthreads();
}


// ---- producer / consumer wave roles (PathX3PC; 512-thread workgroups) -- an EXPERIMENT of round 5, not the default ---------------------
// The loop above gives every wave both jobs -- load, batch-norm / ReLU, three-way split, LDS stores AND the six products -- in ONE in-order
// instruction stream: ~135 vector instructions per wave and 16-deep k-tile queue behind and in front of 24 MFMAs, and the counters say
// the matrix pipe idles 0.6 of a launch while a SIMD has nothing to issue 0.4 of it (docs/EXPERIMENTS.md, round 4).  Here a workgroup is
// EIGHT waves: waves 0-3 (consumers, tid 0..255: the 2 x 2 wave grid of the output tile, the only ones with accumulators and an
// epilogue) issue nothing but fragment reads and MFMAs; waves 4-7 (producers) run the staging pass of the whole tile and end.  A SIMD
// hosts one wave of each role.  Three LDS stages (101 KB: ONE workgroup per CU), one workgroup barrier per k-tile, tile j in stage j % 3:
//     iteration t   consumers: read the fragments of tile t + 1 into the other register set, multiply tile t from registers
//                   producers: convert + store tile t + 2 (register slot (t + 2) % 3), request tile t + 5 into that slot
//     barrier t + 1: tile t + 2 is complete; every fragment of tile t has been consumed
// Stage (t + 2) % 3 held tile t - 1, whose fragments were read in iteration t - 2: no wave can still be reading it.  Same products in the
// same order as the loop above: results are bit-identical (tests/test_kernels_gpu.py).
// MEASURED (tools/bench_x3_pc.py, MI355X, forward 512 -> 256 at M = 32768): 78-83 us against 56-58 us for the loop above.  Taken apart
// with timing ablations (tools/pc_abl.sh): the consumers alone (producers only keeping the barrier count) 53 us; the producers alone
// 52 us; the products alone, no fragment reads and no barrier, 41-44 us -- of which the chip's matrix pipe accounts for 27.7 us (it
// sustains 1.86 PFLOP/s dense bf16 at the 1.85 GHz it holds under this load, tools/micro/mfma_rate.hip, not the 2.5 PFLOP/s of the
// data sheet) and the rest is the un-overlapped prologue and epilogue of the one workgroup a CU can hold.  The two roles of a SIMD do NOT
// run beside each other for free: together they take 1.5 x the longer of the two, whatever the priorities (s_setprio on either role:
// 78.4 / 82.9 us).  Kept behind T3D_X3_PC=1 for the measurement; nothing selects it.
constexpr int X3_RING = 3;
#ifdef T3D_ABL_PC_NOBAR      // timing ablation (wrong results; with _NOPROD and _NOFRAG: the products alone)
#define T3D_PC_BAR() do {} while (0)
#else
#define T3D_PC_BAR() __syncthreads()
#endif
template <bool SYM, int TM, int TN, class SA, class SB, class LA, class LB, bool AR, int DIMA, bool BR, int DIMB>
__device__ __forceinline__ void gemm_mainloop_x3_pc(SA& sa, SB& sb, const LA& la, const LB& lb, float* smem_f, int red_begin, int red_end,
                                                    int a0, int b0, f32x16 (&acc)[TM][TN], int tid) {
  static_assert(!SYM || TM * TN == 1, "symmetric accumulation: 64 x 64 tiles (three accumulator sets)");
  static_assert(SA::PF == X3_RING && SB::PF == X3_RING, "three register slots: slot and LDS stage of a tile share their period");
  bf16_t* smem = reinterpret_cast<bf16_t*>(smem_f);
  constexpr int STAGE = SA::LDS_ELEMS + SB::LDS_ELEMS;
  const int last = red_end - BKX;
  const int nt = (red_end - red_begin) / BKX;
  auto tile_red = [&](int j) { return min(red_begin + j * BKX, last); };      // past the end the last tile is re-read, never used
  const bool producer = __builtin_amdgcn_readfirstlane((int)threadIdx.x) >= NT;
#ifdef T3D_PC_PRIO      // 1: the consumers (matrix instructions) above the producers; 2: the producers above the consumers
  if ((T3D_PC_PRIO == 1) != producer) __builtin_amdgcn_s_setprio(2);
#endif
  if (producer) {
#ifdef T3D_ABL_PC_NOPROD      // timing ablation (wrong results): producers that only keep the barrier count
#define T3D_PC_PRODUCE(S_, J_) do {} while (0)
#else
#define T3D_PC_PRODUCE(S_, J_)                                                                         \
  do {                                                                                                 \
    bf16_t* At_ = smem + (S_) * STAGE;                                                                 \
    bf16_t* Bt_ = At_ + SA::LDS_ELEMS;                                                                 \
    const int rf_ = tile_red((J_) + X3_RING);                                                          \
    _Pragma("unroll") for (int q = 0; q < SA::NV; ++q) {                                               \
      sa.template store_piece<S_>(la, At_, tid, q);                                                    \
      sa.template fetch_piece<S_>(la, rf_, tid, q);                                                    \
    }                                                                                                  \
    sa.template fetch_head<S_>(la, rf_, tid);                                                          \
    _Pragma("unroll") for (int q = 0; q < SB::NV; ++q) {                                               \
      sb.template store_piece<S_>(lb, Bt_, tid, q);                                                    \
      sb.template fetch_piece<S_>(lb, rf_, tid, q);                                                    \
    }                                                                                                  \
    sb.template fetch_head<S_>(lb, rf_, tid);                                                          \
  } while (0)
#endif
    sa.template fetch<0>(la, tile_red(0), tid);
    sb.template fetch<0>(lb, tile_red(0), tid);
    sa.template fetch<1>(la, tile_red(1), tid);
    sb.template fetch<1>(lb, tile_red(1), tid);
    sa.template fetch<2>(la, tile_red(2), tid);
    sb.template fetch<2>(lb, tile_red(2), tid);
    T3D_PC_PRODUCE(0, 0);
    T3D_PC_PRODUCE(1, 1);
    T3D_PC_BAR();
    // Branch-free body of three iterations (a conditional staging pass is a join in front of which hipcc waits for nearly every load in
    // flight: `s_waitcnt vmcnt(2)` with twelve outstanding): tiles past the end are the last tile again, stored into stages whose tiles
    // have been consumed (tile nt + i lands on tile nt + i - 3), and the consumers pad their barrier count to the same multiple of three.
    for (int t = 0; t < nt; t += X3_RING) {
      T3D_PC_PRODUCE(2, t + 2);
      T3D_PC_BAR();
      T3D_PC_PRODUCE(0, t + 3);
      T3D_PC_BAR();
      T3D_PC_PRODUCE(1, t + 4);
      T3D_PC_BAR();
    }
#undef T3D_PC_PRODUCE
    __builtin_amdgcn_endpgm();      // a producer has no accumulators and no epilogue; s_barrier counts the surviving waves only
  }
  f32x16 accb[SYM ? TM : 1][SYM ? TN : 1], accc[SYM ? TM : 1][SYM ? TN : 1];
  if constexpr (SYM) { zero_acc<TM, TN>(accb); zero_acc<TM, TN>(accc); }
  FragsX3<TM, TN> f0, f1;
  const int lane = tid & 63;
  T3D_PC_BAR();
  load_frags_x3<AR, DIMA, SA::PLANE, BR, DIMB, SB::PLANE>(smem, smem + SA::LDS_ELEMS, a0, b0, lane, f0);
  int nxt = 1;      // LDS stage of tile t + 1
  // (the fragments of the tile behind the last one are read from a stage that holds a repeat of the last tile, and never used)
#ifdef T3D_ABL_PC_NOMMA       // timing ablations (wrong results): consumers without products / without fragment reads
#define T3D_PC_MMA(FC_) do {} while (0)
#else
#define T3D_PC_MMA(FC_) mma_x3_f<SYM, TM, TN>(FC_, acc, accb, accc, [](int) {})
#endif
#ifdef T3D_ABL_PC_NOFRAG
#define T3D_PC_LOADFRAGS(FN_) do {} while (0)
#else
#define T3D_PC_LOADFRAGS(FN_) load_frags_x3<AR, DIMA, SA::PLANE, BR, DIMB, SB::PLANE>(smem + nxt * STAGE, smem + nxt * STAGE + SA::LDS_ELEMS, a0, b0, lane, FN_)
#endif
#define T3D_PC_CONSUME(FC_, FN_)                                                                                                       \
  do {                                                                                                                                 \
    T3D_PC_LOADFRAGS(FN_);                                                                                                             \
    __builtin_amdgcn_sched_barrier(0);      /* the reads in front of the products (hipcc sinks them to the end, right before their wait) */ \
    T3D_PC_MMA(FC_);                                                                                                                   \
    __builtin_amdgcn_sched_barrier(0);      /* the barrier behind the products, not in front of them (hipcc hoists it over the MFMAs) */ \
    T3D_PC_BAR();                                                                                                                   \
    nxt = nxt == X3_RING - 1 ? 0 : nxt + 1;                                                                                            \
  } while (0)
  int t = 0;
  for (; t + 2 <= nt; t += 2) {
    T3D_PC_CONSUME(f0, f1);
    T3D_PC_CONSUME(f1, f0);
  }
  if (t < nt) { T3D_PC_CONSUME(f0, f1); ++t; }
#undef T3D_PC_CONSUME
#undef T3D_PC_MMA
#undef T3D_PC_LOADFRAGS
  for (const int tb = (nt + X3_RING - 1) / X3_RING * X3_RING; t < tb; ++t) T3D_PC_BAR();      // the producers' barrier count
  if constexpr (SYM) {
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
      for (int tn = 0; tn < TN; ++tn)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[tm][tn][r] = (acc[tm][tn][r] + accb[tm][tn][r]) + accc[tm][tn][r];
  }
}

// Arithmetic of a GEMM kernel: which staging / MFMA loop, and the element type T of the layer tensors it writes.
struct PathF32 {
  typedef float T;
  template <bool HAS_SUB, class XT> using Act = ActLoader<HAS_SUB>;
  template <bool POOLED> using Dy = DyLoader<POOLED>;
  typedef WLoaderT<float> WL;            // loader of the layer's weight matrix
  typedef WLoaderT<float> WLX;           // ... when every tile lies inside the matrix (the fp32 path keeps its one loader)
  static constexpr bool BF16 = false;
  static constexpr bool X3 = false;
  static constexpr bool PC = false;
  static constexpr int RED = BK;         // reduction depth of an LDS stage
  template <int DIM, bool TYPE_R, class L, int PF, bool IS_A = false> using Stg = Stager<DIM, TYPE_R, L, PF>;
};
struct PathBF16 {
  typedef bf16_t T;
  template <bool HAS_SUB, class XT> using Act = ActLoaderT<HAS_SUB, XT>;
  template <bool POOLED> using Dy = DyLoaderT<POOLED, bf16_t>;
  typedef WLoaderT<bf16_t> WL;           // `w` points at the bf16 copy of the weights (t3d_cast_bf16)
  typedef WLoaderT<bf16_t, true> WLX;
  static constexpr bool BF16 = true;
  static constexpr bool X3 = false;
  static constexpr bool PC = false;
  static constexpr int RED = BKH;
  // the first operand of every GEMM here is the [M, C] stream from HBM: two register slots (prefetch distance 2); the second
  // (weights from L2, or the fatter dy operand of the weight gradient) one
  // (PF == 0: one slot for the first operand too -- the 128-column data-gradient tilings, whose second slot spilled 10-27 VGPRs to scratch)
  template <int DIM, bool TYPE_R, class L, int PF, bool IS_A = false> using Stg = StagerHSel<DIM, TYPE_R, L, (IS_A && PF != 0) ? 2 : 1>;
};
// The fp32 activation loader for layers whose channel count is a multiple of the k-tile (the launchers take the x3 path only then):
// no clamped addresses, no column masks -- four selects and a min per chunk less in a staging pass that bounds these kernels.
// (a layer input without a batch-norm in front of it: scale = 1, shift = 0 from these tables, so that the k-loop has no branch on
// `scale != nullptr` -- hipcc kept that branch, nine v_mov and two conditional loads, inside every k-tile)
// (NOT const: a const table lives in the constant address space, the select between it and the caller's pointer is then a generic
// pointer, and the load a flat_load -- which returns out of order: s_waitcnt vmcnt(0) in front of every use)
__device__ float t3d_ident_scale[4096] = {
#define T3D_R16_(x) x, x, x, x, x, x, x, x, x, x, x, x, x, x, x, x
#define T3D_R256_(x) T3D_R16_(x), T3D_R16_(x), T3D_R16_(x), T3D_R16_(x), T3D_R16_(x), T3D_R16_(x), T3D_R16_(x), T3D_R16_(x), \
                     T3D_R16_(x), T3D_R16_(x), T3D_R16_(x), T3D_R16_(x), T3D_R16_(x), T3D_R16_(x), T3D_R16_(x), T3D_R16_(x)
    T3D_R256_(1.f), T3D_R256_(1.f), T3D_R256_(1.f), T3D_R256_(1.f), T3D_R256_(1.f), T3D_R256_(1.f), T3D_R256_(1.f), T3D_R256_(1.f),
    T3D_R256_(1.f), T3D_R256_(1.f), T3D_R256_(1.f), T3D_R256_(1.f), T3D_R256_(1.f), T3D_R256_(1.f), T3D_R256_(1.f), T3D_R256_(1.f)};
#undef T3D_R256_
#undef T3D_R16_
__device__ float t3d_ident_shift[4096] = {0.f};
constexpr int T3D_IDENT_MAX = 4096;
struct ActLoaderE {
  static constexpr bool EXACT = true;
  t3d_act_src s;
  int K;
  int rpf;
  struct Raw { float4 x; };
  struct Coef { float4 sc, sh; };
  static constexpr bool HAS_AT = true;      // (see DyLoader::fetch_at)
  __device__ __forceinline__ static void pin(Raw& r) { asm volatile("" : "+v"(r.x.x), "+v"(r.x.y), "+v"(r.x.z), "+v"(r.x.w)); }
  __device__ __forceinline__ int ld_elems() const { return s.ldx; }
  __device__ __forceinline__ const float* base_ptr() const { return s.x + s.coff; }
  __device__ __forceinline__ Raw fetch_at(size_t uoff, unsigned lbytes) const {
    Raw r;
    r.x = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(s.x + s.coff + uoff) + lbytes);
    return r;
  }
  __device__ __forceinline__ Coef fetch_coef_at(int ucol, unsigned lbytes) const {      // K <= T3D_IDENT_MAX when scale == nullptr (launchers)
    const float* scp = s.scale != nullptr ? s.scale : t3d_ident_scale;
    const float* shp = s.scale != nullptr ? s.shift : t3d_ident_shift;
    Coef c;
    c.sc = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(scp + ucol) + lbytes);
    c.sh = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(shp + ucol) + lbytes);
    return c;
  }
  static constexpr bool HAS_CTAB = true;      // (see DyLoader::ctab_fill)
  __device__ __forceinline__ int ctab_floats() const { return 2 * K; }
  __device__ __forceinline__ void ctab_fill(float* tab, int tid, int nthreads) const {
    const float* scp = s.scale != nullptr ? s.scale : t3d_ident_scale;
    const float* shp = s.scale != nullptr ? s.shift : t3d_ident_shift;
    for (int i = tid * 4; i < K; i += nthreads * 4) {
      *reinterpret_cast<float4*>(tab + i) = *reinterpret_cast<const float4*>(scp + i);
      *reinterpret_cast<float4*>(tab + K + i) = *reinterpret_cast<const float4*>(shp + i);
    }
  }
  __device__ __forceinline__ Coef ctab_coef(const float* tab, int col) const {
    Coef c;
    c.sc = *reinterpret_cast<const float4*>(tab + col);
    c.sh = *reinterpret_cast<const float4*>(tab + K + col);
    return c;
  }
  __device__ __forceinline__ Coef fetch_coef(int col) const {
    Coef c;
    c.sc = make_float4(1.f, 1.f, 1.f, 1.f);
    c.sh = f4zero();
    if (s.scale != nullptr) {
      c.sc = *reinterpret_cast<const float4*>(s.scale + col);
      c.sh = *reinterpret_cast<const float4*>(s.shift + col);
    }
    return c;
  }
  __device__ __forceinline__ Raw fetch(int row, int col) const {
    Raw r;
    r.x = *reinterpret_cast<const float4*>(s.x + (size_t)row * s.ldx + s.coff + col);
    return r;
  }
  __device__ __forceinline__ float4 xform(const Raw& r, const Coef& c, int, int) const {
    const float floor_ = s.relu ? 0.f : -INFINITY;
    return make_float4(fmaxf(fmaf(r.x.x, c.sc.x, c.sh.x), floor_), fmaxf(fmaf(r.x.y, c.sc.y, c.sh.y), floor_),
                       fmaxf(fmaf(r.x.z, c.sc.z, c.sh.z), floor_), fmaxf(fmaf(r.x.w, c.sc.w, c.sh.w), floor_));
  }
};

// fp32 storage and loaders, bf16 x 3 arithmetic (see above)
struct PathX3 {
  typedef float T;
  template <bool HAS_SUB, class XT> using Act = std::conditional_t<HAS_SUB, ActLoader<true>, ActLoaderE>;
  template <bool POOLED> using Dy = DyLoader<POOLED>;
  typedef WLoaderT<float> WL;
  typedef WLoaderT<float, true> WLX;
  static constexpr bool BF16 = false;
  static constexpr bool X3 = true;
  static constexpr bool PC = false;
  static constexpr int RED = BKX;
  template <int DIM, bool TYPE_R, class L, int PF, bool IS_A = false> using Stg = StagerX3<DIM, TYPE_R, L>;
};
// the loader of a layer's [K, N] weight matrix for a path (fp32 / bf16 copy: rows x cols with bounds; x3 planes: whole tiles)
template <class WL, bool FWD = true, class Args>      // FWD: the forward arrangement of the fragment planes (lane index = N); else the data gradient's (= K)
__device__ __forceinline__ WL make_wloader(const Args& p) {
  if constexpr (IsFrag<WL>::value) return WL{reinterpret_cast<const bf16_t*>(p.w_x3), (long)p.w_x3_stride, FWD ? p.N / 32 : p.K / 32};
  else return WL{p.w, p.N, p.K, p.N};
}
struct PathX3P : PathX3 {      // ... with the layer's weight matrix split beforehand, in fragment order (t3d_pointmlp_fwd_args.w_x3: WLoaderX3F)
  typedef WLoaderX3F WL;
  typedef WLoaderX3F WLX;
  template <int DIM, bool TYPE_R, class L, int PF, bool IS_A = false>
  using Stg = std::conditional_t<IsFrag<L>::value, StagerX3F<DIM, TYPE_R, L>, StagerX3<DIM, TYPE_R, L>>;
};
// ... with EIGHT waves per workgroup (2 x 4 waves of 64 x 64: a 128 x 256 output tile).  The staged A tile serves twice the columns: a
// quarter fewer staged elements -- loads, batch-norm / ReLU, three-way splits, LDS stores -- per MFMA than two 128 x 128 workgroups, which
// is what the additive model of these kernels (docs/EXPERIMENTS.md, round 5) says their time depends on.  92 KB of LDS: one workgroup per
// CU, the same eight waves per CU as two four-wave workgroups.
struct PathX3W : PathX3 {
  static constexpr int WAVES = 8;
  template <int DIM, bool TYPE_R, class L, int PF, bool IS_A = false> using Stg = StagerX3<DIM, TYPE_R, L, T3D_X3_PF, 2 * NT>;
};
struct PathX3WP : PathX3W {      // ... and the weights in fragment order
  typedef WLoaderX3F WL;
  typedef WLoaderX3F WLX;
  template <int DIM, bool TYPE_R, class L, int PF, bool IS_A = false>
  using Stg = std::conditional_t<IsFrag<L>::value, StagerX3F<DIM, TYPE_R, L>, StagerX3<DIM, TYPE_R, L, T3D_X3_PF, 2 * NT>>;
};
template <class PR, class = void> struct WavesOf { static constexpr int value = 4; };
template <class PR> struct WavesOf<PR, typename std::enable_if<(PR::WAVES > 0)>::type> { static constexpr int value = PR::WAVES; };
// ... in 512-thread workgroups with producer and consumer waves (gemm_mainloop_x3_pc): three register slots per operand
struct PathX3PC : PathX3 {
  static constexpr bool PC = true;
  template <int DIM, bool TYPE_R, class L, int PF, bool IS_A = false> using Stg = StagerX3<DIM, TYPE_R, L, X3_RING>;
};
template <class PR, int TM, int TN, class SA, class SB, class LA, class LB, bool AR, int DIMA, bool BR, int DIMB, bool SYM = false>
__device__ __forceinline__ void run_mainloop(SA& sa, SB& sb, const LA& la, const LB& lb, float* smem, int red_begin, int red_end,
                                             int a0, int b0, f32x16 (&acc)[TM][TN], int tid) {
  if constexpr (PR::PC) gemm_mainloop_x3_pc<SYM, TM, TN, SA, SB, LA, LB, AR, DIMA, BR, DIMB>(sa, sb, la, lb, smem, red_begin, red_end, a0, b0, acc, tid);
  else if constexpr (PR::X3) gemm_mainloop_x3<SYM, TM, TN, SA, SB, LA, LB, AR, DIMA, BR, DIMB>(sa, sb, la, lb, smem, red_begin, red_end, a0, b0, acc, tid);
  else if constexpr (PR::BF16) gemm_mainloop_h<TM, TN, SA, SB, LA, LB, AR, DIMA, BR, DIMB>(sa, sb, la, lb, smem, red_begin, red_end, a0, b0, acc, tid);
  else gemm_mainloop<TM, TN, SA, SB, LA, LB, AR, DIMA, BR, DIMB>(sa, sb, la, lb, smem, red_begin, red_end, a0, b0, acc, tid);
}

// ---------------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------------
// PR: arithmetic + element type of y (PathF32 / PathBF16); XT: element type of the input tensor (fp32 for the raw inputs)
template <int BN, bool HAS_SUB, class PR = PathF32, class XT = float>
__device__ __forceinline__ void fwd_body(const t3d_pointmlp_fwd_args& p, float* smem, const int bid, const int nblk) {
  constexpr int BM = 128, TM = 2;
  constexpr int WN = WavesOf<PR>::value / 2, WCOLS = BN / WN, TN = WCOLS / 32;      // wave grid 2 x WN, a wave owns 64 x WCOLS
  static_assert(TN == 1 || TN == 2, "a wave owns 32 or 64 columns");
  using LA = typename PR::template Act<HAS_SUB, XT>;
  using YT = typename PR::T;
  constexpr int PF = BN == 64 ? T3D_PF_NARROW : T3D_PF_WIDE;
  using WL = std::conditional_t<LA::EXACT, typename PR::WLX, typename PR::WL>;      // K % 64 == 0: every weight tile is whole
  // (bf16 arithmetic on an fp32 source -- a raw input wider than 4 channels, e.g. the Box-PC representation -- at 128 columns: one slot)
  using SA = typename PR::template Stg<BM, true, LA, (PR::BF16 && BN == 128 && !Elem<XT>::BF16) ? 0 : PF, true>;
  using SB = typename PR::template Stg<BN, false, WL, PF>;

  // (producer / consumer workgroups: the second 256 threads are the staging waves of the same tile coordinates; they end in the main loop)
  const int tid = PR::PC ? (int)(threadIdx.x & (NT - 1)) : (int)threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid / WN, wn = wid % WN;
  const int tiles_n = p.N / BN;
  const int lin = xcd_remap(bid, nblk);
  const int tile_m = lin / tiles_n, tile_n = lin % tiles_n;
  const int row0 = tile_m * BM, col0 = tile_n * BN;
  T3D_TRACE_MARK(0);

  LA la{p.a, p.K, p.rows_per_frustum};
  WL lb = make_wloader<WL>(p);
  SA sa; SB sb;
  sa.init(la, row0, tid);
  sb.init(lb, col0, tid);

  // the epilogue's per-column additive terms (bias, conv6's per-frustum row bias), requested BEFORE the main loop: loaded at the
  // head of the epilogue they were a dependent L2 round trip in front of every launch's stores (round 3, tools/trace_fwd_res.py)
  // (the two terms stay apart until the epilogue adds them: summed here, the sum would wait for both loads at the kernel's head)
  float addv[TN], addr[TN];
  {
    const int l31_ = lane & 31, b_ = row0 / p.rows_per_frustum;
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
      const int col = col0 + wn * WCOLS + tn * 32 + l31_;
      addv[tn] = p.bias ? p.bias[col] : 0.f;
      addr[tn] = p.rowbias ? p.rowbias[(size_t)b_ * p.N + col] : 0.f;
    }
  }
  const bool has_rowbias = p.rowbias != nullptr;
  f32x16 acc[TM][TN];
  zero_acc<TM, TN>(acc);
  const int kred = (p.K + PR::RED - 1) / PR::RED * PR::RED;
  run_mainloop<PR, TM, TN, SA, SB, LA, WL, true, BM, false, BN>(sa, sb, la, lb, smem, 0, kred, wm * 64,
                                                                     wn * WCOLS, acc, tid);
  T3D_TRACE_MARK(1);

#ifdef T3D_ABL_NOEPI
  if (p.M > 0) {          // diagnostic build: skip the epilogue but keep the accumulators live
    if (p.K < 0) {
#pragma unroll
      for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
#pragma unroll
          for (int r = 0; r < 16; ++r) p.y[tm * 32 + tn * 16 + r] = acc[tm][tn][r];
    }
    return;
  }
#endif
  // epilogue: + bias (+ per-frustum row bias), store y, column statistics, optional pool partials
  const int l31 = lane & 31, h = lane >> 5;
  const int b = row0 / p.rows_per_frustum;
  const bool pool = p.pmax != nullptr;
  const bool store_y = p.y != nullptr;      // Gram-form backward: statistics and pool partials only
  float* red = smem;   // [2][BN] x 6 quantities
  constexpr int YLD = BN + 8;                                       // bf16 output tile [128][BN + 8] behind `red`
  bf16_t* ytile = reinterpret_cast<bf16_t*>(smem + 12 * BN);
  float csum[TN], csq[TN], cmax[TN], cmin[TN];
  int amax[TN], amin[TN];
  // keep flags of this lane's 32 rows, loaded once (they were re-read per column group inside the compare chain)
  unsigned keepbits = 0xffffffffu;
  if (pool && p.rowmask) {
    static_assert(TM == 2, "keep_bits32 covers two 32-row blocks");
    keepbits = keep_bits32(p.rowmask, row0 + wm * 64 + 4 * h);
  }
  const int rin_base = row0 - b * p.rows_per_frustum + wm * 64 + 4 * h;
  // `store_y` and `pool` are compile-time inside the element loop (round 3): as run-time flags hipcc compiled each into a BRANCH
  // PER ELEMENT -- two s_cbranch per output element in every forward kernel, fp32 and bf16 alike (tools/kernel_branches.py: 155
  // branches in k_pointmlp_fwd<128>), and the fp32 store addresses into a 64-bit multiply-add per element.  The arithmetic and its
  // order are unchanged.
  const unsigned n4 = (unsigned)p.N * 4u;                            // row pitch in bytes (uniform)
  char* const yb = reinterpret_cast<char*>(p.y);
  auto epi = [&](auto sy_tag, auto pl_tag) {
    constexpr bool SY = decltype(sy_tag)::value, PL = decltype(pl_tag)::value;
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
      const int col = col0 + wn * WCOLS + tn * 32 + l31;
      const float add = has_rowbias ? addv[tn] + addr[tn] : addv[tn];
      float s = 0.f, ss = 0.f, mx = -INFINITY, mn = INFINITY;
      int ax = -1, an = -1;
      // 32-bit byte offsets (M*N < 2^30 is checked by the launcher): one VGPR add per store on top of the uniform base; the
      // per-lane part is opaque to the optimiser so that it is not folded back into a 64-bit product per element
      unsigned boff0 = ((unsigned)(row0 + wm * 64 + 4 * h) * (unsigned)p.N + (unsigned)col) * 4u;
      asm volatile("" : "+v"(boff0));
#pragma unroll
      for (int tm = 0; tm < TM; ++tm) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          // bf16 storage: statistics (and the pool's candidates) of the value as it is stored, i.e. as every reader sees it
          const float v = SY ? Elem<YT>::rnd(acc[tm][tn][r] + add) : acc[tm][tn][r] + add;
          if constexpr (SY) {
            if constexpr (Elem<YT>::BF16) {
              // bf16: the tile goes through LDS (the stages are free now) and leaves as 16-byte row-contiguous stores below -- 64
              // two-byte global stores per thread were half of a workgroup's lifetime on the narrow layers (tools/trace_blocks.py)
              ytile[(wm * 64 + 4 * h + tm * 32 + (r & 3) + 8 * (r >> 2)) * YLD + wn * WCOLS + tn * 32 + l31] = (bf16_t)v;
            } else {
              *reinterpret_cast<float*>(yb + (boff0 + (unsigned)(tm * 32 + (r & 3) + 8 * (r >> 2)) * n4)) = v;
            }
          }
          s += v;
          ss = fmaf(v, v, ss);
          if constexpr (PL) {                          // selects: same results as the branches they replace, no exec-mask juggling
            const bool keep = (keepbits >> (tm * 16 + r)) & 1u;
            const int rin = rin_base + tm * 32 + (r & 3) + 8 * (r >> 2);
            const bool up = keep & (v > mx), dn = keep & (v < mn);
            mx = up ? v : mx;
            ax = up ? rin : ax;
            mn = dn ? v : mn;
            an = dn ? rin : an;
          }
        }
      }
      // combine the two lane halves (rows +4): lower row index wins ties
      s += __shfl_xor(s, 32, 64);
      ss += __shfl_xor(ss, 32, 64);
      if constexpr (PL) {
        const float omx = __shfl_xor(mx, 32, 64), omn = __shfl_xor(mn, 32, 64);
        const int oax = __shfl_xor(ax, 32, 64), oan = __shfl_xor(an, 32, 64);
        if (oax >= 0 && (omx > mx || ax < 0 || (omx == mx && oax < ax))) { mx = omx; ax = oax; }
        if (oan >= 0 && (omn < mn || an < 0 || (omn == mn && oan < an))) { mn = omn; an = oan; }
      }
      csum[tn] = s; csq[tn] = ss; cmax[tn] = mx; cmin[tn] = mn; amax[tn] = ax; amin[tn] = an;
    }
  };
  if (store_y) { if (pool) epi(std::true_type{}, std::true_type{}); else epi(std::true_type{}, std::false_type{}); }
  else { if (pool) epi(std::false_type{}, std::true_type{}); else epi(std::false_type{}, std::false_type{}); }
  __syncthreads();
  if (h == 0) {
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
      const int c = wn * WCOLS + tn * 32 + l31;
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
  // bf16: the output tile leaves the LDS image LAST, as 16-byte row-contiguous stores behind the last barrier (they drain while
  // the CU already runs the next workgroup).  The fp32 path keeps its in-loop stores: moving them behind the barriers measured
  // 1-2 % SLOWER at B=32 (1.628 vs 1.60 ms per step) -- with two workgroups per CU the wait at the barrier is covered by the other
  // workgroup's MFMAs, and the second pass over the accumulators costs more than it saves.
  if (store_y) {
    if constexpr (Elem<YT>::BF16) {
      constexpr int CPR = BN / 8;                                   // 16-byte chunks per tile row
      bf16_t* yg = reinterpret_cast<bf16_t*>(p.y);
#pragma unroll
      for (int i = 0; i < 128 * CPR / NT; ++i) {
        const int c = tid + NT * i, row = c / CPR, ch = c % CPR;
        const bf16x8 v = *reinterpret_cast<const bf16x8*>(ytile + row * YLD + ch * 8);
        *reinterpret_cast<bf16x8*>(yg + (size_t)(row0 + row) * p.N + col0 + ch * 8) = v;
      }
    }
  }
  T3D_TRACE_MARK(2);
}

template <int BN, bool HAS_SUB, class PR = PathF32, class XT = float>
__global__ __launch_bounds__(NT, T3D_WAVES) void k_pointmlp_fwd(const t3d_pointmlp_fwd_args p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  fwd_body<BN, HAS_SUB, PR, XT>(p, smem, blockIdx.x, gridDim.x);
}

// The eight-wave kernels hold ONE workgroup per CU, so a launch of more tiles than CUs ran in rounds with the CU idle between a
// workgroup's exit and its successor's first instruction (1.9 us in the timeline of the 256 -> 512 pooled layer, tools/trace_blocks.py)
// on top of the epilogue and prologue on either side.  T3D_W8_PERSIST: the launch is min(tiles, CUs) workgroups and a workgroup walks
// tiles b, b + G, b + 2 G, ... itself (same tile -> XCD assignment as the dispatcher's round robin: xcd_remap sees the same index).
#ifndef T3D_W8_PERSIST
#define T3D_W8_PERSIST 1
#endif
template <int BN, class PR>      // eight waves, 128 x 256 tile (PathX3W)
__global__ __launch_bounds__(2 * NT) void k_pointmlp_fwd_w8(const t3d_pointmlp_fwd_args p, const int n_tiles) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  for (int b = blockIdx.x; b < n_tiles; b += gridDim.x) {
    int bb = b;
    asm volatile("" : "+s"(bb));      // (nothing derived from the tile index is hoisted out of the loop and kept live across a tile)
    fwd_body<BN, false, PR, float>(p, smem, bb, n_tiles);
    __syncthreads();                  // the next tile's staging pass overwrites the epilogue's LDS scratch
  }
}
template <int BN, class PR>      // ... hosting riders: the rider workgroups run on their first four waves (the bodies are 256-thread programs)
__global__ __launch_bounds__(2 * NT) void k_pointmlp_fwd_w8_r(const t3d_pointmlp_fwd_args p, const t3d_rider_set r, const int n_tiles) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  if ((int)blockIdx.x < r.n_wg) {
    if (threadIdx.x >= NT) return;      // (s_barrier counts the surviving waves)
    run_riders(r, smem);
    return;
  }
  for (int b = blockIdx.x - r.n_wg; b < n_tiles; b += gridDim.x - r.n_wg) {
    int bb = b;
    asm volatile("" : "+s"(bb));
    fwd_body<BN, false, PR, float>(p, smem, bb, n_tiles);
    __syncthreads();
  }
}

template <int BN, class PR>      // producer / consumer form (gemm_mainloop_x3_pc): eight waves, one workgroup per CU
__global__ __launch_bounds__(2 * NT) void k_pointmlp_fwd_pc(const t3d_pointmlp_fwd_args p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  fwd_body<BN, false, PR, float>(p, smem, blockIdx.x, gridDim.x);
}

// Rider forms (rider_dev.h): the launch's first r.n_wg workgroups run a set of small ops of an independent chain and leave; the
// GEMM's tiles are the workgroups behind them.  fp32 split-form kernels only; the plain kernels above and below are untouched.
template <int BN, bool HAS_SUB, class PR = PathF32>
__global__ __launch_bounds__(NT, T3D_WAVES) void k_pointmlp_fwd_r(const t3d_pointmlp_fwd_args p, const t3d_rider_set r) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  if ((int)blockIdx.x < r.n_wg) { run_riders(r, smem); return; }
  fwd_body<BN, HAS_SUB, PR, float>(p, smem, blockIdx.x - r.n_wg, gridDim.x - r.n_wg);
}

// ---------------------------------------------------------------------------------------------
// bf16 forward, activation-resident: one workgroup owns a 128-row tile for ALL column tiles
// ---------------------------------------------------------------------------------------------
// The T3D_BF16 kernels are VALU-issue-bound (2 300 VALU instructions per wave against 128 MFMAs in the generic
// 512 -> 256 forward): with the MFMA 16x faster than in fp32, what a SIMD spends its time on is the staging pass -- widen, fma, max,
// round, >= 3.5 operations per element -- and the generic kernel repeats that pass for every column tile of a row panel (8 times
// for the 128 -> 1024 layer).  Here the activated, rounded input panel act(x)[128, K] is staged into LDS ONCE (K <= 256: at most four
// R images), every load of it in flight together; then, per column tile, only bf16 weight tiles stream through a two-stage ring
// (copied untouched, from L2) and the epilogue of the generic kernel runs on the tile.  The epilogue's LDS scratch (partial sums +
// the bf16 output tile) aliases the weight ring: 2 x 64 x (BN + 32) x 2 bytes = 12 x BN x 4 + 128 x (BN + 8) x 2 exactly for BN = 128.
template <int BN, int KT>      // KT = K / 64 k-tiles resident
__global__ __launch_bounds__(NT, T3D_WAVES) void k_pointmlp_fwd_res(const t3d_pointmlp_fwd_args p) {
  constexpr int TM = 2, TN = BN / 64, ST = BKH / 16;
  using LA = ActLoaderT<false, bf16_t>;
  using WL = WLoaderT<bf16_t, true>;
  using SA = StagerHSel<128, true, LA, KT>;
  using SB = StagerHSel<BN, false, WL, 1>;
  constexpr int A_ELEMS = SA::LDS_ELEMS, B_ELEMS = SB::LDS_ELEMS;
  static_assert((size_t)2 * B_ELEMS * 2 >= (size_t)12 * BN * 4 + (size_t)128 * (BN + 8) * 2, "the epilogue scratch aliases the weight ring");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  bf16_t* panel = reinterpret_cast<bf16_t*>(smem);
  bf16_t* bst = panel + KT * A_ELEMS;

  // Everything derived from the thread index is re-derived inside the row loop from a laundered copy: hoisted out of the loop
  // (LICM) the address arithmetic of the epilogue and of the stagers stays live across the whole tile -- 171 -> 256 VGPRs + 48
  // spilled when the loop was first wrapped around the kernel.
  const int tid0 = threadIdx.x;
  int tid = tid0, lane = tid & 63, wid = tid >> 6;
  int wm = wid >> 1, wn = wid & 1, l31 = lane & 31, h = lane >> 5;
  const bool pool = p.pmax != nullptr, store_y = p.y != nullptr;
  // Round 3: the workgroup is PERSISTENT over row tiles slot, slot + G, slot + 2 G, ... (G = gridDim.x, two workgroups per CU) and
  // every global load is requested one phase ahead of its use.  A phase trace of the one-tile-per-workgroup form (tools/trace_fwd_res.py,
  // 128 -> 128 at M = 262144) showed a tile as a chain of exposed latencies: 3.0 us until the panel is staged, 3.1 us until the first
  // weight tile is in LDS, 1.3 us of MFMAs, 3.4 us of epilogue that starts with the bias load -- 12 us per tile, two tiles per CU at a
  // time.  vmcnt retires IN ORDER (loads and stores alike on gfx9), so the order of issue is what makes a prefetch a prefetch:
  //   * the next row tile's panel is requested right behind the LAST weight-tile fetch of the current row tile (nothing younger is
  //     waited for until the next row tile's panel store);
  //   * the first weight tile and the bias / row-bias terms of the next column tile (or of the next row tile's first column tile)
  //     are requested AHEAD of the current tile's y stores, so waiting for them does not wait for the stores.
  const int tiles_m = p.M / 128, G = gridDim.x;
  LA la{p.a, p.K, p.rows_per_frustum};
  SA sa;
  // KT <= 2 (K <= 128): the next panel's raw chunks (16 VGPRs per k-tile) travel in registers across the current tile's column
  // tiles; its per-channel coefficients (another 16 per k-tile) are requested late, with the next weight tile -- both held across the
  // epilogue would spill.  KT = 4 (off by default): no prefetch, the panel is requested at the top of its tile.
  constexpr bool PREF = KT <= 2;
  auto fetch_panel = [&](int t_m) {
    sa.init(la, t_m * 128, tid);
    sa.template fetch_raw<0>(la, 0, tid);
    if constexpr (KT > 1) sa.template fetch_raw<1>(la, BKH, tid);
    if constexpr (KT > 2) { sa.template fetch_raw<2>(la, 2 * BKH, tid); sa.template fetch_raw<3>(la, 3 * BKH, tid); }
  };
  auto fetch_panel_coefs = [&]() {
    sa.template fetch_coefs<0>(la, 0, tid);
    if constexpr (KT > 1) sa.template fetch_coefs<1>(la, BKH, tid);
    if constexpr (KT > 2) { sa.template fetch_coefs<2>(la, 2 * BKH, tid); sa.template fetch_coefs<3>(la, 3 * BKH, tid); }
  };
  WL lb{p.w, p.N, p.K, p.N};
  SB sb;
  const int n_tiles = p.N / BN;
  float addv[TN], addr[TN];                 // bias and conv6's per-frustum row bias of this lane's columns in the current column tile
  const bool has_rowbias = p.rowbias != nullptr;          // (added in the epilogue: summed at the load they would be waited for there)
  auto fetch_w0 = [&](int c0, int bb) {
    sb.init(lb, c0, tid);
    sb.template fetch<0>(lb, 0, tid);
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
      const int col = c0 + wn * (BN / 2) + tn * 32 + l31;
      addv[tn] = p.bias ? p.bias[col] : 0.f;
      addr[tn] = has_rowbias ? p.rowbias[(size_t)bb * p.N + col] : 0.f;
    }
  };
  int tile_m = xcd_remap(blockIdx.x, G);
  T3D_TRACE_MARK8(0);
  if constexpr (PREF) { fetch_panel(tile_m); fetch_panel_coefs(); }
  fetch_w0(0, (tile_m * 128) / p.rows_per_frustum);
#pragma unroll 1
 for (; tile_m < tiles_m; tile_m += G) {
  const bool tr_first = tile_m < G;
  tid = tid0;
  asm volatile("" : "+v"(tid));
  lane = tid & 63; wid = tid >> 6; wm = wid >> 1; wn = wid & 1; l31 = lane & 31; h = lane >> 5;
  const int row0 = tile_m * 128;
  const int b = row0 / p.rows_per_frustum;
  // ---- the activated input panel, once per row tile: transform + round + store (the previous tile's MFMAs have left the panel:
  // its k-loop ends with a barrier) ----
  if constexpr (!PREF) { fetch_panel(tile_m); fetch_panel_coefs(); }
  sa.template store<0>(la, panel, tid);
  if constexpr (KT > 1) sa.template store<1>(la, panel + A_ELEMS, tid);
  if constexpr (KT > 2) { sa.template store<2>(la, panel + 2 * A_ELEMS, tid); sa.template store<3>(la, panel + 3 * A_ELEMS, tid); }
  if (tr_first) T3D_TRACE_MARK8(1);
  unsigned keepbits = 0xffffffffu;          // keep flags of this lane's 32 rows (the same rows for every column tile)
  if (pool && p.rowmask) {
    static_assert(TM == 2, "keep_bits32 covers two 32-row blocks");
    keepbits = keep_bits32(p.rowmask, row0 + wm * 64 + 4 * h);
  }
  const int rin_base = row0 - b * p.rows_per_frustum + wm * 64 + 4 * h;
  float* red = reinterpret_cast<float*>(bst);
  constexpr int YLD = BN + 8, CPR = BN / 8;
  bf16_t* ytile = reinterpret_cast<bf16_t*>(red + 12 * BN);
  bf16_t* yg = reinterpret_cast<bf16_t*>(p.y);

  // Prefetches are UNCONDITIONAL on clamped indices (a conditional one keeps the stale registers alive through the other arm: the
  // raw chunks and coefficients of the panel stayed live across every column tile, 256 VGPRs + 38 spilled): the last column tile
  // is its own copy of the body (LAST), and a workgroup without a next row tile re-requests its own tile (L2 hits).
  const int tile_nx = tile_m + G < tiles_m ? tile_m + G : tile_m;
  auto col_tile = [&](const int nt, auto last_tag) {
    constexpr bool LAST = decltype(last_tag)::value;
    const int col0 = nt * BN;
    constexpr bool pf_panel = PREF && LAST;      // behind this column tile's last weight fetch
    f32x16 acc[TM][TN];
    zero_acc<TM, TN>(acc);
    __syncthreads();                          // the panel is complete / the previous tile's epilogue has left the ring
    sb.template store<0>(lb, bst, tid);       // (requested one column tile ago: fetch_w0)
    if constexpr (KT > 1) sb.template fetch<0>(lb, BKH, tid);
    if constexpr (KT <= 2 && pf_panel) fetch_panel(tile_nx);
    __syncthreads();
    if (tr_first && nt == 0) T3D_TRACE_MARK8(2);
#pragma unroll
    for (int t = 0; t < KT; ++t) {
      if (t + 1 < KT) {                       // the next weight tile goes into the other stage first, then its registers are refilled
        sb.template store<0>(lb, bst + ((t + 1) & 1) * B_ELEMS, tid);
        if (t + 2 < KT) {
          sb.template fetch<0>(lb, (t + 2) * BKH, tid);
          if constexpr (pf_panel) { if (t + 2 == KT - 1) fetch_panel(tile_nx); }
        }
      }
      mma_steps_h<TM, TN, true, 128, false, BN, 0, ST>(panel + t * A_ELEMS, bst + (t & 1) * B_ELEMS, wm * 64, wn * (BN / 2), acc, lane,
                                                       [](int) {});
      __syncthreads();
    }
    if (tr_first && nt == 0) T3D_TRACE_MARK8(3);
    // ---- epilogue of the column tile (the generic kernel's, bf16 form) ----
    float csum[TN], csq[TN], cmax[TN], cmin[TN];
    int amax[TN], amin[TN];
    // `store_y` and `pool` are compile-time inside the element loop: as run-time flags hipcc turned each into a BRANCH PER ELEMENT
    // (s_cbranch around every ds_write_b16 and around every max / min update: the 3.3 us "epilogue math" of the phase trace).
    auto epi = [&](auto sy_tag, auto pl_tag) {
      constexpr bool SY = decltype(sy_tag)::value, PL = decltype(pl_tag)::value;
#pragma unroll
      for (int tn = 0; tn < TN; ++tn) {
        const float add = has_rowbias ? addv[tn] + addr[tn] : addv[tn];
        float s_ = 0.f, ss = 0.f, mx = -INFINITY, mn = INFINITY;
        int ax = -1, an = -1;
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const float v = SY ? Elem<bf16_t>::rnd(acc[tm][tn][r] + add) : acc[tm][tn][r] + add;
            if constexpr (SY) ytile[(wm * 64 + 4 * h + tm * 32 + (r & 3) + 8 * (r >> 2)) * YLD + wn * (BN / 2) + tn * 32 + l31] = (bf16_t)v;
            s_ += v;
            ss = fmaf(v, v, ss);
            if constexpr (PL) {
              const bool keep = (keepbits >> (tm * 16 + r)) & 1u;
              const int rin = rin_base + tm * 32 + (r & 3) + 8 * (r >> 2);
              const bool up = keep & (v > mx), dn = keep & (v < mn);
              mx = up ? v : mx;
              ax = up ? rin : ax;
              mn = dn ? v : mn;
              an = dn ? rin : an;
            }
          }
        }
        s_ += __shfl_xor(s_, 32, 64);
        ss += __shfl_xor(ss, 32, 64);
        if constexpr (PL) {
          const float omx = __shfl_xor(mx, 32, 64), omn = __shfl_xor(mn, 32, 64);
          const int oax = __shfl_xor(ax, 32, 64), oan = __shfl_xor(an, 32, 64);
          if (oax >= 0 && (omx > mx || ax < 0 || (omx == mx && oax < ax))) { mx = omx; ax = oax; }
          if (oan >= 0 && (omn < mn || an < 0 || (omn == mn && oan < an))) { mn = omn; an = oan; }
        }
        csum[tn] = s_; csq[tn] = ss; cmax[tn] = mx; cmin[tn] = mn; amax[tn] = ax; amin[tn] = an;
      }
    };
    if (store_y) { if (pool) epi(std::true_type{}, std::true_type{}); else epi(std::true_type{}, std::false_type{}); }
    else { if (pool) epi(std::false_type{}, std::true_type{}); else epi(std::false_type{}, std::false_type{}); }
    if (h == 0) {                             // (the k-loop's last barrier has freed the ring: `red` may be written)
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
    if (tr_first && nt == 0) T3D_TRACE_MARK8(4);
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
    if (tr_first && nt == 0) T3D_TRACE_MARK8(5);
    // the next column tile's (or the next row tile's first) weight tile and additive terms, AHEAD of the y stores
    if constexpr (LAST) {
      fetch_w0(0, (tile_nx * 128) / p.rows_per_frustum);
      if constexpr (PREF) fetch_panel_coefs();
    } else {
      fetch_w0(col0 + BN, b);
    }
    if (store_y) {
#pragma unroll
      for (int i = 0; i < 128 * CPR / NT; ++i) {
        const int c = tid + NT * i, row = c / CPR, ch = c % CPR;
        *reinterpret_cast<bf16x8*>(yg + (size_t)(row0 + row) * p.N + col0 + ch * 8) = *reinterpret_cast<const bf16x8*>(ytile + row * YLD + ch * 8);
      }
    }
    if (tr_first && nt == 0) T3D_TRACE_MARK8(6);
  };
#pragma unroll 1
  for (int nt = 0; nt < n_tiles - 1; ++nt) col_tile(nt, std::false_type{});
  col_tile(n_tiles - 1, std::true_type{});
 }
  T3D_TRACE_MARK8(7);
}

// ---------------------------------------------------------------------------------------------
// bf16 forward of the first layer of a net: K <= 4 raw input channels (xyz [+ 1]), fp32 source, N = 64 or 128
// ---------------------------------------------------------------------------------------------
// The generic kernel pads K to a 64-deep k-tile and runs it through the staging / MFMA machinery: 41 us per launch at B=128 N=2048
// for 3 x 128 multiply-adds per row, against 11 us of HBM time for the output.  Here a thread owns 8 output columns: the (<= 4) x 8
// weights sit in registers, a row costs one 16-byte load, K x 8 FMAs on operands rounded to bf16 exactly as the MFMA path rounds them
// (products of two bf16 values are exact in fp32; the sum over <= 4 terms runs in k order), the bias, the output rounding, the two
// statistics, one 16-byte store.
// YT = float (round 3): the same kernel for the fp32 path -- no operand / output rounding, the <= 4-term sum as an fma chain in k
// order (the MFMA path sums the same products in a permuted k order: results agree to the last bits, not bit for bit).  At
// B=32 N=1024 the generic kernel took 9.2 us per launch for the two 3 -> 128 layers and 7.6 us for 4 -> 64, against 2.7 / 1.4 us of
// HBM time for the output.
template <int N, class YT = bf16_t>
__global__ __launch_bounds__(NT) void k_pointmlp_fwd_tinyk(const t3d_pointmlp_fwd_args p) {
  constexpr bool H = Elem<YT>::BF16;
  constexpr int CPR = N / 8, RPP = NT / CPR, NP = 128 / RPP;      // column chunks per row, rows per pass, passes
  __shared__ float red[2][RPP][N];
  const int tid = threadIdx.x, ch = tid % CPR, r0 = tid / CPR;
  const int tile = blockIdx.x, row0 = tile * 128, b = row0 / p.rows_per_frustum;
  float w[4][8], add[8];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    if (H) {
      const bf16x8 v = *reinterpret_cast<const bf16x8*>(reinterpret_cast<const bf16_t*>(p.w) + (size_t)min(k, p.K - 1) * p.N + ch * 8);
#pragma unroll
      for (int e = 0; e < 8; ++e) w[k][e] = k < p.K ? (float)v[e] : 0.f;
    } else {
      const float4 v0 = *reinterpret_cast<const float4*>(p.w + (size_t)min(k, p.K - 1) * p.N + ch * 8);
      const float4 v1 = *reinterpret_cast<const float4*>(p.w + (size_t)min(k, p.K - 1) * p.N + ch * 8 + 4);
      const float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
#pragma unroll
      for (int e = 0; e < 8; ++e) w[k][e] = k < p.K ? v[e] : 0.f;
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) add[e] = p.bias ? p.bias[ch * 8 + e] : 0.f;
  float sc[4] = {1.f, 1.f, 1.f, 1.f}, sh[4] = {0.f, 0.f, 0.f, 0.f}, sub[4] = {0.f, 0.f, 0.f, 0.f};
  if (p.a.scale != nullptr) {      // (raw inputs carry none; kept for the interface)
    const float4 a = *reinterpret_cast<const float4*>(p.a.scale), c = *reinterpret_cast<const float4*>(p.a.shift);
    sc[0] = a.x; sc[1] = a.y; sc[2] = a.z; sc[3] = a.w; sh[0] = c.x; sh[1] = c.y; sh[2] = c.z; sh[3] = c.w;
  }
  if (p.a.sub != nullptr) {
#pragma unroll
    for (int k = 0; k < 4; ++k) sub[k] = p.a.sub[(size_t)b * p.a.sub_ld + min(k, p.K - 1)];
  }
  const float floor_ = p.a.relu ? 0.f : -INFINITY;
  float s1[8], s2[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { s1[e] = 0.f; s2[e] = 0.f; }
  float4 xs[NP];
#pragma unroll
  for (int i = 0; i < NP; ++i) xs[i] = *reinterpret_cast<const float4*>(p.a.x + (size_t)(row0 + r0 + RPP * i) * p.a.ldx + p.a.coff);
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    const float xv[4] = {xs[i].x, xs[i].y, xs[i].z, xs[i].w};
    float a[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float t = fmaxf(fmaf(xv[k], sc[k], sh[k]), floor_) - sub[k];
      a[k] = k < p.K ? Elem<YT>::rnd(t) : 0.f;      // operand rounding of the bf16 GEMM (identity in fp32)
    }
    float o[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float acc = a[0] * w[0][e];
      acc = fmaf(a[1], w[1][e], acc);
      acc = fmaf(a[2], w[2][e], acc);
      acc = fmaf(a[3], w[3][e], acc);
      const float v = Elem<YT>::rnd(acc + add[e]);
      o[e] = v;
      s1[e] += v;
      s2[e] = fmaf(v, v, s2[e]);
    }
    const size_t off = (size_t)(row0 + r0 + RPP * i) * p.N + ch * 8;
    if (H) {
      bf16x8 ob;
#pragma unroll
      for (int e = 0; e < 8; ++e) ob[e] = (bf16_t)o[e];
      *reinterpret_cast<bf16x8*>(reinterpret_cast<bf16_t*>(p.y) + off) = ob;
    } else {
      *reinterpret_cast<float4*>(p.y + off) = make_float4(o[0], o[1], o[2], o[3]);
      *reinterpret_cast<float4*>(p.y + off + 4) = make_float4(o[4], o[5], o[6], o[7]);
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) { red[0][r0][ch * 8 + e] = s1[e]; red[1][r0][ch * 8 + e] = s2[e]; }
  __syncthreads();
  if (tid < N) {
    float t1 = 0.f, t2 = 0.f;
#pragma unroll
    for (int r = 0; r < RPP; ++r) { t1 += red[0][r][tid]; t2 += red[1][r][tid]; }
    p.psum[(size_t)tile * p.N + tid] = t1;
    p.psumsq[(size_t)tile * p.N + tid] = t2;
  }
}

// ---------------------------------------------------------------------------------------------
// data gradient
// ---------------------------------------------------------------------------------------------
// Shared epilogue of the two data-gradient kernels: + add_in (+ per-column constant), ReLU mask of the producing
// layer, that layer's batch-norm-backward partial sums, store.
struct DgradEpilogue {
  const float* add_in;
  const int32_t* add_live;     // [M] or NULL: add_in is read only on rows whose flag is non-zero (the sparse arg-max rows)
  const float* colconst;
  const float* prev_y;
  const float* prev_scale;
  const float* prev_shift;
  float* out;
  float* psum_dz;
  float* psum_dzy;
  int K;
};

// Element offsets are 32-bit (the launchers check M*K < 2^30): one VGPR per element instead of a 64-bit address pair,
// and the compiler can use the scalar-base + vector-offset addressing form.  ADD / MASK are compile-time so that the
// loads are unconditional (a uniform `ptr ? load : 0` becomes a branch around every load), and the loads of a batch
// are issued ahead of the batch's stores: `out` may alias the inputs as far as the compiler knows, so a load placed
// after a store is never hoisted above it and every element would pay a full memory round trip.
// ADD: 0 = no add_in, 1 = dense add_in, 2 = add_in gated by the per-row flags (rows without a flag are never read).
// fp32 epilogue: the code of round 1, kept textually apart from the bf16 variant below (sharing one body behind `if constexpr`
// changed the compiler's schedule of this one: k_pointmlp_bwd<64,64,64> 28.4 -> 30.6 us at B=32)
template <int BN, int TM, int TN, int ADD, bool MASK>
__device__ __forceinline__ void dgrad_epilogue_body(const DgradEpilogue& p, f32x16 (&acc)[TM][TN], float* red, int tid,
                                                    int row0, int col0, int tile_m) {
  const int lane = tid & 63, wid = tid >> 6, wm = wid >> 1, wn = wid & 1;
  const int l31 = lane & 31, h = lane >> 5;
  const bool stats = MASK && p.psum_dz != nullptr;
  const unsigned K = (unsigned)p.K;
  float cs1[TN], cs2[TN];
  unsigned live = 0u;                    // bit tm*16 + r: the lane's accumulator row (tm, r) has something to add
  if (ADD == 2) {      // all flag loads requested together, then the compares (cf. keep_bits32)
    int4 fl[TM][4];
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
      for (int j = 0; j < 4; ++j) fl[tm][j] = *reinterpret_cast<const int4*>(p.add_live + row0 + wm * 64 + 4 * h + tm * 32 + 8 * j);
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int4 f = fl[tm][j];
        live |= (unsigned)((f.x != 0) | ((f.y != 0) << 1) | ((f.z != 0) << 2) | ((f.w != 0) << 3)) << (tm * 16 + 4 * j);
      }
  }
#pragma unroll
  for (int tn = 0; tn < TN; ++tn) {
    const int col = col0 + wn * (BN / 2) + tn * 32 + l31;
    const float psc = MASK ? p.prev_scale[col] : 0.f, psh = MASK ? p.prev_shift[col] : 0.f;
    const float cc = p.colconst ? p.colconst[col] : 0.f;
    // prev_y, add_in and out share ONE 32-bit byte offset per element on top of their uniform bases (M*K < 2^30: launcher); as
    // element indices into three pointers hipcc formed three 64-bit addresses per element (v_lshlrev_b64 + v_lshl_add_u64 each)
    unsigned boff0 = ((unsigned)(row0 + wm * 64 + 4 * h) * K + (unsigned)col) * 4u;
    asm volatile("" : "+v"(boff0));
    const unsigned k4 = K * 4u;
    const char* const pyb = reinterpret_cast<const char*>(p.prev_y);
    const char* const adb = reinterpret_cast<const char*>(p.add_in);
    char* const outb = reinterpret_cast<char*>(p.out);
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
      constexpr int EB = 8;
#pragma unroll
      for (int r0 = 0; r0 < 16; r0 += EB) {
        float yp[EB], ad[EB];
#pragma unroll
        for (int e = 0; e < EB; ++e) {
          const int r = r0 + e;
          const unsigned o = boff0 + (unsigned)(tm * 32 + (r & 3) + 8 * (r >> 2)) * k4;
#ifdef T3D_ABL_DG_NOLOAD
          yp[e] = psc + (float)r;
#else
          yp[e] = MASK ? *reinterpret_cast<const float*>(pyb + o) : 0.f;
#endif
          if (ADD == 2) ad[e] = ((live >> (tm * 16 + r)) & 1u) ? *reinterpret_cast<const float*>(adb + o) : 0.f;
          else ad[e] = ADD ? *reinterpret_cast<const float*>(adb + o) : 0.f;
        }
#pragma unroll
        for (int e = 0; e < EB; ++e) {
          const int r = r0 + e;
          const unsigned o = boff0 + (unsigned)(tm * 32 + (r & 3) + 8 * (r >> 2)) * k4;
          float v = acc[tm][tn][r] + cc + ad[e];
          if (MASK) {
            if (!(fmaf(yp[e], psc, psh) > 0.f)) v = 0.f;
            s1 += v;
            s2 = fmaf(v, yp[e], s2);
          }
          *reinterpret_cast<float*>(outb + o) = v;
        }
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

template <int BN, int TM, int TN, int ADD, bool MASK, class T>     // T: element type of prev_y, out and a dense add_in
__device__ __forceinline__ void dgrad_epilogue_body_h(const DgradEpilogue& p, f32x16 (&acc)[TM][TN], float* red, int tid,
                                                    int row0, int col0, int tile_m) {
  const int lane = tid & 63, wid = tid >> 6, wm = wid >> 1, wn = wid & 1;
  const int l31 = lane & 31, h = lane >> 5;
  const bool stats = MASK && p.psum_dz != nullptr;
  const unsigned K = (unsigned)p.K;
  float cs1[TN], cs2[TN];
  unsigned live = 0u;                    // bit tm*16 + r: the lane's accumulator row (tm, r) has something to add
  if (ADD == 2) {      // all flag loads requested together, then the compares (cf. keep_bits32)
    int4 fl[TM][4];
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
      for (int j = 0; j < 4; ++j) fl[tm][j] = *reinterpret_cast<const int4*>(p.add_live + row0 + wm * 64 + 4 * h + tm * 32 + 8 * j);
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int4 f = fl[tm][j];
        live |= (unsigned)((f.x != 0) | ((f.y != 0) << 1) | ((f.z != 0) << 2) | ((f.w != 0) << 3)) << (tm * 16 + 4 * j);
      }
  }
  // bf16: the producer's raw output (and a dense add_in) come in and the gradient goes out through LDS tiles, moved with 16-byte
  // row-contiguous accesses -- 64 two-byte gathers and 64 two-byte stores per thread were most of a workgroup's epilogue.
  // `red` holds 2 x 2 x BN floats; the tiles sit behind the forward epilogue's 12 x BN floats (one allocation rule for both).
  constexpr int YLD = BN + 8, CPR = BN / 8;
  bf16_t* ptile = reinterpret_cast<bf16_t*>(red + 12 * BN);         // prev_y in, out (same element, same thread) out
  bf16_t* atile = ptile + 128 * YLD;                                // dense add_in
  if constexpr (Elem<T>::BF16) {
    const bf16_t* pg = reinterpret_cast<const bf16_t*>(p.prev_y);
    const bf16_t* ag = reinterpret_cast<const bf16_t*>(p.add_in);
#pragma unroll
    for (int i = 0; i < 128 * CPR / NT; ++i) {
      const int c = tid + NT * i, row = c / CPR, ch = c % CPR;
      const size_t go = (size_t)(row0 + row) * K + col0 + ch * 8;
      if (MASK) *reinterpret_cast<bf16x8*>(ptile + row * YLD + ch * 8) = *reinterpret_cast<const bf16x8*>(pg + go);
      if (ADD == 1) *reinterpret_cast<bf16x8*>(atile + row * YLD + ch * 8) = *reinterpret_cast<const bf16x8*>(ag + go);
    }
    __syncthreads();
  }
#pragma unroll
  for (int tn = 0; tn < TN; ++tn) {
    const int col = col0 + wn * (BN / 2) + tn * 32 + l31;
    const float psc = MASK ? p.prev_scale[col] : 0.f, psh = MASK ? p.prev_shift[col] : 0.f;
    const float cc = p.colconst ? p.colconst[col] : 0.f;
    const unsigned off0 = (unsigned)(row0 + wm * 64 + 4 * h) * K + (unsigned)col;
    float s1 = 0.f, s2 = 0.f;
    if constexpr (Elem<T>::BF16) {
#pragma unroll
      for (int tm = 0; tm < TM; ++tm) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int lr = wm * 64 + 4 * h + tm * 32 + (r & 3) + 8 * (r >> 2), li = lr * YLD + wn * (BN / 2) + tn * 32 + l31;
          const float ypv = MASK ? (float)ptile[li] : 0.f;
          float ad = 0.f;
          if (ADD == 1) ad = (float)atile[li];
          if (ADD == 2) ad = ((live >> (tm * 16 + r)) & 1u) ? p.add_in[off0 + (unsigned)(tm * 32 + (r & 3) + 8 * (r >> 2)) * K] : 0.f;
          float v = Elem<T>::rnd(acc[tm][tn][r] + cc + ad);
          if (MASK) {
            if (!(fmaf(ypv, psc, psh) > 0.f)) v = 0.f;
            s1 += v;
            s2 = fmaf(v, ypv, s2);
          }
          ptile[li] = (bf16_t)v;
        }
      }
    } else {
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
      constexpr int EB = 8;
#pragma unroll
      for (int r0 = 0; r0 < 16; r0 += EB) {
        float yp[EB], ad[EB];
#pragma unroll
        for (int e = 0; e < EB; ++e) {
          const int r = r0 + e;
          const unsigned o = off0 + (unsigned)(tm * 32 + (r & 3) + 8 * (r >> 2)) * K;
#ifdef T3D_ABL_DG_NOLOAD
          yp[e] = psc + (float)r;
#else
          yp[e] = MASK ? Elem<T>::ld1(p.prev_y, o) : 0.f;
#endif
          if (ADD == 2) ad[e] = ((live >> (tm * 16 + r)) & 1u) ? p.add_in[o] : 0.f;      // the sparse rows S: fp32
          else ad[e] = ADD ? Elem<T>::ld1(p.add_in, o) : 0.f;
        }
#pragma unroll
        for (int e = 0; e < EB; ++e) {
          const int r = r0 + e;
          const unsigned o = off0 + (unsigned)(tm * 32 + (r & 3) + 8 * (r >> 2)) * K;
          float v = Elem<T>::rnd(acc[tm][tn][r] + cc + ad[e]);
          if (MASK) {
            if (!(fmaf(yp[e], psc, psh) > 0.f)) v = 0.f;
            s1 += v;
            s2 = fmaf(v, yp[e], s2);
          }
          Elem<T>::st1(p.out, o, v);
        }
      }
    }
    }
    s1 += __shfl_xor(s1, 32, 64);
    s2 += __shfl_xor(s2, 32, 64);
    cs1[tn] = s1; cs2[tn] = s2;
  }
  // bf16: the gradient tile leaves its LDS image behind the last barrier (see the forward epilogue); fp32 stored it in the loop
  if (stats || Elem<T>::BF16) __syncthreads();      // bf16: the LDS tile is complete
  if (stats) {
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
  if constexpr (Elem<T>::BF16) {
    bf16_t* og = reinterpret_cast<bf16_t*>(p.out);
#pragma unroll
    for (int i = 0; i < 128 * CPR / NT; ++i) {
      const int c = tid + NT * i, row = c / CPR, ch = c % CPR;
      *reinterpret_cast<bf16x8*>(og + (size_t)(row0 + row) * K + col0 + ch * 8) = *reinterpret_cast<const bf16x8*>(ptile + row * YLD + ch * 8);
    }
  }
}

template <int BN, int TM, int TN, class T = float>
__device__ __forceinline__ void dgrad_epilogue(const DgradEpilogue& p, f32x16 (&acc)[TM][TN], float* red, int tid, int row0,
                                               int col0, int tile_m) {
  const bool add = p.add_in != nullptr, mask = p.prev_y != nullptr;      // workgroup-uniform
  if constexpr (Elem<T>::BF16) {
    if (mask) {
      if (add && p.add_live) dgrad_epilogue_body_h<BN, TM, TN, 2, true, T>(p, acc, red, tid, row0, col0, tile_m);
      else if (add) dgrad_epilogue_body_h<BN, TM, TN, 1, true, T>(p, acc, red, tid, row0, col0, tile_m);
      else dgrad_epilogue_body_h<BN, TM, TN, 0, true, T>(p, acc, red, tid, row0, col0, tile_m);
    } else {
      if (add && p.add_live) dgrad_epilogue_body_h<BN, TM, TN, 2, false, T>(p, acc, red, tid, row0, col0, tile_m);
      else if (add) dgrad_epilogue_body_h<BN, TM, TN, 1, false, T>(p, acc, red, tid, row0, col0, tile_m);
      else dgrad_epilogue_body_h<BN, TM, TN, 0, false, T>(p, acc, red, tid, row0, col0, tile_m);
    }
  } else {
    if (mask) {
      if (add && p.add_live) dgrad_epilogue_body<BN, TM, TN, 2, true>(p, acc, red, tid, row0, col0, tile_m);
      else if (add) dgrad_epilogue_body<BN, TM, TN, 1, true>(p, acc, red, tid, row0, col0, tile_m);
      else dgrad_epilogue_body<BN, TM, TN, 0, true>(p, acc, red, tid, row0, col0, tile_m);
    } else {
      if (add && p.add_live) dgrad_epilogue_body<BN, TM, TN, 2, false>(p, acc, red, tid, row0, col0, tile_m);
      else if (add) dgrad_epilogue_body<BN, TM, TN, 1, false>(p, acc, red, tid, row0, col0, tile_m);
      else dgrad_epilogue_body<BN, TM, TN, 0, false>(p, acc, red, tid, row0, col0, tile_m);
    }
  }
}

template <int BN, bool POOLED, class PR = PathF32>   // BN = tile width over the layer's INPUT channels K
__device__ __forceinline__ void dgrad_body(const t3d_pointmlp_dgrad_args& p, float* smem, int bid, int nblocks) {
  constexpr int BM = 128, TM = 2, TN = BN / 64;
  constexpr int PF = BN == 64 ? T3D_PF_NARROW : T3D_PF_WIDE;
  using LA = typename PR::template Dy<POOLED>;
  using WL = typename PR::WLX;            // bf16: K % 64 == 0 and N % 64 == 0 (launcher-checked), every tile is whole
  using SA = typename PR::template Stg<BM, true, LA, (PR::BF16 && BN == 128) ? 0 : PF, true>;
  using SB = typename PR::template Stg<BN, true, WL, PF>;

  const int tid = threadIdx.x, wid = tid >> 6;
  const int wm = wid >> 1, wn = wid & 1;
  const int tiles_n = p.K / BN;
  const int lin = nblocks > 0 ? xcd_remap(bid, nblocks) : bid;      // nblocks == 0: `bid` already is the logical tile index
  const int tile_m = lin / tiles_n, tile_n = lin % tiles_n;
  const int row0 = tile_m * BM, col0 = tile_n * BN;

  LA la{p.dy, p.N, p.rows_per_frustum};
  WL lb = make_wloader<WL, false>(p);
  SA sa; SB sb;
  sa.init(la, row0, tid);
  sb.init(lb, col0, tid);

  f32x16 acc[TM][TN];
  zero_acc<TM, TN>(acc);
  const int nred = (p.N + PR::RED - 1) / PR::RED * PR::RED;
  run_mainloop<PR, TM, TN, SA, SB, LA, WL, true, BM, true, BN>(sa, sb, la, lb, smem, 0, nred, wm * 64,
                                                                    wn * (BN / 2), acc, tid);

  DgradEpilogue e{p.add_in, nullptr, nullptr, p.prev_y, p.prev_scale, p.prev_shift, p.out, p.psum_dz, p.psum_dzy, p.K};
  dgrad_epilogue<BN, TM, TN, typename PR::T>(e, acc, smem, tid, row0, col0, tile_m);
}

template <int BN, bool POOLED, class PR = PathF32>
__global__ __launch_bounds__(NT, T3D_WAVES) void k_pointmlp_dgrad(const t3d_pointmlp_dgrad_args p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  dgrad_body<BN, POOLED, PR>(p, smem, blockIdx.x, gridDim.x);
}

// Gram-form data gradient of a max-pooled layer: out = act(a) . P + rowconst + S  (see t3d.h K11e); the operand
// side is the forward kernel's (activations type R, the K x K matrix type C), the epilogue is the dgrad one.
template <int BN, class PR = PathF32>
__device__ __forceinline__ void dgrad_gram_body(const t3d_pointmlp_dgrad_gram_args& p, float* smem, int bid, int nblocks) {
  constexpr int BM = 128, TM = 2, TN = BN / 64;
  using LA = typename PR::template Act<false, typename PR::T>;
  constexpr int PF = BN == 64 ? T3D_PF_NARROW : T3D_PF_GRAM128;
  using SA = typename PR::template Stg<BM, true, LA, (PR::BF16 && BN == 128) ? 0 : PF, true>;
  using SB = typename PR::template Stg<BN, false, WLoader, PF>;
  const int tid = threadIdx.x, wid = tid >> 6;
  const int wm = wid >> 1, wn = wid & 1;
  const int tiles_n = p.K / BN;
  const int lin = xcd_remap(bid, nblocks);
  const int tile_m = lin / tiles_n, tile_n = lin % tiles_n;
  const int row0 = tile_m * BM, col0 = tile_n * BN;
  LA la{p.a, p.K, p.rows_per_frustum};
  WLoader lb{p.p, p.K, p.K, p.K};
  SA sa; SB sb;
  sa.init(la, row0, tid);
  sb.init(lb, col0, tid);
  f32x16 acc[TM][TN];
  zero_acc<TM, TN>(acc);
  run_mainloop<PR, TM, TN, SA, SB, LA, WLoader, true, BM, false, BN>(sa, sb, la, lb, smem, 0, p.K, wm * 64, wn * (BN / 2), acc,
                                                                     tid);
  DgradEpilogue e{p.add_in, p.add_live, p.rowconst, p.prev_y, p.prev_scale, p.prev_shift, p.out, p.psum_dz, p.psum_dzy, p.K};
  dgrad_epilogue<BN, TM, TN, typename PR::T>(e, acc, smem, tid, row0, col0, tile_m);
}

template <int BN, class PR = PathF32>
__global__ __launch_bounds__(NT, T3D_WAVES) void k_pointmlp_dgrad_gram(const t3d_pointmlp_dgrad_gram_args p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  dgrad_gram_body<BN, PR>(p, smem, blockIdx.x, gridDim.x);
}

// ---------------------------------------------------------------------------------------------
// weight gradient (split over rows)
// ---------------------------------------------------------------------------------------------
// slab[split][k0.., n0..] = sum over the split's rows of A[m,k] B[m,n]; both operands type C (row index = reduction).
template <int BMK, int BN, class PR = PathF32, bool SYM = false, class LA, class LB>      // SYM: see mma_x3 (Gram matrices on the x3 path)
__device__ __forceinline__ void wgrad_body(const LA& la, const LB& lb, float* slabs, int K, int N, int rows_per_split,
                                           float* smem, int bid, int nblocks) {
  constexpr int TM = BMK / 64, TN = BN / 64;
  constexpr int PF = (BMK == 64 && BN == 64) ? T3D_PF_NARROW : T3D_PF_WIDE;
  using SA = typename PR::template Stg<BMK, false, LA, (PR::BF16 && BMK == 128 && BN == 128) ? 0 : PF, true>;      // (one slot: see PathBF16::Stg)
  using SB = typename PR::template Stg<BN, false, LB, PF>;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid >> 1, wn = wid & 1;
  const int tiles_k = (K + BMK - 1) / BMK, tiles_n = N / BN;
  const int lin = nblocks > 0 ? xcd_remap(bid, nblocks) : bid;      // nblocks == 0: `bid` already is the logical tile index
  const int split = lin / (tiles_k * tiles_n);
  const int t = lin % (tiles_k * tiles_n);
  const int k0 = (t / tiles_n) * BMK, n0 = (t % tiles_n) * BN;
  SA sa; SB sb;
  sa.init(la, k0, tid);
  sb.init(lb, n0, tid);
  f32x16 acc[TM][TN];
  zero_acc<TM, TN>(acc);
  const int m_begin = split * rows_per_split;
  run_mainloop<PR, TM, TN, SA, SB, LA, LB, false, BMK, false, BN, SYM && PR::X3>(sa, sb, la, lb, smem, m_begin, m_begin + rows_per_split,
                                                                                 wm * (BMK / 2), wn * (BN / 2), acc, tid);
  const int l31 = lane & 31, h = lane >> 5;
  float* slab = slabs + (size_t)split * K * N;
  if (k0 + BMK <= K) {
    // the tile lies inside the matrix (every layer but the K <= 4 first ones): no bounds test per element -- as a test it was an
    // exec-mask BRANCH and a 64-bit multiply-add per stored element (tools/kernel_branches.py) -- and one 32-bit offset add per store
    // on top of the uniform slab base (a slab is K x N <= 2^21 floats, t3d_wgrad_plan)
    char* const sbase = reinterpret_cast<char*>(slab);
    const unsigned n4 = (unsigned)N * 4u;
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
      unsigned boff = ((unsigned)(k0 + wm * (BMK / 2) + 4 * h) * (unsigned)N + (unsigned)(n0 + wn * (BN / 2) + tn * 32 + l31)) * 4u;
      asm volatile("" : "+v"(boff));
#pragma unroll
      for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          *reinterpret_cast<float*>(sbase + (boff + (unsigned)(tm * 32 + (r & 3) + 8 * (r >> 2)) * n4)) = acc[tm][tn][r];
    }
    return;
  }
#pragma unroll
  for (int tn = 0; tn < TN; ++tn) {
    const int col = n0 + wn * (BN / 2) + tn * 32 + l31;
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int k = k0 + wm * (BMK / 2) + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (k < K) slab[(size_t)k * N + col] = acc[tm][tn][r];
      }
    }
  }
}

// XT: element type of the layer input `a` (fp32 for the raw inputs: the first layer of each net)
template <int BMK, int BN, bool HAS_SUB, bool POOLED, class PR = PathF32, class XT = float>
__global__ __launch_bounds__(NT, T3D_WAVES) void k_pointmlp_wgrad(const t3d_pointmlp_wgrad_args p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  typename PR::template Act<HAS_SUB, XT> la{p.a, p.K, p.rows_per_frustum};
  typename PR::template Dy<POOLED> lb{p.dy, p.N, p.rows_per_frustum};
  wgrad_body<BMK, BN, PR>(la, lb, p.slabs, p.K, p.N, p.rows_per_split, smem, blockIdx.x, gridDim.x);
}

// Weight gradient of a first layer (K <= 4 raw input channels, fp32, dense dy): the generic kernel pads K to a 64-row tile (10.7 us per
// launch at M = 32768, N = 128 for 2 x 16.8 MB of input: 0.4 of the HBM rate).  Here a thread owns 8 columns x K accumulators in
// registers, walks every RPP-th row of its split -- two 16-byte loads each of dz and y, one of the input row -- and the row groups
// meet through LDS in a fixed order.  One slab [K, N] per split, as the generic kernel writes them.
template <int N, bool HAS_SUB>
__global__ __launch_bounds__(NT) void k_pointmlp_wgrad_tinyk(const t3d_pointmlp_wgrad_args p) {
  constexpr int CPR = N / 8, RPP = NT / CPR;      // column chunks per row, rows per pass
  __shared__ float red[RPP][4][N];
  const int tid = threadIdx.x, ch = tid % CPR, r0 = tid / CPR;
  const int split = blockIdx.x, row_begin = split * p.rows_per_split, row_end = row_begin + p.rows_per_split;
  float c0[8], c1[8], c2[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    c0[e] = p.dy.coef[ch * 8 + e];
    c1[e] = p.dy.coef[p.N + ch * 8 + e];
    c2[e] = p.dy.coef[2 * p.N + ch * 8 + e];
  }
  float sc[4] = {1.f, 1.f, 1.f, 1.f}, sh[4] = {0.f, 0.f, 0.f, 0.f};
  if (p.a.scale != nullptr) {
    const float4 a = *reinterpret_cast<const float4*>(p.a.scale), c = *reinterpret_cast<const float4*>(p.a.shift);
    sc[0] = a.x; sc[1] = a.y; sc[2] = a.z; sc[3] = a.w; sh[0] = c.x; sh[1] = c.y; sh[2] = c.z; sh[3] = c.w;
  }
  const float floor_ = p.a.relu ? 0.f : -INFINITY;
  float acc[4][8];
#pragma unroll
  for (int k = 0; k < 4; ++k)
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[k][e] = 0.f;
  constexpr int U = 4;                              // rows in flight per thread
  for (int r = row_begin + r0; r < row_end; r += RPP * U) {
    float4 dz[U][2], y[U][2], x[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int rr = min(r + u * RPP, row_end - 1);      // clamped: no branch around the loads
      const size_t off = (size_t)rr * p.N + ch * 8;
      dz[u][0] = *reinterpret_cast<const float4*>(p.dy.dz + off);
      dz[u][1] = *reinterpret_cast<const float4*>(p.dy.dz + off + 4);
      y[u][0] = *reinterpret_cast<const float4*>(p.dy.y + off);
      y[u][1] = *reinterpret_cast<const float4*>(p.dy.y + off + 4);
      x[u] = *reinterpret_cast<const float4*>(p.a.x + (size_t)rr * p.a.ldx + p.a.coff);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const bool live = r + u * RPP < row_end;
      const float xv[4] = {x[u].x, x[u].y, x[u].z, x[u].w};
      float a[4];
      float sub[4] = {0.f, 0.f, 0.f, 0.f};
      if (HAS_SUB) {
        const int b = min(r + u * RPP, row_end - 1) / p.rows_per_frustum;
#pragma unroll
        for (int k = 0; k < 4; ++k) sub[k] = p.a.sub[(size_t)b * p.a.sub_ld + min(k, p.K - 1)];
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) a[k] = (live && k < p.K) ? fmaxf(fmaf(xv[k], sc[k], sh[k]), floor_) - sub[k] : 0.f;
      const float dzv[8] = {dz[u][0].x, dz[u][0].y, dz[u][0].z, dz[u][0].w, dz[u][1].x, dz[u][1].y, dz[u][1].z, dz[u][1].w};
      const float yv[8] = {y[u][0].x, y[u][0].y, y[u][0].z, y[u][0].w, y[u][1].x, y[u][1].y, y[u][1].z, y[u][1].w};
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float d = fmaf(c0[e], dzv[e], fmaf(c1[e], yv[e], c2[e]));
#pragma unroll
        for (int k = 0; k < 4; ++k) acc[k][e] = fmaf(a[k], d, acc[k][e]);
      }
    }
  }
#pragma unroll
  for (int k = 0; k < 4; ++k)
#pragma unroll
    for (int e = 0; e < 8; ++e) red[r0][k][ch * 8 + e] = acc[k][e];
  __syncthreads();
  float* slab = p.slabs + (size_t)split * p.K * p.N;
  for (int i = tid; i < p.K * N; i += NT) {
    const int k = i / N, n = i % N;
    float t = 0.f;
#pragma unroll
    for (int g = 0; g < RPP; ++g) t += red[g][k][n];
    slab[(size_t)k * p.N + n] = t;
  }
}

// Gram matrix of a layer input, G = a^T a, as split-row slabs (t3d.h K11e).
template <int BMK, int BN, class PR = PathF32>
__global__ __launch_bounds__(NT, T3D_WAVES) void k_pointmlp_gram(const t3d_pointmlp_gram_args p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  typename PR::template Act<false, typename PR::T> la{p.a, p.K, p.rows_per_frustum};
  wgrad_body<BMK, BN, PR, true>(la, la, p.slabs, p.K, p.K, p.rows_per_split, smem, blockIdx.x, gridDim.x);
}

// One launch for both gradients of a dense layer: the first `n_wgrad` workgroups run weight-gradient tiles, the rest
// data-gradient tiles.  The two are independent (both read dy = c0*dz + c1*y + c2), so sharing a launch removes one
// kernel's fill/drain latency per layer and lets the tiles of one kind fill the holes the other leaves on a CU.
template <int DBN, int WBMK, int WBN, class PR = PathF32>
__global__ __launch_bounds__(NT, T3D_WAVES) void k_pointmlp_bwd(const t3d_pointmlp_dgrad_args d, const t3d_pointmlp_wgrad_args w,
                                                                const int n_wgrad, const int interleave) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  typename PR::template Act<false, typename PR::T> la{w.a, w.K, w.rows_per_frustum};
  typename PR::template Dy<false> lb{w.dy, w.N, w.rows_per_frustum};
#ifdef T3D_TRACE      // (tools/trace_bwd.py: entry, exit and the kind of tile of every workgroup; slot 4 = its main loop's prologue done)
  T3D_TRACE_MARK(0);
  if (threadIdx.x == 0 && t3d_trace_ptr) t3d_trace_ptr[(size_t)blockIdx.x * T3D_TRACE_STRIDE + 1] = (int)blockIdx.x < n_wgrad ? 1 : 2;
  if (!interleave) {
    if ((int)blockIdx.x < n_wgrad) wgrad_body<WBMK, WBN, PR>(la, lb, w.slabs, w.K, w.N, w.rows_per_split, smem, blockIdx.x, n_wgrad);
    else dgrad_body<DBN, false, PR>(d, smem, blockIdx.x - n_wgrad, gridDim.x - n_wgrad);
    T3D_TRACE_MARK(2);
    return;
  }
#endif
  if (interleave) {
    // logical order: per row split, its weight-gradient tiles followed by the data-gradient tiles of the same rows; the
    // XCD remap hands each XCD a contiguous piece of that order, so both readers of a dy row range share one L2
    const int wt = ((w.K + WBMK - 1) / WBMK) * (w.N / WBN), tiles_nd = d.K / DBN;
    const int dt = (w.rows_per_split / 128) * tiles_nd, grp_sz = wt + dt;
    const int l = xcd_remap(blockIdx.x, gridDim.x);
    const int grp = l / grp_sz, r = l % grp_sz;
    if (r < wt) wgrad_body<WBMK, WBN, PR>(la, lb, w.slabs, w.K, w.N, w.rows_per_split, smem, grp * wt + r, 0);
    else dgrad_body<DBN, false, PR>(d, smem, grp * dt + (r - wt), 0);
  } else if ((int)blockIdx.x < n_wgrad) {
    wgrad_body<WBMK, WBN, PR>(la, lb, w.slabs, w.K, w.N, w.rows_per_split, smem, blockIdx.x, n_wgrad);
  } else {
    dgrad_body<DBN, false, PR>(d, smem, blockIdx.x - n_wgrad, gridDim.x - n_wgrad);
  }
}

#ifndef T3D_R64_WAVES
// (experiment of round 6: 3 caps the rider-hosting <64,64,64> backward at 168 VGPRs -- the GEMM bodies need 148 and the plain kernel runs three
// workgroups per CU -- but the rider bodies then spill 500 bytes per lane and the hosted launch goes from 28.5 to 35.8 us, the step from 1.15 to 1.25 ms)
#define T3D_R64_WAVES T3D_WAVES
#endif
template <int DBN, int WBMK, int WBN, class PR = PathF32>
__global__ __launch_bounds__(NT, (DBN == 64 && WBMK == 64 && WBN == 64) ? T3D_R64_WAVES : T3D_WAVES) void k_pointmlp_bwd_r(const t3d_pointmlp_dgrad_args d, const t3d_pointmlp_wgrad_args w,
                                                                  const int n_wgrad, const t3d_rider_set r) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  if ((int)blockIdx.x < r.n_wg) { run_riders(r, smem); return; }
  const int bid = blockIdx.x - r.n_wg, nblk = gridDim.x - r.n_wg;
  typename PR::template Act<false, float> la{w.a, w.K, w.rows_per_frustum};
  typename PR::template Dy<false> lb{w.dy, w.N, w.rows_per_frustum};
  if (bid < n_wgrad) wgrad_body<WBMK, WBN, PR>(la, lb, w.slabs, w.K, w.N, w.rows_per_split, smem, bid, n_wgrad);
  else dgrad_body<DBN, false, PR>(d, smem, bid - n_wgrad, nblk - n_wgrad);
}

// first layer of a net (raw points in, no data gradient): the weight gradient alone
template <int BMK, int BN, bool HAS_SUB, bool POOLED>
__global__ __launch_bounds__(NT, T3D_WAVES) void k_pointmlp_wgrad_r(const t3d_pointmlp_wgrad_args p, const t3d_rider_set r) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  if ((int)blockIdx.x < r.n_wg) { run_riders(r, smem); return; }
  typename PathF32::template Act<HAS_SUB, float> la{p.a, p.K, p.rows_per_frustum};
  typename PathF32::template Dy<POOLED> lb{p.dy, p.N, p.rows_per_frustum};
  wgrad_body<BMK, BN, PathF32>(la, lb, p.slabs, p.K, p.N, p.rows_per_split, smem, blockIdx.x - r.n_wg, gridDim.x - r.n_wg);
}

// ---------------------------------------------------------------------------------------------
// one-pass backward of a dense bf16 layer with K, N in {64, 128}, or 256 x 128 / 128 x 256
// ---------------------------------------------------------------------------------------------
// The split form above reads dz, y and the layer input twice (once per kind of workgroup) and transforms every element twice --
// and the bf16 kernels are bound by exactly that element-wise VALU work and by HBM.  Here ONE workgroup owns
// a run of 128-row tiles (rows_per_split rows): per tile it builds dy = c0*dz + c1*y + c2 and a = relu(x*scale + shift) ONCE as
// LDS images, runs dW += a^T dy (accumulators live in registers across all its tiles) and dX = dy W^T from those images -- the
// weights of the data gradient sit in registers as MFMA fragments for the whole kernel -- and finishes dX (rounding, ReLU mask of
// the producing layer, its batch-norm-backward partial sums) against the raw input tile it already holds in LDS.  Every input
// element is read from HBM once and transformed once.  One fp32 slab per workgroup leaves at the end.  The arithmetic per element
// is that of the split form (same operand rounding, same MFMA step order along n and along the rows of a tile), so dX is
// bit-identical and dW differs only by where the row splits fall.
//
// ONE image serves both GEMMs: [128 rows][DIM] bf16, no padding, the 16-byte slot index of row m XOR-ed with swz(m):
//   * dX needs dy with the reduction index n contiguous per lane (one ds_read_b128, 16 lanes = 16 rows per bank pass): rows are
//     256 B (or 128 B) apart, so without the swizzle all 16 would hit the same slot; swz is a bijection of the row's low four bits;
//   * dW needs both images through the transposing read (ds_read_b64_tr_b16; reduction index = row): a half-wave addresses 4 rows
//     x 64 B, and swz's high bits move the four rows to four different 64-byte bank segments.
template <int DIM>
__device__ __forceinline__ int swz(int m) {
  return DIM >= 128 ? (((m & 3) << 2) | ((m >> 2) & 3)) : ((((m >> 1) & 1) << 2) | ((m >> 2) & 3));
}
template <int DIM>
__device__ __forceinline__ int img_off(int m, int col) {      // bf16 element offset of (row m, column col)
  return m * DIM + ((((col >> 3) ^ swz<DIM>(m))) << 3) + (col & 7);
}
// operand rows r0 + (lane & 31), reduction indices 16 st + 8 h .. + 7 along the image row
template <int DIM>
__device__ __forceinline__ bf16x8 frag_r1(const bf16_t* img, int r0, int st, int lane) {
  return *reinterpret_cast<const bf16x8*>(img + img_off<DIM>(r0 + (lane & 31), 16 * st + 8 * (lane >> 5)));
}
// operand columns c0 + (lane & 31), reduction indices = image rows 16 st + 8 h .. + 7 (transposing read, see frag_h)
template <int DIM>
__device__ __forceinline__ bf16x8 frag_c1(const bf16_t* img, int c0, int st, int lane) {
  const int row = 16 * st + 8 * (lane >> 5) + ((lane & 15) >> 2), col = c0 + 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(img + img_off<DIM>(row, col)));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(img + img_off<DIM>(row + 4, col)));
  return __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
}

// 512 threads, one workgroup per CU, row tiles of BM rows (128; 64 for the 256-wide layers, whose images would not fit).
// Wave roles (wv = 0..7), 32 x 32 MFMA tiles:
//   dW (K/32 x N/32 tiles): 32 tiles -> 2 x 2 per wave; 16 -> two k-tiles x one n-tile; 8 -> one per wave; 4 (64 x 64) -> one per
//      wave PAIR, each wave reducing over half of the tile's rows (the pair's sums meet in LDS once, before the slab is written)
//   dX (BM/32 x K/32 tiles): the K/32 column tiles go to 8 / RG waves each ... RG = 8 / (K/32) row groups of BM / RG rows
// The raw bf16 chunks of tile t+1 (dz, y, x: 32-48 VGPRs) are requested right after tile t's images are complete and land under
// its MFMAs and epilogue: ~96 KB in flight per CU, which is what the HBM share of a CU needs at ~2 us of latency.
// LDS: D [BM][N], A [BM][K] (activated), X [2][BM][K + 8] (the producer's RAW output: mask + statistics of the epilogue; the
// masked gradient overwrites it element by element and leaves as 16-byte rows; two copies so that tile t+1 is staged while slow
// threads still copy tile t out), 2 x 2 x RG x K floats of cross-wave statistics (the batch-norm partials are per 128 rows: two
// 64-row tiles share one).  Two barriers per tile.
constexpr int NT1 = 512;
template <int K, int N, int BM>
__global__ __launch_bounds__(NT1) void k_pointmlp_bwd1(const t3d_pointmlp_dgrad_args d, const t3d_pointmlp_wgrad_args w) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  bf16_t* Dimg = reinterpret_cast<bf16_t*>(smem);
  bf16_t* Aimg = Dimg + BM * N;
  bf16_t* Ximg = Aimg + BM * K;
  constexpr int XLD = K + 8;                           // X is never an MFMA operand: plain rows, 16 bytes of padding
  constexpr int CT = K / 32, RG = 8 / CT, TMD = BM / (32 * RG);      // dX: column tiles, row groups, row tiles per wave
  static_assert(CT == 2 || CT == 4 || CT == 8, "K in {64, 128, 256}");
  static_assert(TMD >= 1 && TMD * 32 * RG == BM, "dX tiles do not cover the row tile");
  float* red = reinterpret_cast<float*>(Ximg + 2 * BM * XLD);      // [BM == 64 ? 2 : 1][2][RG][K]
  constexpr int TNn = N / 32, TILES = CT * TNn;
  static_assert(TILES == 4 || TILES == 8 || TILES == 16 || TILES == 32, "K x N tiles");
  constexpr int TMW = TILES >= 16 ? 2 : 1, TNW = TILES == 32 ? 2 : 1;
  constexpr int RS = TILES >= 8 ? 1 : 8 / TILES;
  static_assert(RS == 1 || BM == 128, "row-split pairs: 128-row tiles only");
  constexpr int STN = N / 16, STW = BM / 16 / RS;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, l31 = lane & 31, h = lane >> 5;
  const int wt = TILES >= 16 ? 0 : wv % TILES;
  // dW tile coordinates (in 32-wide tiles): 32 tiles: wave grid (K/64) x (N/64); 16: (K/64) x (N/32); else one tile
  const int kt0 = TILES == 32 ? (wv / (N / 64)) * 2 : TILES == 16 ? (wv / TNn) * 2 : wt / TNn;
  const int nt0 = TILES == 32 ? (wv % (N / 64)) * 2 : TILES == 16 ? (wv % TNn) : wt % TNn;
  const int rs = TILES >= 8 ? 0 : wv / TILES;
  const int rg = wv / CT, ct = wv % CT;
  const int xr0 = rg * (BM / RG);
  const int split = blockIdx.x;
  const int row_begin = split * w.rows_per_split, n_tiles = w.rows_per_split / BM;

  const bf16_t* dzg = reinterpret_cast<const bf16_t*>(d.dy.dz);
  const bf16_t* yg = reinterpret_cast<const bf16_t*>(d.dy.y);
  const bf16_t* xg = reinterpret_cast<const bf16_t*>(w.a.x);
  const bf16_t* wg = reinterpret_cast<const bf16_t*>(d.w);
  const bf16_t* addg = reinterpret_cast<const bf16_t*>(d.add_in);
  bf16_t* outg = reinterpret_cast<bf16_t*>(d.out);
  const bool mask = d.prev_y != nullptr, stats = mask && d.psum_dz != nullptr, add = d.add_in != nullptr;      // uniform

  // weights of the data gradient as B fragments: B[n][k] = W[k][n], lane (k = l31, h) holds n = 16 st + 8 h .. + 7
  bf16x8 wf[STN];
#pragma unroll
  for (int st = 0; st < STN; ++st)
    wf[st] = *reinterpret_cast<const bf16x8*>(wg + (size_t)(ct * 32 + l31) * N + 16 * st + 8 * h);

  // staging maps: a thread owns one 8-column chunk (the same for every row it touches) of each tensor
  constexpr int CPD = N / 8, RPD = NT1 / CPD, NID = BM / RPD;
  constexpr int CPA = K / 8, RPA = NT1 / CPA, NIA = BM / RPA;
  static_assert(NID >= 1 && NIA >= 1, "row tile shorter than one staging pass");
  const int chd = tid % CPD, rd = tid / CPD, cha = tid % CPA, ra = tid / CPA;
  const float floor_ = w.a.relu ? 0.f : -INFINITY;
  const float psc = mask ? d.prev_scale[ct * 32 + l31] : 0.f, psh = mask ? d.prev_shift[ct * 32 + l31] : 0.f;

  bf16x8 rz[NID], ry[NID], rx[NIA];
  auto load_raw = [&](int row0) {
#pragma unroll
    for (int i = 0; i < NID; ++i) {
      const size_t go = (size_t)(row0 + rd + RPD * i) * N + chd * 8;
      rz[i] = *reinterpret_cast<const bf16x8*>(dzg + go);
      ry[i] = *reinterpret_cast<const bf16x8*>(yg + go);
    }
#pragma unroll
    for (int i = 0; i < NIA; ++i)
      rx[i] = *reinterpret_cast<const bf16x8*>(xg + (size_t)(row0 + ra + RPA * i) * w.a.ldx + w.a.coff + cha * 8);
  };

  f32x16 accw[TMW][TNW];
  zero_acc<TMW, TNW>(accw);

  load_raw(row_begin);
  for (int t = 0; t < n_tiles; ++t) {
    const int row0 = row_begin + t * BM;
    bf16_t* Xc = Ximg + (t & 1) * BM * XLD;
    float* redc = red + (BM == 64 ? (t & 1) : 0) * 2 * RG * K;
    // registers -> images: dy = c0 * dz + c1 * y + c2 and a = relu(x * scale + shift), rounded to bf16 after the fp32 arithmetic
    // (as the split form's loaders do); x itself.  The per-column constants are re-read per tile (L1): 40 VGPRs not held.
    {
      float c0[8], c1[8], c2[8];
#pragma unroll
      for (int e = 0; e < 8; e += 4) {
        const float4 a0 = *reinterpret_cast<const float4*>(d.dy.coef + chd * 8 + e);
        const float4 a1 = *reinterpret_cast<const float4*>(d.dy.coef + N + chd * 8 + e);
        const float4 a2 = *reinterpret_cast<const float4*>(d.dy.coef + 2 * N + chd * 8 + e);
        c0[e] = a0.x; c0[e + 1] = a0.y; c0[e + 2] = a0.z; c0[e + 3] = a0.w;
        c1[e] = a1.x; c1[e + 1] = a1.y; c1[e + 2] = a1.z; c1[e + 3] = a1.w;
        c2[e] = a2.x; c2[e + 1] = a2.y; c2[e + 2] = a2.z; c2[e + 3] = a2.w;
      }
#pragma unroll
      for (int i = 0; i < NID; ++i) {
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (bf16_t)fmaf(c0[e], (float)rz[i][e], fmaf(c1[e], (float)ry[i][e], c2[e]));
        *reinterpret_cast<bf16x8*>(Dimg + img_off<N>(rd + RPD * i, chd * 8)) = o;
      }
    }
    {
      float sc[8], sh[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) { sc[e] = 1.f; sh[e] = 0.f; }
      if (w.a.scale != nullptr) {
#pragma unroll
        for (int e = 0; e < 8; e += 4) {
          const float4 b0 = *reinterpret_cast<const float4*>(w.a.scale + cha * 8 + e);
          const float4 b1 = *reinterpret_cast<const float4*>(w.a.shift + cha * 8 + e);
          sc[e] = b0.x; sc[e + 1] = b0.y; sc[e + 2] = b0.z; sc[e + 3] = b0.w;
          sh[e] = b1.x; sh[e + 1] = b1.y; sh[e + 2] = b1.z; sh[e + 3] = b1.w;
        }
      }
#pragma unroll
      for (int i = 0; i < NIA; ++i) {
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (bf16_t)fmaxf(fmaf((float)rx[i][e], sc[e], sh[e]), floor_);
        *reinterpret_cast<bf16x8*>(Aimg + img_off<K>(ra + RPA * i, cha * 8)) = o;
        *reinterpret_cast<bf16x8*>(Xc + (ra + RPA * i) * XLD + cha * 8) = rx[i];
      }
    }
    __syncthreads();
    if (t + 1 < n_tiles) load_raw(row0 + BM);      // workgroup-uniform

    // dW += a^T dy: reduction over the tile's rows (this wave's share of them)
#pragma unroll
    for (int s_ = 0; s_ < STW; ++s_) {
      const int st = rs * STW + s_;
      bf16x8 fa[TMW], fb[TNW];
#pragma unroll
      for (int tm = 0; tm < TMW; ++tm) fa[tm] = frag_c1<K>(Aimg, (kt0 + tm) * 32, st, lane);
#pragma unroll
      for (int tn = 0; tn < TNW; ++tn) fb[tn] = frag_c1<N>(Dimg, (nt0 + tn) * 32, st, lane);
#pragma unroll
      for (int tm = 0; tm < TMW; ++tm)
#pragma unroll
        for (int tn = 0; tn < TNW; ++tn)
          accw[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[tm], fb[tn], accw[tm][tn], 0, 0, 0);
    }
    // dX = dy W^T
    f32x16 accd[TMD];
#pragma unroll
    for (int tm = 0; tm < TMD; ++tm)
#pragma unroll
      for (int r = 0; r < 16; ++r) accd[tm][r] = 0.f;
#pragma unroll
    for (int st = 0; st < STN; ++st) {
      bf16x8 fa[TMD];
#pragma unroll
      for (int tm = 0; tm < TMD; ++tm) fa[tm] = frag_r1<N>(Dimg, xr0 + tm * 32, st, lane);
#pragma unroll
      for (int tm = 0; tm < TMD; ++tm) accd[tm] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[tm], wf[st], accd[tm], 0, 0, 0);
    }
    constexpr int ALD = K + 8;
    if (add) {      // dense add_in tile through LDS (the D and A images are dead once every wave is past its MFMAs)
      __syncthreads();
#pragma unroll
      for (int i = 0; i < NIA; ++i)
        *reinterpret_cast<bf16x8*>(Dimg + (ra + RPA * i) * ALD + cha * 8) =
            *reinterpret_cast<const bf16x8*>(addg + (size_t)(row0 + ra + RPA * i) * K + cha * 8);
      __syncthreads();
    }
    // epilogue: (+ add_in,) round, ReLU mask of the producing layer, its batch-norm-backward partial sums, gradient -> X in place
    float s1 = 0.f, s2 = 0.f;
    auto epi = [&](auto mask_c, auto add_c) {      // compile-time flags: no branch around the per-element LDS reads
      constexpr bool MASK = decltype(mask_c)::value, ADD = decltype(add_c)::value;
      bf16_t* xb = Xc + (xr0 + 4 * h) * XLD + ct * 32 + l31;
      const bf16_t* ab = Dimg + (xr0 + 4 * h) * ALD + ct * 32 + l31;
#pragma unroll
      for (int tm = 0; tm < TMD; ++tm)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int ro = tm * 32 + (r & 3) + 8 * (r >> 2);
          const float ypv = MASK ? (float)xb[ro * XLD] : 0.f;
          const float ad = ADD ? (float)ab[ro * ALD] : 0.f;
          float v = Elem<bf16_t>::rnd(accd[tm][r] + ad);
          if (MASK) {
            if (!(fmaf(ypv, psc, psh) > 0.f)) v = 0.f;
            s1 += v;
            s2 = fmaf(v, ypv, s2);
          }
          xb[ro * XLD] = (bf16_t)v;
        }
    };
    if (mask) { if (add) epi(std::true_type(), std::true_type()); else epi(std::true_type(), std::false_type()); }
    else { if (add) epi(std::false_type(), std::true_type()); else epi(std::false_type(), std::false_type()); }
    s1 += __shfl_xor(s1, 32, 64);
    s2 += __shfl_xor(s2, 32, 64);
    if (stats && h == 0) {
      redc[(0 * RG + rg) * K + ct * 32 + l31] = s1;
      redc[(1 * RG + rg) * K + ct * 32 + l31] = s2;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NIA; ++i)
      *reinterpret_cast<bf16x8*>(outg + (size_t)(row0 + ra + RPA * i) * K + cha * 8) =
          *reinterpret_cast<const bf16x8*>(Xc + (ra + RPA * i) * XLD + cha * 8);
    if (stats && tid < K && (BM == 128 || (t & 1))) {      // 64-row tiles: the pair (t - 1, t) makes one 128-row partial
      float t1 = 0.f, t2 = 0.f;
#pragma unroll
      for (int hb = 0; hb < (BM == 64 ? 2 : 1); ++hb)
#pragma unroll
        for (int g = 0; g < RG; ++g) {
          t1 += red[(hb * 2 * RG + 0 * RG + g) * K + tid];
          t2 += red[(hb * 2 * RG + 1 * RG + g) * K + tid];
        }
      const size_t o = (size_t)(row0 / 128) * K + tid;
      d.psum_dz[o] = t1;
      d.psum_dzy[o] = t2;
    }
    // no barrier here: the next tile is staged into D, A (every wave is past its reads) and the OTHER X copy; a `red` half is
    // written again only behind a later tile's first barrier, which every reader above reaches first
  }

  if (RS == 2) {      // 64 x 64: the two row halves of a tile meet
    float* cmb = reinterpret_cast<float*>(Dimg);
    __syncthreads();
    if (rs == 1) {
#pragma unroll
      for (int r = 0; r < 16; ++r) cmb[(wt * 16 + r) * 64 + lane] = accw[0][0][r];
    }
    __syncthreads();
    if (rs == 0) {
#pragma unroll
      for (int r = 0; r < 16; ++r) accw[0][0][r] += cmb[(wt * 16 + r) * 64 + lane];
    }
  }
  if (rs == 0) {
    float* slab = w.slabs + (size_t)split * K * N;
#pragma unroll
    for (int tn = 0; tn < TNW; ++tn) {
      const int col = (nt0 + tn) * 32 + l31;
#pragma unroll
      for (int tm = 0; tm < TMW; ++tm)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          slab[(size_t)((kt0 + tm) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h) * N + col] = accw[tm][tn][r];
    }
  }
}

// ---------------------------------------------------------------------------------------------
// one-pass backward of a dense fp32 layer with K, N in {64, 128}
// ---------------------------------------------------------------------------------------------
// The fp32 twin of k_pointmlp_bwd1 on v_mfma_f32_32x32x2_f32: ONE 512-thread workgroup per CU owns rows_per_split rows and walks
// them in 64-row tiles.  Per tile it builds dy = c0*dz + c1*y + c2 and a = relu(x*scale + shift) once as padded row-major LDS
// images ([64][DIM + 4]: a ds_read_b128 along the row for the data gradient's A operand, conflict-free ds_read_b32 down a column
// for both operands of the weight gradient), keeps the raw x tile beside them for the epilogue (ReLU mask and batch-norm-backward
// partial sums of the producing layer), and holds W as register fragments for the whole kernel.  dz, y and x are read from HBM
// once; the raw float4 chunks of tile t+1 are requested right behind the barrier that completes tile t's images and land under
// its MFMAs.  dW accumulates in registers across the tiles; one K x N slab per workgroup leaves at the end.
// MFMA steps run in the split form's order (k-groups of 8 ascending, lane half h takes indices 8g + 4h + i), so dX is
// bit-identical to k_pointmlp_dgrad and dW differs only by where the row splits fall.
// Wave roles (wv = 0..7): every 64-row tile has (64/32) x (K/32) data-gradient tiles of N/2 MFMA steps and (K/32) x (N/32)
// weight-gradient tiles of 32 steps -- the same MFMA count on both sides for all four shapes:
//   K = 128: all eight waves own one dX tile (row group wv / 4, column tile wv % 4) AND 16 / 8 (N = 128) or 8 / 8 dW tiles;
//   K = 64 : waves 0-3 own the four dX tiles, waves 4-7 the dW tiles (two k-tiles of one n-tile each for N = 128, one for N = 64).
template <int K, int N, bool ADD>
__global__ __launch_bounds__(NT1) void k_pointmlp_bwd1f(const t3d_pointmlp_dgrad_args d, const t3d_pointmlp_wgrad_args w) {
  static_assert((K == 64 || K == 128) && (N == 64 || N == 128), "K, N in {64, 128}");
  constexpr int BM = 64, DLD = N + 4, ALD = K + 4;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Dimg = smem;                      // [BM][DLD]  dy
  float* Aimg = Dimg + BM * DLD;           // [BM][ALD]  activated input
  float* Ximg = Aimg + BM * ALD;           // [2][BM][ALD]  raw input; the epilogue overwrites it with the finished gradient, which
                                           // leaves as whole rows one tile later (see the loop)
  constexpr int CT = K / 32, RG = 2;       // dX: column tiles, row groups of 32 rows
  float* red = Ximg + 2 * BM * ALD;        // [2 tiles of a 128-row pair][2 sums][RG][K]
  float* cof = red + 2 * 2 * RG * K;       // [3][N] dy coefficients, [2][K] scale / shift of the input: read from HBM/L2 ONCE -- a
                                           // global load at the top of a tile would be waited for with vmcnt(0), i.e. behind the
                                           // previous tile's stores and a full L2 round trip per tile
  constexpr bool ROLES = K == 64;          // K = 64: 4 dX tiles -> waves 0-3, dW -> waves 4-7
  constexpr int TNn = N / 32;
  constexpr int TMW = (K == 128 && N == 64) || (K == 64 && N == 64) ? 1 : 2;      // k-tiles of one n-tile per dW wave
  constexpr int NG = N / 8;                // k-groups of the data gradient
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, l31 = lane & 31, h = lane >> 5;
  const bool do_dx = !ROLES || wv < 4, do_dw = !ROLES || wv >= 4;      // wave-uniform
  const int xw = ROLES ? (wv & 3) : wv;
  const int rg = xw / CT, ct = xw % CT, xr0 = rg * 32;
  const int ww = ROLES ? (wv & 3) : wv;
  // dW tiles of this wave: k-tiles kt0 .. kt0 + TMW - 1 of n-tile nt0
  const int kt0 = K == 128 ? (N == 128 ? (ww / TNn) * 2 : ww / TNn) : (N == 128 ? 0 : ww / TNn);
  const int nt0 = ww % TNn;
  const int split = blockIdx.x;
  const int row_begin = split * w.rows_per_split, n_tiles = w.rows_per_split / BM;

  const bool mask = d.prev_y != nullptr, stats = mask && d.psum_dz != nullptr;      // uniform

  // weights of the data gradient as B fragments: lane (k = ct*32 + l31, h) holds W[k][8g + 4h .. + 3]
  float4 wf[NG];
  if (do_dx) {
#pragma unroll
    for (int g = 0; g < NG; ++g) wf[g] = *reinterpret_cast<const float4*>(d.w + (size_t)(ct * 32 + l31) * N + 8 * g + 4 * h);
  } else {
#pragma unroll
    for (int g = 0; g < NG; ++g) wf[g] = f4zero();
  }

  // staging maps: a thread owns one float4 column chunk (the same for every row it touches) of each tensor
  constexpr int CPD = N / 4, RPD = NT1 / CPD, NID = BM / RPD;
  constexpr int CPA = K / 4, RPA = NT1 / CPA, NIA = BM / RPA;
  static_assert(NID >= 1 && NIA >= 1, "row tile shorter than one staging pass");
  const int chd = tid % CPD, rd = tid / CPD, cha = tid % CPA, ra = tid / CPA;
  const float floor_ = w.a.relu ? 0.f : -INFINITY;
  const float psc = mask ? d.prev_scale[ct * 32 + l31] : 0.f, psh = mask ? d.prev_shift[ct * 32 + l31] : 0.f;

  float4 rz[NID], ry[NID], rx[NIA];
  auto load_raw = [&](int row0) {
#pragma unroll
    for (int i = 0; i < NID; ++i) {
      const size_t go = (size_t)(row0 + rd + RPD * i) * N + chd * 4;
      rz[i] = *reinterpret_cast<const float4*>(d.dy.dz + go);
      ry[i] = *reinterpret_cast<const float4*>(d.dy.y + go);
    }
#pragma unroll
    for (int i = 0; i < NIA; ++i)
      rx[i] = *reinterpret_cast<const float4*>(w.a.x + (size_t)(row0 + ra + RPA * i) * w.a.ldx + w.a.coff + cha * 4);
  };

  f32x16 accw[TMW];
#pragma unroll
  for (int tm = 0; tm < TMW; ++tm)
#pragma unroll
    for (int r = 0; r < 16; ++r) accw[tm][r] = 0.f;

  load_raw(row_begin);
  for (int i = tid; i < 3 * N + 2 * K; i += NT1) {
    float v;
    if (i < 3 * N) v = d.dy.coef[i];
    else if (w.a.scale == nullptr) v = i < 3 * N + K ? 1.f : 0.f;
    else v = i < 3 * N + K ? w.a.scale[i - 3 * N] : w.a.shift[i - 3 * N - K];
    cof[i] = v;
  }
  __syncthreads();
  // Order of the memory operations of one tile: [rows of tile t-1 out] [add_in of tile t] [raw chunks of tile t+1] -- all behind
  // the barrier that completes tile t's images, i.e. under its MFMAs.  vmcnt is one in-order counter for loads AND stores, and
  // the waits at the top of the loop are vmcnt(0) whatever was issued (the compiler merges the loop's entry states): stores placed
  // behind the prefetch, e.g. in the epilogue, would be waited for -- a full write round trip per tile -- before the next tile
  // could even be staged.
  auto rows_out = [&](int row0, const float* Xp) {
#pragma unroll
    for (int i = 0; i < NIA; ++i)
      *reinterpret_cast<float4*>(d.out + (size_t)(row0 + ra + RPA * i) * K + cha * 4) =
          *reinterpret_cast<const float4*>(Xp + (ra + RPA * i) * ALD + cha * 4);
  };
#ifdef T3D_TRACE
  unsigned long long tr_a = 0, tr_b = 0, tr_c = 0, tr_0 = wall_clock64(), tr_t = tr_0;      // staging / MFMA / epilogue, 10 ns ticks
#define T3D_PHASE(acc) do { const unsigned long long n_ = wall_clock64(); acc += n_ - tr_t; tr_t = n_; } while (0)
#else
#define T3D_PHASE(acc) do {} while (0)
#endif
  for (int t = 0; t < n_tiles; ++t) {
    const int row0 = row_begin + t * BM;
    float* Xc = Ximg + (t & 1) * BM * ALD;
    float* redc = red + (t & 1) * 2 * RG * K;
    {      // registers -> images (the per-column constants come from LDS per tile: 20 VGPRs not held)
      const float4 c0 = *reinterpret_cast<const float4*>(cof + chd * 4);
      const float4 c1 = *reinterpret_cast<const float4*>(cof + N + chd * 4);
      const float4 c2 = *reinterpret_cast<const float4*>(cof + 2 * N + chd * 4);
#pragma unroll
      for (int i = 0; i < NID; ++i) {
        const float4 v = make_float4(fmaf(c0.x, rz[i].x, fmaf(c1.x, ry[i].x, c2.x)), fmaf(c0.y, rz[i].y, fmaf(c1.y, ry[i].y, c2.y)),
                                     fmaf(c0.z, rz[i].z, fmaf(c1.z, ry[i].z, c2.z)), fmaf(c0.w, rz[i].w, fmaf(c1.w, ry[i].w, c2.w)));
        *reinterpret_cast<float4*>(Dimg + (rd + RPD * i) * DLD + chd * 4) = v;
      }
      const float4 sc = *reinterpret_cast<const float4*>(cof + 3 * N + cha * 4);
      const float4 sh = *reinterpret_cast<const float4*>(cof + 3 * N + K + cha * 4);
#pragma unroll
      for (int i = 0; i < NIA; ++i) {
        const float4 v = make_float4(fmaxf(fmaf(rx[i].x, sc.x, sh.x), floor_), fmaxf(fmaf(rx[i].y, sc.y, sh.y), floor_),
                                     fmaxf(fmaf(rx[i].z, sc.z, sh.z), floor_), fmaxf(fmaf(rx[i].w, sc.w, sh.w), floor_));
        *reinterpret_cast<float4*>(Aimg + (ra + RPA * i) * ALD + cha * 4) = v;
        *reinterpret_cast<float4*>(Xc + (ra + RPA * i) * ALD + cha * 4) = rx[i];
      }
    }
    __syncthreads();
    T3D_PHASE(tr_a);
    if (t > 0) rows_out(row0 - BM, Ximg + ((t - 1) & 1) * BM * ALD);
    f32x16 accd;
#pragma unroll
    for (int r = 0; r < 16; ++r) accd[r] = 0.f;
    float ad[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) ad[r] = 0.f;
    if (ADD && do_dx) {
      const unsigned ob = (unsigned)(row0 + xr0 + 4 * h) * (unsigned)K + (unsigned)(ct * 32 + l31);
#pragma unroll
      for (int r = 0; r < 16; ++r) ad[r] = d.add_in[ob + (unsigned)((r & 3) + 8 * (r >> 2)) * K];
    }
    if (t + 1 < n_tiles) load_raw(row0 + BM);      // workgroup-uniform
    if (do_dx) {
      // dX = dy W^T: reduction over n
      const float* ap = Dimg + (xr0 + l31) * DLD + 4 * h;
      float4 fa[2];
      fa[0] = *reinterpret_cast<const float4*>(ap);
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        if (g + 1 < NG) fa[(g + 1) & 1] = *reinterpret_cast<const float4*>(ap + 8 * (g + 1));
#ifndef T3D_BWD1F_NOPIN
        __builtin_amdgcn_sched_barrier(0);      // keep the next group's read AHEAD of this group's four dependent MFMAs
#endif
        const float4 a = fa[g & 1], b = wf[g];
        accd = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, accd, 0, 0, 0);
        accd = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, accd, 0, 0, 0);
        accd = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, accd, 0, 0, 0);
        accd = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, accd, 0, 0, 0);
      }
    }
    if (do_dw) {
      // dW += a^T dy: reduction over the tile's rows, k-groups of 8 rows (lane half h: rows 8g + 4h + i)
      const float* ap = Aimg + 4 * h * ALD + kt0 * 32 + l31;
      const float* bp = Dimg + 4 * h * DLD + nt0 * 32 + l31;
      float fa[2][TMW][4], fb[2][4];
      auto ldf = [&](int s_, int g) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          fb[s_][i] = bp[(8 * g + i) * DLD];
#pragma unroll
          for (int tm = 0; tm < TMW; ++tm) fa[s_][tm][i] = ap[(8 * g + i) * ALD + tm * 32];
        }
      };
      ldf(0, 0);
#pragma unroll
      for (int g = 0; g < BM / 8; ++g) {
        if (g + 1 < BM / 8) ldf((g + 1) & 1, g + 1);
#ifndef T3D_BWD1F_NOPIN
        __builtin_amdgcn_sched_barrier(0);
#endif
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int tm = 0; tm < TMW; ++tm)
            accw[tm] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[g & 1][tm][i], fb[g & 1][i], accw[tm], 0, 0, 0);
      }
    }
    // epilogue of the data gradient: (+ add_in,) ReLU mask of the producing layer, its batch-norm-backward partial sums; the
    // finished gradient replaces the raw input element it was masked with
    T3D_PHASE(tr_b);
    if (do_dx) {
      float s1 = 0.f, s2 = 0.f;
      float* xb = Xc + (xr0 + 4 * h) * ALD + ct * 32 + l31;
      auto epi = [&](auto mask_c) {
        constexpr bool MASK = decltype(mask_c)::value;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int ro = (r & 3) + 8 * (r >> 2);
          const float ypv = MASK ? xb[ro * ALD] : 0.f;
          float v = accd[r] + ad[r];
          if (MASK) {
            if (!(fmaf(ypv, psc, psh) > 0.f)) v = 0.f;
            s1 += v;
            s2 = fmaf(v, ypv, s2);
          }
          xb[ro * ALD] = v;
        }
      };
      if (mask) epi(std::true_type()); else epi(std::false_type());
      s1 += __shfl_xor(s1, 32, 64);
      s2 += __shfl_xor(s2, 32, 64);
      if (stats && h == 0) {
        redc[(0 * RG + rg) * K + ct * 32 + l31] = s1;
        redc[(1 * RG + rg) * K + ct * 32 + l31] = s2;
      }
    }
    __syncthreads();      // D and A free for the next tile; gradient rows and statistics of this tile visible
    T3D_PHASE(tr_c);
    if (stats && tid < K && (t & 1)) {      // the pair (t - 1, t) makes one 128-row partial
      float t1 = 0.f, t2 = 0.f;
#pragma unroll
      for (int hb = 0; hb < 2; ++hb)
#pragma unroll
        for (int g = 0; g < RG; ++g) {
          t1 += red[(hb * 2 * RG + 0 * RG + g) * K + tid];
          t2 += red[(hb * 2 * RG + 1 * RG + g) * K + tid];
        }
      const size_t o = (size_t)(row0 / 128) * K + tid;
      d.psum_dz[o] = t1;
      d.psum_dzy[o] = t2;
    }
    // a `red` half is written again only behind a later tile's first barrier, which every reader above reaches first; the X copy
    // of this tile is staged again two tiles on, behind the next tile's second barrier -- its rows leave before that
  }
  rows_out(row_begin + (n_tiles - 1) * BM, Ximg + ((n_tiles - 1) & 1) * BM * ALD);
#ifdef T3D_TRACE
  if (tid == 0 && t3d_trace_ptr) {
    t3d_trace_ptr[(size_t)blockIdx.x * 4 + 0] = tr_a;
    t3d_trace_ptr[(size_t)blockIdx.x * 4 + 1] = tr_b;
    t3d_trace_ptr[(size_t)blockIdx.x * 4 + 2] = tr_c;
    t3d_trace_ptr[(size_t)blockIdx.x * 4 + 3] = wall_clock64() - tr_0;
  }
#endif
#undef T3D_PHASE

  if (do_dw) {
    float* slab = w.slabs + (size_t)split * K * N;
    const int col = nt0 * 32 + l31;
#pragma unroll
    for (int tm = 0; tm < TMW; ++tm)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        slab[(size_t)((kt0 + tm) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h) * N + col] = accw[tm][r];
  }
}

// Gram-form backward of a pooled layer, stage 1: the three jobs that need nothing but the layer input and the
// batch-norm-backward coefficients -- Gram slabs a^T a, column sums of a, and the P / rowconst slabs (+ wc) -- in one launch.
template <int GT, class PR = PathF32>
__global__ __launch_bounds__(NT, T3D_WAVES) void k_pool_bwd_stage1(const t3d_pointmlp_gram_args g, const t3d_act_colsum_args c,
                                                                   const t3d_pool_bwd_prep_args q, const int n_gram,
                                                                   const int n_colsum) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int b = blockIdx.x;
  if (b < n_gram) {
    typename PR::template Act<false, typename PR::T> la{g.a, g.K, g.rows_per_frustum};
    wgrad_body<GT, GT, PR, true>(la, la, g.slabs, g.K, g.K, g.rows_per_split, smem, b, n_gram);
  } else if (b < n_gram + n_colsum) {
    act_colsum_body<typename PR::T>(c, smem, b - n_gram);
  } else {
    const int r = b - n_gram - n_colsum, kb = q.K / 32;
    pool_bwd_prep_body(q, smem, r % kb, (r / kb) % kb, r / (kb * kb));
  }
}

// Stage 2: the weight-gradient assembly and the input-gradient GEMM (both after the slab reduction, independent of each
// other) in one launch.
template <int BN, class PR = PathF32>
__global__ __launch_bounds__(NT, T3D_WAVES) void k_pool_bwd_stage2(const t3d_pool_wgrad_finish_args f,
                                                                   const t3d_pointmlp_dgrad_gram_args d, const int n_finish) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int b = blockIdx.x;
  if (b < n_finish) {
    const int kb = f.K / FK;
    pool_wgrad_finish_body<typename PR::T>(f, smem, b % kb, b / kb);
  } else {
    dgrad_gram_body<BN, PR>(d, smem, b - n_finish, gridDim.x - n_finish);
  }
}

template <int GT, class PR = PathF32>
__global__ __launch_bounds__(NT, T3D_WAVES) void k_pool_bwd_stage1_r(const t3d_pointmlp_gram_args g, const t3d_act_colsum_args c,
                                                                     const t3d_pool_bwd_prep_args q, const int n_gram,
                                                                     const int n_colsum, const t3d_rider_set r) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  if ((int)blockIdx.x < r.n_wg) { run_riders(r, smem); return; }
  const int b = blockIdx.x - r.n_wg;
  if (b < n_gram) {
    typename PR::template Act<false, float> la{g.a, g.K, g.rows_per_frustum};
    wgrad_body<GT, GT, PR, true>(la, la, g.slabs, g.K, g.K, g.rows_per_split, smem, b, n_gram);
  } else if (b < n_gram + n_colsum) {
    act_colsum_body<float>(c, smem, b - n_gram);
  } else {
    const int rr = b - n_gram - n_colsum, kb = q.K / 32;
    pool_bwd_prep_body(q, smem, rr % kb, (rr / kb) % kb, rr / (kb * kb));
  }
}

template <int BN, class PR = PathF32>
__global__ __launch_bounds__(NT, T3D_WAVES) void k_pool_bwd_stage2_r(const t3d_pool_wgrad_finish_args f,
                                                                     const t3d_pointmlp_dgrad_gram_args d, const int n_finish,
                                                                     const t3d_rider_set r) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  if ((int)blockIdx.x < r.n_wg) { run_riders(r, smem); return; }
  const int b = blockIdx.x - r.n_wg;
  if (b < n_finish) {
    const int kb = f.K / FK;
    pool_wgrad_finish_body<float>(f, smem, b % kb, b / kb);
  } else {
    dgrad_gram_body<BN, PR>(d, smem, b - n_finish, (int)gridDim.x - r.n_wg - n_finish);
  }
}

// ---------------------------------------------------------------------------------------------
// one-pass forms of the two GEMMs of the Gram-form backward (bf16, K = 128 or 256 input channels)
// ---------------------------------------------------------------------------------------------
// Same idea as k_pointmlp_bwd1: 512 threads, one workgroup per CU walking row tiles, the layer input read from HBM once per kernel
// and activated once, the raw chunks of the next tile requested while the current one is in the MFMAs.
//   gram1:  G = a^T a slabs (both operands through the transposing read from ONE image) + the per-128-row column sums of a
//           (t3d_act_colsum's output) from the fp32 values the staging pass holds anyway.  Two images: one barrier per tile.
//   dgram1: out = (act(a) . P + rowconst + S) masked, with the producer's batch-norm-backward partials; P sits in registers as
//           B fragments, the raw input tile (= the producer's raw output the mask and the partials need) stays in LDS.
template <int K, int BM>
__device__ __forceinline__ void gram1_body(const t3d_pointmlp_gram_args& g, float* part, float* smem, int split) {
  constexpr int CPA = K / 8, RPA = NT1 / CPA, NIA = BM / RPA;
  constexpr int KT = K / 32, TMW = KT / 2, TNW = KT / 4;      // wave grid 2 x 4 over the KT x KT tiles
  bf16_t* Aimg = reinterpret_cast<bf16_t*>(smem);             // [2][BM][K]
  float* csum = reinterpret_cast<float*>(Aimg + 2 * BM * K);  // [2][RPA][K]
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, l31 = lane & 31, h = lane >> 5;
  const int kt0 = (wv >> 2) * TMW, nt0 = (wv & 3) * TNW;
  const int cha = tid % CPA, ra = tid / CPA;
  const int row_begin = split * g.rows_per_split, n_tiles = g.rows_per_split / BM;
  const bf16_t* xg = reinterpret_cast<const bf16_t*>(g.a.x);
  const float floor_ = g.a.relu ? 0.f : -INFINITY;

  bf16x8 rx[NIA];
  auto load_raw = [&](int row0) {
#pragma unroll
    for (int i = 0; i < NIA; ++i)
      rx[i] = *reinterpret_cast<const bf16x8*>(xg + (size_t)(row0 + ra + RPA * i) * g.a.ldx + g.a.coff + cha * 8);
  };
  f32x16 acc[TMW][TNW];
  zero_acc<TMW, TNW>(acc);
  float pend = 0.f;      // 64-row tiles: the first half of a 128-row column sum
  load_raw(row_begin);
  for (int t = 0; t < n_tiles; ++t) {
    const int row0 = row_begin + t * BM;
    bf16_t* Ac = Aimg + (t & 1) * BM * K;
    float* cs = csum + (t & 1) * RPA * K;
    {
      float sc[8], sh[8], colp[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) { sc[e] = 1.f; sh[e] = 0.f; colp[e] = 0.f; }
      if (g.a.scale != nullptr) {
#pragma unroll
        for (int e = 0; e < 8; e += 4) {
          const float4 b0 = *reinterpret_cast<const float4*>(g.a.scale + cha * 8 + e);
          const float4 b1 = *reinterpret_cast<const float4*>(g.a.shift + cha * 8 + e);
          sc[e] = b0.x; sc[e + 1] = b0.y; sc[e + 2] = b0.z; sc[e + 3] = b0.w;
          sh[e] = b1.x; sh[e + 1] = b1.y; sh[e + 2] = b1.z; sh[e + 3] = b1.w;
        }
      }
#pragma unroll
      for (int i = 0; i < NIA; ++i) {
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float v = fmaxf(fmaf((float)rx[i][e], sc[e], sh[e]), floor_);
          colp[e] += v;
          o[e] = (bf16_t)v;
        }
        *reinterpret_cast<bf16x8*>(Ac + img_off<K>(ra + RPA * i, cha * 8)) = o;
      }
      if (part != nullptr) {
        *reinterpret_cast<float4*>(cs + ra * K + cha * 8) = make_float4(colp[0], colp[1], colp[2], colp[3]);
        *reinterpret_cast<float4*>(cs + ra * K + cha * 8 + 4) = make_float4(colp[4], colp[5], colp[6], colp[7]);
      }
    }
    __syncthreads();
    if (t + 1 < n_tiles) load_raw(row0 + BM);
#pragma unroll
    for (int st = 0; st < BM / 16; ++st) {
      bf16x8 fa[TMW], fb[TNW];
#pragma unroll
      for (int tm = 0; tm < TMW; ++tm) fa[tm] = frag_c1<K>(Ac, (kt0 + tm) * 32, st, lane);
#pragma unroll
      for (int tn = 0; tn < TNW; ++tn) fb[tn] = frag_c1<K>(Ac, (nt0 + tn) * 32, st, lane);
#pragma unroll
      for (int tm = 0; tm < TMW; ++tm)
#pragma unroll
        for (int tn = 0; tn < TNW; ++tn)
          acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[tm], fb[tn], acc[tm][tn], 0, 0, 0);
    }
    if (part != nullptr && tid < K) {
      float s_ = 0.f;
#pragma unroll
      for (int r = 0; r < RPA; ++r) s_ += cs[r * K + tid];
      if (BM == 128) part[(size_t)(row0 / 128) * K + tid] = s_;
      else if (t & 1) part[(size_t)(row0 / 128) * K + tid] = pend + s_;
      else pend = s_;
    }
    // one barrier per tile: tile t+1 goes into the other image / the other partial buffer, whose readers (tile t-1) every wave left
    // before it passed this tile's barrier
  }
  float* slab = g.slabs + (size_t)split * K * K;
#pragma unroll
  for (int tn = 0; tn < TNW; ++tn) {
    const int col = (nt0 + tn) * 32 + l31;
#pragma unroll
    for (int tm = 0; tm < TMW; ++tm)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        slab[(size_t)((kt0 + tm) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h) * K + col] = acc[tm][tn][r];
  }
}
constexpr size_t lds_gram1(int k, int bm) { return (size_t)2 * bm * k * 2 + (size_t)2 * (NT1 / (k / 8)) * k * 4; }

template <int K, int BM>
__device__ __forceinline__ void dgram1_body(const t3d_pointmlp_dgrad_gram_args& p, float* smem, int wg, int nwg) {
  constexpr int XLD = K + 8, SUB = 128 / BM;
  constexpr int CT = K / 32, RG = 8 / CT, TMD = BM / (32 * RG), STK = K / 16;
  constexpr int CPA = K / 8, RPA = NT1 / CPA, NIA = BM / RPA;
  bf16_t* Aimg = reinterpret_cast<bf16_t*>(smem);                       // [BM][K] activated
  bf16_t* Ximg = Aimg + BM * K;                                         // [2][BM][K + 8] raw / gradient out
  float* red = reinterpret_cast<float*>(Ximg + 2 * BM * XLD);           // [SUB][2][RG][K]
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, l31 = lane & 31, h = lane >> 5;
  const int rg = wv / CT, ct = wv % CT, xr0 = rg * (BM / RG);
  const int cha = tid % CPA, ra = tid / CPA;
  const bf16_t* xg = reinterpret_cast<const bf16_t*>(p.a.x);
  bf16_t* outg = reinterpret_cast<bf16_t*>(p.out);
  const bool mask = p.prev_y != nullptr, stats = mask && p.psum_dz != nullptr, add = p.add_in != nullptr;      // uniform
  const float floor_ = p.a.relu ? 0.f : -INFINITY;
  const int col = ct * 32 + l31;
  const float psc = mask ? p.prev_scale[col] : 0.f, psh = mask ? p.prev_shift[col] : 0.f;
  const float cc = p.rowconst ? p.rowconst[col] : 0.f;

  // P as B fragments: B[k'][k] = P[k'][k], lane (k = col, h) holds k' = 16 st + 8 h .. + 7 (rounded to bf16 like the split form's loader)
  bf16x8 pf[STK];
#pragma unroll
  for (int st = 0; st < STK; ++st) {
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = p.p[(size_t)(16 * st + 8 * h + j) * K + col];
#pragma unroll
    for (int j = 0; j < 8; ++j) pf[st][j] = (bf16_t)v[j];
  }

  bf16x8 rx[NIA];
  auto load_raw = [&](int row0) {
#pragma unroll
    for (int i = 0; i < NIA; ++i)
      rx[i] = *reinterpret_cast<const bf16x8*>(xg + (size_t)(row0 + ra + RPA * i) * p.a.ldx + p.a.coff + cha * 8);
  };
  const int n_super = p.M / 128;
  if (wg < n_super) load_raw(wg * 128);
  int it = 0;
  for (int su = wg; su < n_super; su += nwg) {
#pragma unroll 1
    for (int sub = 0; sub < SUB; ++sub, ++it) {
      const int row0 = su * 128 + sub * BM;
      bf16_t* Xc = Ximg + (it & 1) * BM * XLD;
      float* redc = red + sub * 2 * RG * K;
      {
        float sc[8], sh[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { sc[e] = 1.f; sh[e] = 0.f; }
        if (p.a.scale != nullptr) {
#pragma unroll
          for (int e = 0; e < 8; e += 4) {
            const float4 b0 = *reinterpret_cast<const float4*>(p.a.scale + cha * 8 + e);
            const float4 b1 = *reinterpret_cast<const float4*>(p.a.shift + cha * 8 + e);
            sc[e] = b0.x; sc[e + 1] = b0.y; sc[e + 2] = b0.z; sc[e + 3] = b0.w;
            sh[e] = b1.x; sh[e + 1] = b1.y; sh[e + 2] = b1.z; sh[e + 3] = b1.w;
          }
        }
#pragma unroll
        for (int i = 0; i < NIA; ++i) {
          bf16x8 o;
#pragma unroll
          for (int e = 0; e < 8; ++e) o[e] = (bf16_t)fmaxf(fmaf((float)rx[i][e], sc[e], sh[e]), floor_);
          *reinterpret_cast<bf16x8*>(Aimg + img_off<K>(ra + RPA * i, cha * 8)) = o;
          *reinterpret_cast<bf16x8*>(Xc + (ra + RPA * i) * XLD + cha * 8) = rx[i];
        }
      }
      __syncthreads();
      {      // raw chunks of the next tile of this workgroup (workgroup-uniform)
        const int nrow = sub + 1 < SUB ? row0 + BM : (su + nwg) * 128;
        if (sub + 1 < SUB || su + nwg < n_super) load_raw(nrow);
      }
      f32x16 accd[TMD];
#pragma unroll
      for (int tm = 0; tm < TMD; ++tm)
#pragma unroll
        for (int r = 0; r < 16; ++r) accd[tm][r] = 0.f;
#pragma unroll
      for (int st = 0; st < STK; ++st) {
        bf16x8 fa[TMD];
#pragma unroll
        for (int tm = 0; tm < TMD; ++tm) fa[tm] = frag_r1<K>(Aimg, xr0 + tm * 32, st, lane);
#pragma unroll
        for (int tm = 0; tm < TMD; ++tm) accd[tm] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[tm], pf[st], accd[tm], 0, 0, 0);
      }
      unsigned live = 0u;      // bit tm*16 + r: the lane's accumulator row (tm, r) has a sparse arg-max row to add
      if (add) {
#pragma unroll
        for (int tm = 0; tm < TMD; ++tm)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int4 f = *reinterpret_cast<const int4*>(p.add_live + row0 + xr0 + 4 * h + tm * 32 + 8 * j);
            live |= (unsigned)((f.x != 0) | ((f.y != 0) << 1) | ((f.z != 0) << 2) | ((f.w != 0) << 3)) << (tm * 16 + 4 * j);
          }
      }
      float s1 = 0.f, s2 = 0.f;
      auto epi = [&](auto mask_c) {
        constexpr bool MASK = decltype(mask_c)::value;
        bf16_t* xb = Xc + (xr0 + 4 * h) * XLD + col;
        const float* sb = p.add_in + (size_t)(row0 + xr0 + 4 * h) * K + col;
#pragma unroll
        for (int tm = 0; tm < TMD; ++tm)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int ro = tm * 32 + (r & 3) + 8 * (r >> 2);
            const float ypv = MASK ? (float)xb[ro * XLD] : 0.f;
            const float ad = ((live >> (tm * 16 + r)) & 1u) ? sb[(size_t)ro * K] : 0.f;
            float v = Elem<bf16_t>::rnd(accd[tm][r] + cc + ad);
            if (MASK) {
              if (!(fmaf(ypv, psc, psh) > 0.f)) v = 0.f;
              s1 += v;
              s2 = fmaf(v, ypv, s2);
            }
            xb[ro * XLD] = (bf16_t)v;
          }
      };
      if (mask) epi(std::true_type()); else epi(std::false_type());
      s1 += __shfl_xor(s1, 32, 64);
      s2 += __shfl_xor(s2, 32, 64);
      if (stats && h == 0) {
        redc[(0 * RG + rg) * K + col] = s1;
        redc[(1 * RG + rg) * K + col] = s2;
      }
      __syncthreads();
#pragma unroll
      for (int i = 0; i < NIA; ++i)
        *reinterpret_cast<bf16x8*>(outg + (size_t)(row0 + ra + RPA * i) * K + cha * 8) =
            *reinterpret_cast<const bf16x8*>(Xc + (ra + RPA * i) * XLD + cha * 8);
      if (stats && tid < K && sub == SUB - 1) {
        float t1 = 0.f, t2 = 0.f;
#pragma unroll
        for (int hb = 0; hb < SUB; ++hb)
#pragma unroll
          for (int gg = 0; gg < RG; ++gg) {
            t1 += red[(hb * 2 * RG + 0 * RG + gg) * K + tid];
            t2 += red[(hb * 2 * RG + 1 * RG + gg) * K + tid];
          }
        p.psum_dz[(size_t)su * K + tid] = t1;
        p.psum_dzy[(size_t)su * K + tid] = t2;
      }
      // (next staging: the A image is rewritten behind this barrier, X alternates, a `red` half is rewritten only behind a later
      // tile's first barrier)
    }
  }
}
constexpr size_t lds_dgram1(int k, int bm) { return (size_t)bm * k * 2 + (size_t)2 * bm * (k + 8) * 2 + (size_t)(128 / bm) * 2 * (8 / (k / 32)) * k * 4; }

template <int K, int BM>
__global__ __launch_bounds__(NT1) void k_gram1(const t3d_pointmlp_gram_args g) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  gram1_body<K, BM>(g, nullptr, smem, blockIdx.x);
}
template <int K, int BM>
__global__ __launch_bounds__(NT1) void k_dgram1(const t3d_pointmlp_dgrad_gram_args p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  dgram1_body<K, BM>(p, smem, blockIdx.x, gridDim.x);
}
// The stage launches of the Gram-form backward with the one-pass GEMMs.  The small jobs keep their 256-thread bodies: a 512-thread
// block runs TWO logical blocks side by side (each half its own LDS region; their barriers span both halves, which execute the same
// barrier sequence; an odd last block repeats its neighbour's work -- same values to the same addresses).
template <int K, int BM>
__global__ __launch_bounds__(NT1) void k_pool_bwd_stage1_h(const t3d_pointmlp_gram_args g, const t3d_act_colsum_args c,
                                                           const t3d_pool_bwd_prep_args q, const int n_gram) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  if ((int)blockIdx.x < n_gram) {
    gram1_body<K, BM>(g, c.part, smem, blockIdx.x);
  } else {
    const int half = threadIdx.x >> 8, kb = q.K / 32, nlog = kb * kb * ((q.N + PCH - 1) / PCH);
    const int r = min(2 * ((int)blockIdx.x - n_gram) + half, nlog - 1);
    pool_bwd_prep_body(q, smem + half * (PREP_LDS / sizeof(float)), r % kb, (r / kb) % kb, r / (kb * kb), threadIdx.x & 255);
  }
}
template <int K, int BM>
__global__ __launch_bounds__(NT1) void k_pool_bwd_stage2_h(const t3d_pool_wgrad_finish_args f, const t3d_pointmlp_dgrad_gram_args d,
                                                           const int n_fin2, const int fin_lds_floats) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  if ((int)blockIdx.x < n_fin2) {
    const int half = threadIdx.x >> 8, kb = f.K / FK, nlog = kb * (f.N / FN);
    const int b = min(2 * (int)blockIdx.x + half, nlog - 1);
    pool_wgrad_finish_body<bf16_t>(f, smem + half * fin_lds_floats, b % kb, b / kb, threadIdx.x & 255);
  } else {
    dgram1_body<K, BM>(d, smem, blockIdx.x - n_fin2, gridDim.x - n_fin2);
  }
}

// ---------------------------------------------------------------------------------------------
// forward of a max-pooled layer (K = 128 or 256 input channels, no [M,N] store): A-resident, persistent over n
// ---------------------------------------------------------------------------------------------
// The pooled layers are the widest (N = 1024 / 512 / 256 columns from K = 128 / 256 inputs) and keep nothing but
// column statistics and pool partials.  One workgroup owns a 128-row tile for ALL column tiles: the activated input
// panel act(a)[128,K] is staged into LDS once (instead of once per column tile), only the weight tiles stream through
// the two-stage pipeline, the MFMA stream never drains between column tiles, and the epilogue of column tile j
// (statistics, masked max/min + arg-max over the tile's rows) runs in the MFMA shadows of column tile j+1 on a copy
// of the accumulators.  One barrier per k-tile; the cross-wave combination of the epilogue rides on those barriers.
template <int K, int BKB, int NW>
struct FwdPool {
  // NW waves per workgroup: 4 = 2x2 waves of 64x64, one wave per SIMD; 8 = 2x4 waves of 64x32, two waves per SIMD (the
  // side work of one -- weight staging, the running epilogue, barrier waits -- hides under the MFMAs of the other)
  static constexpr int NTP = 64 * NW, WN = NW / 2, TN = 64 / (32 * (WN / 2)), WCOLS = 128 / WN;
  static constexpr int LDA = K + 4, KT = K / BKB, G = BKB / 8, SLOTS = G * 4, NVB = BKB * 32 / NTP;
  static constexpr int NQUADS = 8 * TN, QSTRIDE = SLOTS >= NQUADS ? SLOTS / NQUADS : 1;
  static constexpr int EPI_TILES = SLOTS >= NQUADS ? 1 : NQUADS / SLOTS;   // k-tiles over which the epilogue quads are spread
  static constexpr int A_FLOATS = 128 * LDA, B_FLOATS = BKB * 128, RED_FLOATS = 6 * 2 * 128;
  static constexpr size_t LDS_BYTES = (size_t)(A_FLOATS + 2 * B_FLOATS + RED_FLOATS) * sizeof(float);
  static_assert(KT % 2 == 0, "the LDS stage of a k-tile is a compile-time constant");
  static_assert(NVB >= 1 && EPI_TILES < KT, "tile shape");

  struct Epi {                       // per-lane column accumulators of the tile being finished
    float s[TN], ss[TN], mx[TN], mn[TN];
    int ax[TN], an[TN];
  };

  const t3d_pointmlp_fwd_args& p;
  float* Ap; float* Bst; float* red;
  int tid, lane, l31, h, wm, wn, row0, tile_m, rin_base;
  unsigned keepbits;
  float4 braw[NVB];

  __device__ __forceinline__ void fetch_b(int nt, int kt) {
#pragma unroll
    for (int q = 0; q < NVB; ++q) {
      const int f = tid + NTP * q, kr = f >> 5, c4 = (f & 31) * 4;
      braw[q] = *reinterpret_cast<const float4*>(p.w + (size_t)(kt * BKB + kr) * p.N + nt * 128 + c4);
    }
  }
  __device__ __forceinline__ void store_b_piece(int stage, int q) {
    const int f = tid + NTP * q, kr = f >> 5, c4 = (f & 31) * 4;
    *reinterpret_cast<float4*>(Bst + stage * B_FLOATS + kr * 128 + c4) = braw[q];
  }

  // epilogue phase A, one quad = 4 accumulator elements of one (tn, tm): same visiting order as k_pointmlp_fwd
  __device__ __forceinline__ void epi_quad(const f32x16 (&accp)[2][TN], Epi& e, const float (&add)[TN], int quad) {
    const int tn = quad >> 3, tm = (quad >> 2) & 1, rq = quad & 3;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int r = rq * 4 + j;
      const float v = accp[tm][tn][r] + add[tn];
      e.s[tn] += v;
      e.ss[tn] = fmaf(v, v, e.ss[tn]);
      const bool keep = (keepbits >> (tm * 16 + r)) & 1u;
      const int rin = rin_base + tm * 32 + (r & 3) + 8 * (r >> 2);
      // selects, not branches: this code sits between MFMAs and must stay one basic block
      const bool up = keep & (v > e.mx[tn]), dn = keep & (v < e.mn[tn]);
      e.mx[tn] = up ? v : e.mx[tn];
      e.ax[tn] = up ? rin : e.ax[tn];
      e.mn[tn] = dn ? v : e.mn[tn];
      e.an[tn] = dn ? rin : e.an[tn];
    }
  }
  __device__ __forceinline__ void epi_reset(Epi& e) {
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) { e.s[tn] = 0.f; e.ss[tn] = 0.f; e.mx[tn] = -INFINITY; e.mn[tn] = INFINITY; e.ax[tn] = -1; e.an[tn] = -1; }
  }
  // end of phase A: combine the two lane halves (rows +4), lower row index wins ties; lanes h == 0 publish to LDS
  __device__ __forceinline__ void epi_publish(Epi& e) {
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
      e.s[tn] += __shfl_xor(e.s[tn], 32, 64);
      e.ss[tn] += __shfl_xor(e.ss[tn], 32, 64);
      const float omx = __shfl_xor(e.mx[tn], 32, 64), omn = __shfl_xor(e.mn[tn], 32, 64);
      const int oax = __shfl_xor(e.ax[tn], 32, 64), oan = __shfl_xor(e.an[tn], 32, 64);
      const bool tx = (oax >= 0) & ((omx > e.mx[tn]) | (e.ax[tn] < 0) | ((omx == e.mx[tn]) & (oax < e.ax[tn])));
      const bool tn_ = (oan >= 0) & ((omn < e.mn[tn]) | (e.an[tn] < 0) | ((omn == e.mn[tn]) & (oan < e.an[tn])));
      e.mx[tn] = tx ? omx : e.mx[tn];
      e.ax[tn] = tx ? oax : e.ax[tn];
      e.mn[tn] = tn_ ? omn : e.mn[tn];
      e.an[tn] = tn_ ? oan : e.an[tn];
      if (h == 0) {
        const int c = wn * WCOLS + tn * 32 + l31;
        red[(0 * 2 + wm) * 128 + c] = e.s[tn];
        red[(1 * 2 + wm) * 128 + c] = e.ss[tn];
        red[(2 * 2 + wm) * 128 + c] = e.mx[tn];
        red[(3 * 2 + wm) * 128 + c] = e.mn[tn];
        reinterpret_cast<int*>(red)[(4 * 2 + wm) * 128 + c] = e.ax[tn];
        reinterpret_cast<int*>(red)[(5 * 2 + wm) * 128 + c] = e.an[tn];
      }
    }
  }
  // phase B (after a barrier): the two row halves of the tile -> global partials of column tile `nt`
  __device__ __forceinline__ void epi_write(int nt) {
    if (tid < 128) {
      const int c = tid;
      const size_t o = (size_t)tile_m * p.N + nt * 128 + c;
      p.psum[o] = red[(0 * 2 + 0) * 128 + c] + red[(0 * 2 + 1) * 128 + c];
      p.psumsq[o] = red[(1 * 2 + 0) * 128 + c] + red[(1 * 2 + 1) * 128 + c];
      float mx = red[(2 * 2 + 0) * 128 + c], mn = red[(3 * 2 + 0) * 128 + c];
      int ax = reinterpret_cast<int*>(red)[(4 * 2 + 0) * 128 + c], an = reinterpret_cast<int*>(red)[(5 * 2 + 0) * 128 + c];
      const float mx1 = red[(2 * 2 + 1) * 128 + c], mn1 = red[(3 * 2 + 1) * 128 + c];
      const int ax1 = reinterpret_cast<int*>(red)[(4 * 2 + 1) * 128 + c], an1 = reinterpret_cast<int*>(red)[(5 * 2 + 1) * 128 + c];
      if (ax1 >= 0 && (ax < 0 || mx1 > mx)) { mx = mx1; ax = ax1; }
      if (an1 >= 0 && (an < 0 || mn1 < mn)) { mn = mn1; an = an1; }
      p.pmax[o] = mx; p.pmin[o] = mn; p.pamax[o] = ax; p.pamin[o] = an;
    }
  }
  __device__ __forceinline__ void load_add(int nt, float (&add)[TN]) {
    const int b = row0 / p.rows_per_frustum;
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
      const int col = nt * 128 + wn * WCOLS + tn * 32 + l31;
      float a = p.bias ? p.bias[col] : 0.f;
      if (p.rowbias) a += p.rowbias[(size_t)b * p.N + col];
      add[tn] = a;
    }
  }

  // one k-tile: MFMAs of (nt, KT_IDX) from stage KT_IDX & 1; fillers: the B tile of the next k-tile into the other
  // stage (second half of the slots) and, when HAS_PREV, the epilogue quads of the previous column tile.
  template <int KT_IDX, bool HAS_PREV>
  __device__ __forceinline__ void ktile(f32x16 (&acc)[2][TN], const f32x16 (&accp)[2][TN], Epi& e, const float (&addp)[TN], int nt,
                                        int n_tiles) {
    constexpr int stage = KT_IDX & 1;
    const float* Bs = Bst + stage * B_FLOATS;
    float fa[2][2][4], fb[2][TN][4];
    auto load_frags = [&](int g, int buf) {
#pragma unroll
      for (int tm = 0; tm < 2; ++tm) {
        const float4 v = *reinterpret_cast<const float4*>(Ap + (wm * 64 + tm * 32 + l31) * LDA + KT_IDX * BKB + 8 * g + 4 * h);
        fa[buf][tm][0] = v.x; fa[buf][tm][1] = v.y; fa[buf][tm][2] = v.z; fa[buf][tm][3] = v.w;
      }
#pragma unroll
      for (int tn = 0; tn < TN; ++tn)
#pragma unroll
        for (int i = 0; i < 4; ++i) fb[buf][tn][i] = Bs[(8 * g + 4 * h + i) * 128 + wn * WCOLS + tn * 32 + l31];
    };
    if (HAS_PREV && KT_IDX == EPI_TILES) epi_write(nt - 1);          // phase B: partials published one barrier ago
    load_frags(0, 0);
#pragma unroll
    for (int g = 0; g < G; ++g) {
      if (g + 1 < G) load_frags(g + 1, (g + 1) & 1);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int tm = 0; tm < 2; ++tm)
#pragma unroll
          for (int tn = 0; tn < TN; ++tn)
            acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[g & 1][tm][i], fb[g & 1][tn][i], acc[tm][tn], 0, 0, 0);
        const int slot = g * 4 + i;
#ifndef T3D_ABL_FP_NOSTAGE
        if (slot >= SLOTS / 2 && slot - SLOTS / 2 < NVB) store_b_piece(stage ^ 1, slot - SLOTS / 2);
#endif
        if (HAS_PREV && KT_IDX < EPI_TILES) {
          const int sidx = KT_IDX * SLOTS + slot;
          if (sidx % QSTRIDE == 0 && sidx / QSTRIDE < NQUADS) epi_quad(accp, e, addp, sidx / QSTRIDE);
        }
        // one MFMA cluster + its share of the side work per scheduling region: left alone, the scheduler gathers the
        // epilogue's VALU into one long run that starves the matrix pipe
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (HAS_PREV && KT_IDX == EPI_TILES - 1) epi_publish(e);
    __builtin_amdgcn_sched_barrier(0);
    // loads of the tile after next (clamped past the end: re-reads the last tile)
    {
      int nt2 = nt, kt2 = KT_IDX + 2;
      if (kt2 >= KT) { kt2 -= KT; nt2 = min(nt + 1, n_tiles - 1); }
#ifndef T3D_ABL_FP_NOSTAGE
      fetch_b(nt2, kt2);
#endif
    }
    __syncthreads();
  }

  template <int KT_IDX, bool HAS_PREV>
  __device__ __forceinline__ void ktiles(f32x16 (&acc)[2][TN], const f32x16 (&accp)[2][TN], Epi& e, const float (&addp)[TN], int nt,
                                         int n_tiles) {
    if constexpr (KT_IDX < KT) {
      ktile<KT_IDX, HAS_PREV>(acc, accp, e, addp, nt, n_tiles);
      ktiles<KT_IDX + 1, HAS_PREV>(acc, accp, e, addp, nt, n_tiles);
    }
  }
};

template <int K, int BKB, int NW>
__global__ __launch_bounds__(64 * NW, NW / 4) void k_pointmlp_fwd_pool(const t3d_pointmlp_fwd_args p) {
  using F = FwdPool<K, BKB, NW>;
  constexpr int TN = F::TN, NTP = F::NTP;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  F f{p, smem, smem + F::A_FLOATS, smem + F::A_FLOATS + 2 * F::B_FLOATS, tid, lane, lane & 31, lane >> 5, wid / F::WN, wid % F::WN,
      0, 0, 0, 0u, {}};
  f.tile_m = xcd_remap(blockIdx.x, gridDim.x);
  f.row0 = f.tile_m * 128;
  const int b = f.row0 / p.rows_per_frustum;
  f.rin_base = f.row0 - b * p.rows_per_frustum + f.wm * 64 + 4 * f.h;
  const int n_tiles = p.N / 128;

  // ---- prologue: the activated input panel, once ----
  {
    ActLoader<false> la{p.a, K, p.rows_per_frustum};
    constexpr int CH = K / 4, NVA = 128 * CH / NTP;       // float4 chunks per row / per thread
    static_assert(NTP % CH == 0, "a thread keeps its column chunk");
    const int c4 = (tid % CH) * 4;
    const typename ActLoader<false>::Coef coef = la.fetch_coef(c4);
    typename ActLoader<false>::Raw raw[NVA];
#pragma unroll
    for (int q = 0; q < NVA; ++q) raw[q] = la.fetch(f.row0 + (tid + NTP * q) / CH, c4);
    f.fetch_b(0, 0);
#pragma unroll
    for (int q = 0; q < NVA; ++q) {
      const int r = (tid + NTP * q) / CH;
      *reinterpret_cast<float4*>(f.Ap + r * F::LDA + c4) = la.xform(raw[q], coef, f.row0 + r, c4);
    }
#pragma unroll
    for (int q = 0; q < F::NVB; ++q) f.store_b_piece(0, q);
    f.fetch_b(0, 1);
    // keep flags of this lane's 32 rows (the same rows for every column tile)
    unsigned bits = 0xffffffffu;
    if (p.rowmask) {
      bits = keep_bits32(p.rowmask, f.row0 + f.wm * 64 + 4 * f.h);
    }
    f.keepbits = bits;
  }
  __syncthreads();

  f32x16 acc[2][TN], accp[2][TN];
  zero_acc<2, TN>(acc);
  zero_acc<2, TN>(accp);
  typename F::Epi e;
  // bias (+ row bias) of a column tile is requested one whole column tile before its epilogue needs it: a wait on it
  // then never drags in the weight-tile loads that were issued only one barrier ago
  float addp[TN], addn[TN];
#pragma unroll
  for (int tn = 0; tn < TN; ++tn) addp[tn] = 0.f;
  f.load_add(0, addn);
  f.template ktiles<0, false>(acc, accp, e, addp, 0, n_tiles);
  for (int nt = 1; nt < n_tiles; ++nt) {
#pragma unroll
    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
      for (int tn = 0; tn < TN; ++tn) { accp[tm][tn] = acc[tm][tn]; }
    zero_acc<2, TN>(acc);
    f.epi_reset(e);
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) addp[tn] = addn[tn];
    f.load_add(nt, addn);
#ifdef T3D_ABL_FP_NOEPI
    f.template ktiles<0, false>(acc, accp, e, addp, nt, n_tiles);
#else
    f.template ktiles<0, true>(acc, accp, e, addp, nt, n_tiles);
#endif
  }
  // the last column tile: its epilogue has no MFMAs left to hide under
  f.epi_reset(e);
#pragma unroll
  for (int tn = 0; tn < TN; ++tn) addp[tn] = addn[tn];
#pragma unroll
  for (int quad = 0; quad < F::NQUADS; ++quad) f.epi_quad(acc, e, addp, quad);
  __syncthreads();                   // phase B of tile n_tiles-2 has read `red`
  f.epi_publish(e);
  __syncthreads();
  f.epi_write(n_tiles - 1);
}

// dynamic-LDS launch: two pipeline stages exceed the 64 KB static limit for the 128-wide tiles
// > 64 KB of dynamic LDS needs the function attribute; set once per kernel and size (a driver call per launch would
// sit on the host path of every eager launch)
void allow_lds(const void* kernel, size_t lds_bytes) {
  if (lds_bytes <= 64 * 1024) return;
  static std::mutex mu;
  static std::unordered_map<const void*, size_t> done;
  std::lock_guard<std::mutex> lock(mu);
  size_t& cur = done[kernel];
  if (cur >= lds_bytes) return;
  (void)hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
  cur = lds_bytes;
}

template <class Args>
void launch_lds(void (*kernel)(const Args), dim3 grid, size_t lds_bytes, hipStream_t s, const Args& a) {
  allow_lds(reinterpret_cast<const void*>(kernel), lds_bytes);
  T3D_LAUNCH(kernel, grid, dim3(NT), lds_bytes, s, a);
}
template <class Args>
void launch_lds_r(void (*kernel)(const Args, const t3d_rider_set), dim3 grid, size_t lds_bytes, hipStream_t s, const Args& a,
                  const t3d_rider_set& r) {
  allow_lds(reinterpret_cast<const void*>(kernel), lds_bytes);
  T3D_LAUNCH(kernel, grid, dim3(NT), lds_bytes, s, a, r);
}
template <class Args>
void launch_lds1(void (*kernel)(const Args), dim3 grid, size_t lds_bytes, hipStream_t s, const Args& a) {      // 512-thread kernels
  allow_lds(reinterpret_cast<const void*>(kernel), lds_bytes);
  T3D_LAUNCH(kernel, grid, dim3(NT1), lds_bytes, s, a);
}
constexpr size_t lds_fwd(int bn) { return 2 * (size_t)(128 * LDR + BK * bn) * sizeof(float); }
constexpr size_t lds_dgrad(int bn) { return 2 * (size_t)(128 * LDR + bn * LDR) * sizeof(float); }
constexpr size_t lds_wgrad(int bmk, int bn) { return 2 * (size_t)BK * (bmk + bn) * sizeof(float); }
// bf16 path: two stages of R images [dim][BKH + 8] / C images [BKH][dim + 32] of 2-byte elements; never less than the epilogues'
// cross-wave scratch (6 x 2 x BN floats)
constexpr size_t lds_epi(int bn) { return (size_t)12 * bn * sizeof(float) + (size_t)128 * (bn + 8) * 2; }      // red + the bf16 output tile
constexpr size_t lds_min(size_t b, int bn) { return b > lds_epi(bn) ? b : lds_epi(bn); }
constexpr size_t lds_fwd_h(int bn) { return lds_min(2 * (size_t)(128 * LDRH + BKH * (bn + 32)) * 2, bn); }
constexpr size_t lds_dgram_h(int bn) { return lds_fwd_h(bn) > (size_t)12 * bn * 4 + 2 * (size_t)128 * (bn + 8) * 2 ? lds_fwd_h(bn) : (size_t)12 * bn * 4 + 2 * (size_t)128 * (bn + 8) * 2; }
constexpr size_t lds_epi2(int bn) { return (size_t)12 * bn * sizeof(float) + 2 * (size_t)128 * (bn + 8) * 2; }  // red + prev_y/out tile + add_in tile
constexpr size_t lds_max(size_t a, size_t b) { return a > b ? a : b; }
constexpr size_t lds_dgrad_h(int bn) { return lds_max(2 * (size_t)(128 * LDRH + bn * LDRH) * 2, lds_epi2(bn)); }
constexpr size_t lds_wgrad_h(int bmk, int bn) { return 2 * (size_t)BKH * (bmk + 32 + bn + 32) * 2; }
// x3 path: two stages of three bf16 planes per operand (R image [dim][LDRX] / C image [BKX][dim + 32])
constexpr size_t lds_x3_r(int dim) { return (size_t)3 * dim * LDRX * 2; }
constexpr size_t lds_x3_c(int dim) { return (size_t)3 * BKX * (dim + LDCX_PAD) * 2; }
constexpr size_t lds_fwd_x3(int bn) { return lds_max(2 * (lds_x3_r(128) + lds_x3_c(bn)), (size_t)12 * bn * sizeof(float)); }
constexpr size_t lds_fwd_x3_pc(int bn) { return lds_max(X3_RING * (lds_x3_r(128) + lds_x3_c(bn)), (size_t)12 * bn * sizeof(float)); }
constexpr size_t lds_dgrad_x3(int bn) { return lds_max(2 * (lds_x3_r(128) + lds_x3_r(bn)), (size_t)12 * bn * sizeof(float)); }
constexpr size_t lds_wgrad_x3(int bmk, int bn) { return 2 * (lds_x3_c(bmk) + lds_x3_c(bn)); }
constexpr size_t lds_bwd1f(int k, int n) { return (size_t)(64 * ((n + 4) + 3 * (k + 4)) + 2 * 2 * 2 * k + 3 * n + 2 * k) * 4; }      // D, A, 2 x X images, statistics scratch, per-column constants
constexpr size_t lds_bwd1(int k, int n, int bm) { return (size_t)bm * (n + k + 2 * (k + 8)) * 2 + (size_t)2 * 2 * 4 * k * 4; }      // D, A, 2 x X images + the statistics scratch

bool dtype_ok(int dt) { return dt == T3D_F32 || dt == T3D_BF16; }
// Arithmetic of an fp32 launch (t3d.h: T3D_ARITH_*).  The request travels in the argument struct (`arith`), fixed by the host when it
// builds its plan; only T3D_ARITH_AUTO (a zero-initialised struct: the tools and the kernel tests) consults the environment -- T3D_X3=0
// the fp32-MFMA kernels, T3D_X3_MINKN / T3D_X3_MINKN_BWD: only launches with K x N at least that (measured on the B=32 N=1024 step, one
// MI355X: 1.440 ms fp32-MFMA, 1.304 with K x N >= 128 x 128, 1.295 with every launch) -- read at every launch so that one process can
// compare both.
bool x3_on(int arith) {
  if (arith == T3D_ARITH_FP32_MFMA) return false;
  if (arith == T3D_ARITH_BF16X3) return true;
  const char* e = getenv("T3D_X3");
  return e ? atoi(e) != 0 : true;
}
long x3_min_kn(int arith) { const char* e = arith == T3D_ARITH_AUTO ? getenv("T3D_X3_MINKN") : nullptr; return e ? atol(e) : 1L; }
long x3_min_kn_bwd(int arith) { const char* e = arith == T3D_ARITH_AUTO ? getenv("T3D_X3_MINKN_BWD") : nullptr; return e ? atol(e) : x3_min_kn(arith); }
// (K, N <= T3D_IDENT_MAX: the identity scale / shift tables of ActLoaderE)
bool x3_layer(int arith, int K, int N) { return x3_on(arith) && (long)K * N >= x3_min_kn(arith) && K <= 4096 && N <= 4096; }            // forward launches
// backward launches (dense and Gram form).  Until round 5 not the layers with few input and many output channels (64 -> 512, conv6's
// per-point part: their data gradient is ONE 64-column tile over a long reduction, and the compiler-scheduled x3 loop lost to the
// fp32-MFMA form, 54.1 vs 50.7 us alone); with the hand-placed iteration the x3 form wins there too (44.0 vs 52.1 us alone at
// M = 32768, 1.1874 vs 1.1947 ms per step, same box: profiles/r06_narrow_bwd.log).  T3D_X3_BWD_NARROW=0 restores the exclusion.
bool x3_bwd_narrow() { const char* e = getenv("T3D_X3_BWD_NARROW"); return e ? atoi(e) != 0 : true; }
bool x3_layer_bwd(int arith, int K, int N) {
  return x3_on(arith) && (long)K * N >= x3_min_kn_bwd(arith) && (N <= 4 * K || K >= 128 || x3_bwd_narrow()) && K <= 4096 && N <= 4096;
}
bool act_ok(const t3d_act_src& a, int K) {
  return a.x != nullptr && (a.ldx % 4) == 0 && (a.coff % 4) == 0 && a.coff + (K + 3) / 4 * 4 <= a.ldx &&
         (a.scale == nullptr || a.shift != nullptr) && dtype_ok(a.dtype) && !(a.dtype == T3D_BF16 && a.sub != nullptr) &&
         !(a.dtype == T3D_BF16 && ((a.ldx | a.coff) & 7));      // bf16 sources are read in 16-byte granules
}
bool dy_ok(const t3d_dy_src& d) {
  return d.y != nullptr && d.coef != nullptr && (d.dz != nullptr || (d.argidx != nullptr && d.dpool != nullptr)) && dtype_ok(d.dtype) &&
         !(d.dtype == T3D_BF16 && d.dz == nullptr);      // bf16: dense form only (pooled layers take the Gram path)
}

bool riders_ok(const t3d_rider_set* r) {
  return r->sync != nullptr && r->n_ops > 0 && r->n_ops <= T3D_RIDER_MAX_OPS && r->n_wg > 0 && r->n_wg <= (r->n_ops == 1 ? RIDER_MAX_WG_WIDE : RIDER_MAX_WG) && r->lds_bytes >= 0;
}
size_t lds_with(size_t lds, const t3d_rider_set* r) { return r && (size_t)r->lds_bytes > lds ? (size_t)r->lds_bytes : lds; }

}  // namespace

// ---------------------------------------------------------------------------------------------
// The T3D_X3 kernels are instantiated in a translation unit of their own (csrc/pointmlp_x3.hip = this file with T3D_X3_TU defined:
// the device templates above, these launch helpers, none of the entry points below): hipcc spends two minutes on this file as it is.
// ---------------------------------------------------------------------------------------------
int t3d_x3_fwd(const t3d_pointmlp_fwd_args* a, const t3d_rider_set* r, hipStream_t s);
int t3d_x3_bwd(const t3d_pointmlp_dgrad_args* d, const t3d_pointmlp_wgrad_args* w, const t3d_rider_set* r, int tk, int tn, bool wide,
               int n_w, int n_d, hipStream_t s);
int t3d_x3_stage1(const t3d_pointmlp_gram_args* g, const t3d_act_colsum_args* c, const t3d_pool_bwd_prep_args* q, const t3d_rider_set* r,
                  int gt, int n_gram, int n_colsum, int n_prep, size_t lds_other, hipStream_t s);
int t3d_x3_stage2(const t3d_pool_wgrad_finish_args* f, const t3d_pointmlp_dgrad_gram_args* d, const t3d_rider_set* r, bool wide,
                  int n_finish, int n_d, size_t lds_other, hipStream_t s);
int t3d_x3_split(const float* src, void* planes, int64_t n, int64_t plane_stride, hipStream_t s);
int t3d_x3_split_frag(const float* params, void* pf, void* pd, int64_t stride, const t3d_x3_frag_entry* tab, int n, int n_blocks, hipStream_t s);
int t3d_x3_dgrad(const t3d_pointmlp_dgrad_args* a, bool wide, hipStream_t s);
int t3d_x3_wgrad(const t3d_pointmlp_wgrad_args* a, int tk, int tn, int n_blocks, hipStream_t s);
int t3d_x3_gram(const t3d_pointmlp_gram_args* a, int tk, int n_blocks, hipStream_t s);
int t3d_x3_dgrad_gram(const t3d_pointmlp_dgrad_gram_args* a, bool wide, hipStream_t s);

#ifdef T3D_X3_TU
namespace {
__global__ __launch_bounds__(256) void k_split_x3(const float* __restrict__ src, bf16_t* __restrict__ planes, long n, long stride) {
  const long step = (long)gridDim.x * 256 * 4;
  for (long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += step) {
    if (i + 4 <= n) {
      bf16x4 h, m, l;
      split3(*reinterpret_cast<const float4*>(src + i), h, m, l);
      *reinterpret_cast<bf16x4*>(planes + i) = h;
      *reinterpret_cast<bf16x4*>(planes + stride + i) = m;
      *reinterpret_cast<bf16x4*>(planes + 2 * stride + i) = l;
    } else {
      for (long j = i; j < n; ++j) {
        bf16x4 h, m, l;
        split3(make_float4(src[j], 0.f, 0.f, 0.f), h, m, l);
        planes[j] = h[0]; planes[stride + j] = m[0]; planes[2 * stride + j] = l[0];
      }
    }
  }
}
}  // namespace
// the weights of every x3 layer as fragment-order planes (WLoaderX3F), both arrangements, ONE launch per step: entry e of the device
// table = a [K, N] matrix at params + off; its planes live at the same offset of the plane buffers (a plane is as long as `params`)
namespace {
__global__ __launch_bounds__(256) void k_split_x3_frag(const float* __restrict__ params, bf16_t* __restrict__ pf, bf16_t* __restrict__ pd, long stride,
                                                       const t3d_x3_frag_entry* __restrict__ tab, int n) {
  // block -> (matrix, its 256 fragments): the table carries each matrix's first block (ascending); every matrix in parallel (one
  // after the other, the ~20 matrices of a step took 16 us of dependent round trips for 20 MB)
  int e = 0;
  while (e + 1 < n && (int)blockIdx.x >= tab[e + 1].blk0) ++e;
  const t3d_x3_frag_entry en = tab[e];
  const float* w = params + en.off;
  const long nfr = (long)en.K * en.N / 8;      // eight-element fragments per plane
  const long f = (long)((int)blockIdx.x - en.blk0) * 256 + threadIdx.x;
  if (f >= nfr) return;
  const int lane = (int)(f & 63);
  const long q = f >> 6;
  float4 v0, v1, u0, u1;
  if (en.fwd) {             // Op[n][k] = w[k][n]: lane index n, reduction k
    const int NB = en.N / 32, nb = (int)(q % NB);
    const long k0 = (q / NB) * 16 + 8 * (lane >> 5);
    const float* src = w + k0 * en.N + nb * 32 + (lane & 31);
    v0 = make_float4(src[0], src[en.N], src[2 * (long)en.N], src[3 * (long)en.N]);
    v1 = make_float4(src[4 * (long)en.N], src[5 * (long)en.N], src[6 * (long)en.N], src[7 * (long)en.N]);
  }
  if (en.dgrad) {           // Op[k][n] = w[k][n]: lane index k, reduction n
    const int KB = en.K / 32, kb = (int)(q % KB);
    const long n0 = (q / KB) * 16 + 8 * (lane >> 5);
    const float* src = w + (long)(kb * 32 + (lane & 31)) * en.N + n0;
    u0 = *reinterpret_cast<const float4*>(src);
    u1 = *reinterpret_cast<const float4*>(src + 4);
  }
  bf16x4 h0, m0, l0, h1, m1, l1;
  if (en.fwd) {
    split3(v0, h0, m0, l0);
    split3(v1, h1, m1, l1);
    bf16_t* d = pf + en.off + f * 8;
    *reinterpret_cast<bf16x8*>(d) = __builtin_shufflevector(h0, h1, 0, 1, 2, 3, 4, 5, 6, 7);
    *reinterpret_cast<bf16x8*>(d + stride) = __builtin_shufflevector(m0, m1, 0, 1, 2, 3, 4, 5, 6, 7);
    *reinterpret_cast<bf16x8*>(d + 2 * stride) = __builtin_shufflevector(l0, l1, 0, 1, 2, 3, 4, 5, 6, 7);
  }
  if (en.dgrad) {
    split3(u0, h0, m0, l0);
    split3(u1, h1, m1, l1);
    bf16_t* d = pd + en.off + f * 8;
    *reinterpret_cast<bf16x8*>(d) = __builtin_shufflevector(h0, h1, 0, 1, 2, 3, 4, 5, 6, 7);
    *reinterpret_cast<bf16x8*>(d + stride) = __builtin_shufflevector(m0, m1, 0, 1, 2, 3, 4, 5, 6, 7);
    *reinterpret_cast<bf16x8*>(d + 2 * stride) = __builtin_shufflevector(l0, l1, 0, 1, 2, 3, 4, 5, 6, 7);
  }
}
}  // namespace
int t3d_x3_split_frag(const float* params, void* pf, void* pd, int64_t stride, const t3d_x3_frag_entry* tab, int n, int n_blocks, hipStream_t s) {
  T3D_LAUNCH(k_split_x3_frag, dim3(n_blocks), dim3(256), 0, s, params, static_cast<bf16_t*>(pf), static_cast<bf16_t*>(pd), (long)stride, tab, n);
  T3D_CHECK_LAUNCH();
  return T3D_OK;
}
int t3d_x3_split(const float* src, void* planes, int64_t n, int64_t plane_stride, hipStream_t s) {
  long blocks = (n + 1023) / 1024;
  if (blocks > 2048) blocks = 2048;
  T3D_LAUNCH(k_split_x3, dim3((unsigned)blocks), dim3(256), 0, s, src, static_cast<bf16_t*>(planes), (long)n, (long)plane_stride);
  T3D_CHECK_LAUNCH();
  return T3D_OK;
}

int t3d_x3_dgrad(const t3d_pointmlp_dgrad_args* a, bool wide, hipStream_t s) {
  const int tiles_m = a->M / 128;
  if (a->w_x3 && (a->K % 32 != 0 || a->N % 16 != 0 || a->w_x3_stride < (int64_t)a->K * a->N)) return T3D_ERR_SHAPE;      // (fragment planes: whole blocks)
  if (a->w_x3 && a->N <= 1536) {      // (dy's coefficient table in LDS: 3 N floats <= 18 KB)
    if (wide) launch_lds(k_pointmlp_dgrad<128, false, PathX3P>, dim3(tiles_m * (a->K / 128)), lds_dgrad_x3(128), s, *a);
    else launch_lds(k_pointmlp_dgrad<64, false, PathX3P>, dim3(tiles_m * (a->K / 64)), lds_dgrad_x3(64), s, *a);
  } else if (wide) launch_lds(k_pointmlp_dgrad<128, false, PathX3>, dim3(tiles_m * (a->K / 128)), lds_dgrad_x3(128), s, *a);
  else launch_lds(k_pointmlp_dgrad<64, false, PathX3>, dim3(tiles_m * (a->K / 64)), lds_dgrad_x3(64), s, *a);
  T3D_CHECK_LAUNCH();
  return T3D_OK;
}

int t3d_x3_wgrad(const t3d_pointmlp_wgrad_args* a, int tk, int tn, int n_blocks, hipStream_t s) {
  const dim3 grid(n_blocks);
#define T3D_WGX(TK, TN_) launch_lds(k_pointmlp_wgrad<TK, TN_, false, false, PathX3, float>, grid, lds_wgrad_x3(TK, TN_), s, *a)
  if (tk == 128 && tn == 128) T3D_WGX(128, 128);
  else if (tk == 128) T3D_WGX(128, 64);
  else if (tn == 128) T3D_WGX(64, 128);
  else T3D_WGX(64, 64);
#undef T3D_WGX
  T3D_CHECK_LAUNCH();
  return T3D_OK;
}

int t3d_x3_gram(const t3d_pointmlp_gram_args* a, int tk, int n_blocks, hipStream_t s) {
  if (tk != 64) return T3D_ERR_ARG;      // (64 x 64 tiles: the symmetric accumulation keeps three accumulator sets)
  launch_lds(k_pointmlp_gram<64, 64, PathX3>, dim3(n_blocks), lds_wgrad_x3(64, 64), s, *a);
  T3D_CHECK_LAUNCH();
  return T3D_OK;
}

int t3d_x3_dgrad_gram(const t3d_pointmlp_dgrad_gram_args* a, bool wide, hipStream_t s) {
  const int tiles_m = a->M / 128;
  if (wide) launch_lds(k_pointmlp_dgrad_gram<128, PathX3>, dim3(tiles_m * (a->K / 128)), lds_fwd_x3(128), s, *a);
  else launch_lds(k_pointmlp_dgrad_gram<64, PathX3>, dim3(tiles_m * (a->K / 64)), lds_fwd_x3(64), s, *a);
  T3D_CHECK_LAUNCH();
  return T3D_OK;
}

int t3d_x3_fwd(const t3d_pointmlp_fwd_args* a, const t3d_rider_set* r, hipStream_t s) {
  const int tiles_m = a->M / 128, nr = r ? r->n_wg : 0;
  // the weights arrive as three bf16 planes in fragment order (t3d_split_x3_frag, forward arrangement); the fragment kernels keep the
  // input's scale / shift table in LDS (2 K floats <= 18 KB: K <= 1536 -- a wider layer takes the in-kernel split)
  const bool pre = a->w_x3 != nullptr && a->K <= 1536;
  if (pre && (a->N % 32 != 0 || a->K % 16 != 0 || a->w_x3_stride < (int64_t)a->K * a->N)) return T3D_ERR_SHAPE;
  const char* e = getenv("T3D_X3_FWD128_MIN");      // (fewest 128-wide tiles for which the forward takes them; experiments)
  const long min_tiles = e ? atol(e) : 512;
  // eight-wave 128 x 256 tiles where a launch has at least two rounds of them (one workgroup per CU: with a single round nothing
  // covers a workgroup's prologue and epilogue, and 512 -> 256 at M = 32768 measured 57.5 us against 56.7; with two or more, 256 -> 512
  // 52.3 against 56.6 and 128 -> 1024 57.3 against 60.5, profiles/r05_w8.log).  T3D_X3_W8=0: never; =2: whenever N % 256 == 0
  const int w8 = []() { const char* e_ = getenv("T3D_X3_W8"); return e_ ? atoi(e_) : T3D_X3_W8_DEFAULT; }();
  if (w8 && a->N % 256 == 0 && (w8 == 2 || (long)tiles_m * (a->N / 256) >= 512)) {
    const int n_tiles = tiles_m * (a->N / 256);
    const dim3 grid((T3D_W8_PERSIST && n_tiles > T3D_FAIR_CUS ? T3D_FAIR_CUS : n_tiles) + nr);
    const size_t lds = lds_with(lds_fwd_x3(256), r);
    if (r && pre) {
      auto kern = k_pointmlp_fwd_w8_r<256, PathX3WP>;
      allow_lds(reinterpret_cast<const void*>(kern), lds);
      T3D_LAUNCH(kern, grid, dim3(2 * NT), lds, s, *a, *r, n_tiles);
    } else if (r) {
      auto kern = k_pointmlp_fwd_w8_r<256, PathX3W>;
      allow_lds(reinterpret_cast<const void*>(kern), lds);
      T3D_LAUNCH(kern, grid, dim3(2 * NT), lds, s, *a, *r, n_tiles);
    } else if (pre) {
      auto kern = k_pointmlp_fwd_w8<256, PathX3WP>;
      allow_lds(reinterpret_cast<const void*>(kern), lds);
      T3D_LAUNCH(kern, grid, dim3(2 * NT), lds, s, *a, n_tiles);
    } else {
      auto kern = k_pointmlp_fwd_w8<256, PathX3W>;
      allow_lds(reinterpret_cast<const void*>(kern), lds);
      T3D_LAUNCH(kern, grid, dim3(2 * NT), lds, s, *a, n_tiles);
    }
    T3D_CHECK_LAUNCH();
    return T3D_OK;
  }
  const int pc = []() { const char* e_ = getenv("T3D_X3_PC"); return e_ ? atoi(e_) : 0; }();      // producer / consumer kernels (experiment; read per launch like T3D_X3)
  if (pc && !r && !pre) {
    const bool wide = a->N % 128 == 0 && (long)tiles_m * (a->N / 128) >= (pc == 2 ? 1 : min_tiles);
    if (wide) {
      auto kern = k_pointmlp_fwd_pc<128, PathX3PC>;
      allow_lds(reinterpret_cast<const void*>(kern), lds_fwd_x3_pc(128));
      T3D_LAUNCH(kern, dim3(tiles_m * (a->N / 128)), dim3(2 * NT), lds_fwd_x3_pc(128), s, *a);
    } else {
      auto kern = k_pointmlp_fwd_pc<64, PathX3PC>;
      allow_lds(reinterpret_cast<const void*>(kern), lds_fwd_x3_pc(64));
      T3D_LAUNCH(kern, dim3(tiles_m * (a->N / 64)), dim3(2 * NT), lds_fwd_x3_pc(64), s, *a);
    }
    T3D_CHECK_LAUNCH();
    return T3D_OK;
  }
  if (a->N % 128 == 0 && (long)tiles_m * (a->N / 128) >= min_tiles) {
    const dim3 grid(tiles_m * (a->N / 128) + nr);
    if (r && pre) launch_lds_r(k_pointmlp_fwd_r<128, false, PathX3P>, grid, lds_with(lds_fwd_x3(128), r), s, *a, *r);
    else if (r) launch_lds_r(k_pointmlp_fwd_r<128, false, PathX3>, grid, lds_with(lds_fwd_x3(128), r), s, *a, *r);
    else if (pre) launch_lds(k_pointmlp_fwd<128, false, PathX3P, float>, grid, lds_fwd_x3(128), s, *a);
    else launch_lds(k_pointmlp_fwd<128, false, PathX3, float>, grid, lds_fwd_x3(128), s, *a);
  } else {
    const dim3 grid(tiles_m * (a->N / 64) + nr);
    if (r && pre) launch_lds_r(k_pointmlp_fwd_r<64, false, PathX3P>, grid, lds_with(lds_fwd_x3(64), r), s, *a, *r);
    else if (r) launch_lds_r(k_pointmlp_fwd_r<64, false, PathX3>, grid, lds_with(lds_fwd_x3(64), r), s, *a, *r);
    else if (pre) launch_lds(k_pointmlp_fwd<64, false, PathX3P, float>, grid, lds_fwd_x3(64), s, *a);
    else launch_lds(k_pointmlp_fwd<64, false, PathX3, float>, grid, lds_fwd_x3(64), s, *a);
  }
  T3D_CHECK_LAUNCH();
  return T3D_OK;
}

int t3d_x3_bwd(const t3d_pointmlp_dgrad_args* d, const t3d_pointmlp_wgrad_args* w, const t3d_rider_set* r, int tk, int tn, bool wide,
               int n_w, int n_d, hipStream_t s) {
  if (d->w_x3 && (d->K % 32 != 0 || d->N % 16 != 0 || d->w_x3_stride < (int64_t)d->K * d->N)) return T3D_ERR_SHAPE;
  const dim3 grid(n_w + n_d + (r ? r->n_wg : 0));
#define T3D_BWDX_P(DBN, TK, TN_, PR_)                                                             \
  do {                                                                                             \
    const size_t lds = lds_max(lds_dgrad_x3(DBN), lds_wgrad_x3(TK, TN_));                          \
    if (r) {                                                                                       \
      auto kern = k_pointmlp_bwd_r<DBN, TK, TN_, PR_>;                                             \
      allow_lds(reinterpret_cast<const void*>(kern), lds_with(lds, r));                            \
      T3D_LAUNCH(kern, grid, dim3(NT), lds_with(lds, r), s, *d, *w, n_w, *r);                      \
    } else {                                                                                       \
      auto kern = k_pointmlp_bwd<DBN, TK, TN_, PR_>;                                               \
      allow_lds(reinterpret_cast<const void*>(kern), lds);                                         \
      T3D_LAUNCH(kern, grid, dim3(NT), lds, s, *d, *w, n_w, 0);                                    \
    }                                                                                              \
  } while (0)
#define T3D_BWDX(DBN, TK, TN_) do { if (d->w_x3 && d->N <= 1536) T3D_BWDX_P(DBN, TK, TN_, PathX3P); else T3D_BWDX_P(DBN, TK, TN_, PathX3); } while (0)
#define T3D_BWDX_W(DBN)                                  \
  do {                                                   \
    if (tk == 128 && tn == 128) T3D_BWDX(DBN, 128, 128); \
    else if (tk == 128) T3D_BWDX(DBN, 128, 64);          \
    else if (tn == 128) T3D_BWDX(DBN, 64, 128);          \
    else T3D_BWDX(DBN, 64, 64);                          \
  } while (0)
  if (wide) T3D_BWDX_W(128);
  else T3D_BWDX_W(64);
#undef T3D_BWDX_W
#undef T3D_BWDX
#undef T3D_BWDX_P
  T3D_CHECK_LAUNCH();
  return T3D_OK;
}

int t3d_x3_stage1(const t3d_pointmlp_gram_args* g, const t3d_act_colsum_args* c, const t3d_pool_bwd_prep_args* q, const t3d_rider_set* r,
                  int gt, int n_gram, int n_colsum, int n_prep, size_t lds_other, hipStream_t s) {
  const dim3 grid(n_gram + n_colsum + n_prep + (r ? r->n_wg : 0));
  const size_t lds = lds_with(lds_max(lds_wgrad_x3(gt, gt), lds_other), r);
#define T3D_ST1X(GT_)                                                                  \
  do {                                                                                 \
    if (r) {                                                                           \
      auto kern = k_pool_bwd_stage1_r<GT_, PathX3>;                                    \
      allow_lds(reinterpret_cast<const void*>(kern), lds);                             \
      T3D_LAUNCH(kern, grid, dim3(NT), lds, s, *g, *c, *q, n_gram, n_colsum, *r);      \
    } else {                                                                           \
      auto kern = k_pool_bwd_stage1<GT_, PathX3>;                                      \
      allow_lds(reinterpret_cast<const void*>(kern), lds);                             \
      T3D_LAUNCH(kern, grid, dim3(NT), lds, s, *g, *c, *q, n_gram, n_colsum);          \
    }                                                                                  \
  } while (0)
  if (gt != 64) return T3D_ERR_ARG;      // (see t3d_x3_gram)
  T3D_ST1X(64);
#undef T3D_ST1X
  T3D_CHECK_LAUNCH();
  return T3D_OK;
}

int t3d_x3_stage2(const t3d_pool_wgrad_finish_args* f, const t3d_pointmlp_dgrad_gram_args* d, const t3d_rider_set* r, bool wide,
                  int n_finish, int n_d, size_t lds_other, hipStream_t s) {
  const dim3 grid(n_finish + n_d + (r ? r->n_wg : 0));
  const size_t lds = lds_with(lds_max(wide ? lds_fwd_x3(128) : lds_fwd_x3(64), lds_other), r);
#define T3D_ST2X(BN_)                                                                  \
  do {                                                                                 \
    if (r) {                                                                           \
      auto kern = k_pool_bwd_stage2_r<BN_, PathX3>;                                    \
      allow_lds(reinterpret_cast<const void*>(kern), lds);                             \
      T3D_LAUNCH(kern, grid, dim3(NT), lds, s, *f, *d, n_finish, *r);                  \
    } else {                                                                           \
      auto kern = k_pool_bwd_stage2<BN_, PathX3>;                                      \
      allow_lds(reinterpret_cast<const void*>(kern), lds);                             \
      T3D_LAUNCH(kern, grid, dim3(NT), lds, s, *f, *d, n_finish);                      \
    }                                                                                  \
  } while (0)
  if (wide) T3D_ST2X(128); else T3D_ST2X(64);
#undef T3D_ST2X
  T3D_CHECK_LAUNCH();
  return T3D_OK;
}
#else

// "would this launch host a rider set inside the GEMM's grid?" is answered by the launchers themselves (t3d_*_hosts_riders below
// call them with this pseudo-stream and a dummy set): at every point where the variant is decided they return 1 (rider form) or 0
// (the set would run as a launch of its own) instead of launching -- the step scheduler needs no mirror of the dispatch rules.
#define T3D_QUERY_STREAM (reinterpret_cast<t3d_stream_t>(static_cast<intptr_t>(-1)))
#define T3D_HOSTED(r, stream) do { if ((stream) == T3D_QUERY_STREAM) return (r) ? 1 : 0; } while (0)
// a launch whose kernel variant has no rider form: the set runs as a launch of its own in front of it
#define T3D_RIDERS_FIRST(r, stream)                         \
  do {                                                      \
    if ((stream) == T3D_QUERY_STREAM) return 0;             \
    if (r) {                                                \
      const int rc__ = t3d_run_riders(r, stream);           \
      if (rc__ != T3D_OK) return rc__;                      \
    }                                                       \
  } while (0)

extern "C" int t3d_pointmlp_fwd(const t3d_pointmlp_fwd_args* a, t3d_stream_t stream) { return t3d_pointmlp_fwd_r(a, nullptr, stream); }

extern "C" int t3d_pointmlp_fwd_r(const t3d_pointmlp_fwd_args* a, const t3d_rider_set* r, t3d_stream_t stream) {
  T3D_ABI_TAKE(pointmlp_fwd_args, a);
  if (a && a->struct_size != sizeof(*a)) return T3D_ERR_ABI;
  if (r && !riders_ok(r)) return T3D_ERR_ARG;
  if (!a || !a->w || !a->psum || !a->psumsq || !act_ok(a->a, a->K)) return T3D_ERR_ARG;
  if (a->pmax && (!a->pmin || !a->pamax || !a->pamin)) return T3D_ERR_ARG;
  if (a->M <= 0 || a->K <= 0 || a->N <= 0 || a->M % T3D_TILE_ROWS || a->rows_per_frustum % T3D_TILE_ROWS ||
      a->M % a->rows_per_frustum || a->N % 64 || (long)a->M * a->N >= (1L << 30))
    return T3D_ERR_SHAPE;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int tiles_m = a->M / 128;
  // 128-wide column tiles only when they still give >= 2 workgroups per CU
  const bool sub = a->a.sub != nullptr;
  if (!dtype_ok(a->dtype)) return T3D_ERR_ARG;
  if (a->dtype == T3D_BF16) {
    T3D_RIDERS_FIRST(r, stream);
    // bf16 storage + bf16 MFMA (configs[4]); the input is fp32 for the raw point cloud / Box-PC representation, bf16 for a layer output
    const bool wide = T3D_FORCE_TILE != 64 && a->N % 128 == 0 && (T3D_FORCE_TILE == 128 || (long)tiles_m * (a->N / 128) >= 512);
    const bool xh = a->a.dtype == T3D_BF16;
    if (xh && a->K % BKH) return T3D_ERR_SHAPE;      // a bf16 input is a layer output: whole 64-deep k-tiles (the loaders do not mask)
    // first layer of a net (xyz [+ 1 channel], fp32 source): the register kernel (T3D_FWD_TINYK=0: the generic one)
    static const bool use_tiny = []() { const char* e = getenv("T3D_FWD_TINYK"); return !(e && e[0] == '0'); }();
    if (use_tiny && !xh && a->K <= 4 && a->y && !a->pmax && !a->rowbias && (a->N == 64 || a->N == 128) && a->a.ldx % 4 == 0 &&
        a->a.coff % 4 == 0 && a->a.coff + 4 <= a->a.ldx) {
      if (a->N == 64) T3D_LAUNCH(k_pointmlp_fwd_tinyk<64>, dim3(tiles_m), dim3(NT), 0, s, *a);
      else T3D_LAUNCH(k_pointmlp_fwd_tinyk<128>, dim3(tiles_m), dim3(NT), 0, s, *a);
      T3D_CHECK_LAUNCH();
      return T3D_OK;
    }
    // activation-resident kernel: the input panel is transformed once for all column tiles (T3D_FWD_RES=0: the generic kernel)
    static const bool use_res = []() { const char* e = getenv("T3D_FWD_RES"); return !(e && e[0] == '0'); }();
    // K = 256 (a 74 KB panel: one workgroup per CU) measured SLOWER than the generic kernel (256 -> 512: 257 vs 194 us, 256 -> 128:
    // 82 vs 61 us at M = 262144) and its instantiation was dropped in round 3; K <= 128 keeps two workgroups per CU: 128 -> 1024
    // 272 -> 219 us, 64 -> 512 144 -> 107 us, 128 -> 256 73 -> 67 us, the small layers unchanged (round 2; round 3: see the kernel).
    if (use_res && xh && !sub && (a->K == 64 || a->K == 128)) {
      // persistent workgroups (two per CU) walking the row tiles; T3D_FWD_RES_WGS=0: one workgroup per tile as before
      static const int res_wgs = []() { const char* e = getenv("T3D_FWD_RES_WGS"); return e ? atoi(e) : 512; }();
      const dim3 grid(res_wgs > 0 && tiles_m > res_wgs ? res_wgs : tiles_m);
#define T3D_FWD_RES(BN_, KT_)                                                                                    \
  do {                                                                                                          \
    constexpr size_t lds = ((size_t)KT_ * 128 * LDRH + 2 * (size_t)BKH * (BN_ + 32)) * 2;                        \
    launch_lds(k_pointmlp_fwd_res<BN_, KT_>, grid, lds, s, *a);                                                 \
  } while (0)
      if (a->N % 128 == 0) { if (a->K == 64) T3D_FWD_RES(128, 1); else T3D_FWD_RES(128, 2); }
      else { if (a->K == 64) T3D_FWD_RES(64, 1); else T3D_FWD_RES(64, 2); }
#undef T3D_FWD_RES
      T3D_CHECK_LAUNCH();
      return T3D_OK;
    }
    const dim3 grid(tiles_m * (a->N / (wide ? 128 : 64)));
    const size_t lds = wide ? lds_fwd_h(128) : lds_fwd_h(64);
#define T3D_FWD_H(BN_)                                                                                          \
  do {                                                                                                          \
    if (sub) launch_lds(k_pointmlp_fwd<BN_, true, PathBF16, float>, grid, lds, s, *a);                          \
    else if (xh) launch_lds(k_pointmlp_fwd<BN_, false, PathBF16, bf16_t>, grid, lds, s, *a);                    \
    else launch_lds(k_pointmlp_fwd<BN_, false, PathBF16, float>, grid, lds, s, *a);                             \
  } while (0)
    if (wide) T3D_FWD_H(128);
    else T3D_FWD_H(64);
#undef T3D_FWD_H
    T3D_CHECK_LAUNCH();
    return T3D_OK;
  }
  if (a->a.dtype != T3D_F32) return T3D_ERR_ARG;
  {   // first layer of a net (xyz [+ 1 channel]): the register kernel (T3D_FWD_TINYK=0: the generic one)
    static const bool use_tiny32 = []() { const char* e = getenv("T3D_FWD_TINYK"); return !(e && e[0] == '0'); }();
    if (use_tiny32 && a->K <= 4 && a->y && !a->pmax && !a->rowbias && (a->N == 64 || a->N == 128) && a->a.ldx % 4 == 0 &&
        a->a.coff % 4 == 0 && a->a.coff + 4 <= a->a.ldx) {
      T3D_RIDERS_FIRST(r, stream);
      if (a->N == 64) T3D_LAUNCH((k_pointmlp_fwd_tinyk<64, float>), dim3(tiles_m), dim3(NT), 0, s, *a);
      else T3D_LAUNCH((k_pointmlp_fwd_tinyk<128, float>), dim3(tiles_m), dim3(NT), 0, s, *a);
      T3D_CHECK_LAUNCH();
      return T3D_OK;
    }
  }
  if (!sub && a->K % BKX == 0 && x3_layer(a->arith, a->K, a->N)) {      // (ahead of the A-resident fp32 kernel)
    T3D_HOSTED(r, stream);
    return t3d_x3_fwd(a, r, s);
  }
  // max-pooled layer without an output tensor: the A-resident persistent kernel (T3D_FWD_POOL=0: the generic one)
  static const bool use_pool_kernel = []() { const char* e = getenv("T3D_FWD_POOL"); return !(e && e[0] == '0'); }();
  // K = 256 needs 16-deep weight tiles to fit the 133 KB panel next to them and measured slower than the generic kernel
  // (109 vs 100 us on 256->512); K = 128: 92 vs 100 us on 128->1024, 26 vs 33 us on 128->256.  T3D_FWD_POOL=2 forces it on.
  static const bool fp_k256 = []() { const char* e = getenv("T3D_FWD_POOL"); return e && e[0] == '2'; }();
  if (use_pool_kernel && !a->y && a->pmax && !sub && a->N % 128 == 0 && a->N >= 256 && (a->K == 128 || (a->K == 256 && fp_k256))) {
    T3D_RIDERS_FIRST(r, stream);      // (a 512-thread kernel)
#ifndef T3D_FP_WAVES
#define T3D_FP_WAVES 8
#endif
    if (a->K == 128) {
      auto kern = k_pointmlp_fwd_pool<128, 32, T3D_FP_WAVES>;
      constexpr size_t lds = FwdPool<128, 32, T3D_FP_WAVES>::LDS_BYTES;
      allow_lds(reinterpret_cast<const void*>(kern), lds);
      T3D_LAUNCH(kern, dim3(tiles_m), dim3(64 * T3D_FP_WAVES), lds, s, *a);
    } else {
      auto kern = k_pointmlp_fwd_pool<256, 16, T3D_FP_WAVES>;
      constexpr size_t lds = FwdPool<256, 16, T3D_FP_WAVES>::LDS_BYTES;
      allow_lds(reinterpret_cast<const void*>(kern), lds);
      T3D_LAUNCH(kern, dim3(tiles_m), dim3(64 * T3D_FP_WAVES), lds, s, *a);
    }
    T3D_CHECK_LAUNCH();
    return T3D_OK;
  }
  T3D_HOSTED(r, stream);
  if (T3D_FORCE_TILE != 64 && a->N % 128 == 0 && (T3D_FORCE_TILE == 128 || (long)tiles_m * (a->N / 128) >= 512)) {
    const dim3 grid(tiles_m * (a->N / 128) + (r ? r->n_wg : 0));
    if (r) {
      if (sub) launch_lds_r(k_pointmlp_fwd_r<128, true>, grid, lds_with(lds_fwd(128), r), s, *a, *r);
      else launch_lds_r(k_pointmlp_fwd_r<128, false>, grid, lds_with(lds_fwd(128), r), s, *a, *r);
    } else if (sub) launch_lds(k_pointmlp_fwd<128, true>, grid, lds_fwd(128), s, *a);
    else launch_lds(k_pointmlp_fwd<128, false>, grid, lds_fwd(128), s, *a);
  } else {
    const dim3 grid(tiles_m * (a->N / 64) + (r ? r->n_wg : 0));
    if (r) {
      if (sub) launch_lds_r(k_pointmlp_fwd_r<64, true>, grid, lds_with(lds_fwd(64), r), s, *a, *r);
      else launch_lds_r(k_pointmlp_fwd_r<64, false>, grid, lds_with(lds_fwd(64), r), s, *a, *r);
    } else if (sub) launch_lds(k_pointmlp_fwd<64, true>, grid, lds_fwd(64), s, *a);
    else launch_lds(k_pointmlp_fwd<64, false>, grid, lds_fwd(64), s, *a);
  }
  T3D_CHECK_LAUNCH();
  return T3D_OK;
}

static int check_dgrad(const t3d_pointmlp_dgrad_args* a) {
  if (a && a->struct_size != sizeof(*a)) return T3D_ERR_ABI;
  if (!a || !a->w || !a->out || !dy_ok(a->dy)) return T3D_ERR_ARG;
  if (a->prev_y && (!a->prev_scale || !a->prev_shift)) return T3D_ERR_ARG;
  if (a->psum_dz && (!a->psum_dzy || !a->prev_y)) return T3D_ERR_ARG;
  if (!dtype_ok(a->dtype) || a->dtype != a->dy.dtype) return T3D_ERR_ARG;
  if (a->M <= 0 || a->M % T3D_TILE_ROWS || a->rows_per_frustum % T3D_TILE_ROWS || a->M % a->rows_per_frustum ||
      a->K % 64 || a->N % 4 || (long)a->M * a->K >= (1L << 30) || (a->dtype == T3D_BF16 && a->N % 64))
    return T3D_ERR_SHAPE;
  return T3D_OK;
}
static bool dgrad_wide(const t3d_pointmlp_dgrad_args* a) {
  // (T3D_DGRAD_WIDE_MIN: fewest 128-wide tiles for which the data gradient takes them; experiments)
  const char* e = getenv("T3D_DGRAD_WIDE_MIN");
  const long min_tiles = e ? atol(e) : 512;
  return T3D_FORCE_TILE != 64 && a->K % 128 == 0 && (T3D_FORCE_TILE == 128 || (long)(a->M / 128) * (a->K / 128) >= min_tiles);
}

extern "C" int t3d_pointmlp_dgrad(const t3d_pointmlp_dgrad_args* a, t3d_stream_t stream) {
  T3D_ABI_TAKE(pointmlp_dgrad_args, a);
  const int rc = check_dgrad(a);
  if (rc != T3D_OK) return rc;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int tiles_m = a->M / 128;
  const bool pooled = a->dy.dz == nullptr;
  if (a->dtype == T3D_BF16) {
    if (dgrad_wide(a)) launch_lds(k_pointmlp_dgrad<128, false, PathBF16>, dim3(tiles_m * (a->K / 128)), lds_dgrad_h(128), s, *a);
    else launch_lds(k_pointmlp_dgrad<64, false, PathBF16>, dim3(tiles_m * (a->K / 64)), lds_dgrad_h(64), s, *a);
    T3D_CHECK_LAUNCH();
    return T3D_OK;
  }
  if (!pooled && a->N % BKX == 0 && x3_layer_bwd(a->arith, a->K, a->N)) return t3d_x3_dgrad(a, dgrad_wide(a), s);
  if (dgrad_wide(a)) {
    const dim3 grid(tiles_m * (a->K / 128));
    if (pooled) launch_lds(k_pointmlp_dgrad<128, true>, grid, lds_dgrad(128), s, *a);
    else launch_lds(k_pointmlp_dgrad<128, false>, grid, lds_dgrad(128), s, *a);
  } else {
    const dim3 grid(tiles_m * (a->K / 64));
    if (pooled) launch_lds(k_pointmlp_dgrad<64, true>, grid, lds_dgrad(64), s, *a);
    else launch_lds(k_pointmlp_dgrad<64, false>, grid, lds_dgrad(64), s, *a);
  }
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
  // number of weight-gradient workgroups aimed at: more splits = shorter workgroups but more slab traffic (every split writes a
  // K x N slab that the slab reducer reads back)
  static const long target = []() { const char* e = getenv("T3D_WGRAD_TARGET"); return e ? atol(e) : 384L; }();
  const int cand[4][2] = {{128, 128}, {128, 64}, {64, 128}, {64, 64}};
  int tk = 64, tn = 64;
  long tiles = (long)((K + 63) / 64) * (N / 64);
  for (int i = 0; i < 4; ++i) {
    const int ck = cand[i][0], cn = cand[i][1];
    if ((ck == 128 && K <= 64) || N % cn) continue;
    const long t = (long)((K + ck - 1) / ck) * (N / cn);
    if (t * cap >= target) { tk = ck; tn = cn; tiles = t; break; }
  }
  long want = (target + tiles - 1) / tiles;
  if (want > cap) want = cap;
  // first layers (K <= 4): bound by reading dz and y once, not by MFMA work -- 64-row splits give the register kernel
  // (k_pointmlp_wgrad_tinyk) two workgroups per CU at M = 32768 (at most 512 splits); their slabs are K x N <= 512 floats each
  if (K <= 4) { const long w4 = M / 64 < 512 ? M / 64 : 512; if (w4 > want) want = w4; }
  int s = 1;
  while ((long)s * 2 <= want && M % (s * 2) == 0 && (M / (s * 2)) % BKH == 0) s *= 2;      // whole k-tiles of either path
  *rows_per_split = M / s;
  *tile_k = tk;
  *tile_n = tn;
  return T3D_OK;
}

// Plan of the fused backward (t3d_pointmlp_bwd): bf16 layers with K, N in {64, 128} (and 256 x 128, 128 x 256) take the one-pass kernel -- one workgroup per
// run of 128-row tiles, one 512-thread workgroup per CU (fewer, longer runs mean fewer K x N slabs) --
// everything else the split form with t3d_wgrad_plan's row split.
static bool bwd1_shape(int M, int K, int N, int dtype) {
  static const bool on = []() { const char* e = getenv("T3D_BWD1"); return !(e && e[0] == '0'); }();
  static const bool onf = []() { const char* e = getenv("T3D_BWD1F"); return !(e && e[0] == '0'); }();
  const bool narrow = (K == 64 || K == 128) && (N == 64 || N == 128), mid = (K == 256 && N == 128) || (K == 128 && N == 256);
  if (dtype == T3D_F32) return onf && narrow && M % 128 == 0;
  return on && dtype == T3D_BF16 && (narrow || mid) && M % 128 == 0;
}
// fp32: the one-pass form pays from about four 64-row tiles per workgroup (measured per shape on MI355X, DESIGN.md section 4: at
// B=32 N=1024 -- 256 workgroups of two tiles -- it ties with the split form, its first load and its phase-synchronous tiles exposed);
// when the rows cannot fill 256 workgroups anyway (M < 32768) the choice does not matter and the one-pass form is taken.
static bool bwd1_split_ok(int M, int rps, int dtype) {
  static const bool force = []() { const char* e = getenv("T3D_BWD1F"); return e && e[0] == '2'; }();
  return dtype != T3D_F32 || force || M / 128 < 256 || rps >= 256;
}
extern "C" int t3d_bwd_plan(int M, int K, int N, int dtype, int* rows_per_split, int* one_pass) {
  if (!rows_per_split || !one_pass) return T3D_ERR_ARG;
  *one_pass = 0;
  if (bwd1_shape(M, K, N, dtype)) {
    static const long target = []() { const char* e = getenv("T3D_BWD1_WGS"); return e ? atol(e) : 256L; }();
    const int tiles = M / 128;
    int per = 1;
    while (tiles / per > target && tiles % (per * 2) == 0) per *= 2;
    if (bwd1_split_ok(M, 128 * per, dtype)) {
      *rows_per_split = 128 * per;
      *one_pass = 1;
      return T3D_OK;
    }
  }
  int tk = 0, tn = 0;
  return t3d_wgrad_plan(M, K, N, rows_per_split, &tk, &tn);
}

static int check_wgrad(const t3d_pointmlp_wgrad_args* a) {
  if (a && a->struct_size != sizeof(*a)) return T3D_ERR_ABI;
  if (!a || !a->slabs || !act_ok(a->a, a->K) || !dy_ok(a->dy)) return T3D_ERR_ARG;
  const int red = a->dy.dtype == T3D_BF16 ? BKH : BK;
  if (a->M <= 0 || a->rows_per_split <= 0 || a->rows_per_split % red || a->M % a->rows_per_split || a->N % 64 ||
      a->rows_per_frustum % red || a->M % a->rows_per_frustum)
    return T3D_ERR_SHAPE;
  if (a->dy.dtype == T3D_F32 && a->a.dtype != T3D_F32) return T3D_ERR_ARG;
  if (a->a.dtype == T3D_BF16 && a->K % 64) return T3D_ERR_SHAPE;      // bf16 input: whole tiles (the loaders do not mask)
  return T3D_OK;
}
// tile choice: the plan's tile if the caller used t3d_wgrad_plan's split, else by shape
static void wgrad_tile(const t3d_pointmlp_wgrad_args* a, int* tk, int* tn) {
  int rps = 0;
  if (a->M % 128 == 0 && t3d_wgrad_plan(a->M, a->K, a->N, &rps, tk, tn) == T3D_OK && rps == a->rows_per_split) return;
  *tk = a->K > 64 ? 128 : 64;
  *tn = a->N % 128 == 0 ? 128 : 64;
}

extern "C" int t3d_pointmlp_wgrad(const t3d_pointmlp_wgrad_args* a, t3d_stream_t stream) { return t3d_pointmlp_wgrad_r(a, nullptr, stream); }

extern "C" int t3d_pointmlp_wgrad_r(const t3d_pointmlp_wgrad_args* a, const t3d_rider_set* r, t3d_stream_t stream) {
  T3D_ABI_TAKE(pointmlp_wgrad_args, a);
  if (r && !riders_ok(r)) return T3D_ERR_ARG;
  const int rc = check_wgrad(a);
  if (rc != T3D_OK) return rc;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int splits = a->M / a->rows_per_split;
  int tk = 0, tn = 0;
  wgrad_tile(a, &tk, &tn);
  const int tiles_k = (a->K + tk - 1) / tk, tiles_n = a->N / tn;
  const bool sub = a->a.sub != nullptr, pooled = a->dy.dz == nullptr;
  // rider form: the 64 x 64 tiling of a first layer's weight gradient (K <= 4 raw channels, fp32) only
  // x3 form: its activation loader (ActLoaderE) has neither column clamp nor mask -- whole k-tiles of an fp32 source only (K = 192 with
  // 128-row tiles would read scale / shift / x past column K)
  const bool x3w = a->dy.dtype != T3D_BF16 && a->a.dtype == T3D_F32 && !sub && !pooled && a->K % tk == 0 && a->K % 64 == 0 &&
                   x3_layer_bwd(a->arith, a->K, a->N);      // (no rider form)
  // first layer of a net (K <= 4 raw channels): the register kernel, which has no rider form either and beats the generic kernel + a rider
  static const bool use_tiny_w = []() { const char* e = getenv("T3D_WGRAD_TINYK"); return !(e && e[0] == '0'); }();
  const bool tiny_w = use_tiny_w && a->dy.dtype != T3D_BF16 && !pooled && a->K <= 4 && (a->N == 64 || a->N == 128) && a->a.dtype == T3D_F32 &&
                      a->a.ldx % 4 == 0 && a->a.coff % 4 == 0 && a->a.coff + 4 <= a->a.ldx && (a->a.scale == nullptr) == (a->a.shift == nullptr);
  const bool host = r && a->dy.dtype != T3D_BF16 && tk == 64 && tn == 64 && !pooled && !x3w && !tiny_w;
  if (r && !host) { T3D_RIDERS_FIRST(r, stream); r = nullptr; }
  T3D_HOSTED(r, stream);
  const dim3 grid(tiles_k * tiles_n * splits + (r ? r->n_wg : 0));
  if (a->dy.dtype == T3D_BF16) {
    const bool xh = a->a.dtype == T3D_BF16;
#define T3D_WGH(TK, TN_)                                                                                                     \
  do {                                                                                                                       \
    if (sub) launch_lds(k_pointmlp_wgrad<TK, TN_, true, false, PathBF16, float>, grid, lds_wgrad_h(TK, TN_), s, *a);         \
    else if (xh) launch_lds(k_pointmlp_wgrad<TK, TN_, false, false, PathBF16, bf16_t>, grid, lds_wgrad_h(TK, TN_), s, *a);   \
    else launch_lds(k_pointmlp_wgrad<TK, TN_, false, false, PathBF16, float>, grid, lds_wgrad_h(TK, TN_), s, *a);            \
  } while (0)
    if (tk == 128 && tn == 128) T3D_WGH(128, 128);
    else if (tk == 128) T3D_WGH(128, 64);
    else if (tn == 128) T3D_WGH(64, 128);
    else T3D_WGH(64, 64);
#undef T3D_WGH
    T3D_CHECK_LAUNCH();
    return T3D_OK;
  }
#define T3D_WG(TK, TN_)                                                                                   \
  do {                                                                                                    \
    if (sub && pooled) launch_lds(k_pointmlp_wgrad<TK, TN_, true, true>, grid, lds_wgrad(TK, TN_), s, *a);        \
    else if (sub) launch_lds(k_pointmlp_wgrad<TK, TN_, true, false>, grid, lds_wgrad(TK, TN_), s, *a);             \
    else if (pooled) launch_lds(k_pointmlp_wgrad<TK, TN_, false, true>, grid, lds_wgrad(TK, TN_), s, *a);          \
    else launch_lds(k_pointmlp_wgrad<TK, TN_, false, false>, grid, lds_wgrad(TK, TN_), s, *a);                     \
  } while (0)
  if (tiny_w) {
    const dim3 gsp(splits);      // one workgroup per row split: the slab layout of the generic kernel
    if (a->N == 64) { if (sub) T3D_LAUNCH((k_pointmlp_wgrad_tinyk<64, true>), gsp, dim3(NT), 0, s, *a); else T3D_LAUNCH((k_pointmlp_wgrad_tinyk<64, false>), gsp, dim3(NT), 0, s, *a); }
    else { if (sub) T3D_LAUNCH((k_pointmlp_wgrad_tinyk<128, true>), gsp, dim3(NT), 0, s, *a); else T3D_LAUNCH((k_pointmlp_wgrad_tinyk<128, false>), gsp, dim3(NT), 0, s, *a); }
    T3D_CHECK_LAUNCH();
    return T3D_OK;
  }
  if (x3w) return t3d_x3_wgrad(a, tk, tn, tiles_k * tiles_n * splits, s);
  if (r) {
    if (sub) launch_lds_r(k_pointmlp_wgrad_r<64, 64, true, false>, grid, lds_with(lds_wgrad(64, 64), r), s, *a, *r);
    else launch_lds_r(k_pointmlp_wgrad_r<64, 64, false, false>, grid, lds_with(lds_wgrad(64, 64), r), s, *a, *r);
  } else if (tk == 128 && tn == 128) T3D_WG(128, 128);
  else if (tk == 128) T3D_WG(128, 64);
  else if (tn == 128) T3D_WG(64, 128);
  else T3D_WG(64, 64);
#undef T3D_WG
  T3D_CHECK_LAUNCH();
  return T3D_OK;
}

// One-pass forms of the Gram-form backward (gram1 / dgram1 above): bf16 input with K = 128 or 256 channels.
static bool gram1_shape(int M, int K, int dtype) {
  static const bool on = []() { const char* e = getenv("T3D_GRAM1"); return !(e && e[0] == '0'); }();
  return on && dtype == T3D_BF16 && (K == 128 || K == 256) && M % 128 == 0;
}
static bool act_chunks_ok(const t3d_act_src& a) { return a.dtype == T3D_BF16 && a.ldx % 8 == 0 && a.coff % 8 == 0 && (a.scale == nullptr) == (a.shift == nullptr); }
extern "C" int t3d_gram_plan(int M, int K, int dtype, int* rows_per_split, int* one_pass) {
  if (!rows_per_split || !one_pass) return T3D_ERR_ARG;
  *one_pass = 0;
  if (gram1_shape(M, K, dtype)) {
    const int tiles = M / 128;
    int per = 1;
    while (tiles / per > 256 && tiles % (per * 2) == 0) per *= 2;
    if (bwd1_split_ok(M, 128 * per, dtype)) {
      *rows_per_split = 128 * per;
      *one_pass = 1;
      return T3D_OK;
    }
  }
  int tk = 0, tn = 0;
  return t3d_wgrad_plan(M, K, K, rows_per_split, &tk, &tn);
}
static bool gram1_ok(const t3d_pointmlp_gram_args* g) {
  const int tiles = g->M / 128;
  return gram1_shape(g->M, g->K, g->a.dtype) && g->rows_per_split % 128 == 0 && g->M / g->rows_per_split >= (tiles < 256 ? tiles : 256) &&
         act_chunks_ok(g->a);
}
static bool dgram1_ok(const t3d_pointmlp_dgrad_gram_args* d) {
  // the raw input tile doubles as the producer's raw output (mask, partials): same tensor, dense rows; the added rows are the sparse form.
  // K = 256 only by default: at K = 128 the split form has a single column tile already (nothing is staged twice) and its 256-thread
  // blocks share the CUs with the dW-assembly blocks of stage 2, whose 24 us of gather latency the 512-thread form leaves exposed
  // (stage 2 at B=128 N=2048: 72 -> 81 us with K = 128, 167 -> 125 us with K = 256).  T3D_DGRAM1=2 takes it for K = 128 as well.
  static const int mode = []() { const char* e = getenv("T3D_DGRAM1"); return e ? atoi(e) : 1; }();
  if (mode == 0 || (mode == 1 && d->K != 256)) return false;
  return gram1_shape(d->M, d->K, d->dtype) && act_chunks_ok(d->a) && d->a.ldx == d->K && d->a.coff == 0 &&
         (d->prev_y == nullptr || d->prev_y == d->a.x) && (d->add_in == nullptr || d->add_live != nullptr);
}

static int check_dgrad_gram(const t3d_pointmlp_dgrad_gram_args* a) {
  if (a && a->struct_size != sizeof(*a)) return T3D_ERR_ABI;
  if (!a || !a->p || !a->out || !act_ok(a->a, a->K) || a->a.sub) return T3D_ERR_ARG;
  if (a->add_live && !a->add_in) return T3D_ERR_ARG;
  if (a->prev_y && (!a->prev_scale || !a->prev_shift)) return T3D_ERR_ARG;
  if (a->psum_dz && (!a->psum_dzy || !a->prev_y)) return T3D_ERR_ARG;
  if (!dtype_ok(a->dtype) || a->dtype != a->a.dtype) return T3D_ERR_ARG;
  if (a->M <= 0 || a->M % T3D_TILE_ROWS || a->rows_per_frustum % T3D_TILE_ROWS || a->M % a->rows_per_frustum ||
      a->K % 64 || (long)a->M * a->K >= (1L << 30))
    return T3D_ERR_SHAPE;
  return T3D_OK;
}
static bool dgrad_gram_wide(const t3d_pointmlp_dgrad_gram_args* a) {
  return a->K % 128 == 0 && (long)(a->M / 128) * (a->K / 128) >= 512;
}

extern "C" int t3d_pointmlp_dgrad_gram(const t3d_pointmlp_dgrad_gram_args* a, t3d_stream_t stream) {
  T3D_ABI_TAKE(pointmlp_dgrad_gram_args, a);
  const int rc = check_dgrad_gram(a);
  if (rc != T3D_OK) return rc;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int tiles_m = a->M / 128;
  if (dgram1_ok(a)) {
    const dim3 grid(tiles_m < 256 ? tiles_m : 256);
    if (a->K == 128) launch_lds1(k_dgram1<128, 128>, grid, lds_dgram1(128, 128), s, *a);
    else launch_lds1(k_dgram1<256, 64>, grid, lds_dgram1(256, 64), s, *a);
    T3D_CHECK_LAUNCH();
    return T3D_OK;
  }
  if (a->dtype == T3D_BF16) {
    if (dgrad_gram_wide(a)) launch_lds(k_pointmlp_dgrad_gram<128, PathBF16>, dim3(tiles_m * (a->K / 128)), lds_dgram_h(128), s, *a);
    else launch_lds(k_pointmlp_dgrad_gram<64, PathBF16>, dim3(tiles_m * (a->K / 64)), lds_dgram_h(64), s, *a);
  } else if (a->K % BKX == 0 && x3_layer_bwd(a->arith, a->K, a->K)) {
    return t3d_x3_dgrad_gram(a, dgrad_gram_wide(a), s);
  } else if (dgrad_gram_wide(a))
    launch_lds(k_pointmlp_dgrad_gram<128>, dim3(tiles_m * (a->K / 128)), lds_fwd(128), s, *a);
  else
    launch_lds(k_pointmlp_dgrad_gram<64>, dim3(tiles_m * (a->K / 64)), lds_fwd(64), s, *a);
  T3D_CHECK_LAUNCH();
  return T3D_OK;
}

static int check_gram(const t3d_pointmlp_gram_args* a) {
  if (a && a->struct_size != sizeof(*a)) return T3D_ERR_ABI;
  if (!a || !a->slabs || !act_ok(a->a, a->K) || a->a.sub) return T3D_ERR_ARG;
  const int red = a->a.dtype == T3D_BF16 ? BKH : BK;
  if (a->M <= 0 || a->rows_per_split <= 0 || a->rows_per_split % red || a->M % a->rows_per_split || a->K % 64 ||
      a->rows_per_frustum % red || a->M % a->rows_per_frustum)
    return T3D_ERR_SHAPE;
  return T3D_OK;
}
static int gram_tile(const t3d_pointmlp_gram_args* a) {      // square tiles only: two instantiations
  int rps = 0, tk = 0, tn = 0;
  if (!(a->M % 128 == 0 && t3d_wgrad_plan(a->M, a->K, a->K, &rps, &tk, &tn) == T3D_OK && rps == a->rows_per_split))
    tk = tn = (a->K % 128 == 0 ? 128 : 64);
  return (tk == 128 && tn == 128) ? 128 : 64;
}

extern "C" int t3d_pointmlp_gram(const t3d_pointmlp_gram_args* a, t3d_stream_t stream) {
  T3D_ABI_TAKE(pointmlp_gram_args, a);
  if (check_gram(a) == T3D_OK && gram1_ok(a)) {
    const dim3 grid(a->M / a->rows_per_split);
    if (a->K == 128) launch_lds1(k_gram1<128, 128>, grid, lds_gram1(128, 128), static_cast<hipStream_t>(stream), *a);
    else launch_lds1(k_gram1<256, 64>, grid, lds_gram1(256, 64), static_cast<hipStream_t>(stream), *a);
    T3D_CHECK_LAUNCH();
    return T3D_OK;
  }
  const int rc = check_gram(a);
  if (rc != T3D_OK) return rc;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int splits = a->M / a->rows_per_split;
  const bool x3g = a->a.dtype != T3D_BF16 && x3_layer_bwd(a->arith, a->K, a->K);
  const int tk = x3g ? 64 : gram_tile(a), tn = tk;
  const dim3 grid((a->K / tk) * (a->K / tn) * splits);
  if (a->a.dtype == T3D_BF16) {
    if (tk == 128) launch_lds(k_pointmlp_gram<128, 128, PathBF16>, grid, lds_wgrad_h(128, 128), s, *a);
    else launch_lds(k_pointmlp_gram<64, 64, PathBF16>, grid, lds_wgrad_h(64, 64), s, *a);
  } else if (x3g) {
    return t3d_x3_gram(a, tk, (a->K / tk) * (a->K / tn) * splits, s);
  } else if (tk == 128) launch_lds(k_pointmlp_gram<128, 128>, grid, lds_wgrad(128, 128), s, *a);
  else launch_lds(k_pointmlp_gram<64, 64>, grid, lds_wgrad(64, 64), s, *a);
  T3D_CHECK_LAUNCH();
  return T3D_OK;
}

extern "C" int t3d_pointmlp_bwd(const t3d_pointmlp_dgrad_args* d, const t3d_pointmlp_wgrad_args* w, t3d_stream_t stream) {
  return t3d_pointmlp_bwd_r(d, w, nullptr, stream);
}

extern "C" int t3d_pointmlp_bwd_r(const t3d_pointmlp_dgrad_args* d, const t3d_pointmlp_wgrad_args* w, const t3d_rider_set* r,
                                  t3d_stream_t stream) {
  T3D_ABI_TAKE(pointmlp_dgrad_args, d);
  T3D_ABI_TAKE(pointmlp_wgrad_args, w);
  if (r && !riders_ok(r)) return T3D_ERR_ARG;
  int rc = check_dgrad(d);
  if (rc != T3D_OK) return rc;
  rc = check_wgrad(w);
  if (rc != T3D_OK) return rc;
  // the fused launch covers the dense form only (pooled layers take the Gram path, raw-point layers have no dgrad)
  if (!d->dy.dz || !w->dy.dz || w->a.sub) return T3D_ERR_ARG;
  if (d->M != w->M || d->K != w->K || d->N != w->N) return T3D_ERR_SHAPE;
  if (d->dtype != w->dy.dtype || w->a.dtype != d->dtype) return T3D_ERR_ARG;      // one element type per layer
  const bool bf16 = d->dtype == T3D_BF16;
  hipStream_t s = static_cast<hipStream_t>(stream);
  {   // one-pass form: eligible layer and a row split that leaves enough workgroups (t3d_bwd_plan's does)
    const int rps = w->rows_per_split, tiles = d->M / 128;
    // the epilogue takes the producer's raw output from the input tile it already holds: prev_y must BE the input tensor
    const bool own_x = d->prev_y == nullptr || (d->prev_y == w->a.x && w->a.coff == 0 && w->a.ldx == d->K);
    const bool shape1 = bwd1_shape(d->M, d->K, d->N, d->dtype) && rps % 128 == 0 && d->M / rps >= (tiles < 256 ? tiles : 256) && own_x &&
                        bwd1_split_ok(d->M, rps, d->dtype) &&
                        (w->a.scale == nullptr) == (w->a.shift == nullptr);
    if (shape1 && !bf16 && w->a.ldx % 4 == 0 && w->a.coff % 4 == 0) {
      T3D_RIDERS_FIRST(r, stream);
      const dim3 grid1(d->M / rps);
#define T3D_BWD1F(K_, N_)                                                                        \
  do {                                                                                           \
    if (d->add_in) {                                                                             \
      auto kern = k_pointmlp_bwd1f<K_, N_, true>;                                                \
      allow_lds(reinterpret_cast<const void*>(kern), lds_bwd1f(K_, N_));                         \
      T3D_LAUNCH(kern, grid1, dim3(NT1), lds_bwd1f(K_, N_), s, *d, *w);                          \
    } else {                                                                                     \
      auto kern = k_pointmlp_bwd1f<K_, N_, false>;                                               \
      allow_lds(reinterpret_cast<const void*>(kern), lds_bwd1f(K_, N_));                         \
      T3D_LAUNCH(kern, grid1, dim3(NT1), lds_bwd1f(K_, N_), s, *d, *w);                          \
    }                                                                                            \
  } while (0)
      if (d->K == 128 && d->N == 128) T3D_BWD1F(128, 128);
      else if (d->K == 128) T3D_BWD1F(128, 64);
      else if (d->N == 128) T3D_BWD1F(64, 128);
      else T3D_BWD1F(64, 64);
#undef T3D_BWD1F
      T3D_CHECK_LAUNCH();
      return T3D_OK;
    }
    if (shape1 && bf16 && w->a.dtype == T3D_BF16 && w->a.ldx % 8 == 0 && w->a.coff % 8 == 0) {
      T3D_RIDERS_FIRST(r, stream);
      const dim3 grid1(d->M / rps);
#define T3D_BWD1(K_, N_, BM_)                                                                    \
  do {                                                                                           \
    auto kern = k_pointmlp_bwd1<K_, N_, BM_>;                                                    \
    allow_lds(reinterpret_cast<const void*>(kern), lds_bwd1(K_, N_, BM_));                       \
    T3D_LAUNCH(kern, grid1, dim3(NT1), lds_bwd1(K_, N_, BM_), s, *d, *w);                        \
  } while (0)
      if (d->K == 256) T3D_BWD1(256, 128, 64);
      else if (d->N == 256) T3D_BWD1(128, 256, 64);
      else if (d->K == 128 && d->N == 128) T3D_BWD1(128, 128, 128);
      else if (d->K == 128) T3D_BWD1(128, 64, 128);
      else if (d->N == 128) T3D_BWD1(64, 128, 128);
      else T3D_BWD1(64, 64, 128);
#undef T3D_BWD1
      T3D_CHECK_LAUNCH();
      return T3D_OK;
    }
  }
  int tk = 0, tn = 0;
  wgrad_tile(w, &tk, &tn);
  const int n_w = ((w->K + tk - 1) / tk) * (w->N / tn) * (w->M / w->rows_per_split);
  const bool wide = dgrad_wide(d);
  const int n_d = (d->M / 128) * (d->K / (wide ? 128 : 64));
  if (bf16) { T3D_RIDERS_FIRST(r, stream); r = nullptr; }
  T3D_HOSTED(r, stream);      // (bf16 has left above: both fp32 forms below host the set)
  if (!bf16 && d->N % BKX == 0 && w->K % 64 == 0 && w->K % tk == 0 && w->a.dtype == T3D_F32 && x3_layer_bwd(d->arith, d->K, d->N)) return t3d_x3_bwd(d, w, r, tk, tn, wide, n_w, n_d, s);
  const dim3 grid(n_w + n_d + (r ? r->n_wg : 0));
  // interleaving the two kinds of tile by row range (so that both readers of a dy row range share an L2) measured SLOWER
  // than weight-gradient tiles first (1.683 vs 1.630 ms per step): the long wgrad tiles are better started early.
  static const bool il = []() { const char* e = getenv("T3D_BWD_INTERLEAVE"); return e && e[0] == '1'; }();
  const int interleave = (il && w->rows_per_split % 128 == 0) ? 1 : 0;
#define T3D_BWD(DBN, TK, TN_)                                                                                      \
  do {                                                                                                              \
    if (bf16) {                                                                                                     \
      const size_t lds = lds_dgrad_h(DBN) > lds_wgrad_h(TK, TN_) ? lds_dgrad_h(DBN) : lds_wgrad_h(TK, TN_);         \
      auto kern = k_pointmlp_bwd<DBN, TK, TN_, PathBF16>;                                                           \
      allow_lds(reinterpret_cast<const void*>(kern), lds);                                                          \
      T3D_LAUNCH(kern, grid, dim3(NT), lds, s, *d, *w, n_w, interleave);                                            \
    } else if (r) {                                                                                                 \
      const size_t lds = lds_with(lds_dgrad(DBN) > lds_wgrad(TK, TN_) ? lds_dgrad(DBN) : lds_wgrad(TK, TN_), r);    \
      auto kern = k_pointmlp_bwd_r<DBN, TK, TN_>;                                                                   \
      allow_lds(reinterpret_cast<const void*>(kern), lds);                                                          \
      T3D_LAUNCH(kern, grid, dim3(NT), lds, s, *d, *w, n_w, *r);                                                    \
    } else {                                                                                                        \
      const size_t lds = lds_dgrad(DBN) > lds_wgrad(TK, TN_) ? lds_dgrad(DBN) : lds_wgrad(TK, TN_);                 \
      auto kern = k_pointmlp_bwd<DBN, TK, TN_>;                                                                     \
      allow_lds(reinterpret_cast<const void*>(kern), lds);                                                          \
      T3D_LAUNCH(kern, grid, dim3(NT), lds, s, *d, *w, n_w, interleave);                                            \
    }                                                                                                               \
  } while (0)
#define T3D_BWD_W(DBN)                              \
  do {                                              \
    if (tk == 128 && tn == 128) T3D_BWD(DBN, 128, 128); \
    else if (tk == 128) T3D_BWD(DBN, 128, 64);      \
    else if (tn == 128) T3D_BWD(DBN, 64, 128);      \
    else T3D_BWD(DBN, 64, 64);                      \
  } while (0)
  if (wide) T3D_BWD_W(128);
  else T3D_BWD_W(64);
#undef T3D_BWD_W
#undef T3D_BWD
  T3D_CHECK_LAUNCH();
  return T3D_OK;
}

extern "C" int t3d_pool_bwd_stage1(const t3d_pointmlp_gram_args* g, const t3d_act_colsum_args* c,
                                   const t3d_pool_bwd_prep_args* q, t3d_stream_t stream) {
  return t3d_pool_bwd_stage1_r(g, c, q, nullptr, stream);
}

extern "C" int t3d_pool_bwd_stage1_r(const t3d_pointmlp_gram_args* g, const t3d_act_colsum_args* c,
                                     const t3d_pool_bwd_prep_args* q, const t3d_rider_set* r, t3d_stream_t stream) {
  T3D_ABI_TAKE(pointmlp_gram_args, g);
  if (r && !riders_ok(r)) return T3D_ERR_ARG;
  int rc = check_gram(g);
  if (rc != T3D_OK) return rc;
  if ((rc = check_colsum(c)) != T3D_OK) return rc;
  if ((rc = check_prep(q)) != T3D_OK) return rc;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (c->a.dtype != g->a.dtype) return T3D_ERR_ARG;
  if (gram1_ok(g) && c->M == g->M && c->K == g->K && c->a.x == g->a.x && c->a.ldx == g->a.ldx && c->a.coff == g->a.coff) {
    T3D_RIDERS_FIRST(r, stream);
    const int n_gram1 = g->M / g->rows_per_split, n_prep1 = (q->K / 32) * (q->K / 32) * ((q->N + PCH - 1) / PCH);
    const dim3 grid1(n_gram1 + (n_prep1 + 1) / 2);
#define T3D_ST1H(K_, BM_)                                                                                  \
  do {                                                                                                     \
    const size_t lds1 = lds_max(lds_gram1(K_, BM_), 2 * PREP_LDS);                                         \
    auto kern = k_pool_bwd_stage1_h<K_, BM_>;                                                              \
    allow_lds(reinterpret_cast<const void*>(kern), lds1);                                                  \
    T3D_LAUNCH(kern, grid1, dim3(NT1), lds1, s, *g, *c, *q, n_gram1);                                      \
  } while (0)
    if (g->K == 128) T3D_ST1H(128, 128); else T3D_ST1H(256, 64);
#undef T3D_ST1H
    T3D_CHECK_LAUNCH();
    return T3D_OK;
  }
  const bool x3g = g->a.dtype != T3D_BF16 && x3_layer_bwd(g->arith, g->K, g->K);
  const int gt = x3g ? 64 : gram_tile(g);      // (x3: 64 x 64 tiles, see mma_x3)
  const int n_gram = (g->K / gt) * (g->K / gt) * (g->M / g->rows_per_split);
  const int n_colsum = c->M / 128;
  const int n_prep = (q->K / 32) * (q->K / 32) * ((q->N + PCH - 1) / PCH);
  const bool bf16 = g->a.dtype == T3D_BF16;
  if (c->a.dtype != g->a.dtype) return T3D_ERR_ARG;
  if (bf16) { T3D_RIDERS_FIRST(r, stream); r = nullptr; }
  const dim3 grid(n_gram + n_colsum + n_prep + (r ? r->n_wg : 0));
  size_t lds = bf16 ? lds_wgrad_h(gt, gt) : lds_wgrad(gt, gt);
  if (PREP_LDS > lds) lds = PREP_LDS;
  if (COLSUM_LDS > lds) lds = COLSUM_LDS;
  T3D_HOSTED(r, stream);
  if (x3g) return t3d_x3_stage1(g, c, q, r, gt, n_gram, n_colsum, n_prep, lds_max(PREP_LDS, COLSUM_LDS), s);
  lds = lds_with(lds, r);
#define T3D_ST1(GT_, PR_)                                                              \
  do {                                                                                 \
    auto kern = k_pool_bwd_stage1<GT_, PR_>;                                           \
    allow_lds(reinterpret_cast<const void*>(kern), lds);                               \
    T3D_LAUNCH(kern, grid, dim3(NT), lds, s, *g, *c, *q, n_gram, n_colsum);            \
  } while (0)
#define T3D_ST1R(GT_)                                                                  \
  do {                                                                                 \
    auto kern = k_pool_bwd_stage1_r<GT_>;                                              \
    allow_lds(reinterpret_cast<const void*>(kern), lds);                               \
    T3D_LAUNCH(kern, grid, dim3(NT), lds, s, *g, *c, *q, n_gram, n_colsum, *r);        \
  } while (0)
  if (r) { if (gt == 128) T3D_ST1R(128); else T3D_ST1R(64); }
  else if (bf16) { if (gt == 128) T3D_ST1(128, PathBF16); else T3D_ST1(64, PathBF16); }
  else { if (gt == 128) T3D_ST1(128, PathF32); else T3D_ST1(64, PathF32); }
#undef T3D_ST1R
#undef T3D_ST1
  T3D_CHECK_LAUNCH();
  return T3D_OK;
}

extern "C" int t3d_pool_bwd_stage2(const t3d_pool_wgrad_finish_args* f, const t3d_pointmlp_dgrad_gram_args* d,
                                   t3d_stream_t stream) {
  return t3d_pool_bwd_stage2_r(f, d, nullptr, stream);
}

extern "C" int t3d_pool_bwd_stage2_r(const t3d_pool_wgrad_finish_args* f, const t3d_pointmlp_dgrad_gram_args* d,
                                     const t3d_rider_set* r, t3d_stream_t stream) {
  T3D_ABI_TAKE(pointmlp_dgrad_gram_args, d);
  if (r && !riders_ok(r)) return T3D_ERR_ARG;
  int rc = check_finish(f);
  if (rc != T3D_OK) return rc;
  if ((rc = check_dgrad_gram(d)) != T3D_OK) return rc;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int n_finish = (f->K / FK) * (f->N / FN);
  if (f->a.dtype != d->a.dtype) return T3D_ERR_ARG;
  if (dgram1_ok(d) && f->K == d->K) {
    T3D_RIDERS_FIRST(r, stream);
    const int n_fin2 = (n_finish + 1) / 2, tiles = d->M / 128;
    const dim3 grid2(n_fin2 + (tiles < 256 ? tiles : 256));
    const int fin_floats = (int)(finish_lds(f->K) / sizeof(float));
#define T3D_ST2H(K_, BM_)                                                                                  \
  do {                                                                                                     \
    const size_t lds2 = lds_max(lds_dgram1(K_, BM_), 2 * finish_lds(K_));                                  \
    auto kern = k_pool_bwd_stage2_h<K_, BM_>;                                                              \
    allow_lds(reinterpret_cast<const void*>(kern), lds2);                                                  \
    T3D_LAUNCH(kern, grid2, dim3(NT1), lds2, s, *f, *d, n_fin2, fin_floats);                               \
  } while (0)
    if (d->K == 128) T3D_ST2H(128, 128); else T3D_ST2H(256, 64);
#undef T3D_ST2H
    T3D_CHECK_LAUNCH();
    return T3D_OK;
  }
  const bool wide = dgrad_gram_wide(d);
  const int n_d = (d->M / 128) * (d->K / (wide ? 128 : 64));
  const bool bf16 = d->dtype == T3D_BF16;
  if (f->a.dtype != d->a.dtype) return T3D_ERR_ARG;
  if (bf16) { T3D_RIDERS_FIRST(r, stream); r = nullptr; }
  const dim3 grid(n_finish + n_d + (r ? r->n_wg : 0));
  size_t lds = bf16 ? (wide ? lds_dgram_h(128) : lds_dgram_h(64)) : (wide ? lds_fwd(128) : lds_fwd(64));
  if (finish_lds(f->K) > lds) lds = finish_lds(f->K);
  T3D_HOSTED(r, stream);
  if (!bf16 && d->K % BKX == 0 && x3_layer_bwd(d->arith, d->K, d->K)) return t3d_x3_stage2(f, d, r, wide, n_finish, n_d, finish_lds(f->K), s);
  lds = lds_with(lds, r);
#define T3D_ST2(BN_, PR_)                                                              \
  do {                                                                                 \
    auto kern = k_pool_bwd_stage2<BN_, PR_>;                                           \
    allow_lds(reinterpret_cast<const void*>(kern), lds);                               \
    T3D_LAUNCH(kern, grid, dim3(NT), lds, s, *f, *d, n_finish);                        \
  } while (0)
#define T3D_ST2R(BN_)                                                                  \
  do {                                                                                 \
    auto kern = k_pool_bwd_stage2_r<BN_>;                                              \
    allow_lds(reinterpret_cast<const void*>(kern), lds);                               \
    T3D_LAUNCH(kern, grid, dim3(NT), lds, s, *f, *d, n_finish, *r);                    \
  } while (0)
  if (r) { if (wide) T3D_ST2R(128); else T3D_ST2R(64); }
  else if (bf16) { if (wide) T3D_ST2(128, PathBF16); else T3D_ST2(64, PathBF16); }
  else { if (wide) T3D_ST2(128, PathF32); else T3D_ST2(64, PathF32); }
#undef T3D_ST2R
#undef T3D_ST2
  T3D_CHECK_LAUNCH();
  return T3D_OK;
}

extern "C" int t3d_split_x3(const float* src, void* planes, int64_t n, int64_t plane_stride, t3d_stream_t stream) {
  if (!src || !planes || n <= 0 || plane_stride < n || (reinterpret_cast<uintptr_t>(src) & 15) || (reinterpret_cast<uintptr_t>(planes) & 7) ||
      (plane_stride & 3))
    return T3D_ERR_ARG;
  return t3d_x3_split(src, planes, n, plane_stride, static_cast<hipStream_t>(stream));
}

extern "C" int t3d_split_x3_frag(const float* params, void* planes_fwd, void* planes_dgrad, int64_t plane_stride, const t3d_x3_frag_entry* table,
                                 int n_entries, int n_blocks, t3d_stream_t stream) {
  if (!params || !planes_fwd || !planes_dgrad || !table || n_entries <= 0 || n_blocks <= 0 || plane_stride <= 0 || (plane_stride & 7) ||
      (reinterpret_cast<uintptr_t>(params) & 15) || (reinterpret_cast<uintptr_t>(planes_fwd) & 15) || (reinterpret_cast<uintptr_t>(planes_dgrad) & 15))
    return T3D_ERR_ARG;
  return t3d_x3_split_frag(params, planes_fwd, planes_dgrad, plane_stride, table, n_entries, n_blocks, static_cast<hipStream_t>(stream));
}

// ---- does the `_r` launcher host a rider set for these arguments? (1 / 0; negative: the arguments are rejected) ----
static const t3d_rider_set* query_set() {
  static const t3d_rider_set q = []() { t3d_rider_set r{}; r.n_ops = 1; r.n_wg = 1; r.sync = reinterpret_cast<unsigned*>(16); return r; }();
  return &q;
}
extern "C" int t3d_gemm_arithmetic(int arith, int dtype, int K, int N, int kind) {
  if (dtype == T3D_BF16) return T3D_ARITH_BF16;
  bool x3 = false;
  switch (kind) {      // each launcher's own shape rule (the element-type / pointer conditions of a launch are the caller's to know)
    case T3D_GEMM_FWD: x3 = K % BKX == 0 && x3_layer(arith, K, N); break;                                                   // t3d_pointmlp_fwd[_r]
    case T3D_GEMM_BWD: x3 = N % BKX == 0 && K % 64 == 0 && (K <= 64 || K % 128 == 0) && x3_layer_bwd(arith, K, N); break;    // t3d_pointmlp_bwd[_r]: both tile kinds
    case T3D_GEMM_DGRAD: x3 = N % BKX == 0 && x3_layer_bwd(arith, K, N); break;                                             // t3d_pointmlp_dgrad (dense dy)
    case T3D_GEMM_WGRAD: x3 = K % 64 == 0 && (K <= 64 || K % 128 == 0) && x3_layer_bwd(arith, K, N); break;                  // t3d_pointmlp_wgrad[_r]: whole k-tiles of an fp32 source
    case T3D_GEMM_GRAM: x3 = x3_layer_bwd(arith, K, K); break;                                                              // t3d_pointmlp_gram, t3d_pool_bwd_stage1 (N ignored)
    case T3D_GEMM_DGRAD_GRAM: x3 = K % BKX == 0 && x3_layer_bwd(arith, K, K); break;                                        // t3d_pointmlp_dgrad_gram, t3d_pool_bwd_stage2
    default: return T3D_ERR_ARG;
  }
  return x3 ? T3D_ARITH_BF16X3 : T3D_ARITH_FP32_MFMA;
}
extern "C" int t3d_pointmlp_fwd_hosts_riders(const t3d_pointmlp_fwd_args* a) { return t3d_pointmlp_fwd_r(a, query_set(), T3D_QUERY_STREAM); }
extern "C" int t3d_pointmlp_wgrad_hosts_riders(const t3d_pointmlp_wgrad_args* a) { return t3d_pointmlp_wgrad_r(a, query_set(), T3D_QUERY_STREAM); }
extern "C" int t3d_pointmlp_bwd_hosts_riders(const t3d_pointmlp_dgrad_args* d, const t3d_pointmlp_wgrad_args* w) {
  return t3d_pointmlp_bwd_r(d, w, query_set(), T3D_QUERY_STREAM);
}
extern "C" int t3d_pool_bwd_stage1_hosts_riders(const t3d_pointmlp_gram_args* g, const t3d_act_colsum_args* c, const t3d_pool_bwd_prep_args* q) {
  return t3d_pool_bwd_stage1_r(g, c, q, query_set(), T3D_QUERY_STREAM);
}
extern "C" int t3d_pool_bwd_stage2_hosts_riders(const t3d_pool_wgrad_finish_args* f, const t3d_pointmlp_dgrad_gram_args* d) {
  return t3d_pool_bwd_stage2_r(f, d, query_set(), T3D_QUERY_STREAM);
}

#ifdef T3D_TRACE
// diagnostic builds only (not part of include/t3d.h): install / remove the per-workgroup trace buffer (each translation unit has its
// own copy of the pointer: the x3 kernels read the one of csrc/pointmlp_x3.hip)
int t3d_x3_set_trace(void* buf);
extern "C" int t3d_set_trace(void* buf) {
  unsigned long long* p = static_cast<unsigned long long*>(buf);
  if (t3d_x3_set_trace(buf) != T3D_OK) return T3D_ERR_LAUNCH;
  return hipMemcpyToSymbol(HIP_SYMBOL(t3d_trace_ptr), &p, sizeof(p)) == hipSuccess ? T3D_OK : T3D_ERR_LAUNCH;
}
#endif

#endif  // !T3D_X3_TU
#if defined(T3D_X3_TU) && defined(T3D_TRACE)
int t3d_x3_set_trace(void* buf) {
  unsigned long long* p = static_cast<unsigned long long*>(buf);
  return hipMemcpyToSymbol(HIP_SYMBOL(t3d_trace_ptr), &p, sizeof(p)) == hipSuccess ? T3D_OK : T3D_ERR_LAUNCH;
}
#endif
