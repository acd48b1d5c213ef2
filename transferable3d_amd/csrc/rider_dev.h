// Riders: small launches of one dependency chain executed by a few extra workgroups of a GEMM launch of an INDEPENDENT chain.
//
// At B = 32 a quarter of the training step is spent in ~60 launches of 1-64 workgroups (batch-norm finalizers, the FC heads of the
// T-Net / box net and their backward, per-frustum column sums) whose cost is the dependent memory round trips of a kernel boundary,
// not their work, while 240 of 256 CUs idle.  The segmentation net's backward needs nothing from the T-Net / box net forward,
// loss and backward (semisup_models.py:150-151: the mask is a hard comparison, no gradient crosses it), so the host (schedule.py)
// interleaves the two chains and hands the small launches of one to the GEMM launches of the other: the first `n_wg` workgroups
// of such a launch run a rider SET -- a run of small ops in chain order -- and leave; the GEMM's own tiles follow behind them.
// Ops of a set that depend on their predecessor (`depends`) are separated by a barrier among the rider workgroups only
// (agent-scope release / acquire, cdna_hip_programming.md Guideline 16); the GEMM's workgroups never wait for anything.
// All rider workgroups are the launch's lowest block indices and far fewer than the chip's resident slots (<= 32 of 512), so they
// are co-resident by the time any of them waits.
//
// The bodies are the stand-alone kernels' own (fc_dev.h, bn_dev.h) run by 256 threads: results are bit-identical to the separate
// launches (tests/test_riders_gpu.py).
#pragma once
#include "fc_dev.h"
#include "bn_dev.h"
#include "poolbwd_dev.h"

namespace {

constexpr int RIDER_NWP = 4;                  // physical waves of a rider workgroup (= the GEMM kernels' 256 threads)
constexpr int RIDER_NT = RIDER_NWP * 64;
constexpr int RIDER_MAX_WG = 32;                // sets with barriers: every rider workgroup must be resident
constexpr int RIDER_MAX_WG_WIDE = 2048;         // a one-op set (no barrier): t3d_pool_bwd_mid rides with one workgroup per block
constexpr unsigned RIDER_SPIN_LIMIT = 1u << 22;

typedef __attribute__((address_space(1))) unsigned rider_gu32;

// Barrier i of a set (in front of op i): sync[2i] arrival count, sync[2i+1] generation; sync[2 * T3D_RIDER_MAX_OPS] is the set's timeout word.
// Self-resetting: the last arriver zeroes the count, then moves the generation; the others wait for the generation to leave the value
// they read BEFORE arriving (it cannot move until they have arrived).  Replays of a captured launch need no memset node.
__device__ __forceinline__ void rider_barrier(unsigned* sync, int i, int n_wg, int n_ops) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // every storing wave drains its stores
  __syncthreads();
  if (threadIdx.x == 0) {
    rider_gu32* cnt = (rider_gu32*)(sync + 2 * i);
    rider_gu32* gen = cnt + 1;
    const unsigned g0 = __hip_atomic_load(gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // the write-back completes before the arrival is visible
    const unsigned old = __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (old + 1u == (unsigned)n_wg) {
      __hip_atomic_store(cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __hip_atomic_fetch_add(gen, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      unsigned spins = 0;
      while (__hip_atomic_load(gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == g0) {
        __builtin_amdgcn_s_sleep(8);
        if (++spins > RIDER_SPIN_LIMIT) {                    // bounded: a poisoned state ends in a flagged wrong answer, not a hang
          __hip_atomic_store((rider_gu32*)(sync + 2 * T3D_RIDER_MAX_OPS), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          break;
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // holds the barrier below until the invalidate has completed
  }
  __syncthreads();
}

// t3d_pool_bwd_mid's grid (bn_optim.hip): gx x n_tensors slab-reduction blocks, then the sparse-row tiles
__host__ __device__ __forceinline__ int mid_gx(int max_numel) {
  int gx = (max_numel / 4 + 31) / 32;
  return gx > 256 ? 256 : (gx < 1 ? 1 : gx);
}
__host__ __device__ __forceinline__ int mid_sparse_blocks(const t3d_pool_sparse_rows_args& sp) {
  return (int)(((long)sp.B * sp.rows_per_frustum) / 128) * (sp.K / SR_KC);
}

// workgroups a rider op is cut into (the stand-alone launchers' grids)
__device__ __forceinline__ int rider_blocks(const t3d_small_op& o, int kind) {
  switch (kind) {
    case T3D_SMALL_BN_BWD_FINALIZE: return (o.u.bn_bwd.N + FC_CH - 1) / FC_CH;
    case T3D_SMALL_BN_FWD_FINALIZE: return (o.u.bn_fwd.N + FC_CH - 1) / FC_CH;
    case T3D_SMALL_FC_FWD: return (o.u.fc_fwd.N + CB - 1) / CB;
    case T3D_SMALL_FC_BWD: return (o.u.fc_bwd.N + CB - 1) / CB;
    case T3D_SMALL_FC_DINPUT: return (o.u.fc_dinput.K + CB - 1) / CB;
    case T3D_SMALL_DY_COLSUM: return (o.u.dy_colsum.B * o.u.dy_colsum.N + 255) / 256;
    case T3D_SMALL_POOL_BWD_MID: return mid_gx(o.u.mid.max_numel) * o.u.mid.n_tensors + mid_sparse_blocks(o.u.mid.sparse);
    default: return 0;
  }
}

// `r` is the kernel's by-value argument: the op structs sit in the kernarg segment like the arguments of the stand-alone kernels and
// are read field by field where needed (scalar loads).  Copies of the structs fetched from a device table instead cost 186 SGPR spills
// and pushed every host kernel to the 256-VGPR cap with scratch; one __noinline__ function per kind cannot carry an occupancy bound
// and took up to 280 registers.  As written: 248 VGPRs, no spill, no scratch in every host kernel (two waves per SIMD, as without riders).
__device__ __forceinline__ void run_riders(const t3d_rider_set& r, float* smem) {
  const int wg = blockIdx.x;
  double (*red)[FC_CH] = reinterpret_cast<double (*)[FC_CH]>(smem);
  float* scsh = smem + 2 * FC_GR * FC_CH;                  // behind the FC_GR x FC_CH doubles
  __builtin_amdgcn_s_setprio(3);                           // a latency chain beside a throughput kernel: its few instructions go first
  for (int i = 0; i < r.n_ops; ++i) {
    const t3d_small_op& o = r.ops[i];
    const int kind = o.kind;
    if (i > 0) {
      if (o.depends) rider_barrier(r.sync, i, r.n_wg, r.n_ops);
      else __syncthreads();
    }
    const int nb = rider_blocks(o, kind);
    for (int b = wg; b < nb; b += r.n_wg) {
      switch (kind) {      // workgroup-uniform
        case T3D_SMALL_BN_BWD_FINALIZE: bn_bwd_finalize_body<FC_GR>(o.u.bn_bwd, red, b, threadIdx.x); break;
        case T3D_SMALL_BN_FWD_FINALIZE: bn_fwd_finalize_body<FC_GR>(o.u.bn_fwd, red, scsh, b); break;
        case T3D_SMALL_FC_FWD: fc_fwd_body<1, RIDER_NWP>(o.u.fc_fwd, smem, b); break;
        case T3D_SMALL_FC_BWD: fc_bwd_body<1, RIDER_NWP>(o.u.fc_bwd, smem, b); break;
        case T3D_SMALL_FC_DINPUT: fc_dinput_body<1, RIDER_NWP>(o.u.fc_dinput, smem, b); break;
        case T3D_SMALL_DY_COLSUM: dy_colsum_body(o.u.dy_colsum, b, threadIdx.x); break;
        case T3D_SMALL_POOL_BWD_MID: {
          const t3d_pool_bwd_mid_args& m = o.u.mid;
          const int gx = mid_gx(m.max_numel), n_reduce = gx * m.n_tensors;
          if (b < n_reduce) {
            reduce_slabs_body(m.slab_base, m.grad_base, m.table_dev, reinterpret_cast<float4(*)[32]>(smem), b % gx, b / gx, gx);
          } else {
            const int rr = b - n_reduce, tiles = m.sparse.B * m.sparse.rows_per_frustum / 128;
            pool_sparse_rows_body(m.sparse, smem, rr % tiles, rr / tiles);
          }
        } break;
        default: break;
      }
      __syncthreads();                                      // the next block of this workgroup reuses the LDS scratch
    }
  }
}

// host side: can this op ride, how many workgroups does it want, how much LDS
inline int rider_op_blocks(const t3d_small_op& o) {
  switch (o.kind) {
    case T3D_SMALL_BN_BWD_FINALIZE:
      if (!o.u.bn_bwd.coef || (o.u.bn_bwd.psum_dz != nullptr && o.u.bn_bwd.n_tiles > 512)) return -1;      // the many-tile forms stay alone
      return (o.u.bn_bwd.N + FC_CH - 1) / FC_CH;
    case T3D_SMALL_BN_FWD_FINALIZE:
      if (o.u.bn_fwd.n_tiles > 512) return -1;
      return (o.u.bn_fwd.N + FC_CH - 1) / FC_CH;
    case T3D_SMALL_FC_FWD: return o.u.fc_fwd.B <= 32 ? (o.u.fc_fwd.N + CB - 1) / CB : -1;
    case T3D_SMALL_FC_BWD: return o.u.fc_bwd.B <= 32 ? (o.u.fc_bwd.N + CB - 1) / CB : -1;
    case T3D_SMALL_FC_DINPUT: return o.u.fc_dinput.B <= 32 ? (o.u.fc_dinput.K + CB - 1) / CB : -1;
    case T3D_SMALL_DY_COLSUM: return (o.u.dy_colsum.B * o.u.dy_colsum.N + 255) / 256;
    case T3D_SMALL_POOL_BWD_MID: {
      const t3d_pool_bwd_mid_args& m = o.u.mid;
      if (!m.slab_base || !m.grad_base || !m.table_dev || m.n_tensors <= 0 || check_sparse_rows(&m.sparse) != T3D_OK) return -1;
      // its LDS (a 128 x 128 fp32 tile + the hit lists: 64 KB + 16 N bytes) becomes the whole launch's: above 76 KB the host GEMM
      // would drop to one workgroup per CU (N = 1024, the seg net's conv5: stays alone)
      if (sparse_rows_lds(m.sparse.N) > 76 * 1024) return -1;
      return mid_gx(m.max_numel) * m.n_tensors + mid_sparse_blocks(m.sparse);
    }
    default: return -1;
  }
}

inline size_t rider_op_lds(const t3d_small_op& o) {
  if (o.kind == T3D_SMALL_POOL_BWD_MID) return sparse_rows_lds(o.u.mid.sparse.N);
  return fc_lds_bytes(32);      // the FC bodies' reduction tiles are the largest scratch of the other kinds
}

}  // namespace
