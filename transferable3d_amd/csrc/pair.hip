// Two INDEPENDENT small launches in one: the backward of the box / T-Net side and the backward of the segmentation net do not depend
// on each other (semisup_models.py:150-151 blocks the gradient), and each carries a dozen launches of 16-64 workgroups -- batch-norm
// backward finalizers, the FC head kernels, per-frustum column sums -- whose cost is the ~3-5 us coherence round trip of a kernel
// boundary, not their work.  One launch that runs the blocks of one op of each chain pays that once (tools/micro/grid_barrier.py:
// 2-4 us saved per pair).  The bodies are the stand-alone kernels' own (fc_dev.h, bn_dev.h): results are bit-identical.
// The host side (nets.pair_small_launches) interleaves the two chains so that their small launches meet.
#include "rider_dev.h"

namespace {

template <int RBT>
__device__ __forceinline__ void run_small(const t3d_small_op& o, float* sm, const int bid) {
  switch (o.kind) {      // workgroup-uniform
    case T3D_SMALL_BN_BWD_FINALIZE: bn_bwd_finalize_body<FC_GR>(o.u.bn_bwd, reinterpret_cast<double (*)[FC_CH]>(sm), bid, threadIdx.x); break;
    case T3D_SMALL_FC_BWD: fc_bwd_body<RBT>(o.u.fc_bwd, sm, bid); break;
    case T3D_SMALL_FC_DINPUT: fc_dinput_body<RBT>(o.u.fc_dinput, sm, bid); break;
    case T3D_SMALL_DY_COLSUM: dy_colsum_body(o.u.dy_colsum, bid, threadIdx.x); break;
    default: break;
  }
}

template <int RBT>
__global__ __launch_bounds__(NTH) void k_small_pair(const t3d_small_op a, const t3d_small_op b, const int n_a) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  if ((int)blockIdx.x < n_a) run_small<RBT>(a, sm, blockIdx.x);
  else run_small<RBT>(b, sm, blockIdx.x - n_a);
}

int small_blocks(const t3d_small_op& o, int* rows) {
  switch (o.kind) {
    case T3D_SMALL_BN_BWD_FINALIZE:
      if (!o.u.bn_bwd.coef || (o.u.bn_bwd.psum_dz != nullptr && o.u.bn_bwd.n_tiles > 512)) return -1;      // (the 64-group form stays alone)
      return (o.u.bn_bwd.N + FC_CH - 1) / FC_CH;
    case T3D_SMALL_FC_BWD: *rows = o.u.fc_bwd.B; return (o.u.fc_bwd.N + CB - 1) / CB;
    case T3D_SMALL_FC_DINPUT: *rows = o.u.fc_dinput.B; return (o.u.fc_dinput.K + CB - 1) / CB;
    case T3D_SMALL_DY_COLSUM: return (o.u.dy_colsum.B * o.u.dy_colsum.N + 255) / 256;
    default: return -1;
  }
}

// a rider set as a launch of its own (what a `_r` launcher falls back to, and the scheduler's choice for a run of small ops that found no
// GEMM to ride in: one launch instead of one per op)
__global__ __launch_bounds__(RIDER_NT) void k_riders(const t3d_rider_set r) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  run_riders(r, sm);
}

}  // namespace

extern "C" int t3d_riders_plan(t3d_rider_set* r) {
  if (!r || r->n_ops <= 0 || r->n_ops > T3D_RIDER_MAX_OPS) return T3D_ERR_ARG;
  int m = 1;
  size_t lds = 0;
  for (int i = 0; i < r->n_ops; ++i) {
    const int nb = rider_op_blocks(r->ops[i]);
    if (nb <= 0) return r->ops[i].kind >= 1 && r->ops[i].kind <= 7 ? T3D_ERR_SHAPE : T3D_ERR_ARG;
    if (r->ops[i].kind == T3D_SMALL_POOL_BWD_MID && r->n_ops != 1) return T3D_ERR_ARG;      // a wide rider is alone in its set
    if (nb > m) m = nb;
    const size_t l = rider_op_lds(r->ops[i]);
    if (l > lds) lds = l;
  }
  const int cap = r->n_ops == 1 ? RIDER_MAX_WG_WIDE : RIDER_MAX_WG;      // no barrier in a one-op set: no residency requirement
  r->n_wg = m < cap ? m : cap;
  if (r->n_ops == 1 && r->ops[0].kind != T3D_SMALL_POOL_BWD_MID && r->n_wg > RIDER_MAX_WG) r->n_wg = RIDER_MAX_WG;
  r->lds_bytes = (int)lds;
  return T3D_OK;
}

extern "C" int t3d_run_riders(const t3d_rider_set* r, t3d_stream_t stream) {
  if (!r || !r->sync || r->n_ops <= 0 || r->n_ops > T3D_RIDER_MAX_OPS || r->n_wg <= 0 || r->n_wg > (r->n_ops == 1 ? RIDER_MAX_WG_WIDE : RIDER_MAX_WG) ||
      r->lds_bytes < 0)
    return T3D_ERR_ARG;
  if (r->lds_bytes > 64 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_riders), hipFuncAttributeMaxDynamicSharedMemorySize, r->lds_bytes);
  T3D_LAUNCH(k_riders, dim3(r->n_wg), dim3(RIDER_NT), (size_t)r->lds_bytes, static_cast<hipStream_t>(stream), *r);
  T3D_CHECK_LAUNCH();
  return T3D_OK;
}

extern "C" int t3d_small_pair(const t3d_small_op* a, const t3d_small_op* b, t3d_stream_t stream) {
  if (!a || !b) return T3D_ERR_ARG;
  int rows_a = 0, rows_b = 0;
  const int na = small_blocks(*a, &rows_a), nb = small_blocks(*b, &rows_b);
  if (na <= 0 || nb <= 0) return T3D_ERR_ARG;
  const int rows = rows_a > rows_b ? rows_a : rows_b;
  if (rows > 32 * MAXRB || (rows_a && rows_b && (rows_a <= 32) != (rows_b <= 32))) return T3D_ERR_SHAPE;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (rows <= 32) T3D_LAUNCH(k_small_pair<1>, dim3(na + nb), dim3(NTH), fc_lds_bytes(32), s, *a, *b, na);
  else T3D_LAUNCH(k_small_pair<MAXRB>, dim3(na + nb), dim3(NTH), fc_lds_bytes(128), s, *a, *b, na);
  T3D_CHECK_LAUNCH();
  return T3D_OK;
}
