// Per-frustum fully-connected layers (B rows): forward, backward, input gradient -- fp32 MFMA.
//
// Replaces tf_util.fully_connected (models/tf_util.py:1463-1499) with its batch-norm over the B rows
// (tf_util.py:1666-1677), activation and the tf_util.dropout that follows it (tf_util.py:1720-1741) at
// semisup_models.py:196-198, 253-261, 385-392 and mlps_with_dropout (44-63).
//
// These layers are latency bound (M = B = 32..128 rows, weights <= 2 MB).  One workgroup owns 32 output
// columns for ALL rows, so the batch-norm reductions over the batch stay inside the workgroup and a layer is a
// single launch.  The reduction dimension is split over the workgroup's 8 waves (interleaved 8-deep k groups);
// every wave feeds v_mfma_f32_32x32x2_f32 straight from global/L2 (operands are tiny and cache resident, no LDS
// staging), the partial 32x32 tiles are summed through LDS, and the epilogue runs on all 512 threads.
#include "fc_dev.h"

namespace {

template <int RBT>
__global__ __launch_bounds__(NTH) void k_fc_fwd(const t3d_fc_fwd_args p, const int cb) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  fc_fwd_body<RBT>(p, sm, blockIdx.x, cb);
}
template <int RBT>
__global__ __launch_bounds__(NTH) void k_fc_bwd(const t3d_fc_bwd_args p) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  fc_bwd_body<RBT>(p, sm, blockIdx.x);
}
template <int RBT>
__global__ __launch_bounds__(NTH) void k_fc_dinput(const t3d_fc_dinput_args p) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  fc_dinput_body<RBT>(p, sm, blockIdx.x);
}

}  // namespace


extern "C" int t3d_fc_fwd(const t3d_fc_fwd_args* a, t3d_stream_t stream) {
  if (!a || !a->in || !a->out || (a->K2 > 0 && !a->in2)) return T3D_ERR_ARG;
  if (!a->w && (a->K != a->N || a->K2 != 0)) return T3D_ERR_SHAPE;      // identity form
  if (a->gamma && (!a->beta || !a->moving_mean || !a->moving_var || !a->mean || !a->invstd || !a->y)) return T3D_ERR_ARG;
  if (a->gamma && a->is_training && !a->decay) return T3D_ERR_ARG;
  if (a->B <= 0 || a->B > 32 * MAXRB || a->N <= 0 || a->K <= 0) return T3D_ERR_SHAPE;
  // columns per workgroup: 32.  T3D_FC_CB=16 / 8 cuts a wide layer with a long reduction into narrower column runs (the
  // 1024 -> 512 row-bias layer of conv6 then runs on 32 / 64 CUs instead of 16) -- measured on MI355X: 12.4 / 12.3 / 12.6 us per launch
  // at 32 / 16 / 8 columns, step 1.4489 / 1.4480 / 1.4476 ms: the launch is a latency chain, not a bandwidth problem per CU; off.
  static const int narrow = []() { const char* e = getenv("T3D_FC_CB"); const int v = e ? atoi(e) : 32; return (v == 8 || v == 16 || v == 32) ? v : 32; }();
  const int cb = (a->w && (long)a->K * a->N >= (1L << 18) && a->N >= 8 * narrow) ? narrow : CB;
  if (a->B <= 32) T3D_LAUNCH(k_fc_fwd<1>, dim3((a->N + cb - 1) / cb), dim3(NTH), fc_lds_bytes(32), static_cast<hipStream_t>(stream), *a, cb);
  else T3D_LAUNCH(k_fc_fwd<MAXRB>, dim3((a->N + cb - 1) / cb), dim3(NTH), fc_lds_bytes(128), static_cast<hipStream_t>(stream), *a, cb);
  T3D_CHECK_LAUNCH();
  return T3D_OK;
}

extern "C" int t3d_fc_bwd(const t3d_fc_bwd_args* a, t3d_stream_t stream) {
  if (!a || !a->dy || (!a->dout && (!a->dy_next || !a->w_next))) return T3D_ERR_ARG;
  if (a->dw && (!a->in || (a->K2 > 0 && !a->in2))) return T3D_ERR_ARG;
  if (a->gamma && (!a->beta || !a->mean || !a->invstd || !a->y)) return T3D_ERR_ARG;
  if (a->act != T3D_ACT_NONE && !a->y) return T3D_ERR_ARG;
  if (a->B <= 0 || a->B > 32 * MAXRB || a->N <= 0) return T3D_ERR_SHAPE;
  if (a->B <= 32) T3D_LAUNCH(k_fc_bwd<1>, dim3((a->N + CB - 1) / CB), dim3(NTH), fc_lds_bytes(32), static_cast<hipStream_t>(stream), *a);
  else T3D_LAUNCH(k_fc_bwd<MAXRB>, dim3((a->N + CB - 1) / CB), dim3(NTH), fc_lds_bytes(128), static_cast<hipStream_t>(stream), *a);
  T3D_CHECK_LAUNCH();
  return T3D_OK;
}

extern "C" int t3d_fc_dinput(const t3d_fc_dinput_args* a, t3d_stream_t stream) {
  if (!a || !a->dy || !a->w || !a->din) return T3D_ERR_ARG;
  if (a->bn_coef && (!a->bn_pooled || !a->bn_ysel || !a->bn_dpool || !a->bn_scale ||
                     (!a->bn_frozen && (!a->bn_gamma || !a->bn_mean || !a->bn_invstd))))
    return T3D_ERR_ARG;
  if (a->B <= 0 || a->B > 32 * MAXRB || a->N <= 0 || a->K <= 0) return T3D_ERR_SHAPE;
  if (a->B <= 32) T3D_LAUNCH(k_fc_dinput<1>, dim3((a->K + CB - 1) / CB), dim3(NTH), fc_lds_bytes(32), static_cast<hipStream_t>(stream), *a);
  else T3D_LAUNCH(k_fc_dinput<MAXRB>, dim3((a->K + CB - 1) / CB), dim3(NTH), fc_lds_bytes(128), static_cast<hipStream_t>(stream), *a);
  T3D_CHECK_LAUNCH();
  return T3D_OK;
}

#ifdef T3D_TRACE
extern "C" int t3d_set_trace_fc(void* buf) {
  unsigned long long* p = static_cast<unsigned long long*>(buf);
  return hipMemcpyToSymbol(HIP_SYMBOL(t3d_trace_fc_ptr), &p, sizeof(p)) == hipSuccess ? T3D_OK : T3D_ERR_LAUNCH;
}
#endif
