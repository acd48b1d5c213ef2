// Stand-alone launches of the Gram-form helpers (t3d.h K11e); the bodies live in poolbwd_dev.h.
#include "poolbwd_dev.h"

namespace {

__global__ __launch_bounds__(256) void k_pool_bwd_prep(const t3d_pool_bwd_prep_args p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  pool_bwd_prep_body(p, smem, blockIdx.x, blockIdx.y, blockIdx.z);
}
__global__ __launch_bounds__(256) void k_pool_sparse_rows(const t3d_pool_sparse_rows_args p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  pool_sparse_rows_body(p, smem, blockIdx.x, blockIdx.y);
}
template <class XT>
__global__ __launch_bounds__(256) void k_act_colsum(const t3d_act_colsum_args p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  act_colsum_body<XT>(p, smem, blockIdx.x);
}
template <class XT>
__global__ __launch_bounds__(256) void k_pool_wgrad_finish(const t3d_pool_wgrad_finish_args p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  pool_wgrad_finish_body<XT>(p, smem, blockIdx.x, blockIdx.y);
}

}  // namespace

extern "C" int t3d_pool_bwd_prep(const t3d_pool_bwd_prep_args* a, t3d_stream_t stream) {
  const int rc = check_prep(a);
  if (rc != T3D_OK) return rc;
  T3D_LAUNCH(k_pool_bwd_prep, dim3(a->K / 32, a->K / 32, (a->N + PCH - 1) / PCH), dim3(256), PREP_LDS,
             static_cast<hipStream_t>(stream), *a);
  T3D_CHECK_LAUNCH();
  return T3D_OK;
}

extern "C" int t3d_pool_sparse_rows(const t3d_pool_sparse_rows_args* a, t3d_stream_t stream) {
  const int rc = check_sparse_rows(a);
  if (rc != T3D_OK) return rc;
  const size_t lds = sparse_rows_lds(a->N);
  static size_t allowed = 0;          // set the > 64 KB attribute once per size, not per launch
  if (lds > 64 * 1024 && lds > allowed) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_pool_sparse_rows), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)lds);
    allowed = lds;
  }
  const long M = (long)a->B * a->rows_per_frustum;
  T3D_LAUNCH(k_pool_sparse_rows, dim3(M / 128, a->K / SR_KC), dim3(256), lds, static_cast<hipStream_t>(stream), *a);
  T3D_CHECK_LAUNCH();
  return T3D_OK;
}

extern "C" int t3d_act_colsum(const t3d_act_colsum_args* a, t3d_stream_t stream) {
  const int rc = check_colsum(a);
  if (rc != T3D_OK) return rc;
  if (a->a.dtype == T3D_BF16) T3D_LAUNCH(k_act_colsum<bf16_t>, dim3(a->M / 128), dim3(256), COLSUM_LDS, static_cast<hipStream_t>(stream), *a);
  else T3D_LAUNCH(k_act_colsum<float>, dim3(a->M / 128), dim3(256), COLSUM_LDS, static_cast<hipStream_t>(stream), *a);
  T3D_CHECK_LAUNCH();
  return T3D_OK;
}

extern "C" int t3d_pool_wgrad_finish(const t3d_pool_wgrad_finish_args* a, t3d_stream_t stream) {
  const int rc = check_finish(a);
  if (rc != T3D_OK) return rc;
  const size_t lds = finish_lds(a->K);
  if (a->a.dtype == T3D_BF16) T3D_LAUNCH(k_pool_wgrad_finish<bf16_t>, dim3(a->K / FK, a->N / FN), dim3(256), lds, static_cast<hipStream_t>(stream), *a);
  else T3D_LAUNCH(k_pool_wgrad_finish<float>, dim3(a->K / FK, a->N / FN), dim3(256), lds, static_cast<hipStream_t>(stream), *a);
  T3D_CHECK_LAUNCH();
  return T3D_OK;
}
