// Segmentation head (dropout + conv10 + softmax-CE + hard mask, forward AND backward in one pass) and the
// per-frustum box losses (forward + backward).
//
// seg head: semisup_models.py:131-136 (dropout, conv10), 150-158 (mask, masked xyz sums),
//           semisup_v1_sunrgbd.py:430-431 (sparse softmax CE).
// box loss: semisup_v1_sunrgbd.py:423-564, model_util.py:94-119,145-167, tf_util.py:1001-1041.
#include "common.h"
#include "boxgeom_dev.h"

namespace {

// ---------------------------------------------------------------------------------------------
// seg head.  One workgroup = one 128-row tile = 16 waves (4 per SIMD, so that the per-row chain load -> dot ->
// wave reduction -> exp/log -> dz store of one wave hides under the others); wave w owns rows 8w..8w+7, two rows
// in flight; lane owns channels 2*lane, 2*lane+1 of the K = 128 wide conv9 output (coalesced 512-B row loads).
// ---------------------------------------------------------------------------------------------
constexpr int SH_WAVES = 16, SH_ROWS = 128 / SH_WAVES, SH_LD = 128 * 4 + 8;

// Sum over the 64 lanes with data-parallel-primitive moves (six VALU adds) instead of six ds_bpermute round trips through the LDS
// crossbar; the total is read from lane 63 and returned in every lane.
template <int CTRL>
__device__ __forceinline__ float dpp_add(float v) {
  return v + __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float wave_sum_dpp(float v) {
  v = dpp_add<0xb1>(v);      // quad_perm [1,0,3,2]
  v = dpp_add<0x4e>(v);      // quad_perm [2,3,0,1]
  v = dpp_add<0x124>(v);     // row_ror:4
  v = dpp_add<0x128>(v);     // row_ror:8   -> every lane holds the sum of its row of 16
  v = dpp_add<0x142>(v);     // row_bcast:15 -> last lane of rows 1..3 adds the previous row's sum
  v = dpp_add<0x143>(v);     // row_bcast:31 -> lane 63 holds the wave's sum
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

__device__ __forceinline__ float drop_keep(uint64_t key, size_t i, float keep) {
  uint64_t x = key + (uint64_t)i * 0xD6E8FEB86659FD93ULL;
  x ^= x >> 33; x *= 0xff51afd7ed558ccdULL;
  x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL;
  x ^= x >> 33;
  const uint32_t r = (uint32_t)(x >> 16);
  return (float)(r >> 8) * (1.0f / 16777216.0f) < keep ? 1.f : 0.f;
}

// T: element type of conv9's raw output y and of dz (t3d_seg_head_args.dtype).  SOFT: the second run of the head behind the weak
// surface loss (t3d_weak_loss): d loss / d soft_mask joins the logit gradients (an instantiation of its own: the usual one is untouched)
template <class T, bool SOFT = false>
__global__ __launch_bounds__(64 * SH_WAVES) void k_seg_head(const t3d_seg_head_args p) {
  __shared__ float red[SH_WAVES][SH_LD];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int tile = blockIdx.x, row0 = tile * 128;
  const int b = row0 / p.rows_per_frustum;
  const int ch = 2 * lane;
  const float2 sc = *reinterpret_cast<const float2*>(p.scale + ch);
  const float2 sh = *reinterpret_cast<const float2*>(p.shift + ch);
  const float w00 = p.w[ch * 2 + 0], w01 = p.w[ch * 2 + 1], w10 = p.w[ch * 2 + 2], w11 = p.w[ch * 2 + 3];
  const float b0 = p.bias[0], b1 = p.bias[1];
  const bool train = p.labels != nullptr;
  const bool bwd = p.dz != nullptr;
  const float wb = train ? p.ce_weight * (float)(1 - p.is_data_2D[b]) / ((float)p.B * (float)p.rows_per_frustum) : 0.f;
  const bool gen = p.drop_mask == nullptr && p.drop_hyper != nullptr && p.keep_prob < 1.f;
  const float inv_keep = (p.drop_mask || gen) ? 1.0f / p.keep_prob : 1.f;
  const uint64_t dkey = gen ? (((uint64_t)p.drop_seed << 32) ^ ((uint64_t)p.drop_hyper[0] * 0x9E3779B97F4A7C15ULL)) : 0ull;

  float sdz0 = 0.f, sdz1 = 0.f, sdzy0 = 0.f, sdzy1 = 0.f, dw00 = 0.f, dw01 = 0.f, dw10 = 0.f, dw11 = 0.f;

  // Phase A -- the wave's SH_ROWS rows, channel-parallel: dropout(relu(bn(y))) and the two logits of every row (wave sums).
  const int rbase = row0 + wid * SH_ROWS;
  float2 y[SH_ROWS], km[SH_ROWS];
  float d0[SH_ROWS], d1[SH_ROWS], q0r[SH_ROWS], q1r[SH_ROWS];
#pragma unroll
  for (int i = 0; i < SH_ROWS; ++i) {
    const size_t o = (size_t)(rbase + i) * 128 + ch;
    y[i] = Elem<T>::ld2(p.y, o);
    if (gen) {                      // same generator and element index as k_dropout_mask (bn_optim.hip)
      km[i].x = drop_keep(dkey, o, p.keep_prob);
      km[i].y = drop_keep(dkey, o + 1, p.keep_prob);
    } else {
      km[i] = p.drop_mask ? *reinterpret_cast<const float2*>(p.drop_mask + o) : make_float2(1.f, 1.f);
    }
  }
#pragma unroll
  for (int i = 0; i < SH_ROWS; ++i) {
    km[i].x *= inv_keep; km[i].y *= inv_keep;
    d0[i] = fmaxf(fmaf(y[i].x, sc.x, sh.x), 0.f) * km[i].x;
    d1[i] = fmaxf(fmaf(y[i].y, sc.y, sh.y), 0.f) * km[i].y;
    q0r[i] = wave_sum_dpp(fmaf(d0[i], w00, d1[i] * w10)) + b0;
    q1r[i] = wave_sum_dpp(fmaf(d0[i], w01, d1[i] * w11)) + b1;
  }

  // Phase B -- row-parallel: lane r < SH_ROWS owns row r.  The soft-max / cross-entropy arithmetic (three transcendentals per row)
  // is issued once for the eight rows instead of once per row with all 64 lanes computing the same numbers.
  float q0 = q0r[0], q1 = q1r[0];
#pragma unroll
  for (int i = 1; i < SH_ROWS; ++i) { q0 = lane == i ? q0r[i] : q0; q1 = lane == i ? q1r[i] : q1; }
  const bool own = lane < SH_ROWS;
  const int row = rbase + (lane & (SH_ROWS - 1));
  const bool oracle = p.oracle_mask != nullptr;      // uniform: logits = stack([1 - m, m]) (semisup_v1_sunrgbd.py:161-162)
  if (oracle) { const float om = (float)p.oracle_mask[row]; q0 = 1.f - om; q1 = om; }
  const float m = q0 < q1 ? 1.f : 0.f;
  if (own) {
    *reinterpret_cast<float2*>(p.logits + (size_t)row * 2) = make_float2(q0, q1);
    p.mask[row] = m;
  }
  const float ownf = own ? 1.f : 0.f;
  const float px = p.pc[(size_t)row * p.ld_pc], py = p.pc[(size_t)row * p.ld_pc + 1], pz = p.pc[(size_t)row * p.ld_pc + 2];
  float cnt = ownf * m, sx = cnt * px, sy = cnt * py, sz = cnt * pz;
  float ce_sum = 0.f, ncorr = 0.f, g0 = 0.f, g1 = 0.f;
  if (train) {
    const int lab = p.labels[row];
    const float mx = fmaxf(q0, q1);
    const float lse = mx + logf(expf(q0 - mx) + expf(q1 - mx));
    ce_sum = ownf * (lse - (lab ? q1 : q0));
    ncorr = ownf * (((q1 > q0 ? 1 : 0) == lab) ? 1.f : 0.f);
    if (bwd) {
      g0 = ownf * wb * (expf(q0 - lse) - (lab == 0 ? 1.f : 0.f));
      g1 = ownf * wb * (expf(q1 - lse) - (lab == 1 ? 1.f : 0.f));
      if (SOFT) {      // soft_mask = softmax(logits)[1]: d soft / d (q0, q1) = p1 (1 - p1) (-1, +1)
        const float p1 = expf(q1 - lse), gs = ownf * p.dsoft[row] * p1 * (1.f - p1);
        g0 -= gs;
        g1 += gs;
      }
      if (oracle) { g0 = 0.f; g1 = 0.f; }      // the stacked logits are a constant: nothing flows back into conv10 / conv9
    }
  }
  cnt = wave_sum_dpp(cnt); sx = wave_sum_dpp(sx); sy = wave_sum_dpp(sy); sz = wave_sum_dpp(sz);
  ce_sum = wave_sum_dpp(ce_sum); ncorr = wave_sum_dpp(ncorr);
  const float db0 = wave_sum_dpp(g0), db1 = wave_sum_dpp(g1);

  // Phase C -- channel-parallel again: the logit gradients of row i come back from lane i.
  if (bwd) {
#pragma unroll
    for (int i = 0; i < SH_ROWS; ++i) {
      const float G0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, g0), i));
      const float G1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, g1), i));
      dw00 = fmaf(d0[i], G0, dw00); dw01 = fmaf(d0[i], G1, dw01);
      dw10 = fmaf(d1[i], G0, dw10); dw11 = fmaf(d1[i], G1, dw11);
      // (bf16: the partial sums below are those of the gradient as stored, i.e. as the backward kernels read it)
      const float dz0 = Elem<T>::rnd(fmaf(y[i].x, sc.x, sh.x) > 0.f ? (G0 * w00 + G1 * w01) * km[i].x : 0.f);
      const float dz1 = Elem<T>::rnd(fmaf(y[i].y, sc.y, sh.y) > 0.f ? (G0 * w10 + G1 * w11) * km[i].y : 0.f);
      Elem<T>::st2(p.dz, (size_t)(rbase + i) * 128 + ch, dz0, dz1);
      sdz0 += dz0; sdz1 += dz1;
      sdzy0 = fmaf(dz0, y[i].x, sdzy0); sdzy1 = fmaf(dz1, y[i].y, sdzy1);
    }
  }
  float* r = red[wid];
  r[ch] = sdz0; r[ch + 1] = sdz1;
  r[128 + ch] = sdzy0; r[128 + ch + 1] = sdzy1;
  r[256 + ch * 2] = dw00; r[256 + ch * 2 + 1] = dw01; r[256 + ch * 2 + 2] = dw10; r[256 + ch * 2 + 3] = dw11;
  if (lane == 0) {
    r[512] = ce_sum; r[513] = cnt; r[514] = sx; r[515] = sy; r[516] = sz; r[517] = db0; r[518] = db1; r[519] = ncorr;
  }
  __syncthreads();
  // fixed-order combination of the 16 wave partials: thread e < 520 owns one quantity
  if (tid < 520) {
    float acc = 0.f;
#pragma unroll
    for (int w = 0; w < SH_WAVES; ++w) acc += red[w][tid];
    if (tid < 128) { if (bwd) p.psum_dz[(size_t)tile * 128 + tid] = acc; }
    else if (tid < 256) { if (bwd) p.psum_dzy[(size_t)tile * 128 + tid - 128] = acc; }
    else if (tid < 512) { if (bwd) p.dw_part[(size_t)tile * 256 + tid - 256] = acc; }
    else p.part[(size_t)tile * 8 + tid - 512] = acc;
  }
}

__global__ __launch_bounds__(256) void k_seg_finalize(const t3d_seg_finalize_args p) {
  __shared__ double red[3][256];
  const int tid = threadIdx.x;
  const int T = p.B * p.tiles_per_frustum;
  for (int b = tid; b < p.B; b += 256) {
    double ce = 0, cnt = 0, sx = 0, sy = 0, sz = 0;
    for (int t = 0; t < p.tiles_per_frustum; ++t) {
      const float* q = p.part + (size_t)(b * p.tiles_per_frustum + t) * 8;
      ce += q[0]; cnt += q[1]; sx += q[2]; sy += q[3]; sz += q[4];
    }
    const double den = cnt > 1.0 ? cnt : 1.0;
    p.mask_xyz_mean[b * 3 + 0] = (float)(sx / den);
    p.mask_xyz_mean[b * 3 + 1] = (float)(sy / den);
    p.mask_xyz_mean[b * 3 + 2] = (float)(sz / den);
    if (p.seg_loss) p.seg_loss[b] = (float)(ce / (double)p.rows_per_frustum);
  }
  // conv10 bias gradient + accuracy counter: strided partial sums, then a fixed-order tree
  double a0 = 0, a1 = 0, a2 = 0;
  for (int t = tid; t < T; t += 256) {
    const float* q = p.part + (size_t)t * 8;
    a0 += q[5]; a1 += q[6]; a2 += q[7];
  }
  red[0][tid] = a0; red[1][tid] = a1; red[2][tid] = a2;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (tid < s) {
      red[0][tid] += red[0][tid + s]; red[1][tid] += red[1][tid + s]; red[2][tid] += red[2][tid + s];
    }
    __syncthreads();
  }
  if (tid == 0) {
    if (p.dbias) { p.dbias[0] = (float)red[0][0]; p.dbias[1] = (float)red[1][0]; }
    if (p.n_correct) p.n_correct[0] = (float)red[2][0];
  }
  // conv10 weight gradient: normally summed by t3d_reduce_slabs (dw_part registered as slabs); this
  // path serves callers that pass dw explicitly.
  if (p.dw != nullptr) {
    for (int e = tid; e < p.K * 2; e += 256) {
      double a = 0;
      for (int t = 0; t < T; ++t) a += p.dw_part[(size_t)t * p.K * 2 + e];
      p.dw[e] = (float)a;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// box losses
// ---------------------------------------------------------------------------------------------
constexpr int NH = 12, NS = 10;
__device__ const float kMeanDims[NS][3] = {   // class2type order: bed,table,sofa,chair,toilet,desk,dresser,night_stand,bookshelf,bathtub
    {2.114256f, 1.620300f, 0.927272f}, {0.791118f, 1.279516f, 0.718182f}, {0.923508f, 1.867419f, 0.845495f},
    {0.591958f, 0.552978f, 0.827272f}, {0.699104f, 0.454178f, 0.756250f}, {0.695190f, 1.346299f, 0.736364f},
    {0.528526f, 1.002642f, 1.172878f}, {0.500618f, 0.632163f, 0.683424f}, {0.404671f, 1.071108f, 1.688889f},
    {0.765840f, 1.398258f, 0.472728f}};

__device__ __forceinline__ float bin_center(int j) { return (float)((double)j * (2.0 * 3.14159265358979323846 / 12.0)); }
__device__ __forceinline__ float huber_f(float e, float delta) {
  const float a = fabsf(e), q = fminf(a, delta);
  return 0.5f * q * q + delta * (a - q);
}
__device__ __forceinline__ float huber_d(float a, float delta) { return a < delta ? a : delta; }   // d/da for a >= 0
__device__ float softmax_ce(const float* z, int n, int ld, int label, float* grad, float gs) {
  float mx = z[0];
  for (int i = 1; i < n; ++i) mx = fmaxf(mx, z[i * ld]);
  float s = 0.f;
  for (int i = 0; i < n; ++i) s += expf(z[i * ld] - mx);
  const float lse = mx + logf(s);
  for (int i = 0; i < n; ++i) grad[i] += gs * (expf(z[i * ld] - lse) - (i == label ? 1.f : 0.f));
  return lse - z[label * ld];
}

// compute_box3d_iou for one frustum (roi_seg_box3d_dataset.py:103-140): predicted box = arg-max bins of the raw heads `o`
// (class2angle: bin centre + residual; class2size: mean size + residual), label box from the label bins.
__device__ float head_iou(const float* o, const float* cen, const float* yc, int j, float yor, int k, const float* ydr, float* iou2d) {
  int js = 0, ks = 0;
  for (int i = 1; i < NH; ++i) if (o[3 + i] > o[3 + js]) js = i;
  for (int i = 1; i < NS; ++i) if (o[3 + 2 * NH + i] > o[3 + 2 * NH + ks]) ks = i;
  const float hp = bin_center(js) + o[3 + NH + js] * (3.14159265358979323846f / NH);
  float sp[3], sl[3];
  for (int d = 0; d < 3; ++d) {
    sp[d] = kMeanDims[ks][d] + o[3 + 2 * NH + NS + 3 * ks + d] * kMeanDims[ks][d];
    sl[d] = kMeanDims[k][d] + ydr[d];
  }
  return boxgeom::box3d_iou_params(cen, sp, hp, yc, sl, bin_center(j) + yor, iou2d);
}

__global__ __launch_bounds__(1024) void k_box_head_iou(const t3d_box_head_iou_args p) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= p.B) return;
  const float* o = p.box + (size_t)b * p.ld_box;
  float cen[3], yc[3], ydr[3];
  for (int d = 0; d < 3; ++d) {
    cen[d] = o[d] + (p.stage1_center ? p.stage1_center[b * 3 + d] : 0.f);
    yc[d] = p.y_center[b * 3 + d];
    ydr[d] = p.y_dims_reg[b * 3 + d];
  }
  float i2;
  p.iou3d[b] = head_iou(o, cen, yc, p.y_orient_cls[b], p.y_orient_reg[b], p.y_dims_cls[b], ydr, &i2);
  p.iou2d[b] = i2;
}

// GL: the per-frustum gradient vector g[67] and the head outputs o[67] live in LDS (B <= 128).  Both are indexed by the label /
// arg-max bins, i.e. dynamically: as private arrays they went to scratch (272 B per thread), every access a memory round trip in
// a program that is one serial chain per frustum; an LDS access costs a quarter of that.  Row stride 67 (odd): conflict-free.
template <bool GL>
__global__ __launch_bounds__(1024) void k_strong_loss(const t3d_strong_loss_args p) {
  __shared__ float red[1024];
  __shared__ float s_norm;
  __shared__ float g_lds[GL ? 128 * 67 : 1];
  __shared__ float o_lds[GL ? 128 * 67 : 1];
  if (GL) {          // the head outputs, staged coalesced
    for (int f = threadIdx.x; f < p.B * 67; f += 1024) o_lds[f] = p.box[(size_t)(f / 67) * p.ld_box + f % 67];
  }
  const int b = threadIdx.x;
  const bool ok = b < p.B;
  const float w3d = ok ? (float)(1 - p.is_data_2D[b]) : 0.f;
  red[b] = w3d;
  __syncthreads();
  if (b == 0) {
    float s = 0.f;
    for (int i = 0; i < p.B; ++i) s += red[i];
    s_norm = p.normalize_by_3d_count ? 1.0f / (s + 1e-3f) : 1.0f / (float)p.B;
  }
  __syncthreads();
  const float norm = s_norm;
  float total = 0.f;
  if (ok) {
    const float* o = GL ? o_lds + b * 67 : p.box + (size_t)b * p.ld_box;
    const t3d_strong_weights& W = p.wts;
    const float gs = w3d * norm;
    float g_priv[GL ? 1 : 67];
    float* g = GL ? g_lds + b * 67 : g_priv;
    for (int i = 0; i < 67; ++i) g[i] = 0.f;
    float gc[3] = {0.f, 0.f, 0.f}, gs1[3] = {0.f, 0.f, 0.f};
    const float s1[3] = {p.stage1_center[b * 3], p.stage1_center[b * 3 + 1], p.stage1_center[b * 3 + 2]};
    const float cen[3] = {o[0] + s1[0], o[1] + s1[1], o[2] + s1[2]};
    const float yc[3] = {p.y_center[b * 3], p.y_center[b * 3 + 1], p.y_center[b * 3 + 2]};
    const int j = p.y_orient_cls[b], k = p.y_dims_cls[b];
    const float yor = p.y_orient_reg[b];
    const float ydr[3] = {p.y_dims_reg[b * 3], p.y_dims_reg[b * 3 + 1], p.y_dims_reg[b * 3 + 2]};
    const float bm = W.box_multiplier;

    // centre (delta 2) and stage-1 centre (delta 1)
    float dist = sqrtf((yc[0] - cen[0]) * (yc[0] - cen[0]) + (yc[1] - cen[1]) * (yc[1] - cen[1]) + (yc[2] - cen[2]) * (yc[2] - cen[2]));
    const float l_center = huber_f(dist, 2.f);
    if (dist > 0.f) {
      const float f = gs * bm * W.center * huber_d(dist, 2.f) / dist;
      for (int d = 0; d < 3; ++d) gc[d] += f * (cen[d] - yc[d]);
    }
    dist = sqrtf((yc[0] - s1[0]) * (yc[0] - s1[0]) + (yc[1] - s1[1]) * (yc[1] - s1[1]) + (yc[2] - s1[2]) * (yc[2] - s1[2]));
    const float l_s1 = huber_f(dist, 1.f);
    if (dist > 0.f) {
      const float f = gs * bm * W.tnet_center * huber_d(dist, 1.f) / dist;
      for (int d = 0; d < 3; ++d) gs1[d] += f * (s1[d] - yc[d]);
    }
    // heading
    const float l_hcls = softmax_ce(o + 3, NH, 1, j, g + 3, gs * bm * W.orient_cls);
    const float hrn = o[3 + NH + j];
    const float eh = hrn - yor / (3.14159265358979323846f / NH);
    const float l_hres = huber_f(eh, 1.f);
    g[3 + NH + j] += gs * bm * W.orient_reg * fmaxf(-1.f, fminf(1.f, eh));
    // size
    const float l_scls = softmax_ce(o + 3 + 2 * NH, NS, 1, k, g + 3 + 2 * NH, gs * bm * W.dims_cls);
    const float* srn = o + 3 + 2 * NH + NS + 3 * k;
    float ds[3], sd = 0.f;
    for (int d = 0; d < 3; ++d) { ds[d] = srn[d] - ydr[d] / kMeanDims[k][d]; sd += ds[d] * ds[d]; }
    sd = sqrtf(sd);
    const float l_sres = huber_f(sd, 1.f);
    if (sd > 0.f) {
      const float f = gs * bm * W.dims_reg * huber_d(sd, 1.f) / sd;
      for (int d = 0; d < 3; ++d) g[3 + 2 * NH + NS + 3 * k + d] += f * ds[d];
    }
    // corners of the GT-bin box; sizes = anchor + 2*residual (model_util.py:158-159)
    const float th = bin_center(j) + hrn * (3.14159265358979323846f / NH);
    const float c = cosf(th), s = sinf(th);
    float sz3[3];
    for (int d = 0; d < 3; ++d) { const float r = srn[d] * kMeanDims[k][d]; sz3[d] = (kMeanDims[k][d] + r) + r; }
    const float hl = bin_center(j) + yor;
    const float cg = cosf(hl), sg = sinf(hl);
    const float cgf = cosf(hl + 3.14159265358979323846f), sgf = sinf(hl + 3.14159265358979323846f);
    const float gl[3] = {kMeanDims[k][0] + ydr[0], kMeanDims[k][1] + ydr[1], kMeanDims[k][2] + ydr[2]};
    float l_corner = 0.f, gth = 0.f, gl_ = 0.f, gw_ = 0.f, gh_ = 0.f;
    for (int i = 0; i < 8; ++i) {
      const float sxi = ((i >> 1) & 1) ? -1.f : 1.f;             // + + - - + + - -
      const float syi = (i >> 2) ? -1.f : 1.f;                    // + + + + - - - -
      const float szi = (((i + 1) >> 1) & 1) ? -1.f : 1.f;        // + - - + + - - +
      const float x = sxi * sz3[0] * 0.5f, y = syi * sz3[2] * 0.5f, z = szi * sz3[1] * 0.5f;
      const float cp[3] = {c * x + s * z + cen[0], y + cen[1], -s * x + c * z + cen[2]};
      const float xg = sxi * gl[0] * 0.5f, yg = syi * gl[2] * 0.5f, zg = szi * gl[1] * 0.5f;
      const float t1[3] = {cg * xg + sg * zg + yc[0], yg + yc[1], -sg * xg + cg * zg + yc[2]};
      const float t2[3] = {cgf * xg + sgf * zg + yc[0], yg + yc[1], -sgf * xg + cgf * zg + yc[2]};
      float d1 = 0.f, d2 = 0.f;
      for (int d = 0; d < 3; ++d) { d1 += (cp[d] - t1[d]) * (cp[d] - t1[d]); d2 += (cp[d] - t2[d]) * (cp[d] - t2[d]); }
      d1 = sqrtf(d1); d2 = sqrtf(d2);
      const bool first = d1 <= d2;
      const float dm = first ? d1 : d2;
      l_corner += huber_f(dm, 1.f) * 0.125f;
      if (dm > 0.f) {
        const float f = gs * W.corner * 0.125f * huber_d(dm, 1.f) / dm;
        const float gv[3] = {f * (cp[0] - (first ? t1[0] : t2[0])), f * (cp[1] - (first ? t1[1] : t2[1])),
                             f * (cp[2] - (first ? t1[2] : t2[2]))};
        for (int d = 0; d < 3; ++d) gc[d] += gv[d];
        gth += gv[0] * (-s * x + c * z) + gv[2] * (-c * x - s * z);
        gl_ += (gv[0] * c - gv[2] * s) * sxi * 0.5f;
        gh_ += gv[1] * syi * 0.5f;
        gw_ += (gv[0] * s + gv[2] * c) * szi * 0.5f;
      }
    }
    g[3 + NH + j] += gth * (3.14159265358979323846f / NH);
    g[3 + 2 * NH + NS + 3 * k + 0] += gl_ * 2.f * kMeanDims[k][0];
    g[3 + 2 * NH + NS + 3 * k + 1] += gw_ * 2.f * kMeanDims[k][1];
    g[3 + 2 * NH + NS + 3 * k + 2] += gh_ * 2.f * kMeanDims[k][2];
    for (int d = 0; d < 3; ++d) g[d] = gc[d];

    const float box_l = bm * (W.center * l_center + W.orient_cls * l_hcls + W.dims_cls * l_scls + W.orient_reg * l_hres +
                              W.dims_reg * l_sres + W.tnet_center * l_s1) + W.corner * l_corner;
    const float seg_l = p.seg_loss ? p.seg_loss[b] : 0.f;
    total = w3d * (W.cross_entropy * seg_l + box_l);
    for (int i = 0; i < 67; ++i) p.dbox[(size_t)b * 67 + i] = g[i];
    for (int d = 0; d < 3; ++d) {
      p.dstage1[b * 3 + d] = gc[d] + gs1[d];
      p.center[b * 3 + d] = cen[d];
    }
    float* t = p.terms + (size_t)b * 8;
    t[0] = seg_l; t[1] = l_center; t[2] = l_s1; t[3] = l_hcls; t[4] = l_hres; t[5] = l_scls; t[6] = l_sres; t[7] = l_corner;
    p.total_losses[b] = total;
    // anchor -> reg of the PREDICTED bins (first arg-max)
    int js = 0, ks = 0;
    for (int i = 1; i < NH; ++i) if (o[3 + i] > o[3 + js]) js = i;
    for (int i = 1; i < NS; ++i) if (o[3 + 2 * NH + i] > o[3 + 2 * NH + ks]) ks = i;
    for (int d = 0; d < 3; ++d)
      p.reg_dims[b * 3 + d] = fmaxf(kMeanDims[ks][d] + o[3 + 2 * NH + NS + 3 * ks + d] * kMeanDims[ks][d], 1e-5f);
    p.reg_theta[b] = bin_center(js) + o[3 + NH + js] * (3.14159265358979323846f / NH);
    if (p.iou3d && p.B > 512) {          // small batches: the summary runs beside the loss, in another wave (below)
      float i2;
      p.iou3d[b] = head_iou(o, cen, yc, j, yor, k, ydr, &i2);
      p.iou2d[b] = i2;
    }
  }
  // the IoU summary of frustum f on thread 512 + f: a different wave than the loss of f, so the two serial chains overlap
  if (p.iou3d && p.B <= 512 && b >= 512 && b - 512 < p.B) {
    const int f = b - 512;
    const float* o = GL ? o_lds + f * 67 : p.box + (size_t)f * p.ld_box;
    float cen[3], yc[3], ydr[3];
    for (int d = 0; d < 3; ++d) {
      cen[d] = o[d] + p.stage1_center[f * 3 + d];
      yc[d] = p.y_center[f * 3 + d];
      ydr[d] = p.y_dims_reg[f * 3 + d];
    }
    float i2;
    p.iou3d[f] = head_iou(o, cen, yc, p.y_orient_cls[f], p.y_orient_reg[f], p.y_dims_cls[f], ydr, &i2);
    p.iou2d[f] = i2;
  }
  __syncthreads();
  red[b] = total;
  __syncthreads();
  if (b == 0) {
    double s = 0.0;
    for (int i = 0; i < p.B; ++i) s += (double)red[i];
    p.loss[0] = (float)(s * (double)norm);
  }
}

}  // namespace

namespace {
// out[m,k] = act(a)[m,k] * mask / keep (t3d.h: standalone tf_util.dropout on a per-point tensor)
__global__ __launch_bounds__(256) void k_act_dropout(const t3d_act_dropout_args p) {
  const size_t n = (size_t)p.M * p.K;
  const float inv_keep = (p.mask && p.keep_prob < 1.f) ? 1.0f / p.keep_prob : 1.f;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const size_t m = i / p.K;
    const int k = (int)(i - m * p.K);
    float v = ld_elem(p.a.x, m * p.a.ldx + p.a.coff + k, p.a.dtype);
    if (p.a.scale) v = fmaf(v, p.a.scale[k], p.a.shift[k]);
    if (p.a.relu) v = fmaxf(v, 0.f);
    if (p.a.sub) v -= p.a.sub[(m / p.rows_per_frustum) * p.a.sub_ld + k];
    if (p.mask) v *= p.mask[i] * inv_keep;
    p.out[i] = v;
  }
}
}  // namespace

extern "C" int t3d_act_dropout(const t3d_act_dropout_args* a, t3d_stream_t stream) {
  if (!a || !a->a.x || !a->out || (a->a.scale && !a->a.shift)) return T3D_ERR_ARG;
  if (a->M <= 0 || a->K <= 0 || a->rows_per_frustum <= 0 || a->keep_prob <= 0.f) return T3D_ERR_SHAPE;
  const size_t n = (size_t)a->M * a->K;
  size_t blocks = (n + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  T3D_LAUNCH(k_act_dropout, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), *a);
  T3D_CHECK_LAUNCH();
  return T3D_OK;
}

extern "C" int t3d_seg_head(const t3d_seg_head_args* a, t3d_stream_t stream) {
  T3D_ABI_TAKE(seg_head_args, a);
  if (a && a->struct_size != sizeof(*a)) return T3D_ERR_ABI;
  if (!a || !a->y || !a->scale || !a->shift || !a->w || !a->bias || !a->pc || !a->logits || !a->mask || !a->part)
    return T3D_ERR_ARG;
  if (a->labels && !a->is_data_2D) return T3D_ERR_ARG;
  if (a->dz && (!a->labels || !a->psum_dz || !a->psum_dzy || !a->dw_part)) return T3D_ERR_ARG;
  if (a->K != 128 || a->M % T3D_TILE_ROWS || a->rows_per_frustum % T3D_TILE_ROWS) return T3D_ERR_SHAPE;
  if (a->dsoft != nullptr) {
    if (!a->labels || !a->dz) return T3D_ERR_ARG;      // the soft-mask gradient belongs to a training backward
    if (a->dtype == T3D_BF16) T3D_LAUNCH((k_seg_head<bf16_t, true>), dim3(a->M / 128), dim3(64 * SH_WAVES), 0, static_cast<hipStream_t>(stream), *a);
    else if (a->dtype == T3D_F32) T3D_LAUNCH((k_seg_head<float, true>), dim3(a->M / 128), dim3(64 * SH_WAVES), 0, static_cast<hipStream_t>(stream), *a);
    else return T3D_ERR_ARG;
    T3D_CHECK_LAUNCH();
    return T3D_OK;
  }
  if (a->dtype == T3D_BF16) T3D_LAUNCH(k_seg_head<bf16_t>, dim3(a->M / 128), dim3(64 * SH_WAVES), 0, static_cast<hipStream_t>(stream), *a);
  else if (a->dtype == T3D_F32) T3D_LAUNCH(k_seg_head<float>, dim3(a->M / 128), dim3(64 * SH_WAVES), 0, static_cast<hipStream_t>(stream), *a);
  else return T3D_ERR_ARG;
  T3D_CHECK_LAUNCH();
  return T3D_OK;
}

extern "C" int t3d_seg_finalize(const t3d_seg_finalize_args* a, t3d_stream_t stream) {
  if (!a || !a->part || !a->mask_xyz_mean) return T3D_ERR_ARG;
  if (a->dw && !a->dw_part) return T3D_ERR_ARG;
  T3D_LAUNCH(k_seg_finalize, dim3(1), dim3(256), 0, static_cast<hipStream_t>(stream), *a);
  T3D_CHECK_LAUNCH();
  return T3D_OK;
}

extern "C" int t3d_strong_loss(const t3d_strong_loss_args* a, t3d_stream_t stream) {
  if (!a || !a->box || !a->stage1_center || !a->y_center || !a->y_orient_cls || !a->y_orient_reg || !a->y_dims_cls ||
      !a->y_dims_reg || !a->is_data_2D || !a->dbox || !a->dstage1 || !a->terms || !a->total_losses || !a->loss ||
      !a->center || !a->reg_dims || !a->reg_theta)
    return T3D_ERR_ARG;
  if (a->B <= 0 || a->B > 1024) return T3D_ERR_SHAPE;
  if ((a->iou2d == nullptr) != (a->iou3d == nullptr)) return T3D_ERR_ARG;
  if (a->B <= 128) T3D_LAUNCH(k_strong_loss<true>, dim3(1), dim3(1024), 0, static_cast<hipStream_t>(stream), *a);
  else T3D_LAUNCH(k_strong_loss<false>, dim3(1), dim3(1024), 0, static_cast<hipStream_t>(stream), *a);
  T3D_CHECK_LAUNCH();
  return T3D_OK;
}

namespace {
__global__ __launch_bounds__(256) void k_box2d_feats(const t3d_box2d_feats_args p) {
  const int W = p.n_oh + 4;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= p.B * W) return;
  const int b = i / W, j = i - b * W;
  if (j < p.n_oh) { p.out[i] = p.one_hot[b * p.n_oh + j]; return; }
  const int q = j - p.n_oh;                                  // 0 left, 1 top, 2 right, 3 bottom
  const float d = (q & 1) ? p.img_dim[b * 2 + 0] : p.img_dim[b * 2 + 1];      // rows for top / bottom, cols for left / right
  p.out[i] = p.box2D[b * 4 + q] / d;
}
}  // namespace

extern "C" int t3d_box2d_feats(const t3d_box2d_feats_args* a, t3d_stream_t stream) {
  if (!a || !a->box2D || !a->img_dim || !a->out || (a->n_oh > 0 && !a->one_hot)) return T3D_ERR_ARG;
  if (a->B <= 0 || a->n_oh < 0) return T3D_ERR_SHAPE;
  const int n = a->B * (a->n_oh + 4);
  T3D_LAUNCH(k_box2d_feats, dim3((n + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream), *a);
  T3D_CHECK_LAUNCH();
  return T3D_OK;
}

extern "C" int t3d_box_head_iou(const t3d_box_head_iou_args* a, t3d_stream_t stream) {
  if (!a || !a->box || !a->y_center || !a->y_orient_cls || !a->y_orient_reg || !a->y_dims_cls || !a->y_dims_reg || !a->iou2d ||
      !a->iou3d)
    return T3D_ERR_ARG;
  if (a->B <= 0 || a->ld_box < 67) return T3D_ERR_SHAPE;
  T3D_LAUNCH(k_box_head_iou, dim3((a->B + 63) / 64), dim3(64), 0, static_cast<hipStream_t>(stream), *a);
  T3D_CHECK_LAUNCH();
  return T3D_OK;
}
