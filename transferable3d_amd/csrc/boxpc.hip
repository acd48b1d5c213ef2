// Box-PC Fit network pieces: the box <-> point-cloud representation (forward and its gradient w.r.t. the box) and
// the Box-PC loss (forward + backward).
//
// representation: models/tf_util.py:764-795 (tf_get_box_pc_representation) with the surface centres / inward normals
//                 of tf_create_3D_box_by_surface_centers (893-942): for a box (centre c, dims (l,w,h), heading th),
//                 with t = xyz - c and the box axes ex = (cos, 0, -sin), ez = (sin, 0, cos):
//                   u = ex.t, v = t.y, q = ez.t
//                   d = (l/2 - u, l/2 + u, h/2 - v, h/2 + v, w/2 - q, w/2 + q)
//                 (n_k.(t - p_k) with p_k = R p_loc, n_k = R n_loc and R orthonormal); rep = [pc channels | d].
// loss:           boxpc_sunrgbd.py:106-193 (2-way softmax CE vs iou > bound; Huber(delta=1) on the centre / size /
//                 angle deltas, weighted .34/.33/.33 and BOXPC_WEIGHT_DELTA).
#include "common.h"

namespace {

__device__ const float kMean[10][3] = {
    {2.114256f, 1.620300f, 0.927272f}, {0.791118f, 1.279516f, 0.718182f}, {0.923508f, 1.867419f, 0.845495f},
    {0.591958f, 0.552978f, 0.827272f}, {0.699104f, 0.454178f, 0.756250f}, {0.695190f, 1.346299f, 0.736364f},
    {0.528526f, 1.002642f, 1.172878f}, {0.500618f, 0.632163f, 0.683424f}, {0.404671f, 1.071108f, 1.688889f},
    {0.765840f, 1.398258f, 0.472728f}};

struct Box { float cx, cy, cz, l, w, h, c, s; };

__device__ __forceinline__ Box load_box(const t3d_boxpc_rep_args& p, int b) {
  Box x;
  x.cx = p.center[b * 3]; x.cy = p.center[b * 3 + 1]; x.cz = p.center[b * 3 + 2];
  float th;
  if (p.y_dims_cls != nullptr) {      // label form: convert_raw_y_box_to_reg_format (boxpc_sunrgbd.py:206-230)
    const int k = p.y_dims_cls[b], j = p.y_orient_cls[b];
    x.l = fmaxf(kMean[k][0] + p.dims[b * 3], 1e-5f);
    x.w = fmaxf(kMean[k][1] + p.dims[b * 3 + 1], 1e-5f);
    x.h = fmaxf(kMean[k][2] + p.dims[b * 3 + 2], 1e-5f);
    th = (float)((double)j * (2.0 * 3.14159265358979323846 / 12.0)) + p.theta[b];
  } else {
    x.l = p.dims[b * 3]; x.w = p.dims[b * 3 + 1]; x.h = p.dims[b * 3 + 2];
    th = p.theta[b];
  }
  x.c = cosf(th); x.s = sinf(th);
  if (p.box_out != nullptr && threadIdx.x == 0 && (blockIdx.x * blockDim.x) % p.rows_per_frustum == 0) {
    float* o = p.box_out + b * 7;
    o[0] = x.cx; o[1] = x.cy; o[2] = x.cz; o[3] = x.l; o[4] = x.w; o[5] = x.h; o[6] = th;
  }
  return x;
}

__global__ __launch_bounds__(256) void k_boxpc_rep(const t3d_boxpc_rep_args p) {
  const int m = blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= p.M) return;
  const int b = m / p.rows_per_frustum;
  const Box x = load_box(p, b);
  const float* src = p.pc + (size_t)m * p.ld_pc;
  float* dst = p.rep + (size_t)m * p.ld_rep;
  for (int i = 0; i < p.C; ++i) dst[i] = src[i];
  const float tx = src[0] - x.cx, ty = src[1] - x.cy, tz = src[2] - x.cz;
  const float u = x.c * tx - x.s * tz, q = x.s * tx + x.c * tz;
  dst[p.C + 0] = 0.5f * x.l - u;
  dst[p.C + 1] = 0.5f * x.l + u;
  dst[p.C + 2] = 0.5f * x.h - ty;
  dst[p.C + 3] = 0.5f * x.h + ty;
  dst[p.C + 4] = 0.5f * x.w - q;
  dst[p.C + 5] = 0.5f * x.w + q;
  for (int i = p.C + 6; i < p.ld_rep; ++i) dst[i] = 0.f;
}

// d(rep distances)/d(box): one workgroup per frustum reduces over its points.
//   g = drep[m, C..C+5];  du = g1 - g0, dv = g3 - g2, dq = g5 - g4
//   d centre = -(du*ex + dv*ey + dq*ez) ; d l = (g0+g1)/2, d h = (g2+g3)/2, d w = (g4+g5)/2
//   d theta = du * d u/d th + dq * d q/d th,  d u/d th = -s tx - c tz = -q,  d q/d th = c tx - s tz = u
__global__ __launch_bounds__(256) void k_boxpc_rep_bwd(const t3d_boxpc_rep_bwd_args p) {
  __shared__ float red[7][256];
  const int b = blockIdx.x, tid = threadIdx.x;
  const float cx = p.box[b * 7], cy = p.box[b * 7 + 1], cz = p.box[b * 7 + 2], th = p.box[b * 7 + 6];
  (void)cy;
  const float c = cosf(th), s = sinf(th);
  float a[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (int n = tid; n < p.rows_per_frustum; n += 256) {
    const size_t m = (size_t)b * p.rows_per_frustum + n;
    const float* g = p.drep + m * p.ld_drep + p.coff;
    const float* src = p.pc + m * p.ld_pc;
    const float tx = src[0] - cx, tz = src[2] - cz;
    const float u = c * tx - s * tz, q = s * tx + c * tz;
    const float du = g[1] - g[0], dv = g[3] - g[2], dq = g[5] - g[4];
    a[0] -= du * c + dq * s;
    a[1] -= dv;
    a[2] -= -du * s + dq * c;
    a[3] += 0.5f * (g[0] + g[1]);     // l
    a[4] += 0.5f * (g[4] + g[5]);     // w
    a[5] += 0.5f * (g[2] + g[3]);     // h
    a[6] += -du * q + dq * u;
  }
  for (int i = 0; i < 7; ++i) red[i][tid] = a[i];
  __syncthreads();
  for (int st = 128; st > 0; st >>= 1) {
    if (tid < st)
      for (int i = 0; i < 7; ++i) red[i][tid] += red[i][tid + st];
    __syncthreads();
  }
  if (tid < 7) p.dbox[b * 7 + tid] = red[tid][0];
}

__device__ __forceinline__ float hub(float e) { const float a = fabsf(e), q = fminf(a, 1.f); return 0.5f * q * q + (a - q); }
__device__ __forceinline__ float hubd(float e) { return fmaxf(-1.f, fminf(1.f, e)); }

__global__ __launch_bounds__(1024) void k_boxpc_loss(const t3d_boxpc_loss_args p) {
  __shared__ float red[1024];
  const int b = threadIdx.x;
  float total = 0.f;
  if (b < p.B) {
    const float* o = p.out + (size_t)b * 9;
    const float inv = 1.0f / (float)p.B;
    float g[9];
    const float l0 = o[7], l1 = o[8];
    const int cls = p.y_box_iou[b] > p.fit_bound ? 1 : 0;
    const float mx = fmaxf(l0, l1);
    const float lse = mx + logf(expf(l0 - mx) + expf(l1 - mx));
    const float ce = lse - (cls ? l1 : l0);
    const float p1 = expf(l1 - lse);
    g[7] = inv * p.w_cls * (expf(l0 - lse) - (cls == 0 ? 1.f : 0.f));
    g[8] = inv * p.w_cls * (p1 - (cls == 1 ? 1.f : 0.f));
    float wl = 1.f;
    if (p.weigh_by_cls_gt) wl = 1.f - p.y_box_iou[b];
    if (p.weigh_by_cls_conf) wl = 1.f - p1;            // logits_for_weigh is stop_gradient (boxpc_sunrgbd.py:74)
    float lc = 0.f, ls = 0.f;
    for (int d = 0; d < 3; ++d) {
      const float ec = o[d] - p.y_center_delta[b * 3 + d], es = o[3 + d] - p.y_dims_delta[b * 3 + d];
      lc += hub(ec) * (1.f / 3.f);
      ls += hub(es) * (1.f / 3.f);
      g[d] = inv * p.w_delta * p.w_center * wl * hubd(ec) * (1.f / 3.f);
      g[3 + d] = inv * p.w_delta * p.w_size * wl * hubd(es) * (1.f / 3.f);
    }
    const float ea = o[6] - p.y_orient_delta[b];
    const float la = hub(ea);
    g[6] = inv * p.w_delta * p.w_angle * wl * hubd(ea);
    const float delta = wl * (p.w_center * lc + p.w_size * ls + p.w_angle * la);
    total = p.w_cls * ce + p.w_delta * delta;
    for (int i = 0; i < 9; ++i) p.dout[(size_t)b * 9 + i] = g[i];
    p.terms[b * 4 + 0] = ce; p.terms[b * 4 + 1] = delta; p.terms[b * 4 + 2] = p1; p.terms[b * 4 + 3] = total;
  }
  red[b] = total;
  __syncthreads();
  if (b == 0) {
    double s = 0.0;
    for (int i = 0; i < p.B; ++i) s += (double)red[i];
    p.loss[0] = (float)(s / (double)p.B);
  }
}

}  // namespace

extern "C" int t3d_boxpc_rep(const t3d_boxpc_rep_args* a, t3d_stream_t stream) {
  if (!a || !a->pc || !a->center || !a->dims || !a->theta || !a->rep) return T3D_ERR_ARG;
  if (a->y_dims_cls && !a->y_orient_cls) return T3D_ERR_ARG;
  if (a->M <= 0 || a->C < 3 || a->ld_rep < a->C + 6 || a->rows_per_frustum % 256) return T3D_ERR_SHAPE;
  T3D_LAUNCH(k_boxpc_rep, dim3((a->M + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream), *a);
  T3D_CHECK_LAUNCH();
  return T3D_OK;
}

extern "C" int t3d_boxpc_rep_bwd(const t3d_boxpc_rep_bwd_args* a, t3d_stream_t stream) {
  if (!a || !a->pc || !a->box || !a->drep || !a->dbox) return T3D_ERR_ARG;
  if (a->B <= 0 || a->rows_per_frustum <= 0) return T3D_ERR_SHAPE;
  T3D_LAUNCH(k_boxpc_rep_bwd, dim3(a->B), dim3(256), 0, static_cast<hipStream_t>(stream), *a);
  T3D_CHECK_LAUNCH();
  return T3D_OK;
}

extern "C" int t3d_boxpc_loss(const t3d_boxpc_loss_args* a, t3d_stream_t stream) {
  if (!a || !a->out || !a->y_box_iou || !a->y_center_delta || !a->y_dims_delta || !a->y_orient_delta || !a->dout ||
      !a->terms || !a->loss)
    return T3D_ERR_ARG;
  if (a->B <= 0 || a->B > 1024) return T3D_ERR_SHAPE;
  T3D_LAUNCH(k_boxpc_loss, dim3(1), dim3(1024), 0, static_cast<hipStream_t>(stream), *a);
  T3D_CHECK_LAUNCH();
  return T3D_OK;
}
