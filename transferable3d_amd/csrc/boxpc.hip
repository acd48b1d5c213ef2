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
#include "boxgeom_dev.h"

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
  const float km = p.rowmask ? p.rowmask[m] : 1.f;      // --mask_pc_for_boxpc (test_semisup.py:103-105): the net sees pc * mask
  for (int i = 0; i < p.C; ++i) dst[i] = src[i] * km;
  const float tx = src[0] * km - x.cx, ty = src[1] * km - x.cy, tz = src[2] * km - x.cz;
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
    if (p.weigh_by_cls_conf) wl = 1.f - p1;            // logits_for_weigh (boxpc_sunrgbd.py:73-74)
    const float wp = p.weigh_pred_by_cls_conf ? 1.f - p1 : 1.f;      // the predicted deltas are weighed (boxpc_sunrgbd.py:84-92)
    const bool mse = p.delta_loss_mse != 0;
    float lc = 0.f, ls = 0.f, dwp = 0.f;               // dwp = d(delta loss / wl) / d wp
    for (int d = 0; d < 3; ++d) {
      const float ec = o[d] * wp - p.y_center_delta[b * 3 + d], es = o[3 + d] * wp - p.y_dims_delta[b * 3 + d];
      lc += (mse ? ec * ec : hub(ec)) * (1.f / 3.f);
      ls += (mse ? es * es : hub(es)) * (1.f / 3.f);
      const float gc = p.w_center * (mse ? 2.f * ec : hubd(ec)) * (1.f / 3.f), gs = p.w_size * (mse ? 2.f * es : hubd(es)) * (1.f / 3.f);
      g[d] = inv * p.w_delta * wl * wp * gc;
      g[3 + d] = inv * p.w_delta * wl * wp * gs;
      dwp += gc * o[d] + gs * o[3 + d];
    }
    const float ea = o[6] * wp - p.y_orient_delta[b];
    const float la = mse ? ea * ea : hub(ea), ga = mse ? 2.f * ea : hubd(ea);
    g[6] = inv * p.w_delta * p.w_angle * wl * wp * ga;
    dwp += p.w_angle * ga * o[6];
    const float unw = p.w_center * lc + p.w_size * ls + p.w_angle * la;
    const float delta = wl * unw;
    if (p.grad_cls_via_delta) {                        // p_fit inside wl / wp is differentiated: d p1 / d(l0, l1) = p1 (1 - p1) (-1, +1)
      float dp1 = 0.f;
      if (p.weigh_by_cls_conf) dp1 -= unw;
      if (p.weigh_pred_by_cls_conf) dp1 -= wl * dwp;
      const float t = inv * p.w_delta * dp1 * p1 * (1.f - p1);
      g[7] -= t;
      g[8] += t;
    }
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

// ---------------------------------------------------------------------------------------------
// stage-c glue (train_semisup_adv.py:331-411, semisup_v1_sunrgbd.py:343-407)
// ---------------------------------------------------------------------------------------------

// out[m, j] = sum_n dy[m,n] * w[(k0+j), n], j < kn <= 8: the input gradient of a per-point layer restricted to a few
// input channels (the six distance channels of the Box-PC representation).  64 rows per workgroup.
__global__ __launch_bounds__(256) void k_dgrad_narrow(const t3d_pointmlp_dgrad_narrow_args p) {
  __shared__ float dy_s[64][129];
  __shared__ float w_s[8][128];
  const int tid = threadIdx.x, row0 = blockIdx.x * 64;
  const int r = tid & 63, kq = tid >> 6;
  float acc0 = 0.f, acc1 = 0.f;
  for (int n0 = 0; n0 < p.N; n0 += 128) {
    __syncthreads();
    for (int i = tid; i < 8 * 128; i += 256) {
      const int k = i >> 7, n = i & 127;
      w_s[k][n] = (k < p.kn) ? p.w[(size_t)(p.k0 + k) * p.N + n0 + n] : 0.f;
    }
    for (int i = tid; i < 64 * 32; i += 256) {
      const int rr = i >> 5, c4 = (i & 31) * 4;
      const size_t o = (size_t)(row0 + rr) * p.N + n0 + c4;
      const float4 dz = ld_elem4(p.dy.dz, o, p.dy.dtype);
      const float4 y = ld_elem4(p.dy.y, o, p.dy.dtype);
      const float4 c0 = *reinterpret_cast<const float4*>(p.dy.coef + n0 + c4);
      const float4 c1 = *reinterpret_cast<const float4*>(p.dy.coef + p.N + n0 + c4);
      const float4 c2 = *reinterpret_cast<const float4*>(p.dy.coef + 2 * p.N + n0 + c4);
      dy_s[rr][c4 + 0] = fmaf(c0.x, dz.x, fmaf(c1.x, y.x, c2.x));
      dy_s[rr][c4 + 1] = fmaf(c0.y, dz.y, fmaf(c1.y, y.y, c2.y));
      dy_s[rr][c4 + 2] = fmaf(c0.z, dz.z, fmaf(c1.z, y.z, c2.z));
      dy_s[rr][c4 + 3] = fmaf(c0.w, dz.w, fmaf(c1.w, y.w, c2.w));
    }
    __syncthreads();
#pragma unroll 8
    for (int n = 0; n < 128; ++n) {
      const float d = dy_s[r][n];
      acc0 = fmaf(d, w_s[kq][n], acc0);
      acc1 = fmaf(d, w_s[kq + 4][n], acc1);
    }
  }
  float* o = p.out + (size_t)(row0 + r) * p.ld_out;
  if (kq < p.kn) o[kq] = acc0;
  if (kq + 4 < p.kn) o[kq + 4] = acc1;
}

__device__ const float kMeanG[10][3] = {
    {2.114256f, 1.620300f, 0.927272f}, {0.791118f, 1.279516f, 0.718182f}, {0.923508f, 1.867419f, 0.845495f},
    {0.591958f, 0.552978f, 0.827272f}, {0.699104f, 0.454178f, 0.756250f}, {0.695190f, 1.346299f, 0.736364f},
    {0.528526f, 1.002642f, 1.172878f}, {0.500618f, 0.632163f, 0.683424f}, {0.404671f, 1.071108f, 1.688889f},
    {0.765840f, 1.398258f, 0.472728f}};

// total = strong + w_weak * intraclass + w_fit * mean_b( -log(0.01 + p_fit) * (only_2d ? is2D : 1) )
// intraclass (weak_losses.py:267-291, huber): mean over the trained classes of mean_{b in class, d} huber(dims - stopgrad(mean)),
// an empty class contributing 0.
__global__ __launch_bounds__(1024) void k_semi_final_loss(const t3d_semi_final_loss_args p) {
  __shared__ float cmean[10][3];
  __shared__ float ccount[10];
  __shared__ float red[2][1024];
  const int b = threadIdx.x;
  int cls = 0;
  if (b < p.B) {
    const float* oh = p.one_hot + (size_t)b * 10;
    for (int i = 1; i < 10; ++i) if (oh[i] > oh[cls]) cls = i;
  }
  if (b < 10) {
    float s[3] = {0.f, 0.f, 0.f}, n = 0.f;
    for (int i = 0; i < p.B; ++i) {
      const float* oh = p.one_hot + (size_t)i * 10;
      int ci = 0;
      for (int k = 1; k < 10; ++k) if (oh[k] > oh[ci]) ci = k;
      if (ci == b) { n += 1.f; for (int d = 0; d < 3; ++d) s[d] += p.reg_dims[i * 3 + d]; }
    }
    ccount[b] = n;
    for (int d = 0; d < 3; ++d) cmean[b][d] = n > 0.f ? s[d] / n : 0.f;
  }
  __syncthreads();
  int T = 0;
  for (int i = 0; i < 10; ++i) T += p.train_classes[i] ? 1 : 0;
  float intra = 0.f, fit = 0.f;
  if (b < p.B) {
    float gd[3] = {0.f, 0.f, 0.f};
    if (p.w_weak != 0.f && T > 0 && p.train_classes[cls]) {
      const float sc = 1.0f / (3.0f * ccount[cls] * (float)T);
      for (int d = 0; d < 3; ++d) {
        const float e = p.reg_dims[b * 3 + d] - cmean[cls][d];
        intra += hub(e) * sc;
        gd[d] = p.w_weak * hubd(e) * sc;
      }
    }
    for (int d = 0; d < 3; ++d) p.d_dims[b * 3 + d] = gd[d];
    const float* o = p.out9 + (size_t)b * 9;
    const float l0 = o[7], l1 = o[8];
    const float mx = fmaxf(l0, l1);
    const float lse = mx + logf(expf(l0 - mx) + expf(l1 - mx));
    const float pf = expf(l1 - lse);
    const float m = p.fit_only_2d ? (float)p.is_data_2D[b] : 1.f;
    fit = -logf(0.01f + pf) * m;
    const float dp = -p.w_fit * m / ((0.01f + pf) * (float)p.B);
    float* g = p.dout9 + (size_t)b * 9;
    for (int i = 0; i < 7; ++i) g[i] = 0.f;
    g[7] = -dp * pf * (1.f - pf);
    g[8] = dp * pf * (1.f - pf);
    p.fit_prob[b] = pf;
  }
  red[0][b] = intra;
  red[1][b] = fit;
  __syncthreads();
  if (b == 0) {
    double si = 0.0, sf = 0.0;
    for (int i = 0; i < p.B; ++i) { si += red[0][i]; sf += red[1][i]; }
    const double fl = sf / (double)p.B;
    p.terms[0] = (float)si;
    p.terms[1] = (float)fl;
    p.loss[0] = (float)((double)p.strong_loss[0] + (double)p.w_weak * si + (double)p.w_fit * fl);
  }
}

// Backward of tf_convert_box_params_from_anchor_to_reg_format (tf_util.py:1017-1031) for the refined heads:
//   centre = box[0:3] + stage1_center            -> dbox[0:3] += g_c,  dstage1 += g_c
//   dims   = max(anchor[k*] + srn[k*]*mean[k*], 1e-5)  -> dbox[srn k*] += g_d * mean[k*] * 1[dims > 1e-5]
//   theta  = bin[j*] + hrn[j*]*pi/NH             -> dbox[hrn j*] += g_t * pi/NH          (k*, j* = first arg-max of the scores)
__global__ __launch_bounds__(1024) void k_anchor_reg_bwd(const t3d_anchor_reg_bwd_args p) {
  const int b = threadIdx.x;
  if (b >= p.B) return;
  const float* o = p.box + (size_t)b * p.ld_box;
  float* g = p.dbox + (size_t)b * 67;
  int js = 0, ks = 0;
  for (int i = 1; i < 12; ++i) if (o[3 + i] > o[3 + js]) js = i;
  for (int i = 1; i < 10; ++i) if (o[27 + i] > o[27 + ks]) ks = i;
  float gc[3] = {0.f, 0.f, 0.f}, gd[3] = {0.f, 0.f, 0.f}, gt = 0.f;
  if (p.dbox7) {
    const float* q = p.dbox7 + (size_t)b * 7;
    for (int d = 0; d < 3; ++d) { gc[d] = q[d]; gd[d] = q[3 + d]; }
    gt = q[6];
  }
  if (p.d_dims) for (int d = 0; d < 3; ++d) gd[d] += p.d_dims[b * 3 + d];
  for (int d = 0; d < 3; ++d) {
    g[d] += gc[d];
    p.dstage1[b * 3 + d] += gc[d];
    const float raw = kMeanG[ks][d] + o[37 + 3 * ks + d] * kMeanG[ks][d];
    if (raw > 1e-5f) g[37 + 3 * ks + d] += gd[d] * kMeanG[ks][d];
  }
  g[15 + js] += gt * (3.14159265358979323846f / 12.0f);
}

// one refinement step of the inference loop (test_semisup.py:101-134): box <- box - w * delta(box, pc)
__global__ __launch_bounds__(256) void k_box_refine_step(const t3d_box_refine_step_args p) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= p.B) return;
  const float* o = p.out9 + (size_t)b * 9;
  const float mx = fmaxf(o[7], o[8]);
  const float e0 = expf(o[7] - mx), e1 = expf(o[8] - mx);
  const float pfit = e1 / (e0 + e1);
  const float w = p.weigh_by_conf == 0 ? 1.f : (p.weigh_by_conf == 1 ? 1.f - pfit : (1.f - pfit) * (1.f - pfit));
  if (p.fit_prob) p.fit_prob[b] = pfit;
  float* tot = p.total + (size_t)b * 7;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const float dc = o[i] * w, ds = o[3 + i] * w;
    p.center_out[b * 3 + i] = p.center_in[b * 3 + i] - dc;
    p.dims_out[b * 3 + i] = p.dims_in[b * 3 + i] - ds;
    tot[i] = (p.first ? 0.f : tot[i]) + dc;
    tot[3 + i] = (p.first ? 0.f : tot[3 + i]) + ds;
  }
  const float da = o[6] * w;
  p.theta_out[b] = p.theta_in[b] - da;
  tot[6] = (p.first ? 0.f : tot[6]) + da;
}

__global__ __launch_bounds__(256) void k_box_refine_step_bwd(const t3d_box_refine_step_bwd_args p) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= p.B) return;
  float tot[7];
#pragma unroll
  for (int k = 0; k < 7; ++k) tot[k] = p.dbox_rep[b * 7 + k] + (p.carry ? p.carry[b * 7 + k] : 0.f);
#pragma unroll
  for (int k = 0; k < 7; ++k) p.tot_out[b * 7 + k] = tot[k];
  if (!p.out9) return;
  const float* o = p.out9 + (size_t)b * 9;
  const float mx = fmaxf(o[7], o[8]);
  const float e0 = expf(o[7] - mx), e1 = expf(o[8] - mx);
  const float pfit = e1 / (e0 + e1), q = 1.f - pfit;
  const int n = p.weigh_by_conf;
  const float w = n == 0 ? 1.f : (n == 1 ? q : q * q);
  float dot = 0.f;                                       // d L / d w = sum_k tot_k * (-out9_k)
#pragma unroll
  for (int k = 0; k < 7; ++k) {
    p.dout9[(size_t)b * 9 + k] = -w * tot[k];
    dot -= tot[k] * o[k];
  }
  float t = 0.f;
  if (p.grad_via_conf && n > 0) {                        // w = q^n, d w / d p = -n q^(n-1); d p / d(l0, l1) = p q (-1, +1)
    const float dw_dp = n == 1 ? -1.f : -2.f * q;
    t = dot * dw_dp * pfit * q;
  }
  p.dout9[(size_t)b * 9 + 7] = -t;
  p.dout9[(size_t)b * 9 + 8] = t;
}

}  // namespace

extern "C" int t3d_box_refine_step_bwd(const t3d_box_refine_step_bwd_args* a, t3d_stream_t stream) {
  if (!a || !a->dbox_rep || !a->tot_out || (a->out9 && !a->dout9)) return T3D_ERR_ARG;
  if (a->B <= 0 || a->weigh_by_conf < 0 || a->weigh_by_conf > 2) return T3D_ERR_SHAPE;
  T3D_LAUNCH(k_box_refine_step_bwd, dim3((a->B + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream), *a);
  T3D_CHECK_LAUNCH();
  return T3D_OK;
}

extern "C" int t3d_box_refine_step(const t3d_box_refine_step_args* a, t3d_stream_t stream) {
  if (!a || !a->out9 || !a->center_in || !a->dims_in || !a->theta_in || !a->center_out || !a->dims_out || !a->theta_out ||
      !a->total)
    return T3D_ERR_ARG;
  if (a->B <= 0) return T3D_ERR_SHAPE;
  T3D_LAUNCH(k_box_refine_step, dim3((a->B + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream), *a);
  T3D_CHECK_LAUNCH();
  return T3D_OK;
}

extern "C" int t3d_boxpc_rep(const t3d_boxpc_rep_args* a, t3d_stream_t stream) {
  T3D_ABI_TAKE(boxpc_rep_args, a);
  if (a && a->struct_size != sizeof(*a)) return T3D_ERR_ABI;
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

extern "C" int t3d_pointmlp_dgrad_narrow(const t3d_pointmlp_dgrad_narrow_args* a, t3d_stream_t stream) {
  if (!a || !a->dy.dz || !a->dy.y || !a->dy.coef || !a->w || !a->out) return T3D_ERR_ARG;
  if (a->M <= 0 || a->M % 64 || a->N % 128 || a->kn <= 0 || a->kn > 8 || a->ld_out < a->kn) return T3D_ERR_SHAPE;
  T3D_LAUNCH(k_dgrad_narrow, dim3(a->M / 64), dim3(256), 0, static_cast<hipStream_t>(stream), *a);
  T3D_CHECK_LAUNCH();
  return T3D_OK;
}

extern "C" int t3d_semi_final_loss(const t3d_semi_final_loss_args* a, t3d_stream_t stream) {
  if (!a || !a->strong_loss || !a->reg_dims || !a->one_hot || !a->is_data_2D || !a->out9 || !a->d_dims || !a->dout9 ||
      !a->fit_prob || !a->terms || !a->loss)
    return T3D_ERR_ARG;
  if (a->B <= 0 || a->B > 1024) return T3D_ERR_SHAPE;
  T3D_LAUNCH(k_semi_final_loss, dim3(1), dim3(1024), 0, static_cast<hipStream_t>(stream), *a);
  T3D_CHECK_LAUNCH();
  return T3D_OK;
}

extern "C" int t3d_anchor_reg_bwd(const t3d_anchor_reg_bwd_args* a, t3d_stream_t stream) {
  if (!a || !a->box || !a->dbox || !a->dstage1) return T3D_ERR_ARG;
  if (a->B <= 0 || a->B > 1024) return T3D_ERR_SHAPE;
  T3D_LAUNCH(k_anchor_reg_bwd, dim3(1), dim3(1024), 0, static_cast<hipStream_t>(stream), *a);
  T3D_CHECK_LAUNCH();
  return T3D_OK;
}

// ---- K13: 3-D IoU (t3d.h) -----------------------------------------------------------------------------------------------
namespace {
__global__ __launch_bounds__(256) void k_box3d_iou(const t3d_box3d_iou_args p) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= p.n) return;
  float c1[3], s1[3], c2[3], s2[3];
  for (int d = 0; d < 3; ++d) {
    c1[d] = p.center1[i * 3 + d]; s1[d] = p.size1[i * 3 + d];
    c2[d] = p.center2[i * 3 + d]; s2[d] = p.size2[i * 3 + d];
  }
  float i2;
  p.iou3d[i] = boxgeom::box3d_iou_params(c1, s1, p.heading1[i], c2, s2, p.heading2[i], &i2);
  if (p.iou2d) p.iou2d[i] = i2;
}
__global__ __launch_bounds__(256) void k_box3d_iou_corners(const t3d_box3d_iou_corners_args p) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= p.n) return;
  float k1[24], k2[24];
  for (int d = 0; d < 24; ++d) { k1[d] = p.corners1[(size_t)i * 24 + d]; k2[d] = p.corners2[(size_t)i * 24 + d]; }
  float i2;
  p.iou3d[i] = boxgeom::box3d_iou_corners(k1, k2, &i2);
  if (p.iou2d) p.iou2d[i] = i2;
}
}  // namespace

extern "C" int t3d_box3d_iou(const t3d_box3d_iou_args* a, t3d_stream_t stream) {
  if (!a || !a->center1 || !a->size1 || !a->heading1 || !a->center2 || !a->size2 || !a->heading2 || !a->iou3d) return T3D_ERR_ARG;
  if (a->n <= 0) return T3D_ERR_SHAPE;
  T3D_LAUNCH(k_box3d_iou, dim3((a->n + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream), *a);
  T3D_CHECK_LAUNCH();
  return T3D_OK;
}

extern "C" int t3d_box3d_iou_corners(const t3d_box3d_iou_corners_args* a, t3d_stream_t stream) {
  if (!a || !a->corners1 || !a->corners2 || !a->iou3d) return T3D_ERR_ARG;
  if (a->n <= 0) return T3D_ERR_SHAPE;
  T3D_LAUNCH(k_box3d_iou_corners, dim3((a->n + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream), *a);
  T3D_CHECK_LAUNCH();
  return T3D_OK;
}
