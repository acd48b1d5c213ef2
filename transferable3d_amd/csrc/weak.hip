// Weak box losses of get_semi_loss_backbone (semisup_v1_sunrgbd.py:270-311): the reprojection loss (models/weak_losses.py:69-229) and
// the surface loss (231-259) of the predicted box S_pred_box_reg = (centre, dims, theta), forward and backward.
//
// Both are small geometric programs per frustum (8 box corners through two rotations and a pinhole projection, min / max or a
// stop-gradient soft-max over them, clips, a Huber band loss) or per point (six ray / plane distances, their minimum), full of
// data-dependent branches.  Their gradients with respect to the 7 box parameters are taken in FORWARD mode: every intermediate is
// a value plus 7 partial derivatives (D7), so the backward pass is the forward program itself and follows its branches by
// construction (an arg-min picks the derivative of the element it picked, a clip zeroes it, ...).  Cost: 8x the arithmetic of a
// few hundred flops per point -- noise next to the GEMMs.  Sums over points are tile partials combined in a fixed order.
#include "common.h"

namespace {

constexpr int ND = 7;      // centre x, y, z; dims l, w, h; theta

struct D7 {
  float v;
  float g[ND];
};
__device__ __forceinline__ D7 dconst(float v) {
  D7 r; r.v = v;
#pragma unroll
  for (int i = 0; i < ND; ++i) r.g[i] = 0.f;
  return r;
}
__device__ __forceinline__ D7 dvar(float v, int i, float seed) { D7 r = dconst(v); r.g[i] = seed; return r; }
__device__ __forceinline__ D7 operator+(const D7& a, const D7& b) {
  D7 r; r.v = a.v + b.v;
#pragma unroll
  for (int i = 0; i < ND; ++i) r.g[i] = a.g[i] + b.g[i];
  return r;
}
__device__ __forceinline__ D7 operator-(const D7& a, const D7& b) {
  D7 r; r.v = a.v - b.v;
#pragma unroll
  for (int i = 0; i < ND; ++i) r.g[i] = a.g[i] - b.g[i];
  return r;
}
__device__ __forceinline__ D7 operator-(const D7& a) {
  D7 r; r.v = -a.v;
#pragma unroll
  for (int i = 0; i < ND; ++i) r.g[i] = -a.g[i];
  return r;
}
__device__ __forceinline__ D7 operator*(const D7& a, const D7& b) {
  D7 r; r.v = a.v * b.v;
#pragma unroll
  for (int i = 0; i < ND; ++i) r.g[i] = a.g[i] * b.v + a.v * b.g[i];
  return r;
}
__device__ __forceinline__ D7 operator*(const D7& a, float s) {
  D7 r; r.v = a.v * s;
#pragma unroll
  for (int i = 0; i < ND; ++i) r.g[i] = a.g[i] * s;
  return r;
}
__device__ __forceinline__ D7 operator*(float s, const D7& a) { return a * s; }
__device__ __forceinline__ D7 operator+(const D7& a, float s) { D7 r = a; r.v += s; return r; }
__device__ __forceinline__ D7 operator-(const D7& a, float s) { D7 r = a; r.v -= s; return r; }
__device__ __forceinline__ D7 operator/(const D7& a, const D7& b) {
  const float inv = 1.0f / b.v;
  D7 r; r.v = a.v * inv;
#pragma unroll
  for (int i = 0; i < ND; ++i) r.g[i] = (a.g[i] - r.v * b.g[i]) * inv;
  return r;
}
__device__ __forceinline__ D7 dsin(const D7& a) { float s, c; sincosf(a.v, &s, &c); D7 r; r.v = s;
#pragma unroll
  for (int i = 0; i < ND; ++i) r.g[i] = c * a.g[i];
  return r; }
__device__ __forceinline__ D7 dcos(const D7& a) { float s, c; sincosf(a.v, &s, &c); D7 r; r.v = c;
#pragma unroll
  for (int i = 0; i < ND; ++i) r.g[i] = -s * a.g[i];
  return r; }
__device__ __forceinline__ D7 dsqrt(const D7& a) {
  D7 r; r.v = sqrtf(a.v);
  const float k = r.v > 0.f ? 0.5f / r.v : 0.f;      // |r| at r = 0: subgradient 0 (TensorFlow's tf.norm yields NaN there; measure zero)
#pragma unroll
  for (int i = 0; i < ND; ++i) r.g[i] = a.g[i] * k;
  return r;
}
__device__ __forceinline__ D7 dabs(const D7& a) { return a.v < 0.f ? -a : (a.v > 0.f ? a : dconst(0.f)); }      // tf.abs: sign(0) = 0
__device__ __forceinline__ D7 dsel(bool c, const D7& a, const D7& b) { return c ? a : b; }
__device__ __forceinline__ D7 dmin(const D7& a, const D7& b) { return b.v < a.v ? b : a; }      // first of equals (ties: measure zero)
__device__ __forceinline__ D7 dmax(const D7& a, const D7& b) { return b.v > a.v ? b : a; }
__device__ __forceinline__ D7 drelu(const D7& a) { return a.v > 0.f ? a : dconst(0.f); }

// tf.losses.huber_loss(labels = target, predictions = val, delta = 1) / mean_squared_error, element-wise
__device__ __forceinline__ D7 elem_loss(const D7& val, float target, int mse) {
  const D7 e = val - target;
  if (mse) return e * e;
  const D7 a = dabs(e);
  if (a.v <= 1.f) return 0.5f * (a * a);
  return a - 0.5f;
}
// weak_losses.py:15-37
__device__ __forceinline__ D7 dev_from_range(const D7& val, float lower_b, float upper_b, int mse) {
  D7 r = dconst(0.f);
  if (val.v < lower_b) r = r + elem_loss(val, lower_b, mse);
  if (val.v > upper_b) r = r + elem_loss(val, upper_b, mse);
  return r;
}
__device__ __forceinline__ D7 dmin1000(const D7& a) { return a.v > 1000.f ? dconst(1000.f) : a; }

struct Box7 { D7 c[3], d[3], th; };
__device__ __forceinline__ Box7 load_box(const t3d_weak_loss_args& p, int b, const int32_t (&train)[3], float dims_scale) {
  Box7 x;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    x.c[i] = dvar(p.center[b * 3 + i], i, train[0] ? 1.f : 0.f);
    x.d[i] = dvar(p.reg_dims[b * 3 + i] * dims_scale, 3 + i, train[1] ? dims_scale : 0.f);
  }
  x.th = dvar(p.reg_theta[b], 6, train[2] ? 1.f : 0.f);
  return x;
}

// ---- surface loss: one thread per point, one block per 128-point tile ------------------------------------------------------
// tf_distance_to_box_surfaces / tf_distance_to_closest_3D_box_surface (tf_util.py:610-709): the MINIMUM OVER THE SIX RAW distances
// | |r| - |r| * ((p0 - l0).n) / (r.n + 1e-5) | (the "cleaned" distances of lines 694-704 are computed and dropped by the reference).
__global__ __launch_bounds__(128) void k_weak_surface(const t3d_weak_loss_args p) {
  __shared__ float red[2][8];
  const int tid = threadIdx.x, tile = blockIdx.x, b = blockIdx.y;
  const int n = tile * 128 + tid;
  const size_t m = (size_t)b * p.N + n;
  const Box7 x = load_box(p, b, p.train_box_surface, p.surface_scale_dims);
  const D7 st = dsin(x.th), ct = dcos(x.th);
  D7 r[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) r[i] = dconst(p.pc[m * p.ld_pc + i]) - x.c[i];
  // inward normals of the six faces, rotated: R = [[c,0,s],[0,1,0],[-s,0,c]] applied to (-+1,0,0), (0,-+1,0), (0,0,-+1)
  const D7 perp[6] = {-(r[0] * ct) + r[2] * st, r[0] * ct - r[2] * st, -r[1], r[1], -(r[0] * st) - r[2] * ct, r[0] * st + r[2] * ct};
  const D7 q[3] = {x.d[0] * -0.5f, x.d[2] * -0.5f, x.d[1] * -0.5f};      // (p0 - l0).n of the x, y, z face pairs: -l/2, -h/2, -w/2
  const D7 rn = dsqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
  const bool at_center = fabsf(r[0].v) + fabsf(r[1].v) + fabsf(r[2].v) == 0.f;
  D7 best = dconst(0.f);
#pragma unroll
  for (int s = 0; s < 6; ++s) {
    D7 dcs = rn * (q[s >> 1] / (perp[s] + 1e-5f));
    if (at_center) dcs = -q[s >> 1];                                      // tf_util.py:655-660: (l/2, l/2, h/2, h/2, w/2, w/2)
    const D7 dist = dabs(rn - dcs);
    best = s == 0 ? dist : dmin(best, dist);
  }
  const D7 e = drelu(best - p.surface_margin);
  // soft_mask = softmax(logits)[:, 1]
  const float q0 = p.logits[m * 2], q1 = p.logits[m * 2 + 1];
  const float soft = 1.0f / (1.0f + expf(q0 - q1));
  const float invN = 1.0f / (float)p.N;
  // scale of d total_loss / d surface_loss[b]: is_data_2D * SEMI_MULTIPLIER * w_surface / B
  const float sb = (p.is_data_2D ? (float)p.is_data_2D[b] : 1.f) * p.multiplier * p.w_surface / (float)p.B;
  if (p.dsoft) p.dsoft[m] = sb * e.v * invN;
  float val[8];
  val[0] = e.v * soft;
#pragma unroll
  for (int i = 0; i < ND; ++i) val[1 + i] = e.g[i] * soft;
#pragma unroll
  for (int i = 0; i < 8; ++i) val[i] = wave_sum(val[i]);
  if ((tid & 63) == 0) {
#pragma unroll
    for (int i = 0; i < 8; ++i) red[tid >> 6][i] = val[i];
  }
  __syncthreads();
  if (tid < 8) p.surf_part[((size_t)b * gridDim.x + tile) * 8 + tid] = (red[0][tid] + red[1][tid]) * invN;
}

// ---- reprojection loss + combination: one thread per frustum ------------------------------------------------------------------
__device__ __forceinline__ void project_corners(const t3d_weak_loss_args& p, int b, const Box7& x, D7 (&u)[8], D7 (&v)[8]) {
  // tf_rot_box_params (tf_util.py:1045-1062): rotate the box back by the frustum angle
  const float A = p.rot_frust[b];
  float sA, cA;
  sincosf(A, &sA, &cA);
  const D7 cx = cA * x.c[0] + sA * x.c[2], cy = x.c[1], cz = -sA * x.c[0] + cA * x.c[2];
  const D7 th = x.th + A;
  // tf_create_3D_box_by_vertices_multi (tf_util.py:841-891): upright depth corners, then X, -Z, Y
  const D7 c = dcos(-th), s = dsin(-th);
  const float* Rt = p.Rtilt + b * 9;
  const float* K = p.K + b * 9;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float sx = (j == 0 || j == 3 || j == 4 || j == 7) ? -0.5f : 0.5f;
    const float sy = (j & 2) ? -0.5f : 0.5f;
    const float sz = j < 4 ? 0.5f : -0.5f;
    const D7 xc = x.d[0] * sx, yc = x.d[1] * sy, zc = x.d[2] * sz;
    const D7 x3 = c * xc - s * yc, y3 = s * xc + c * yc;
    const D7 P[3] = {x3 + cx, -zc + cy, y3 + cz};                     // upright camera coordinates
    const D7 pd[3] = {P[0], P[2], -P[1]};                             // flip_axis_to_depth (tf_util.py:826-830)
    D7 t[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) t[i] = Rt[0 * 3 + i] * pd[0] + Rt[1 * 3 + i] * pd[1] + Rt[2 * 3 + i] * pd[2];      // Rtilt^T pd
    const D7 cam[3] = {t[0], -t[2], t[1]};                            // flip_axis_to_camera
    D7 uv[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) uv[i] = K[i * 3 + 0] * cam[0] + K[i * 3 + 1] * cam[1] + K[i * 3 + 2] * cam[2];
    u[j] = uv[0] / uv[2];
    v[j] = uv[1] / uv[2];
  }
}

// soft extreme of tf_get_2D_softmax_bbox_of_points (tf_util.py:379-414): sum_j x_j * stop_gradient(softmax(closeness_j / extent * scale))
__device__ __forceinline__ D7 soft_extreme(const D7 (&x)[8], float far_bound, float sign, float extent, float scale) {
  float z[8], mx = -INFINITY;
#pragma unroll
  for (int j = 0; j < 8; ++j) { z[j] = sign * (far_bound - x[j].v) / extent * scale; mx = fmaxf(mx, z[j]); }
  float den = 0.f;
#pragma unroll
  for (int j = 0; j < 8; ++j) { z[j] = expf(z[j] - mx); den += z[j]; }
  D7 r = dconst(0.f);
#pragma unroll
  for (int j = 0; j < 8; ++j) r = r + x[j] * (z[j] / den);
  return r;
}

__global__ __launch_bounds__(256) void k_weak_finish(const t3d_weak_loss_args p, const int tiles) {
  __shared__ float tot[256];
  __shared__ float iv[256];
  __shared__ float ccount[10];
  const int b = threadIdx.x;
  float contrib = 0.f, ivc = 0.f;
  // inactive-volume loss (weak_losses.py:39-67, stage c): class of a frustum = arg-max of its one-hot vector
  int cls = 0, T = 0;
  const bool inact = p.w_inactive != 0.f && p.one_hot != nullptr;
  if (inact) {
    if (b < p.B) { const float* oh = p.one_hot + (size_t)b * 10; for (int i = 1; i < 10; ++i) if (oh[i] > oh[cls]) cls = i; }
    if (b < 10) {
      float n = 0.f;
      for (int i = 0; i < p.B; ++i) {
        const float* oh = p.one_hot + (size_t)i * 10;
        int ci = 0;
        for (int k = 1; k < 10; ++k) if (oh[k] > oh[ci]) ci = k;
        n += ci == b ? 1.f : 0.f;
      }
      ccount[b] = n;
    }
    for (int i = 0; i < 10; ++i) T += p.inactive_train[i] ? 1 : 0;
    __syncthreads();
  }
  if (b < p.B) {
    // surface: tile partials in ascending order
    float sv[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) sv[i] = 0.f;
    for (int t = 0; t < tiles; ++t)
#pragma unroll
      for (int i = 0; i < 8; ++i) sv[i] += p.surf_part[((size_t)b * tiles + t) * 8 + i];
    // reprojection
    D7 rl = dconst(0.f);
    if (p.Rtilt != nullptr) {      // evaluated whenever its inputs are given (the reference logs it at any weight)
      const Box7 x = load_box(p, b, p.train_box_reproj, 1.f);
      D7 u[8], v[8];
      project_corners(p, b, x, u, v);
      D7 pb[4];
      if (p.use_softmax_proj) {
        float lb = u[0].v, rb = u[0].v, tb = v[0].v, bb = v[0].v;
#pragma unroll
        for (int j = 1; j < 8; ++j) { lb = fminf(lb, u[j].v); rb = fmaxf(rb, u[j].v); tb = fminf(tb, v[j].v); bb = fmaxf(bb, v[j].v); }
        const float width = fabsf(rb - lb), height = fabsf(bb - tb);
        pb[0] = soft_extreme(u, rb, 1.f, width, p.softmax_scale);      // left:   closeness = right_bound - u
        pb[1] = soft_extreme(v, bb, 1.f, height, p.softmax_scale);     // top
        pb[2] = soft_extreme(u, lb, -1.f, width, p.softmax_scale);     // right:  closeness = u - left_bound
        pb[3] = soft_extreme(v, tb, -1.f, height, p.softmax_scale);    // bottom
      } else {
        pb[0] = u[0]; pb[1] = v[0]; pb[2] = u[0]; pb[3] = v[0];
#pragma unroll
        for (int j = 1; j < 8; ++j) { pb[0] = dmin(pb[0], u[j]); pb[1] = dmin(pb[1], v[j]); pb[2] = dmax(pb[2], u[j]); pb[3] = dmax(pb[3], v[j]); }
      }
      // 2-D boxes: the label box clipped to the image, the dilated box (tf_util.py:486-514: height = top - bottom, as written) and its clip
      const float* bx = p.box2D + b * 4;
      const float rows = p.img_dim[b * 2], cols = p.img_dim[b * 2 + 1];
      const float small[4] = {fmaxf(0.f, bx[0]), fmaxf(0.f, bx[1]), fminf(cols, bx[2]), fminf(rows, bx[3])};
      const float cxm = (bx[0] + bx[2]) * 0.5f, cym = (bx[1] + bx[3]) * 0.5f;
      const float nw = p.dilate * (bx[2] - bx[0]), nh = p.dilate * (bx[1] - bx[3]);
      const float big[4] = {cxm - nw * 0.5f, cym + nh * 0.5f, cxm + nw * 0.5f, cym - nh * 0.5f};
      const float bigc[4] = {fmaxf(0.f, big[0]), fmaxf(0.f, big[1]), fminf(cols, big[2]), fminf(rows, big[3])};
      const int mse = p.loss_mse;
      if (p.clip_pred_box) {
        // the projected box clipped to the image too (gradient passes where it was not clipped)
        D7 pc4[4] = {drelu(pb[0]), drelu(pb[1]), pb[2].v < cols ? pb[2] : dconst(cols), pb[3].v < rows ? pb[3] : dconst(rows)};
        // tf.maximum(0, x) / tf.minimum(c, x): at equality TF splits the gradient; measure zero, the kept branch is taken here
        rl = dev_from_range(pc4[0], bigc[0], small[0], mse) + dev_from_range(pc4[1], bigc[1], small[1], mse) +
             dev_from_range(pc4[2], small[2], bigc[2], mse) + dev_from_range(pc4[3], small[3], bigc[3], mse);
      } else if (p.clip_lower_b_loss) {
        rl = dconst(0.f);
        if (big[0] == bigc[0]) rl = rl + dev_from_range(pb[0], bigc[0], small[0], mse);
        if (big[1] == bigc[1]) rl = rl + dev_from_range(pb[1], bigc[1], small[1], mse);
        if (big[2] == bigc[2]) rl = rl + dev_from_range(pb[2], small[2], bigc[2], mse);
        if (big[3] == bigc[3]) rl = rl + dev_from_range(pb[3], small[3], bigc[3], mse);
        rl = dmin1000(rl);
      } else {
        rl = dconst(0.f);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const bool less_inside = i < 2;      // left / top: no inner violation below the inner bound, no outer violation above the outer one
          D7 side = dconst(0.f);
          const bool inner_on = less_inside ? !(pb[i].v < small[i]) : !(pb[i].v > small[i]);
          const bool outer_on = (less_inside ? !(pb[i].v > bigc[i]) : !(pb[i].v < bigc[i])) && big[i] == bigc[i];
          if (inner_on) side = side + elem_loss(pb[i], small[i], mse);
          if (outer_on) side = side + elem_loss(pb[i], bigc[i], mse);
          rl = rl + dmin1000(side);
        }
      }
    }
    const float is2d = p.is_data_2D ? (float)p.is_data_2D[b] : 1.f;      // NULL: every sample (stage c without ..._ONLY_ON_2D_CLS)
    const float weak = p.w_reproj * rl.v + p.w_surface * sv[0];
    if (p.reproj) p.reproj[b] = rl.v;
    if (p.surface) p.surface[b] = sv[0];
    const float sb = is2d * p.multiplier / (float)p.B;
#pragma unroll
    for (int i = 0; i < ND; ++i) p.dbox7[b * 7 + i] = sb * (p.w_reproj * rl.g[i] + p.w_surface * sv[1 + i]);
    const float add = is2d * p.multiplier * weak;
    if (p.total_losses) p.total_losses[b] += add;
    contrib = add / (float)p.B;
    if (inact && T > 0 && p.inactive_train[cls]) {      // mean over the trained classes of the class mean of max(0, margin - l w h)
      const float l = p.reg_dims[b * 3], w = p.reg_dims[b * 3 + 1], h = p.reg_dims[b * 3 + 2];
      const float vol = l * w * h, viol = p.inactive_margins[cls] - vol;
      const float wgt = 1.0f / (ccount[cls] * (float)T);
      if (viol > 0.f) {
        ivc = viol * wgt;
        const float k = -p.multiplier * p.w_inactive * wgt;      // a scalar term of the loss: no is_data_2D factor, no 1/B
        p.dbox7[b * 7 + 3] += k * w * h;
        p.dbox7[b * 7 + 4] += k * l * h;
        p.dbox7[b * 7 + 5] += k * l * w;
      }
    }
  }
  tot[threadIdx.x] = contrib;
  iv[threadIdx.x] = ivc;
  __syncthreads();
  if (threadIdx.x == 0) {
    float s = 0.f, si = 0.f;
    for (int i = 0; i < p.B; ++i) { s += tot[i]; si += iv[i]; }      // ascending b: reproducible
    if (p.inactive) p.inactive[0] = si;
    p.loss[0] += s + p.multiplier * p.w_inactive * si;
  }
}

}  // namespace

extern "C" int t3d_weak_loss(const t3d_weak_loss_args* a, t3d_stream_t stream) {
  if (!a || !a->center || !a->reg_dims || !a->reg_theta || !a->dbox7 || !a->loss) return T3D_ERR_ARG;
  const bool surf = a->pc != nullptr, rep = a->Rtilt != nullptr;      // a loss is evaluated when its inputs are given, whatever its weight
  if (a->w_surface != 0.f && !surf) return T3D_ERR_ARG;
  if (a->w_reproj != 0.f && !rep) return T3D_ERR_ARG;
  if (surf && (!a->logits || !a->surf_part)) return T3D_ERR_ARG;
  if (a->w_inactive != 0.f && !a->one_hot) return T3D_ERR_ARG;
  if (rep && (!a->K || !a->rot_frust || !a->box2D || !a->img_dim)) return T3D_ERR_ARG;
  if (a->B <= 0 || a->B > 256 || a->N <= 0 || a->N % 128) return T3D_ERR_SHAPE;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int tiles = a->N / 128;
  if (surf) {
    T3D_LAUNCH(k_weak_surface, dim3(tiles, a->B), dim3(128), 0, s, *a);
    T3D_CHECK_LAUNCH();
  }
  T3D_LAUNCH(k_weak_finish, dim3(1), dim3(256), 0, s, *a, surf ? tiles : 0);
  T3D_CHECK_LAUNCH();
  return T3D_OK;
}
