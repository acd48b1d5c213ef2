// 3-D IoU of two upright boxes (device code shared by the loss kernel, the stand-alone IoU entries and the Box-PC sampler).
//
// Replaces box_util.box3d_iou as called from roi_seg_box3d_dataset.py:103-140 (compute_box3d_iou, the per-step tf.py_func of
// semisup_v1_sunrgbd.py:236-246), box_pc_fit_dataset.py:38-42 (perturb_box_to_diff_ious) and eval_det.py:60-66.  `box_util`
// itself is not in the reference tree (it is the Frustum-PointNets train/box_util.py the reference's sys.path points at,
// unpinned); its published algorithm: ground-plane (x,z) rectangles from corners 3,2,1,0 of get_3d_box, polygon area by the
// shoelace formula, intersection polygon by Sutherland-Hodgman clipping (area of its convex hull), iou_2d = inter / (a1 + a2 -
// inter); height overlap max(0, min(ymax) - max(ymin)); iou_3d = inter*height / (vol1 + vol2 - inter*height).
//
// Here the intersection area is the boundary integral  1/2 * sum cross(a, b)  over the pieces of each rectangle's edges that lie
// inside the other rectangle (Cyrus-Beck clip of a segment against four half-planes): no vertex lists, no data-dependent
// indexing, every lane runs the same 8 x 4 clip steps.  Edges of P are kept where they touch Q's boundary (to within 1e-5 of
// the edge length), edges of Q are not, so coincident edges (a box against itself) are counted once.
#pragma once
#include <hip/hip_runtime.h>

namespace boxgeom {

struct Quad { float x[4], z[4]; };      // counter-clockwise in the (x, z) plane

// Ground rectangle of get_3d_box(size=(l,w,h), heading, center) (roi_seg_box3d_dataset.py:86-101): corners
// (+-l/2, +-w/2) rotated by roty(heading), listed counter-clockwise for cross(a,b) = ax*bz - az*bx.
__device__ __forceinline__ Quad ground_rect(float cx, float cz, float l, float w, float heading) {
  const float c = cosf(heading), s = sinf(heading);
  const float hl = 0.5f * fabsf(l), hw = 0.5f * fabsf(w);
  const float lx[4] = {hl, -hl, -hl, hl}, lz[4] = {hw, hw, -hw, -hw};
  Quad q;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    q.x[i] = c * lx[i] + s * lz[i] + cx;       // roty: x' = c x + s z,  z' = -s x + c z
    q.z[i] = -s * lx[i] + c * lz[i] + cz;
  }
  return q;
}

__device__ __forceinline__ float quad_area2(const Quad& q) {     // twice the signed area
  float a = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) { const int j = (i + 1) & 3; a += q.x[i] * q.z[j] - q.z[i] * q.x[j]; }
  return a;
}

__device__ __forceinline__ void make_ccw(Quad& q) {
  if (quad_area2(q) < 0.f) {
    float t = q.x[1]; q.x[1] = q.x[3]; q.x[3] = t;
    t = q.z[1]; q.z[1] = q.z[3]; q.z[3] = t;
  }
}

// 1/2 sum of cross(a', b') over the parts [a', b'] of P's edges inside Q.  CLOSED: points on Q's boundary count as inside.
template <bool CLOSED>
__device__ __forceinline__ float boundary_inside(const Quad& P, const Quad& Q) {
  float acc = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int j = (i + 1) & 3;
    const float ax = P.x[i], az = P.z[i], bx = P.x[j], bz = P.z[j];
    float t0 = 0.f, t1 = 1.f;
    bool alive = true;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int f = (e + 1) & 3;
      const float ex = Q.x[f] - Q.x[e], ez = Q.z[f] - Q.z[e];
      // signed distance (times |edge|) to the left of Q's edge e -> f: inside a CCW polygon is >= 0
      const float da = ex * (az - Q.z[e]) - ez * (ax - Q.x[e]);
      const float db = ex * (bz - Q.z[e]) - ez * (bx - Q.x[e]);
      // a point within `tol` of the edge line is ON it: inside for the closed test, outside for the open one, so that edges
      // that coincide up to rounding (a box against its own half turn, a prediction equal to its label) count exactly once
      const float tol = 1e-5f * (ex * ex + ez * ez) + 1e-12f;
      const bool ina = CLOSED ? da >= -tol : da > tol, inb = CLOSED ? db >= -tol : db > tol;
      const float t = fminf(fmaxf(da / (da - db), 0.f), 1.f);   // crossing parameter (used only when the sides differ)
      // a segment lying ON the edge line bounds the intersection only if both interiors are on the same side of it, i.e. the
      // two (counter-clockwise) edges point the same way; opposite directions mean the rectangles merely touch there
      if (CLOSED) {
        const bool on_line = fabsf(da) <= tol && fabsf(db) <= tol;
        alive = alive && !(on_line && (bx - ax) * ex + (bz - az) * ez <= 0.f);
      }
      alive = alive && (ina || inb);
      t0 = (!ina && inb) ? fmaxf(t0, t) : t0;
      t1 = (ina && !inb) ? fminf(t1, t) : t1;
    }
    alive = alive && t1 > t0;
    const float px = ax + t0 * (bx - ax), pz = az + t0 * (bz - az);
    const float qx = ax + t1 * (bx - ax), qz = az + t1 * (bz - az);
    acc += alive ? 0.5f * (px * qz - pz * qx) : 0.f;
  }
  return acc;
}

__device__ __forceinline__ float quad_intersection_area(Quad P, Quad Q) {
  make_ccw(P);
  make_ccw(Q);
  // shift both to P's first corner: the cross products then involve small numbers (boxes sit metres from the origin)
  const float ox = P.x[0], oz = P.z[0];
#pragma unroll
  for (int i = 0; i < 4; ++i) { P.x[i] -= ox; P.z[i] -= oz; Q.x[i] -= ox; Q.z[i] -= oz; }
  const float a = boundary_inside<true>(P, Q) + boundary_inside<false>(Q, P);
  return fmaxf(a, 0.f);
}

// The common tail of box3d_iou: areas, height overlap, volumes.
__device__ __forceinline__ float iou_from_parts(float inter_area, float area1, float area2, float ymax1, float ymin1, float ymax2,
                                                float ymin2, float vol1, float vol2, float* iou2d) {
  *iou2d = inter_area / (area1 + area2 - inter_area);
  const float inter_vol = inter_area * fmaxf(0.f, fminf(ymax1, ymax2) - fmaxf(ymin1, ymin2));
  return inter_vol / (vol1 + vol2 - inter_vol);
}

// Boxes in parameter form: centre (x,y,z), size (l,w,h), heading about the y axis.
__device__ __forceinline__ float box3d_iou_params(const float* c1, const float* s1, float h1, const float* c2, const float* s2, float h2,
                                                  float* iou2d) {
  const Quad P = ground_rect(c1[0], c1[2], s1[0], s1[1], h1), Q = ground_rect(c2[0], c2[2], s2[0], s2[1], h2);
  const float a1 = fabsf(s1[0] * s1[1]), a2 = fabsf(s2[0] * s2[1]);
  // signed half heights: a negative h puts get_3d_box's "top" face below its "bottom" face, the height overlap of box3d_iou is
  // then never positive and iou3d = 0 -- kept, because an untrained size head does produce negative sizes.  Negative l / w only
  // mirror the ground rectangle (same area, same clip), hence the magnitudes above.
  const float hh1 = 0.5f * s1[2], hh2 = 0.5f * s2[2];
  return iou_from_parts(quad_intersection_area(P, Q), a1, a2, c1[1] + hh1, c1[1] - hh1, c2[1] + hh2, c2[1] - hh2, a1 * fabsf(s1[2]),
                        a2 * fabsf(s2[2]), iou2d);
}

// Boxes as 8 corners in get_3d_box order (box3d_iou's own argument form): ground rectangle = corners 3,2,1,0 (x,z);
// ymax = corner 0, ymin = corner 4; volume from the three edge lengths at corner 0 (box3d_vol).
__device__ __forceinline__ float box3d_iou_corners(const float* k1, const float* k2, float* iou2d) {
  Quad P, Q;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    P.x[i] = k1[(3 - i) * 3 + 0]; P.z[i] = k1[(3 - i) * 3 + 2];
    Q.x[i] = k2[(3 - i) * 3 + 0]; Q.z[i] = k2[(3 - i) * 3 + 2];
  }
  auto edge = [](const float* k, int a, int b) {
    const float dx = k[a * 3] - k[b * 3], dy = k[a * 3 + 1] - k[b * 3 + 1], dz = k[a * 3 + 2] - k[b * 3 + 2];
    return sqrtf(dx * dx + dy * dy + dz * dz);
  };
  const float a1 = 0.5f * fabsf(quad_area2(P)), a2 = 0.5f * fabsf(quad_area2(Q));
  const float v1 = edge(k1, 0, 1) * edge(k1, 1, 2) * edge(k1, 0, 4), v2 = edge(k2, 0, 1) * edge(k2, 1, 2) * edge(k2, 0, 4);
  return iou_from_parts(quad_intersection_area(P, Q), a1, a2, k1[1], k1[4 * 3 + 1], k2[1], k2[4 * 3 + 1], v1, v2, iou2d);
}

}  // namespace boxgeom
