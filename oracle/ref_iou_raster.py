"""TEST INFRASTRUCTURE ONLY -- a second opinion for the rotated-box IoU: rasterisation, no polygon clipping at all.

`box_util.box3d_iou` is the one numeric routine of the path with no pin whatsoever (the module is absent from the reference's
tree: roi_seg_box3d_dataset.py:15 imports it from Frustum PointNets' train/box_util.py), and both restatements in this repo -- the
oracle's Sutherland-Hodgman clip (ref_box.py) and the device's boundary integral (csrc/boxgeom_dev.h) -- are CLIPPING algorithms
written by the same author.  This file computes the same quantity by counting: the ground rectangle of box 1 is covered, in its own
frame, by an n x n grid of cells with one sample each (so box 1's own boundary is represented exactly), every sample is mapped into the
frame of box 2 and tested with two absolute-value comparisons; the height overlap is a 1-D interval intersection.  No edge, vertex or
orientation convention enters.  Box parametrisation as roi_seg_box3d_dataset.get_3d_box (86-101): size (l, w, h), heading about +y,
x = l/2 c + w/2 s ..., i.e. corners = roty(heading) . (+-l/2, +-h/2, +-w/2) + centre.

Accuracy: only cells cut by box 2's boundary are uncertain; with one stratified sample per cell their errors are independent, so
the error of the overlap fraction is ~ sqrt(4 n) / n^2 = 6e-5 at n = 1024 (tests pin it on closed forms first).
"""
import numpy as np


def ground_overlap_fraction(size1, heading1, center1, size2, heading2, center2, n=1024):
    """Fraction of box 1's ground rectangle (x-z plane) that lies inside box 2's (closed) ground rectangle."""
    l1, w1 = abs(float(size1[0])), abs(float(size1[1]))
    l2, w2 = abs(float(size2[0])), abs(float(size2[1]))
    # one sample per cell, at a pseudo-random place inside the cell (stratified Monte-Carlo, fixed seed): with samples at the cell
    # centres the errors of symmetric configurations add up coherently (a square against its 45-degree turn: 7e-4 at n = 1024)
    rng = np.random.RandomState(12345)
    k = np.arange(n, dtype=np.float64)
    a = ((k[:, None] + rng.uniform(size=(n, n))) / n - 0.5) * l1   # local coordinates of box 1: a along l, b along w
    b = ((k[None, :] + rng.uniform(size=(n, n))) / n - 0.5) * w1
    # roty(t) maps local (a, ., b) to world (x, z) = (c a + s b, -s a + c b)
    c1, s1 = np.cos(heading1), np.sin(heading1)
    x = c1 * a + s1 * b + center1[0]
    z = -s1 * a + c1 * b + center1[2]
    # world -> local of box 2: inverse rotation
    c2, s2 = np.cos(heading2), np.sin(heading2)
    dx, dz = x - center2[0], z - center2[2]
    a2 = c2 * dx - s2 * dz
    b2 = s2 * dx + c2 * dz
    inside = (np.abs(a2) <= 0.5 * l2) & (np.abs(b2) <= 0.5 * w2)
    return float(inside.mean())


def box3d_iou_raster(center1, size1, heading1, center2, size2, heading2, n=1024):
    """(iou_3d, iou_2d) of two boxes given as (centre (x, y, z), size (l, w, h), heading); the argument order of
    box_pc_fit_dataset.get_box3d_iou (38-42).  Sizes count by magnitude in the ground plane; a negative h inverts the height range
    (top below bottom), which the reference's min / max of the corner heights turns into an empty overlap."""
    center1, center2 = np.asarray(center1, np.float64), np.asarray(center2, np.float64)
    a1, a2 = abs(float(size1[0]) * float(size1[1])), abs(float(size2[0]) * float(size2[1]))
    inter = a1 * ground_overlap_fraction(size1, heading1, center1, size2, heading2, center2, n)
    iou2d = inter / (a1 + a2 - inter)
    top = min(center1[1] + 0.5 * size1[2], center2[1] + 0.5 * size2[2])      # corners 0-3 carry +h/2, corners 4-7 -h/2
    bot = max(center1[1] - 0.5 * size1[2], center2[1] - 0.5 * size2[2])
    inter_vol = inter * max(0.0, top - bot)
    v1, v2 = a1 * abs(float(size1[2])), a2 * abs(float(size2[2]))
    return inter_vol / (v1 + v2 - inter_vol), iou2d
