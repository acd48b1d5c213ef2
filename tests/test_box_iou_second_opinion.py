"""A second opinion for the rotated-box IoU (the one numeric routine with no pin at all: box_util is absent upstream,
roi_seg_box3d_dataset.py:15, 102-139): oracle/ref_iou_raster.py counts grid cells instead of clipping polygons.  It must agree with
the oracle's Sutherland-Hodgman restatement (ref_box.box3d_iou), with the specification of the device algorithm, and -- on the GPU
-- with t3d_box3d_iou itself, to 1e-3 on 200 random rotated pairs incl. touching and nested boxes."""
import ctypes as C
import math

import numpy as np
import pytest

from oracle import ref_box as RB
from oracle.ref_iou_raster import box3d_iou_raster


def pairs(seed=7, n=200):
    """200 pairs: 100 near pairs (high IoU, many edge crossings), 40 random, 20 nested, 20 touching along an edge, 20 disjoint."""
    r = np.random.RandomState(seed)
    c1 = r.normal(0, 1.0, size=(n, 3)) + np.array([0, 0, 3.0])
    s1 = r.uniform(0.4, 2.5, size=(n, 3))
    h1 = r.uniform(-np.pi, np.pi, size=n)
    c2 = r.normal(0, 1.0, size=(n, 3)) + np.array([0, 0, 3.0])
    s2 = r.uniform(0.4, 2.5, size=(n, 3))
    h2 = r.uniform(-np.pi, np.pi, size=n)
    k = np.arange(n)
    near = k < 100
    c2[near] = c1[near] + r.normal(0, 0.2, size=(100, 3))
    s2[near] = s1[near] * r.uniform(0.8, 1.25, size=(100, 3))
    h2[near] = h1[near] + r.uniform(-0.6, 0.6, size=100)
    nested = (k >= 140) & (k < 160)              # box 2 strictly inside box 1, turned
    small = np.minimum(s1[nested, 0], s1[nested, 1])[:, None]      # half diagonal + offset < half the shorter side: inside at any turn
    c2[nested] = c1[nested] + r.uniform(-0.05, 0.05, size=(20, 3)) * np.concatenate([small, s1[nested, 2:3], small], 1)
    s2[nested] = np.concatenate([small * r.uniform(0.2, 0.3, size=(20, 2)), s1[nested, 2:3] * r.uniform(0.2, 0.45, size=(20, 1))], 1)
    h2[nested] = h1[nested] + r.uniform(-np.pi, np.pi, size=20)
    touch = (k >= 160) & (k < 180)               # same heading, shifted by exactly one length along l: a shared face, no volume
    s2[touch] = s1[touch]
    h2[touch] = h1[touch]
    c2[touch] = c1[touch] + np.stack([np.cos(h1[touch]) * s1[touch, 0], np.zeros(20), -np.sin(h1[touch]) * s1[touch, 0]], 1)
    far = k >= 180
    c2[far] = c1[far] + np.array([8.0, 0, 0])
    return c1, s1, h1, c2, s2, h2


def test_raster_iou_closed_forms():
    unit = (1.0, 1.0, 1.0)
    i3, i2 = box3d_iou_raster((0, 0, 0), unit, 0.0, (0.5, 0, 0), unit, 0.0)                 # half overlap: 1/3
    assert abs(i2 - 1 / 3) < 2e-4 and abs(i3 - 1 / 3) < 2e-4
    a = 2 * (math.sqrt(2) - 1)                                                                # square vs its 45-degree turn: octagon
    i3, i2 = box3d_iou_raster((0, 0, 0), unit, 0.0, (0, 0, 0), unit, math.pi / 4)
    assert abs(i2 - a / (2 - a)) < 2e-4
    i3, i2 = box3d_iou_raster((1.2, 0.5, 5.1), (1, 1, 1), 0.3, (1, 0, 5), (4, 2, 2), 0.3)     # containment: volume ratio
    assert abs(i2 - 1 / 8) < 1e-9 and abs(i3 - 1 / 16) < 1e-9                                 # (box 1 entirely inside: every sample counts)
    i3, i2 = box3d_iou_raster((1.2, 0.5, 5.1), (1, 1, 1), 0.3, (1, 1.5, 5), (4, 2, 2), 0.3)   # heights overlap by 1/2
    assert abs(i3 - 0.5 / (16 + 1 - 0.5)) < 1e-9
    assert box3d_iou_raster((0, 0, 0), unit, 0.2, (3, 0, 0), unit, 1.0) == (0.0, 0.0)
    # a rotated rectangle pair with a hand-computed intersection: 2 x 1 rectangle and its quarter turn about the same centre -> unit square
    i3, i2 = box3d_iou_raster((0, 0, 0), (2, 1, 1), 0.0, (0, 0, 0), (2, 1, 1), math.pi / 2)
    assert abs(i2 - 1.0 / 3.0) < 2e-4


def test_clipping_oracle_and_device_specification_agree_with_the_rasterised_iou():
    from fake_t3d import box3d_iou_spec
    c1, s1, h1, c2, s2, h2 = pairs()
    worst = 0.0
    n_pos = n_ill = 0
    for i in range(len(h1)):
        ras = box3d_iou_raster(c1[i], s1[i], h1[i], c2[i], s2[i], h2[i])
        with np.errstate(all='ignore'):
            clip = RB.get_box3d_iou(c1[i], s1[i], h1[i], c2[i], s2[i], h2[i])
        spec = box3d_iou_spec(c1[i], s1[i], h1[i], c2[i], s2[i], h2[i])
        touching = 160 <= i < 180
        if touching and not (np.isfinite(clip[0]) and abs(clip[0] - ras[0]) < 1e-3):
            # Boxes that share a face: two edges of the ground rectangles are collinear, and the published Sutherland-Hodgman step
            # intersects two (nearly) parallel lines there -- 1 / (dc x dp) with dc x dp = 0 up to rounding: the clip of box_util is
            # ill-conditioned exactly here (nan, or a vertex far away: IoU 0.005 on pair 164).  The device algorithm keeps a coincident edge once and
            # returns 0, which is what counting cells gives; nothing to compare the clip with.
            n_ill += 1
        else:
            assert abs(clip[0] - ras[0]) < 1e-3 and abs(clip[1] - ras[1]) < 1e-3, (i, clip, ras)
            worst = max(worst, abs(clip[0] - ras[0]), abs(clip[1] - ras[1]))
        assert abs(spec[0] - ras[0]) < 1e-3 and abs(spec[1] - ras[1]) < 1e-3, (i, spec, ras)
        worst = max(worst, abs(spec[0] - ras[0]), abs(spec[1] - ras[1]))
        n_pos += ras[0] > 0.05
    assert n_pos >= 100
    # nested: the inner box's volume over the outer's; touching and far pairs: nothing
    for i in range(140, 160):
        want = np.prod(s2[i]) / np.prod(s1[i])
        assert abs(RB.get_box3d_iou(c1[i], s1[i], h1[i], c2[i], s2[i], h2[i])[0] - want) < 1e-9
        assert abs(box3d_iou_raster(c1[i], s1[i], h1[i], c2[i], s2[i], h2[i])[0] - want) < 1e-3
    for i in range(160, 200):
        assert box3d_iou_raster(c1[i], s1[i], h1[i], c2[i], s2[i], h2[i])[0] < 1e-3
        assert box3d_iou_spec(c1[i], s1[i], h1[i], c2[i], s2[i], h2[i])[0] < 1e-9
    for i in range(180, 200):
        assert RB.get_box3d_iou(c1[i], s1[i], h1[i], c2[i], s2[i], h2[i])[0] == 0.0
    print('largest |clip or spec - raster| over 200 pairs: %.2e; touching pairs on which the published clip is ill-conditioned: %d of 20' % (worst, n_ill))


@pytest.mark.gpu
def test_device_box3d_iou_agrees_with_the_rasterised_iou(hip_lib):
    import torch
    from transferable3d_amd import abi
    from transferable3d_amd.abi import fptr
    c1, s1, h1, c2, s2, h2 = pairs()
    n = len(h1)
    t = [torch.as_tensor(np.ascontiguousarray(a, dtype=np.float32)).cuda() for a in (c1, s1, h1, c2, s2, h2)]
    i3, i2 = torch.zeros(n, device='cuda'), torch.zeros(n, device='cuda')
    a = abi.Box3dIouArgs(*[fptr(x) for x in t], fptr(i3), fptr(i2), n)
    assert hip_lib.t3d_box3d_iou(C.byref(a), C.c_void_p(torch.cuda.current_stream().cuda_stream)) == 0
    torch.cuda.synchronize()
    i3, i2 = i3.cpu().numpy().astype(np.float64), i2.cpu().numpy().astype(np.float64)
    f32 = lambda x: np.asarray(x, np.float32).astype(np.float64)          # the boxes the device actually saw
    for i in range(n):
        ras = box3d_iou_raster(f32(c1[i]), f32(s1[i]), float(f32(h1[i])), f32(c2[i]), f32(s2[i]), float(f32(h2[i])))
        assert abs(i3[i] - ras[0]) < 1e-3 and abs(i2[i] - ras[1]) < 1e-3, (i, i3[i], i2[i], ras)
