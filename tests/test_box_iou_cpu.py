"""CPU: the 3-D box IoU (t3d.h K13) -- oracle (Sutherland-Hodgman restatement of box_util.box3d_iou) pinned on closed-form
cases, and the specification of the device algorithm (boundary integral, tests/fake_t3d.py) against the oracle."""
import math

import numpy as np

from fake_t3d import box3d_iou_spec
from oracle import ref_box as RB


def test_oracle_closed_forms():
    unit = lambda cx, cz, h=0.0, s=(1.0, 1.0, 1.0), cy=0.0: RB.get_3d_box(s, h, (cx, cy, cz))
    # unit cubes offset by half an edge: inter 1/2, union 3/2
    i3, i2 = RB.box3d_iou(unit(0, 0), unit(0.5, 0))
    assert abs(i2 - 1 / 3) < 1e-12 and abs(i3 - 1 / 3) < 1e-12
    # a square against its 45-degree turn: regular octagon of area 2(sqrt2 - 1)
    i3, i2 = RB.box3d_iou(unit(0, 0), unit(0, 0, math.pi / 4))
    a = 2 * (math.sqrt(2) - 1)
    assert abs(i2 - a / (2 - a)) < 1e-12
    # containment: small box inside a big one -> ratio of volumes; vertical offset scales the 3-D value only
    big, small = RB.get_3d_box((4, 2, 2), 0.3, (1, 0, 5)), RB.get_3d_box((1, 1, 1), 0.3, (1.2, 0.5, 5.1))
    i3, i2 = RB.box3d_iou(small, big)
    assert abs(i2 - 1 / 8) < 1e-12 and abs(i3 - 1 / 16) < 1e-12
    i3, _ = RB.box3d_iou(small, RB.get_3d_box((4, 2, 2), 0.3, (1, 1.5, 5)))           # heights overlap by 1/2
    assert abs(i3 - 0.5 / (16 + 1 - 0.5)) < 1e-12
    # disjoint
    assert RB.box3d_iou(unit(0, 0), unit(3, 0)) == (0.0, 0.0)
    assert RB.box3d_iou(unit(0, 0), unit(0, 0, cy=2.0))[0] == 0.0


def random_boxes(r, n, spread=1.0):
    c = r.normal(0, spread, size=(n, 3)) + np.array([0, 0, 3.0])
    s = r.uniform(0.3, 2.5, size=(n, 3))
    h = r.uniform(-np.pi, np.pi, size=n)
    return c, s, h


def test_device_algorithm_matches_oracle_on_random_pairs():
    r = np.random.RandomState(0)
    c1, s1, h1 = random_boxes(r, 400)
    c2, s2, h2 = random_boxes(r, 400)
    c2[:200] = c1[:200] + r.normal(0, 0.2, size=(200, 3))          # near pairs: high IoU, many edge crossings
    s2[:200] = s1[:200] * r.uniform(0.8, 1.2, size=(200, 3))
    h2[:200] = h1[:200] + r.uniform(0, 0.5, size=200)
    nz = 0
    for i in range(400):
        want = RB.get_box3d_iou(c1[i], s1[i], h1[i], c2[i], s2[i], h2[i])
        got = box3d_iou_spec(c1[i], s1[i], h1[i], c2[i], s2[i], h2[i])
        assert abs(got[0] - want[0]) < 1e-9 and abs(got[1] - want[1]) < 1e-9, (i, got, want)
        nz += want[0] > 0
    assert nz > 250


def test_device_algorithm_degenerate_cases():
    c, s, h = np.array([0.3, -0.2, 4.0]), np.array([1.7, 0.9, 1.1]), 0.7
    assert abs(box3d_iou_spec(c, s, h, c, s, h)[0] - 1.0) < 1e-12          # a box against itself: coincident edges count once
    assert abs(box3d_iou_spec(c, s, h, c, s, h + 2 * np.pi)[0] - 1.0) < 1e-9
    assert abs(box3d_iou_spec(c, s, h, c, s, h + np.pi)[0] - 1.0) < 1e-9    # a box is symmetric under a half turn
    a = box3d_iou_spec(c, s, h, c + 0.1, s * 1.1, h + 0.2)
    b = box3d_iou_spec(c + 0.1, s * 1.1, h + 0.2, c, s, h)
    assert abs(a[0] - b[0]) < 1e-12 and abs(a[1] - b[1]) < 1e-12            # symmetric
    # shared edge, no overlap -- whichever corner the vertex lists start at (the boundary integral needs a closed loop)
    assert box3d_iou_spec([0, 0, 0], [1, 1, 1], 0.0, [1, 0, 0], [1, 1, 1], 0.0)[0] == 0.0
    from fake_t3d import iou_from_quads
    sq = np.array([[0.5, 0.5], [-0.5, 0.5], [-0.5, -0.5], [0.5, -0.5]])
    for k in range(4):
        for m in range(4):
            assert iou_from_quads(np.roll(sq, k, 0), np.roll(sq + [1.0, 0.0], m, 0), 0.5, -0.5, 0.5, -0.5, 1.0, 1.0) == (0.0, 0.0)
            half = iou_from_quads(np.roll(sq, k, 0), np.roll(sq + [0.5, 0.0], m, 0), 0.5, -0.5, 0.5, -0.5, 1.0, 1.0)   # collinear, same way
            assert abs(half[0] - 1 / 3) < 1e-12 and abs(half[1] - 1 / 3) < 1e-12
    # negative l / w mirror the ground rectangle (same IoU); a negative h inverts the height range: iou3d = 0, iou2d unchanged
    assert abs(box3d_iou_spec(c, s * [-1, -1, 1], h, c, s, h)[0] - 1.0) < 1e-12
    full = box3d_iou_spec(c, -s, h, c, s, h)
    assert full[0] == 0.0 and abs(full[1] - 1.0) < 1e-12
    neg = RB.get_box3d_iou(c, s * [-1, 1, -1], h, c + 0.1, s, h + 0.3)
    pos = RB.get_box3d_iou(c, s, h, c + 0.1, s, h + 0.3)
    assert neg[0] == 0.0 and abs(neg[1] - pos[1]) < 1e-12                  # the oracle (reference behaviour) agrees
    got = box3d_iou_spec(c, s * [-1, 1, -1], h, c + 0.1, s, h + 0.3)
    assert got[0] == 0.0 and abs(got[1] - pos[1]) < 1e-9


def test_compute_box3d_iou_oracle_uses_argmax_bins_and_label_bins():
    r = np.random.RandomState(3)
    B = 6
    hl, sl = r.randn(B, 12), r.randn(B, 10)
    hr, sr = r.randn(B, 12) * 0.1, r.randn(B, 10, 3) * 0.1
    cp, cl = r.randn(B, 3) * 0.2, r.randn(B, 3) * 0.2
    hcl, scl = r.randint(0, 12, B), r.randint(0, 10, B)
    hrl, srl = r.randn(B) * 0.1, r.randn(B, 3) * 0.1
    i2, i3 = RB.compute_box3d_iou(cp, hl, hr, sl, sr, cl, hcl, hrl, scl, srl)
    assert i2.dtype == np.float32 and i2.shape == (B,) and np.all(i3 <= i2 + 1e-6) and np.all(i3 >= 0)
    b = 2
    k, j = int(np.argmax(sl[b])), int(np.argmax(hl[b]))
    want = RB.get_box3d_iou(cp[b], RB.class2size(k, sr[b, k]), RB.class2angle(j, hr[b, j]), cl[b], RB.class2size(scl[b], srl[b]),
                            RB.class2angle(hcl[b], hrl[b]))
    assert abs(i3[b] - want[0]) < 1e-6
