"""TEST INFRASTRUCTURE ONLY -- NumPy (fp64) restatement of the 3-D box IoU the reference calls.  PARITY UNPINNED for box3d_iou
(the module is absent upstream); roty / get_3d_box / class2angle / class2size are pinned on the reference's own functions
(tests/test_reference_vectors.py).

`box_util.box3d_iou` is imported by the reference (roi_seg_box3d_dataset.py:15, box_pc_fit_dataset.py:17, eval_det.py:59,
evaluate.py:19) but is not in its tree: it is train/box_util.py of Frustum PointNets (charlesq34/frustum-pointnets, no version
pinned -- the reference appends '../../train' to sys.path).  This file restates that module's published algorithm:

    box3d_iou(corners1, corners2):  rect_k = [(corners_k[i,0], corners_k[i,2]) for i in 3,2,1,0]
        area_k  = shoelace(rect_k);   inter = Sutherland-Hodgman clip of rect1 by rect2, area of its convex hull
        iou_2d  = inter_area / (area1 + area2 - inter_area)
        ymax = min(corners1[0,1], corners2[0,1]);  ymin = max(corners1[4,1], corners2[4,1])
        inter_vol = inter_area * max(0, ymax - ymin);  vol_k = product of the three edge lengths at corner 0
        iou = inter_vol / (vol1 + vol2 - inter_vol)

and the reference's own callers: get_3d_box (roi_seg_box3d_dataset.py:86-101), class2angle / class2size (66-84),
compute_box3d_iou (103-140).  The reference holds no test vectors for any of this; the tests pin this file on closed-form
cases (offset squares, a square against its 45-degree turn, containment, disjoint boxes).
"""
import numpy as np

from .ref_constants import MEAN_DIMS_ARR, NUM_HEADING_BIN


def roty(t):
    c, s = np.cos(t), np.sin(t)
    return np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]])


def get_3d_box(box_size, heading_angle, center):
    """roi_seg_box3d_dataset.py:86-101: (8,3) corners, 0-3 the +h/2 face, 4-7 the -h/2 face."""
    l, w, h = box_size
    x = [l / 2, l / 2, -l / 2, -l / 2, l / 2, l / 2, -l / 2, -l / 2]
    y = [h / 2, h / 2, h / 2, h / 2, -h / 2, -h / 2, -h / 2, -h / 2]
    z = [w / 2, -w / 2, -w / 2, w / 2, w / 2, -w / 2, -w / 2, w / 2]
    c = np.dot(roty(heading_angle), np.vstack([x, y, z]))
    return (c + np.asarray(center, np.float64).reshape(3, 1)).T


def class2angle(pred_cls, residual, num_class=NUM_HEADING_BIN, to_label_format=True):
    """roi_seg_box3d_dataset.py:66-73."""
    angle = pred_cls * (2 * np.pi / float(num_class)) + residual
    if to_label_format and angle > np.pi:
        angle = angle - 2 * np.pi
    return angle


def class2size(pred_cls, residual):
    """roi_seg_box3d_dataset.py:81-84."""
    return MEAN_DIMS_ARR[pred_cls] + residual


def poly_area(x, y):
    return 0.5 * np.abs(np.dot(x, np.roll(y, 1)) - np.dot(y, np.roll(x, 1)))


def polygon_clip(subject, clip):
    """Sutherland-Hodgman: `subject` clipped by the convex polygon `clip` (both lists of (x, y), `clip` counter-clockwise).
    None when nothing is left."""
    def inside(p):
        return (cp2[0] - cp1[0]) * (p[1] - cp1[1]) > (cp2[1] - cp1[1]) * (p[0] - cp1[0])

    def intersection():
        dc = [cp1[0] - cp2[0], cp1[1] - cp2[1]]
        dp = [s[0] - e[0], s[1] - e[1]]
        n1 = cp1[0] * cp2[1] - cp1[1] * cp2[0]
        n2 = s[0] * e[1] - s[1] * e[0]
        n3 = 1.0 / (dc[0] * dp[1] - dc[1] * dp[0])
        return [(n1 * dp[0] - n2 * dc[0]) * n3, (n1 * dp[1] - n2 * dc[1]) * n3]

    out = list(subject)
    cp1 = clip[-1]
    for cp2 in clip:
        inp, out = out, []
        s = inp[-1]
        for e in inp:
            if inside(e):
                if not inside(s):
                    out.append(intersection())
                out.append(e)
            elif inside(s):
                out.append(intersection())
            s = e
        cp1 = cp2
        if not out:
            return None
    return out


def convex_hull_intersection(p1, p2):
    inter = polygon_clip(p1, p2)
    if inter is None or len(inter) < 3:
        return 0.0
    pts = np.asarray(inter, np.float64)          # the clip of two convex polygons is convex: hull area == polygon area
    return float(poly_area(pts[:, 0], pts[:, 1]))


def box3d_vol(c):
    a = np.sqrt(np.sum((c[0] - c[1]) ** 2))
    b = np.sqrt(np.sum((c[1] - c[2]) ** 2))
    h = np.sqrt(np.sum((c[0] - c[4]) ** 2))
    return a * b * h


def box3d_iou(corners1, corners2):
    corners1, corners2 = np.asarray(corners1, np.float64), np.asarray(corners2, np.float64)
    rect1 = [(corners1[i, 0], corners1[i, 2]) for i in range(3, -1, -1)]
    rect2 = [(corners2[i, 0], corners2[i, 2]) for i in range(3, -1, -1)]
    area1 = poly_area(np.array(rect1)[:, 0], np.array(rect1)[:, 1])
    area2 = poly_area(np.array(rect2)[:, 0], np.array(rect2)[:, 1])
    inter_area = convex_hull_intersection(rect1, rect2)
    iou_2d = inter_area / (area1 + area2 - inter_area)
    ymax = min(corners1[0, 1], corners2[0, 1])
    ymin = max(corners1[4, 1], corners2[4, 1])
    inter_vol = inter_area * max(0.0, ymax - ymin)
    iou = inter_vol / (box3d_vol(corners1) + box3d_vol(corners2) - inter_vol)
    return iou, iou_2d


def get_box3d_iou(center_a, size_a, heading_a, center_b, size_b, heading_b):
    """box_pc_fit_dataset.py:38-42."""
    return box3d_iou(get_3d_box(size_a, heading_a, center_a), get_3d_box(size_b, heading_b, center_b))


def compute_box3d_iou(center_pred, heading_logits, heading_residuals, size_logits, size_residuals, center_label, heading_class_label,
                      heading_residual_label, size_class_label, size_residual_label):
    """roi_seg_box3d_dataset.py:103-140 -> (iou2ds, iou3ds)."""
    B = heading_logits.shape[0]
    hc, sc = np.argmax(heading_logits, 1), np.argmax(size_logits, 1)
    i2, i3 = [], []
    for i in range(B):
        c = get_3d_box(class2size(sc[i], size_residuals[i, sc[i]]), class2angle(hc[i], heading_residuals[i, hc[i]]), center_pred[i])
        g = get_3d_box(class2size(size_class_label[i], size_residual_label[i]),
                       class2angle(heading_class_label[i], heading_residual_label[i]), center_label[i])
        a, b = box3d_iou(c, g)
        i3.append(a)
        i2.append(b)
    return np.array(i2, np.float32), np.array(i3, np.float32)
