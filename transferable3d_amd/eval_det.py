"""Detection evaluation: per-class precision / recall / average precision with 3-D IoU matching.

Mirrors the interface of the reference's sunrgbd/sunrgbd_detection/eval_det.py (`voc_ap` 25-57, `get_iou` 61-67, `eval_det_cls`
69-151, `eval_det` 153-199) and the prediction -> box conversion of evaluate.py:56-72.  The matching rule is the PASCAL-VOC one the
reference follows: detections of a class in decreasing score order, each takes the ground-truth box of its image it overlaps
most; a true positive if that overlap exceeds the threshold and the box is still unclaimed, otherwise a false positive.

What differs is where the work happens: every (detection, ground-truth box of the same image and class) pair goes to the device in
ONE t3d_box3d_iou_corners launch per class (box_util.box3d_iou per pair on the host in the reference); the greedy claim pass over
the sorted detections, which is sequential by definition, stays on the host and touches only the precomputed overlaps.
"""
import ctypes as C

import numpy as np
import torch

from . import abi
from .abi import fptr
from .constants import MEAN_DIMS_ARR, NUM_HEADING_BIN


def voc_ap(rec, prec, use_07_metric=False):
    """Area under the precision-recall curve: the 11-point VOC07 average, or the exact area under the monotone envelope."""
    rec, prec = np.asarray(rec, np.float64), np.asarray(prec, np.float64)
    if use_07_metric:
        pts = [prec[rec >= t].max() if np.any(rec >= t) else 0.0 for t in np.arange(0.0, 1.1, 0.1)]
        return float(np.sum(pts) / 11.0)
    r = np.concatenate(([0.0], rec, [1.0]))
    p = np.concatenate(([0.0], prec, [0.0]))
    p = np.maximum.accumulate(p[::-1])[::-1]                 # precision envelope: best precision at any recall >= r
    step = np.nonzero(r[1:] != r[:-1])[0]
    return float(np.sum((r[step + 1] - r[step]) * p[step + 1]))


def box3d_iou_batch(corners1, corners2, rt):
    """[n,8,3] x [n,8,3] -> (iou3d [n], iou2d [n]) on the device (t3d_box3d_iou_corners)."""
    n = len(corners1)
    if n == 0:
        return np.zeros(0, np.float32), np.zeros(0, np.float32)
    k1 = torch.as_tensor(np.ascontiguousarray(corners1, np.float32)).to(rt.device)
    k2 = torch.as_tensor(np.ascontiguousarray(corners2, np.float32)).to(rt.device)
    i3, i2 = torch.zeros(n, device=rt.device), torch.zeros(n, device=rt.device)
    a = abi.Box3dIouCornersArgs(fptr(k1), fptr(k2), fptr(i3), fptr(i2), n)
    rc = rt.lib.t3d_box3d_iou_corners(C.byref(a), rt.stream())
    if rc != 0:
        raise abi.T3DError('t3d_box3d_iou_corners failed: %d' % rc)
    return i3.cpu().numpy(), i2.cpu().numpy()


def get_iou(bb1, bb2, rt):
    """3-D IoU of two boxes given as (8,3) corners (eval_det.py:61-67)."""
    return float(box3d_iou_batch(np.asarray(bb1)[None], np.asarray(bb2)[None], rt)[0][0])


def eval_det_cls(pred, gt, ovthresh=0.25, use_07_metric=False, rt=None):
    """One class.  pred: {img_id: [(bbox (8,3), score)]}, gt: {img_id: [bbox]} -> (rec [nd], prec [nd], ap)."""
    gt_boxes = {i: [np.asarray(b, np.float64) for b in boxes] for i, boxes in gt.items()}
    npos = sum(len(b) for b in gt_boxes.values())
    img, score, box = [], [], []
    for i, dets in pred.items():
        for b, s in dets:
            img.append(i)
            score.append(s)
            box.append(np.asarray(b, np.float64))
    nd = len(img)
    order = np.argsort(-np.asarray(score, np.float64), kind='stable') if nd else np.zeros(0, np.int64)
    # every (detection, candidate ground-truth box) pair of the class -> one launch
    pair_det, pair_gt, k1, k2 = [], [], [], []
    for d in order:
        for j, g in enumerate(gt_boxes.get(img[d], [])):
            pair_det.append(d)
            pair_gt.append(j)
            k1.append(box[d])
            k2.append(g)
    iou = box3d_iou_batch(np.asarray(k1).reshape(-1, 8, 3), np.asarray(k2).reshape(-1, 8, 3), rt)[0] if k1 else np.zeros(0)
    best = {}
    for d, j, v in zip(pair_det, pair_gt, iou):                     # first maximum in ground-truth order, as `iou > ovmax` picks
        if d not in best or v > best[d][0]:
            best[d] = (float(v), j)
    claimed = {i: [False] * len(b) for i, b in gt_boxes.items()}
    tp, fp = np.zeros(nd), np.zeros(nd)
    for rank, d in enumerate(order):
        ov, j = best.get(d, (-np.inf, -1))
        if ov > ovthresh and not claimed[img[d]][j]:
            tp[rank] = 1.0
            claimed[img[d]][j] = True
        else:
            fp[rank] = 1.0
    tp, fp = np.cumsum(tp), np.cumsum(fp)
    rec = tp / float(npos) if npos else np.full(nd, np.nan)          # the reference divides by npos as is
    prec = tp / np.maximum(tp + fp, np.finfo(np.float64).eps)
    return rec, prec, voc_ap(rec, prec, use_07_metric)


def eval_det(pred_all, gt_all, ovthresh=0.25, use_07_metric=False, rt=None):
    """pred_all: {img_id: [(classname, bbox, score)]}, gt_all: {img_id: [(classname, bbox)]}; ovthresh a scalar or {classname: t}
    -> ({class: rec}, {class: prec}, {class: ap}) over the classes that have ground truth (eval_det.py:153-199)."""
    if rt is None:
        from .engine import Runtime
        rt = Runtime()
    pred, gt = {}, {}
    for i, dets in pred_all.items():
        for name, b, s in dets:
            pred.setdefault(name, {}).setdefault(i, []).append((b, s))
            gt.setdefault(name, {}).setdefault(i, [])
    for i, boxes in gt_all.items():
        for name, b in boxes:
            gt.setdefault(name, {}).setdefault(i, []).append(b)
    rec, prec, ap = {}, {}, {}
    for name in gt:
        t = ovthresh[name] if isinstance(ovthresh, dict) else ovthresh
        rec[name], prec[name], ap[name] = eval_det_cls(pred.get(name, {}), gt[name], t, use_07_metric, rt)
    return rec, prec, ap


# ---- evaluate.py: predictions (test_semisup's 14-list) -> boxes -----------------------------------------------------------------
def get_3d_box(box_size, heading_angle, center):
    """(8,3) corners, rows 0-3 the +h/2 face (roi_seg_box3d_dataset.py:86-101)."""
    l, w, h = [float(v) for v in box_size]
    c, s = np.cos(heading_angle), np.sin(heading_angle)
    x = np.array([l, l, -l, -l, l, l, -l, -l]) / 2
    y = np.array([h, h, h, h, -h, -h, -h, -h]) / 2
    z = np.array([w, -w, -w, w, w, -w, -w, w]) / 2
    return np.stack([c * x + s * z + center[0], y + center[1], -s * x + c * z + center[2]], 1)


def predictions_to_boxes(predictions, classes, test_classes=None):
    """The `B) Get PRED boxes` half of evaluate_predictions (evaluate.py:56-72): class2angle / class2size / get_3d_box in the
    centre view, rotated back by -rot_angle.  -> {img_id: [(classname, corners (8,3), score)]}."""
    _, _, _, center_l, hcls_l, hres_l, scls_l, sres_l, rot_l, score_l, cls_l, file_l = predictions[:12]
    out = {}
    for i in range(len(center_l)):
        name = classes[int(cls_l[i])]
        if test_classes is not None and name not in test_classes:
            raise Exception('Not supposed to have class: %s' % name)              # evaluate.py:60
        heading = int(hcls_l[i]) * (2 * np.pi / NUM_HEADING_BIN) + float(hres_l[i])
        if heading > np.pi:
            heading -= 2 * np.pi
        k = get_3d_box(MEAN_DIMS_ARR[int(scls_l[i])] + np.asarray(sres_l[i], np.float64), heading, np.asarray(center_l[i], np.float64).squeeze())
        c, s = np.cos(-float(rot_l[i])), np.sin(-float(rot_l[i]))                 # rotate_pc_along_y(corners, -rot_angle)
        k[:, [0, 2]] = np.stack([c * k[:, 0] - s * k[:, 2], s * k[:, 0] + c * k[:, 2]], 1)
        out.setdefault(file_l[i], []).append((name, k, float(score_l[i])))
    return out


def evaluate_predictions(predictions, gt_all, classes, test_classes=None, ovthresh=0.25, rt=None):
    """evaluate.py:27-76 with the ground-truth boxes given ({img_id: [(classname, corners)]}, what its part A builds from the
    SUN-RGBD label files) -> (rec, prec, ap, mean_ap)."""
    rec, prec, ap = eval_det(predictions_to_boxes(predictions, classes, test_classes), gt_all, ovthresh, rt=rt)
    return rec, prec, ap, float(np.mean([ap[k] for k in ap]))


def get_ap_info(ap, mean_ap):
    """evaluate.py:96-103."""
    lines = ['Average Precision:'] + ['%11s: [%.1f]' % (k, 100.0 * ap[k]) for k in sorted(ap)]
    return '\n'.join(lines) + '\n    Mean AP:  %.1f' % (100.0 * mean_ap)
