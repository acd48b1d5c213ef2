"""TEST INFRASTRUCTURE ONLY -- restatement of the reference's detection evaluation loop.  PINNED on outputs of the reference's own
eval_det.py executed in the build container (voc_ap, eval_det_cls, eval_det on axis-aligned boxes, where the IoU of the absent
box_util is a closed form: tests/golden/make_reference_vectors.py -> tests/test_reference_vectors.py).

Follows sunrgbd/sunrgbd_detection/eval_det.py: voc_ap 25-57 (sentinels, backward running maximum, sum over recall steps; 11-point
variant), eval_det_cls 69-151 (detections by decreasing confidence; per detection a scan over the image's ground-truth boxes with
`iou > ovmax`; a box can be claimed once), eval_det 153-199 (regrouping by class).  Plain loops, one IoU at a time.
"""
import numpy as np

from oracle.ref_box import box3d_iou


def voc_ap(rec, prec, use_07_metric=False):
    if use_07_metric:
        ap = 0.0
        for t in np.arange(0.0, 1.1, 0.1):
            p = 0 if np.sum(rec >= t) == 0 else np.max(prec[rec >= t])
            ap += p / 11.0
        return ap
    mrec = np.concatenate(([0.0], rec, [1.0]))
    mpre = np.concatenate(([0.0], prec, [0.0]))
    for i in range(mpre.size - 1, 0, -1):
        mpre[i - 1] = max(mpre[i - 1], mpre[i])
    idx = np.where(mrec[1:] != mrec[:-1])[0]
    return float(np.sum((mrec[idx + 1] - mrec[idx]) * mpre[idx + 1]))


def eval_det_cls(pred, gt, ovthresh=0.25, use_07_metric=False):
    recs, npos = {}, 0
    for img_id, boxes in gt.items():
        recs[img_id] = {'bbox': [np.asarray(b, np.float64) for b in boxes], 'det': [False] * len(boxes)}
        npos += len(boxes)
    for img_id in pred:
        recs.setdefault(img_id, {'bbox': [], 'det': []})
    ids, conf, bbs = [], [], []
    for img_id, dets in pred.items():
        for box, score in dets:
            ids.append(img_id)
            conf.append(score)
            bbs.append(np.asarray(box, np.float64))
    order = np.argsort(-np.asarray(conf), kind='stable')
    nd = len(ids)
    tp, fp = np.zeros(nd), np.zeros(nd)
    for d, k in enumerate(order):
        R = recs[ids[k]]
        ovmax, jmax = -np.inf, -1
        for j, g in enumerate(R['bbox']):
            iou = box3d_iou(bbs[k], g)[0]
            if iou > ovmax:
                ovmax, jmax = iou, j
        if ovmax > ovthresh and not R['det'][jmax]:
            tp[d] = 1.0
            R['det'][jmax] = True
        else:
            fp[d] = 1.0
    fp, tp = np.cumsum(fp), np.cumsum(tp)
    with np.errstate(invalid='ignore', divide='ignore'):
        rec = tp / float(npos)
    prec = tp / np.maximum(tp + fp, np.finfo(np.float64).eps)
    return rec, prec, voc_ap(rec, prec, use_07_metric)


def eval_det(pred_all, gt_all, ovthresh=0.25, use_07_metric=False):
    pred, gt = {}, {}
    for img_id, dets in pred_all.items():
        for name, box, score in dets:
            pred.setdefault(name, {}).setdefault(img_id, []).append((box, score))
            gt.setdefault(name, {}).setdefault(img_id, [])
    for img_id, boxes in gt_all.items():
        for name, box in boxes:
            gt.setdefault(name, {}).setdefault(img_id, []).append(box)
    rec, prec, ap = {}, {}, {}
    for name in gt:
        t = ovthresh[name] if isinstance(ovthresh, dict) else ovthresh
        rec[name], prec[name], ap[name] = eval_det_cls(pred.get(name, {}), gt[name], t, use_07_metric)
    return rec, prec, ap
