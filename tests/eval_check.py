"""Shared by the CPU (specification library) and GPU (HIP library) evaluation tests."""
import numpy as np

from oracle import ref_box as RB
from oracle import ref_eval as RE
from transferable3d_amd import eval_det as E
from transferable3d_amd.constants import MEAN_DIMS_ARR

CLASSES = ['bed', 'table', 'sofa', 'chair', 'toilet', 'desk', 'dresser', 'night_stand', 'bookshelf', 'bathtub']


def synthetic_scene(seed, n_img=12):
    """Ground-truth boxes per image and detections that are noisy copies of them (some missed, some duplicated, some spurious)."""
    r = np.random.RandomState(seed)
    gt_all, pred_all = {}, {}
    for i in range(n_img):
        gt_all[i], pred_all[i] = [], []
        for _ in range(r.randint(1, 5)):
            k = int(r.randint(0, 4))
            c, s, h = r.normal(size=3) * [2, 0.2, 2] + [0, 0, 4], MEAN_DIMS_ARR[k] * r.uniform(0.8, 1.2, 3), r.uniform(-np.pi, np.pi)
            gt_all[i].append((CLASSES[k], RB.get_3d_box(s, h, c)))
            for _ in range(r.randint(0, 3)):                              # 0 = missed, 2 = a duplicate detection
                q = r.uniform(0.0, 0.6)
                pred_all[i].append((CLASSES[k], RB.get_3d_box(s * (1 + r.normal(size=3) * q * 0.3), h + r.normal() * q, c + r.normal(size=3) * q * 0.5),
                                    float(1 - q + r.normal() * 0.05)))
        for _ in range(r.randint(0, 3)):                                  # spurious detections, possibly of a class without GT here
            k = int(r.randint(0, 5))
            pred_all[i].append((CLASSES[k], RB.get_3d_box(MEAN_DIMS_ARR[k], r.uniform(-3, 3), r.normal(size=3) * [2, 0.2, 2] + [0, 0, 4]),
                                float(r.uniform(0, 0.7))))
    pred_all[n_img] = [(CLASSES[0], RB.get_3d_box(MEAN_DIMS_ARR[0], 0.3, [0, 0, 3]), 0.9)]          # an image without any ground truth
    return pred_all, gt_all


def check_eval_det(rt):
    for seed, thr, m07 in ((0, 0.25, False), (1, 0.5, False), (2, {c: 0.25 for c in CLASSES}, True)):
        pred_all, gt_all = synthetic_scene(seed)
        rec, prec, ap = E.eval_det(pred_all, gt_all, thr, use_07_metric=m07, rt=rt)
        wrec, wprec, wap = RE.eval_det(pred_all, gt_all, thr, use_07_metric=m07)
        assert sorted(ap) == sorted(wap) and len(ap) >= 4
        for k in wap:
            # same claims, detection by detection (a class with detections but no ground truth has 0/0 recall, as in the reference)
            assert np.array_equal(rec[k], wrec[k], equal_nan=True) and np.array_equal(prec[k], wprec[k]), k
            assert (np.isnan(ap[k]) and np.isnan(wap[k])) or abs(ap[k] - wap[k]) < 1e-12, k
        assert 0.05 < np.nanmean(list(ap.values())) < 0.95
    return ap


def check_predictions_round_trip(rt):
    """Boxes encoded the way test_semisup writes them (centre view, bin + residual) evaluate to AP 1 against themselves."""
    r = np.random.RandomState(5)
    n = 20
    cls = r.randint(0, 10, n)
    center = r.normal(size=(n, 3)) * [1, 0.2, 1] + [0, 0, 4]
    hcls, hres = r.randint(0, 12, n), r.uniform(-1, 1, n) * np.pi / 12
    sres = r.normal(size=(n, 3)) * 0.1
    rot = r.uniform(-0.5, 0.5, n) + np.pi / 2
    preds = [[None] * n, [None] * n, [None] * n, list(center), list(hcls), list(hres), list(cls), list(sres), list(rot), list(r.uniform(size=n)),
             list(cls), list(range(n)), None, None]
    boxes = E.predictions_to_boxes(preds, CLASSES)
    gt_all = {i: [(nm, k) for nm, k, _ in v] for i, v in boxes.items()}
    # the conversion agrees with the oracle's get_3d_box + class2angle + a rotation by -rot_angle
    i = 3
    want = RB.get_3d_box(MEAN_DIMS_ARR[cls[i]] + sres[i], RB.class2angle(hcls[i], hres[i]), center[i])
    c, s = np.cos(-rot[i]), np.sin(-rot[i])
    want[:, [0, 2]] = np.stack([c * want[:, 0] - s * want[:, 2], s * want[:, 0] + c * want[:, 2]], 1)
    assert np.abs(boxes[i][0][1] - want).max() < 1e-12
    rec, prec, ap, mean_ap = E.evaluate_predictions(preds, gt_all, CLASSES, rt=rt)
    assert abs(mean_ap - 1.0) < 1e-12 and all(abs(v - 1) < 1e-12 for v in ap.values())
    assert E.get_ap_info(ap, mean_ap).endswith('Mean AP:  100.0')
