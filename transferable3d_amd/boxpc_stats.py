"""Per-class statistics of the Box-PC Fit net, as the reference's stage-b driver reports them
(sunrgbd/sunrgbd_detection/train_boxpc.py: ClassificationStats 651-714, BoxDeltaIOUStats 507-649): precision / recall / F1 of the
fit / no-fit decision, and the 3-D IoU with the label box before and after the predicted centre / size / angle deltas are taken
off the box the net was shown.  The IoUs are computed on the device (t3d_box3d_iou), one launch per report.
"""
import ctypes as C

import numpy as np
import torch

from . import abi
from .abi import fptr
from .constants import MEAN_DIMS_ARR, NUM_HEADING_BIN, class2type


class ClassificationStats:
    def __init__(self, classes):
        self.classes = list(classes)
        self.loss_sum, self.num_batches = 0.0, 0
        self.pred, self.y, self.cls = [], [], []

    def add_loss(self, loss):
        self.loss_sum += float(loss)
        self.num_batches += 1

    def add_prediction(self, pred_fit, y_fit, y_cls):
        self.pred.extend(np.asarray(pred_fit).tolist())
        self.y.extend(np.asarray(y_fit).tolist())
        self.cls.extend(np.asarray(y_cls).tolist())

    def get_mean_loss(self):
        return self.loss_sum / max(self.num_batches, 1)

    def get_batch_stats(self):
        pred, y, cls = np.array(self.pred), np.array(self.y), np.array(self.cls)
        stats = {}
        for c, name in enumerate(self.classes):
            sel = cls == c
            if not sel.any():
                continue
            tp = float(np.sum((y == 1) & (pred == 1) & sel))
            fp = float(np.sum((y == 0) & (pred == 1) & sel))
            fn = float(np.sum((y == 1) & (pred == 0) & sel))
            prec, rec = tp / (tp + fp + 1e-3), tp / (tp + fn + 1e-3)             # train_boxpc.py:687-689
            stats[name] = (prec, rec, 2 * prec * rec / (prec + rec + 1e-3), int(sel.sum()))
        return stats

    @staticmethod
    def summarize_stats(stats):
        rows = ['%11s %6s %6s %4s %5s' % ('  Classname', 'Prec', 'Recall', ' F1 ', 'Supp')]
        rows += ['%11s: %.3f %.3f %.3f %5d' % ((k,) + stats[k]) for k in sorted(stats)]
        cols = np.array([stats[k] for k in stats], np.float64).reshape(-1, 4)
        rows.append('%11s: %.3f %.3f %.3f %5d' % (('      MEAN ',) + tuple(cols.mean(0))))
        return '\n'.join(rows) + '\n'


class BoxDeltaIOUStats:
    """Boxes in label form (centre, heading bin + residual, size class + residual), as the driver holds them."""

    def __init__(self, classes, rt):
        self.classes, self.rt = list(classes), rt
        self.ori, self.dele, self.y, self.cls = [], [], [], []

    @staticmethod
    def _params(box):
        center, hcls, hres, scls, sres = [np.asarray(v) for v in box]
        return (center.reshape(-1, 3).astype(np.float32),
                (MEAN_DIMS_ARR[scls.astype(int)] + sres.reshape(-1, 3)).astype(np.float32),
                (hcls.astype(np.float64) * (2 * np.pi / NUM_HEADING_BIN) + hres).astype(np.float32))

    def add_prediction(self, ori_box, ori_box_aft_delta, y_box, y_cls):
        self.ori.append(self._params(ori_box))
        self.dele.append(self._params(ori_box_aft_delta))
        self.y.append(self._params(y_box))
        self.cls.extend(np.asarray(y_cls).tolist())

    def _iou(self, a, b):
        n = len(a[0])
        dev = self.rt.device
        t = [torch.as_tensor(np.ascontiguousarray(v)).to(dev) for v in (a[0], a[1], a[2], b[0], b[1], b[2])]
        out = torch.zeros(n, device=dev)
        args = abi.Box3dIouArgs(*[fptr(v) for v in t], fptr(out), fptr(None), n)
        abi.check(self.rt.lib.t3d_box3d_iou(C.byref(args), self.rt.stream()), 't3d_box3d_iou')
        return out.cpu().numpy()

    def get_batch_stats(self):
        cat = lambda lst: tuple(np.concatenate([x[i] for x in lst]) for i in range(3))
        y = cat(self.y)
        iou_ori, iou_del = self._iou(y, cat(self.ori)), self._iou(y, cat(self.dele))      # box3d_iou(y_box3d, .) train_boxpc.py:588-589
        cls = np.array(self.cls)
        stats = {}
        for c, name in enumerate(self.classes):
            sel = cls == c
            if sel.any():
                stats[name] = (float(iou_ori[sel].mean()), float(iou_del[sel].mean()), float(iou_del[sel].mean() - iou_ori[sel].mean()),
                               int(sel.sum()))
        return stats

    @staticmethod
    def summarize_stats(stats):
        sign = lambda v: (' +' if v > 0 else ' ') + '%.3f' % v
        rows = ['%11s %6s %6s %6s %5s' % ('  Classname', 'Before', 'After ', ' +/- ', 'Supp')]
        rows += ['%11s: %.3f %.3f %4s %5d' % (k, stats[k][0], stats[k][1], sign(stats[k][2]), stats[k][3]) for k in sorted(stats)]
        cols = np.array([stats[k] for k in stats], np.float64).reshape(-1, 4)
        m = cols.mean(0)
        rows.append('%11s: %.3f %.3f %4s %5d' % ('      MEAN ', m[0], m[1], sign(m[2]), m[3]))
        return '\n'.join(rows) + '\n'


ALL_CLASSES = [class2type[i] for i in range(10)]


def record_batch(cls_stats, pos_stats, neg_stats, fit_lo, loss, pred_fit, deltas, inputs):
    """The per-batch bookkeeping of train_boxpc.py:369-395 from the fetched outputs and the batch that produced them (`inputs`: the
    device-resident feed buffers, nets.Inputs).  The label box is the shown box minus the true deltas."""
    g = lambda k: getattr(inputs, k).cpu().numpy()
    x_center, x_hcls, x_hres, x_scls, x_sres = g('y_center'), g('y_orient_cls'), g('y_orient_reg'), g('y_dims_cls'), g('y_dims_reg')
    iou, t_dc, t_ds, t_da = g('y_box_iou'), g('y_center_delta'), g('y_dims_delta'), g('y_orient_delta')
    y_cls = np.argmax(g('one_hot_vec'), axis=1)
    del_center, del_size, del_angle = deltas
    cls_stats.add_prediction(pred_fit, (iou > fit_lo).astype(int), y_cls)
    cls_stats.add_loss(loss)
    ori = (x_center, x_hcls, x_hres, x_scls, x_sres)
    dele = (x_center - del_center, x_hcls, x_hres - del_angle, x_scls, x_sres - del_size)
    ybox = (x_center - t_dc, x_hcls, x_hres - t_da, x_scls, x_sres - t_ds)
    for stats, sel in ((pos_stats, iou >= fit_lo), (neg_stats, iou < fit_lo)):
        if sel.any():
            stats.add_prediction([v[sel] for v in ori], [v[sel] for v in dele], [v[sel] for v in ybox], y_cls[sel])
