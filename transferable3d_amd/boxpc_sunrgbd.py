"""Box-PC Fit model builder under the reference's names and signatures
(sunrgbd/sunrgbd_detection/boxpc_sunrgbd.py: placeholder_inputs 33-54, get_model 56-100, get_loss 106-128,
convert_raw_y_box_to_reg_format 206-230)."""
import numpy as np

from . import api
from .constants import MEAN_DIMS_ARR, NUM_CLASS, ORIENT_ANCHORS


def placeholder_inputs(batch_size, num_point, num_channels):
    """The reference's 12 placeholders, same order."""
    ctx = api.get_default_graph()
    ctx.ensure_engine(batch_size, num_point, num_channels)
    B, N, C = batch_size, num_point, num_channels
    P = api.placeholder
    return (P('pc', (B, N, C)), P('one_hot_vec', (B, NUM_CLASS)), P('y_seg', (B, N)), P('y_center', (B, 3)),
            P('y_orient_cls', (B,)), P('y_orient_reg', (B,)), P('y_dims_cls', (B,)), P('y_dims_reg', (B, 3)),
            P('y_box_iou', (B,)), P('y_center_delta', (B, 3)), P('y_dims_delta', (B, 3)), P('y_orient_delta', (B,)))


class BoxRegHandle(tuple):
    """(centers, dims, orients) of convert_raw_y_box_to_reg_format, carried symbolically: the conversion runs inside
    the representation kernel (t3d_boxpc_rep, label form)."""


def convert_raw_y_box_to_reg_format(y_box, one_hot_vec):
    y_centers, y_orient_cls, y_orient_reg, y_dims_cls, y_dims_reg = y_box
    if isinstance(y_centers, api.Tensor):
        h = BoxRegHandle((y_centers, y_dims_reg, y_orient_reg))
        h.labels = y_box
        return h
    dims = np.maximum(MEAN_DIMS_ARR.astype(np.float32)[np.asarray(y_dims_cls)] + np.asarray(y_dims_reg), 1e-5)
    return np.asarray(y_centers), dims, ORIENT_ANCHORS.astype(np.float32)[np.asarray(y_orient_cls)] + np.asarray(y_orient_reg)


def get_model(boxpc, is_training, one_hot_vec, use_one_hot_vec=False, bn_decay=None, c=None):
    """boxpc = (box_reg, pc).  Returns (pred, end_points) with pred = (fit logits, (delta centre, size, angle))
    (boxpc_sunrgbd.py:56-100)."""
    from . import semisup_models
    from .semisup_models import SlicedTensor
    box_reg, pc = boxpc
    ctx = pc.ctx
    B = ctx.engine.B
    oh_pl = one_hot_vec
    end_points = {'class_ids': SlicedTensor(pc, lambda: np.argmax(ctx.inputs.one_hot_vec.cpu().numpy(), axis=1).astype(np.int32),
                                            (B,), 'class_ids')}
    if not use_one_hot_vec:
        one_hot_vec = None
    output, feats = semisup_models.box_pc_mask_features_model(box_reg, pc, None, 2 + 7, is_training, end_points=end_points,
                                                              reuse=False, bn_for_output=False, one_hot_vec=one_hot_vec,
                                                              norm_box2D=None, bn_decay=bn_decay, c=c, scope='box_pc_mask_model')
    out = output.buf
    T = lambda buf, shape, name: api.Tensor(ctx, buf, shape, name)
    logits = T(out[:, 7:9], (B, 2), 'boxpc_fit_logits')
    terms = ctx.assembly.loss_op.terms
    lw = T(terms[:, 2], (B,), 'logits_for_weigh')
    if c.BOXPC_WEIGH_DELTA_PRED_BY_CLS_CONF:       # boxpc_sunrgbd.py:84-92: the deltas the model hands on are (1 - p_fit) * raw; the loss kernel
        # weighs its own copy (t3d_boxpc_loss.weigh_pred_by_cls_conf), these handles are evaluated on the host at fetch time
        full = T(out, (B, 9), 'boxpc_out')
        wd = lambda: 1.0 - lw.numpy()
        dc = SlicedTensor(full, lambda: full.numpy()[:, 0:3] * wd()[:, None], (B, 3), 'boxpc_delta_center')
        ds = SlicedTensor(full, lambda: full.numpy()[:, 3:6] * wd()[:, None], (B, 3), 'boxpc_delta_size')
        da = SlicedTensor(full, lambda: full.numpy()[:, 6] * wd(), (B,), 'boxpc_delta_angle')
    else:
        dc, ds, da = T(out[:, 0:3], (B, 3), 'boxpc_delta_center'), T(out[:, 3:6], (B, 3), 'boxpc_delta_size'), T(out[:, 6], (B,), 'boxpc_delta_angle')
    end_points.update({'boxpc_feats_dict': feats, 'boxpc_fit_logits': logits,
                       'pred_boxpc_fit': SlicedTensor(lw, lambda: (lw.numpy() > 0.5).astype(np.int32), (B,), 'pred_boxpc_fit'),
                       'logits_for_weigh': lw, 'boxpc_delta_center': dc, 'boxpc_delta_size': ds, 'boxpc_delta_angle': da})
    return (logits, (dc, ds, da)), end_points


def get_loss(pred, labels, end_points, reduce_loss=True, c=None):
    ctx = pred[0].ctx
    lop = ctx.assembly.loss_op
    B = ctx.engine.B
    ctx.loss = api.Tensor(ctx, lop.loss, (), 'boxpc_loss') if reduce_loss else api.Tensor(ctx, lop.terms[:, 3], (B,), 'boxpc_losses')
    end_points['boxpc_cls_losses'] = api.Tensor(ctx, lop.terms[:, 0], (B,), 'boxpc_cls_losses')
    end_points['boxpc_delta_losses'] = api.Tensor(ctx, lop.terms[:, 1], (B,), 'boxpc_delta_losses')
    return ctx.loss
