"""Operator wrappers under the reference's names (models/tf_util.py:1258-1323 conv2d, 1463-1499 fully_connected,
1501-1524 max_pool2d, 1666-1705 batch_norm_for_*, 1720-1741 dropout; 466-484 tf_normalize_2D_bboxes;
1001-1041 anchor->reg conversion), backed by the fused HIP layers of engine.py instead of TF ops.

Semantics that differ from an eager op library, all consequences of fusing ("lazy tensors"):
  * conv2d returns a LAZY activation: the layer writes its raw matmul output and per-tile statistics; batch-norm
    and ReLU are applied by whichever op consumes it.
  * a global max-pool over the points is an epilogue of the producing conv2d, so it must be requested there
    (`pool_over_points=True`, optional `rowmask`); `max_pool2d` then just returns that pooled tensor.
Unsupported argument combinations raise NotImplementedError naming the reference behaviour that is missing."""
import numpy as np

from . import api
from .constants import MEAN_DIMS_ARR, NUM_HEADING_BIN, ORIENT_ANCHORS
from .engine import ActSpec, FcLayer, PointLayer


class PointTensor(api.Tensor):
    """(B, N, 1, C) per-point activation held lazily as an engine.ActSpec."""

    def __init__(self, ctx, spec, layer=None):
        e = ctx.engine
        api.Tensor.__init__(self, ctx, None, (e.B, e.rpf, 1, spec.K), 'point_tensor', producer=layer)
        self.spec, self.layer = spec, layer


def _as_spec(ctx, inputs):
    if isinstance(inputs, PointTensor):
        return inputs.spec
    if isinstance(inputs, api.LazyPoints):
        e = ctx.engine
        return ActSpec(inputs.pc.buf, e.ldpc, inputs.ncols, sub=inputs.sub.buf if inputs.sub is not None else None, sub_ld=3)
    if isinstance(inputs, api.Placeholder):          # the raw point cloud (B,N,C)
        e = ctx.engine
        return ActSpec(inputs.buf, e.ldpc, e.C)
    raise TypeError('conv2d input must be a point tensor, got %r' % (inputs,))


def conv2d(inputs, num_output_channels, kernel_size, scope, stride=[1, 1], padding='SAME', data_format='NHWC',
           use_xavier=True, stddev=1e-3, weight_decay=None, activation_fn='relu', bn=False, bn_decay=None,
           is_training=None, pool_over_points=False, rowmask=None):
    """2-D convolution with a 1x1 (or [1,D] over a one-channel image) kernel == the per-point shared MLP layer."""
    ctx = api.get_default_graph()
    if list(stride) != [1, 1] or data_format != 'NHWC' or not use_xavier or weight_decay is not None:
        raise NotImplementedError('conv2d: only stride 1, NHWC, xavier init, no weight decay are on the hot path')
    if not bn or activation_fn not in ('relu',):
        raise NotImplementedError('conv2d without batch-norm+ReLU (the conv10 logits layer) is fused into the '
                                  'segmentation head: use semisup_models.v1_inst_seg')
    spec = _as_spec(ctx, inputs)
    kh, kw = kernel_size
    if kh != 1 or kw not in (1, spec.K):
        raise NotImplementedError('conv2d: kernel must be [1,1] or [1,D]')
    layer = PointLayer(ctx.engine, scope, spec.K, num_output_channels, kernel_1xD=(kw != 1), pool=pool_over_points)
    out = layer.fwd(ctx.engine.fwd, spec, bool(is_training), rowmask=rowmask.buf if rowmask is not None else None)
    return PointTensor(ctx, out, layer)


def max_pool2d(inputs, kernel_size, scope, stride=[2, 2], padding='VALID'):
    """Global max over the point axis of a conv2d output built with pool_over_points=True -> (B,1,1,C)."""
    layer = getattr(inputs, 'layer', None)
    ctx = api.get_default_graph()
    if layer is None or not layer.pool or kernel_size[0] != ctx.engine.rpf or kernel_size[1] != 1:
        raise NotImplementedError('max_pool2d: only the global pool over all N points, fused into the producing '
                                  'conv2d(pool_over_points=True)')
    return api.Tensor(ctx, layer.pooled, (ctx.engine.B, 1, 1, layer.N), scope, producer=layer)


def fully_connected(inputs, num_outputs, scope, use_xavier=True, stddev=1e-3, weight_decay=None, activation_fn='relu',
                    bn=False, bn_decay=None, is_training=None):
    ctx = api.get_default_graph()
    K = inputs.shape[-1]
    layer = FcLayer(ctx.engine, scope, K, num_outputs, bn=bn, act=activation_fn)
    out = layer.fwd(ctx.engine.fwd, inputs.buf, K, bool(is_training))
    return api.Tensor(ctx, out, (ctx.engine.B, num_outputs), scope, producer=layer)


def dropout(inputs, is_training, scope, keep_prob=0.5, noise_shape=None):
    raise NotImplementedError('dropout is fused into its producer (FcLayer keep_prob / the segmentation head); '
                              'standalone tf_util.dropout nodes are not on the hot path')


def batch_norm_for_conv2d(inputs, is_training, bn_decay, scope, data_format='NHWC'):
    raise NotImplementedError('batch-norm is part of conv2d(bn=True) (statistics in its epilogue, apply on load)')


def batch_norm_for_fc(inputs, is_training, bn_decay, scope):
    raise NotImplementedError('batch-norm is part of fully_connected(bn=True)')


# ---- host-side geometry helpers (tiny, per-frustum; run on NumPy arrays) ----------------------------------------
def tf_normalize_2D_bboxes(box2D, image_dim):
    """tf_util.py:466-484: [left/cols, top/rows, right/cols, bottom/rows]."""
    box2D, image_dim = np.asarray(box2D, np.float32), np.asarray(image_dim, np.float32)
    rows, cols = image_dim[:, 0], image_dim[:, 1]
    return np.stack([box2D[:, 0] / cols, box2D[:, 1] / rows, box2D[:, 2] / cols, box2D[:, 3] / rows], axis=1)


def tf_convert_box_params_from_anchor_to_reg_format_multi(box_params, y_classes=None, dims_anchors=None, orient_anchors=None):
    """tf_util.py:1001-1041 on NumPy arrays: argmax bins -> (center, dims=max(anchor+res,1e-5), theta).  `y_classes`
    is accepted and ignored, as in the reference (tf_util.py:1017-1023)."""
    center, dims_cls, dims_reg, orient_cls, orient_reg = [np.asarray(x) for x in box_params]
    dims_anchors = MEAN_DIMS_ARR.astype(np.float32) if dims_anchors is None else np.asarray(dims_anchors)
    orient_anchors = ORIENT_ANCHORS.astype(np.float32) if orient_anchors is None else np.asarray(orient_anchors)
    k, j = dims_cls.argmax(1), orient_cls.argmax(1)
    ar = np.arange(center.shape[0])
    dims = np.maximum(dims_anchors[k] + dims_reg[ar, k], 1e-5)
    return center, dims, orient_anchors[j] + orient_reg[ar, j]


def tf_expand_tile(tensor, axis, tile):
    return np.tile(np.expand_dims(np.asarray(tensor), axis=axis), tile)
