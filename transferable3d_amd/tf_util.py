"""Operator wrappers under the reference's names (models/tf_util.py:1258-1323 conv2d, 1463-1499 fully_connected,
1501-1524 max_pool2d, 1666-1705 batch_norm_for_*, 1720-1741 dropout; 466-484 tf_normalize_2D_bboxes;
1001-1041 anchor->reg conversion), backed by the fused HIP layers of engine.py instead of TF ops.

Semantics that differ from an eager op library, all consequences of fusing ("lazy tensors"):
  * conv2d returns a LAZY activation: the layer writes its raw matmul output and per-tile statistics; batch-norm
    and ReLU are applied by whichever op consumes it.
  * a global max-pool over the points is an epilogue of the producing conv2d: `max_pool2d(net, [num_point, 1])` -- the reference's
    call -- switches it on in that layer's recorded launches (PointLayer.enable_pool) and returns the pooled tensor.
  * dropout / batch_norm_for_conv2d / batch_norm_for_fc exist as nodes of their own (small launches, or lazy scale/shift fused
    into the consumer); on the hot path they are parts of the fused layers.
Ops built here go into the graph's forward schedule (forward-only surface); the training step is emitted by the fused
sub-networks of semisup_models.  Unsupported argument combinations raise NotImplementedError naming the reference behaviour that
is missing."""
import numpy as np

from . import abi, api
from .abi import fptr
from .constants import BN_EPS, MEAN_DIMS_ARR, NUM_HEADING_BIN, ORIENT_ANCHORS
from .engine import ActSpec, FcLayer, PointLayer


class PointTensor(api.Tensor):
    """(B, N, 1, C) per-point activation held lazily as an engine.ActSpec."""

    def __init__(self, ctx, spec, layer=None):
        e = ctx.engine
        api.Tensor.__init__(self, ctx, None, (e.B, e.rpf, 1, spec.K), 'point_tensor', producer=layer)
        self.spec, self.layer = spec, layer

    def numpy(self):
        """The value the consumers see: relu?(x * scale + shift) - sub, evaluated at fetch time on the host (the tensor itself is
        never materialised on the device)."""
        s, e = self.spec, self.ctx.engine
        v = s.x.detach().float().cpu()[:, s.coff:s.coff + s.K]
        if s.scale is not None:
            v = v * s.scale.detach().cpu() + s.shift.detach().cpu()
        if s.relu:
            v = v.clamp(min=0)
        if s.sub is not None:
            v = v - s.sub.detach().cpu()[:, :s.K].repeat_interleave(e.rpf, 0)
        return v.numpy().reshape(self._shape)


def _as_spec(ctx, inputs):
    if isinstance(inputs, PointTensor):
        return inputs.spec
    if isinstance(inputs, api.LazyPoints):
        e = ctx.engine
        return ActSpec(inputs.pc.buf, e.ldpc, inputs.ncols, sub=inputs.sub.buf if inputs.sub is not None else None, sub_ld=3)
    if isinstance(inputs, api.Placeholder):          # the raw point cloud (B,N,C)
        e = ctx.engine
        return ActSpec(inputs.buf, e.ldpc, e.C)
    if isinstance(inputs, api.MaskedPoints):
        raise NotImplementedError('a masked per-point tensor (api.multiply) can only feed tf_util.max_pool2d (semisup_models.py:184-188)')
    raise TypeError('conv2d input must be a point tensor, got %r' % (inputs,))


def conv2d(inputs, num_output_channels, kernel_size, scope, stride=[1, 1], padding='SAME', data_format='NHWC',
           use_xavier=True, stddev=1e-3, weight_decay=None, activation_fn='relu', bn=False, bn_decay=None,
           is_training=None, pool_over_points=False, rowmask=None):
    """2-D convolution with a 1x1 (or [1,D] over a one-channel image) kernel == the per-point shared MLP layer
    (tf_util.py:1258-1323).  bn=True: batch-norm statistics in the GEMM's epilogue, batch-norm + ReLU applied by whichever op
    consumes the result.  bn=False (tf_util.py:1316 skipped): the bare convolution + bias, activation_fn None or relu (applied on
    load by the consumer); a following batch_norm_for_conv2d turns it into the bn=True form.  `activation_fn`: api.nn.relu (the
    reference's default tf.nn.relu), None, or the strings.  `inputs` may be the lazy concat of a point tensor with a tiled
    per-frustum vector (api.concat of api.tile, semisup_models.py:107-108): the layer is then emitted in its split form -- a
    per-point GEMM on the point part plus ONE [B, Cg] x [Cg, N] product added as a per-frustum row bias -- over the one weight
    variable <scope>/weights [1,1,Cp+Cg,N] the reference has.  A following max_pool2d(net, [num_point,1]) fuses into this layer
    (PointLayer.enable_pool); `pool_over_points` / `rowmask` request that at build time (the fused sub-networks' own use).
    Forward-only on this surface: the training step of the hot path is emitted by the fused sub-networks (semisup_models)."""
    ctx = api.get_default_graph()
    scope = api.scoped(scope)
    activation_fn = api.activation_name(activation_fn)
    if list(stride) != [1, 1] or data_format != 'NHWC' or not use_xavier or weight_decay is not None:
        raise NotImplementedError('conv2d: only stride 1, NHWC, xavier init, no weight decay are on the hot path')
    if activation_fn not in ('relu', None) or (bn and activation_fn != 'relu'):
        raise NotImplementedError('conv2d: activation_fn is tf.nn.relu or None at every call site of the reference')
    kh, kw = kernel_size
    if isinstance(inputs, api.ConcatPointGlobal):
        if not bn or (kh, kw) != (1, 1) or pool_over_points:
            raise NotImplementedError('conv2d on a [point | tiled global] concat: the batch-normed 1x1 form of semisup_models.py:111')
        e, vs = ctx.engine, ctx.engine.vars
        pspec, parts = _as_spec(ctx, inputs.point), inputs.tiled.vec.parts
        if len(parts) > 2:
            raise NotImplementedError('conv2d: a tiled global vector of at most two blocks (pooled feature [+ one-hot])')
        Kp, Kg = pspec.K, sum(n for _, n in parts)
        N = num_output_channels
        w = vs.xavier(scope + '/weights', (1, 1, Kp + Kg, N), Kp + Kg, N).view(Kp + Kg, N)
        layer = PointLayer(e, scope, Kp, N, w=w[0:Kp], w_name=scope + '/weights', w_row0=0, gram=False)
        glob = FcLayer(e, scope + '/global', parts[0][1], N, bn=False, act=None, K2=parts[1][1] if len(parts) == 2 else 0, w=w[Kp:],
                       bias=None)
        rb = glob.fwd(e.fwd, parts[0][0], parts[0][0].shape[-1], bool(is_training),
                      in2=parts[1][0] if len(parts) == 2 else None, ld_in2=parts[1][0].shape[-1] if len(parts) == 2 else 0)
        out = layer.fwd(e.fwd, pspec, bool(is_training), rowbias=rb)
        return PointTensor(ctx, out, layer)
    spec = _as_spec(ctx, inputs)
    if kh != 1 or kw not in (1, spec.K):
        raise NotImplementedError('conv2d: kernel must be [1,1] or [1,D]')
    if not bn and pool_over_points:
        raise NotImplementedError('conv2d: the fused max-pool follows a batch-normed layer at every call site of the reference')
    n_alloc = (num_output_channels + 63) // 64 * 64
    layer = PointLayer(ctx.engine, scope, spec.K, num_output_channels, kernel_1xD=(kw != 1), pool=pool_over_points, bn=bn,
                       n_alloc=None if n_alloc == num_output_channels else n_alloc, gram=False)      # (the output tensor exists: an op graph may read it)
    out = layer.fwd(ctx.engine.fwd, spec, bool(is_training), rowmask=rowmask.buf if rowmask is not None else None)
    if not bn:
        out.relu = activation_fn == 'relu'
    return PointTensor(ctx, out, layer)


def max_pool2d(inputs, kernel_size, scope, stride=[2, 2], padding='VALID'):
    """tf_util.max_pool2d(net, [num_point, 1], padding='VALID', scope=...) (tf_util.py:1501-1524 at semisup_models.py:96, 188, 244,
    375): the global max over the point axis -> (B,1,1,C).  The pool is an epilogue of the conv2d that produced `inputs` (per-tile
    max / min / arg partials, picked in the batch-norm finalize): it is switched on in that layer's recorded launches here, when the
    reference asks for it.  `inputs` multiplied by the mask first (api.multiply): the mask becomes the pool's row mask."""
    layer = getattr(inputs, 'layer', None)
    ctx = api.get_default_graph()
    if layer is None or not layer.bn or kernel_size[0] != ctx.engine.rpf or kernel_size[1] != 1 or padding != 'VALID':
        raise NotImplementedError('max_pool2d: only the global pool over all N points of a batch-normed conv2d output '
                                  '(every call site of the reference)')
    mask = getattr(inputs, 'rowmask', None)
    if not layer.pool or mask is not None:
        layer.enable_pool(rowmask=mask)
    return api.Tensor(ctx, layer.pooled, (ctx.engine.B, 1, 1, layer.N), api.scoped(scope), producer=layer)


def fully_connected(inputs, num_outputs, scope, use_xavier=True, stddev=1e-3, weight_decay=None, activation_fn='relu',
                    bn=False, bn_decay=None, is_training=None):
    ctx = api.get_default_graph()
    scope, activation_fn = api.scoped(scope), api.activation_name(activation_fn)
    K = inputs.shape[-1]
    layer = FcLayer(ctx.engine, scope, K, num_outputs, bn=bn, act=activation_fn)
    out = layer.fwd(ctx.engine.fwd, inputs.buf, K, bool(is_training))
    return api.Tensor(ctx, out, (ctx.engine.B, num_outputs), scope, producer=layer)


def dropout(inputs, is_training, scope, keep_prob=0.5, noise_shape=None):
    """tf.cond(is_training, tf.nn.dropout(inputs, keep_prob), inputs) (tf_util.py:1720-1741) as a node of its own.  The keep mask
    lives in a buffer registered under `scope` (generated per step by t3d_dropout_mask in a training schedule, or fed through
    Session.run(feed_dict={scope: mask})).  [B,N] tensors: an identity-weight t3d_fc_fwd; per-point tensors: t3d_act_dropout
    materialises the dropped-out activation (on the hot path the reference's one such node, conv9 -> dp1 -> conv10, is inside
    t3d_seg_head)."""
    if noise_shape is not None:
        raise NotImplementedError('dropout: noise_shape is None at every call site of the reference')
    if isinstance(inputs, api.MaskedPoints):
        raise NotImplementedError('a masked per-point tensor (api.multiply) can only feed tf_util.max_pool2d (semisup_models.py:184-188)')
    ctx = api.get_default_graph()
    scope = api.scoped(scope)
    e = ctx.engine
    if isinstance(inputs, PointTensor):
        spec = inputs.spec
        out = e.rt.zeros(e.M, spec.K)
        a = abi.ActDropoutArgs()
        a.a, a.out, a.M, a.K, a.rows_per_frustum, a.keep_prob = spec.struct(), fptr(out), e.M, spec.K, e.rpf, float(keep_prob)
        if bool(is_training):
            mask = e.rt.full((e.M, spec.K), 1.0)
            e.dropout_masks[scope] = (mask, float(keep_prob))
            a.mask = fptr(mask)
        a._keep = (spec, out)
        e.fwd.add('t3d_act_dropout', a)
        return PointTensor(ctx, ActSpec(out, spec.K, spec.K), None)
    N = inputs.shape[-1]
    layer = FcLayer(e, scope, N, N, bn=False, act=None, w=FcLayer.IDENTITY, bias=None, keep_prob=float(keep_prob), drop_scope=scope)
    out = layer.fwd(e.fwd, inputs.buf, N, bool(is_training))
    return api.Tensor(ctx, out, (e.B, N), scope, producer=layer)


def batch_norm_for_conv2d(inputs, is_training, bn_decay, scope, data_format='NHWC'):
    """tf.contrib.layers.batch_norm over (batch, points) per channel (tf_util.py:1693-1705 -> batch_norm_template 1645-1664) on
    the output of conv2d(bn=False, activation_fn=None): the statistics come from the partial sums that GEMM's epilogue already
    wrote (t3d_bn_fwd_finalize; variables <scope>/{beta,gamma,moving_mean,moving_variance}), the normalisation is applied by the
    consumer while it loads the tensor."""
    if data_format != 'NHWC':
        raise NotImplementedError('batch_norm_for_conv2d: NHWC only')
    if not isinstance(inputs, PointTensor) or inputs.layer is None or inputs.layer.bn or inputs.spec.relu:
        raise NotImplementedError('batch_norm_for_conv2d follows conv2d(bn=False, activation_fn=None)')
    ctx = api.get_default_graph()
    scope = api.scoped(scope)
    e, lay, vs = ctx.engine, inputs.layer, ctx.engine.vars
    N = lay.N
    if isinstance(bn_decay, (int, float)):
        e.hyper[2] = float(bn_decay)
    gamma, beta = vs.const(scope + '/gamma', (N,), 1.0), vs.const(scope + '/beta', (N,), 0.0)
    mm = vs.const(scope + '/moving_mean', (N,), 0.0, trainable=False)
    mv = vs.const(scope + '/moving_variance', (N,), 1.0, trainable=False)
    scale, shift, mean, invstd = (e.rt.zeros(N) for _ in range(4))
    f = abi.BnFwdFinalizeArgs()
    f.psum, f.psumsq, f.n_tiles, f.count, f.N = fptr(lay.psum), fptr(lay.psumsq), lay.T, lay.M, N
    if lay.NA != N:            # padded GEMM: the partials have NA columns per tile
        raise NotImplementedError('batch_norm_for_conv2d on a layer with fewer than 64 channels')
    f.gamma, f.beta, f.moving_mean, f.moving_var = fptr(gamma), fptr(beta), fptr(mm), fptr(mv)
    f.decay, f.eps, f.is_training, f.unbiased_ema = fptr(e.bn_decay_ptr), BN_EPS, int(bool(is_training)), int(e.unbiased_ema)
    f.scale, f.shift, f.mean, f.invstd = fptr(scale), fptr(shift), fptr(mean), fptr(invstd)
    f._keep = (gamma, beta, mm, mv, scale, shift, mean, invstd)
    e.fwd.add('t3d_bn_fwd_finalize', f)
    return PointTensor(ctx, ActSpec(lay.y, N, N, 0, scale, shift, False, producer=lay), lay)


def batch_norm_for_fc(inputs, is_training, bn_decay, scope):
    """Batch-norm over the B rows of a [B,N] tensor (tf_util.py:1666-1677) as a node of its own: an identity-weight t3d_fc_fwd
    (variables <scope>/{beta,gamma,moving_mean,moving_variance}), no activation."""
    ctx = api.get_default_graph()
    scope = api.scoped(scope)
    e = ctx.engine
    if isinstance(bn_decay, (int, float)):
        e.hyper[2] = float(bn_decay)
    N = inputs.shape[-1]
    layer = FcLayer(e, scope, N, N, bn=True, act=None, w=FcLayer.IDENTITY, bias=None, bn_scope=scope)
    out = layer.fwd(e.fwd, inputs.buf, N, bool(is_training))
    return api.Tensor(ctx, out, (e.B, N), scope, producer=layer)


# ---- host-side geometry helpers (tiny, per-frustum; run on NumPy arrays) ----------------------------------------
class NormBox2D(api.Tensor):
    """tf_normalize_2D_bboxes of the box2D / img_dim placeholders: a lazy handle -- the division runs in t3d_box2d_feats at the head of
    the forward, which writes it straight into the [one_hot | norm_box2D] input block of the T-Net / box net FC layers."""

    def __init__(self, ctx, box2D, image_dim):
        api.Tensor.__init__(self, ctx, None, (ctx.engine.B, 4), 'norm_box2D')
        self.box2D, self.image_dim = box2D, image_dim

    def numpy(self):
        return tf_normalize_2D_bboxes(self.box2D.numpy(), self.image_dim.numpy())


def tf_normalize_2D_bboxes(box2D, image_dim):
    """tf_util.py:466-484: [left/cols, top/rows, right/cols, bottom/rows]."""
    if isinstance(box2D, api.Tensor):
        if getattr(box2D, 'field', None) != 'box2D' or getattr(image_dim, 'field', None) != 'img_dim':
            raise NotImplementedError('tf_normalize_2D_bboxes takes the box2D and img_dim placeholders (train_semisup.py:240)')
        return NormBox2D(box2D.ctx, box2D, image_dim)
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
