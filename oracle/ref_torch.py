"""ORACLE (test infrastructure only) -- stock-PyTorch CPU restatement of the reference hot path.

PARITY UNPINNED against TensorFlow: the reference's arithmetic lives in TensorFlow 1.x (version
unpinned, not installable here) and the reference ships no tests or golden
vectors.  This file restates the reference graph from its source lines plus documented TF-1 op
semantics, using only stock torch CPU ops + autograd.  It is cross-checked against the independent
NumPy restatement in `oracle/ref_np.py` (tests/test_oracle.py) and frozen into tests/golden/*.npz.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the
product path (transferable3d_amd/) never does.

Tensors are (B, N, C) channel-fastest, exactly the reference's NHWC with H=N, W=1.  Parameters are
a flat dict keyed by the reference's TF variable names (`<scope>/weights`, `<scope>/biases`,
`<scope>/bn/{beta,gamma,moving_mean,moving_variance}`).
"""
import math
import re
from types import SimpleNamespace

import numpy as np
import torch
import torch.nn.functional as F

from .ref_constants import NUM_HEADING_BIN, NUM_SIZE_CLUSTER, NUM_CLASS, MEAN_DIMS_ARR, ORIENT_ANCHORS, BN_EPS, BOX_OUT_DIMS


# ----------------------------------------------------------------------------------------------
# parameters
# ----------------------------------------------------------------------------------------------
# (scope, kind, Cin, Cout, bn) in creation order.  kind: 'conv' = tf_util.conv2d per-point layer,
# 'fc' = tf_util.fully_connected.  `cin` of the first conv of each net is filled from the channels.
def layer_table(num_channels, model='A', use_one_hot=False, prefix_agnostic='', boxpc_channels=None, norm_box2D=False):
    """Layer list of the nets on the path.

    model 'A': inst_seg + tnet + box_est (semisup_models.py:69-291).
    model 'F': same under `class_agnostic/` + `class_dependent/box_refine` (semisup_v1_sunrgbd.py:132-230).
    model 'boxpc': `box_pc_mask_model` (semisup_models.py:326-398).
    """
    oh = NUM_CLASS if use_one_hot else 0
    L = []
    if model in ('A', 'F'):
        p = 'class_agnostic/' if model == 'F' else ''
        ohA = oh if model == 'A' else 0      # model F never feeds one_hot to the agnostic nets
        nb = 4 if norm_box2D else 0          # USE_NORMALIZED_BOX2D_AS_FEATS: fc1 of the T-Net / box net reads 4 more columns (semisup_models.py:194-195, 251-252)
        seg = [('conv1', num_channels, 64), ('conv2', 64, 64), ('conv3', 64, 64), ('conv4', 64, 128),
               ('conv5', 128, 1024), ('conv6', 64 + 1024 + ohA, 512), ('conv7', 512, 256),
               ('conv8', 256, 128), ('conv9', 128, 128)]
        for n, ci, co in seg:
            L.append((p + 'inst_seg/' + n, 'conv', ci, co, True))
        L.append((p + 'inst_seg/conv10', 'conv', 128, 2, False))
        for n, ci, co in [('conv-reg1-stage1', 3, 128), ('conv-reg2-stage1', 128, 128),
                          ('conv-reg3-stage1', 128, 256)]:
            L.append((p + 'tnet/' + n, 'conv', ci, co, True))
        L.append((p + 'tnet/fc1-stage1', 'fc', 256 + ohA + nb, 256, True))
        L.append((p + 'tnet/fc2-stage1', 'fc', 256, 128, True))
        L.append((p + 'tnet/fc3-stage1', 'fc', 128, 3, False))
        for n, ci, co in [('conv-reg1', 3, 128), ('conv-reg2', 128, 128), ('conv-reg3', 128, 256),
                          ('conv-reg4', 256, 512)]:
            L.append((p + 'box_est/' + n, 'conv', ci, co, True))
        L.append((p + 'box_est/fc1', 'fc', 512 + ohA + nb, 512, True))
        L.append((p + 'box_est/fc2', 'fc', 512, 256, True))
        L.append((p + 'box_est/fc3', 'fc', 256, BOX_OUT_DIMS, False))
        if model == 'F':
            q = 'class_dependent/box_refine/'
            L.append((q + 'fc0', 'fc', 512 + oh, 512, True))
            L.append((q + 'fc1', 'fc', 512, 256, True))
            L.append((q + 'fc2', 'fc', 256, BOX_OUT_DIMS, False))
    elif model == 'boxpc':
        p = prefix_agnostic + 'box_pc_mask_model/'
        d = (boxpc_channels if boxpc_channels is not None else num_channels) + 6
        for n, ci, co in [('conv-reg1', d, 128), ('conv-reg2', 128, 128), ('conv-reg3', 128, 256),
                          ('conv-reg4', 256, 512)]:
            L.append((p + n, 'conv', ci, co, True))
        L.append((p + 'fc1', 'fc', 512 + oh, 512, True))
        L.append((p + 'fc2', 'fc', 512, 256, True))
        L.append((p + 'fc3', 'fc', 256, 9, False))
    else:
        raise ValueError(model)
    return L


def init_params(rng, table, dtype=torch.float64, first_conv_kernel_is_1xD=True):
    """Xavier-uniform weights, zero biases, gamma=1, beta=0, moving_mean=0, moving_variance=1
    (tf_util.py:1165-1190,1312; tf.contrib.layers.batch_norm defaults).

    xavier_initializer(): limit = sqrt(6/(fan_in+fan_out)); for a conv kernel [kh,kw,Cin,Cout],
    fan_in = kh*kw*Cin and fan_out = kh*kw*Cout.  The first layer of each net that reads the raw
    point channels is a `[1,D]` kernel over a 1-channel image (semisup_models.py:76, 354), i.e.
    shape [1,D,1,Cout]: fan_in = D, fan_out = D*Cout.  1x1 kernels [1,1,Cin,Cout]: fan_in = Cin,
    fan_out = Cout.
    """
    P = {}
    for scope, kind, ci, co, bn in table:
        one_x_d = kind == 'conv' and (scope.endswith('inst_seg/conv1')
                                      or scope.endswith('box_pc_mask_model/conv-reg1'))
        if one_x_d:
            fan_in, fan_out = ci, ci * co
            shape = (1, ci, 1, co)
        elif kind == 'conv':
            fan_in, fan_out = ci, co
            shape = (1, 1, ci, co)
        else:
            fan_in, fan_out = ci, co
            shape = (ci, co)
        lim = math.sqrt(6.0 / (fan_in + fan_out))
        w = rng.uniform(-lim, lim, size=shape)
        P[scope + '/weights'] = torch.tensor(w, dtype=dtype)
        P[scope + '/biases'] = torch.zeros(co, dtype=dtype)
        if bn:
            P[scope + '/bn/beta'] = torch.zeros(co, dtype=dtype)
            P[scope + '/bn/gamma'] = torch.ones(co, dtype=dtype)
            P[scope + '/bn/moving_mean'] = torch.zeros(co, dtype=dtype)
            P[scope + '/bn/moving_variance'] = torch.ones(co, dtype=dtype)
    return P


class SharedParams(dict):
    """Variables looked up under a RE-USED scope (tf.variable_scope('D_boxpc_branch', reuse=True), train_semisup_adv.py:341-369): the
    k-th further evaluation of the Box-PC net runs under the scope tag `D_boxpc_branch@k/`, which reads the variables of
    `D_boxpc_branch/` while keeping its per-scope records (forced decisions, margins) apart from the other evaluations'."""

    @staticmethod
    def _k(k):
        return re.sub(r'@\d+/', '/', k)

    def __getitem__(self, k):
        return dict.__getitem__(self, self._k(k))

    def __contains__(self, k):
        return dict.__contains__(self, self._k(k))

    def get(self, k, d=None):
        return dict.get(self, self._k(k), d)


def trainable_names(P):
    return [k for k in P if not (k.endswith('moving_mean') or k.endswith('moving_variance'))]


# ----------------------------------------------------------------------------------------------
# layer wrappers (tf_util.py:1258-1323, 1463-1499, 1501-1524, 1645-1705, 1720-1741)
# ----------------------------------------------------------------------------------------------
class Ctx:
    """Carries params, training flag, bn_decay, injected dropout masks and collects EMA updates."""

    def __init__(self, P, is_training=True, bn_decay=None, dropout_masks=None, ema_unbiased=True):
        self.P = P
        self.is_training = is_training
        self.bn_decay = 0.9 if bn_decay is None else bn_decay   # tf_util.py:1659
        self.dropout_masks = dropout_masks or {}
        self.ema_unbiased = ema_unbiased
        self.ema_updates = {}
        self.is_training_override = {}   # scope prefix -> bool (stage c: frozen BoxPC net)
        # Flip-aware gradient checks (tests/model_check.py): an fp32 implementation and this fp64 restatement can disagree on the
        # side of a ReLU input that is within fp32 rounding of zero, or on which of two near-equal rows is the max-pool's arg-max;
        # either moves every gradient below it by one element's contribution.  A test may therefore hand over the decisions the
        # implementation under test actually took -- `forced_gates[scope]` (1 where its ReLU passed, 0 where not, -1 = not handed over) and `forced_argmax[scope]`
        # ((B, C) row index, -1 = keep the natural one) -- and the restatement differentiates THAT branch of the piecewise-linear
        # function; forward values move by the size of the disputed pre-activations (~1e-6).  `flips` counts the disagreements.
        self.forced_gates = {}
        self.forced_argmax = {}
        # bf16 emulation (BASELINE configs[4]; tests/test_bf16_gpu.py): the roundings of the T3D_BF16 path injected at the points the
        # kernels round -- both operands of every per-point GEMM (the activated input and the weights) and every STORED raw layer
        # output; the max-pooled layers (no stored output), the conv10 logits (computed in the segmentation head from the stored
        # conv9 output), the fully-connected heads, statistics and losses stay unrounded.  Roundings are straight-through for
        # autograd: the gradient is that of the rounded forward, without the backward pass's own roundings (dz / dy to bf16).
        self.bf16 = False
        self.keep_raw = None         # dict: conv2d records its raw (pre-batch-norm) outputs here, for layer-wise comparisons
        self.forced_mask = None      # (B, N) 0/1: the hard segmentation mask the implementation under test took (logit0 < logit1 with
        #                              the two logits within fp32 rounding of each other is the same kind of decision)
        self.flips = {}
        # What a forced decision may cost: per site (numel, flips, worst margin, scale).  The margin of a flipped decision is THIS
        # restatement's own distance from the boundary -- |post-batch-norm pre-activation| of a flipped ReLU gate, natural maximum
        # minus the forced row's value of a flipped arg-max, |logit0 - logit1| of a flipped mask point -- and `scale` the magnitude
        # of the tensor it is measured in.  An implementation that decides differently only where the decision is within rounding of
        # the boundary has margins of ~1e-6 * scale and a handful of flips; one that gates, pools or masks WRONGLY on some subset
        # shows margins of the order of the activations themselves.  tests/model_check.check_decision_margins bounds both, so the
        # product cannot drag this oracle's gradient along a branch the reference would not take.
        self.margins = {}

    def training_for(self, scope):
        for pre, val in self.is_training_override.items():
            if scope.startswith(pre):
                return val
        return self.is_training


def batch_norm(ctx, y, scope):
    """tf.contrib.layers.batch_norm(center, scale, eps=1e-3, decay, updates_collections=None).

    Train: biased batch variance normalises; EMA `m <- m*d + batch*(1-d)`, with the Bessel-corrected
    variance on the fused path (default; SURVEY Appendix D / E.3, switchable).  Eval: moving stats.
    """
    P = ctx.P
    g, b = P[scope + '/gamma'], P[scope + '/beta']
    flat = y.reshape(-1, y.shape[-1])
    if ctx.training_for(scope):
        mean = flat.mean(0)
        var = ((flat - mean) ** 2).mean(0)
        n = flat.shape[0]
        d = ctx.bn_decay
        var_ema = var * (n / max(n - 1, 1)) if ctx.ema_unbiased else var
        ctx.ema_updates[scope + '/moving_mean'] = (P[scope + '/moving_mean'] * d + mean.detach() * (1 - d))
        ctx.ema_updates[scope + '/moving_variance'] = (P[scope + '/moving_variance'] * d + var_ema.detach() * (1 - d))
    else:
        mean, var = P[scope + '/moving_mean'], P[scope + '/moving_variance']
    return (y - mean) / torch.sqrt(var + BN_EPS) * g + b


def _act(x, activation, ctx=None, scope=None):
    if activation is None:
        return x
    gate = ctx.forced_gates.get(scope) if ctx is not None else None
    if gate is not None and activation in ('relu', 'leaky_relu'):
        gate = torch.as_tensor(gate).reshape(x.shape).to(x.dtype)
        nat = (x.detach() > 0).to(x.dtype)
        gate = torch.where(gate < 0, nat, gate)                  # -1: no decision handed over for this element
        diff = nat != gate
        ctx.flips[scope] = int(diff.sum())
        ctx.margins[scope] = (x.numel(), int(diff.sum()), float(x.detach().abs()[diff].max()) if bool(diff.any()) else 0.0,
                              float(x.detach().abs().max()))
        return x * gate if activation == 'relu' else x * (gate + 0.2 * (1 - gate))
    if activation == 'relu':
        return torch.relu(x)
    if activation == 'leaky_relu':
        return F.leaky_relu(x, 0.2)         # tf.nn.leaky_relu default alpha
    if activation == 'tanh':
        return torch.tanh(x)
    raise ValueError(activation)


def round_bf16(x):
    """x rounded to bfloat16 (round to nearest even), straight-through for autograd."""
    r = x.detach().to(torch.float32).to(torch.bfloat16).to(x.dtype)
    return x + (r - x.detach())


BF16_UNSTORED = ('/conv5', '/conv-reg3-stage1', '/conv-reg4')      # max-pooled layers: statistics / pool from the fp32 accumulators


def conv2d(ctx, x, scope, bn=True, activation='relu', x_global=None):
    """Per-point 1x1 (or [1,D]) VALID conv == (B*N, Cin) x (Cin, Cout) + bias (+BN) (+act).
    x_global (conv6 only): the per-frustum part of the input, (B, Cg), multiplying the LAST Cg rows of the weights -- the same
    arithmetic as the tile + concat of semisup_models.py:107-108, written so that the bf16 emulation can round the per-point
    operands only (the global part is an fp32 fully-connected product in the T3D_BF16 path)."""
    W = ctx.P[scope + '/weights']
    W = W.reshape(-1, W.shape[-1])
    head = ctx.bf16 and not bn                       # conv10: inside the segmentation head, fp32 arithmetic on the stored conv9 output
    Wp, Wg = (W, None) if x_global is None else (W[:W.shape[0] - x_global.shape[1]], W[W.shape[0] - x_global.shape[1]:])
    if ctx.bf16 and not head:
        x, Wp = round_bf16(x), round_bf16(Wp)
    y = x @ Wp + ctx.P[scope + '/biases']
    if x_global is not None:
        y = y + (x_global @ Wg)[:, None, :]
    if ctx.bf16 and not head and not scope.endswith(BF16_UNSTORED):
        y = round_bf16(y)
    if ctx.keep_raw is not None:
        ctx.keep_raw[scope] = y.detach()              # the layer's raw output as the implementation stores it
    if bn:
        y = batch_norm(ctx, y, scope + '/bn')
    return _act(y, activation, ctx, scope)


def fully_connected(ctx, x, scope, bn=False, activation='relu'):
    y = x @ ctx.P[scope + '/weights'] + ctx.P[scope + '/biases']
    if bn:
        y = batch_norm(ctx, y, scope + '/bn')
    return _act(y, activation, ctx, scope)


def dropout(ctx, x, scope, keep_prob):
    """tf.nn.dropout: x * mask / keep in training (mask injected, 0/1), identity otherwise."""
    if not ctx.training_for(scope):
        return x
    m = ctx.dropout_masks[scope]
    return x * m.to(x.dtype) / keep_prob


def max_pool_points(x, ctx=None, scope=None):
    """tf.nn.max_pool ksize [1,N,1,1] == max over the point axis.  (`scope` = the conv layer that feeds the pool: key of
    Ctx.forced_argmax, see there.)"""
    forced = ctx.forced_argmax.get(scope) if ctx is not None else None
    if forced is None:
        return x.max(dim=1).values
    nat = x.detach().argmax(dim=1)
    idx = torch.as_tensor(forced).reshape(nat.shape).to(nat.dtype)
    idx = torch.where(idx < 0, nat, idx)
    ctx.flips[scope + '#argmax'] = int((idx != nat).sum())
    picked = torch.gather(x, 1, idx[:, None, :]).squeeze(1)
    gap = (x.detach().max(dim=1).values - picked.detach())                 # >= 0: how far below the natural maximum the forced row is
    ctx.margins[scope + '#argmax'] = (idx.numel(), int((idx != nat).sum()), float(gap.max()), float(x.detach().abs().max()))
    return picked


# ----------------------------------------------------------------------------------------------
# sub-networks (semisup_models.py)
# ----------------------------------------------------------------------------------------------
def v1_inst_seg(ctx, pc, one_hot_vec, scope='inst_seg', ep=None):
    """semisup_models.py:69-139."""
    B, N, _ = pc.shape
    net = conv2d(ctx, pc, scope + '/conv1')
    net = conv2d(ctx, net, scope + '/conv2')
    point_feat = conv2d(ctx, net, scope + '/conv3')
    net = conv2d(ctx, point_feat, scope + '/conv4')
    net = conv2d(ctx, net, scope + '/conv5')
    global_feat = max_pool_points(net, ctx, scope + '/conv5')          # (B,1024)
    if ep is not None:
        ep['seg_global_feat'] = global_feat
    if one_hot_vec is not None:
        global_feat = torch.cat([global_feat, one_hot_vec], dim=1)
    if ctx.bf16:
        net = conv2d(ctx, point_feat, scope + '/conv6', x_global=global_feat)
    else:
        concat = torch.cat([point_feat, global_feat[:, None, :].expand(B, N, global_feat.shape[1])], dim=2)
        net = conv2d(ctx, concat, scope + '/conv6')
    net = conv2d(ctx, net, scope + '/conv7')
    net = conv2d(ctx, net, scope + '/conv8')
    net = conv2d(ctx, net, scope + '/conv9')
    net = dropout(ctx, net, scope + '/dp1', 0.5)
    logits = conv2d(ctx, net, scope + '/conv10', bn=False, activation=None)
    return logits                                                    # (B,N,2)


def subtract_points_mean(pc, logits, ctx=None):
    """semisup_models.py:145-162.  mask is a hard comparison: no gradient."""
    mask = (logits[:, :, 0:1] < logits[:, :, 1:2]).to(pc.dtype)      # (B,N,1)
    if ctx is not None and ctx.forced_mask is not None:
        forced = torch.as_tensor(ctx.forced_mask).reshape(mask.shape).to(pc.dtype)
        diff = forced != mask
        ctx.flips['mask'] = int(diff.sum())
        dl = (logits[:, :, 0:1] - logits[:, :, 1:2]).detach().abs()
        ctx.margins['mask'] = (mask.numel(), int(diff.sum()), float(dl[diff].max()) if bool(diff.any()) else 0.0,
                               float(logits.detach().abs().max()))
        mask = forced
    mask_count = mask.sum(dim=1, keepdim=True).expand(-1, -1, 3)
    xyz = pc[:, :, 0:3]
    mean = (mask * xyz).sum(dim=1, keepdim=True) / torch.clamp(mask_count, min=1.0)
    return mask, mean, xyz, xyz - mean


def tf_normalize_2D_bboxes(box2D, image_dim):
    """tf_util.py:466-484: image_dim = (rows, cols); [left/cols, top/rows, right/cols, bottom/rows]."""
    rows, cols = image_dim[:, 0], image_dim[:, 1]
    return torch.stack([box2D[:, 0] / cols, box2D[:, 1] / rows, box2D[:, 2] / cols, box2D[:, 3] / rows], dim=1)


def batch_norm_box2D(batch, c, dtype):
    """norm_box2D as the drivers build it (train_semisup.py:240), or None unless USE_NORMALIZED_BOX2D_AS_FEATS
    (semisup_v1_sunrgbd.py:97,145)."""
    if not getattr(c, 'USE_NORMALIZED_BOX2D_AS_FEATS', False):
        return None
    return tf_normalize_2D_bboxes(torch.as_tensor(batch['box2D'], dtype=dtype), torch.as_tensor(batch['img_dim'], dtype=dtype))


def v1_tnet(ctx, xyz_stage1, mask, mask_xyz_mean, one_hot_vec, ep, scope='tnet', norm_box2D=None):
    """semisup_models.py:164-202."""
    net = conv2d(ctx, xyz_stage1, scope + '/conv-reg1-stage1')
    net = conv2d(ctx, net, scope + '/conv-reg2-stage1')
    net = conv2d(ctx, net, scope + '/conv-reg3-stage1')
    net = max_pool_points(net * mask, ctx, scope + '/conv-reg3-stage1')
    ep['tnet_feats'] = net
    if one_hot_vec is not None:
        net = torch.cat([net, one_hot_vec], dim=1)
    if norm_box2D is not None:
        net = torch.cat([net, norm_box2D], dim=1)
    net = fully_connected(ctx, net, scope + '/fc1-stage1', bn=True)
    net = fully_connected(ctx, net, scope + '/fc2-stage1', bn=True)
    c = fully_connected(ctx, net, scope + '/fc3-stage1', activation=None)
    c = c + mask_xyz_mean[:, 0, :]
    ep['stage1_center'] = c
    return c


def _slice_box_heads(out, stage1_center, ep, prefix, dtype):
    """semisup_models.py:265-290 / semisup_v1_sunrgbd.py:203-222."""
    B = out.shape[0]
    NH, NS = NUM_HEADING_BIN, NUM_SIZE_CLUSTER
    mean_dims = torch.tensor(MEAN_DIMS_ARR, dtype=torch.float32).to(dtype)
    ep[prefix + 'center'] = out[:, 0:3] + stage1_center
    ep[prefix + 'heading_scores'] = out[:, 3:3 + NH]
    hrn = out[:, 3 + NH:3 + 2 * NH]
    ep[prefix + 'heading_residuals_normalized'] = hrn
    ep[prefix + 'heading_residuals'] = hrn * (np.pi / NH)
    ep[prefix + 'size_scores'] = out[:, 3 + 2 * NH:3 + 2 * NH + NS]
    srn = out[:, 3 + 2 * NH + NS:3 + 2 * NH + 4 * NS].reshape(B, NS, 3)
    ep[prefix + 'size_residuals_normalized'] = srn
    ep[prefix + 'size_residuals'] = srn * mean_dims[None]
    return (ep[prefix + 'center'], ep[prefix + 'size_scores'], ep[prefix + 'size_residuals'],
            ep[prefix + 'heading_scores'], ep[prefix + 'heading_residuals'])


def v1_box_est(ctx, xyz_submean, stage1_center, mask, one_hot_vec, ep, prefix='', scope='box_est', norm_box2D=None):
    """semisup_models.py:215-291."""
    net = conv2d(ctx, xyz_submean, scope + '/conv-reg1')
    net = conv2d(ctx, net, scope + '/conv-reg2')
    net = conv2d(ctx, net, scope + '/conv-reg3')
    net = conv2d(ctx, net, scope + '/conv-reg4')
    net = max_pool_points(net * mask, ctx, scope + '/conv-reg4')
    ep[prefix + 'feats_lv1'] = net
    if one_hot_vec is not None:
        net = torch.cat([net, one_hot_vec], dim=1)
    if norm_box2D is not None:
        net = torch.cat([net, norm_box2D], dim=1)
    net = fully_connected(ctx, net, scope + '/fc1', bn=True)
    ep[prefix + 'feats_lv2'] = net
    net = fully_connected(ctx, net, scope + '/fc2', bn=True)
    ep[prefix + 'feats_lv3'] = net
    out = fully_connected(ctx, net, scope + '/fc3', activation=None)
    ep[prefix + 'box_params'] = out
    return _slice_box_heads(out, stage1_center, ep, prefix, out.dtype)


def anchor_to_reg(pred_box, dtype):
    """tf_util.py:1001-1041: argmax size/heading bin -> dims=max(anchor+res,1e-5), theta=bin+res."""
    center, size_scores, size_res, heading_scores, heading_res = pred_box
    B = center.shape[0]
    anchors = torch.tensor(MEAN_DIMS_ARR, dtype=torch.float32).to(dtype)
    orient = torch.tensor(ORIENT_ANCHORS, dtype=torch.float32).to(dtype)
    k = torch.argmax(size_scores, dim=1)
    j = torch.argmax(heading_scores, dim=1)
    ar = torch.arange(B)
    dims = torch.clamp(anchors[k] + size_res[ar, k], min=1e-5)
    theta = orient[j] + heading_res[ar, j]
    return center, dims, theta


def get_semi_model_backbone(ctx, pc, one_hot_vec, use_one_hot=False, norm_box2D=None):
    """semisup_v1_sunrgbd.py:81-130 (SEMI_MODEL A)."""
    ep = {'point_cloud': pc, 'class_one_hot': one_hot_vec,
          'class_ids': torch.argmax(one_hot_vec, dim=1).to(torch.int32)}
    oh = one_hot_vec if use_one_hot else None
    logits = v1_inst_seg(ctx, pc, oh, 'inst_seg', ep)
    ep['logits'] = logits
    ep['soft_mask'] = torch.softmax(logits, dim=-1)[:, :, 1]
    mask, mean, xyz, xyz1 = subtract_points_mean(pc, logits, ctx)
    ep['mask'] = mask
    ep['mask_xyz_mean'] = mean
    s1 = v1_tnet(ctx, xyz1, mask, mean, oh, ep, 'tnet', norm_box2D=norm_box2D)
    xyz2 = xyz - s1[:, None, :]
    pred_box = v1_box_est(ctx, xyz2, s1, mask, oh, ep, '', 'box_est', norm_box2D=norm_box2D)
    ep['S_pred_box'] = pred_box
    ep['S_pred_box_reg'] = anchor_to_reg(pred_box, pc.dtype)
    return (logits, pred_box), ep


def mlps_with_dropout(ctx, x, scope, n_layers, activations, keep_probs):
    """semisup_models.py:44-63: hidden = FC+BN+act then dropout; last = FC, no BN."""
    net = x
    for i in range(n_layers):
        last = i == n_layers - 1
        net = fully_connected(ctx, net, '%s/fc%d' % (scope, i), bn=not last, activation=activations[i])
        if not last:
            net = dropout(ctx, net, '%s/dp%d' % (scope, i), keep_probs[i])
    return net


def get_semi_model_final(ctx, pc, one_hot_vec, use_one_hot, c, norm_box2D=None, oracle_mask=None):
    """semisup_v1_sunrgbd.py:132-230 (SEMI_MODEL F).  oracle_mask [B,N] (161-162): logits = stack([1 - m, m], axis=2)."""
    ep = {'point_cloud': pc, 'class_one_hot': one_hot_vec,
          'class_ids': torch.argmax(one_hot_vec, dim=1).to(torch.int32)}
    p = 'class_agnostic/'
    logits = v1_inst_seg(ctx, pc, None, p + 'inst_seg', ep)
    if oracle_mask is not None:
        om = torch.as_tensor(oracle_mask).to(pc.dtype)
        logits = torch.stack([1 - om, om], dim=2)
    ep['logits'] = logits
    mask, mean, xyz, xyz1 = subtract_points_mean(pc, logits, ctx)
    ep['mask'] = mask
    ep['mask_xyz_mean'] = mean
    s1 = v1_tnet(ctx, xyz1, mask, mean, None, ep, p + 'tnet', norm_box2D=norm_box2D)
    xyz2 = xyz - s1[:, None, :]
    W_pred_box = v1_box_est(ctx, xyz2, s1, mask, None, ep, '', p + 'box_est', norm_box2D=norm_box2D)
    feat = ep['feats_lv1']
    if use_one_hot:
        feat = torch.cat([feat, one_hot_vec], dim=1)
    act = 'leaky_relu' if c.SEMI_ADV_LEAKY_RELU else 'relu'
    last = 'tanh' if c.SEMI_ADV_TANH_FOR_LAST_LAYER_OF_G else act
    dp = c.SEMI_ADV_DROPOUTS_FOR_G
    out = mlps_with_dropout(ctx, feat, 'class_dependent/box_refine', 3, [act, last, None], [dp, dp, None])
    ep['F_box_params'] = out
    F_pred_box = _slice_box_heads(out, s1, ep, 'F_', out.dtype)
    ep['F_pred_box_reg'] = anchor_to_reg(F_pred_box, pc.dtype)
    return (logits, W_pred_box, F_pred_box), ep


# ----------------------------------------------------------------------------------------------
# Box-PC Fit net (tf_util.py:764-795, 893-955; semisup_models.py:326-398; boxpc_sunrgbd.py)
# ----------------------------------------------------------------------------------------------
def box_pc_representation(box_reg, pc):
    """Six signed point-to-face distances of every point to the (centre, (l,w,h), theta) box,
    concatenated to the raw point channels (tf_util.py:764-795, 893-942)."""
    center, dims, theta = box_reg
    l, w, h = dims[:, 0], dims[:, 1], dims[:, 2]
    s, c = torch.sin(theta), torch.cos(theta)
    z = torch.zeros_like(s)
    o = torch.ones_like(s)
    R = torch.stack([torch.stack([c, z, s], 1), torch.stack([z, o, z], 1), torch.stack([-s, z, c], 1)], 1)  # (B,3,3)
    pts = torch.stack([torch.stack([l / 2, -l / 2, z, z, z, z], 1),
                       torch.stack([z, z, h / 2, -h / 2, z, z], 1),
                       torch.stack([z, z, z, z, w / 2, -w / 2], 1)], 1)                                     # (B,3,6)
    nrm = torch.tensor([[-1., 1., 0., 0., 0., 0.], [0., 0., -1., 1., 0., 0.], [0., 0., 0., 0., -1., 1.]],
                       dtype=pc.dtype)
    spts = torch.matmul(R, pts).transpose(1, 2)                     # (B,6,3)
    snrm = torch.matmul(R, nrm[None].expand(R.shape[0], 3, 6)).transpose(1, 2)
    t = pc[:, :, 0:3] - center[:, None, :]                           # (B,N,3)
    ray = t[:, :, None, :] - spts[:, None, :, :]                     # (B,N,6,3)
    d = (snrm[:, None, :, :] * ray).sum(-1)                          # (B,N,6)
    return torch.cat([pc, d], dim=2)


def boxpc_get_model(ctx, box_reg, pc, one_hot_vec, use_one_hot_vec, c, scope_prefix=''):
    """boxpc_sunrgbd.py:56-100 + semisup_models.py:326-398 (representation 'A')."""
    ep = {'class_ids': torch.argmax(one_hot_vec, dim=1).to(torch.int32)}
    sc = scope_prefix + 'box_pc_mask_model'
    rep = box_pc_representation(box_reg, pc)
    ep['box_pc_rep'] = rep
    net = conv2d(ctx, rep, sc + '/conv-reg1')
    net = conv2d(ctx, net, sc + '/conv-reg2')
    net = conv2d(ctx, net, sc + '/conv-reg3')
    net = conv2d(ctx, net, sc + '/conv-reg4')
    net = max_pool_points(net, ctx, sc + '/conv-reg4')
    if use_one_hot_vec:
        net = torch.cat([net, one_hot_vec], dim=1)
    f1 = net
    net = fully_connected(ctx, net, sc + '/fc1', bn=True)
    f2 = net
    net = dropout(ctx, net, sc + '/dp1', 0.7)
    net = fully_connected(ctx, net, sc + '/fc2', bn=True)
    f3 = net
    net = dropout(ctx, net, sc + '/dp2', 0.7)
    out = fully_connected(ctx, net, sc + '/fc3', activation=None)
    ep['boxpc_out'] = out
    ep['boxpc_feats_dict'] = {'box_pc_mask_model_feats_lv1': f1, 'box_pc_mask_model_feats_lv2': f2,
                              'box_pc_mask_model_feats_lv3': f3}
    fit_logits = out[:, -2:]
    p1 = torch.softmax(fit_logits, dim=-1)[:, 1]
    ep['boxpc_fit_logits'] = fit_logits
    ep['pred_boxpc_fit'] = (p1 > 0.5).to(torch.int32)
    lw = p1.detach() if c.BOXPC_STOP_GRAD_OF_CLS_VIA_DELTA else p1
    ep['logits_for_weigh'] = lw
    dc, ds, da = out[:, 0:3], out[:, 3:6], out[:, 6]
    if c.BOXPC_WEIGH_DELTA_PRED_BY_CLS_CONF:
        wd = 1.0 - lw
        dc, ds, da = dc * wd[:, None], ds * wd[:, None], da * wd
    ep['boxpc_delta_center'], ep['boxpc_delta_size'], ep['boxpc_delta_angle'] = dc, ds, da
    return (fit_logits, (dc, ds, da)), ep


def tf_huber(labels, pred, delta=1.0):
    """tf.losses.huber_loss elementwise (reduction NONE)."""
    e = (pred - labels).abs()
    q = torch.clamp(e, max=delta)
    return 0.5 * q * q + delta * (e - q)


def boxpc_get_loss(pred, labels, ep, c, reduce_loss=True):
    """boxpc_sunrgbd.py:106-193."""
    logits, (dc, ds, da) = pred
    y_iou, (ydc, yds, yda) = labels
    cls = (y_iou > c.BOXPC_FIT_BOUNDS[0]).long()
    cls_losses = F.cross_entropy(logits, cls, reduction='none')
    if c.BOXPC_DELTA_LOSS_TYPE == 'huber':                 # boxpc_sunrgbd.py:151-157
        elem = tf_huber
    elif c.BOXPC_DELTA_LOSS_TYPE == 'mse':                 # 158-164: tf.losses.mean_squared_error, reduction NONE = squared difference
        elem = lambda labels, pred: (pred - labels) ** 2
    else:
        raise ValueError(c.BOXPC_DELTA_LOSS_TYPE)
    assert not (c.BOXPC_WEIGH_DELTA_LOSS_BY_CLS_CONF and c.BOXPC_WEIGH_DELTA_LOSS_BY_CLS_GT)     # 166
    lc = elem(ydc, dc).mean(1)
    ls = elem(yds, ds).mean(1)
    la = elem(yda, da)
    wl = 1.0
    if c.BOXPC_WEIGH_DELTA_LOSS_BY_CLS_CONF:
        wl = 1.0 - ep['logits_for_weigh']
    if c.BOXPC_WEIGH_DELTA_LOSS_BY_CLS_GT:
        wl = 1.0 - y_iou
    delta_losses = (c.BOXPC_WEIGHT_DELTA_CENTER_PERCENT * lc * wl + c.BOXPC_WEIGHT_DELTA_SIZE_PERCENT * ls * wl
                    + c.BOXPC_WEIGHT_DELTA_ANGLE_PERCENT * la * wl)
    ep['boxpc_cls_losses'] = cls_losses
    ep['boxpc_delta_losses'] = delta_losses
    total = c.BOXPC_WEIGHT_CLS * cls_losses + c.BOXPC_WEIGHT_DELTA * delta_losses
    return total.mean() if reduce_loss else total


def convert_raw_y_box_to_reg_format(y_box, dtype):
    """boxpc_sunrgbd.py:206-230: one-hot the GT bins, then the same anchor->reg op."""
    yc, yoc, yor, ydc, ydr = y_box
    anchors = torch.tensor(MEAN_DIMS_ARR, dtype=torch.float32).to(dtype)
    orient = torch.tensor(ORIENT_ANCHORS, dtype=torch.float32).to(dtype)
    dims = torch.clamp(anchors[ydc.long()] + ydr, min=1e-5)
    theta = orient[yoc.long()] + yor
    return yc, dims, theta


# ----------------------------------------------------------------------------------------------
# losses (semisup_v1_sunrgbd.py:248-564, model_util.py:94-167, weak_losses.py:267-291)
# ----------------------------------------------------------------------------------------------
def huber_loss(error, delta):
    """semisup_v1_sunrgbd.py:555-564."""
    a = error.abs()
    q = torch.clamp(a, max=delta)
    return 0.5 * q ** 2 + delta * (a - q)


def box3d_corners_helper(centers, headings, sizes):
    """model_util.py:94-119.  (N,3),(N,),(N,3) -> (N,8,3)."""
    l, w, h = sizes[:, 0:1], sizes[:, 1:2], sizes[:, 2:3]
    xc = torch.cat([l / 2, l / 2, -l / 2, -l / 2, l / 2, l / 2, -l / 2, -l / 2], 1)
    yc = torch.cat([h / 2, h / 2, h / 2, h / 2, -h / 2, -h / 2, -h / 2, -h / 2], 1)
    zc = torch.cat([w / 2, -w / 2, -w / 2, w / 2, w / 2, -w / 2, -w / 2, w / 2], 1)
    corners = torch.stack([xc, yc, zc], 1)                          # (N,3,8)
    c, s = torch.cos(headings), torch.sin(headings)
    z, o = torch.zeros_like(c), torch.ones_like(c)
    R = torch.stack([torch.stack([c, z, s], 1), torch.stack([z, o, z], 1), torch.stack([-s, z, c], 1)], 1)
    out = torch.matmul(R, corners) + centers[:, :, None]
    return out.transpose(1, 2)


def get_strong_loss(pred, labels, ep, c, prefix=''):
    """semisup_v1_sunrgbd.py:423-553.  Returns per-frustum (mask_losses, box_losses)."""
    pred_seg, _ = pred
    y_seg, y_center, y_orient_cls, y_orient_reg, y_dims_cls, y_dims_reg = labels
    dtype = pred_seg.dtype
    B, N, _ = pred_seg.shape
    NH, NS = NUM_HEADING_BIN, NUM_SIZE_CLUSTER
    mean_dims = torch.tensor(MEAN_DIMS_ARR, dtype=torch.float32).to(dtype)

    mask_losses = F.cross_entropy(pred_seg.reshape(B * N, 2), y_seg.reshape(-1).long(),
                                  reduction='none').reshape(B, N).mean(1)
    center_dist = torch.norm(y_center - ep[prefix + 'center'], dim=-1)
    center_losses = huber_loss(center_dist, 2.0)
    s1_dist = torch.norm(y_center - ep['stage1_center'], dim=-1)
    s1_losses = huber_loss(s1_dist, 1.0)

    hcls_losses = F.cross_entropy(ep[prefix + 'heading_scores'], y_orient_cls.long(), reduction='none')
    hoh = F.one_hot(y_orient_cls.long(), NH).to(dtype)
    hres_label = y_orient_reg / (np.pi / NH)
    hres_losses = huber_loss((ep[prefix + 'heading_residuals_normalized'] * hoh).sum(1) - hres_label, 1.0)

    scls_losses = F.cross_entropy(ep[prefix + 'size_scores'], y_dims_cls.long(), reduction='none')
    soh = F.one_hot(y_dims_cls.long(), NS).to(dtype)
    pred_srn = (ep[prefix + 'size_residuals_normalized'] * soh[:, :, None]).sum(1)
    mean_size_label = (soh[:, :, None] * mean_dims[None]).sum(1)
    srn_label = y_dims_reg / mean_size_label
    sres_losses = huber_loss(torch.norm(srn_label - pred_srn, dim=-1), 1.0)

    # Corner loss.  get_box3d_corners_sunrgbd adds the size residual TWICE (model_util.py:158-159).
    center = ep[prefix + 'center']
    hres = ep[prefix + 'heading_residuals']
    sres = ep[prefix + 'size_residuals']
    bins = torch.tensor(ORIENT_ANCHORS, dtype=torch.float32).to(dtype)
    headings = hres + bins[None]                                    # (B,NH)
    sizes = (mean_dims[None] + sres) + sres                         # (B,NS,3)
    sizes_t = sizes[:, None].expand(B, NH, NS, 3).reshape(-1, 3)
    head_t = headings[:, :, None].expand(B, NH, NS).reshape(-1)
    cent_t = center[:, None, None, :].expand(B, NH, NS, 3).reshape(-1, 3)
    corners = box3d_corners_helper(cent_t, head_t, sizes_t).reshape(B, NH, NS, 8, 3)
    gt_mask = hoh[:, :, None] * soh[:, None, :]
    corners_pred = (gt_mask[:, :, :, None, None] * corners).sum(dim=(1, 2))       # (B,8,3)
    heading_label = ((y_orient_reg[:, None] + bins[None]) * hoh).sum(1)
    size_label = (soh[:, :, None] * (mean_dims[None] + y_dims_reg[:, None, :])).sum(1)
    cg = box3d_corners_helper(y_center, heading_label, size_label)
    cgf = box3d_corners_helper(y_center, heading_label + np.pi, size_label)
    cdist = torch.minimum(torch.norm(corners_pred - cg, dim=-1), torch.norm(corners_pred - cgf, dim=-1))
    corner_losses = huber_loss(cdist, 1.0).mean(1)

    terms = dict(mask=mask_losses, center=center_losses, stage1=s1_losses, hcls=hcls_losses,
                 hres=hres_losses, scls=scls_losses, sres=sres_losses, corner=corner_losses)
    ep[prefix + 'loss_terms'] = terms
    mask_l = c.STRONG_WEIGHT_CROSS_ENTROPY * mask_losses
    box_l = c.STRONG_BOX_MULTIPLER * (c.STRONG_WEIGHT_CENTER * center_losses
                                      + c.STRONG_WEIGHT_ORIENT_CLS * hcls_losses
                                      + c.STRONG_WEIGHT_DIMS_CLS * scls_losses
                                      + c.STRONG_WEIGHT_ORIENT_REG * hres_losses
                                      + c.STRONG_WEIGHT_DIMS_REG * sres_losses
                                      + c.STRONG_WEIGHT_TNET_CENTER * s1_losses) \
        + c.STRONG_WEIGHT_CORNER * corner_losses
    return mask_l, box_l


WEAK_DEFAULTS = dict(       # models/config.py:111-162 of the reference
    WEAK_TRAIN_SEG_W_SURFACE=False, WEAK_TRAIN_BOX_W_REPROJECTION=[True, True, True], WEAK_TRAIN_BOX_W_SURFACE=[True, False, True],
    WEAK_REPROJECTION_USE_SOFTMAX_PROJ=False, WEAK_REPROJECTION_SOFTMAX_SCALE=10., WEAK_REPROJECTION_ONLY_ON_2D_CLS=False,
    WEAK_REPROJECTION_CLIP_LOWERB_LOSS=True, WEAK_REPROJECTION_CLIP_PRED_BOX=False, WEAK_REPROJECTION_LOSS_TYPE='huber',
    WEAK_REPROJECTION_DILATE_FACTOR=1.5, WEAK_SURFACE_MARGIN=0., WEAK_SURFACE_LOSS_WT_FOR_INNER_PTS=0.8,
    WEAK_SURFACE_LOSS_SCALE_DIMS=0.9)


def weak_flag(c, name):
    return getattr(c, name, WEAK_DEFAULTS[name])


def get_semi_loss_backbone(pred, labels, ep, c):
    """semisup_v1_sunrgbd.py:256-321.  The weak reprojection / surface losses (oracle/ref_weak.py) are evaluated when their weight
    is non-zero; their camera inputs (Rtilt, K, rot_frust, box2D, img_dim) come in through ep['weak_inputs']."""
    y_seg, y_center, yoc, yor, ydc, ydr, is2d = labels
    mask_l, box_l = get_strong_loss(pred, (y_seg, y_center, yoc, yor, ydc, ydr), ep, c)
    w3 = (1 - is2d).to(mask_l.dtype)
    total = w3 * (mask_l + box_l)
    if c.WEAK_WEIGHT_REPROJECTION != 0 or c.WEAK_WEIGHT_SURFACE != 0:
        from . import ref_weak as W
        wi = ep['weak_inputs']
        box = ep['S_pred_box_reg']
        if ep.get('forced_S_box') is not None:      # flip-aware checks: the branches of the weak losses at the box the product predicted
            box = tuple(o + (torch.as_tensor(f, dtype=o.dtype).reshape(o.shape) - o).detach() for o, f in zip(box, ep['forced_S_box']))
        reproj = W.get_reprojection_loss(
            box, wi['box2D'], wi['Rtilt'], wi['K'], wi['img_dim'], wi['rot_frust'],
            weak_flag(c, 'WEAK_REPROJECTION_USE_SOFTMAX_PROJ'), weak_flag(c, 'WEAK_REPROJECTION_SOFTMAX_SCALE'),
            weak_flag(c, 'WEAK_REPROJECTION_DILATE_FACTOR'), weak_flag(c, 'WEAK_REPROJECTION_CLIP_LOWERB_LOSS'),
            weak_flag(c, 'WEAK_REPROJECTION_CLIP_PRED_BOX'), weak_flag(c, 'WEAK_REPROJECTION_LOSS_TYPE'),
            weak_flag(c, 'WEAK_TRAIN_BOX_W_REPROJECTION'), ep=ep)
        surf = W.get_surface_loss(box, ep['point_cloud'][:, :, 0:3], ep['soft_mask'], weak_flag(c, 'WEAK_SURFACE_MARGIN'),
                                  weak_flag(c, 'WEAK_SURFACE_LOSS_SCALE_DIMS'), weak_flag(c, 'WEAK_TRAIN_BOX_W_SURFACE'), ep=ep)
        weak = c.WEAK_WEIGHT_REPROJECTION * reproj + c.WEAK_WEIGHT_SURFACE * surf
        ep['reprojection_loss'], ep['surface_loss'], ep['weak_loss_fns'] = reproj, surf, weak
        total = total + is2d.to(mask_l.dtype) * (weak * c.SEMI_MULTIPLIER_FOR_WEAK_LOSS)
    ep['total_losses'] = total
    return total.mean()


def intraclass_variance_loss(dims_reg, class_ids, train_classes):
    """weak_losses.py:267-291 (huber): per trained class, tf.losses.huber_loss(mean over elements,
    0 for an empty partition) of the dims to the stop-gradient class-batch mean; mean over the
    trained classes."""
    losses = []
    for i, on in enumerate(train_classes):
        if not on:
            continue
        sel = dims_reg[class_ids.long() == i]
        if sel.shape[0] == 0:
            losses.append(torch.zeros((), dtype=dims_reg.dtype))
            continue
        m = sel.mean(0).detach()
        losses.append(tf_huber(m[None].expand_as(sel), sel).mean())
    return torch.stack(losses).mean()


def get_semi_loss_final(pred, labels, ep, c):
    """semisup_v1_sunrgbd.py:323-421 (SEMI_MODEL F)."""
    pred_seg, _, F_pred_box = pred
    y_seg, y_center, yoc, yor, ydc, ydr, is2d = labels
    mask_l, box_l = get_strong_loss((pred_seg, F_pred_box), (y_seg, y_center, yoc, yor, ydc, ydr), ep, c,
                                    prefix='F_')
    w3 = (1 - is2d).to(mask_l.dtype)
    denom = w3.sum() + 1e-3
    strong = (mask_l * w3).sum() / denom + (box_l * w3).sum() / denom
    weak = torch.zeros((), dtype=mask_l.dtype)
    if c.WEAK_WEIGHT_INTRACLASSVAR != 0:
        _, F_dims, _ = ep['F_pred_box_reg']
        icv = intraclass_variance_loss(F_dims, ep['class_ids'], ep['intraclsdims_train_classes'])
        ep['intraclass_variance_loss'] = icv
        weak = weak + c.WEAK_WEIGHT_INTRACLASSVAR * icv
    if getattr(c, 'WEAK_WEIGHT_INACTIVE_VOLUME', 0) != 0:       # semisup_v1_sunrgbd.py:348-359
        from . import ref_weak as W
        _, F_dims, _ = ep['F_pred_box_reg']
        iv = W.get_inactive_volume_loss_v1(F_dims, ep['class_ids'], ep['inactive_vol_train_classes'],
                                           torch.as_tensor(c.WEAK_INACTIVE_VOL_LOSS_MARGINS, dtype=F_dims.dtype))
        ep['inactive_vol_loss'] = iv
        weak = weak + c.WEAK_WEIGHT_INACTIVE_VOLUME * iv
    if c.WEAK_WEIGHT_REPROJECTION != 0:                         # semisup_v1_sunrgbd.py:373-392
        from . import ref_weak as W
        wi = ep['weak_inputs']
        box = ep['F_pred_box_reg']
        if ep.get('forced_S_box') is not None:                  # see get_semi_loss_backbone
            box = tuple(o + (torch.as_tensor(f, dtype=o.dtype).reshape(o.shape) - o).detach() for o, f in zip(box, ep['forced_S_box']))
        reproj = W.get_reprojection_loss(
            box, wi['box2D'], wi['Rtilt'], wi['K'], wi['img_dim'], wi['rot_frust'],
            weak_flag(c, 'WEAK_REPROJECTION_USE_SOFTMAX_PROJ'), weak_flag(c, 'WEAK_REPROJECTION_SOFTMAX_SCALE'),
            weak_flag(c, 'WEAK_REPROJECTION_DILATE_FACTOR'), weak_flag(c, 'WEAK_REPROJECTION_CLIP_LOWERB_LOSS'),
            weak_flag(c, 'WEAK_REPROJECTION_CLIP_PRED_BOX'), weak_flag(c, 'WEAK_REPROJECTION_LOSS_TYPE'),
            weak_flag(c, 'WEAK_TRAIN_BOX_W_REPROJECTION'), ep=ep)
        if weak_flag(c, 'WEAK_REPROJECTION_ONLY_ON_2D_CLS'):
            reproj = reproj * is2d.to(reproj.dtype)
        ep['reprojection_loss'] = reproj
        weak = weak + (c.WEAK_WEIGHT_REPROJECTION * reproj).mean()      # weak_loss = reduce_mean(scalar terms + per-sample term)
    total = strong + c.SEMI_MULTIPLIER_FOR_WEAK_LOSS * weak
    if c.SEMI_WEIGHT_BOXPC_FIT_LOSS != 0:
        fit = -torch.log(0.01 + ep['boxpc_fit_prob'])
        if c.SEMI_BOXPC_FIT_ONLY_ON_2D_CLS:
            fit = fit * is2d.to(fit.dtype)
        total = total + c.SEMI_WEIGHT_BOXPC_FIT_LOSS * fit.mean()
    return total


# ----------------------------------------------------------------------------------------------
# optimiser + schedules (train_semisup.py:127-145, 229-231; tf.train.AdamOptimizer)
# ----------------------------------------------------------------------------------------------
def learning_rate(step, batch_size, base=1e-3, decay_step=800000, decay_rate=0.5):
    """exponential_decay(staircase=True); the clip at train_semisup.py:134 is a no-op (typo)."""
    return base * decay_rate ** math.floor(step * batch_size / decay_step)


def bn_decay(step, batch_size, decay_step=800000.0):
    """train_semisup.py:137-145: min(0.99, 1 - 0.5*0.5^floor(step*B/decay_step))."""
    return min(0.99, 1 - 0.5 * 0.5 ** math.floor(step * batch_size / decay_step))


def adam_tf_step(P, grads, m, v, t, lr, beta1=0.9, beta2=0.999, eps=1e-8):
    """TF form: lr_t = lr*sqrt(1-b2^t)/(1-b1^t); w -= lr_t*m/(sqrt(v)+eps); t starts at 1."""
    lr_t = lr * math.sqrt(1 - beta2 ** t) / (1 - beta1 ** t)
    for k, g in grads.items():
        m[k] = beta1 * m[k] + (1 - beta1) * g
        v[k] = beta2 * v[k] + (1 - beta2) * g * g
        P[k] = P[k] - lr_t * m[k] / (torch.sqrt(v[k]) + eps)


def momentum_tf_step(P, grads, accum, lr, momentum=0.9):
    """tf.train.MomentumOptimizer(learning_rate, momentum) of `--optimizer momentum` (train_semisup.py:226-228), use_nesterov=False
    (TF's ApplyMomentum kernel): accum = momentum * accum + g;  w -= lr * accum."""
    for k, g in grads.items():
        accum[k] = momentum * accum[k] + g
        P[k] = P[k] - lr * accum[k]


# ----------------------------------------------------------------------------------------------
# one full training step (what bench.py's cpu_baseline leg times and the fixtures freeze)
# ----------------------------------------------------------------------------------------------
def default_config(**over):
    c = SimpleNamespace(
        SEMI_MODEL='A', WEAK_WEIGHT_REPROJECTION=0.0, WEAK_WEIGHT_SURFACE=0.0, WEAK_WEIGHT_INACTIVE_VOLUME=0.0,
        WEAK_WEIGHT_INTRACLASSVAR=0.0, SEMI_MULTIPLIER_FOR_WEAK_LOSS=1.0, SEMI_WEIGHT_BOXPC_FIT_LOSS=1.0,
        SEMI_BOXPC_FIT_ONLY_ON_2D_CLS=False, SEMI_ADV_LEAKY_RELU=True, SEMI_ADV_TANH_FOR_LAST_LAYER_OF_G=True,
        SEMI_ADV_DROPOUTS_FOR_G=0.5, SEMI_REFINE_USING_BOXPC_DELTA_NUM=1, SEMI_BOXPC_MIN_FIT_LOSS_AFT_REFINE=False,
        SEMI_WEIGH_BOXPC_DELTA_DURING_TEST=False,
        STRONG_WEIGHT_CROSS_ENTROPY=1.0, STRONG_BOX_MULTIPLER=0.1, STRONG_WEIGHT_CENTER=1.0,
        STRONG_WEIGHT_ORIENT_CLS=1.0, STRONG_WEIGHT_ORIENT_REG=20.0, STRONG_WEIGHT_DIMS_CLS=1.0,
        STRONG_WEIGHT_DIMS_REG=20.0, STRONG_WEIGHT_TNET_CENTER=1.0, STRONG_WEIGHT_CORNER=1.0,
        BOXPC_FIT_BOUNDS=[0.7, 1.0], BOXPC_STOP_GRAD_OF_CLS_VIA_DELTA=True,
        BOXPC_WEIGH_DELTA_PRED_BY_CLS_CONF=False, BOXPC_WEIGH_DELTA_LOSS_BY_CLS_CONF=False,
        BOXPC_WEIGH_DELTA_LOSS_BY_CLS_GT=False, BOXPC_WEIGHT_CLS=1.0, BOXPC_WEIGHT_DELTA=1.0,
        BOXPC_WEIGHT_DELTA_CENTER_PERCENT=0.34, BOXPC_WEIGHT_DELTA_SIZE_PERCENT=0.33,
        BOXPC_WEIGHT_DELTA_ANGLE_PERCENT=0.33, BOXPC_DELTA_LOSS_TYPE='huber', BOX_PC_MASK_REPRESENTATION='A',
        USE_NORMALIZED_BOX2D_AS_FEATS=False)
    for k, val in over.items():
        setattr(c, k, val)
    return c


def _labels_to_torch(batch, dtype):
    return (torch.as_tensor(batch['y_seg']), torch.as_tensor(batch['y_center'], dtype=dtype),
            torch.as_tensor(batch['y_orient_cls']), torch.as_tensor(batch['y_orient_reg'], dtype=dtype),
            torch.as_tensor(batch['y_dims_cls']), torch.as_tensor(batch['y_dims_reg'], dtype=dtype),
            torch.as_tensor(batch['is_data_2D']))


def _apply_forced(ctx, forced):
    """forced = {'gates': {scope: 0/1 array}, 'argmax': {scope: (B, C) int array}} -- see Ctx.forced_gates."""
    if forced:
        ctx.forced_gates = dict(forced.get('gates', {}))
        ctx.forced_argmax = dict(forced.get('argmax', {}))
        ctx.forced_mask = forced.get('mask')
        ctx.bf16 = bool(forced.get('bf16', False))
        ctx.keep_raw = forced.get('keep_raw')


def model_a_forward_backward(P, batch, c, bn_decay_val=0.5, dtype=torch.float64, use_one_hot=False,
                             is_training=True, want_grads=True, forced=None):
    """fwd (+bwd) of SEMI_MODEL A on one batch.  Returns (loss, end_points, grads, ema_updates)."""
    names = trainable_names(P)
    Pl = {k: (val.detach().to(dtype).requires_grad_(k in names and want_grads)) for k, val in P.items()}
    masks = {k: torch.as_tensor(val) for k, val in batch.get('dropout_masks', {}).items()}
    ctx = Ctx(Pl, is_training=is_training, bn_decay=bn_decay_val, dropout_masks=masks)
    _apply_forced(ctx, forced)
    pc = torch.as_tensor(batch['pc'], dtype=dtype)
    oh = torch.as_tensor(batch['one_hot_vec'], dtype=dtype)
    pred, ep = get_semi_model_backbone(ctx, pc, oh, use_one_hot, norm_box2D=batch_norm_box2D(batch, c, dtype))
    if 'Rtilt' in batch:
        ep['weak_inputs'] = {k: torch.as_tensor(batch[k], dtype=dtype) for k in ('Rtilt', 'K', 'rot_frust', 'box2D', 'img_dim')}
    if forced and forced.get('S_box') is not None:
        ep['forced_S_box'] = forced['S_box']
    loss = get_semi_loss_backbone(pred, _labels_to_torch(batch, dtype), ep, c)
    grads = {}
    if want_grads:
        gl = torch.autograd.grad(loss, [Pl[k] for k in names], allow_unused=True)
        grads = {k: (g if g is not None else torch.zeros_like(Pl[k])) for k, g in zip(names, gl)}
    ep['__flips__'] = dict(ctx.flips)
    ep['__margins__'] = dict(ctx.margins)
    return loss, ep, grads, ctx.ema_updates


def boxpc_forward_backward(P, batch, c, bn_decay_val=0.5, dtype=torch.float64, is_training=True,
                           want_grads=True, forced=None):
    """fwd (+bwd) of the Box-PC Fit net (train_boxpc.py path) on one batch."""
    names = trainable_names(P)
    Pl = {k: (val.detach().to(dtype).requires_grad_(k in names and want_grads)) for k, val in P.items()}
    masks = {k: torch.as_tensor(val) for k, val in batch.get('dropout_masks', {}).items()}
    ctx = Ctx(Pl, is_training=is_training, bn_decay=bn_decay_val, dropout_masks=masks)
    _apply_forced(ctx, forced)
    pc = torch.as_tensor(batch['pc'], dtype=dtype)
    oh = torch.as_tensor(batch['one_hot_vec'], dtype=dtype)
    y_box = (torch.as_tensor(batch['y_center'], dtype=dtype), torch.as_tensor(batch['y_orient_cls']),
             torch.as_tensor(batch['y_orient_reg'], dtype=dtype), torch.as_tensor(batch['y_dims_cls']),
             torch.as_tensor(batch['y_dims_reg'], dtype=dtype))
    box_reg = convert_raw_y_box_to_reg_format(y_box, dtype)
    pred, ep = boxpc_get_model(ctx, box_reg, pc, oh, False, c)
    labels = (torch.as_tensor(batch['y_box_iou'], dtype=dtype),
              (torch.as_tensor(batch['y_center_delta'], dtype=dtype),
               torch.as_tensor(batch['y_dims_delta'], dtype=dtype),
               torch.as_tensor(batch['y_orient_delta'], dtype=dtype)))
    loss = boxpc_get_loss(pred, labels, ep, c)
    grads = {}
    if want_grads:
        gl = torch.autograd.grad(loss, [Pl[k] for k in names], allow_unused=True)
        grads = {k: (g if g is not None else torch.zeros_like(Pl[k])) for k, g in zip(names, gl)}
    ep['__flips__'] = dict(ctx.flips)
    ep['__margins__'] = dict(ctx.margins)
    return loss, ep, grads, ctx.ema_updates


def stage_c_forward_backward(P, batch, c, train_classes, bn_decay_val=0.5, dtype=torch.float64, use_one_hot=True,
                             var_prefixes=('class_dependent', 'class_agnostic/tnet', 'class_agnostic/box'), want_grads=True, forced=None):
    """One stage-c step (train_semisup_adv.py:308-422, SEMI_MODEL F): class-agnostic nets + box_refine, the frozen
    Box-PC net (`D_boxpc_branch/`, is_training_D = False when SEMI_TRAIN_BOXPC_MODEL = 0) applied to F_pred_box_reg,
    get_semi_loss_final, gradients w.r.t. the var_list (regex-prefix semantics of get_scope_vars)."""
    names = [k for k in trainable_names(P) if any(k.startswith(p) for p in var_prefixes)]
    Pl = {k: (val.detach().to(dtype).requires_grad_(k in names and want_grads)) for k, val in P.items()}
    masks = {k: torch.as_tensor(val) for k, val in batch.get('dropout_masks', {}).items()}
    ctx = Ctx(SharedParams(Pl), is_training=True, bn_decay=bn_decay_val, dropout_masks=masks)
    ctx.is_training_override['D_boxpc_branch'] = False           # is_training_D (train_semisup_adv.py:337), every evaluation
    _apply_forced(ctx, forced)
    pc = torch.as_tensor(batch['pc'], dtype=dtype)
    oh = torch.as_tensor(batch['one_hot_vec'], dtype=dtype)
    pred, ep = get_semi_model_final(ctx, pc, oh, use_one_hot, c, norm_box2D=batch_norm_box2D(batch, c, dtype))
    _, ep_b = boxpc_get_model(ctx, ep['F_pred_box_reg'], pc, oh, False, c, scope_prefix='D_boxpc_branch/')
    fit_prob = torch.softmax(ep_b['boxpc_fit_logits'], dim=-1)[:, 1]
    ep['boxpc_out'] = ep_b['boxpc_out']
    ep['box_pc_rep'] = ep_b['box_pc_rep']
    # the refinement loop of the training graph (train_semisup_adv.py:362-399).  Its first evaluation sees the same box as the one
    # above (same variables, same input: the same values and the same gradient), so that evaluation is shared here.
    cur = ep['F_pred_box_reg']
    tot_c, tot_s, tot_a = torch.zeros_like(cur[0]), torch.zeros_like(cur[1]), torch.zeros_like(cur[2])
    ep_i, outs = ep_b, []
    for i in range(int(c.SEMI_REFINE_USING_BOXPC_DELTA_NUM)):
        if i > 0:
            _, ep_i = boxpc_get_model(ctx, cur, pc, oh, False, c, scope_prefix='D_boxpc_branch@%d/' % i)
        outs.append(ep_i['boxpc_out'])
        w = (1 - ep_i['logits_for_weigh']) if c.SEMI_WEIGH_BOXPC_DELTA_DURING_TEST else torch.ones_like(fit_prob)
        dc, da, ds = ep_i['boxpc_delta_center'] * w[:, None], ep_i['boxpc_delta_angle'] * w, ep_i['boxpc_delta_size'] * w[:, None]
        cur = (cur[0] - dc, cur[1] - ds, cur[2] - da)
        tot_c, tot_s, tot_a = tot_c + dc, tot_s + ds, tot_a + da
    if c.SEMI_BOXPC_MIN_FIT_LOSS_AFT_REFINE:                     # train_semisup_adv.py:388-389: the LAST evaluation's fit probability
        fit_prob = torch.softmax(ep_i['boxpc_fit_logits'], dim=-1)[:, 1]
    ep['boxpc_fit_prob'] = fit_prob
    ep['boxpc_outs'] = outs
    ep['total_delta'] = torch.cat([tot_c, tot_s, tot_a[:, None]], dim=1)
    ep['F2_center'] = ep['F_center'] - tot_c
    ep['F2_heading_residuals'] = ep['F_heading_residuals'] - tot_a[:, None]
    ep['F2_size_residuals'] = ep['F_size_residuals'] - tot_s[:, None, :]
    ep['intraclsdims_train_classes'] = train_classes
    ep['inactive_vol_train_classes'] = train_classes      # train_semisup_adv.py sets both lists from the same classes
    if 'Rtilt' in batch:
        ep['weak_inputs'] = {k: torch.as_tensor(batch[k], dtype=dtype) for k in ('Rtilt', 'K', 'rot_frust', 'box2D', 'img_dim')}
    if forced and forced.get('S_box') is not None:
        ep['forced_S_box'] = forced['S_box']
    loss = get_semi_loss_final(pred, _labels_to_torch(batch, dtype), ep, c)
    grads = {}
    if want_grads:
        gl = torch.autograd.grad(loss, [Pl[k] for k in names], allow_unused=True)
        grads = {k: (g if g is not None else torch.zeros_like(Pl[k])) for k, g in zip(names, gl)}
    ep['__flips__'] = dict(ctx.flips)
    ep['__margins__'] = dict(ctx.margins)
    return loss, ep, grads, ctx.ema_updates


def stage_c_inference(P, batch, c, refine_num, dtype=torch.float64, use_one_hot=True, use_oracle_mask=False, mask_pc_for_boxpc=False):
    """The inference graph of test_semisup.py:61-149 for SEMI_MODEL F: every net in inference mode (moving statistics,
    no dropout); the F_ box in regression form is refined `refine_num` times by the Box-PC net,
    box <- box - w * delta(box, pc), w = 1 - p_fit if SEMI_WEIGH_BOXPC_DELTA_DURING_TEST else 1; the F2_ heads are
    the F_ heads minus the accumulated deltas (136-142).  use_oracle_mask (test_semisup.py:75): y_seg replaces the seg logits;
    mask_pc_for_boxpc (103-105): the Box-PC net sees pc * float(argmax(logits, 2))."""
    Pl = {k: val.detach().to(dtype) for k, val in P.items()}
    ctx = Ctx(Pl, is_training=False, bn_decay=0.5, dropout_masks={})
    pc = torch.as_tensor(batch['pc'], dtype=dtype)
    oh = torch.as_tensor(batch['one_hot_vec'], dtype=dtype)
    pred, ep = get_semi_model_final(ctx, pc, oh, use_one_hot, c, norm_box2D=batch_norm_box2D(batch, c, dtype),
                                    oracle_mask=batch['y_seg'] if use_oracle_mask else None)
    cur = ep['F_pred_box_reg']
    tot_c, tot_s, tot_a = torch.zeros_like(cur[0]), torch.zeros_like(cur[1]), torch.zeros_like(cur[2])
    fit = None
    pc_boxpc = pc * torch.argmax(ep['logits'], dim=2).to(dtype)[:, :, None] if mask_pc_for_boxpc else pc
    for _ in range(int(refine_num)):
        _, ep_b = boxpc_get_model(ctx, cur, pc_boxpc, oh, False, c, scope_prefix='D_boxpc_branch/')
        fit = torch.softmax(ep_b['boxpc_fit_logits'], dim=-1)[:, 1]
        w = (1 - ep_b['logits_for_weigh']) if c.SEMI_WEIGH_BOXPC_DELTA_DURING_TEST else torch.ones_like(fit)
        dc, da, ds = ep_b['boxpc_delta_center'] * w[:, None], ep_b['boxpc_delta_angle'] * w, ep_b['boxpc_delta_size'] * w[:, None]
        cur = (cur[0] - dc, cur[1] - ds, cur[2] - da)
        tot_c, tot_s, tot_a = tot_c + dc, tot_s + ds, tot_a + da
    ep['boxpc_fit_prob'] = fit
    ep['F2_center'] = ep['F_center'] - tot_c
    ep['F2_heading_scores'] = ep['F_heading_scores']
    ep['F2_heading_residuals'] = ep['F_heading_residuals'] - tot_a[:, None]
    ep['F2_size_scores'] = ep['F_size_scores']
    ep['F2_size_residuals'] = ep['F_size_residuals'] - tot_s[:, None, :]
    ep['refined_box'] = cur
    return pred, ep


def inference_scores(logits, heading_scores, size_scores, boxpc_fit_prob=None):
    """Detection confidence of test_semisup.py:233-246 (NumPy): log(mean mask prob + .01) + log(max heading prob + .01) +
    log(max size prob + .01) [+ log(p_fit + .01)], mask mean over the predicted-foreground points with the reference's
    `+ 1` in the denominator."""
    import numpy as np

    def softmax(x):
        e = np.exp(x - x.max(axis=-1, keepdims=True))
        return e / e.sum(axis=-1, keepdims=True)
    seg_prob = softmax(logits)[:, :, 1]
    seg_mask = np.argmax(logits, 2)
    mask_mean_prob = (seg_prob * seg_mask).sum(1) / (seg_mask.sum(1) + 1)
    s = np.log(mask_mean_prob + 0.01) + np.log(softmax(heading_scores).max(1) + 0.01) + np.log(softmax(size_scores).max(1) + 0.01)
    if boxpc_fit_prob is not None:
        s = s + np.log(boxpc_fit_prob + 0.01)
    return s


def stage_c_params(rng, num_channels, dtype=torch.float64, use_one_hot=True, norm_box2D=False):
    """Variables of the stage-c graph: model F under class_agnostic/ + class_dependent/, Box-PC under D_boxpc_branch/."""
    P = init_params(rng, layer_table(num_channels, 'F', use_one_hot=use_one_hot, norm_box2D=norm_box2D), dtype)
    P.update(init_params(rng, layer_table(num_channels, 'boxpc', prefix_agnostic='D_boxpc_branch/'), dtype))
    return P
