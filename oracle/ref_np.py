"""ORACLE (test infrastructure only) -- independent float64 NumPy restatement of the forward path.

PARITY UNPINNED against TensorFlow (see oracle/ref_torch.py header and SURVEY.md 8c).  Written
from the same reference lines as ref_torch.py but without sharing any code with it (explicit loops
over layers, explicit per-frustum loss arithmetic), so that the two agree only if both read the
reference the same way.  tests/test_oracle.py requires agreement to 1e-9 in fp64 and checks the
torch-autograd gradients against central finite differences of THIS forward.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import numpy as np

from .ref_constants import NUM_HEADING_BIN as NH, NUM_SIZE_CLUSTER as NS, MEAN_DIMS_ARR, ORIENT_ANCHORS, BN_EPS

MEAN32 = MEAN_DIMS_ARR.astype(np.float32).astype(np.float64)      # tf.constant(..., dtype=tf.float32)
BINS32 = ORIENT_ANCHORS.astype(np.float32).astype(np.float64)


def _bn_train(y, gamma, beta):
    """Training-mode batch norm over all rows (tf.contrib.layers.batch_norm, eps 1e-3, biased var)."""
    flat = y.reshape(-1, y.shape[-1])
    mu = flat.sum(0) / flat.shape[0]
    var = ((flat - mu) ** 2).sum(0) / flat.shape[0]
    return (y - mu) * (gamma / np.sqrt(var + BN_EPS)) + beta, mu, var


def _bn_eval(y, gamma, beta, mm, mv):
    return (y - mm) * (gamma / np.sqrt(mv + BN_EPS)) + beta


def layer(P, x, scope, bn=True, relu=True, training=True, stats=None):
    """tf_util.conv2d 1x1 / fully_connected: x.W + b, optional BN, optional ReLU."""
    W = P[scope + '/weights']
    W = W.reshape(-1, W.shape[-1])
    y = np.matmul(x, W) + P[scope + '/biases']
    if bn:
        g, b = P[scope + '/bn/gamma'], P[scope + '/bn/beta']
        if training:
            y, mu, var = _bn_train(y, g, b)
            if stats is not None:
                stats[scope] = (mu, var, int(np.prod(x.shape[:-1])))
        else:
            y = _bn_eval(y, g, b, P[scope + '/bn/moving_mean'], P[scope + '/bn/moving_variance'])
    if relu:
        y = np.maximum(y, 0.0)
    return y


def inst_seg(P, pc, drop_mask, stats, pre='inst_seg'):
    """semisup_models.py:69-139 (one_hot_vec=None)."""
    B, N, _ = pc.shape
    n = pc
    for name in ('conv1', 'conv2', 'conv3'):
        n = layer(P, n, '%s/%s' % (pre, name), stats=stats)
    pf = n
    n = layer(P, n, pre + '/conv4', stats=stats)
    n = layer(P, n, pre + '/conv5', stats=stats)
    g = n.max(axis=1)                                               # (B,1024)
    cat = np.concatenate([pf, np.repeat(g[:, None, :], N, axis=1)], axis=2)
    n = cat
    for name in ('conv6', 'conv7', 'conv8', 'conv9'):
        n = layer(P, n, '%s/%s' % (pre, name), stats=stats)
    n = n * drop_mask / 0.5
    return layer(P, n, pre + '/conv10', bn=False, relu=False), g


def huber(e, delta):
    a = np.abs(e)
    q = np.minimum(a, delta)
    return 0.5 * q * q + delta * (a - q)


def softmax_ce(logits, label):
    z = logits - logits.max(-1, keepdims=True)
    lse = np.log(np.exp(z).sum(-1))
    return lse - np.take_along_axis(z, label[..., None].astype(np.int64), -1)[..., 0]


def roty_corners(center, heading, size):
    """model_util.py:94-119 for one box: (3,), scalar, (l,w,h) -> (8,3)."""
    l, w, h = size
    x = np.array([l, l, -l, -l, l, l, -l, -l]) / 2
    y = np.array([h, h, h, h, -h, -h, -h, -h]) / 2
    z = np.array([w, -w, -w, w, w, -w, -w, w]) / 2
    c, s = np.cos(heading), np.sin(heading)
    R = np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]])
    return (R @ np.stack([x, y, z])).T + center


def model_a_forward(P, batch, c, training=True):
    """SEMI_MODEL A forward + loss in float64 (semisup_v1_sunrgbd.py:81-130, 256-321, 423-553).

    Returns (loss, out dict).  P: dict of float64 arrays keyed by TF variable names.
    """
    pc = batch['pc'].astype(np.float64)
    B, N, _ = pc.shape
    stats = {}
    drop = batch['dropout_masks']['inst_seg/dp1'].astype(np.float64) if training else np.full((B, N, 128), 0.5)
    logits, gfeat = inst_seg(P, pc, drop, stats)
    mask = (logits[:, :, 0] < logits[:, :, 1]).astype(np.float64)   # (B,N)
    xyz = pc[:, :, :3]
    cnt = np.maximum(mask.sum(1), 1.0)
    mean = (mask[:, :, None] * xyz).sum(1) / cnt[:, None]           # (B,3)

    n = xyz - mean[:, None, :]
    for name in ('conv-reg1-stage1', 'conv-reg2-stage1', 'conv-reg3-stage1'):
        n = layer(P, n, 'tnet/' + name, stats=stats)
    f = (n * mask[:, :, None]).max(axis=1)
    tnet_feats = f
    f = layer(P, f, 'tnet/fc1-stage1', stats=stats)
    f = layer(P, f, 'tnet/fc2-stage1', stats=stats)
    s1 = layer(P, f, 'tnet/fc3-stage1', bn=False, relu=False) + mean

    n = xyz - s1[:, None, :]
    for name in ('conv-reg1', 'conv-reg2', 'conv-reg3', 'conv-reg4'):
        n = layer(P, n, 'box_est/' + name, stats=stats)
    f1 = (n * mask[:, :, None]).max(axis=1)
    f = layer(P, f1, 'box_est/fc1', stats=stats)
    f = layer(P, f, 'box_est/fc2', stats=stats)
    out = layer(P, f, 'box_est/fc3', bn=False, relu=False)         # (B,67)

    center = out[:, 0:3] + s1
    hs = out[:, 3:3 + NH]
    hrn = out[:, 3 + NH:3 + 2 * NH]
    ss = out[:, 3 + 2 * NH:3 + 2 * NH + NS]
    srn = out[:, 3 + 2 * NH + NS:].reshape(B, NS, 3)
    hres = hrn * (np.pi / NH)
    sres = srn * MEAN32[None]

    # losses, frustum by frustum
    tot = np.zeros(B)
    terms = {k: np.zeros(B) for k in ('mask', 'center', 'stage1', 'hcls', 'hres', 'scls', 'sres', 'corner')}
    for b in range(B):
        terms['mask'][b] = softmax_ce(logits[b], batch['y_seg'][b]).mean()
        yc = batch['y_center'][b].astype(np.float64)
        terms['center'][b] = huber(np.sqrt(((yc - center[b]) ** 2).sum()), 2.0)
        terms['stage1'][b] = huber(np.sqrt(((yc - s1[b]) ** 2).sum()), 1.0)
        j = int(batch['y_orient_cls'][b])
        k = int(batch['y_dims_cls'][b])
        yor = float(batch['y_orient_reg'][b])
        ydr = batch['y_dims_reg'][b].astype(np.float64)
        terms['hcls'][b] = softmax_ce(hs[b], np.array(j))
        terms['hres'][b] = huber(hrn[b, j] - yor / (np.pi / NH), 1.0)
        terms['scls'][b] = softmax_ce(ss[b], np.array(k))
        terms['sres'][b] = huber(np.sqrt(((ydr / MEAN32[k] - srn[b, k]) ** 2).sum()), 1.0)
        # corners: predicted box of the GT bins, sizes = anchor + 2*residual (model_util.py:158-159)
        cp = roty_corners(center[b], BINS32[j] + hres[b, j], MEAN32[k] + 2.0 * sres[b, k])
        hl = BINS32[j] + yor
        sl = MEAN32[k] + ydr
        cg = roty_corners(yc, hl, sl)
        cf = roty_corners(yc, hl + np.pi, sl)
        d = np.minimum(np.sqrt(((cp - cg) ** 2).sum(1)), np.sqrt(((cp - cf) ** 2).sum(1)))
        terms['corner'][b] = huber(d, 1.0).mean()
        box = c.STRONG_BOX_MULTIPLER * (c.STRONG_WEIGHT_CENTER * terms['center'][b]
                                        + c.STRONG_WEIGHT_ORIENT_CLS * terms['hcls'][b]
                                        + c.STRONG_WEIGHT_DIMS_CLS * terms['scls'][b]
                                        + c.STRONG_WEIGHT_ORIENT_REG * terms['hres'][b]
                                        + c.STRONG_WEIGHT_DIMS_REG * terms['sres'][b]
                                        + c.STRONG_WEIGHT_TNET_CENTER * terms['stage1'][b]) \
            + c.STRONG_WEIGHT_CORNER * terms['corner'][b]
        tot[b] = (1 - int(batch['is_data_2D'][b])) * (c.STRONG_WEIGHT_CROSS_ENTROPY * terms['mask'][b] + box)
    loss = tot.mean()

    # anchor -> reg (tf_util.py:1001-1041)
    ks = ss.argmax(1)
    js = hs.argmax(1)
    ar = np.arange(B)
    dims = np.maximum(MEAN32[ks] + sres[ar, ks], 1e-5)
    theta = BINS32[js] + hres[ar, js]
    outd = dict(logits=logits, mask=mask, mask_xyz_mean=mean, stage1_center=s1, center=center,
                heading_scores=hs, heading_residuals_normalized=hrn, size_scores=ss,
                size_residuals_normalized=srn, box_params=out, feats_lv1=f1, tnet_feats=tnet_feats,
                seg_global_feat=gfeat, loss_terms=terms, total_losses=tot, S_dims=dims, S_theta=theta,
                bn_stats=stats)
    return loss, outd
