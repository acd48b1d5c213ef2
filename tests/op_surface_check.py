"""Shared by the CPU (specification library) and GPU tests: the reference's operator wrappers called one by one
(models/tf_util.py: conv2d 1258-1323 with and without bn, batch_norm_for_conv2d 1693-1705, batch_norm_for_fc 1666-1677, dropout
1720-1741, max_pool2d 1501-1524, fully_connected 1463-1499) on a small graph, against the oracle's restatements of the same ops."""
import numpy as np
import torch

from oracle import ref_torch as R
from transferable3d_amd import api, tf_util
from transferable3d_amd.synthetic import make_batch


def check_operator_surface(rt, is_training=True, seed=5):
    B, N, C = 4, 128, 4
    batch = make_batch(B, N, C, seed=seed)
    r = np.random.RandomState(seed)
    masks = {'dp_points': (r.uniform(size=(B, N, 64)) < 0.6).astype(np.float32), 'dp_fc': (r.uniform(size=(B, 128)) < 0.7).astype(np.float32)}
    with api.Graph(rt=rt, seed=seed).as_default() as g:
        g.ensure_engine(B, N, C)
        pc = api.placeholder('pc', (B, N, C))
        it = is_training
        x = tf_util.conv2d(pc, 64, [1, C], scope='c1', bn=True, is_training=it)                        # [1,D] kernel + BN + ReLU
        x = tf_util.conv2d(x, 64, [1, 1], scope='c2', bn=False, activation_fn=None, is_training=it)    # bare convolution
        x = tf_util.batch_norm_for_conv2d(x, it, None, scope='c2/bn')                                  # ... its batch-norm as a node
        xr = tf_util.conv2d(x, 64, [1, 1], scope='c3', bn=False, activation_fn='relu', is_training=it)  # conv + ReLU, no BN
        xd = tf_util.dropout(xr, it, scope='dp_points', keep_prob=0.6)
        logits = tf_util.conv2d(xd, 2, [1, 1], scope='c4', bn=False, activation_fn=None, is_training=it)   # 2 output channels
        xp = tf_util.conv2d(xr, 128, [1, 1], scope='c5', bn=True, is_training=it, pool_over_points=True)
        pooled = tf_util.max_pool2d(xp, [N, 1], scope='pool')
        pooled2 = api.Tensor(g, pooled.buf, (B, 128), 'pooled')
        f = tf_util.fully_connected(pooled2, 128, scope='f1', bn=False, activation_fn=None, is_training=it)
        fb = tf_util.batch_norm_for_fc(f, it, None, scope='f1/bn')
        fd = tf_util.dropout(fb, it, scope='dp_fc', keep_prob=0.7)
        out = tf_util.fully_connected(fd, 3, scope='f2', bn=False, activation_fn=None, is_training=it)
        # a few non-trivial parameter values (biases and moving statistics are 0 / 1 at init)
        rr = np.random.RandomState(seed + 1)
        sd = g.vars.state_dict()
        for k in sd:
            if k.endswith(('biases', 'beta', 'moving_mean')):
                sd[k] = rr.normal(0, 0.2, size=sd[k].shape).astype(np.float32)
            elif k.endswith(('gamma', 'moving_variance')):
                sd[k] = (0.5 + rr.uniform(size=sd[k].shape)).astype(np.float32)
        g.vars.load_state_dict(sd)
        sess = api.Session(use_hip_graph=False)
        feed = {pc: batch['pc']}
        if it:
            feed.update({k: v for k, v in masks.items()})
        got = sess.run([logits, pooled2, out], feed_dict=feed)
        state = g.vars.state_dict()
    # ---- oracle: the same ops, fp64 ----
    P = {k: torch.as_tensor(v.astype(np.float64)) for k, v in sd.items()}
    # standalone batch-norm variables live directly under their scope; the oracle's batch_norm takes that scope
    ctx = R.Ctx(P, is_training=it, bn_decay=0.5, dropout_masks={k: torch.as_tensor(v) for k, v in masks.items()})
    x0 = torch.as_tensor(batch['pc'], dtype=torch.float64)
    h = R.conv2d(ctx, x0, 'c1')
    h = R.conv2d(ctx, h, 'c2', bn=False, activation=None)
    h = R.batch_norm(ctx, h, 'c2/bn')
    hr = R.conv2d(ctx, h, 'c3', bn=False, activation='relu')
    hd = R.dropout(ctx, hr, 'dp_points', 0.6)
    lg = R.conv2d(ctx, hd, 'c4', bn=False, activation=None)
    hp = R.max_pool_points(R.conv2d(ctx, hr, 'c5'))
    ff = R.fully_connected(ctx, hp, 'f1', bn=False, activation=None)
    ff = R.batch_norm(ctx, ff, 'f1/bn')
    ff = R.dropout(ctx, ff, 'dp_fc', 0.7)
    oo = R.fully_connected(ctx, ff, 'f2', activation=None)
    for name, mine, ref in (('logits', got[0], lg), ('pooled', got[1], hp), ('out', got[2], oo)):
        ref = ref.detach().numpy().reshape(mine.shape)
        assert np.abs(mine - ref).max() < 1e-4 * max(1.0, np.abs(ref).max()), (name, float(np.abs(mine - ref).max()))
    if it:      # the standalone batch-norm nodes update THEIR moving statistics (updates_collections=None)
        for k, v in ctx.ema_updates.items():
            assert np.abs(state[k].reshape(v.shape) - v.numpy()).max() < 1e-4 * max(1.0, float(v.abs().max())), k
        assert 'c2/bn/moving_mean' in ctx.ema_updates and 'f1/bn/moving_variance' in ctx.ema_updates
    return True
