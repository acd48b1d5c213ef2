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


def inst_seg_by_the_reference_calls(point_cloud, one_hot_vec, is_training, bn_decay, scope):
    """The call sequence of the reference's v1_inst_seg (sunrgbd/sunrgbd_detection/semisup_models.py:69-139) with `tf.` spelled
    `api.`: an ordinary conv2d followed by max_pool2d(net, [num_point,1], padding='VALID'), the one-hot concat, tile + concat of the
    global feature, dropout, the conv10 logits -- every argument as the reference passes it."""
    with api.variable_scope(scope) as sc:       # noqa: F841
        batch_size, num_point, D = point_cloud.get_shape().as_list()
        pc_image = api.expand_dims(point_cloud, -1)
        kw = dict(padding='VALID', stride=[1, 1], bn=True, is_training=is_training, bn_decay=bn_decay)
        net = tf_util.conv2d(pc_image, 64, [1, D], scope='conv1', **kw)
        net = tf_util.conv2d(net, 64, [1, 1], scope='conv2', **kw)
        point_feat = tf_util.conv2d(net, 64, [1, 1], scope='conv3', **kw)
        net = tf_util.conv2d(point_feat, 128, [1, 1], scope='conv4', **kw)
        net = tf_util.conv2d(net, 1024, [1, 1], scope='conv5', **kw)
        global_feat = tf_util.max_pool2d(net, [num_point, 1], padding='VALID', scope='maxpool')
        if one_hot_vec is not None:
            global_feat = api.concat([global_feat, api.expand_dims(api.expand_dims(one_hot_vec, 1), 1)], axis=3)
        global_feat_expand = api.tile(global_feat, [1, num_point, 1, 1])
        concat_feat = api.concat(axis=3, values=[point_feat, global_feat_expand])
        net = tf_util.conv2d(concat_feat, 512, [1, 1], scope='conv6', **kw)
        net = tf_util.conv2d(net, 256, [1, 1], scope='conv7', **kw)
        net = tf_util.conv2d(net, 128, [1, 1], scope='conv8', **kw)
        net = tf_util.conv2d(net, 128, [1, 1], scope='conv9', **kw)
        net = tf_util.dropout(net, is_training, 'dp1', keep_prob=0.5)
        logits = tf_util.conv2d(net, 2, [1, 1], padding='VALID', stride=[1, 1], activation_fn=None, scope='conv10')
        logits = api.squeeze(logits, [2])
    return logits, global_feat


def check_reference_inst_seg_call_sequence(rt, use_one_hot, is_training=True, seed=9):
    """The operator surface takes the reference's own calls: the graph built by them equals oracle.v1_inst_seg (same variables under
    the same names) within 1e-4; minimize on such a graph says what to use instead."""
    B, N, C = 4, 128, 4
    batch = make_batch(B, N, C, seed=seed, dropout_scopes={'inst_seg/dp1': ((B, N, 128), 0.5)})
    with api.Graph(rt=rt, seed=seed).as_default() as g:
        g.ensure_engine(B, N, C)
        pc = api.placeholder('pc', (B, N, C))
        oh = api.placeholder('one_hot_vec', (B, 10)) if use_one_hot else None
        logits, global_feat = inst_seg_by_the_reference_calls(pc, oh, is_training, None, 'inst_seg')
        assert tuple(logits.shape) == (B, N, 2) and global_feat.shape[-1] == 1024 + (10 if use_one_hot else 0)
        rr = np.random.RandomState(seed + 1)
        sd = g.vars.state_dict()
        assert tuple(sd['inst_seg/conv6/weights'].shape) == (1, 1, 64 + 1024 + (10 if use_one_hot else 0), 512)
        assert tuple(sd['inst_seg/conv1/weights'].shape) == (1, C, 1, 64) and 'inst_seg/conv10/biases' in sd
        for k in sd:
            if k.endswith(('biases', 'beta', 'moving_mean')):
                sd[k] = rr.normal(0, 0.2, size=sd[k].shape).astype(np.float32)
            elif k.endswith(('gamma', 'moving_variance')):
                sd[k] = (0.5 + rr.uniform(size=sd[k].shape)).astype(np.float32)
        g.vars.load_state_dict(sd)
        sess = api.Session(use_hip_graph=False)
        feed = {pc: batch['pc']}
        if use_one_hot:
            feed[oh] = batch['one_hot_vec']
        if is_training:
            feed['inst_seg/dp1'] = batch['dropout_masks']['inst_seg/dp1']
        got_logits, got_global = sess.run([logits, global_feat], feed_dict=feed)
        try:
            api.AdamOptimizer(1e-3).minimize(logits)
            raise AssertionError('minimize on an operator-level graph must say that the surface is forward-only')
        except NotImplementedError as err:
            assert 'forward-only' in str(err)
    P = {k: torch.as_tensor(v.astype(np.float64)) for k, v in sd.items()}
    ctx = R.Ctx(P, is_training=is_training, bn_decay=0.5,
                dropout_masks={k: torch.as_tensor(v) for k, v in batch['dropout_masks'].items()})
    ep = {}
    ref = R.v1_inst_seg(ctx, torch.as_tensor(batch['pc'], dtype=torch.float64),
                        torch.as_tensor(batch['one_hot_vec'], dtype=torch.float64) if use_one_hot else None, ep=ep)
    ref = ref.detach().numpy()
    assert np.abs(got_logits - ref).max() < 1e-4 * max(1.0, np.abs(ref).max()), float(np.abs(got_logits - ref).max())
    gref = ep['seg_global_feat'].detach().numpy()
    assert np.abs(got_global.reshape(B, -1)[:, :1024] - gref).max() < 1e-4 * max(1.0, np.abs(gref).max())
    return True
