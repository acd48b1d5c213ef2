"""Sub-network builders under the reference's names and signatures
(sunrgbd/sunrgbd_detection/semisup_models.py: mlps_with_dropout 44, v1_inst_seg 69, subtract_points_mean 145,
v1_tnet 164, subtract_1st_stage_center 204, v1_box_est 215, box_pc_mask_features_model 297).  Each call allocates the sub-network (variables in the reference's
scopes, HBM buffers) on the default graph and links it into the graph's ModelAssembly; the fused launch
schedule is emitted when a Session first runs."""
import numpy as np

from . import api
from .constants import BOX_OUT_DIMS, MEAN_DIMS_ARR, NUM_HEADING_BIN, NUM_SIZE_CLUSTER
from .nets import BoxEstNet, InstSegNet, TNet


def _asm(ctx, use_one_hot):
    asm = ctx.assembly
    assert asm is not None, 'call semisup_v1_sunrgbd.get_semi_model (or api.Graph.ensure_assembly) first'
    return asm


def _box2d_feats(asm, one_hot_vec, norm_box2D):
    """norm_box2D joins the FC input of the T-Net / box net behind the one-hot vector (semisup_models.py:192-195, 249-252): the
    assembly gets the [one_hot | norm_box2D] block (nets.ExtraFeats) those layers read."""
    if norm_box2D is None:
        return
    from .nets import ExtraFeats
    from .constants import NUM_CLASS
    n_oh = NUM_CLASS if one_hot_vec is not None else 0
    if asm.extra is None or asm.extra.n_oh != n_oh:
        asm.extra = ExtraFeats(asm.g, n_oh)
    asm.box2d = True


def v1_inst_seg(point_cloud, img_feats, one_hot_vec, end_points, is_training, bn_decay=None, scope=None):
    """Instance-segmentation PointNet -> per-point logits (B,N,2)."""
    ctx = point_cloud.ctx
    asm = _asm(ctx, one_hot_vec is not None)
    asm.seg = InstSegNet(ctx.engine, scope, one_hot_vec is not None)
    ctx.is_training = is_training if isinstance(is_training, api.BoolPlaceholder) else bool(is_training)
    e = ctx.engine
    logits = api.Tensor(ctx, asm.seg.logits, (e.B, e.rpf, 2), scope + '/logits', producer=asm.seg)
    end_points['seg_global_feat'] = api.Tensor(ctx, asm.seg.L5.pooled, (e.B, 1024), scope + '/global_feat')
    return logits


def subtract_points_mean(point_cloud, logits, scope=None):
    """mask = logit0 < logit1 (hard, no gradient); masked xyz mean; recentred xyz.  Computed by the segmentation-head
    kernel; the recentred cloud stays lazy."""
    ctx = point_cloud.ctx
    seg = logits.producer
    e = ctx.engine
    mask = api.Tensor(ctx, seg.mask, (e.B, e.rpf, 1), 'mask', producer=seg)
    mean = api.Tensor(ctx, seg.mask_xyz_mean, (e.B, 1, 3), 'mask_xyz_mean', producer=seg)
    xyz = api.LazyPoints(ctx, point_cloud, 3)
    return mask, mean, xyz, api.LazyPoints(ctx, point_cloud, 3, sub=mean)


def v1_tnet(point_cloud_xyz_stage1, mask, mask_xyz_mean, one_hot_vec, end_points, is_training, norm_box2D=None,
            bn_decay=None, scope=None):
    ctx = point_cloud_xyz_stage1.ctx
    asm = _asm(ctx, one_hot_vec is not None)
    _box2d_feats(asm, one_hot_vec, norm_box2D)
    asm.tnet = TNet(ctx.engine, scope, one_hot_vec is not None, box2d=norm_box2D is not None)
    s1 = api.Tensor(ctx, asm.tnet.F3.out, (ctx.engine.B, 3), scope + '/stage1_center', producer=asm.tnet)
    end_points['stage1_center'] = s1
    return s1


def subtract_1st_stage_center(point_cloud_xyz, stage1_center, scope=None):
    return api.LazyPoints(point_cloud_xyz.ctx, point_cloud_xyz.pc, 3, sub=stage1_center)


def v1_box_est(point_cloud_xyz_submean, stage1_center, mask, one_hot_vec, end_points, is_training, norm_box2D=None,
               bn_decay=None, prefix='', c=None, scope=None):
    ctx = point_cloud_xyz_submean.ctx
    asm = _asm(ctx, one_hot_vec is not None)
    _box2d_feats(asm, one_hot_vec, norm_box2D)
    asm.box = BoxEstNet(ctx.engine, scope, one_hot_vec is not None, box2d=norm_box2D is not None)
    B = ctx.engine.B
    NH, NS = NUM_HEADING_BIN, NUM_SIZE_CLUSTER
    box = asm.box
    out = api.Tensor(ctx, box.G3.out, (B, BOX_OUT_DIMS), scope + '/box_params', producer=box)
    end_points[prefix + 'feats_lv1'] = api.Tensor(ctx, box.B4.pooled, (B, 512), 'feats_lv1')
    end_points[prefix + 'feats_lv2'] = api.Tensor(ctx, box.G1.out, (B, 512), 'feats_lv2')
    end_points[prefix + 'feats_lv3'] = api.Tensor(ctx, box.G2.out, (B, 256), 'feats_lv3')
    end_points[prefix + 'box_params'] = out
    heads = BoxHeads(out, stage1_center, prefix)
    end_points.update(heads.end_points())
    return heads.pred_box()


class SlicedTensor(api.Tensor):
    """Column slice / affine view of a (B,67) head output, evaluated on the host at fetch time
    (semisup_models.py:265-290 slicing and scaling)."""

    def __init__(self, src, fn, shape, name):
        api.Tensor.__init__(self, src.ctx, None, shape, name)
        self.src, self.fn = src, fn

    def numpy(self):
        return self.fn()


class BoxHeads:
    def __init__(self, out, stage1_center, prefix):
        self.out, self.s1, self.prefix = out, stage1_center, prefix

    def end_points(self):
        o, s1, p = self.out, self.s1, self.prefix
        B = o.shape[0]
        NH, NS = NUM_HEADING_BIN, NUM_SIZE_CLUSTER
        mean = MEAN_DIMS_ARR.astype(np.float32)
        mk = lambda fn, shape, name: SlicedTensor(o, fn, shape, p + name)
        return {
            p + 'center': mk(lambda: o.numpy()[:, 0:3] + s1.numpy(), (B, 3), 'center'),
            p + 'heading_scores': mk(lambda: o.numpy()[:, 3:3 + NH], (B, NH), 'heading_scores'),
            p + 'heading_residuals_normalized': mk(lambda: o.numpy()[:, 3 + NH:3 + 2 * NH], (B, NH), 'heading_residuals_normalized'),
            p + 'heading_residuals': mk(lambda: o.numpy()[:, 3 + NH:3 + 2 * NH] * np.float32(np.pi / NH), (B, NH), 'heading_residuals'),
            p + 'size_scores': mk(lambda: o.numpy()[:, 3 + 2 * NH:3 + 2 * NH + NS], (B, NS), 'size_scores'),
            p + 'size_residuals_normalized': mk(lambda: o.numpy()[:, 3 + 2 * NH + NS:].reshape(B, NS, 3), (B, NS, 3),
                                                'size_residuals_normalized'),
            p + 'size_residuals': mk(lambda: o.numpy()[:, 3 + 2 * NH + NS:].reshape(B, NS, 3) * mean[None], (B, NS, 3),
                                     'size_residuals'),
        }

    def pred_box(self):
        ep, p = self.end_points(), self.prefix
        return (ep[p + 'center'], ep[p + 'size_scores'], ep[p + 'size_residuals'], ep[p + 'heading_scores'],
                ep[p + 'heading_residuals'])


def mlps_with_dropout(input_feat, layers, activation_fns, keep_probs, is_training, bn=True, bn_decay=None, c=None, scope=None,
                      reuse=None):
    """FC stack with dropout after every hidden layer (semisup_models.py:44-63).  Its one call site on the hot path is
    the class-dependent `box_refine` head of SEMI_MODEL F (semisup_v1_sunrgbd.py:183-197: feats_lv1 (+ one_hot) -> 512 ->
    256 -> 67); the stack is part of the fused stage-c assembly (nets.SemiModelF.R0-R2) and this returns its output."""
    assert(len(layers) == len(activation_fns) == len(keep_probs))
    from .nets import SemiModelF
    ctx = input_feat.ctx
    m = ctx.assembly
    if not isinstance(m, SemiModelF) or tuple(layers) != (512, 256, BOX_OUT_DIMS) or scope != 'box_refine':
        raise NotImplementedError('mlps_with_dropout: only the box_refine head of SEMI_MODEL F is on the hot path '
                                  '(layers (512, 256, %d)); got %r in scope %r' % (BOX_OUT_DIMS, tuple(layers), scope))
    if not bn:
        raise NotImplementedError('box_refine is built with batch-norm on its hidden layers (semisup_v1_sunrgbd.py:196)')
    return api.Tensor(ctx, m.R2.out, (ctx.engine.B, layers[-1]), 'class_dependent/' + scope + '/fc2')


def box_pc_mask_features_model(box, pc, logits, num_outputs, is_training, end_points, reuse, bn_for_output, normalize_pc=False,
                               normalize_method='SD', one_hot_vec=None, norm_box2D=None, bn_decay=None, c=None, scope=None):
    """The Box-PC Fit net on a (box, point cloud) pair -> ((B, num_outputs) output, {feature level: tensor})
    (semisup_models.py:297-324 -> combined_box_pc_mask_features_model 326-398).  Representation 'A' (the only one in a
    published recipe): the 6 signed face distances per point appended to the raw channels, conv D->128->128->256->512,
    max-pool, FC 512 -> 256 -> num_outputs with dropout 0.7.  Variables live under the literal scope box_pc_mask_model/."""
    from .boxpc_sunrgbd import BoxRegHandle
    from .nets import BoxPCModel
    if c.BOX_PC_MASK_REPRESENTATION == 'B':
        raise NotImplementedError('BOX_PC_MASK_REPRESENTATION B (independent_box_pc_mask_features_model) is in no published recipe')
    if c.BOX_PC_MASK_REPRESENTATION != 'A':
        raise Exception('Box pc mask representation not implemented: %s' % c.BOX_PC_MASK_REPRESENTATION)
    if logits is not None or normalize_pc or norm_box2D is not None or bn_for_output:
        raise NotImplementedError('mask / normalize_pc / norm_box2D / bn_for_output are off at every call site of the reference')
    if num_outputs != 9:
        raise NotImplementedError('the fit net has 2 fit logits + 7 delta terms (boxpc_sunrgbd.py:60,65)')
    ctx = pc.ctx
    e = ctx.engine
    if not isinstance(box, BoxRegHandle):
        raise NotImplementedError('a GT box enters through convert_raw_y_box_to_reg_format (train_boxpc.py:233); a predicted box '
                                  'enters inside SEMI_MODEL F (get_semi_model_final), which owns its own frozen Box-PC branch')
    ctx.assembly = BoxPCModel(e, c, one_hot_vec is not None, inputs=ctx.inputs)
    ctx.is_training = is_training if isinstance(is_training, api.BoolPlaceholder) else bool(is_training)
    if isinstance(bn_decay, (int, float)):
        e.hyper[2] = float(bn_decay)
    net = ctx.assembly.net
    B = e.B
    T = lambda buf, shape, name: api.Tensor(ctx, buf, shape, name)
    feats = {'box_pc_mask_model_feats_lv1': T(net.P4.pooled, (B, 512), 'feats_lv1'),
             'box_pc_mask_model_feats_lv2': T(net.F1.out, (B, 512), 'feats_lv2'),
             'box_pc_mask_model_feats_lv3': T(net.F2.out, (B, 256), 'feats_lv3')}
    return T(net.F3.out, (B, 9), 'box_pc_mask_model/output'), feats
