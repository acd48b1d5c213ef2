"""Model builder under the reference's names and signatures
(sunrgbd/sunrgbd_detection/semisup_v1_sunrgbd.py: placeholder_inputs 37-67, get_semi_model 69-79,
get_semi_model_backbone 81-130, get_semi_loss 248-254, get_semi_loss_backbone 256-321,
convert_raw_y_box_to_reg_format 584-608)."""
import numpy as np

from . import api, semisup_models
from .constants import MEAN_DIMS_ARR, NUM_CLASS, NUM_HEADING_BIN, ORIENT_ANCHORS

INPUT_IMG_CHANNELS = 3


def placeholder_inputs(batch_size, num_point, num_channel):
    """The reference's 18 placeholders, same order.  bg_pc, img and the KITTI matrices (R0_rect, P) feed nothing on the SUN-RGBD
    path and are accepted and ignored; Rtilt, K, rot_frust, box2D, img_dim feed the weak reprojection loss."""
    ctx = api.get_default_graph()
    ctx.ensure_engine(batch_size, num_point, num_channel)
    B, N, C = batch_size, num_point, num_channel
    P = api.placeholder
    return (P('pc', (B, N, C)), P(None, (B, N, C), name='bg_pc'), P(None, (B, None, None, INPUT_IMG_CHANNELS), name='img'),
            P('one_hot_vec', (B, NUM_CLASS)), P('y_seg', (B, N)), P('y_center', (B, 3)), P('y_orient_cls', (B,)),
            P('y_orient_reg', (B,)), P('y_dims_cls', (B,)), P('y_dims_reg', (B, 3)), P(None, (B, 3, 3), name='R0_rect'),
            P(None, (B, 3, 4), name='P'), P('Rtilt', (B, 3, 3)), P('K', (B, 3, 3)),
            P('rot_frust', (B, 1)), P('box2D', (B, 4)), P('img_dim', (B, 2)),
            P('is_data_2D', (B,)))


def get_semi_model(pc, bg_pc, img, one_hot_vec, is_training, use_one_hot, oracle_mask=None, norm_box2D=None,
                   bn_decay=None, c=None):
    if c.SEMI_MODEL == 'A':
        return get_semi_model_backbone(pc, bg_pc, img, one_hot_vec, is_training, use_one_hot, oracle_mask=oracle_mask,
                                       norm_box2D=norm_box2D, bn_decay=bn_decay, c=c)
    elif c.SEMI_MODEL == 'F':
        return get_semi_model_final(pc, bg_pc, img, one_hot_vec, is_training, use_one_hot, oracle_mask=oracle_mask,
                                    norm_box2D=norm_box2D, bn_decay=bn_decay, c=c)
    else:
        raise Exception('Not implemented SEMI_MODEL: %s' % c.SEMI_MODEL)


def get_semi_model_final(pc, bg_pc, img, one_hot_vec, is_training, use_one_hot, oracle_mask=None, norm_box2D=None,
                         bn_decay=None, c=None):
    """SEMI_MODEL F (semisup_v1_sunrgbd.py:132-230): class-agnostic seg / T-Net / box nets + class-dependent box_refine.
    The stage-c training graph also contains the frozen Box-PC branch (train_semisup_adv.py:331-411); it is built here as
    part of the same assembly (nets.SemiModelF) and exposed through end_points exactly as the driver adds it."""
    from .nets import SemiModelF
    ctx = pc.ctx
    e = ctx.engine
    if oracle_mask is not None and getattr(oracle_mask, 'field', None) != 'y_seg':
        # (the one call site of the reference, test_semisup.py:75, passes y_seg_pl)
        raise NotImplementedError('oracle_mask must be the y_seg placeholder (semisup_v1_sunrgbd.py:161-162, test_semisup.py:75)')
    if isinstance(bn_decay, (int, float)):
        e.hyper[2] = float(bn_decay)
    train_classes = getattr(c, 'intraclsdims_train_classes', None)
    if c.USE_NORMALIZED_BOX2D_AS_FEATS and norm_box2D is None:        # semisup_v1_sunrgbd.py:145,168,176
        raise ValueError('USE_NORMALIZED_BOX2D_AS_FEATS needs norm_box2D = tf_util.tf_normalize_2D_bboxes(box2D_pl, img_dim_pl)')
    m = SemiModelF(e, c, use_one_hot=use_one_hot, train_classes=train_classes, inputs=ctx.inputs, oracle_mask=oracle_mask is not None,
                   mask_pc_for_boxpc=bool(getattr(c, 'mask_pc_for_boxpc', False)))
    ctx.assembly = m
    ctx.is_training = is_training if isinstance(is_training, api.BoolPlaceholder) else bool(is_training)
    B, N = e.B, e.rpf
    T = lambda buf, shape, name: api.Tensor(ctx, buf, shape, name)
    logits = T(m.seg.logits, (B, N, 2), 'logits')
    s1 = T(m.tnet.F3.out, (B, 3), 'stage1_center')
    end_points = {'point_cloud': pc, 'class_one_hot': one_hot_vec, 'stage1_center': s1,
                  'dims_anchors': MEAN_DIMS_ARR.astype(np.float32), 'orient_anchors': ORIENT_ANCHORS.astype(np.float32),
                  'feats_lv1': T(m.box.B4.pooled, (B, 512), 'feats_lv1')}
    _common_end_points(ctx, end_points, logits)
    W = semisup_models.BoxHeads(T(m.box.G3.out, (B, 67), 'box_params'), s1, '')
    F = semisup_models.BoxHeads(T(m.R2.out, (B, 67), 'F_box_params'), s1, 'F_')
    end_points.update(W.end_points())
    end_points.update(F.end_points())
    lo = m.loss_op
    end_points['F_pred_box_reg'] = (T(lo.center, (B, 3), 'F_center_reg'), T(lo.reg_dims, (B, 3), 'F_dims_reg'),
                                    T(lo.reg_theta, (B,), 'F_orient_reg'))
    # Box-PC branch outputs and the refined F2_ heads (train_semisup_adv.py:376-411), evaluated on the host at fetch time
    out9 = T(m.boxpc.F3.out, (B, 9), 'boxpc_out')
    end_points['boxpc_fit_prob'] = T(m.fit_prob, (B,), 'boxpc_fit_prob')
    ST = semisup_models.SlicedTensor

    def delta_heads(buf):
        """boxpc_delta_{center,size,angle} of one Box-PC evaluation (boxpc_sunrgbd.py:79-95): columns of its output, times
        (1 - p_fit) under BOXPC_WEIGH_DELTA_PRED_BY_CLS_CONF (evaluated on the host at fetch time)."""
        if not c.BOXPC_WEIGH_DELTA_PRED_BY_CLS_CONF:
            return T(buf[:, 0:3], (B, 3), 'boxpc_delta_center'), T(buf[:, 3:6], (B, 3), 'boxpc_delta_size'), T(buf[:, 6], (B,), 'boxpc_delta_angle')
        full = T(buf, (B, 9), 'boxpc_out')

        def wd():
            l = full.numpy()[:, 7:9].astype(np.float64)
            e = np.exp(l - l.max(1, keepdims=True))
            return (1.0 - e[:, 1] / e.sum(1)).astype(np.float32)
        return (ST(full, lambda: full.numpy()[:, 0:3] * wd()[:, None], (B, 3), 'boxpc_delta_center'),
                ST(full, lambda: full.numpy()[:, 3:6] * wd()[:, None], (B, 3), 'boxpc_delta_size'),
                ST(full, lambda: full.numpy()[:, 6] * wd(), (B,), 'boxpc_delta_angle'))
    end_points['boxpc_delta_center'], end_points['boxpc_delta_size'], end_points['boxpc_delta_angle'] = delta_heads(m.boxpc.F3.out)
    if not ctx.is_training:
        # inference graph (test_semisup.py:95-149): SEMI_REFINE_USING_BOXPC_DELTA_NUM refinement steps on the device,
        # F2_ = F_ - accumulated deltas
        m.refine_num = int(c.SEMI_REFINE_USING_BOXPC_DELTA_NUM)
        tot = T(m.total_delta, (B, 7), 'total_delta')
        end_points['F2_center'] = ST(tot, lambda: end_points['F_center'].numpy() - tot.numpy()[:, 0:3], (B, 3), 'F2_center')
        end_points['F2_heading_scores'] = end_points['F_heading_scores']
        end_points['F2_heading_residuals'] = ST(tot, lambda: end_points['F_heading_residuals'].numpy() - tot.numpy()[:, 6:7],
                                                (B, 12), 'F2_heading_residuals')
        end_points['F2_size_scores'] = end_points['F_size_scores']
        end_points['F2_size_residuals'] = ST(tot, lambda: end_points['F_size_residuals'].numpy() - tot.numpy()[:, None, 3:6],
                                             (B, 10, 3), 'F2_size_residuals')
        return (logits, W.pred_box(), F.pred_box()), end_points

    # training form of the stage-c glue (train_semisup_adv.py:362-399).  With SEMI_REFINE_USING_BOXPC_DELTA_NUM = 1 (the published
    # recipe) one Box-PC evaluation on F_pred_box_reg serves the fit probability and the refinement step, and
    # SEMI_BOXPC_MIN_FIT_LOSS_AFT_REFINE (set in recipe c) changes nothing: the loop's evaluation sees the same unrefined box.
    if m.refine_train > 1:
        # one evaluation per step on the device (nets.SemiModelF.boxpc_nets); the fit probability is the first evaluation's, or the
        # last one's with SEMI_BOXPC_MIN_FIT_LOSS_AFT_REFINE (train_semisup_adv.py:388-389); delta heads = the last evaluation's
        end_points['boxpc_delta_center'], end_points['boxpc_delta_size'], end_points['boxpc_delta_angle'] = \
            delta_heads(m.boxpc_nets[-1].F3.out)
        tot = T(m.total_delta, (B, 7), 'total_delta')
        end_points['F2_center'] = ST(tot, lambda: end_points['F_center'].numpy() - tot.numpy()[:, 0:3], (B, 3), 'F2_center')
        end_points['F2_heading_scores'] = end_points['F_heading_scores']
        end_points['F2_heading_residuals'] = ST(tot, lambda: end_points['F_heading_residuals'].numpy() - tot.numpy()[:, 6:7],
                                                (B, 12), 'F2_heading_residuals')
        end_points['F2_size_scores'] = end_points['F_size_scores']
        end_points['F2_size_residuals'] = ST(tot, lambda: end_points['F_size_residuals'].numpy() - tot.numpy()[:, None, 3:6],
                                             (B, 10, 3), 'F2_size_residuals')
        return (logits, W.pred_box(), F.pred_box()), end_points

    def wgt():
        n = int(bool(c.SEMI_WEIGH_BOXPC_DELTA_DURING_TEST)) + int(bool(c.BOXPC_WEIGH_DELTA_PRED_BY_CLS_CONF))    # boxpc_sunrgbd.py:84-92
        return (1.0 - end_points['boxpc_fit_prob'].numpy()) ** n
    end_points['F2_center'] = ST(out9, lambda: end_points['F_center'].numpy() - out9.numpy()[:, 0:3] * wgt()[:, None], (B, 3), 'F2_center')
    end_points['F2_heading_scores'] = end_points['F_heading_scores']
    end_points['F2_heading_residuals'] = ST(out9, lambda: end_points['F_heading_residuals'].numpy() - (out9.numpy()[:, 6] * wgt())[:, None],
                                            (B, 12), 'F2_heading_residuals')
    end_points['F2_size_scores'] = end_points['F_size_scores']
    end_points['F2_size_residuals'] = ST(out9, lambda: end_points['F_size_residuals'].numpy() - (out9.numpy()[:, 3:6] * wgt()[:, None])[:, None, :],
                                         (B, 10, 3), 'F2_size_residuals')
    return (logits, W.pred_box(), F.pred_box()), end_points


def _common_end_points(ctx, end_points, logits):
    """class_ids = argmax(one_hot) (semisup_v1_sunrgbd.py:91,143) and soft_mask = softmax(logits)[:,:,1] (102,165),
    evaluated on the host at fetch time."""
    B, N = ctx.engine.B, ctx.engine.rpf
    ST = semisup_models.SlicedTensor
    end_points['class_ids'] = ST(logits, lambda: np.argmax(ctx.inputs.one_hot_vec.cpu().numpy(), axis=1).astype(np.int32), (B,),
                                 'class_ids')

    def soft():
        l = logits.numpy().astype(np.float64)
        e = np.exp(l - l.max(axis=2, keepdims=True))
        return (e[:, :, 1] / e.sum(axis=2)).astype(np.float32)
    end_points['soft_mask'] = ST(logits, soft, (B, N), 'soft_mask')


def get_semi_model_backbone(pc, bg_pc, img, one_hot_vec, is_training, use_one_hot, oracle_mask=None, norm_box2D=None,
                            bn_decay=None, c=None):
    """seg PointNet -> masked centroid -> T-Net -> box PointNet (semisup_v1_sunrgbd.py:81-130)."""
    ctx = pc.ctx
    e = ctx.engine
    ctx.ensure_assembly(c, use_one_hot)
    if isinstance(bn_decay, (int, float)):
        e.hyper[2] = float(bn_decay)
    end_points = {'point_cloud': pc, 'class_one_hot': one_hot_vec,
                  'dims_anchors': MEAN_DIMS_ARR.astype(np.float32), 'orient_anchors': ORIENT_ANCHORS.astype(np.float32)}
    if not use_one_hot:
        one_hot_vec = None
    if oracle_mask is not None:
        raise NotImplementedError
    if not c.USE_NORMALIZED_BOX2D_AS_FEATS:
        norm_box2D = None
    elif norm_box2D is None:
        raise ValueError('USE_NORMALIZED_BOX2D_AS_FEATS needs norm_box2D = tf_util.tf_normalize_2D_bboxes(box2D_pl, img_dim_pl)')
    logits = semisup_models.v1_inst_seg(pc, None, one_hot_vec, end_points, is_training, bn_decay=bn_decay, scope='inst_seg')
    mask, mask_xyz_mean, pc_xyz, pc_xyz_stage1 = semisup_models.subtract_points_mean(pc, logits, scope='subtract_points_mean')
    stage1_center = semisup_models.v1_tnet(pc_xyz_stage1, mask, mask_xyz_mean, one_hot_vec, end_points, is_training,
                                           norm_box2D=norm_box2D, bn_decay=bn_decay, scope='tnet')
    pc_xyz_submean = semisup_models.subtract_1st_stage_center(pc_xyz, stage1_center, scope='subtract_tnet_center')
    pred_box = semisup_models.v1_box_est(pc_xyz_submean, stage1_center, mask, one_hot_vec, end_points, is_training,
                                         norm_box2D=norm_box2D, bn_decay=bn_decay, c=c, scope='box_est')
    end_points['S_pred_box'] = pred_box
    _common_end_points(ctx, end_points, logits)
    end_points['mask'] = mask
    end_points['mask_xyz_mean'] = mask_xyz_mean
    asm = ctx.assembly
    from .nets import StrongLoss
    asm.loss_op = StrongLoss(e)
    B = e.B
    T = lambda buf, shape, name: api.Tensor(ctx, buf, shape, name)
    end_points['S_pred_box_reg'] = (T(asm.loss_op.center, (B, 3), 'reg_center'), T(asm.loss_op.reg_dims, (B, 3), 'reg_dims'),
                                    T(asm.loss_op.reg_theta, (B,), 'reg_theta'))
    return (logits, pred_box), end_points


def get_semi_loss(pred, labels, end_points, reduce_loss=True, c=None):
    if c.SEMI_MODEL == 'A':
        return get_semi_loss_backbone(pred, labels, end_points, reduce_loss=reduce_loss, c=c)
    elif c.SEMI_MODEL == 'F':
        return get_semi_loss_final(pred, labels, end_points, reduce_loss=reduce_loss, c=c)
    else:
        raise Exception('Not implemented SEMI_MODEL: %s' % c.SEMI_MODEL)


def get_semi_loss_final(pred, labels, end_points, reduce_loss=True, c=None):
    """strong(F_ heads, normalised by the 3-D count) + SEMI_MULTIPLIER*(WEAK_WEIGHT_INTRACLASSVAR*intraclass +
    WEAK_WEIGHT_INACTIVE_VOLUME*inactive volume + WEAK_WEIGHT_REPROJECTION*mean reprojection) + SEMI_WEIGHT_BOXPC_FIT_LOSS*fit
    (semisup_v1_sunrgbd.py:323-421)."""
    if not reduce_loss:
        raise Exception('Not implemented')                                  # semisup_v1_sunrgbd.py:420-421
    ctx = pred[0].ctx
    m = ctx.assembly
    if 'intraclsdims_train_classes' in end_points:
        m.train_classes = list(end_points['intraclsdims_train_classes'])
    if 'inactive_vol_train_classes' in end_points:
        m.inactive_train_classes = list(end_points['inactive_vol_train_classes'])
    from .nets import WeakLoss
    if WeakLoss.active_final(c) and m.weak is None:      # reprojection of the refined box / inactive volume (semisup_v1_sunrgbd.py:345-392)
        m.weak = WeakLoss(ctx.engine)
    if m.weak is not None:
        end_points['reproj_loss'] = api.Tensor(ctx, m.weak.reproj, (ctx.engine.B,), 'reproj_loss')
    ctx.loss = api.Tensor(ctx, m.loss, (), 'semi_loss')
    B = ctx.engine.B
    end_points['loss_terms'] = api.Tensor(ctx, m.loss_op.terms, (B, 8), 'loss_terms')
    # get_iou_summary(W_pred_box, ..., 'W_') and (F_pred_box, ..., '') (semisup_v1_sunrgbd.py:414-416)
    for key, buf in (('W_iou2ds', m.W_iou2d), ('W_iou3ds', m.W_iou3d), ('iou2ds', m.loss_op.iou2d), ('iou3ds', m.loss_op.iou3d)):
        end_points[key] = api.Tensor(ctx, buf, (B,), key)
    return ctx.loss


def get_semi_loss_backbone(pred, labels, end_points, reduce_loss=True, c=None):
    """mean_b [(1-is2D)*(seg CE + strong box loss) + is2D*SEMI_MULTIPLIER*(w_r*reprojection + w_s*surface)]
    (semisup_v1_sunrgbd.py:256-321).  The weak losses (nets.WeakLoss, t3d_weak_loss) are part of the graph whenever their weight
    is non-zero -- the reference's default flags; recipe a of its README sets both to 0."""
    logits = pred[0]
    ctx = logits.ctx
    asm, B = ctx.assembly, ctx.engine.B
    T = lambda buf, shape, name: api.Tensor(ctx, buf, shape, name)
    end_points['loss_terms'] = T(asm.loss_op.terms, (B, 8), 'loss_terms')
    end_points['center'] = T(asm.loss_op.center, (B, 3), 'center')
    end_points['iou2ds'] = T(asm.loss_op.iou2d, (B,), 'iou2ds')          # get_iou_summary (semisup_v1_sunrgbd.py:236-246,316)
    end_points['iou3ds'] = T(asm.loss_op.iou3d, (B,), 'iou3ds')
    from .nets import WeakLoss
    if WeakLoss.wanted(c) and asm.weak is None:      # non-zero weight, or evaluated for the `Weak_Loss/...` summaries (c.WEAK_LOSS_SUMMARIES)
        asm.weak = WeakLoss(ctx.engine)
    if asm.weak is not None:
        end_points['reproj_loss'] = T(asm.weak.reproj, (B,), 'reproj_loss')          # weak_losses.py:226
        end_points['surface_loss'] = T(asm.weak.surface, (B,), 'surface_loss')       # weak_losses.py:257
    total = T(asm.loss_op.loss, (), 'semi_loss')
    ctx.loss = total if reduce_loss else T(asm.loss_op.total_losses, (B,), 'semi_losses')
    return ctx.loss


def convert_raw_y_box_to_reg_format(y_box, one_hot_vec):
    """GT (center, heading bin/residual, size bin/residual) -> (center, dims, theta), NumPy
    (semisup_v1_sunrgbd.py:584-608)."""
    y_centers, y_orient_cls, y_orient_reg, y_dims_cls, y_dims_reg = [np.asarray(x) for x in y_box]
    dims = np.maximum(MEAN_DIMS_ARR.astype(np.float32)[y_dims_cls] + y_dims_reg, 1e-5)
    return y_centers, dims, ORIENT_ANCHORS.astype(np.float32)[y_orient_cls] + y_orient_reg
