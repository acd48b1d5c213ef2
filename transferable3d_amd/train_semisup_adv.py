#!/usr/bin/env python3
"""Stage-c training driver (SEMI_MODEL F + frozen Box-PC net) with the reference's command line
(sunrgbd/sunrgbd_detection/train_semisup_adv.py: flags 30-60, graph 267-433, partial restore 224-237/450-467,
ALTERNATE_BATCH loop 520-620).  Synthetic frustums replace the SUN-RGBD pickles.

  python -m transferable3d_amd.train_semisup_adv --SEMI_MODEL F --BOX_PC_MASK_REPRESENTATION A --use_one_hot \
      --SEMI_TRAIN_BOX_TRAIN_CLASS_AG_TNET 1 --SEMI_TRAIN_BOX_TRAIN_CLASS_AG_BOX 1 --SEMI_BOXPC_FIT_ONLY_ON_2D_CLS 1 \
      --WEAK_WEIGHT_INTRACLASSVAR 2 --WEAK_WEIGHT_REPROJECTION 0 --SEMI_MULTIPLIER_FOR_WEAK_LOSS 0.05 \
      --init_class_ag_path logA/model_epoch_30.npz --init_boxpc_path logB/model_epoch_30.npz --num_point 1024 --num_channels 4
"""
import os
import sys
import time

import numpy as np

if __package__ in (None, ''):
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from transferable3d_amd import api, tf_util, semisup_v1_sunrgbd as MODEL        # noqa: E402
from transferable3d_amd.config import make_parser                        # noqa: E402
from transferable3d_amd.constants import type2class                      # noqa: E402
from transferable3d_amd.synthetic import make_batch                      # noqa: E402
from transferable3d_amd.tf_checkpoint import Saver, load_state, restore_model   # noqa: E402
from transferable3d_amd.train_semisup import ap_by_label_kind                        # noqa: E402

ALL_CLASSES = ['bed', 'table', 'sofa', 'chair', 'toilet', 'desk', 'dresser', 'night_stand', 'bookshelf', 'bathtub']


def build_flags(argv=None):
    cfg = make_parser()
    cfg.add_argument('--train_data', type=str, default='synthetic')
    cfg.add_argument('--gpu', type=int, default=0)
    cfg.add_argument('--model', default='semisup_v1_sunrgbd')
    cfg.add_argument('--log_dir', default='log_adv')
    cfg.add_argument('--num_point', type=int, default=2048)
    cfg.add_argument('--max_epoch', type=int, default=31)
    cfg.add_argument('--batch_size', type=int, default=32)
    cfg.add_argument('--learning_rate', type=float, default=0.001)
    cfg.add_argument('--momentum', type=float, default=0.9)
    cfg.add_argument('--optimizer', default='adam', help='adam or momentum [default: adam]')
    cfg.add_argument('--decay_step', type=int, default=800000)
    cfg.add_argument('--decay_rate', type=float, default=0.5)
    cfg.add_argument('--use_one_hot', action='store_true')
    cfg.add_argument('--use_one_hot_boxpc', action='store_true')
    cfg.add_argument('--no_rgb', action='store_true')
    cfg.add_argument('--init_class_ag_path', default=None, help='stage-a state dict (class-agnostic branch)')
    cfg.add_argument('--init_boxpc_path', default=None, help='stage-b state dict (Box-PC Fit net)')
    cfg.add_argument('--restore_model_path', default=None)
    cfg.add_argument('--eval_file', default=None, help='held-out frustum file of the reference for eval_one_epoch (with --frustum_file / --device_data)')
    cfg.add_argument('--frustum_file', default=None, help='train from a frustum file of the reference (frustums/*.zip.pickle) held in HBM')
    cfg.add_argument('--eval_batches', type=int, default=0, help='held-out synthetic batches evaluated after every epoch (eval_one_epoch)')
    cfg.add_argument('--ckpt_format', default='npz', choices=['npz', 'tf'], help='tf: TensorFlow Saver bundle')
    cfg.add_argument('--synthetic', action='store_true')
    cfg.add_argument('--num_channels', type=int, default=None)
    cfg.add_argument('--steps_per_epoch', type=int, default=100)
    cfg.add_argument('--device_data', type=int, default=0, metavar='F',
                     help='keep a synthetic data set of F ragged frustums in HBM; batches (ALTERNATE_BATCH: weak / strong on '
                          'alternate steps) are assembled on the device by t3d_batch_assemble')
    cfg.add_argument('--seed', type=int, default=0)
    cfg.add_argument('--dtype', default='f32', choices=['f32', 'bf16'],
                     help='element type of the per-point layer tensors and GEMM operands (bf16: BASELINE configs[4]; weights, statistics, heads, losses and Adam stay fp32)')
    FLAGS = cfg.parse_special_args(argv)
    FLAGS.NUM_CHANNELS = FLAGS.num_channels if FLAGS.num_channels else (3 if FLAGS.no_rgb else 6)
    FLAGS.TEST_CLS = FLAGS.SUNRGBD_SEMI_TEST_CLS
    return FLAGS


def load_variable_scopes_from_ckpt(vars_, path, scope, adam_scopes=()):
    """train_semisup_adv.py:224-237, 450-457: restore every variable under `scope/` from a checkpoint whose names lack that prefix
    (stage-a / stage-b checkpoints are saved without `class_agnostic/` / `D_boxpc_branch/`).  As `Saver.restore` does, a variable
    of the scope that the checkpoint does not hold is an error (a wrong or swapped --init_*_path must not train from random
    initial weights).  The reference builds its restore list with `trainable_only=False` AFTER `minimize()`, so the Adam slots
    `<var>/Adam`, `<var>/Adam_1` of the sub-scopes that stage c trains (`adam_scopes`) are part of it: they are copied into the
    optimiser state whenever the checkpoint holds them (Saver bundles do; a plain .npz state dict of weights does not)."""
    import torch
    sd = load_state(path)
    n, missing = 0, []
    for name in list(vars_.index):
        if not name.startswith(scope + '/'):
            continue
        src = name[len(scope) + 1:]
        if src not in sd:
            missing.append(src)
            continue
        vars_.load_state_dict({name: sd[src]})
        n += 1
        off, shape, trainable = vars_.index[name]
        if trainable and src + '/Adam' in sd and any(name.startswith(a) for a in adam_scopes):
            k = int(np.prod(shape))
            vars_.adam_m[off:off + k].copy_(torch.as_tensor(np.asarray(sd[src + '/Adam'], np.float32)).reshape(-1))
            vars_.adam_v[off:off + k].copy_(torch.as_tensor(np.asarray(sd[src + '/Adam_1'], np.float32)).reshape(-1))
    if missing:
        raise KeyError('checkpoint %s holds no tensor for %d variable(s) of scope %s/ (first: %s): NotFoundError in the '
                       "reference's Saver.restore" % (path, len(missing), scope, missing[:3]))
    return n


def eval_one_epoch(sess, pls, is_training_pl, logits_t, end_points, FLAGS, epoch, log, source=None):
    """train_semisup_adv.py:663-709 on held-out synthetic frustums (is_training fed False, every frustum with its 3-D label): AP
    of the intermediate `F_` boxes and of the `F2_` boxes refined by the Box-PC deltas, plus the `W_` / `F_` box IoU summaries."""
    from transferable3d_amd.constants import MEAN_DIMS_ARR, NUM_HEADING_BIN, class2type
    from transferable3d_amd.eval_det import eval_det, get_3d_box, get_ap_info
    from transferable3d_amd.test_semisup import detection_scores
    B, N, C = FLAGS.batch_size, FLAGS.num_point, FLAGS.NUM_CHANNELS
    classes = [class2type[i] for i in range(10)]
    dets, gt_all = {'F_': {}, 'F2_': {}}, {}
    heads = lambda p: [end_points[p + k] for k in ('center', 'heading_scores', 'heading_residuals', 'size_scores', 'size_residuals')]
    sums = np.zeros(4)
    log('---- EPOCH %03d EVALUATION ----' % epoch)
    for i in range(FLAGS.eval_batches):
        if source is not None:
            source.load(i)
            feed = {}
        else:
            b = make_batch(B, N, C, seed=FLAGS.seed * 1000003 + 900000 + i)
            feed = {pls[0]: b['pc'], pls[3]: b['one_hot_vec'], pls[4]: b['y_seg'], pls[5]: b['y_center'], pls[6]: b['y_orient_cls'],
                    pls[7]: b['y_orient_reg'], pls[8]: b['y_dims_cls'], pls[9]: b['y_dims_reg'], pls[17]: np.zeros(B, np.int32),
                    pls[12]: b['Rtilt'], pls[13]: b['K'], pls[14]: b['rot_frust'], pls[15]: b['box2D'], pls[16]: b['img_dim']}
        feed[is_training_pl] = False
        out = sess.run([logits_t, end_points['iou2ds'], end_points['iou3ds'], end_points['W_iou2ds'], end_points['W_iou3ds']]
                       + heads('F_') + heads('F2_'), feed_dict=feed)
        logits = out[0]
        sums += [float(np.sum(v)) for v in out[1:5]]
        x = sess.g.inputs
        lab = {k: getattr(x, k).cpu().numpy() for k in ('y_center', 'y_orient_cls', 'y_orient_reg', 'y_dims_cls', 'y_dims_reg', 'one_hot_vec')}
        for p, (cen, hs, hr, ss, sr) in (('F_', out[5:10]), ('F2_', out[10:15])):
            hc, sc, score = np.argmax(hs, 1), np.argmax(ss, 1), detection_scores(logits, hs, ss)
            for k in range(B):
                dets[p][i * B + k] = [(classes[int(np.argmax(lab['one_hot_vec'][k]))],
                                       get_3d_box(MEAN_DIMS_ARR[sc[k]] + sr[k, sc[k]], hc[k] * (2 * np.pi / NUM_HEADING_BIN) + hr[k, hc[k]], cen[k]),
                                       float(score[k]))]
        for k in range(B):
            gt_all[i * B + k] = [(classes[int(np.argmax(lab['one_hot_vec'][k]))],
                                  get_3d_box(MEAN_DIMS_ARR[int(lab['y_dims_cls'][k])] + lab['y_dims_reg'][k],
                                             int(lab['y_orient_cls'][k]) * (2 * np.pi / NUM_HEADING_BIN) + float(lab['y_orient_reg'][k]),
                                             lab['y_center'][k]))]
    n = float(FLAGS.eval_batches * B)
    log('eval box IoU (ground/3D)     : %f / %f   class-agnostic heads: %f / %f' % (sums[0] / n, sums[1] / n, sums[2] / n, sums[3] / n))
    out = {}
    for p, tag in (('F_', 'intermediate (F_)'), ('F2_', 'refined by the Box-PC deltas (F2_)')):
        _, _, ap = eval_det(dets[p], gt_all, 0.25, rt=sess.g.rt)
        out[p] = float(np.mean(list(ap.values())))
        log('%s\n%s' % (tag, get_ap_info(ap, out[p])))
        log(ap_by_label_kind(ap, FLAGS))
    return out


def train(FLAGS, rt=None, log=print):
    # data parallel (BASELINE configs[3] is the 8-GPU config): one process per GPU, every replica its own batches, ONE
    # all-reduce of the var_list's gradient range per optimiser step, rank 0 logs and checkpoints
    world, rank, pg = api.init_data_parallel(rt, FLAGS.gpu)
    if rank != 0:
        log = lambda *a, **k: None
    B, N, C = FLAGS.batch_size, FLAGS.num_point, FLAGS.NUM_CHANNELS
    os.makedirs(FLAGS.log_dir, exist_ok=True)
    if FLAGS.SEMI_TRAIN_BOXPC_MODEL or FLAGS.SEMI_ADV_ITERS_FOR_D:
        raise NotImplementedError('training the Box-PC branch in stage c is dead code in the reference (SEMI_ADV_ITERS_FOR_D = 0)')
    with api.Graph(rt=rt, seed=FLAGS.seed, inline_dropout=True, dtype=FLAGS.dtype).as_default() as g:
        pls = MODEL.placeholder_inputs(B, N, C)
        is_training_pl = api.is_training_placeholder()                    # train_semisup_adv.py:300 (is_training_D stays False)
        norm_box2D = tf_util.tf_normalize_2D_bboxes(pls[15], pls[16])       # train_semisup_adv.py:315
        pred, end_points = MODEL.get_semi_model(pls[0], pls[1], pls[2], pls[3], is_training_pl, use_one_hot=FLAGS.use_one_hot,
                                                norm_box2D=norm_box2D, c=FLAGS)
        intraclsdims_train_classes = [(cls_type in FLAGS.TEST_CLS) for cls_type in ALL_CLASSES] \
            if FLAGS.SEMI_INTRACLSDIMS_ONLY_ON_2D_CLS else [True] * len(ALL_CLASSES)
        inactive_vol_train_classes = [(cls_type in FLAGS.TEST_CLS) for cls_type in ALL_CLASSES] \
            if getattr(FLAGS, 'WEAK_INACTIVE_VOL_ONLY_ON_2D_CLS', True) else [True] * len(ALL_CLASSES)      # train_semisup_adv.py:322-325
        end_points.update({'intraclsdims_train_classes': intraclsdims_train_classes,
                           'inactive_vol_train_classes': inactive_vol_train_classes})
        semi_loss = MODEL.get_semi_loss(pred, tuple(pls[4:]), end_points, c=FLAGS)
        train_vars = ['class_dependent']
        if FLAGS.SEMI_TRAIN_BOX_TRAIN_CLASS_AG_TNET:
            train_vars.append('class_agnostic/tnet')
        if FLAGS.SEMI_TRAIN_BOX_TRAIN_CLASS_AG_BOX:
            train_vars.append('class_agnostic/box')
        train_op = api.make_optimizer(FLAGS, world_size=world).minimize(semi_loss, var_list=train_vars)      # train_semisup_adv.py:296-298, 415-422
        sess = api.Session(process_group=pg, dropout_seed=1234 + rank)
        saver = Saver(max_to_keep=5)      # train_semisup_adv.py:433
        if FLAGS.init_class_ag_path:
            log('restored %d class_agnostic variables' % load_variable_scopes_from_ckpt(
                g.vars, FLAGS.init_class_ag_path, 'class_agnostic', adam_scopes=[v for v in train_vars if v.startswith('class_agnostic')]))
        if FLAGS.init_boxpc_path:
            log('restored %d D_boxpc_branch variables' % load_variable_scopes_from_ckpt(g.vars, FLAGS.init_boxpc_path, 'D_boxpc_branch'))
        if FLAGS.restore_model_path:
            restore_model(g, FLAGS.restore_model_path)
        test_ids = [type2class[t] for t in FLAGS.TEST_CLS]
        train_ids = [i for i in range(10) if i not in test_ids]
        step, mean_loss = 0, 0.0
        iters = 2 if FLAGS.SEMI_SAMPLING_METHOD == 'ALTERNATE_BATCH' else 1
        ds = eval_source = None
        from transferable3d_amd.dataset import open_training_set
        ds = open_training_set(g.rt, FLAGS, C, classes=None, seed=FLAGS.seed + 17 * rank)
        if ds is not None:
            if iters == 2:
                ds.split_by_class(test_ids)
            g.use_device_dataset(ds, seed=FLAGS.seed * 7919 + rank, alternate=(iters == 2),
                                 equal_class_prob=float(FLAGS.SEMI_SAMPLE_EQUAL_CLASS_WITH_PROB))   # train_semisup_adv.py:557,573
        for epoch in range(FLAGS.max_epoch):
            t0, loss_sum = time.time(), 0.0
            if ds is not None:
                ds.shuffle(FLAGS.seed * 1000003 + epoch)
                n_steps, n_logged = FLAGS.steps_per_epoch * iters, 0
                for it in range(n_steps):
                    if it % 10 >= 8 or it >= n_steps - 2:              # a weak and a strong step out of every ten
                        loss_val, _ = sess.run([semi_loss, train_op])
                        loss_sum += float(loss_val)
                        n_logged += 1
                    else:
                        sess.run([train_op])
                    step += 1
                mean_loss = loss_sum / n_logged
                log('**** EPOCH %03d ****  mean loss: %f  (%.1f frustums/s, batches assembled on the device)' % (
                    epoch, mean_loss, n_steps * B * world / (time.time() - t0)))
            for _ in range(0 if ds is not None else FLAGS.steps_per_epoch):
                for iteration in range(iters):
                    b = make_batch(B, N, C, seed=FLAGS.seed * 1000003 + step * world + rank)
                    if iters == 2:                         # all-2D batch (classes without 3-D labels), then all-3D batch
                        ids = test_ids if iteration == 0 else train_ids
                        cls = np.asarray(ids)[np.random.RandomState(step * world + rank).randint(0, len(ids), size=B)]
                        b['one_hot_vec'] = np.eye(10, dtype=np.float32)[cls]
                        b['y_dims_cls'] = cls.astype(np.int32)
                        b['is_data_2D'][:] = 1 if iteration == 0 else 0
                    feed = {pls[0]: b['pc'], pls[3]: b['one_hot_vec'], pls[4]: b['y_seg'], pls[5]: b['y_center'], pls[6]: b['y_orient_cls'],
                            pls[7]: b['y_orient_reg'], pls[8]: b['y_dims_cls'], pls[9]: b['y_dims_reg'], pls[17]: b['is_data_2D'],
                            pls[12]: b['Rtilt'], pls[13]: b['K'], pls[14]: b['rot_frust'], pls[15]: b['box2D'], pls[16]: b['img_dim']}
                    feed[is_training_pl] = True
                    loss_val, _ = sess.run([semi_loss, train_op], feed_dict=feed)
                    loss_sum += float(loss_val)
                    step += 1
            if rank == 0 and (FLAGS.eval_batches > 0 or FLAGS.eval_file):
                if ds is not None and eval_source is None:
                    from transferable3d_amd.dataset import open_eval_source
                    eval_source = open_eval_source(g, FLAGS, classes=list(FLAGS.TEST_CLS))
                eval_one_epoch(sess, pls, is_training_pl, pred[0], end_points, FLAGS, epoch, log, eval_source)
            if ds is None:
                mean_loss = loss_sum / (FLAGS.steps_per_epoch * iters)
                log('**** EPOCH %03d ****  mean loss: %f  (%.1f frustums/s incl. host batch synthesis)' % (
                    epoch, mean_loss, FLAGS.steps_per_epoch * iters * B * world / (time.time() - t0)))
            if epoch % 5 == 0 and rank == 0:
                sess.check_riders()      # never checkpoint weights a timed-out rider barrier may have corrupted
                path = saver.save(FLAGS.log_dir, epoch, g, FLAGS.ckpt_format, optimizer_scopes=train_vars)
                log('Model saved in file: %s' % path)
        sess.check_riders()
        final = g.vars.state_dict()
    api.finish_data_parallel(world)
    return final, mean_loss


if __name__ == '__main__':
    train(build_flags())
