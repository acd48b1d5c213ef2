#!/usr/bin/env python3
"""Stage-a training driver (SEMI_MODEL A) with the reference's command line
(sunrgbd/sunrgbd_detection/train_semisup.py: flags 28-48, graph build 199-260, epoch loop 305-318, train_one_epoch
320-436).  Differences, all forced by the environment and documented in DESIGN.md:
  * `--synthetic` batches (transferable3d_amd/synthetic.py) replace the SUN-RGBD frustum pickles, which are not
    available; `--train_data` is therefore optional.
  * one process per GPU: launch with `python -m torch.distributed.run --nproc-per-node N ... train_semisup.py` for
    data-parallel training (gradient all-reduce on RCCL); `--gpu` selects the device in the single-process case.
  * checkpoints are `.npz` state dicts keyed by the reference's TF variable names or, with
    `--ckpt_format tf`, TensorFlow Saver bundles (`model_epoch_<n>.ckpt.index/.data-*`, tf_checkpoint.py); `--restore_model_path`
    takes either.

Example (README.md:58-67 recipe a, synthetic data):
  python -m transferable3d_amd.train_semisup --SEMI_MODEL A --WEAK_WEIGHT_REPROJECTION 0 --WEAK_WEIGHT_SURFACE 0 \
      --num_point 1024 --no_rgb_channels 4 --max_epoch 1 --steps_per_epoch 100
"""
import os
import sys
import time

import numpy as np

if __package__ in (None, ''):
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from transferable3d_amd import api, tf_util, semisup_v1_sunrgbd as MODEL       # noqa: E402
from transferable3d_amd.config import make_parser                        # noqa: E402
from transferable3d_amd.synthetic import make_batch                      # noqa: E402
from transferable3d_amd.tf_checkpoint import Saver, restore_model   # noqa: E402


def build_flags(argv=None):
    cfg = make_parser()
    cfg.add_argument('--train_data', type=str, default='synthetic', choices=['train_mini', 'train_aug5x', 'trainval_aug5x', 'synthetic'])
    cfg.add_argument('--train_data3D_keep_prob', type=float, default=1)
    cfg.add_argument('--add3D_for_classes2D_prob', type=float, default=-1)
    cfg.add_argument('--gpu', type=int, default=0, help='GPU to use [default: GPU 0]')
    cfg.add_argument('--model', default='semisup_v1_sunrgbd', help='Model name [default: model]')
    cfg.add_argument('--log_dir', default='log', help='Log dir [default: log]')
    cfg.add_argument('--num_point', type=int, default=2048, help='Point Number [default: 2048]')
    cfg.add_argument('--max_epoch', type=int, default=31, help='Epoch to run [default: 51]')
    cfg.add_argument('--batch_size', type=int, default=32, help='Batch Size during training [default: 32]')
    cfg.add_argument('--learning_rate', type=float, default=0.001, help='Initial learning rate [default: 0.001]')
    cfg.add_argument('--momentum', type=float, default=0.9)
    cfg.add_argument('--optimizer', default='adam', help='adam or momentum [default: adam]')
    cfg.add_argument('--decay_step', type=int, default=800000, help='Decay step for lr decay [default: 200000]')
    cfg.add_argument('--decay_rate', type=float, default=0.5, help='Decay rate for lr decay [default: 0.7]')
    cfg.add_argument('--use_mini', action='store_true')
    cfg.add_argument('--use_one_hot', action='store_true', help='Use one hot vector during training')
    cfg.add_argument('--no_aug', action='store_true')
    cfg.add_argument('--no_rgb', action='store_true', help='Only use XYZ for training')
    cfg.add_argument('--init_model_path', default=None)
    cfg.add_argument('--restore_model_path', default=None, help='Restore model path e.g. log/model_epoch_0.ckpt or .npz')
    cfg.add_argument('--ckpt_format', default='npz', choices=['npz', 'tf'], help='tf: TensorFlow Saver bundle')
    # additions
    cfg.add_argument('--synthetic', action='store_true', help='synthetic frustums (the only data source available)')
    cfg.add_argument('--num_channels', type=int, default=None, help='override point channels (reference: 6, or 3 with --no_rgb)')
    cfg.add_argument('--steps_per_epoch', type=int, default=None,
                     help='steps of an epoch [default: one pass over the data set, len / (batch_size * replicas) as in the reference; '
                          '100 for --synthetic batches]')
    cfg.add_argument('--device_data', type=int, default=0, metavar='F',
                     help='keep a synthetic data set of F ragged frustums in HBM and assemble every batch on the device '
                          '(t3d_batch_assemble: resample / centre-view rotation / flip / shift / labels)')
    cfg.add_argument('--seed', type=int, default=0)
    cfg.add_argument('--dtype', default='f32', choices=['f32', 'bf16'],
                     help='element type of the per-point layer tensors and GEMM operands (bf16: BASELINE configs[4]; weights, statistics, heads, losses and Adam stay fp32)')
    cfg.add_argument('--eval_file', default=None, help='held-out frustum file of the reference for eval_one_epoch (with --frustum_file / --device_data)')
    cfg.add_argument('--frustum_file', default=None, help='train from a frustum file of the reference (frustums/*.zip.pickle) held in HBM')
    cfg.add_argument('--eval_batches', type=int, default=0, help='held-out synthetic batches evaluated after every epoch (eval_one_epoch)')
    cfg.add_argument('--weak_loss_summaries', action='store_true',
                     help='evaluate the weak reprojection / surface losses also when both of their weights are 0, for the reference\'s '
                          'Weak_Loss/... summaries (semisup_v1_sunrgbd.py:270-293; logged per epoch).  Off by default since round 6: the two '
                          'launches cost 22 us of a 1.2 ms step (+1.9 %%) and change nothing that is trained -- SURVEY Appendix E item 4 '
                          'sanctions skipping zero-weight terms; with a non-zero weight they are always evaluated')
    cfg.add_argument('--no_weak_loss_summaries', action='store_true', help='(accepted for round-5 command lines: the default now)')
    FLAGS = cfg.parse_special_args(argv)
    FLAGS.WEAK_LOSS_SUMMARIES = bool(FLAGS.weak_loss_summaries) and not FLAGS.no_weak_loss_summaries
    FLAGS.NUM_CHANNELS = FLAGS.num_channels if FLAGS.num_channels else (3 if FLAGS.no_rgb else 6)
    return FLAGS


def eval_one_epoch(sess, ops, FLAGS, epoch, log, source=None):
    """train_semisup.py:436-545 on held-out synthetic frustums: the SAME graph with is_training fed False (moving batch-norm
    statistics, no dropout, no parameter or EMA update) -- loss, segmentation accuracy / class accuracy / IoU, box IoU."""
    pls, is_training_pl, semi_loss, n_correct, end_points = ops
    pc_pl, _, _, one_hot_vec_pl, y_seg_pl, y_centers_pl, y_orient_cls_pl, y_orient_reg_pl, y_dims_cls_pl, y_dims_reg_pl = pls[:10]
    is_data_2D_pl = pls[-1]
    B, N, C = FLAGS.batch_size, FLAGS.num_point, FLAGS.NUM_CHANNELS
    from transferable3d_amd.constants import MEAN_DIMS_ARR, NUM_HEADING_BIN, class2type
    from transferable3d_amd.eval_det import eval_det, get_3d_box, get_ap_info
    from transferable3d_amd.test_semisup import detection_scores
    classes = [class2type[i] for i in range(10)]
    det_all, gt_all = {}, {}
    log('---- EPOCH %03d EVALUATION ----' % epoch)
    loss_sum = iou2 = iou3 = 0.0
    seen, correct = np.zeros(2), np.zeros(2)
    shape_ious = []
    for i in range(FLAGS.eval_batches):
        if source is not None:                               # batch assembled on the device: nothing but the mode is fed
            lab = source.load(i)
            feed = {is_training_pl: False}
        else:
            b = make_batch(B, N, C, seed=FLAGS.seed * 1000003 + 900000 + i)
            lab = b['y_seg']
            feed = {pc_pl: b['pc'], one_hot_vec_pl: b['one_hot_vec'], y_seg_pl: b['y_seg'], y_centers_pl: b['y_center'],
                    y_orient_cls_pl: b['y_orient_cls'], y_orient_reg_pl: b['y_orient_reg'], y_dims_cls_pl: b['y_dims_cls'],
                    y_dims_reg_pl: b['y_dims_reg'], is_data_2D_pl: np.zeros(B, np.int32), is_training_pl: False,
                    pls[12]: b['Rtilt'], pls[13]: b['K'], pls[14]: b['rot_frust'], pls[15]: b['box2D'], pls[16]: b['img_dim']}
        heads = [end_points[k] for k in ('center', 'heading_scores', 'heading_residuals', 'size_scores', 'size_residuals')]
        loss_val, logits, i2, i3, cen, hs, hr, ss, sr = sess.run(
            [semi_loss, end_points_logits(end_points, sess), end_points['iou2ds'], end_points['iou3ds']] + heads, feed_dict=feed)
        pred = np.argmax(logits, 2)
        # detections of this batch and their label boxes (main_batch + evaluate_predictions, train_semisup.py:459-466); every
        # frustum is its own image, boxes stay in the frustum's centre view
        x = sess.g.inputs
        lab_box = {k: getattr(x, k).cpu().numpy() for k in ('y_center', 'y_orient_cls', 'y_orient_reg', 'y_dims_cls', 'y_dims_reg', 'one_hot_vec')}
        hc, sc = np.argmax(hs, 1), np.argmax(ss, 1)
        score = detection_scores(logits, hs, ss)
        for k in range(B):
            cls = classes[int(np.argmax(lab_box['one_hot_vec'][k]))]
            img = i * B + k
            det_all[img] = [(cls, get_3d_box(MEAN_DIMS_ARR[sc[k]] + sr[k, sc[k]], hc[k] * (2 * np.pi / NUM_HEADING_BIN) + hr[k, hc[k]], cen[k]),
                             float(score[k]))]
            gt_all[img] = [(cls, get_3d_box(MEAN_DIMS_ARR[int(lab_box['y_dims_cls'][k])] + lab_box['y_dims_reg'][k],
                                            int(lab_box['y_orient_cls'][k]) * (2 * np.pi / NUM_HEADING_BIN) + float(lab_box['y_orient_reg'][k]),
                                            lab_box['y_center'][k]))]
        loss_sum += float(loss_val)
        iou2, iou3 = iou2 + float(np.sum(i2)), iou3 + float(np.sum(i3))
        for l in range(2):
            seen[l] += np.sum(lab == l)
            correct[l] += np.sum((pred == l) & (lab == l))
        for k in range(B):                                   # per-frustum part IoU; an absent part that is not predicted counts 1
            shape_ious.append([1.0 if not (np.any(lab[k] == l) or np.any(pred[k] == l)) else
                               np.sum((lab[k] == l) & (pred[k] == l)) / float(np.sum((lab[k] == l) | (pred[k] == l))) for l in range(2)])
    n = float(FLAGS.eval_batches)
    log('eval mean loss: %f' % (loss_sum / n))
    log('eval accuracy: %f' % (correct.sum() / seen.sum()))
    log('eval avg class acc: %f' % np.mean(correct / np.maximum(seen, 1)))
    log('eval mIoU: %f' % np.mean(shape_ious))
    log('eval box IoU (ground/3D)     : %f / %f' % (iou2 / (n * B), iou3 / (n * B)))
    _, _, ap = eval_det(det_all, gt_all, 0.25, rt=sess.g.rt)
    log(get_ap_info(ap, float(np.mean(list(ap.values())))))
    log(ap_by_label_kind(ap, FLAGS))
    return loss_sum / n


def ap_by_label_kind(ap, FLAGS):
    """Mean AP over the classes trained with 3-D labels and over the classes with 2-D labels only (the reference evaluates the
    latter: TEST_DATASET holds FLAGS.TEST_CLS, train_semisup.py:113-117)."""
    weak = [ap[k] for k in ap if k in FLAGS.SUNRGBD_SEMI_TEST_CLS]
    strong = [ap[k] for k in ap if k not in FLAGS.SUNRGBD_SEMI_TEST_CLS]
    m = lambda v: 100.0 * float(np.mean(v)) if v else float('nan')
    return '    Mean AP, classes with 3-D labels: %.1f   classes with 2-D labels only: %.1f' % (m(strong), m(weak))


def end_points_logits(end_points, sess):
    g = sess.g
    return api.Tensor(g, g.assembly.seg.logits, (g.engine.B, g.engine.rpf, 2), 'logits')


def train(FLAGS, rt=None, log=print):
    world, rank, pg = api.init_data_parallel(rt, FLAGS.gpu)
    B, N, C = FLAGS.batch_size, FLAGS.num_point, FLAGS.NUM_CHANNELS
    os.makedirs(FLAGS.log_dir, exist_ok=True)
    if rank == 0:
        log(FLAGS.config_str)
    with api.Graph(rt=rt, seed=FLAGS.seed, inline_dropout=True, dtype=FLAGS.dtype).as_default() as g:
        pls = MODEL.placeholder_inputs(B, N, C)
        pc_pl, bg_pc_pl, img_pl, one_hot_vec_pl, y_seg_pl, y_centers_pl, y_orient_cls_pl, y_orient_reg_pl, y_dims_cls_pl, \
            y_dims_reg_pl, R0_rect_pl, P_pl, Rtilt_pl, K_pl, rot_frust_pl, box2D_pl, img_dim_pl, is_data_2D_pl = pls
        is_training_pl = api.is_training_placeholder()                     # train_semisup.py:210
        norm_box2D = tf_util.tf_normalize_2D_bboxes(box2D_pl, img_dim_pl)   # train_semisup.py:240 (dropped unless USE_NORMALIZED_BOX2D_AS_FEATS)
        pred, end_points = MODEL.get_semi_model(pc_pl, bg_pc_pl, img_pl, one_hot_vec_pl, is_training_pl, use_one_hot=FLAGS.use_one_hot,
                                                norm_box2D=norm_box2D, bn_decay=None, c=FLAGS)
        labels = (y_seg_pl, y_centers_pl, y_orient_cls_pl, y_orient_reg_pl, y_dims_cls_pl, y_dims_reg_pl, R0_rect_pl, P_pl,
                  Rtilt_pl, K_pl, rot_frust_pl, box2D_pl, img_dim_pl, is_data_2D_pl)
        semi_loss = MODEL.get_semi_loss(pred, labels, end_points, c=FLAGS)
        optimizer = api.make_optimizer(FLAGS, world_size=world)      # train_semisup.py:226-231 (--optimizer adam | momentum)
        train_op = optimizer.minimize(semi_loss)
        sess = api.Session(process_group=pg, dropout_seed=1234 + rank)
        saver = Saver(max_to_keep=5)      # train_semisup.py:259
        if FLAGS.restore_model_path:
            restore_model(g, FLAGS.restore_model_path)
        n_correct = api.Tensor(g, g.assembly.seg.n_correct, (1,), 'n_correct')
        step = 0
        ds = eval_source = None
        from transferable3d_amd.dataset import open_training_set
        ds = open_training_set(g.rt, FLAGS, C, classes=list(FLAGS.SUNRGBD_SEMI_TRAIN_CLS) + list(FLAGS.SUNRGBD_SEMI_TEST_CLS),
                               seed=FLAGS.seed + 17 * rank)
        if ds is not None:
            # SEMI_SAMPLING_METHOD BATCH over the combined data set (train_semisup.py:97-110, 343-349): frustums of the classes that
            # have 2-D labels only (SUNRGBD_SEMI_TEST_CLS) run through the net with is_data_2D = 1 -- no strong loss, their points
            # still enter the batch statistics.  (SEMI_USE_LABELS2D_OF_CLASSES3D also repeats the 3-D-label frustums as zero-loss
            # 2-D samples; not reproduced.)
            from transferable3d_amd.constants import type2class
            ds.mark_2d_classes([type2class[t] for t in FLAGS.SUNRGBD_SEMI_TEST_CLS])
            g.use_device_dataset(ds, seed=FLAGS.seed * 7919 + rank)
            # an epoch = ONE pass (train_semisup.py:330-349: num_batches = len(TRAIN_DATASET) / BATCH_SIZE); data parallel: every
            # replica walks its own slice of the common epoch permutation, so the replicas see disjoint frustums
            n = ds.partition(rank, world, B, FLAGS.steps_per_epoch)
            if FLAGS.steps_per_epoch and n < FLAGS.steps_per_epoch and rank == 0:
                log('--steps_per_epoch %d exceeds one pass over the data set: %d steps per epoch' % (FLAGS.steps_per_epoch, n))
            FLAGS.steps_per_epoch = n
        elif not FLAGS.steps_per_epoch:
            FLAGS.steps_per_epoch = 100
        for epoch in range(FLAGS.max_epoch):
            t0 = time.time()
            loss_sum, correct = 0.0, 0.0
            iou2_sum = iou3_sum = 0.0                 # 'Strong Box IoU (ground/3D)' of train_semisup.py:414-431
            iou2ds, iou3ds = end_points['iou2ds'], end_points['iou3ds']
            # tf.summary.scalar('Weak_Loss/reprojection_loss' | 'Weak_Loss/surface_loss') of get_semi_loss_backbone: batch means, logged
            # per epoch (evaluated at zero weight too: --no_weak_loss_summaries)
            weak_t = [end_points[k] for k in ('reproj_loss', 'surface_loss') if k in end_points]
            weak_sum, weak_n = np.zeros(len(weak_t)), 0
            if ds is not None:
                # device pipeline: nothing is fed; the loss is fetched (a D2H sync) every 10th step only
                ds.shuffle(FLAGS.seed * 1000003 + epoch)      # train_semisup.py:343 (the same permutation on every replica)
                n_logged = 0
                for it in range(FLAGS.steps_per_epoch):
                    if it % 10 == 9 or it == FLAGS.steps_per_epoch - 1:
                        loss_val, nc, i2, i3, _, *wk = sess.run([semi_loss, n_correct, iou2ds, iou3ds, train_op] + weak_t)
                        weak_sum, weak_n = weak_sum + np.array([float(np.mean(v)) for v in wk]), weak_n + 1
                        loss_sum += float(loss_val)
                        correct += float(nc[0])
                        iou2_sum, iou3_sum = iou2_sum + float(np.sum(i2)), iou3_sum + float(np.sum(i3))
                        n_logged += 1
                    else:
                        sess.run([train_op])
                    step += 1
                if rank == 0:
                    log('**** EPOCH %03d ****  mean loss: %f  accuracy: %f  (%.1f frustums/s, batches assembled on the device)' % (
                        epoch, loss_sum / n_logged, correct / (n_logged * B * N), FLAGS.steps_per_epoch * B * world / (time.time() - t0)))
                    log('Strong Box IoU (ground/3D): %f / %f' % (iou2_sum / (n_logged * B), iou3_sum / (n_logged * B)))
                loss_sum = loss_sum / n_logged * FLAGS.steps_per_epoch
            for it in range(0 if ds is not None else FLAGS.steps_per_epoch):
                batch = make_batch(B, N, C, seed=FLAGS.seed * 1000003 + step * world + rank)
                feed = {pc_pl: batch['pc'], one_hot_vec_pl: batch['one_hot_vec'], y_seg_pl: batch['y_seg'],
                        y_centers_pl: batch['y_center'], y_orient_cls_pl: batch['y_orient_cls'],
                        y_orient_reg_pl: batch['y_orient_reg'], y_dims_cls_pl: batch['y_dims_cls'],
                        y_dims_reg_pl: batch['y_dims_reg'], is_data_2D_pl: batch['is_data_2D'],
                        # camera side of the weak losses (their default weights are non-zero: models/config.py:112-113)
                        Rtilt_pl: batch['Rtilt'], K_pl: batch['K'], rot_frust_pl: batch['rot_frust'], box2D_pl: batch['box2D'],
                        img_dim_pl: batch['img_dim']}
                feed[is_training_pl] = True
                loss_val, nc, i2, i3, _, *wk = sess.run([semi_loss, n_correct, iou2ds, iou3ds, train_op] + weak_t, feed_dict=feed)
                weak_sum, weak_n = weak_sum + np.array([float(np.mean(v)) for v in wk]), weak_n + 1
                loss_sum += float(loss_val)
                correct += float(nc[0])
                iou2_sum, iou3_sum = iou2_sum + float(np.sum(i2)), iou3_sum + float(np.sum(i3))
                step += 1
            if rank == 0 and ds is None:
                log('**** EPOCH %03d ****  mean loss: %f  accuracy: %f  (%.1f frustums/s incl. host batch synthesis)' % (
                    epoch, loss_sum / FLAGS.steps_per_epoch, correct / (FLAGS.steps_per_epoch * B * N),
                    FLAGS.steps_per_epoch * B * world / (time.time() - t0)))
                log('Strong Box IoU (ground/3D): %f / %f' % (iou2_sum / (FLAGS.steps_per_epoch * B), iou3_sum / (FLAGS.steps_per_epoch * B)))
            if rank == 0 and weak_n and len(weak_t) == 2:
                log('Weak_Loss/reprojection_loss: %f  Weak_Loss/surface_loss: %f  (weights %g / %g)' % (
                    weak_sum[0] / weak_n, weak_sum[1] / weak_n, FLAGS.WEAK_WEIGHT_REPROJECTION, FLAGS.WEAK_WEIGHT_SURFACE))
            if rank == 0 and (FLAGS.eval_batches > 0 or FLAGS.eval_file):
                if ds is not None and eval_source is None:
                    from transferable3d_amd.dataset import open_eval_source
                    eval_source = open_eval_source(g, FLAGS, classes=list(FLAGS.SUNRGBD_SEMI_TEST_CLS))   # TEST_DATASET, train_semisup.py:113
                eval_one_epoch(sess, (pls, is_training_pl, semi_loss, n_correct, end_points), FLAGS, epoch, log, eval_source)
            if rank == 0:
                if epoch % 5 == 0:                       # train_semisup.py:316-318
                    sess.check_riders()      # never checkpoint weights a timed-out rider barrier may have corrupted
                    path = saver.save(FLAGS.log_dir, epoch, g, FLAGS.ckpt_format)
                    log('Model saved in file: %s' % path)
        sess.check_riders()
        final = g.vars.state_dict()
    api.finish_data_parallel(world)
    return final, loss_sum / max(FLAGS.steps_per_epoch, 1)


if __name__ == '__main__':
    train(build_flags())
