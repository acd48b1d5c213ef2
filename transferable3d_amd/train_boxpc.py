#!/usr/bin/env python3
"""Stage-b training driver (Box-PC Fit net) with the reference's command line
(sunrgbd/sunrgbd_detection/train_boxpc.py: flags 28-47, graph 219-261, loop 301-366).  `--device_data F` keeps a
frustum set in HBM and makes every sample on the device: batch assembly, class-balanced composition and the perturbed box with its
IoU target (box_pc_fit_dataset.py; t3d_batch_assemble, t3d_sample_equal_classes, t3d_boxpc_perturb).  Without it, `--synthetic`
host batches carry a perturbed box in label form with drawn IoU / delta labels (transferable3d_amd/synthetic.py).  Per epoch the
driver prints the reference's statistics (boxpc_stats.py): precision / recall / F1 of the fit decision per class and the 3-D IoU
with the label box before / after the predicted deltas; `--eval_batches n` adds eval_one_epoch on held-out samples.

  python -m transferable3d_amd.train_boxpc --BOX_PC_MASK_REPRESENTATION A --BOXPC_WEIGHT_DELTA 4 --num_point 1024 \
      --num_channels 4 --max_epoch 1 --steps_per_epoch 100
"""
import os
import sys
import time

import numpy as np

if __package__ in (None, ''):
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from transferable3d_amd import api, boxpc_sunrgbd as MODEL            # noqa: E402
from transferable3d_amd.config import make_parser                       # noqa: E402
from transferable3d_amd.synthetic import make_batch                     # noqa: E402
from transferable3d_amd.tf_checkpoint import Saver, restore_model  # noqa: E402


def build_flags(argv=None):
    cfg = make_parser()
    cfg.add_argument('--train_data', type=str, default='synthetic')
    cfg.add_argument('--gpu', type=int, default=0)
    cfg.add_argument('--model', default='boxpc_sunrgbd')
    cfg.add_argument('--log_dir', default='log_boxpc')
    cfg.add_argument('--num_point', type=int, default=2048)
    cfg.add_argument('--max_epoch', type=int, default=31)
    cfg.add_argument('--batch_size', type=int, default=32)
    cfg.add_argument('--learning_rate', type=float, default=0.001)
    cfg.add_argument('--momentum', type=float, default=0.9)
    cfg.add_argument('--optimizer', default='adam', help='adam or momentum [default: adam]')
    cfg.add_argument('--decay_step', type=int, default=800000)
    cfg.add_argument('--decay_rate', type=float, default=0.5)
    cfg.add_argument('--use_one_hot', action='store_true')
    cfg.add_argument('--no_rgb', action='store_true')
    cfg.add_argument('--restore_model_path', default=None)
    cfg.add_argument('--ckpt_format', default='npz', choices=['npz', 'tf'], help='tf: TensorFlow Saver bundle')
    cfg.add_argument('--synthetic', action='store_true')
    cfg.add_argument('--num_channels', type=int, default=None)
    cfg.add_argument('--steps_per_epoch', type=int, default=None, help='[default: one pass over the data set; 100 for --synthetic batches]')
    cfg.add_argument('--seed', type=int, default=0)
    cfg.add_argument('--dtype', default='f32', choices=['f32', 'bf16'],
                     help='element type of the per-point layer tensors and GEMM operands (bf16: BASELINE configs[4]; weights, statistics, heads, losses and Adam stay fp32)')
    cfg.add_argument('--eval_file', default=None, help='held-out frustum file of the reference for eval_one_epoch (with --frustum_file / --device_data)')
    cfg.add_argument('--frustum_file', default=None, help='train from a frustum file of the reference (frustums/*.zip.pickle) held in HBM')
    cfg.add_argument('--eval_batches', type=int, default=0, help='held-out synthetic batches evaluated after every epoch (eval_one_epoch)')
    cfg.add_argument('--device_data', type=int, default=0, metavar='F',
                     help='F > 0: F synthetic frustums resident in HBM; batches and the perturbed-box samples are made on the device')
    FLAGS = cfg.parse_special_args(argv)
    FLAGS.NUM_CHANNELS = FLAGS.num_channels if FLAGS.num_channels else (3 if FLAGS.no_rgb else 6)
    return FLAGS


def train(FLAGS, rt=None, log=print):
    # data parallel: one process per GPU, own samples per replica, one all-reduce of the gradients per step; rank 0 logs
    world, rank, pg = api.init_data_parallel(rt, FLAGS.gpu)
    if rank != 0:
        log = lambda *a, **k: None
    B, N, C = FLAGS.batch_size, FLAGS.num_point, FLAGS.NUM_CHANNELS
    os.makedirs(FLAGS.log_dir, exist_ok=True)
    with api.Graph(rt=rt, seed=FLAGS.seed, dtype=FLAGS.dtype).as_default() as g:
        pls = MODEL.placeholder_inputs(B, N, C)
        pc_pl, one_hot_vec_pl, y_seg_pl, x_center_pl, x_orient_cls_pl, x_orient_reg_pl, x_dims_cls_pl, x_dims_reg_pl, \
            y_box_iou_pl, y_center_delta_pl, y_dims_delta_pl, y_orient_delta_pl = pls
        box_reg = MODEL.convert_raw_y_box_to_reg_format((x_center_pl, x_orient_cls_pl, x_orient_reg_pl, x_dims_cls_pl, x_dims_reg_pl),
                                                        one_hot_vec_pl)
        is_training_pl = api.is_training_placeholder()                   # train_boxpc.py:228
        pred, end_points = MODEL.get_model((box_reg, pc_pl), is_training_pl, one_hot_vec_pl, use_one_hot_vec=FLAGS.use_one_hot, c=FLAGS)
        loss = MODEL.get_loss(pred, (y_box_iou_pl, (y_center_delta_pl, y_dims_delta_pl, y_orient_delta_pl)), end_points, c=FLAGS)
        train_op = api.make_optimizer(FLAGS, world_size=world).minimize(loss)      # train_boxpc.py:245-250
        sess = api.Session(process_group=pg, dropout_seed=1234 + rank)
        saver = Saver(max_to_keep=5)      # train_boxpc.py:261
        if FLAGS.restore_model_path:
            restore_model(g, FLAGS.restore_model_path)
        step, mean_loss = 0, 0.0
        from transferable3d_amd.boxpc_stats import ALL_CLASSES, BoxDeltaIOUStats, ClassificationStats, record_batch
        fit_lo = float(FLAGS.BOXPC_FIT_BOUNDS[0])
        fetch_stats = [loss, end_points['pred_boxpc_fit'], end_points['boxpc_delta_center'], end_points['boxpc_delta_size'],
                       end_points['boxpc_delta_angle']]

        def new_stats():
            return ClassificationStats(ALL_CLASSES), BoxDeltaIOUStats(ALL_CLASSES, g.rt), BoxDeltaIOUStats(ALL_CLASSES, g.rt)

        def report(tag, stats):
            cls_stats, pos, neg = stats
            log('%s mean loss: %f' % (tag, cls_stats.get_mean_loss()))
            log(cls_stats.summarize_stats(cls_stats.get_batch_stats()))
            log('Box IoU before / after the predicted deltas, fit samples (IoU >= %.2f):' % fit_lo)
            log(pos.summarize_stats(pos.get_batch_stats()))
            log('Box IoU before / after the predicted deltas, no-fit samples:')
            log(neg.summarize_stats(neg.get_batch_stats()))

        def eval_one_epoch(epoch):
            """train_boxpc.py:398-487: held-out samples, is_training fed False."""
            log('---- EPOCH %03d EVALUATION ----' % epoch)
            stats = new_stats()
            for i in range(FLAGS.eval_batches):
                if eval_source is not None:                  # held-out frustums, samples made on the device: only the mode is fed
                    eval_source.load(i)
                    feed = {}
                else:
                    feed = feed_of(make_batch(B, N, C, seed=FLAGS.seed * 1000003 + 900000 + i, boxpc=True))
                feed[is_training_pl] = False
                out = sess.run(fetch_stats, feed_dict=feed)
                record_batch(stats[0], stats[1], stats[2], fit_lo, out[0], out[1], out[2:5], g.inputs)
            report('eval', stats)

        def feed_of(b):
            return {pc_pl: b['pc'], one_hot_vec_pl: b['one_hot_vec'], x_center_pl: b['y_center'], x_orient_cls_pl: b['y_orient_cls'],
                    x_orient_reg_pl: b['y_orient_reg'], x_dims_cls_pl: b['y_dims_cls'], x_dims_reg_pl: b['y_dims_reg'],
                    y_box_iou_pl: b['y_box_iou'], y_center_delta_pl: b['y_center_delta'], y_dims_delta_pl: b['y_dims_delta'],
                    y_orient_delta_pl: b['y_orient_delta']}
        ds = eval_source = None
        from transferable3d_amd.dataset import open_eval_source, open_training_set
        # BoxPCFitDataset(classes=FLAGS.TRAIN_CLS, ...) (train_boxpc.py:100-108)
        ds = open_training_set(g.rt, FLAGS, C, classes=list(FLAGS.SUNRGBD_SEMI_TRAIN_CLS) if FLAGS.frustum_file else None, seed=FLAGS.seed + 17 * rank)
        if ds is not None:
            if FLAGS.eval_batches > 0 or FLAGS.eval_file:
                eval_source = open_eval_source(g, FLAGS, classes=list(FLAGS.SUNRGBD_SEMI_TRAIN_CLS), boxpc_perturb=FLAGS)
            # BOXPC_SAMPLING_METHOD 'SAMPLE': class-balanced batches with probability BOXPC_SAMPLE_EQUAL_CLASS_WITH_PROB
            # (train_boxpc.py:323-328); 'BATCH': the epoch permutation
            eq = float(FLAGS.BOXPC_SAMPLE_EQUAL_CLASS_WITH_PROB) if FLAGS.BOXPC_SAMPLING_METHOD == 'SAMPLE' else 0.0
            g.use_device_dataset(ds, seed=FLAGS.seed * 7919 + rank, boxpc_perturb=FLAGS, equal_class_prob=eq)
            if eq == 0.0:            # 'BATCH': an epoch is at most one pass over the data set (whole batches, train_boxpc.py:316-321)
                FLAGS.steps_per_epoch = ds.partition(rank, world, B, FLAGS.steps_per_epoch)
        if not FLAGS.steps_per_epoch:
            FLAGS.steps_per_epoch = 100
        for epoch in range(FLAGS.max_epoch):
            t0, loss_sum = time.time(), 0.0
            if ds is not None:
                # box_pc_fit_dataset.py 'BATCH' sampling: an epoch permutation; the loss is fetched every 10th step only
                ds.shuffle(FLAGS.seed * 1000003 + epoch)
                n_logged = 0
                stats = new_stats()
                for it in range(FLAGS.steps_per_epoch):
                    if it % 10 == 9 or it == FLAGS.steps_per_epoch - 1:
                        out = sess.run(fetch_stats + [train_op])
                        record_batch(stats[0], stats[1], stats[2], fit_lo, out[0], out[1], out[2:5], g.inputs)
                        loss_sum += float(out[0])
                        n_logged += 1
                    else:
                        sess.run([train_op])
                    step += 1
                mean_loss = loss_sum / n_logged
                log('**** EPOCH %03d ****  mean loss: %f  (%.1f frustums/s, samples made on the device)' % (
                    epoch, mean_loss, FLAGS.steps_per_epoch * B / (time.time() - t0)))
                report('train (every 10th step)', stats)
                if FLAGS.eval_batches > 0 and rank == 0:
                    eval_one_epoch(epoch)
                if epoch % 5 == 0 and rank == 0:
                    sess.check_riders()      # never checkpoint weights a timed-out rider barrier may have corrupted
                    log('Model saved in file: %s' % saver.save(FLAGS.log_dir, epoch, g, FLAGS.ckpt_format))
                continue
            stats = new_stats()
            for _ in range(FLAGS.steps_per_epoch):
                feed = feed_of(make_batch(B, N, C, seed=FLAGS.seed * 1000003 + step * world + rank, boxpc=True))
                feed[is_training_pl] = True
                out = sess.run(fetch_stats + [train_op], feed_dict=feed)
                record_batch(stats[0], stats[1], stats[2], fit_lo, out[0], out[1], out[2:5], g.inputs)
                loss_sum += float(out[0])
                step += 1
            mean_loss = loss_sum / FLAGS.steps_per_epoch
            log('**** EPOCH %03d ****  mean loss: %f  (%.1f frustums/s incl. host batch synthesis)' % (
                epoch, mean_loss, FLAGS.steps_per_epoch * B / (time.time() - t0)))
            report('train', stats)
            if FLAGS.eval_batches > 0 and rank == 0:
                eval_one_epoch(epoch)
            if epoch % 5 == 0 and rank == 0:
                sess.check_riders()      # never checkpoint weights a timed-out rider barrier may have corrupted
                path = saver.save(FLAGS.log_dir, epoch, g, FLAGS.ckpt_format)
                log('Model saved in file: %s' % path)
        sess.check_riders()
        final = g.vars.state_dict()
    api.finish_data_parallel(world)
    return final, mean_loss


if __name__ == '__main__':
    train(build_flags())
