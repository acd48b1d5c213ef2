#!/usr/bin/env python3
"""Inference + scoring driver with the reference's command line (sunrgbd/sunrgbd_detection/test_semisup.py: flags
527-544, get_model 61-180, inference 188-262, predictions layout 509-511).  The SUN-RGBD frustum pickles and the
box-IoU / MATLAB evaluation are out of scope: `--synthetic` frustums stand in for the dataset
and the predictions are written as the reference's 14-list (entries that need the dataset are None).

  python -m transferable3d_amd.test_semisup --semi_type F --model semisup_v1_sunrgbd --model_path log_adv/model_epoch_0.npz \
      --pred_prefix F2_ --refine 2 --use_one_hot --num_point 1024 --num_channels 4 --output preds.pickle --test AB --synthetic
"""
import gzip
import os
import pickle
import sys

import numpy as np

if __package__ in (None, ''):
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from transferable3d_amd import api, tf_util, semisup_v1_sunrgbd as MODEL        # noqa: E402
from transferable3d_amd.config import make_parser                        # noqa: E402
from transferable3d_amd.constants import NUM_HEADING_BIN, NUM_SIZE_CLUSTER   # noqa: E402
from transferable3d_amd.synthetic import make_batch                      # noqa: E402
from transferable3d_amd.tf_checkpoint import load_state                  # noqa: E402


def build_flags(argv=None):
    cfg = make_parser()
    cfg.add_argument('--test', type=str, nargs='+', default=['AB'])
    cfg.add_argument('--semi_type', type=str, choices=['A', 'F'], default='F')
    cfg.add_argument('--gpu', type=int, default=0)
    cfg.add_argument('--num_point', type=int, default=2048)
    cfg.add_argument('--model', default='semisup_v1_sunrgbd')
    cfg.add_argument('--model_path', default=None, help='state dict (.npz) or TensorFlow checkpoint prefix written by train_semisup*.py')
    cfg.add_argument('--boxpc_model_path', default=None)
    cfg.add_argument('--evaluate', action='store_true', help='AP of the predictions against the label boxes (evaluate.py)')
    cfg.add_argument('--pred_prefix', default='F2_')
    cfg.add_argument('--refine', default=None)
    cfg.add_argument('--output', default=None)
    cfg.add_argument('--no_rgb', action='store_true')
    cfg.add_argument('--mask_pc_for_boxpc', action='store_true', help='Mask the PC before giving to BoxPC network.')
    cfg.add_argument('--use_oracle_mask', action='store_true', help='ground-truth segmentation instead of the seg net\'s (use_oracle_mask of the reference\'s get_model / test)')
    cfg.add_argument('--use_boxpc_fit_prob', action='store_true')
    cfg.add_argument('--use_one_hot', action='store_true')
    cfg.add_argument('--batch_size', type=int, default=32)
    cfg.add_argument('--synthetic', action='store_true')
    cfg.add_argument('--from_rgb_detection', action='store_true', help='--data_path holds frustums of 2-D detections (7 lists, no 3-D labels)')
    cfg.add_argument('--result_dir', default=None, help='write <class>_pred.txt files for the MATLAB evaluation')
    cfg.add_argument('--gt_path', default=None, help='frustum file with the 3-D labels to evaluate detections against (evaluate.py --gt_path)')
    cfg.add_argument('--data_path', default=None, help='frustum file of the reference (frustums/*.zip.pickle); classes: --SUNRGBD_SEMI_TEST_CLS')
    cfg.add_argument('--num_channels', type=int, default=None)
    cfg.add_argument('--num_frustums', type=int, default=64)
    cfg.add_argument('--seed', type=int, default=0)
    cfg.add_argument('--dtype', default='f32', choices=['f32', 'bf16'],
                     help='element type of the per-point layer tensors and GEMM operands (bf16: BASELINE configs[4]; weights, statistics, heads, losses and Adam stay fp32)')
    FLAGS = cfg.parse_special_args(argv)
    FLAGS.NUM_CHANNELS = FLAGS.num_channels if FLAGS.num_channels else (3 if FLAGS.no_rgb else 6)
    FLAGS.SEMI_MODEL = FLAGS.semi_type
    return FLAGS


def get_model(FLAGS, batch_size, num_point, num_channel, rt=None, state_dict=None, use_oracle_mask=False):
    """test_semisup.py:61-180: the inference graph; returns (sess, ops).  `use_oracle_mask` (test_semisup.py:61,75): the ground-truth
    segmentation replaces the seg net's logits (SEMI_MODEL F only, as in the reference).  FLAGS.mask_pc_for_boxpc
    (test_semisup.py:103-105): the Box-PC net of the refinement loop sees pc * mask."""
    FLAGS.SEMI_REFINE_USING_BOXPC_DELTA_NUM = int(FLAGS.refine) if FLAGS.refine is not None else 0
    FLAGS.SEMI_WEIGH_BOXPC_DELTA_DURING_TEST = False                       # test_semisup.py:93
    FLAGS.BOX_PC_MASK_REPRESENTATION = 'A'
    graph = api.Graph(rt=rt, seed=FLAGS.seed, dtype=FLAGS.dtype)
    with graph.as_default():
        pls = MODEL.placeholder_inputs(batch_size, num_point, num_channel)
        norm_box2D = tf_util.tf_normalize_2D_bboxes(pls[15], pls[16])       # test_semisup.py:74
        pred, end_points = MODEL.get_semi_model(pls[0], pls[1], pls[2], pls[3], False, use_one_hot=FLAGS.use_one_hot,
                                                oracle_mask=pls[4] if use_oracle_mask else None, norm_box2D=norm_box2D, c=FLAGS)
        sess = api.Session()
    if state_dict is not None:
        graph.vars.load_state_dict(state_dict, strict=False)
    ops = {'pc_pl': pls[0], 'one_hot_vec_pl': pls[3], 'y_seg_pl': pls[4], 'logits': pred[0], 'end_points': end_points, 'graph': graph}
    return sess, ops


def softmax(x):
    probs = np.exp(x - np.max(x, axis=len(x.shape) - 1, keepdims=True))
    probs /= np.sum(probs, axis=len(x.shape) - 1, keepdims=True)
    return probs


def detection_scores(logits, heading_logits, size_logits, fit_prob=None):
    """test_semisup.py:238-252: log(mean mask probability + .01) + log(max heading prob + .01) + log(max size prob + .01)
    [+ log(fit prob + .01)] per frustum."""
    seg_prob = softmax(logits)[:, :, 1]
    seg_mask = np.argmax(logits, 2)
    mask_mean_prob = np.sum(seg_prob * seg_mask, 1) / (np.sum(seg_mask, 1) + 1)
    s = np.log(mask_mean_prob + 0.01) + np.log(np.max(softmax(heading_logits), 1) + 0.01) + np.log(np.max(softmax(size_logits), 1) + 0.01)
    return s if fit_prob is None else s + np.log(fit_prob + 0.01)


def inference(sess, ops, pc, one_hot_vec, batch_size, prefix='', use_boxpc_fit_prob=False, source=None, n_batches=None, oracle_mask=None):
    """test_semisup.py:188-262, same return tuple: (pred_seg, centers, orient_cls, orient_reg, dims_cls, dims_reg, scores).
    `source` (dataset.DeviceEvalSource): the batches are assembled on the device from a frustum file instead of being fed.
    `oracle_mask` [n, N] (test_semisup.py:207-208): fed to y_seg_pl of a graph built with use_oracle_mask."""
    if source is not None:
        n, npts = n_batches * batch_size, sess.g.engine.rpf
    else:
        assert pc.shape[0] % batch_size == 0
        n, npts = pc.shape[0], pc.shape[1]
    logits = np.zeros((n, npts, 2))
    centers = np.zeros((n, 3))
    heading_logits, heading_residuals = np.zeros((n, NUM_HEADING_BIN)), np.zeros((n, NUM_HEADING_BIN))
    size_logits, size_residuals = np.zeros((n, NUM_SIZE_CLUSTER)), np.zeros((n, NUM_SIZE_CLUSTER, 3))
    scores = np.zeros((n,))
    ep = ops['end_points']
    for i in range(n // batch_size):
        sl = slice(i * batch_size, (i + 1) * batch_size)
        run_ops = [ops['logits'], ep[prefix + 'center'], ep[prefix + 'heading_scores'], ep[prefix + 'heading_residuals'],
                   ep[prefix + 'size_scores'], ep[prefix + 'size_residuals']]
        if use_boxpc_fit_prob:
            run_ops.append(ep['boxpc_fit_prob'])
        if source is not None:
            source.load(i)
            out = sess.run(run_ops)
        else:
            feed = {ops['pc_pl']: pc[sl], ops['one_hot_vec_pl']: one_hot_vec[sl]}
            if oracle_mask is not None:
                feed[ops['y_seg_pl']] = np.asarray(oracle_mask[sl], np.int32)
            out = sess.run(run_ops, feed_dict=feed)
        logits[sl], centers[sl], heading_logits[sl], heading_residuals[sl], size_logits[sl], size_residuals[sl] = out[:6]
        scores[sl] = detection_scores(out[0], out[2], out[4], out[6] if use_boxpc_fit_prob else None)
    heading_cls, size_cls = np.argmax(heading_logits, 1), np.argmax(size_logits, 1)
    pred_orient_reg = heading_residuals[np.arange(n), heading_cls]
    pred_dims_reg = size_residuals[np.arange(n), size_cls, :]
    return np.argmax(logits, 2), centers, heading_cls, pred_orient_reg, size_cls, pred_dims_reg, scores


def test(FLAGS, rt=None, log=print):
    B, N, C = FLAGS.batch_size, FLAGS.num_point, FLAGS.NUM_CHANNELS
    sd = load_state(FLAGS.model_path) if FLAGS.model_path else None
    if FLAGS.boxpc_model_path:
        sd = dict(sd or {})
        sd.update({'D_boxpc_branch/' + k: v for k, v in load_state(FLAGS.boxpc_model_path).items()})
    use_oracle = bool(getattr(FLAGS, 'use_oracle_mask', False))
    sess, ops = get_model(FLAGS, B, N, C, rt=rt, state_dict=sd, use_oracle_mask=use_oracle)
    if FLAGS.data_path:
        return test_on_frustum_file(FLAGS, sess, ops, log)
    n = (FLAGS.num_frustums + B - 1) // B * B                     # the reference pads the last batch (test_semisup.py:450-471)
    batches = [make_batch(B, N, C, seed=FLAGS.seed * 1000003 + i) for i in range(n // B)]
    pc = np.concatenate([b['pc'] for b in batches])
    oh = np.concatenate([b['one_hot_vec'] for b in batches])
    seg_gt = np.concatenate([b['y_seg'] for b in batches])
    seg, centers, hcls, hres, scls, sres, scores = inference(sess, ops, pc, oh, B, prefix=FLAGS.pred_prefix,
                                                             use_boxpc_fit_prob=FLAGS.use_boxpc_fit_prob,
                                                             oracle_mask=seg_gt if use_oracle else None)
    iou = np.mean([(np.logical_and(seg[i], seg_gt[i]).sum() + 1e-9) / (np.logical_or(seg[i], seg_gt[i]).sum() + 1e-9) for i in range(n)])
    log('Mean segmentation IOU: %f' % iou)
    # test_semisup.py:509-511: [ps, seg_gt, seg_pred, center, heading_cls, heading_res, size_cls, size_res, rot_angle, score, cls,
    #                           file_num, box2d, box3d]
    # synthetic frustums are already in their centre view (rot_angle 0) and each is its own "image" (file_num = index)
    predictions = [list(pc), list(seg_gt), list(seg), list(centers), list(hcls), list(hres), list(scls), list(sres), [0.0] * n,
                   list(scores), list(np.argmax(oh, 1)), list(range(n)), None, None]
    if FLAGS.evaluate:
        # evaluate.py:27-76 against the label boxes of the same frustums (the reference builds them from the SUN-RGBD label files)
        from transferable3d_amd.constants import MEAN_DIMS_ARR, class2type
        from transferable3d_amd.eval_det import evaluate_predictions, get_3d_box, get_ap_info
        classes = [class2type[i] for i in range(10)]
        lab = {k: np.concatenate([b[k] for b in batches]) for k in ('y_center', 'y_orient_cls', 'y_orient_reg', 'y_dims_cls', 'y_dims_reg')}
        gt_all = {i: [(classes[int(np.argmax(oh[i]))],
                       get_3d_box(MEAN_DIMS_ARR[int(lab['y_dims_cls'][i])] + lab['y_dims_reg'][i],
                                  int(lab['y_orient_cls'][i]) * (2 * np.pi / NUM_HEADING_BIN) + float(lab['y_orient_reg'][i]), lab['y_center'][i]))]
                  for i in range(n)}
        _, _, ap, mean_ap = evaluate_predictions(predictions, gt_all, classes, rt=sess.g.rt)
        log(get_ap_info(ap, mean_ap))
    if FLAGS.output:
        with gzip.open(FLAGS.output, 'wb') as f:
            pickle.dump(predictions, f, -1)
        log('predictions written to %s' % FLAGS.output)
    return predictions


def from_prediction_to_label_format(center, angle_class, angle_res, size_class, size_res, rot_angle):
    """roi_seg_box3d_dataset.py:461-466: (h, w, l, tx, ty, tz, ry) in the camera frame, ty at the box bottom."""
    from transferable3d_amd.constants import MEAN_DIMS_ARR
    l, w, h = MEAN_DIMS_ARR[int(size_class)] + np.asarray(size_res, np.float64)
    ry = int(angle_class) * (2 * np.pi / NUM_HEADING_BIN) + float(angle_res)
    ry = (ry - 2 * np.pi if ry > np.pi else ry) + rot_angle
    c, s = np.cos(-rot_angle), np.sin(-rot_angle)                      # rotate_pc_along_y(center, -rot_angle)
    cx, cy, cz = [float(v) for v in np.asarray(center).reshape(3)]
    return h, w, l, c * cx - s * cz, cy + h / 2.0, s * cx + c * cz, ry


def write_detection_results(result_dir, test_classes, predictions, class_names):
    """test_semisup.py:262-296: one `<class>_pred.txt` per test class for the MATLAB evaluation
    (`evaluation/sunrgbd/detection/script_3Deval.m`): `idx cls -1 -1 -10 box2d(4) h w l tx ty tz ry score` per detection."""
    os.makedirs(result_dir, exist_ok=True)
    files = {c: open(os.path.join(result_dir, c + '_pred.txt'), 'w') for c in test_classes}
    _, _, _, center_l, hcls_l, hres_l, scls_l, sres_l, rot_l, score_l, _, id_l, box2d_l, _ = predictions
    for i in range(len(center_l)):
        box2d = box2d_l[i] if box2d_l is not None else (0.0, 0.0, 0.0, 0.0)
        vals = from_prediction_to_label_format(center_l[i], hcls_l[i], hres_l[i], scls_l[i], sres_l[i], float(rot_l[i]))
        files[class_names[i]].write('%d %s -1 -1 -10 %f %f %f %f %f %f %f %f %f %f %f %f\n' % (
            (int(id_l[i]), class_names[i], box2d[0], box2d[1], box2d[2], box2d[3]) + vals + (float(score_l[i]),)))
    for f in files.values():
        f.close()


def test_on_frustum_file(FLAGS, sess, ops, log):
    """main_batch (test_semisup.py:404-511) over a frustum file of the reference: the file's frustums of the test classes live in HBM,
    every batch is resampled to N points and rotated to its centre view on the device (no augmentation), the last batch is padded
    by wrapping around; predictions in the 14-list layout with the real image ids and rotation angles; --evaluate scores them
    against the file's own label boxes (evaluate.py builds the same boxes from the SUN-RGBD label files)."""
    from transferable3d_amd.constants import class2type
    from transferable3d_amd.dataset import DeviceEvalSource, DeviceFrustumSet
    from transferable3d_amd.eval_det import evaluate_predictions, get_ap_info
    g = ops['graph']
    B = FLAGS.batch_size
    test_classes = list(FLAGS.SUNRGBD_SEMI_TEST_CLS) or None
    if FLAGS.from_rgb_detection:                # main_batch_from_rgb_detection (test_semisup.py:337-423)
        ds = DeviceFrustumSet.from_detection_pickle(g.rt, FLAGS.data_path, classes=test_classes)
    else:
        ds = DeviceFrustumSet.from_pickle(g.rt, FLAGS.data_path, classes=test_classes)
    source = DeviceEvalSource(g, dataset=ds, seed=FLAGS.seed)
    n_batches = (ds.F + B - 1) // B
    seg, centers, hcls, hres, scls, sres, scores = inference(sess, ops, None, None, B, prefix=FLAGS.pred_prefix,
                                                             use_boxpc_fit_prob=FLAGS.use_boxpc_fit_prob, source=source, n_batches=n_batches)
    keep = slice(0, ds.F)                                          # drop the padding of the last batch
    cls = ds.cls.cpu().numpy()
    rot = np.pi / 2.0 + ds.frustum_angle.cpu().numpy().astype(np.float64)
    if FLAGS.from_rgb_detection:                # the score is the 2-D detection's (test_semisup.py:404-406); no labels in the file
        predictions = [None, None, list(seg[keep]), list(centers[keep]), list(hcls[keep]), list(hres[keep]), list(scls[keep]),
                       list(sres[keep]), list(rot), list(ds.prob), list(cls), list(ds.image_ids), list(ds.box2d), None]
    else:
        predictions = [None, None, list(seg[keep]), list(centers[keep]), list(hcls[keep]), list(hres[keep]), list(scls[keep]),
                       list(sres[keep]), list(rot), list(scores[keep]), list(cls), list(ds.image_ids), None, list(ds.box3d)]
    log('%d frustums of %s from %s' % (ds.F, sorted(set(ds.class_names)), FLAGS.data_path))
    if FLAGS.result_dir:
        write_detection_results(FLAGS.result_dir, test_classes or sorted(set(ds.class_names)), predictions, ds.class_names)
        log('detection results written to %s' % FLAGS.result_dir)
    if FLAGS.evaluate:
        classes = [class2type[i] for i in range(10)]
        gt_all = {}
        gt_ds = ds
        if FLAGS.gt_path:                       # evaluate.py --gt_path: the labelled frustum file of the same images
            from transferable3d_amd.dataset import load_zipped_pickle

            class _GT:
                pass
            lst = load_zipped_pickle(FLAGS.gt_path)
            gt_ds = _GT()
            sel = [i for i, t in enumerate(lst[6]) if test_classes is None or t in test_classes]
            gt_ds.image_ids, gt_ds.class_names = [lst[0][i] for i in sel], [lst[6][i] for i in sel]
            gt_ds.box3d = [np.asarray(lst[2][i], np.float64) for i in sel]
        elif FLAGS.from_rgb_detection:
            raise ValueError('--evaluate on detections needs --gt_path (the detection file holds no 3-D labels)')
        for img, name, k in zip(gt_ds.image_ids, gt_ds.class_names, gt_ds.box3d):
            k = k if k[0, 1] >= k[4, 1] else k[[4, 5, 6, 7, 0, 1, 2, 3]]      # y-max face first, as evaluate.py:49-52 arranges it
            gt_all.setdefault(img, []).append((name, k))
        _, _, ap, mean_ap = evaluate_predictions(predictions, gt_all, classes, rt=g.rt)
        log(get_ap_info(ap, mean_ap))
    if FLAGS.output:
        with gzip.open(FLAGS.output, 'wb') as f:
            pickle.dump(predictions, f, -1)
        log('predictions written to %s' % FLAGS.output)
    return predictions


if __name__ == '__main__':
    test(build_flags())
