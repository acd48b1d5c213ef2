"""CPU: detection evaluation (eval_det.py / evaluate.py surface) on the specification library against the oracle's loop."""
import numpy as np

from eval_check import check_eval_det, check_predictions_round_trip
from fake_t3d import FakeLib
from oracle import ref_eval as RE
from transferable3d_amd import eval_det as E
from transferable3d_amd.engine import Runtime


def test_voc_ap_known_answers_and_oracle():
    # one detection, correct: AP = 1; precision 1 up to recall 1/2 then nothing: AP = 1/2
    assert E.voc_ap(np.array([1.0]), np.array([1.0])) == 1.0
    assert E.voc_ap(np.array([0.5]), np.array([1.0])) == 0.5
    # the envelope: a dip in precision is filled by a later higher value
    rec, prec = np.array([0.25, 0.5, 0.5, 0.75, 1.0]), np.array([1.0, 1.0, 2 / 3, 0.75, 0.8])
    assert abs(E.voc_ap(rec, prec) - (0.5 * 1.0 + 0.5 * 0.8)) < 1e-12
    assert abs(E.voc_ap(rec, prec, True) - (6 * 1.0 + 5 * 0.8) / 11) < 1e-12
    r = np.random.RandomState(0)
    for _ in range(20):
        n = r.randint(1, 30)
        tp = np.cumsum(r.uniform(size=n) < 0.6)
        rec, prec = tp / max(tp[-1] + r.randint(0, 3), 1), tp / np.arange(1, n + 1)
        for m07 in (False, True):
            assert abs(E.voc_ap(rec, prec, m07) - RE.voc_ap(rec, prec, m07)) < 1e-12


def test_eval_det_against_the_reference_loop():
    check_eval_det(Runtime(device='cpu', lib=FakeLib()))


def test_predictions_to_boxes_and_perfect_detections():
    check_predictions_round_trip(Runtime(device='cpu', lib=FakeLib()))


def test_test_semisup_evaluate_flag_reports_average_precision():
    """test_semisup --evaluate: inference on synthetic frustums, predictions in the 14-list layout, AP table of evaluate.py."""
    from transferable3d_amd import test_semisup as TS
    logs = []
    FLAGS = TS.build_flags(['--semi_type', 'F', '--use_one_hot', '--num_point', '128', '--num_channels', '4', '--batch_size', '4',
                            '--num_frustums', '8', '--refine', '1', '--pred_prefix', 'F2_', '--test', 'AB', '--synthetic', '--evaluate'])
    preds = TS.test(FLAGS, rt=Runtime(device='cpu', lib=FakeLib()), log=logs.append)
    assert len(preds) == 14 and preds[8] == [0.0] * 8 and preds[11] == list(range(8))
    table = [l for l in logs if str(l).startswith('Average Precision:')]
    assert len(table) == 1 and 'Mean AP:' in table[0]


def _write_frustum_file(path, n=10, seed=4):
    """A file in the reference's 13-list format whose label corners come from get_3d_box in camera coordinates."""
    import gzip
    import pickle
    from oracle import ref_box as RB
    from transferable3d_amd.constants import MEAN_DIMS_ARR, class2type
    r = np.random.RandomState(seed)
    cls = r.randint(0, 10, n)
    size = MEAN_DIMS_ARR[cls] * r.uniform(0.8, 1.2, (n, 3))
    heading = r.uniform(-np.pi, np.pi, n)
    fang = r.uniform(-0.5, 0.5, n) - np.pi / 2
    depth = r.uniform(2, 5, n)
    center = np.stack([depth * np.cos(-fang - np.pi / 2 + np.pi / 2) * 0 + depth * np.sin(fang + np.pi / 2) * 0, np.zeros(n), depth], 1)
    center[:, 0] = r.normal(size=n) * 0.5
    box3d = [RB.get_3d_box(size[i], heading[i], center[i]) for i in range(n)]
    counts = r.randint(150, 400, n)
    lists = [list(range(100, 100 + n)), [np.zeros(4)] * n, box3d, [None] * n,
             [np.concatenate([center[i] + r.normal(size=(counts[i], 3)) * 0.4, r.uniform(size=(counts[i], 3))], 1) for i in range(n)],
             [(r.uniform(size=counts[i]) < 0.3).astype(np.float64) for i in range(n)], [class2type[int(c)].encode() for c in cls],
             list(heading), list(size), [np.eye(3)] * n, [np.eye(3)] * n, list(fang), [np.array([640, 480])] * n]
    with gzip.open(path, 'wb') as f:
        pickle.dump(lists, f, 2)
    return box3d, cls


def test_frustum_file_labels_round_trip_through_the_centre_view(tmp_path):
    """Frame conventions of the file path of test_semisup: the label box of every frustum, as the device assembly hands it to the
    net (centre view, bin + residual), written out as a 'prediction' and rotated back by evaluate.py's rule, is the file's own
    label box: IoU 1 with it, AP 1."""
    from transferable3d_amd.constants import class2type
    from transferable3d_amd.dataset import DeviceEvalSource, DeviceFrustumSet
    from transferable3d_amd.nets import Graph, Inputs
    path = str(tmp_path / 'val.zip.pickle')
    box3d, cls = _write_frustum_file(path, n=10)
    rt = Runtime(device='cpu', lib=FakeLib())
    B, N, C = 4, 128, 4

    class G:                                         # the two attributes DeviceEvalSource needs of an api.Graph
        pass
    g = G()
    g.rt, g.engine = rt, Graph(B, N, C, rt=rt)
    g.inputs = Inputs(g.engine)
    ds = DeviceFrustumSet.from_pickle(rt, path)
    src = DeviceEvalSource(g, dataset=ds)
    cen, hc, hr, sc, sr = [], [], [], [], []
    for i in range(3):                               # 10 frustums, B = 4: the last batch wraps around
        src.load(i)
        x = g.inputs
        cen.append(x.y_center.numpy().copy()); hc.append(x.y_orient_cls.numpy().copy()); hr.append(x.y_orient_reg.numpy().copy())
        sc.append(x.y_dims_cls.numpy().copy()); sr.append(x.y_dims_reg.numpy().copy())
    assert np.array_equal(src.frustums_of(2), [8, 9, 0, 1])
    cat = lambda v: np.concatenate(v)[:10]
    rot = np.pi / 2 + ds.frustum_angle.numpy().astype(np.float64)
    preds = [None, None, None, list(cat(cen)), list(cat(hc)), list(cat(hr)), list(cat(sc)), list(cat(sr)), list(rot), [1.0] * 10, list(cls),
             list(ds.image_ids), None, None]
    classes = [class2type[i] for i in range(10)]
    boxes = E.predictions_to_boxes(preds, classes)
    gt_all = {img: [(name, k)] for img, name, k in zip(ds.image_ids, ds.class_names, ds.box3d)}
    for img in ds.image_ids:
        assert E.get_iou(boxes[img][0][1], gt_all[img][0][1], rt) > 0.999
    _, _, ap, mean_ap = E.evaluate_predictions(preds, gt_all, classes, rt=rt)
    assert abs(mean_ap - 1.0) < 1e-9


def test_test_semisup_on_a_frustum_file(tmp_path):
    from transferable3d_amd import test_semisup as TS
    path = str(tmp_path / 'val.zip.pickle')
    _write_frustum_file(path, n=10)
    logs = []
    FLAGS = TS.build_flags(['--semi_type', 'F', '--use_one_hot', '--num_point', '128', '--num_channels', '4', '--batch_size', '4', '--refine', '1',
                            '--pred_prefix', 'F2_', '--test', 'AB', '--data_path', path, '--evaluate', '--SUNRGBD_SEMI_TEST_CLS', 'bed', 'table',
                            'sofa', 'chair', 'toilet', 'desk', 'dresser', 'night_stand', 'bookshelf', 'bathtub',
                            '--output', str(tmp_path / 'pred.zip.pickle')])
    preds = TS.test(FLAGS, rt=Runtime(device='cpu', lib=FakeLib()), log=logs.append)
    assert len(preds) == 14 and len(preds[3]) == 10 and preds[11] == list(range(100, 110)) and len(preds[13]) == 10
    assert any(str(l).startswith('Average Precision:') for l in logs)
    from transferable3d_amd.dataset import load_zipped_pickle
    back = load_zipped_pickle(str(tmp_path / 'pred.zip.pickle'))
    assert len(back) == 14 and np.allclose(back[8], preds[8])


def test_training_clis_read_frustum_files(tmp_path):
    """--frustum_file / --eval_file: the three drivers train from a frustum file of the reference held in HBM and evaluate on
    another one."""
    from transferable3d_amd import train_boxpc, train_semisup, train_semisup_adv
    tr, ev = str(tmp_path / 'train.zip.pickle'), str(tmp_path / 'val.zip.pickle')
    _write_frustum_file(tr, n=40, seed=1)
    _write_frustum_file(ev, n=9, seed=2)
    rt = lambda: Runtime(device='cpu', lib=FakeLib())
    small = ['--num_point', '128', '--batch_size', '4', '--num_channels', '4', '--max_epoch', '1', '--steps_per_epoch', '2', '--frustum_file', tr,
             '--eval_file', ev]
    logs = []
    train_semisup.train(train_semisup.build_flags(['--SEMI_MODEL', 'A', '--WEAK_WEIGHT_REPROJECTION', '0', '--WEAK_WEIGHT_SURFACE', '0',
                                                   '--log_dir', str(tmp_path / 'a')] + small), rt=rt(), log=logs.append)
    assert any('Mean AP' in str(l) for l in logs) and any('assembled on the device' in str(l) for l in logs)
    logs = []
    train_boxpc.train(train_boxpc.build_flags(['--BOX_PC_MASK_REPRESENTATION', 'A', '--BOXPC_WEIGHT_DELTA', '4', '--log_dir', str(tmp_path / 'b')]
                                              + small), rt=rt(), log=logs.append)
    assert any('eval mean loss' in str(l) for l in logs)
    logs = []
    train_semisup_adv.train(train_semisup_adv.build_flags(
        ['--SEMI_MODEL', 'F', '--BOX_PC_MASK_REPRESENTATION', 'A', '--use_one_hot', '--WEAK_WEIGHT_INTRACLASSVAR', '2', '--WEAK_WEIGHT_REPROJECTION', '0',
         '--SEMI_MULTIPLIER_FOR_WEAK_LOSS', '0.05', '--SEMI_SAMPLE_EQUAL_CLASS_WITH_PROB', '1', '--log_dir', str(tmp_path / 'c')] + small),
        rt=rt(), log=logs.append)
    assert any('refined by the Box-PC deltas' in str(l) for l in logs)


def test_test_semisup_from_rgb_detection(tmp_path):
    """--from_rgb_detection: a 7-list detection file (no 3-D labels); the prediction score is the detector's; --gt_path evaluates
    against the labelled file of the same images."""
    import gzip
    import pickle
    from transferable3d_amd import test_semisup as TS
    from transferable3d_amd.dataset import load_zipped_pickle
    gt = str(tmp_path / 'val.zip.pickle')
    _write_frustum_file(gt, n=8, seed=7)
    lab = load_zipped_pickle(gt)
    det = [lab[0], lab[1], [None] * 8, lab[4], [t.encode() for t in lab[6]], lab[11], list(np.linspace(0.2, 0.9, 8))]
    dpath = str(tmp_path / 'det.zip.pickle')
    with gzip.open(dpath, 'wb') as f:
        pickle.dump(det, f, 2)
    logs = []
    all_cls = ['bed', 'table', 'sofa', 'chair', 'toilet', 'desk', 'dresser', 'night_stand', 'bookshelf', 'bathtub']
    FLAGS = TS.build_flags(['--semi_type', 'F', '--use_one_hot', '--num_point', '128', '--num_channels', '4', '--batch_size', '4', '--refine', '1',
                            '--pred_prefix', 'F2_', '--test', 'B', '--data_path', dpath, '--from_rgb_detection', '--evaluate', '--gt_path', gt,
                            '--result_dir', str(tmp_path / 'res'), '--SUNRGBD_SEMI_TEST_CLS'] + all_cls)
    preds = TS.test(FLAGS, rt=Runtime(device='cpu', lib=FakeLib()), log=logs.append)
    assert np.allclose(preds[9], np.linspace(0.2, 0.9, 8)) and preds[11] == lab[0] and len(preds[12]) == 8 and preds[13] is None
    assert any(str(l).startswith('Average Precision:') for l in logs)
    # the MATLAB-evaluation text files: one line per detection, 17 fields, the box back in the camera frame with ty at its bottom
    lines = [l.split() for c in all_cls for l in open(tmp_path / 'res' / (c + '_pred.txt'))]
    assert len(lines) == 8 and all(len(l) == 17 and l[2:5] == ['-1', '-1', '-10'] for l in lines)
    i = [int(l[0]) for l in lines].index(lab[0][0])
    h, w, l_, tx, ty, tz, ry = TS.from_prediction_to_label_format(preds[3][0], preds[4][0], preds[5][0], preds[6][0], preds[7][0], preds[8][0])
    assert np.allclose([float(v) for v in lines[i][9:16]], [h, w, l_, tx, ty, tz, ry], atol=1e-5)
    from oracle import ref_data as RD
    back = RD.rotate_pc_along_y(np.asarray(preds[3][0], np.float64).reshape(1, 3), -preds[8][0]).squeeze()
    assert np.allclose([tx, ty - h / 2, tz], back, atol=1e-9)
