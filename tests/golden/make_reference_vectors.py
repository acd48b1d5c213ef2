"""Generates tests/golden/reference_vectors.npz + reference_config.json by EXECUTING the TensorFlow-free modules of the reference
itself (run in the build container only, where /root/reference exists: `python tests/golden/make_reference_vectors.py`).

What runs is the reference's own code, loaded from where it lies (nothing of it is copied here):
    models/config.py                                   the flag parser and its defaults
    sunrgbd/sunrgbd_detection/eval_det.py              voc_ap, eval_det_cls, eval_det
    sunrgbd/sunrgbd_detection/roi_seg_box3d_dataset.py angle2class, class2angle, size2class, class2size, rotate_pc_along_y,
                                                       get_3d_box, from_prediction_to_label_format, ROISegBoxDataset.__getitem__
    sunrgbd/sunrgbd_detection/roi_semi_dataset.py      ROISemiDataset.__init__ / get_classes3D / get_batch
    sunrgbd/sunrgbd_data/utils.py                      roty, load_zipped_pickle, save_zipped_pickle
    sunrgbd/sunrgbd_detection/train_boxpc.py           ONLY the two NumPy statistics classes ClassificationStats / BoxDeltaIOUStats:
                                                       the module imports TensorFlow, so the two `class` statements are taken out of
                                                       its syntax tree and executed on their own (get_batch_stats; their
                                                       summarize_stats sorts dict.keys() in place and is Python-2 only)
    sunrgbd/sunrgbd_detection/test_semisup.py          ONLY the functions softmax, inference and write_detection_results, taken out of
                                                       the syntax tree the same way.  `inference` gets a session object whose run()
                                                       hands back prepared arrays in place of network outputs -- what is exercised is
                                                       the reference's NumPy post-processing (detection score, arg-max decode), not a
                                                       network; its Python-2 `pc.shape[0]/batch_size` is kept an integer by passing
                                                       batch_size as an int whose reflected division floors

Three modules those files import do not exist in this image and are given placeholders so that the `import` lines pass:
    cv2       an empty module (nothing recorded here calls it);
    cPickle   the standard library's pickle (its Python-3 name);
    box_util  the module the reference imports from Frustum PointNets and does not ship.  Its box3d_iou is replaced by the closed-form
              IoU of AXIS-ALIGNED boxes written out below, and the detection fixtures use axis-aligned boxes only, so the recorded
              precision / recall / AP pin the reference's matching loop, not an IoU routine.  For the Box-PC sample generator
              (box_pc_fit_dataset.py BoxPCFitDataset.get / perturb_box_to_diff_ious, whose boxes are rotated) the placeholder
              forwards to the oracle's own restatement of box3d_iou (oracle/ref_box.py): those vectors pin the generator's law --
              which draws it takes, how it scales and applies them, the rejection test, the label format -- GIVEN that IoU; they
              say nothing about the IoU routine itself, which stays unpinned.
Everything that needs the TensorFlow graph (the networks, the losses) cannot run and stays unpinned (oracle/README.md).

The reference draws its augmentation from the global np.random stream; the draws are recorded next to the outputs, so that the
oracle (which takes the draws as arguments) and the device kernels can be checked on the same draws.
"""
import importlib.util
import json
import os
import pickle
import sys
import types

import numpy as np

sys.dont_write_bytecode = True       # the reference checkout is read-only for this project: importing from it must not leave __pycache__ there
REF = '/root/reference'
HERE = os.path.dirname(os.path.abspath(__file__))


def axis_aligned_iou(c1, c2):
    """(iou3d, iou2d) of two boxes given as (8,3) corners whose edges are parallel to the axes; ground plane = x-z, height = y."""
    lo1, hi1, lo2, hi2 = c1.min(0), c1.max(0), c2.min(0), c2.max(0)
    ov = np.maximum(0.0, np.minimum(hi1, hi2) - np.maximum(lo1, lo2))
    d1, d2 = hi1 - lo1, hi2 - lo2
    inter2, inter3 = ov[0] * ov[2], ov[0] * ov[1] * ov[2]
    iou2d = inter2 / (d1[0] * d1[2] + d2[0] * d2[2] - inter2)
    iou3d = inter3 / (d1.prod() + d2.prod() - inter3)
    return iou3d, iou2d


IOU_IMPL = [axis_aligned_iou]          # what the box_util placeholder forwards to (switched for the Box-PC generator, see header)


def load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


class Py2Int(int):
    """int / Py2Int floors, as `/` between two ints does in the Python 2 the reference was written for."""

    def __rtruediv__(self, other):
        return other // int(self)


def reference_classes(path, names, namespace):
    """Executes the named top-level `class` / `def` statements of a reference file without running the rest of the module."""
    import ast
    with open(path) as f:
        tree = ast.parse(f.read(), filename=path)
    body = [n for n in tree.body if isinstance(n, (ast.ClassDef, ast.FunctionDef)) and n.name in names]
    assert len(body) == len(names)
    exec(compile(ast.Module(body=body, type_ignores=[]), path, 'exec'), namespace)
    return [namespace[n] for n in names]


def reference_modules():
    sys.modules['cv2'] = types.ModuleType('cv2')
    sys.modules['cPickle'] = pickle
    bu = types.ModuleType('box_util')
    bu.box3d_iou = lambda c1, c2: IOU_IMPL[0](c1, c2)
    sys.modules['box_util'] = bu
    det = os.path.join(REF, 'sunrgbd', 'sunrgbd_detection')
    utils = load('utils', os.path.join(REF, 'sunrgbd', 'sunrgbd_data', 'utils.py'))
    seg = load('roi_seg_box3d_dataset', os.path.join(det, 'roi_seg_box3d_dataset.py'))
    semi = load('roi_semi_dataset', os.path.join(det, 'roi_semi_dataset.py'))
    ev = load('eval_det', os.path.join(det, 'eval_det.py'))
    bp = load('box_pc_fit_dataset', os.path.join(det, 'box_pc_fit_dataset.py'))
    return utils, seg, semi, ev, bp


class DrawRecorder:
    """Records what the reference takes from the global np.random stream, in order."""

    NAMES = ('choice', 'random', 'randn', 'rand', 'uniform')

    def __init__(self):
        self.log = []
        self.orig = {}

    def __enter__(self):
        for n in self.NAMES:
            self.orig[n] = getattr(np.random, n)
            setattr(np.random, n, self._wrap(n))
        return self

    def _wrap(self, n):
        def f(*a, **k):
            v = self.orig[n](*a, **k)
            self.log.append((n, np.array(v)))
            return v
        return f

    def __exit__(self, *exc):
        for n in self.NAMES:
            setattr(np.random, n, self.orig[n])

    def take(self, name):
        return [v for n, v in self.log if n == name]


CONFIG_LINES = {          # the README's three training recipes (flags only) + the defaults
    'defaults': [],
    'recipe_a': ['--SEMI_MODEL', 'A', '--WEAK_WEIGHT_REPROJECTION', '0', '--WEAK_WEIGHT_SURFACE', '0'],
    'recipe_b': ['--BOX_PC_MASK_REPRESENTATION', 'A', '--BOXPC_WEIGHT_DELTA', '4'],
    'recipe_c': ['--SEMI_MODEL', 'F', '--BOX_PC_MASK_REPRESENTATION', 'A', '--WEAK_WEIGHT_INTRACLASSVAR', '2',
                 '--WEAK_WEIGHT_REPROJECTION', '0', '--SEMI_MULTIPLIER_FOR_WEAK_LOSS', '0.05', '--SEMI_BOXPC_FIT_ONLY_ON_2D_CLS', '1',
                 '--SEMI_WEIGHT_BOXPC_FIT_LOSS', '1'],
    'lists_and_bools': ['--BOXPC_NOFIT_BOUNDS', '0.05', '0.3', '--BOXPC_FIT_BOUNDS', '0.6', '0.9', '--SEMI_USE_LABELS2D_OF_CLASSES3D', 'true',
                        '--SEMI_SAMPLE_EQUAL_CLASS_WITH_PROB', '0.5', '--SEMI_SAMPLING_METHOD', 'BATCH'],
}


def jsonable(v):
    if isinstance(v, (np.floating, np.integer)):
        return v.item()
    if isinstance(v, np.ndarray):
        return v.tolist()
    if isinstance(v, (list, tuple)):
        return [jsonable(x) for x in v]
    return v


def config_vectors():
    out = {}
    argv = sys.argv
    for name, line in CONFIG_LINES.items():
        sys.argv = ['prog'] + line
        for m in ('config',):
            sys.modules.pop(m, None)
        cfgmod = load('config', os.path.join(REF, 'models', 'config.py'))
        flags = cfgmod.cfg.parse_special_args()
        out[name] = {'argv': line, 'flags': {k: jsonable(v) for k, v in sorted(vars(flags).items()) if k != 'config_str'},
                     'config_str': flags.config_str}
    sys.argv = argv
    return out


def synthetic_frustums(seg_mod, n_frustums, seed):
    """A frustum file in the layout the reference's readers expect (13 parallel lists), with random content."""
    r = np.random.RandomState(seed)
    types_ = sorted(seg_mod.type2class)
    L = [[] for _ in range(13)]
    for i in range(n_frustums):
        cls = types_[r.randint(len(types_))]
        npts = int(r.randint(40, 200))
        pts = np.concatenate([r.normal(size=(npts, 3)) * [0.8, 0.5, 0.8] + [0.3, 0.1, 3.0], r.uniform(size=(npts, 3))], 1)
        size = seg_mod.type_mean_size[cls] * r.uniform(0.8, 1.2, size=3)
        heading = float(r.uniform(-np.pi, np.pi))
        center = np.array([0.3, 0.1, 3.0]) + r.normal(size=3) * 0.2
        box3d = seg_mod.get_3d_box(size, heading, center)
        items = (i, r.uniform(0, 300, size=4), box3d, None, pts, (r.uniform(size=npts) < 0.4).astype(np.float64), cls, heading, size,
                 np.eye(3) + r.normal(size=(3, 3)) * 0.01, np.array([[500.0, 0, 320], [0, 500.0, 240], [0, 0, 1]]),
                 float(r.uniform(-np.pi, 0)), np.array([480.0, 640.0]))
        for lst, it in zip(L, items):
            lst.append(it)
    return L


def main():
    utils, seg, semi, ev, bp = reference_modules()
    out = {}
    r = np.random.RandomState(7)

    # ---- constants ----------------------------------------------------------------------------------------------------------
    types_ = [seg.class2type[i] for i in range(seg.NUM_CLASS)]
    out['const/type_names'] = np.array(types_)
    out['const/mean_size'] = np.stack([seg.type_mean_size[t] for t in types_])
    out['const/num_heading_bin'], out['const/num_size_cluster'], out['const/num_class'] = seg.NUM_HEADING_BIN, seg.NUM_SIZE_CLUSTER, seg.NUM_CLASS

    # the anchor table the TensorFlow graph itself uses (models/model_util.py imports TensorFlow: its constant assignments are taken
    # out of the syntax tree -- every top-level assignment / loop over names starting with SUN_ or sun_)
    import ast
    mu = os.path.join(REF, 'models', 'model_util.py')
    with open(mu) as f:
        src = f.read()
    tree = ast.parse(src, filename=mu)
    body = [n for n in tree.body if isinstance(n, (ast.Assign, ast.For)) and
            any(isinstance(x, ast.Name) and x.id.startswith(('SUN_', 'sun_')) for x in ast.walk(n))]
    ns_mu = {'np': np}
    exec(compile(ast.Module(body=body, type_ignores=[]), mu, 'exec'), ns_mu)
    out['const/graph_mean_size_arr'] = ns_mu['sun_mean_size_arr']
    out['const/graph_num_heading_bin'], out['const/graph_num_size_cluster'] = ns_mu['SUN_NUM_HEADING_BIN'], ns_mu['SUN_NUM_SIZE_CLUSTER']
    assert [ns_mu['sun_class2type'][i] for i in range(10)] == types_

    # ---- angle / size / box helpers -----------------------------------------------------------------------------------------
    ang = np.concatenate([r.uniform(-2 * np.pi, 4 * np.pi, size=40), [0.0, np.pi, -np.pi, 2 * np.pi - 1e-9, np.pi / 12, -np.pi / 12]])
    a2c = np.array([seg.angle2class(a, seg.NUM_HEADING_BIN) for a in ang])
    out['angle/in'], out['angle/cls'], out['angle/res'] = ang, a2c[:, 0].astype(np.int64), a2c[:, 1]
    cls = r.randint(0, seg.NUM_HEADING_BIN, size=40)
    res = r.uniform(-np.pi / 12, np.pi / 12, size=40)
    out['class2angle/cls'], out['class2angle/res'] = cls, res
    out['class2angle/label_format'] = np.array([seg.class2angle(c, x, seg.NUM_HEADING_BIN) for c, x in zip(cls, res)])
    out['class2angle/raw'] = np.array([seg.class2angle(c, x, seg.NUM_HEADING_BIN, to_label_format=False) for c, x in zip(cls, res)])
    sizes = r.uniform(0.3, 2.5, size=(20, 3))
    scls = r.randint(0, seg.NUM_CLASS, size=20)
    s2c = [seg.size2class(s, types_[c]) for s, c in zip(sizes, scls)]
    out['size/in'], out['size/type'] = sizes, scls
    out['size/cls'], out['size/res'] = np.array([x[0] for x in s2c]), np.stack([x[1] for x in s2c])
    out['size/back'] = np.stack([seg.class2size(x[0], x[1]) for x in s2c])
    pc = r.normal(size=(16, 6))
    rots = r.uniform(-np.pi, np.pi, size=5)
    out['rotate/pc'], out['rotate/angle'] = pc, rots
    out['rotate/out'] = np.stack([seg.rotate_pc_along_y(pc.copy(), a) for a in rots])
    heads, cents = r.uniform(-np.pi, np.pi, size=8), r.normal(size=(8, 3))
    out['box/size'], out['box/heading'], out['box/center'] = sizes[:8], heads, cents.copy()
    out['box/corners'] = np.stack([seg.get_3d_box(s, h, c) for s, h, c in zip(sizes[:8], heads, cents)])
    out['box/roty'] = np.stack([utils.roty(h) for h in heads])
    sres = r.normal(size=(8, 3)) * 0.1
    out['p2l/center'], out['p2l/angle_cls'], out['p2l/angle_res'], out['p2l/size_cls'], out['p2l/size_res'], out['p2l/rot'] = \
        cents.copy(), cls[:8], res[:8], scls[:8], sres, rots[[0, 1, 2, 3, 4, 0, 1, 2]]
    # (the reference rotates `center` in place through the view np.expand_dims returns: hand it a copy)
    out['p2l/out'] = np.array([seg.from_prediction_to_label_format(cents[i].copy(), cls[i], res[i], scls[i], sres[i], out['p2l/rot'][i])
                               for i in range(8)])

    # ---- voc_ap --------------------------------------------------------------------------------------------------------------
    for i in range(6):
        n = int(r.randint(1, 30))
        tp = (r.uniform(size=n) < 0.6).astype(np.float64)
        tpc, fpc = np.cumsum(tp), np.cumsum(1 - tp)
        rec, prec = tpc / max(tpc[-1] + r.randint(0, 4), 1.0), tpc / np.maximum(tpc + fpc, np.finfo(np.float64).eps)
        out['voc_ap/%d/rec' % i], out['voc_ap/%d/prec' % i] = rec, prec
        out['voc_ap/%d/ap' % i] = ev.voc_ap(rec, prec)
        out['voc_ap/%d/ap07' % i] = ev.voc_ap(rec, prec, use_07_metric=True)

    # ---- eval_det_cls / eval_det on axis-aligned boxes --------------------------------------------------------------------------
    def aabox():
        return seg.get_3d_box(r.uniform(0.5, 1.5, size=3), 0.0, r.uniform(-1.5, 1.5, size=3))

    names = ['bed', 'chair', 'table', 'sofa']
    gt_all, pred_all = {}, {}
    flat_gt, flat_pred = [], []
    for img in range(12):
        gts = [(names[r.randint(3)], aabox()) for _ in range(r.randint(0, 4))]          # 'sofa' never has ground truth
        preds = []
        for cn, b in gts:
            if r.uniform() < 0.8:                                                        # a detection near the object
                jit = b + r.normal(size=3) * 0.15
                preds.append((cn if r.uniform() < 0.85 else names[r.randint(4)], jit, float(r.uniform())))
            if r.uniform() < 0.3:                                                        # a duplicate detection of the same object
                preds.append((cn, b + r.normal(size=3) * 0.05, float(r.uniform())))
        for _ in range(r.randint(0, 3)):                                                 # clutter
            preds.append((names[r.randint(4)], aabox(), float(r.uniform())))
        if gts:
            gt_all[img] = gts
        if preds:
            pred_all[img] = preds
        flat_gt += [(img, names.index(cn), b) for cn, b in gts]
        flat_pred += [(img, names.index(cn), b, s) for cn, b, s in preds]
    out['det/names'] = np.array(names)
    out['det/gt_img'], out['det/gt_cls'] = np.array([g[0] for g in flat_gt]), np.array([g[1] for g in flat_gt])
    out['det/gt_box'] = np.stack([g[2] for g in flat_gt])
    out['det/pred_img'], out['det/pred_cls'] = np.array([p[0] for p in flat_pred]), np.array([p[1] for p in flat_pred])
    out['det/pred_box'], out['det/pred_score'] = np.stack([p[2] for p in flat_pred]), np.array([p[3] for p in flat_pred])
    for tag, thr, m07 in (('t25', 0.25, False), ('t50', 0.5, False), ('t25_07', 0.25, True)):
        rec, prec, ap = ev.eval_det(pred_all, gt_all, ovthresh=thr, use_07_metric=m07)
        out['det/%s/classes' % tag] = np.array(sorted(ap))
        for cn in ap:
            out['det/%s/%s/rec' % (tag, cn)], out['det/%s/%s/prec' % (tag, cn)], out['det/%s/%s/ap' % (tag, cn)] = rec[cn], prec[cn], ap[cn]
    rec, prec, ap = ev.eval_det(pred_all, gt_all, ovthresh={'bed': 0.25, 'chair': 0.5, 'table': 0.1, 'sofa': 0.25}, use_07_metric=False)
    for cn in ap:
        out['det/per_class_thresh/%s/ap' % cn] = ap[cn]

    # ---- the frustum file readers and the per-sample assembly -------------------------------------------------------------------
    L = synthetic_frustums(seg, 24, seed=3)
    path = os.path.join(HERE, 'reference_frustums.zip.pickle')
    utils.save_zipped_pickle(L, path, protocol=2)    # the reference's own writer; protocol 2 = what its Python 2 `-1` means
    back = utils.load_zipped_pickle(path)
    assert len(back) == 13 and len(back[0]) == 24
    # and the reference's reader takes what the product's writer produces
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    from transferable3d_amd.dataset import save_zipped_pickle as product_save
    tmp = os.path.join(HERE, '_roundtrip.zip.pickle')
    product_save(L, tmp)
    again = utils.load_zipped_pickle(tmp)
    os.remove(tmp)
    assert len(again) == 13 and all(np.array_equal(a, b) for a, b in zip(again[4], L[4])) and list(again[6]) == list(L[6])
    N = 128          # the device graph takes multiples of 128 points
    classes = sorted(seg.type2class)
    ds = seg.ROISegBoxDataset(classes, N, 'train', random_flip=True, random_shift=True, rotate_to_center=True, overwritten_data_path=path,
                              one_hot=True)
    np.random.seed(11)
    for i in range(len(ds)):
        with DrawRecorder() as rec_:
            item = ds[i]
        p = 'getitem/%d/' % i
        out[p + 'choice'], out[p + 'flip_u'] = rec_.take('choice')[0], rec_.take('random')[0]
        out[p + 'shift_randn'], out[p + 'height_u'] = rec_.take('randn')[0], rec_.take('random')[1]
        for k, v in zip(('point_set', None, 'seg', 'center', 'angle_cls', 'angle_res', 'size_cls', 'size_res', 'box2d', 'rtilt', 'k',
                         'rot_angle', 'img_dims', 'one_hot'), item):
            if k is not None:
                out[p + k] = np.asarray(v)
    out['getitem/count'], out['getitem/npoints'] = len(ds), N
    ds2 = seg.ROISegBoxDataset(classes, N, 'val', rotate_to_center=True, overwritten_data_path=path, one_hot=True)    # no augmentation
    np.random.seed(12)
    with DrawRecorder() as rec_:
        b = ds2.get_batch(list(range(len(ds2))), 4, 12, N, 6)
    out['get_batch/choice'] = np.stack(rec_.take('choice'))
    for k, v in zip(('pc', None, 'seg', 'center', 'angle_cls', 'angle_res', 'size_cls', 'size_res', 'box2d', 'rtilt', 'k', 'rot_angle',
                     'img_dims', 'one_hot'), b):
        if k is not None:
            out['get_batch/' + k] = np.asarray(v)

    # the semi-supervised data set: which frustums land in the 3-D and the 2-D lists, and one assembled batch
    c3, c2 = ['bed', 'chair', 'table', 'sofa', 'toilet'], ['desk', 'dresser', 'night_stand', 'bookshelf', 'bathtub']
    sd = semi.ROISemiDataset(c3, c2, N, random_flip=True, random_shift=True, rotate_to_center=True, overwritten_data_path=path)
    out['semi/classes3D'], out['semi/classes2D'] = np.array(c3), np.array(c2)
    out['semi/idx_3D'], out['semi/idx_2D'] = np.array(sd.idx_3Dl), np.array(sd.idx_2Dl)
    for cn, ids in sd.cls_to_idx_map3D.items():
        out['semi/map3D/' + cn] = np.array(ids)
    for cn, ids in sd.cls_to_idx_map2D.items():
        out['semi/map2D/' + cn] = np.array(ids)
    np.random.seed(13)
    n3 = sd.get_len_classes3D()
    with DrawRecorder() as rec_:
        items = [sd.get_classes3D(i) for i in range(n3)]
    out['semi/get3D/choice'] = np.stack(rec_.take('choice'))
    rnd = rec_.take('random')
    out['semi/get3D/flip_u'], out['semi/get3D/height_u'] = np.array(rnd[0::2]), np.array(rnd[1::2])
    out['semi/get3D/shift_randn'] = np.array(rec_.take('randn'))
    for j, k in ((0, 'point_set'), (2, 'seg'), (3, 'center'), (4, 'angle_cls'), (5, 'angle_res'), (6, 'size_cls'), (7, 'size_res'),
                 (11, 'rot_angle'), (13, 'one_hot')):
        out['semi/get3D/' + k] = np.stack([np.asarray(it[j]) for it in items])

    # the 2-D-label list: no augmentation, zero labels
    np.random.seed(14)
    n2 = sd.get_len_classes2D()
    with DrawRecorder() as rec_:
        items = [sd.get_classes2D(i) for i in range(n2)]
    assert not rec_.take('random') and not rec_.take('randn')
    out['semi/get2D/choice'] = np.stack(rec_.take('choice'))
    for j, k in ((0, 'point_set'), (2, 'seg'), (3, 'center'), (4, 'angle_cls'), (5, 'angle_res'), (6, 'size_cls'), (7, 'size_res'),
                 (11, 'rot_angle'), (13, 'one_hot')):
        out['semi/get2D/' + k] = np.stack([np.asarray(it[j]) for it in items])
    # get_batch over the combined index space (3-D list first, 2-D list behind it), SEMI_SAMPLING_METHOD BATCH
    np.random.seed(15)
    idxs = np.random.RandomState(5).permutation(n3 + n2)
    with DrawRecorder() as rec_:
        b = sd.get_batch(idxs, 2, 14, N, 4)
    out['semi/batch/idxs'] = idxs[2:14]
    out['semi/batch/n3'] = n3
    log = rec_.log
    ch, fl, sh, hu = [], [], [], []
    k = 0
    for i in idxs[2:14]:                       # per slot: choice, then (3-D list only) flip u, shift randn, height u
        assert log[k][0] == 'choice'
        ch.append(log[k][1]); k += 1
        if i < n3:
            fl.append(float(log[k][1])); sh.append(float(log[k + 1][1])); hu.append(float(log[k + 2][1])); k += 3
        else:
            fl.append(0.0); sh.append(0.0); hu.append(0.0)
    assert k == len(log)
    out['semi/batch/choice'], out['semi/batch/flip_u'] = np.stack(ch), np.array(fl)
    out['semi/batch/shift_randn'], out['semi/batch/height_u'] = np.array(sh), np.array(hu)
    for j, kname in ((0, 'pc'), (2, 'seg'), (3, 'center'), (4, 'angle_cls'), (5, 'angle_res'), (6, 'size_cls'), (7, 'size_res'),
                     (11, 'rot_angle'), (13, 'one_hot'), (14, 'is_data_2D')):
        out['semi/batch/' + kname] = np.asarray(b[j])

    # ---- the Box-PC Fit sample generator (IoU = the oracle's restatement, see header) ----------------------------------------------
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    from oracle.ref_box import box3d_iou as oracle_iou
    IOU_IMPL[0] = oracle_iou
    cp, sp, ap = 0.8, 0.2, np.pi                                       # config.py BOXPC_*_PERTURBATION defaults
    nofit, fit = [0.01, 0.25], [0.7, 1.0]
    bd = bp.BoxPCFitDataset(classes, N, cp, sp, ap, random_flip=True, random_shift=True, rotate_to_center=True, overwritten_data_path=path)
    np.random.seed(16)
    out['boxpc/perturbation'], out['boxpc/nofit'], out['boxpc/fit'], out['boxpc/count'] = np.array([cp, sp, ap]), np.array(nofit), np.array(fit), 16
    for i in range(16):
        is_fit = i % 2 == 0
        with DrawRecorder() as rec_:
            item = bd.get(i, is_fit, nofit, fit)
        pfx = 'boxpc/%d/' % i
        rnd = rec_.take('random')
        out[pfx + 'is_fit'], out[pfx + 'choice'] = is_fit, rec_.take('choice')[0]
        out[pfx + 'flip_u'], out[pfx + 'height_u'], out[pfx + 'shift_randn'] = rnd[0], rnd[1], rec_.take('randn')[0]
        scale = 1 - np.mean(fit if is_fit else nofit)
        u = rec_.take('uniform')
        assert len(u) % 3 == 0
        cand = []
        for t in range(len(u) // 3):                                    # back to uniforms in [0,1): centre, size, angle of candidate t
            cand.append(np.concatenate([(u[3 * t] + cp * scale) / (2 * cp * scale), (u[3 * t + 1] + sp * scale) / (2 * sp * scale),
                                        [u[3 * t + 2] / (ap * scale)]]))
        out[pfx + 'cand_u'] = np.stack(cand)
        for j, k in ((0, 'point_set'), (2, 'seg'), (3, 'center'), (4, 'angle_cls'), (5, 'angle_res'), (6, 'size_cls'), (7, 'size_res'),
                     (13, 'one_hot'), (14, 'new_center'), (15, 'new_angle_cls'), (16, 'new_angle_res'), (17, 'new_size_cls'),
                     (18, 'new_size_res'), (19, 'box_iou'), (20, 'y_center_delta'), (21, 'y_size_delta'), (22, 'y_angle_delta')):
            out[pfx + k] = np.asarray(item[j])
    # ---- the stage-b statistics classes of train_boxpc.py (IoU = the oracle's, as above) ------------------------------------------
    ns = {'np': np, 'class2size': seg.class2size, 'class2angle': seg.class2angle, 'get_3d_box': seg.get_3d_box,
          'box3d_iou': lambda a, b: IOU_IMPL[0](a, b)}
    BoxDeltaIOUStats, ClassificationStats = reference_classes(
        os.path.join(REF, 'sunrgbd', 'sunrgbd_detection', 'train_boxpc.py'), ['BoxDeltaIOUStats', 'ClassificationStats'], ns)
    n = 48
    scls = r.randint(0, 6, size=n)
    pred, yfit = (r.uniform(size=n) < 0.5).astype(np.int64), (r.uniform(size=n) < 0.5).astype(np.int64)
    cs = ClassificationStats(types_)
    cs.add_prediction(pred[:20], yfit[:20], scls[:20])
    cs.add_prediction(pred[20:], yfit[20:], scls[20:])
    cs.add_loss(1.5); cs.add_loss(0.25); cs.add_loss(2.0)
    out['stats/cls/pred'], out['stats/cls/y_fit'], out['stats/cls/y_cls'], out['stats/cls/mean_loss'] = pred, yfit, scls, cs.get_mean_loss()
    st = cs.get_batch_stats()
    out['stats/cls/classes'] = np.array(sorted(st))
    out['stats/cls/table'] = np.array([st[k] for k in sorted(st)], dtype=np.float64)

    def boxes(noise):
        return (c0 + r.normal(size=(n, 3)) * noise, hc.copy(), hr + r.normal(size=n) * noise, scls.copy(), sr + r.normal(size=(n, 3)) * noise * 0.3)
    hc, hr, sr = r.randint(0, 12, size=n), r.uniform(-0.2, 0.2, size=n), r.normal(size=(n, 3)) * 0.1
    c0 = r.normal(size=(n, 3)) * 0.3 + [0, 0, 3]
    ybox, ori, dele = (c0, hc, hr, scls, sr), boxes(0.25), boxes(0.08)
    bs = BoxDeltaIOUStats(types_)
    half = lambda b, s: tuple(x[s] for x in b)
    bs.add_prediction(half(ori, slice(0, 30)), half(dele, slice(0, 30)), half(ybox, slice(0, 30)), scls[:30])
    bs.add_prediction(half(ori, slice(30, n)), half(dele, slice(30, n)), half(ybox, slice(30, n)), scls[30:])
    st = bs.get_batch_stats()
    for tag, b in (('y', ybox), ('ori', ori), ('del', dele)):
        for nm, arr_ in zip(('center', 'heading_cls', 'heading_res', 'size_cls', 'size_res'), b):
            out['stats/box/%s/%s' % (tag, nm)] = np.asarray(arr_)
    out['stats/box/classes'] = np.array(sorted(st))
    out['stats/box/table'] = np.array([st[k] for k in sorted(st)], dtype=np.float64)
    IOU_IMPL[0] = axis_aligned_iou

    # ---- test_semisup.py: detection scores, decode, result files -------------------------------------------------------------------
    ns = {'np': np, 'os': os, 'NUM_HEADING_BIN': seg.NUM_HEADING_BIN, 'NUM_SIZE_CLUSTER': seg.NUM_SIZE_CLUSTER, 'roi_seg_box3d_dataset': seg}
    softmax, inference, write_results = reference_classes(os.path.join(REF, 'sunrgbd', 'sunrgbd_detection', 'test_semisup.py'),
                                                          ['softmax', 'inference', 'write_detection_results'], ns)
    nb, bsz, npt = 3, 4, 32
    tot = nb * bsz
    net = dict(logits=r.normal(size=(tot, npt, 2)) * 2, center=r.normal(size=(tot, 3)), hs=r.normal(size=(tot, 12)) * 3,
               hr=r.normal(size=(tot, 12)) * 0.1, ss=r.normal(size=(tot, 10)) * 3, sr=r.normal(size=(tot, 10, 3)) * 0.1,
               fit=r.uniform(size=tot))
    net['logits'][5, :, 1] = -9.0                                        # a frustum whose mask is empty

    class Session:                                                       # hands the prepared arrays back batch by batch
        def __init__(self):
            self.i = 0

        def run(self, run_ops, feed_dict=None):
            sl = slice(self.i * bsz, (self.i + 1) * bsz)
            self.i += 1
            return [net[k][sl] for k in run_ops]
    ops = {'pc_pl': 'pc', 'one_hot_vec_pl': 'oh', 'is_training_pl': 'tr', 'logits': 'logits',
           'end_points': {'F_center': 'center', 'F_heading_scores': 'hs', 'F_heading_residuals': 'hr', 'F_size_scores': 'ss',
                          'F_size_residuals': 'sr', 'boxpc_fit_prob': 'fit'}}
    pcs, ohs = np.zeros((tot, npt, 4)), np.zeros((tot, 10))
    for k, v in net.items():
        out['infer/net/' + k] = v
    for tag, use_fit in (('plain', False), ('with_fit', True)):
        res = inference(Session(), ops, pcs, ohs, Py2Int(bsz), prefix='F_', use_boxpc_fit_prob=use_fit)
        for nm, v in zip(('seg', 'center', 'heading_cls', 'heading_res', 'size_cls', 'size_res', 'score'), res):
            out['infer/%s/%s' % (tag, nm)] = np.asarray(v)
    out['infer/softmax'] = softmax(net['hs'])
    import shutil
    import tempfile
    tmpd = tempfile.mkdtemp()
    test_classes = ['bed', 'chair', 'desk']
    tcls = [test_classes[i] for i in r.randint(0, 3, size=tot)]
    box2d = r.uniform(0, 500, size=(tot, 4))
    rot = r.uniform(-np.pi, np.pi, size=tot)
    ids = r.randint(1, 9999, size=tot)
    res = [out['infer/plain/' + k] for k in ('center', 'heading_cls', 'heading_res', 'size_cls', 'size_res', 'score')]
    write_results(os.path.join(tmpd, 'res'), test_classes, list(ids), tcls, list(box2d), [c.copy() for c in res[0]], list(res[1]),
                  list(res[2]), list(res[3]), list(res[4]), list(rot), list(res[5]))
    out['results/classes'], out['results/type'], out['results/box2d'], out['results/rot'], out['results/ids'] = \
        np.array(test_classes), np.array(tcls), box2d, rot, ids
    for c in test_classes:
        with open(os.path.join(tmpd, 'res', c + '_pred.txt')) as f:
            out['results/file/' + c] = np.array(f.read())
    shutil.rmtree(tmpd)

    np.savez_compressed(os.path.join(HERE, 'reference_vectors.npz'), **out)
    with open(os.path.join(HERE, 'reference_config.json'), 'w') as f:
        json.dump(config_vectors(), f, indent=1, sort_keys=True)
    print('wrote %d arrays, frustum file %d bytes' % (len(out), os.path.getsize(path)))


if __name__ == '__main__':
    main()
