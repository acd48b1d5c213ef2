"""A data set of ragged frustums resident in HBM + the launch that assembles a training batch from it on the device.

The reference keeps the frustums as Python lists and builds every batch on the host (ROISemiDataset.get_batch ->
get_classes3D, sunrgbd/sunrgbd_detection/roi_semi_dataset.py:283-347, 482-535): at ~20k frustums/s per GPU that loop, not
the GPU, bounds training.  Here the whole data set (points, per-point labels, per-frustum box labels) is uploaded once
-- SUN-RGBD's training frustums are a few GB of the 288 GB -- and `t3d_batch_assemble` (csrc/data.hip) draws, rotates,
augments and labels a batch straight into the model's input buffers.  With an epoch permutation walked by the device step
counter, the input pipeline becomes part of the captured step.

The SUN-RGBD pickle reader itself is out of scope (gzip-pickle + cv2 + Python-2 cPickle); `from_lists`
takes the same per-frustum fields the reference's loader produces, `synthetic` makes frustums of the SURVEY 8d distribution.
"""
import numpy as np
import torch

from . import abi
from .abi import fptr, iptr
from .constants import MEAN_DIMS_ARR, NUM_CLASS, type2class


class DeviceFrustumSet:
    def __init__(self, rt, points, seg, offsets, frustum_angle, box_center, heading, size, cls):
        """points [total, C_src] fp32, seg [total] int32, offsets [F+1] int64 (ragged rows of frustum f:
        offsets[f]..offsets[f+1]), per-frustum frustum_angle [F], box_center [F,3], heading [F], size [F,3] (l,w,h), cls [F]."""
        dev = rt.device
        up = lambda a, dt: torch.as_tensor(np.ascontiguousarray(a)).to(dt).to(dev)
        self.rt = rt
        self.points, self.seg, self.offsets = up(points, torch.float32), up(seg, torch.int32), up(offsets, torch.int64)
        self.frustum_angle, self.box_center = up(frustum_angle, torch.float32), up(box_center, torch.float32)
        self.heading, self.size, self.cls = up(heading, torch.float32), up(size, torch.float32), up(cls, torch.int32)
        self.F, self.C_src = int(self.cls.shape[0]), int(self.points.shape[1])
        assert self.offsets.shape[0] == self.F + 1 and int(self.offsets[-1]) == self.points.shape[0]
        self.perm = torch.arange(self.F, dtype=torch.int32, device=dev)
        rt.allocs.extend([self.points, self.seg, self.offsets, self.frustum_angle, self.box_center, self.heading, self.size, self.cls,
                          self.perm])

    def set_camera(self, rtilt, k, box2d, img_dims):
        """Camera side of the weak losses, per frustum: Rtilt [F,3,3], K [F,3,3], 2-D box [F,4] (left, top, right, bottom), image
        (rows, cols) [F,2] (roi_semi_dataset.py:243-246).  t3d_batch_assemble copies them to the batch slots."""
        up = lambda a, n: torch.as_tensor(np.ascontiguousarray(np.asarray(a, np.float32).reshape(self.F, n))).to(self.rt.device)
        self.cam = (up(rtilt, 9), up(k, 9), up(box2d, 4), up(img_dims, 2))
        self.rt.allocs.extend(self.cam)
        return self

    @classmethod
    def from_lists(cls, rt, points_l, label_l, frustum_angle_l, box3d_center_l, heading_l, size_l, cls_id_l):
        """The reference's per-frustum lists (roi_semi_dataset.py:240-252; box centre = (corner0 + corner6) / 2)."""
        counts = np.array([len(p) for p in points_l], dtype=np.int64)
        offsets = np.concatenate([[0], np.cumsum(counts)])
        return cls(rt, np.concatenate(points_l).astype(np.float32), np.concatenate(label_l).astype(np.int32), offsets,
                   np.asarray(frustum_angle_l), np.asarray(box3d_center_l), np.asarray(heading_l), np.asarray(size_l), np.asarray(cls_id_l))

    @classmethod
    def from_pickle(cls, rt, path, classes=None):
        """A frustum file of the reference (`frustums/*.zip.pickle`, written by sunrgbd_data.py:193-195 and read by
        roi_semi_dataset.py:204-252 / box_pc_fit_dataset.py:59-60): a gzip'd pickle of 13 lists [idx, box2d, box3d (8,3), image_crop,
        points (n,6), label (n,), cls_type (str), heading, size (l,w,h), rtilt, k, frustum_angle, img_dims].  `classes`: keep only these
        class names (the dataset classes' `classes` argument)."""
        idx_l, box2d_l, box3d_l, _, points_l, label_l, cls_type_l, heading_l, size_l, rtilt_l, k_l, frustum_angle_l, img_dims_l = \
            load_zipped_pickle(path)
        keep = [i for i, t in enumerate(cls_type_l) if classes is None or t in classes]
        if not keep:
            raise ValueError('%s: no frustum of classes %s' % (path, classes))
        pick = lambda lst: [lst[i] for i in keep]
        centers = [(np.asarray(b)[0, :] + np.asarray(b)[6, :]) / 2.0 for b in pick(box3d_l)]      # roi_semi_dataset.py:356-358
        ds = cls.from_lists(rt, [np.asarray(p) for p in pick(points_l)], [np.asarray(l) for l in pick(label_l)], pick(frustum_angle_l),
                            centers, pick(heading_l), pick(size_l), [type2class[t] for t in pick(cls_type_l)])
        ds.image_ids = pick(idx_l)
        ds.box3d = [np.asarray(b, np.float64) for b in pick(box3d_l)]        # label corners in camera coordinates (evaluation)
        ds.class_names = pick(cls_type_l)
        ds.set_camera(pick(rtilt_l), pick(k_l), pick(box2d_l), [np.asarray(d)[:2] for d in pick(img_dims_l)])
        return ds

    @classmethod
    def from_detection_pickle(cls, rt, path, classes=None):
        """A frustum file extracted from 2-D detections (`--from_rgb_detection`, roi_seg_box3d_dataset.py:207-222): a gzip'd pickle of 7
        lists [idx, box2d, image_crop, points (n,6), cls_type, frustum_angle, prob].  There are no 3-D labels: the label slots hold a
        zero centre / heading and the class' mean size (zero residual); `prob` is the detection score."""
        idx_l, box2d_l, _, points_l, cls_type_l, frustum_angle_l, prob_l = load_zipped_pickle(path)
        cls_type_l = [t.decode() if isinstance(t, bytes) else t for t in cls_type_l]
        keep = [i for i, t in enumerate(cls_type_l) if classes is None or t in classes]
        if not keep:
            raise ValueError('%s: no frustum of classes %s' % (path, classes))
        pick = lambda lst: [lst[i] for i in keep]
        ids = [type2class[t] for t in pick(cls_type_l)]
        pts = [np.asarray(p) for p in pick(points_l)]
        ds = cls.from_lists(rt, pts, [np.zeros(len(p), np.int32) for p in pts], pick(frustum_angle_l), np.zeros((len(keep), 3)),
                            np.zeros(len(keep)), MEAN_DIMS_ARR[ids], ids)
        ds.image_ids, ds.class_names, ds.box2d, ds.prob = pick(idx_l), pick(cls_type_l), pick(box2d_l), [float(p) for p in pick(prob_l)]
        return ds

    @classmethod
    def synthetic(cls, rt, n_frustums, num_channel=6, seed=0, min_points=400, max_points=3000):
        """Frustums of the SURVEY 8d distribution with ragged point counts (real frustums have a few hundred to a few thousand)."""
        host = synthetic_frustums(n_frustums, num_channel, seed, min_points, max_points)
        cam = synthetic_cameras(n_frustums, seed)
        return cls(rt, **host).set_camera(**cam)

    def partition(self, rank, world, batch_size, steps=None):
        """One pass over the data set per epoch, shared between `world` data-parallel replicas: every replica draws the SAME epoch
        permutation and walks its own slice perm[rank::world], cut to whole batches -- the reference's
        num_batches = len(TRAIN_DATASET) / BATCH_SIZE drops the remainder too (train_semisup.py:330-349).  The walk length is a
        multiple of the batch size, so the device-side position (step * B + b) % length wraps exactly at the epoch boundaries and
        no frustum is repeated or skipped inside a pass.  Returns the number of steps of an epoch (<= `steps` when given)."""
        per_rank = self.F // world
        n = per_rank // batch_size
        if steps:
            n = min(n, int(steps))
        if n <= 0:
            raise ValueError('data set of %d frustums is smaller than one batch of %d on each of %d replicas' % (self.F, batch_size, world))
        self.rank, self.world, self.walk_len = rank, world, n * batch_size
        return n

    def shuffle(self, seed):
        """New epoch order (the reference shuffles train_idxs once per epoch, train_semisup.py:343).  After partition(): the
        replica's slice of the common permutation (the seed must not depend on the rank)."""
        r = np.random.RandomState(seed)
        perm = r.permutation(self.F).astype(np.int32)
        if getattr(self, 'walk_len', None):
            mine = perm[self.rank::self.world][:self.walk_len]
            self.perm[:self.walk_len].copy_(torch.as_tensor(mine))
        else:
            self.perm.copy_(torch.as_tensor(perm))
        for lst, dev in getattr(self, 'subsets', []):
            dev.copy_(torch.as_tensor(lst[r.permutation(len(lst))]))

    def class_groups(self, subset=None):
        """Per-class index lists of the frustums in `subset` (default: all), cls_to_idx_map of the reference data sets
        (roi_semi_dataset.py:222-238), on the device for t3d_sample_equal_classes."""
        cls = self.cls.cpu().numpy()
        idx = np.arange(self.F, dtype=np.int32) if subset is None else np.asarray(subset, np.int32)
        present = sorted(set(int(c) for c in cls[idx]))
        members = np.concatenate([idx[cls[idx] == c] for c in present]).astype(np.int32)
        offsets = np.concatenate([[0], np.cumsum([int((cls[idx] == c).sum()) for c in present])]).astype(np.int32)
        dev = self.rt.device
        t = (torch.as_tensor(members).to(dev), torch.as_tensor(offsets).to(dev), len(present))
        self.rt.allocs.extend(t[:2])
        return t

    def sample_equal_args(self, hyper, B, sample_out, is_data_2D=None, seed=0, equal_prob=1.0, alternate=False, order_draws=None,
                          member_draws=None, prob_draw=None):
        """Argument struct of t3d_sample_equal_classes writing the frustum index of every batch slot into `sample_out` [B]."""
        a = abi.SampleEqualClassesArgs()
        if alternate:
            sets = [(lst, dev) for lst, dev in self.subsets]
        else:
            sets = [(None, self.perm)]
        keep = []
        for i, (lst, perm) in enumerate(sets):
            m, o, n = self.class_groups(lst)
            a.set[i] = abi.ClassGroups(iptr(m), iptr(o), n, iptr(perm), int(perm.numel()))
            keep += [m, o]
        a.B, a.seed, a.hyper = B, seed, fptr(hyper)
        a.order_draws, a.member_draws, a.equal_prob, a.prob_draw = fptr(order_draws), fptr(member_draws), float(equal_prob), fptr(prob_draw)
        a.sample, a.is_data_2D = iptr(sample_out), iptr(is_data_2D)
        a._keep = (keep, order_draws, member_draws, prob_draw, sample_out)
        return a

    def mark_2d_classes(self, classes_2d):
        """SEMI_SAMPLING_METHOD BATCH over the combined data set (roi_semi_dataset.py:482-535: the 3-D-label list followed by the
        2-D-label list): frustums of `classes_2d` carry is_data_2D = 1, i.e. their 3-D labels are withheld from the loss."""
        flag = np.isin(self.cls.cpu().numpy(), list(classes_2d)).astype(np.int32)
        self.is_2D = torch.as_tensor(flag).to(self.rt.device)
        self.rt.allocs.append(self.is_2D)
        return self

    def split_by_class(self, classes_2d):
        """ALTERNATE_BATCH sampling (train_semisup_adv.py:538-565): the frustums of `classes_2d` (class ids whose 3-D labels
        are withheld, SUNRGBD_SEMI_TEST_CLS) form the weak list, the rest the strong list; each is walked in its own shuffled
        order on alternate steps."""
        cls = self.cls.cpu().numpy()
        weak = np.nonzero(np.isin(cls, list(classes_2d)))[0].astype(np.int32)
        strong = np.nonzero(~np.isin(cls, list(classes_2d)))[0].astype(np.int32)
        assert len(weak) and len(strong), 'both lists must be non-empty'
        dev = self.rt.device
        self.subsets = [(weak, torch.as_tensor(weak).to(dev)), (strong, torch.as_tensor(strong).to(dev))]
        self.rt.allocs.extend([d for _, d in self.subsets])
        return self

    def assemble_args(self, inputs, hyper, B, N, C, seed=0, sample=None, choice=None, aug=None, rotate_to_center=True,
                      random_flip=True, random_shift=True, alternate=False):
        """Argument struct that writes a batch into `inputs` (nets.Inputs).  sample=None: walk the epoch permutation with the
        device step counter; choice / aug given: explicit draws (parity tests)."""
        a = abi.BatchAssembleArgs()
        a.points, a.seg, a.offsets = fptr(self.points), iptr(self.seg), abi.C.cast(abi.C.c_void_p(self.offsets.data_ptr()),
                                                                               abi.C.POINTER(abi.C.c_int64))
        a.frustum_angle, a.box_center, a.heading, a.size, a.cls = fptr(self.frustum_angle), fptr(self.box_center), fptr(self.heading), \
            fptr(self.size), iptr(self.cls)
        if sample is None:
            a.sample, a.sample_len = iptr(self.perm), (getattr(self, 'walk_len', None) or self.F)
        else:
            a.sample, a.sample_len = iptr(sample), 0
        a.choice, a.aug = iptr(choice), fptr(aug)
        a.C_src, a.C, a.B, a.N = self.C_src, C, B, N
        a.rotate_to_center, a.random_flip, a.random_shift = int(rotate_to_center), int(random_flip), int(random_shift)
        a.seed, a.hyper = seed, fptr(hyper)
        a.pc, a.y_seg, a.y_center = fptr(inputs.pc), iptr(inputs.y_seg), fptr(inputs.y_center)
        a.y_orient_cls, a.y_orient_reg = iptr(inputs.y_orient_cls), fptr(inputs.y_orient_reg)
        a.y_dims_cls, a.y_dims_reg, a.one_hot = iptr(inputs.y_dims_cls), fptr(inputs.y_dims_reg), fptr(inputs.one_hot_vec)
        a.is_data_2D = iptr(inputs.is_data_2D)
        a.rot_angle = fptr(getattr(inputs, 'rot_frust', None))          # the reference's rot_frust feed is the batch's rot_angle
        cam = getattr(self, 'cam', None)
        if cam is not None and getattr(inputs, 'Rtilt', None) is not None:
            a.cam_rtilt, a.cam_k, a.cam_box2d, a.cam_img_dim = fptr(cam[0]), fptr(cam[1]), fptr(cam[2]), fptr(cam[3])
            a.Rtilt, a.K, a.box2D, a.img_dim = fptr(inputs.Rtilt), fptr(inputs.K), fptr(inputs.box2D), fptr(inputs.img_dim)
        a.frustum_is_2D = iptr(getattr(self, 'is_2D', None))
        a.ld_pc = int(inputs.pc.shape[1])
        if alternate:
            (_, weak), (_, strong) = self.subsets
            a.sample, a.sample_len, a.sample2, a.sample2_len = iptr(weak), int(weak.numel()), iptr(strong), int(strong.numel())
        a._keep = (sample, choice, aug)
        return a


def load_zipped_pickle(path):
    """sunrgbd_data/utils.py:345-348.  The reference's files are Python-2 cPickle streams holding NumPy arrays and byte strings:
    `latin1` decodes both (str class names included) under Python 3."""
    import gzip
    import pickle
    with gzip.open(path, 'rb') as f:
        obj = pickle.load(f, encoding='latin1')
    return [[t.decode() if isinstance(t, bytes) else t for t in lst] if i == 6 else lst for i, lst in enumerate(obj)] \
        if isinstance(obj, (list, tuple)) and len(obj) == 13 else obj


def save_zipped_pickle(obj, path, protocol=2):
    """sunrgbd_data/utils.py:341-343 (the reference passes -1, the highest protocol of its Python 2: protocol 2)."""
    import gzip
    import pickle
    with gzip.open(path, 'wb') as f:
        pickle.dump(obj, f, protocol)


def boxpc_perturb_args(inputs, hyper, B, c, seed=0, max_rounds=16, fit_draw=None, cand_draws=None):
    """Argument struct of t3d_boxpc_perturb working in place on `inputs` (nets.Inputs) after t3d_batch_assemble: the label box
    becomes the perturbed box the Box-PC net sees, y_box_iou / y_*_delta its targets.  `c`: the BOXPC_* flags (config.py:22-29)."""
    a = abi.BoxPcPerturbArgs()
    a.center, a.orient_cls, a.orient_reg = fptr(inputs.y_center), iptr(inputs.y_orient_cls), fptr(inputs.y_orient_reg)
    a.dims_cls, a.dims_reg = iptr(inputs.y_dims_cls), fptr(inputs.y_dims_reg)
    a.y_box_iou, a.y_center_delta, a.y_dims_delta, a.y_orient_delta = fptr(inputs.y_box_iou), fptr(inputs.y_center_delta), \
        fptr(inputs.y_dims_delta), fptr(inputs.y_orient_delta)
    a.center_perturbation, a.size_perturbation, a.angle_perturbation = float(c.BOXPC_CENTER_PERTURBATION), \
        float(c.BOXPC_SIZE_PERTURBATION), float(c.BOXPC_ANGLE_PERTURBATION)
    (a.fit_lo, a.fit_hi), (a.nofit_lo, a.nofit_hi) = [float(v) for v in c.BOXPC_FIT_BOUNDS], [float(v) for v in c.BOXPC_NOFIT_BOUNDS]
    a.proportion_fit = float(c.BOXPC_PROPORTION_OF_BOXPC_FIT)
    a.fit_draw, a.cand_draws, a.max_rounds, a.seed, a.hyper, a.B = fptr(fit_draw), fptr(cand_draws), max_rounds, seed, fptr(hyper), B
    a._keep = (fit_draw, cand_draws)
    return a


def synthetic_frustums(n_frustums, num_channel=6, seed=0, min_points=400, max_points=3000):
    """Host arrays for DeviceFrustumSet: xyz in camera-like coordinates before the centre-view rotation, extra channels
    U(0,1), ~30 % foreground points clustered around the box centre, heading / size / class as in synthetic.make_batch."""
    r = np.random.RandomState(seed)
    counts = r.randint(min_points, max_points + 1, size=n_frustums).astype(np.int64)
    offsets = np.concatenate([[0], np.cumsum(counts)])
    total = int(offsets[-1])
    fang = r.uniform(-0.6, 0.6, size=n_frustums) - np.pi / 2          # centre-view rotation pi/2 + angle in (-0.6, 0.6)
    depth = r.uniform(1.5, 5.5, size=n_frustums)
    cls_id = r.randint(0, NUM_CLASS, size=n_frustums).astype(np.int32)
    size = MEAN_DIMS_ARR[cls_id] + r.normal(size=(n_frustums, 3)) * 0.1
    heading = r.uniform(-np.pi, np.pi, size=n_frustums)
    pts = np.zeros((total, num_channel), np.float32)
    seg = np.zeros(total, np.int32)
    center = np.zeros((n_frustums, 3))
    for f in range(n_frustums):
        lo, hi = offsets[f], offsets[f + 1]
        n = hi - lo
        rot = np.pi / 2 + fang[f]
        fg = r.uniform(size=n) < 0.3
        xc = np.where(fg[:, None], r.normal(size=(n, 3)) * 0.3 + [0.0, 0.2, depth[f]],
                      np.stack([r.normal(size=n) * 0.6, r.normal(size=n) * 0.6, r.uniform(1.0, 6.0, size=n)], 1))
        cen_c = np.array([0.0, 0.2, depth[f]]) + r.normal(size=3) * 0.1
        # un-rotate: the stored frame is the one the centre-view rotation maps onto xc (inverse of [[c,-s],[s,c]])
        c, s = np.cos(rot), np.sin(rot)
        un = lambda v: np.stack([v[..., 0] * c + v[..., 2] * s, v[..., 1], -v[..., 0] * s + v[..., 2] * c], -1)
        pts[lo:hi, :3] = un(xc)
        pts[lo:hi, 3:] = r.uniform(size=(n, num_channel - 3))
        seg[lo:hi] = fg
        center[f] = un(cen_c)
    return dict(points=pts, seg=seg, offsets=offsets, frustum_angle=fang, box_center=center, heading=heading, size=size, cls=cls_id)


def synthetic_cameras(n_frustums, seed=0):
    """A SUN-RGBD-like calibration per synthetic frustum: K of the data set's Kinect v2 images, a small tilt, a 2-D box."""
    r = np.random.RandomState(seed + 7919)
    tilt = r.normal(0, 0.05, size=n_frustums)
    rtilt = np.stack([np.array([[1, 0, 0], [0, np.cos(t), -np.sin(t)], [0, np.sin(t), np.cos(t)]]) for t in tilt])
    k = np.tile(np.array([[529.5, 0, 365.0], [0, 529.5, 265.0], [0, 0, 1.0]]), (n_frustums, 1, 1))
    cx, cy, hw, hh = r.uniform(150, 580, n_frustums), r.uniform(100, 430, n_frustums), r.uniform(40, 220, n_frustums), \
        r.uniform(40, 200, n_frustums)
    return dict(rtilt=rtilt, k=k, box2d=np.stack([cx - hw, cy - hh, cx + hw, cy + hh], 1),
                img_dims=np.tile(np.array([530.0, 730.0]), (n_frustums, 1)))


def open_training_set(rt, FLAGS, num_channel, classes=None, seed=0):
    """`--frustum_file path` (a frustum file of the reference, frustums/*.zip.pickle, restricted to `classes`) or
    `--device_data F` (F synthetic frustums): the training set resident in HBM, or None for host-fed synthetic batches."""
    if getattr(FLAGS, 'frustum_file', None):
        return DeviceFrustumSet.from_pickle(rt, FLAGS.frustum_file, classes=classes)
    if FLAGS.device_data:
        return DeviceFrustumSet.synthetic(rt, FLAGS.device_data, num_channel=max(num_channel, 6), seed=seed)
    return None


class DeviceEvalSource:
    """Held-out frustums from the generator of the training set, resident in HBM; batch i = frustums [i*B, (i+1)*B) assembled by
    t3d_batch_assemble without augmentation (the reference's TEST_DATASET: random_flip / random_shift off) into the graph's feed
    buffers.  `boxpc_perturb` (the BOXPC_* flags): followed by the Box-PC Fit sample generator, as BoxPCFitDataset does for its
    test split (train_boxpc.py:119-124)."""

    def __init__(self, graph, n_frustums=None, seed=0, boxpc_perturb=None, dataset=None):
        from .engine import Plan
        e = graph.engine
        self.g, self.B = graph, e.B
        self.ds = dataset if dataset is not None else DeviceFrustumSet.synthetic(graph.rt, n_frustums, num_channel=max(e.C, 6), seed=seed)
        self.ds.perm.copy_(torch.arange(self.ds.F, dtype=torch.int32))      # file order; the walk wraps around past the end
        self.counter = graph.rt.zeros(4)
        self.plan = Plan(graph.rt)
        self.plan.add('t3d_batch_assemble', self.ds.assemble_args(graph.inputs, self.counter, e.B, e.rpf, e.C, seed=seed, random_flip=False,
                                                                 random_shift=False))
        if boxpc_perturb is not None:
            self.plan.add('t3d_boxpc_perturb', boxpc_perturb_args(graph.inputs, self.counter, e.B, boxpc_perturb, seed=seed ^ 0x5bd1e995))

    def frustums_of(self, i):
        """Indices (file order) of the frustums in batch i."""
        return (i * self.B + np.arange(self.B)) % self.ds.F

    def load(self, i):
        """Assemble batch i; returns its per-point labels [B, N]."""
        self.counter[0] = float(i)
        self.plan.run()
        return self.g.inputs.y_seg.view(self.B, -1).cpu().numpy()


def open_eval_source(graph, FLAGS, classes=None, boxpc_perturb=None):
    """Held-out frustums for eval_one_epoch: `--eval_file path` (a frustum file of the reference restricted to `classes`; every frustum
    is visited, FLAGS.eval_batches is set accordingly) or FLAGS.eval_batches synthetic batches from the generator of the training
    set."""
    B = graph.engine.B
    if getattr(FLAGS, 'eval_file', None):
        ds = DeviceFrustumSet.from_pickle(graph.rt, FLAGS.eval_file, classes=classes)
        FLAGS.eval_batches = (ds.F + B - 1) // B
        return DeviceEvalSource(graph, dataset=ds, seed=FLAGS.seed, boxpc_perturb=boxpc_perturb)
    return DeviceEvalSource(graph, FLAGS.eval_batches * B, FLAGS.seed + 424243, boxpc_perturb=boxpc_perturb)
