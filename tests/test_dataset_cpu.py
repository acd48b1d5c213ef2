"""Device-side batch assembly (t3d_batch_assemble): the NumPy specification against the oracle's restatement of the
reference's get_classes3D / get_batch on identical draws, and the statistics of the generated draws."""
import ctypes as C

import numpy as np
import torch

from fake_t3d import FakeLib
from oracle import ref_data as D
from transferable3d_amd.dataset import DeviceFrustumSet, synthetic_frustums
from transferable3d_amd.engine import Runtime
from transferable3d_amd.nets import Graph, Inputs


def assemble(rt, host, B, N, Cc, sample=None, choice=None, aug=None, step=0, seed=7, **kw):
    g = Graph(B, N, Cc, rt=rt)
    x = Inputs(g)
    ds = DeviceFrustumSet(rt, **host)
    dev = rt.device
    t = lambda a, dt: None if a is None else torch.as_tensor(np.ascontiguousarray(a)).to(dt).to(dev)
    g.hyper[0] = step
    a = ds.assemble_args(x, g.hyper, B, N, Cc, seed=seed, sample=t(sample, torch.int32), choice=t(choice, torch.int32),
                         aug=t(aug, torch.float32), **kw)
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream) if dev.type == 'cuda' else None
    assert rt.lib.t3d_batch_assemble(C.byref(a), stream) == 0
    if dev.type == 'cuda':
        torch.cuda.synchronize()
    num = lambda v: v.detach().cpu().numpy()
    pc = num(x.pc).reshape(B, N, -1)
    assert not pc[:, :, Cc:].any()                      # rows padded to 16 bytes; the padding is written as zeros
    return {'pc': pc[:, :, :Cc], 'y_seg': num(x.y_seg).reshape(B, N), 'y_center': num(x.y_center),
            'y_orient_cls': num(x.y_orient_cls), 'y_orient_reg': num(x.y_orient_reg), 'y_dims_cls': num(x.y_dims_cls),
            'y_dims_reg': num(x.y_dims_reg), 'one_hot_vec': num(x.one_hot_vec)}, ds


def check_against_oracle(rt, flags=(True, True, True)):
    B, N, Cc = 6, 256, 4
    host = synthetic_frustums(20, num_channel=6, seed=3, min_points=50, max_points=400)
    r = np.random.RandomState(1)
    sample = r.randint(0, 20, size=B)
    counts = (host['offsets'][1:] - host['offsets'][:-1])[sample]
    choice = np.stack([r.randint(0, c, size=N) for c in counts])
    flip, rn, hu = (r.uniform(size=B) > 0.5), r.normal(size=B), r.uniform(size=B)
    kw = dict(rotate_to_center=flags[0], random_flip=flags[1], random_shift=flags[2])
    got, _ = assemble(rt, host, B, N, Cc, sample=sample, choice=choice, aug=np.stack([flip, rn, hu], 1), **kw)
    ref = D.get_batch(host, sample, choice, flip, rn, hu, Cc, **kw)
    assert np.abs(got['pc'] - ref['pc']).max() < 2e-5
    assert np.array_equal(got['y_seg'], ref['y_seg'])
    assert np.abs(got['y_center'] - ref['y_center']).max() < 2e-5
    assert np.abs(got['y_dims_reg'] - ref['y_dims_reg']).max() < 1e-6 and np.array_equal(got['y_dims_cls'], ref['y_dims_cls'])
    assert np.array_equal(got['one_hot_vec'], ref['one_hot_vec'])
    # heading: class*bin + residual reproduces the angle even where fp32 puts a boundary case in the neighbouring bin
    per = 2 * np.pi / 12
    ang = lambda c, rr: (c * per + rr) % (2 * np.pi)
    d = np.abs(ang(got['y_orient_cls'], got['y_orient_reg']) - ang(ref['y_orient_cls'], ref['y_orient_reg']))
    assert np.minimum(d, 2 * np.pi - d).max() < 1e-5
    assert (got['y_orient_cls'] == ref['y_orient_cls']).mean() >= 0.8 and np.abs(got['y_orient_reg']).max() <= per / 2 + 1e-5


def check_generated_draws(rt):
    """No explicit draws: the batch follows the epoch permutation by the step counter; resampling indices are uniform over
    each frustum's points, flips come up about half the time, the z shift is within the reference's clip bounds."""
    B, N, Cc = 32, 1024, 4
    host = synthetic_frustums(100, num_channel=6, seed=5, min_points=200, max_points=900)
    got0, ds = assemble(rt, host, B, N, Cc, step=0)
    got0b, _ = assemble(rt, host, B, N, Cc, step=0)
    got1, _ = assemble(rt, host, B, N, Cc, step=1)
    assert np.array_equal(got0['pc'], got0b['pc'])                       # a function of (seed, step) only
    assert not np.array_equal(got0['pc'], got1['pc'])
    # slot b of step s holds frustum perm[(s*B + b) % F] (identity permutation here): its class id is the label
    assert np.array_equal(got0['y_dims_cls'], host['cls'][:B]) and np.array_equal(got1['y_dims_cls'], host['cls'][B:2 * B])
    # every drawn point is a point of the right frustum: the intensity channel (untouched by the augmentation) must occur there
    for b in (0, 7, 31):
        lo, hi = host['offsets'][b], host['offsets'][b + 1]
        assert np.isin(got0['pc'][b, :, 3], host['points'][lo:hi, 3]).all()
        frac_distinct = len(np.unique(got0['pc'][b, :, 3])) / min(N, hi - lo)
        assert frac_distinct > 0.55                                      # with-replacement sampling covers ~63 % when N = count
    # per-frustum statistics over many steps: flip rate and shift bounds
    flips, ok = [], True
    for step in range(2, 8):
        g, _ = assemble(rt, host, B, N, Cc, step=step)
        nf, _ = assemble(rt, host, B, N, Cc, step=step, random_flip=False, random_shift=False)
        flips.extend(np.sign(g['y_center'][:, 0]) != np.sign(nf['y_center'][:, 0]))
        dist = np.sqrt(nf['y_center'][:, 0] ** 2 + nf['y_center'][:, 1] ** 2)     # of the centre before the shifts
        shift = g['y_center'][:, 2] - nf['y_center'][:, 2]
        ok &= bool(((shift >= 0.8 * dist - 1e-4) & (shift <= 1.2 * dist + 1e-4)).all())
        assert np.abs((g['y_center'][:, 1] - nf['y_center'][:, 1])).max() <= 0.2 + 1e-5
    assert 0.3 < np.mean(flips) < 0.7 and ok


def test_batch_assembly_matches_the_reference_restatement():
    for flags in ((True, True, True), (False, False, False), (True, False, True)):
        check_against_oracle(Runtime(device='cpu', lib=FakeLib()), flags)


def test_generated_draws_and_permutation_walk():
    check_generated_draws(Runtime(device='cpu', lib=FakeLib()))


def check_alternate_batch(rt):
    """ALTERNATE_BATCH: even steps are pure weak (2-D-label classes, is_data_2D = 1), odd steps pure strong batches; a batch
    holds distinct frustums (the reference samples without replacement within a batch)."""
    B, N, Cc = 8, 128, 4
    host = synthetic_frustums(120, num_channel=6, seed=2, min_points=64, max_points=200)
    weak_cls = (1, 2, 6, 7, 8)
    g = Graph(B, N, Cc, rt=rt)
    x = Inputs(g)
    ds = DeviceFrustumSet(rt, **host).split_by_class(weak_cls)
    ds.shuffle(5)
    a = ds.assemble_args(x, g.hyper, B, N, Cc, seed=1, alternate=True)
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream) if rt.device.type == 'cuda' else None
    seen = {0: [], 1: []}
    for step in range(6):
        g.hyper[0] = step
        assert rt.lib.t3d_batch_assemble(C.byref(a), stream) == 0
        if rt.device.type == 'cuda':
            torch.cuda.synchronize()
        cls, is2d = x.one_hot_vec.cpu().numpy().argmax(1), x.is_data_2D.cpu().numpy()
        weak = step % 2 == 0
        assert (is2d == (1 if weak else 0)).all()
        assert np.isin(cls, weak_cls).all() if weak else (~np.isin(cls, weak_cls)).all()
        # a weak batch carries no 3-D label at all (get_classes2D returns zeros), a strong batch does
        lab = [x.y_seg, x.y_center, x.y_orient_cls, x.y_orient_reg, x.y_dims_cls, x.y_dims_reg]
        assert all(not bool(t.cpu().numpy().any()) for t in lab) if weak else all(bool(t.cpu().numpy().any()) for t in lab)
        key = x.pc.cpu().numpy().reshape(B, N, -1)[:, :, :3].mean(1).round(4)        # distinct frustums -> distinct point clouds
        assert len(np.unique(key, axis=0)) == B
        seen[step % 2].append(cls.copy())
    assert not np.array_equal(seen[0][0], seen[0][1])               # the lists advance


def test_alternate_batch_sampling():
    check_alternate_batch(Runtime(device='cpu', lib=FakeLib()))


def _stream(lib_is_fake):
    if lib_is_fake:
        return None
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _perturb_inputs(r, B):
    return dict(center=(r.normal(size=(B, 3)) * 0.3 + [0, 0, 3]).astype(np.float32), ocls=r.randint(0, 12, B).astype(np.int32),
                oreg=(r.uniform(-1, 1, B) * np.pi / 12).astype(np.float32), cls=r.randint(0, 10, B).astype(np.int32),
                dreg=(r.normal(size=(B, 3)) * 0.1).astype(np.float32))


def run_perturb(lib, dev, base, B, rounds, fit_u=None, cand=None, step=0, seed=77):
    from transferable3d_amd import abi
    from transferable3d_amd.abi import fptr, iptr
    t = {k: torch.as_tensor(v.copy()).to(dev) for k, v in base.items()}
    o = dict(iou=torch.zeros(B, device=dev), dc=torch.zeros(B, 3, device=dev), ds=torch.zeros(B, 3, device=dev), da=torch.zeros(B, device=dev))
    hyper = torch.tensor([float(step), 0, 0, 0], device=dev)
    fu = torch.as_tensor(fit_u).to(dev) if fit_u is not None else None
    cd = torch.as_tensor(cand).to(dev) if cand is not None else None
    a = abi.BoxPcPerturbArgs(fptr(t['center']), iptr(t['ocls']), fptr(t['oreg']), iptr(t['cls']), fptr(t['dreg']), fptr(o['iou']), fptr(o['dc']),
                             fptr(o['ds']), fptr(o['da']), 0.8, 0.2, np.pi, 0.7, 1.0, 0.01, 0.25, 0.5, fptr(fu), fptr(cd), rounds, seed, fptr(hyper), B)
    assert lib.t3d_boxpc_perturb(C.byref(a), _stream(isinstance(lib, FakeLib))) == 0
    if dev != 'cpu':
        torch.cuda.synchronize()
    return {k: v.cpu() for k, v in t.items()}, {k: v.cpu() for k, v in o.items()}


def check_boxpc_perturb_against_oracle(lib, dev):
    """t3d_boxpc_perturb (first accepted candidate of a 64-wide candidate stream) against the oracle's restatement of
    BoxPCFitDataset.perturb_box_to_diff_ious (sequential rejection sampling) consuming the same uniforms."""
    from oracle import ref_data as RD
    from transferable3d_amd.constants import MEAN_DIMS_ARR
    r = np.random.RandomState(4)
    B, R = 24, 6
    base = _perturb_inputs(r, B)
    fit_u = r.uniform(size=B).astype(np.float32)
    cand = r.uniform(size=(B, R * 64, 7)).astype(np.float32)
    t, o = run_perturb(lib, dev, base, B, R, fit_u, cand)
    n_fit = 0
    for b in range(B):
        heading = base['ocls'][b] * (2 * np.pi / 12) + float(base['oreg'][b])
        size = MEAN_DIMS_ARR[base['cls'][b]] + base['dreg'][b].astype(np.float64)
        is_fit = fit_u[b] < 0.5
        n_fit += is_fit
        want = RD.boxpc_sample_labels(base['center'][b].astype(np.float64), heading, size, int(base['cls'][b]), is_fit, (0.7, 1.0), (0.01, 0.25),
                                      (0.8, 0.2, np.pi), cand[b].astype(np.float64))
        assert want is not None, b
        lo, hi = ((0.7, 1.0) if is_fit else (0.01, 0.25))
        assert lo < float(o['iou'][b]) < hi
        assert abs(float(o['iou'][b]) - want['y_box_iou']) < 2e-5
        assert np.abs(t['center'][b].numpy() - want['x_center']).max() < 1e-5
        assert int(t['ocls'][b]) == want['x_orient_cls'] and abs(float(t['oreg'][b]) - want['x_orient_reg']) < 1e-5
        assert np.abs(t['dreg'][b].numpy() - want['x_dims_reg']).max() < 1e-5
        assert np.abs(o['dc'][b].numpy() - want['y_center_delta']).max() < 1e-6
        assert np.abs(o['ds'][b].numpy() - want['y_dims_delta']).max() < 1e-6
        assert abs(float(o['da'][b]) - want['y_orient_delta']) < 1e-6
    assert 0 < n_fit < B


def check_boxpc_perturb_generated(lib, dev):
    """Generated draws (hash of seed, step, frustum, candidate): every sample lands inside its bounds, about half are "fit", the
    deltas reproduce the perturbed box, and a different step gives different samples."""
    from fake_t3d import box3d_iou_spec
    from transferable3d_amd.constants import MEAN_DIMS_ARR
    B = 64
    base = _perturb_inputs(np.random.RandomState(9), B)
    t, o = run_perturb(lib, dev, base, B, 4, step=3)
    iou = o['iou'].numpy()
    fit = iou > 0.7
    assert np.all(((iou > 0.7) & (iou < 1.0)) | ((iou > 0.01) & (iou < 0.25))) and 16 < fit.sum() < 48
    for b in range(0, B, 7):
        h0 = base['ocls'][b] * (2 * np.pi / 12) + base['oreg'][b]
        s0 = MEAN_DIMS_ARR[base['cls'][b]] + base['dreg'][b]
        h1 = int(t['ocls'][b]) * (2 * np.pi / 12) + float(t['oreg'][b])
        s1 = MEAN_DIMS_ARR[base['cls'][b]] + t['dreg'][b].numpy()
        assert np.abs(t['center'][b].numpy() - base['center'][b] - o['dc'][b].numpy()).max() < 1e-6
        assert np.abs(s1 - s0 - o['ds'][b].numpy()).max() < 1e-6
        assert abs(np.angle(np.exp(1j * (h1 - h0 - float(o['da'][b]))))) < 1e-5
        assert abs(box3d_iou_spec(base['center'][b], s0, h0, t['center'][b].numpy(), s1, h1)[0] - iou[b]) < 2e-5
    _, o2 = run_perturb(lib, dev, base, B, 4, step=4)
    assert not np.allclose(o2['iou'].numpy(), iou)
    return base, t, o


def test_boxpc_perturb_spec_against_oracle_on_identical_draws():
    check_boxpc_perturb_against_oracle(FakeLib(), 'cpu')


def test_boxpc_perturb_generated_draws_follow_the_sampling_law():
    check_boxpc_perturb_generated(FakeLib(), 'cpu')


def test_frustum_pickle_reader(tmp_path):
    """A file in the reference's format -- gzip'd protocol-2 pickle of the 13 lists, class names as Python-2 byte strings -- loads
    into a DeviceFrustumSet: ragged point lists, box centre = (corner0 + corner6) / 2, class filter."""
    import gzip
    import pickle
    from transferable3d_amd.dataset import load_zipped_pickle
    r = np.random.RandomState(2)
    names = [b'bed', b'chair', b'table', b'chair', b'sofa']
    n = [300, 120, 77, 500, 64]
    box3d = [r.normal(size=(8, 3)) for _ in names]
    lists = [list(range(5)), [r.uniform(size=4) for _ in names], box3d, [None] * 5, [r.normal(size=(k, 6)) for k in n],
             [(r.uniform(size=k) < 0.3).astype(np.float64) for k in n], names, list(r.uniform(-3, 3, 5)), [r.uniform(0.5, 2, 3) for _ in names],
             [np.eye(3)] * 5, [np.eye(3)] * 5, list(r.uniform(-2, -1, 5)), [np.array([640, 480])] * 5]
    path = str(tmp_path / 'train_mini.zip.pickle')
    with gzip.open(path, 'wb') as f:
        pickle.dump(lists, f, 2)
    back = load_zipped_pickle(path)
    assert back[6] == ['bed', 'chair', 'table', 'chair', 'sofa'] and np.array_equal(back[4][2], lists[4][2])
    rt = Runtime(device='cpu', lib=FakeLib())
    ds = DeviceFrustumSet.from_pickle(rt, path, classes=['chair', 'sofa'])
    assert ds.F == 3 and ds.image_ids == [1, 3, 4]
    assert ds.offsets.tolist() == [0, 120, 620, 684] and ds.cls.tolist() == [3, 3, 2]
    assert np.allclose(ds.box_center[1].numpy(), (box3d[3][0] + box3d[3][6]) / 2, atol=1e-6)
    assert np.allclose(ds.points[120:620].numpy(), lists[4][3].astype(np.float32)) and ds.seg[620:].tolist() == lists[5][4].astype(int).tolist()
    all_ds = DeviceFrustumSet.from_pickle(rt, path)
    assert all_ds.F == 5


def run_equal_sampler(lib, dev, ds, B, step, seed=5, alternate=False, equal_prob=1.0, order=None, member=None, prob=None):
    hyper = torch.tensor([float(step), 0, 0, 0], device=dev)
    out = torch.zeros(B, dtype=torch.int32, device=dev)
    flag = torch.full((B,), -1, dtype=torch.int32, device=dev)
    mk = lambda v: None if v is None else torch.as_tensor(np.asarray(v, np.float32)).to(dev)
    a = ds.sample_equal_args(hyper, B, out, is_data_2D=flag, seed=seed, equal_prob=equal_prob, alternate=alternate, order_draws=mk(order),
                             member_draws=mk(member), prob_draw=mk(prob))
    assert lib.t3d_sample_equal_classes(C.byref(a), _stream(isinstance(lib, FakeLib))) == 0
    if dev != 'cpu':
        torch.cuda.synchronize()
    return out.cpu().numpy(), flag.cpu().numpy()


def check_equal_class_sampler(rt):
    """t3d_sample_equal_classes against the oracle's restatement of `equal_samples_per_class` on identical draws, the law of the
    generated draws, the two-list (ALTERNATE_BATCH) mode and the not-balanced fallback."""
    lib, dev = rt.lib, rt.device.type if rt.device.type == 'cpu' else 'cuda'
    host = synthetic_frustums(120, num_channel=4, seed=3, min_points=130, max_points=200)
    ds = DeviceFrustumSet(rt, **host)
    cls = host['cls']
    present = sorted(set(int(c) for c in cls))
    lists = [np.nonzero(cls == c)[0] for c in present]
    n = len(present)
    r = np.random.RandomState(1)
    for B in (32, 7, 60):
        order, member = r.uniform(size=n).astype(np.float32), r.uniform(size=B).astype(np.float32)
        got, flag = run_equal_sampler(lib, dev, ds, B, step=3, order=order, member=member)
        # the oracle's shuffled np.array_split: the `B % n` groups with the smallest keys are the larger ones
        rank = np.argsort(np.argsort(order, kind='stable'), kind='stable')
        split = [np.ones(B // n + (1 if rank[i] < B % n else 0)) for i in range(n)]
        assert sorted(len(s) for s in split) == sorted(len(s) for s in np.array_split([1] * B, n))
        want = D.sample_equal_per_class(B, lists, split, member.astype(np.float64))
        assert got.tolist() == [int(w) for w in want] and (flag == 0).all()
    # generated draws: every class gets floor(B/n) or ceil(B/n) slots, slots grouped by class, different steps differ
    B = 32
    a, _ = run_equal_sampler(lib, dev, ds, B, step=10)
    b, _ = run_equal_sampler(lib, dev, ds, B, step=11)
    counts = np.bincount(cls[a], minlength=10)[present]
    assert counts.min() >= B // n and counts.max() <= B // n + 1 and (np.diff(cls[a]) >= 0).all()
    assert not np.array_equal(a, b)
    # over many steps every frustum of a class is drawn about equally often
    hits = np.zeros(len(cls))
    for s in range(200):
        np.add.at(hits, run_equal_sampler(lib, dev, ds, B, step=s)[0], 1)
    per_class = [hits[l] / hits[l].sum() for l in lists]
    assert all(np.abs(p - 1.0 / len(p)).max() < 4.0 / np.sqrt(200 * B / n) / np.sqrt(len(p)) + 0.05 for p in per_class)
    # two lists: even steps from the weak (2-D label) classes with is_data_2D = 1, odd steps from the others with 0
    weak_ids = [1, 2, 6, 7, 8]
    ds.split_by_class(weak_ids)
    e, fe = run_equal_sampler(lib, dev, ds, B, step=4, alternate=True)
    o, fo = run_equal_sampler(lib, dev, ds, B, step=5, alternate=True)
    assert np.isin(cls[e], weak_ids).all() and (fe == 1).all() and (~np.isin(cls[o], weak_ids)).all() and (fo == 0).all()
    # equal_prob = 0: the next B entries of the epoch permutation (B distinct frustums)
    p0, _ = run_equal_sampler(lib, dev, DeviceFrustumSet(rt, **host), B, step=2, equal_prob=0.0)
    ds2 = DeviceFrustumSet(rt, **host)
    p1, _ = run_equal_sampler(lib, dev, ds2, B, step=2, equal_prob=0.0)
    assert np.array_equal(p0, p1) and len(set(p1.tolist())) == B and np.array_equal(p1, ds2.perm.cpu().numpy()[2 * B:3 * B])
    # a draw above equal_prob takes the permutation, below it the balanced composition
    hi, _ = run_equal_sampler(lib, dev, ds2, B, step=2, equal_prob=0.5, prob=[0.7])
    lo, _ = run_equal_sampler(lib, dev, ds2, B, step=2, equal_prob=0.5, prob=[0.3])
    assert np.array_equal(hi, p1) and (np.diff(cls[lo]) >= 0).all()


def test_equal_samples_per_class_sampler():
    check_equal_class_sampler(Runtime(device='cpu', lib=FakeLib()))


def test_combined_data_set_marks_two_d_frustums():
    """SEMI_SAMPLING_METHOD BATCH: is_data_2D follows the per-frustum flag of the combined data set."""
    rt = Runtime(device='cpu', lib=FakeLib())
    host = synthetic_frustums(40, num_channel=4, seed=6, min_points=130, max_points=160)
    ds = DeviceFrustumSet(rt, **host).mark_2d_classes([1, 2, 6, 7, 8])
    B, N, Cc = 8, 128, 4
    g = Graph(B, N, Cc, rt=rt)
    x = Inputs(g)
    g.hyper[0] = 2.0
    a = ds.assemble_args(x, g.hyper, B, N, Cc, seed=3)
    assert rt.lib.t3d_batch_assemble(C.byref(a), None) == 0
    f = ds.perm.numpy()[2 * B:3 * B]
    assert np.array_equal(x.is_data_2D.numpy(), np.isin(host['cls'][f], [1, 2, 6, 7, 8]).astype(np.int32))
    assert 0 < x.is_data_2D.sum() < B


def test_epoch_is_one_pass_and_replicas_walk_disjoint_slices():
    """train_semisup.py:330-349: an epoch visits every frustum once (whole batches; the remainder is dropped as in the reference);
    data-parallel replicas share the epoch permutation and take disjoint slices of it; the device-side walk position
    (step * B + b) % walk_len stays aligned with the epoch boundaries."""
    from fake_t3d import FakeLib
    from transferable3d_amd.dataset import DeviceFrustumSet
    from transferable3d_amd.engine import Runtime
    F, B, world = 50, 4, 2
    seen = []
    for rank in range(world):
        ds = DeviceFrustumSet.synthetic(Runtime(device='cpu', lib=FakeLib()), F, 4, seed=1, min_points=8, max_points=16)
        steps = ds.partition(rank, world, B)
        assert steps == (F // world) // B == 6 and ds.walk_len == 24
        per_epoch = []
        for epoch in range(2):
            ds.shuffle(1000 + epoch)                  # the same seed on every replica
            perm = ds.perm.numpy()
            walk = [int(perm[((epoch * steps + s) * B + b) % ds.walk_len]) for s in range(steps) for b in range(B)]
            assert len(set(walk)) == len(walk) == 24          # nothing repeated inside a pass
            per_epoch.append(set(walk))
        seen.append(per_epoch)
        assert ds.partition(rank, world, B, steps=3) == 3 and ds.walk_len == 12
    for epoch in range(2):
        assert not (seen[0][epoch] & seen[1][epoch])           # replicas see disjoint frustums
        assert len(seen[0][epoch] | seen[1][epoch]) == 48      # 50 frustums, 2 dropped with the incomplete batch
