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
    return {'pc': num(x.pc).reshape(B, N, Cc), 'y_seg': num(x.y_seg).reshape(B, N), 'y_center': num(x.y_center),
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
        cls, is2d = x.y_dims_cls.cpu().numpy(), x.is_data_2D.cpu().numpy()
        weak = step % 2 == 0
        assert (is2d == (1 if weak else 0)).all()
        assert np.isin(cls, weak_cls).all() if weak else (~np.isin(cls, weak_cls)).all()
        key = x.y_center.cpu().numpy()[:, 1].round(4)                # distinct frustums -> distinct (shifted) centres
        assert len(np.unique(key)) == B
        seen[step % 2].append(cls.copy())
    assert not np.array_equal(seen[0][0], seen[0][1])               # the lists advance


def test_alternate_batch_sampling():
    check_alternate_batch(Runtime(device='cpu', lib=FakeLib()))
