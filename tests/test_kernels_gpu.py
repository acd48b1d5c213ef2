"""GPU: every HIP entry point of libt3d.so against the NumPy executable specification (tests/fake_t3d.py)
on identical seeded inputs, called through the C ABI (ctypes).  Sizes are small enough that the fp64 spec
finishes in seconds; tolerances are fp32 accumulation bounds, written per test."""
import ctypes as C

import numpy as np
import pytest
import torch

from fake_t3d import FakeLib
from transferable3d_amd import abi
from transferable3d_amd.abi import fptr, iptr

pytestmark = pytest.mark.gpu


def _dev(name):
    return torch.device(name)


def _mk(dev, a):
    return torch.as_tensor(np.ascontiguousarray(a)).to(dev)


def _run_both(hip_lib, make, name):
    """make(dev) -> (args, outputs dict).  Runs the spec on CPU and the HIP kernel on the GPU."""
    fake = FakeLib()
    a_c, out_c = make(_dev('cpu'))
    assert getattr(fake, name)(C.byref(a_c), None) == 0
    a_g, out_g = make(_dev('cuda'))
    rc = getattr(hip_lib, name)(C.byref(a_g), C.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0, rc
    torch.cuda.synchronize()
    return out_c, {k: v.cpu() for k, v in out_g.items()}


def _close(a, b, rtol, atol, what):
    a, b = a.double().numpy(), b.double().numpy()
    err = np.abs(a - b)
    tol = atol + rtol * np.abs(a)
    assert (err <= tol).all(), (what, float(err.max()), float(np.abs(a).max()), int((err > tol).sum()))


# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('M,K,N,rpf,mode', [
    (512, 64, 64, 256, 'bn'), (512, 4, 64, 128, 'raw'), (256, 3, 128, 128, 'sub'), (512, 128, 256, 256, 'bn_pool'),
    (256, 64, 512, 128, 'bn_rowbias'), (256, 128, 1024, 128, 'bn_pool_nomask'), (384, 512, 256, 128, 'bn')])
def test_pointmlp_fwd(hip_lib, gemm_arithmetic, M, K, N, rpf, mode):
    r = np.random.RandomState(hash((M, K, N)) % 1000)
    B, T = M // rpf, M // 128
    ldx = 4 if K <= 4 else K
    x = r.normal(size=(M, ldx)).astype(np.float32)
    w = (r.normal(size=(K, N)) / np.sqrt(K)).astype(np.float32)
    bias = r.normal(size=N).astype(np.float32) * 0.1
    sc = (0.5 + r.uniform(size=K)).astype(np.float32)
    sc[::3] *= -1
    sh = r.normal(size=K).astype(np.float32) * 0.2
    sub = r.normal(size=(B, 3)).astype(np.float32)
    rb = r.normal(size=(B, N)).astype(np.float32)
    mask = (r.uniform(size=M) < 0.4).astype(np.float32)
    mask[:128] = 0                                                   # one tile with no kept row

    def make(dev):
        t = {k: _mk(dev, v) for k, v in dict(x=x, w=w, bias=bias, sc=sc, sh=sh, sub=sub, rb=rb, mask=mask).items()}
        o = dict(y=torch.zeros(M, N, device=dev), psum=torch.zeros(T, N, device=dev), psumsq=torch.zeros(T, N, device=dev))
        a = abi.PointMlpFwdArgs()
        a.a = abi.ActSrc(fptr(t['x']), ldx, 0, fptr(t['sc'] if 'bn' in mode else None), fptr(t['sh'] if 'bn' in mode else None),
                         int('bn' in mode), fptr(t['sub'] if mode == 'sub' else None), 3)
        a.w, a.bias, a.y, a.psum, a.psumsq = fptr(t['w']), fptr(t['bias']), fptr(o['y']), fptr(o['psum']), fptr(o['psumsq'])
        if 'rowbias' in mode:
            a.rowbias = fptr(t['rb'])
        if 'pool' in mode:
            o.update(pmax=torch.zeros(T, N, device=dev), pmin=torch.zeros(T, N, device=dev),
                     pamax=torch.zeros(T, N, dtype=torch.int32, device=dev), pamin=torch.zeros(T, N, dtype=torch.int32, device=dev))
            a.pmax, a.pmin, a.pamax, a.pamin = fptr(o['pmax']), fptr(o['pmin']), iptr(o['pamax']), iptr(o['pamin'])
            if 'nomask' not in mode:
                a.rowmask = fptr(t['mask'])
        a.M, a.K, a.N, a.rows_per_frustum = M, K, N, rpf
        a._keep = (t, o)
        return a, o

    c, g = _run_both(hip_lib, make, 't3d_pointmlp_fwd')
    _close(c['y'], g['y'], 2e-5, 2e-5, 'y')
    _close(c['psum'], g['psum'], 1e-4, 2e-3, 'psum')
    _close(c['psumsq'], g['psumsq'], 1e-4, 2e-3, 'psumsq')
    if 'pool' in mode:
        _close(c['pmax'][c['pamax'] >= 0], g['pmax'][g['pamax'] >= 0], 2e-5, 2e-5, 'pmax')
        _close(c['pmin'][c['pamin'] >= 0], g['pmin'][g['pamin'] >= 0], 2e-5, 2e-5, 'pmin')
        assert ((c['pamax'] >= 0) == (g['pamax'] >= 0)).all()
        # arg indices may differ only where two rows tie to within rounding: check the value at the GPU's index
        yv = g['y'].reshape(T, 128, N)
        sel = g['pamax'].long() % 128
        got = torch.gather(yv, 1, sel.clamp(min=0)[:, None, :])[:, 0]
        assert torch.equal(got[g['pamax'] >= 0], g['pmax'][g['pamax'] >= 0])
        assert (c['pamax'] == g['pamax']).float().mean() > 0.999


@pytest.mark.parametrize('M,K,N,rpf,mode', [(512, 64, 128, 256, 'dense'), (256, 128, 256, 128, 'pooled'),
                                            (256, 64, 512, 128, 'raw_addin'), (384, 512, 256, 128, 'dense'),
                                            (256, 256, 128, 128, 'dense_addin')])
def test_pointmlp_dgrad(hip_lib, gemm_arithmetic, M, K, N, rpf, mode):
    r = np.random.RandomState(hash((M, K, N, 1)) % 1000)
    B, T = M // rpf, M // 128
    dz = (r.normal(size=(M, N)) * 1e-2).astype(np.float32)
    y = r.normal(size=(M, N)).astype(np.float32)
    coef = r.normal(size=(3, N)).astype(np.float32)
    coef[2] *= 1e-3
    w = (r.normal(size=(K, N)) / np.sqrt(N)).astype(np.float32)
    argidx = r.randint(-1, rpf, size=(B, N)).astype(np.int32)
    dpool = r.normal(size=(B, N)).astype(np.float32)
    add = (r.normal(size=(M, K)) * 1e-2).astype(np.float32)
    py = r.normal(size=(M, K)).astype(np.float32)
    psc = (0.5 + r.uniform(size=K)).astype(np.float32)
    psh = (r.normal(size=K) * 0.3).astype(np.float32)

    def make(dev):
        t = {k: _mk(dev, v) for k, v in dict(dz=dz, y=y, coef=coef, w=w, argidx=argidx, dpool=dpool, add=add, py=py,
                                             psc=psc, psh=psh).items()}
        o = dict(out=torch.zeros(M, K, device=dev), s1=torch.zeros(T, K, device=dev), s2=torch.zeros(T, K, device=dev))
        a = abi.PointMlpDgradArgs()
        if mode == 'pooled':
            a.dy = abi.DySrc(fptr(None), fptr(t['y']), fptr(t['coef']), iptr(t['argidx']), fptr(t['dpool']))
        else:
            a.dy = abi.DySrc(fptr(t['dz']), fptr(t['y']), fptr(t['coef']), iptr(None), fptr(None))
        a.w, a.out = fptr(t['w']), fptr(o['out'])
        if 'addin' in mode:
            a.add_in = fptr(t['add'])
        if not mode.startswith('raw'):
            a.prev_y, a.prev_scale, a.prev_shift = fptr(t['py']), fptr(t['psc']), fptr(t['psh'])
            a.psum_dz, a.psum_dzy = fptr(o['s1']), fptr(o['s2'])
        a.M, a.K, a.N, a.rows_per_frustum = M, K, N, rpf
        a._keep = (t, o)
        return a, o

    c, g = _run_both(hip_lib, make, 't3d_pointmlp_dgrad')
    scale = float(c['out'].abs().max())
    _close(c['out'], g['out'], 1e-4, 2e-5 * scale, 'out')
    if not mode.startswith('raw'):
        _close(c['s1'], g['s1'], 1e-3, 1e-3 * scale, 'psum_dz')
        _close(c['s2'], g['s2'], 1e-3, 2e-3 * scale, 'psum_dzy')


@pytest.mark.parametrize('M,K,N,rpf,rps,mode', [(512, 64, 64, 256, 128, 'dense'), (512, 3, 128, 128, 256, 'sub'),
                                                (256, 128, 1024, 128, 64, 'pooled'), (512, 256, 128, 256, 512, 'dense'),
                                                (256, 4, 64, 128, 32, 'raw'), (256, 512, 256, 128, 128, 'dense')])
def test_pointmlp_wgrad(hip_lib, gemm_arithmetic, M, K, N, rpf, rps, mode):
    r = np.random.RandomState(hash((M, K, N, 2)) % 1000)
    B, S = M // rpf, M // rps
    ldx = 4 if K <= 4 else K
    x = r.normal(size=(M, ldx)).astype(np.float32)
    sc = (0.5 + r.uniform(size=K)).astype(np.float32)
    sh = r.normal(size=K).astype(np.float32) * 0.2
    sub = r.normal(size=(B, 3)).astype(np.float32)
    dz = (r.normal(size=(M, N)) * 1e-2).astype(np.float32)
    y = r.normal(size=(M, N)).astype(np.float32)
    coef = r.normal(size=(3, N)).astype(np.float32)
    coef[2] *= 1e-3
    argidx = r.randint(-1, rpf, size=(B, N)).astype(np.int32)
    dpool = r.normal(size=(B, N)).astype(np.float32)

    def make(dev):
        t = {k: _mk(dev, v) for k, v in dict(x=x, sc=sc, sh=sh, sub=sub, dz=dz, y=y, coef=coef, argidx=argidx, dpool=dpool).items()}
        o = dict(slabs=torch.zeros(S, K, N, device=dev))
        a = abi.PointMlpWgradArgs()
        bn = mode in ('dense', 'pooled')
        a.a = abi.ActSrc(fptr(t['x']), ldx, 0, fptr(t['sc'] if bn else None), fptr(t['sh'] if bn else None), int(bn),
                         fptr(t['sub'] if mode == 'sub' else None), 3)
        if mode == 'pooled':
            a.dy = abi.DySrc(fptr(None), fptr(t['y']), fptr(t['coef']), iptr(t['argidx']), fptr(t['dpool']))
        else:
            a.dy = abi.DySrc(fptr(t['dz']), fptr(t['y']), fptr(t['coef']), iptr(None), fptr(None))
        a.slabs = fptr(o['slabs'])
        a.M, a.K, a.N, a.rows_per_frustum, a.rows_per_split = M, K, N, rpf, rps
        a._keep = (t, o)
        return a, o

    c, g = _run_both(hip_lib, make, 't3d_pointmlp_wgrad')
    scale = float(c['slabs'].abs().max())
    _close(c['slabs'], g['slabs'], 1e-4, 3e-5 * scale, 'slabs')


def test_bn_finalizers_pool_colsum(hip_lib):
    r = np.random.RandomState(5)
    B, tpf, N = 4, 2, 192
    T, M = B * tpf, B * tpf * 128
    psum = r.normal(size=(T, N)).astype(np.float32) * 128
    psumsq = (np.abs(r.normal(size=(T, N))) * 128 + psum ** 2 / 128).astype(np.float32)
    gamma = (r.normal(size=N)).astype(np.float32)
    beta = r.normal(size=N).astype(np.float32)
    mm, mv = r.normal(size=N).astype(np.float32), (0.5 + r.uniform(size=N)).astype(np.float32)
    for training in (1, 0):
        def make(dev):
            t = {k: _mk(dev, v) for k, v in dict(psum=psum, psumsq=psumsq, gamma=gamma, beta=beta, mm=mm.copy(), mv=mv.copy(),
                                                 decay=np.array([0.7], np.float32)).items()}
            o = dict(scale=torch.zeros(N, device=dev), shift=torch.zeros(N, device=dev), mean=torch.zeros(N, device=dev),
                     invstd=torch.zeros(N, device=dev), mm=t['mm'], mv=t['mv'])
            a = abi.BnFwdFinalizeArgs(fptr(t['psum']), fptr(t['psumsq']), T, M, N, fptr(t['gamma']), fptr(t['beta']),
                                      fptr(t['mm']), fptr(t['mv']), fptr(t['decay']), 1e-3, training, 1,
                                      fptr(o['scale']), fptr(o['shift']), fptr(o['mean']), fptr(o['invstd']))
            a._keep = (t, o)
            return a, o
        c, g = _run_both(hip_lib, make, 't3d_bn_fwd_finalize')
        for k in c:
            _close(c[k], g[k], 1e-5, 1e-6, 'bn_fwd ' + k)

    # pool finalize
    scale = r.normal(size=N).astype(np.float32)
    shift = r.normal(size=N).astype(np.float32)
    pmax = r.normal(size=(T, N)).astype(np.float32)
    pmin = (pmax - np.abs(r.normal(size=(T, N)))).astype(np.float32)
    pamax = r.randint(0, 128, size=(T, N)).astype(np.int32) + (np.arange(T) % tpf)[:, None].astype(np.int32) * 128
    pamin = r.randint(0, 128, size=(T, N)).astype(np.int32) + (np.arange(T) % tpf)[:, None].astype(np.int32) * 128
    dead = r.uniform(size=(T, N)) < 0.3
    pamax[dead] = -1
    pamin[dead] = -1

    def make_pool(dev):
        t = {k: _mk(dev, v) for k, v in dict(scale=scale, shift=shift, pmax=pmax, pmin=pmin, pamax=pamax, pamin=pamin).items()}
        o = dict(pooled=torch.zeros(B, N, device=dev), argidx=torch.zeros(B, N, dtype=torch.int32, device=dev),
                 ysel=torch.zeros(B, N, device=dev))
        a = abi.PoolFinalizeArgs(fptr(t['scale']), fptr(t['shift']), fptr(t['pmax']), fptr(t['pmin']), iptr(t['pamax']),
                                 iptr(t['pamin']), B, N, tpf, fptr(o['pooled']), N, iptr(o['argidx']), fptr(o['ysel']))
        a._keep = (t, o)
        return a, o
    c, g = _run_both(hip_lib, make_pool, 't3d_pool_finalize')
    _close(c['pooled'], g['pooled'], 1e-6, 1e-6, 'pooled')
    assert torch.equal(c['argidx'], g['argidx'])
    _close(c['ysel'], g['ysel'], 0, 0, 'ysel')

    # bn backward finalize (dense, pooled, frozen) + colsum
    s1 = r.normal(size=(T, N)).astype(np.float32)
    s2 = r.normal(size=(T, N)).astype(np.float32)
    mean, invstd = r.normal(size=N).astype(np.float32), (0.5 + r.uniform(size=N)).astype(np.float32)
    dpin = r.normal(size=(B, N)).astype(np.float32)
    pooled = np.maximum(r.normal(size=(B, N)), 0).astype(np.float32)
    ysel = r.normal(size=(B, N)).astype(np.float32)
    for form in ('dense', 'pooled', 'frozen'):
        def make_b(dev):
            t = {k: _mk(dev, v) for k, v in dict(s1=s1, s2=s2, mean=mean, invstd=invstd, gamma=gamma, scale=scale, dpin=dpin,
                                                 pooled=pooled, ysel=ysel).items()}
            o = dict(coef=torch.zeros(3, N, device=dev), dgamma=torch.zeros(N, device=dev), dbeta=torch.zeros(N, device=dev),
                     dpool=torch.zeros(B, N, device=dev))
            a = abi.BnBwdFinalizeArgs()
            if form == 'pooled':
                a.dpool_in, a.ld_dpool_in, a.pooled, a.ld_pooled = fptr(t['dpin']), N, fptr(t['pooled']), N
                a.ysel, a.dpool, a.B = fptr(t['ysel']), fptr(o['dpool']), B
            else:
                a.psum_dz, a.psum_dzy, a.n_tiles = fptr(t['s1']), fptr(t['s2']), T
            a.count, a.N = M, N
            a.gamma, a.mean, a.invstd, a.scale = fptr(t['gamma']), fptr(t['mean']), fptr(t['invstd']), fptr(t['scale'])
            a.frozen = int(form == 'frozen')
            a.dgamma, a.dbeta, a.coef = fptr(o['dgamma']), fptr(o['dbeta']), fptr(o['coef'])
            a._keep = (t, o)
            return a, o
        c, g = _run_both(hip_lib, make_b, 't3d_bn_bwd_finalize')
        for k in c:
            _close(c[k], g[k], 1e-5, 1e-6, 'bn_bwd %s %s' % (form, k))

    coef = r.normal(size=(3, N)).astype(np.float32)

    def make_c(dev):
        t = {k: _mk(dev, v) for k, v in dict(s1=s1, psum=psum, coef=coef).items()}
        o = dict(out=torch.zeros(B, N, device=dev))
        a = abi.DyColsumArgs(fptr(t['s1']), fptr(t['psum']), fptr(t['coef']), B, N, tpf, tpf * 128, -1.0, fptr(o['out']))
        a._keep = (t, o)
        return a, o
    c, g = _run_both(hip_lib, make_c, 't3d_dy_colsum')
    _close(c['out'], g['out'], 1e-5, 1e-4, 'colsum')


@pytest.mark.parametrize('B,K,K2,N,bn,act,drop,train', [
    (32, 256, 0, 256, True, 'relu', False, 1), (32, 512, 10, 512, True, 'leaky_relu', True, 1),
    (32, 256, 0, 67, False, None, False, 1), (8, 128, 0, 3, False, None, False, 1),
    (64, 512, 0, 256, True, 'tanh', True, 1), (32, 1024, 0, 512, False, None, False, 1),
    (32, 512, 0, 512, True, 'relu', False, 0), (128, 256, 0, 9, False, None, False, 1)])
def test_fc_fwd_bwd_dinput(hip_lib, B, K, K2, N, bn, act, drop, train):
    r = np.random.RandomState(hash((B, K, N)) % 1000)
    x = r.normal(size=(B, K)).astype(np.float32)
    x2 = r.normal(size=(B, max(K2, 1))).astype(np.float32)
    w = (r.normal(size=(K + K2, N)) / np.sqrt(K)).astype(np.float32)
    bias = (r.normal(size=N) * 0.1).astype(np.float32)
    gamma, beta = (0.5 + r.uniform(size=N)).astype(np.float32), (r.normal(size=N) * 0.1).astype(np.float32)
    mm, mv = r.normal(size=N).astype(np.float32) * 0.1, (0.5 + r.uniform(size=N)).astype(np.float32)
    mask = (r.uniform(size=(B, N)) < 0.7).astype(np.float32)
    add = r.normal(size=(B, 3)).astype(np.float32)
    dout = r.normal(size=(B, N)).astype(np.float32)
    Nn = 96
    dyn = r.normal(size=(B, Nn)).astype(np.float32)
    wn = (r.normal(size=(N, Nn)) / np.sqrt(N)).astype(np.float32)
    saved = {}

    def make_f(dev):
        t = {k: _mk(dev, v) for k, v in dict(x=x, x2=x2, w=w, bias=bias, gamma=gamma, beta=beta, mm=mm.copy(), mv=mv.copy(),
                                             mask=mask, add=add, decay=np.array([0.6], np.float32)).items()}
        o = dict(y=torch.zeros(B, N, device=dev), out=torch.zeros(B, N, device=dev), mean=torch.zeros(N, device=dev),
                 invstd=torch.ones(N, device=dev), mm=t['mm'], mv=t['mv'])
        a = abi.FcFwdArgs()
        a.in_, a.ld_in, a.K, a.in2, a.ld_in2, a.K2 = fptr(t['x']), K, K, fptr(t['x2'] if K2 else None), max(K2, 1), K2
        a.w, a.bias = fptr(t['w']), fptr(t['bias'])
        if bn:
            a.gamma, a.beta, a.moving_mean, a.moving_var = fptr(t['gamma']), fptr(t['beta']), fptr(t['mm']), fptr(t['mv'])
            a.mean, a.invstd = fptr(o['mean']), fptr(o['invstd'])
        a.decay, a.eps, a.is_training, a.unbiased_ema = fptr(t['decay']), 1e-3, train, 1
        a.act, a.leaky_alpha = abi.ACT_BY_NAME[act], 0.2
        if drop:
            a.drop_mask, a.keep_prob = fptr(t['mask']), 0.7
        if N >= 3 and not bn:
            a.add_in, a.ld_add, a.add_n = fptr(t['add']), 3, 3
        a.y, a.out, a.ld_out, a.B, a.N = fptr(o['y']), fptr(o['out']), N, B, N
        a._keep = (t, o)
        saved[dev.type] = (t, o)
        return a, o
    c, g = _run_both(hip_lib, make_f, 't3d_fc_fwd')
    for k in c:
        _close(c[k], g[k], 2e-4, 2e-4, 'fc_fwd ' + k)

    for src in ('dout', 'next'):
        def make_b(dev):
            tf, of = saved[dev.type]
            t = {k: _mk(dev, v) for k, v in dict(dout=dout, dyn=dyn, wn=wn).items()}
            yc = _mk(dev, saved['cpu'][1]['y'].numpy())          # identical saved activations on both sides
            mc, ic = _mk(dev, saved['cpu'][1]['mean'].numpy()), _mk(dev, saved['cpu'][1]['invstd'].numpy())
            o = dict(dy=torch.zeros(B, N, device=dev), dw=torch.zeros(K + K2, N, device=dev), db=torch.zeros(N, device=dev),
                     dgamma=torch.zeros(N, device=dev), dbeta=torch.zeros(N, device=dev))
            a = abi.FcBwdArgs()
            if src == 'dout':
                a.dout, a.ld_dout = fptr(t['dout']), N
            else:
                a.dy_next, a.w_next, a.N_next = fptr(t['dyn']), fptr(t['wn']), Nn
            a.in_, a.ld_in, a.K, a.in2, a.ld_in2, a.K2 = fptr(tf['x']), K, K, fptr(tf['x2'] if K2 else None), max(K2, 1), K2
            a.y, a.out, a.ld_out = fptr(yc), fptr(of['out']), N
            if bn:
                a.gamma, a.beta, a.mean, a.invstd = fptr(tf['gamma']), fptr(tf['beta']), fptr(mc), fptr(ic)
            a.bn_training, a.act, a.leaky_alpha = train, abi.ACT_BY_NAME[act], 0.2
            if drop:
                a.drop_mask, a.keep_prob = fptr(tf['mask']), 0.7
            a.dy, a.dw, a.dbias, a.dgamma, a.dbeta = fptr(o['dy']), fptr(o['dw']), fptr(o['db']), fptr(o['dgamma']), fptr(o['dbeta'])
            a.B, a.N = B, N
            a._keep = (t, o, yc, mc, ic)
            return a, o
        c, g = _run_both(hip_lib, make_b, 't3d_fc_bwd')
        for k in c:
            sc = max(float(c[k].abs().max()), 1e-6)
            _close(c[k], g[k], 5e-4, 2e-4 * sc, 'fc_bwd %s %s' % (src, k))

    def make_d(dev):
        t = {k: _mk(dev, v) for k, v in dict(dy=dout, w=w, add=r.normal(size=(B, K)).astype(np.float32) * 0 + 0.5).items()}
        o = dict(din=torch.zeros(B, K, device=dev))
        a = abi.FcDinputArgs(fptr(t['dy']), N, fptr(t['w']), fptr(t['add']), K, -1.0, fptr(o['din']), K, B, K)
        a._keep = (t, o)
        return a, o
    c, g = _run_both(hip_lib, make_d, 't3d_fc_dinput')
    _close(c['din'], g['din'], 2e-4, 2e-4 * float(c['din'].abs().max()), 'fc_dinput')


@pytest.mark.parametrize('mode', ['train', 'infer', 'train_nodrop'])
def test_seg_head_and_finalize(hip_lib, mode):
    r = np.random.RandomState(9)
    B, rpf, K, Cc = 3, 256, 128, 4
    M, T, tpf = B * rpf, B * rpf // 128, rpf // 128
    y = r.normal(size=(M, K)).astype(np.float32)
    sc, sh = (0.5 + r.uniform(size=K)).astype(np.float32), (r.normal(size=K) * 0.3).astype(np.float32)
    dm = (r.uniform(size=(M, K)) < 0.5).astype(np.float32)
    w, b = (r.normal(size=(K, 2)) * 0.2).astype(np.float32), np.array([0.1, -0.2], np.float32)
    lab = (r.uniform(size=M) < 0.3).astype(np.int32)
    is2d = np.array([0, 1, 0], np.int32)
    pc = r.normal(size=(M, Cc)).astype(np.float32)
    saved = {}

    def make(dev):
        t = {k: _mk(dev, v) for k, v in dict(y=y, sc=sc, sh=sh, dm=dm, w=w, b=b, lab=lab, is2d=is2d, pc=pc).items()}
        o = dict(logits=torch.zeros(M, 2, device=dev), mask=torch.zeros(M, device=dev), part=torch.zeros(T, 8, device=dev))
        a = abi.SegHeadArgs()
        a.y, a.scale, a.shift = fptr(t['y']), fptr(t['sc']), fptr(t['sh'])
        if mode == 'train':
            a.drop_mask, a.keep_prob = fptr(t['dm']), 0.5
        a.w, a.bias, a.pc, a.ld_pc, a.ce_weight = fptr(t['w']), fptr(t['b']), fptr(t['pc']), Cc, 1.0
        if mode != 'infer':
            a.labels, a.is_data_2D = iptr(t['lab']), iptr(t['is2d'])
            o.update(dz=torch.zeros(M, K, device=dev), s1=torch.zeros(T, K, device=dev), s2=torch.zeros(T, K, device=dev),
                     dwp=torch.zeros(T, K, 2, device=dev))
            a.dz, a.psum_dz, a.psum_dzy, a.dw_part = fptr(o['dz']), fptr(o['s1']), fptr(o['s2']), fptr(o['dwp'])
        a.logits, a.mask, a.part = fptr(o['logits']), fptr(o['mask']), fptr(o['part'])
        a.M, a.K, a.rows_per_frustum, a.B = M, K, rpf, B
        a._keep = (t, o)
        saved[dev.type] = o
        return a, o
    c, g = _run_both(hip_lib, make, 't3d_seg_head')
    _close(c['logits'], g['logits'], 1e-5, 2e-5, 'logits')
    # the hard mask may legitimately differ only where |l0-l1| is at rounding level
    diff = (c['mask'] != g['mask'])
    assert (c['logits'][diff, 0] - c['logits'][diff, 1]).abs().max().item() < 1e-4 if diff.any() else True
    if not diff.any():
        _close(c['part'], g['part'], 1e-4, 1e-3, 'part')
    if mode != 'infer':
        _close(c['dz'], g['dz'], 1e-4, 1e-8, 'dz')
        _close(c['s1'], g['s1'], 1e-3, 1e-6, 'psum_dz')
        _close(c['s2'], g['s2'], 1e-3, 1e-6, 'psum_dzy')
        _close(c['dwp'], g['dwp'], 1e-3, 1e-6, 'dw_part')

    def make_f(dev):
        o_prev = saved[dev.type]
        part = _mk(dev, saved['cpu']['part'].numpy())
        o = dict(mean=torch.zeros(B, 3, device=dev), seg=torch.zeros(B, device=dev), dw=torch.zeros(K, 2, device=dev),
                 db=torch.zeros(2, device=dev), nc=torch.zeros(1, device=dev))
        a = abi.SegFinalizeArgs()
        a.part, a.B, a.tiles_per_frustum, a.rows_per_frustum, a.K = fptr(part), B, tpf, rpf, K
        a.mask_xyz_mean, a.seg_loss, a.n_correct = fptr(o['mean']), fptr(o['seg']), fptr(o['nc'])
        if mode != 'infer':
            dwp = _mk(dev, saved['cpu']['dwp'].numpy())
            a.dw_part, a.dw, a.dbias = fptr(dwp), fptr(o['dw']), fptr(o['db'])
            a._keep2 = dwp
        a._keep = (part, o)
        return a, o
    c, g = _run_both(hip_lib, make_f, 't3d_seg_finalize')
    for k in c:
        _close(c[k], g[k], 1e-5, 1e-6, 'seg_finalize ' + k)


@pytest.mark.parametrize('norm3d', [0, 1])
def test_strong_loss(hip_lib, norm3d):
    r = np.random.RandomState(21)
    B = 32
    box = r.normal(size=(B, 67)).astype(np.float32) * 0.5
    s1 = r.normal(size=(B, 3)).astype(np.float32)
    seg = np.abs(r.normal(size=B)).astype(np.float32)
    yc = (s1 + r.normal(size=(B, 3)) * 1.5).astype(np.float32)
    yoc, ydc = r.randint(0, 12, size=B).astype(np.int32), r.randint(0, 10, size=B).astype(np.int32)
    yor = r.uniform(-0.26, 0.26, size=B).astype(np.float32)
    ydr = (r.normal(size=(B, 3)) * 0.1).astype(np.float32)
    is2d = (r.uniform(size=B) < 0.3).astype(np.int32)

    def make(dev):
        t = {k: _mk(dev, v) for k, v in dict(box=box, s1=s1, seg=seg, yc=yc, yoc=yoc, ydc=ydc, yor=yor, ydr=ydr, is2d=is2d).items()}
        o = dict(dbox=torch.zeros(B, 67, device=dev), ds1=torch.zeros(B, 3, device=dev), terms=torch.zeros(B, 8, device=dev),
                 tot=torch.zeros(B, device=dev), loss=torch.zeros(1, device=dev), center=torch.zeros(B, 3, device=dev),
                 dims=torch.zeros(B, 3, device=dev), theta=torch.zeros(B, device=dev))
        a = abi.StrongLossArgs()
        a.box, a.ld_box, a.stage1_center, a.seg_loss = fptr(t['box']), 67, fptr(t['s1']), fptr(t['seg'])
        a.y_center, a.y_orient_cls, a.y_orient_reg = fptr(t['yc']), iptr(t['yoc']), fptr(t['yor'])
        a.y_dims_cls, a.y_dims_reg, a.is_data_2D = iptr(t['ydc']), fptr(t['ydr']), iptr(t['is2d'])
        a.wts = abi.StrongWeights(1.0, 1.0, 20.0, 1.0, 20.0, 1.0, 1.0, 0.1, 1.0)
        a.normalize_by_3d_count = norm3d
        a.dbox, a.dstage1, a.terms, a.total_losses, a.loss = fptr(o['dbox']), fptr(o['ds1']), fptr(o['terms']), fptr(o['tot']), fptr(o['loss'])
        a.center, a.reg_dims, a.reg_theta, a.B = fptr(o['center']), fptr(o['dims']), fptr(o['theta']), B
        a._keep = (t, o)
        return a, o
    c, g = _run_both(hip_lib, make, 't3d_strong_loss')
    for k in c:
        sc = max(float(c[k].abs().max()), 1e-6)
        _close(c[k], g[k], 1e-4, 2e-5 * sc, 'strong_loss ' + k)


def test_reduce_slabs_adam_schedule_dropout(hip_lib):
    r = np.random.RandomState(4)
    fake = FakeLib()
    s = torch.cuda.current_stream().cuda_stream
    # reduce slabs
    sizes = [(5, 640), (3, 67), (16, 4096)]
    slab = np.concatenate([r.normal(size=ns * ne) for ns, ne in sizes]).astype(np.float32)
    table = (abi.SlabDesc * 3)()
    so, go = 0, 8
    for i, (ns, ne) in enumerate(sizes):
        table[i] = abi.SlabDesc(so, go, ne, ns)
        so += ns * ne
        go += ne + 4
    gc, gg = torch.zeros(go), torch.zeros(go, device='cuda')
    sc_, sg = torch.as_tensor(slab), torch.as_tensor(slab).cuda()
    fake.t3d_reduce_slabs(fptr(sc_), fptr(gc), table, 3, 4096, None)
    tab_dev = torch.as_tensor(np.frombuffer(bytes(table), dtype=np.uint8).copy()).cuda()
    assert hip_lib.t3d_reduce_slabs(fptr(sg), fptr(gg), C.cast(C.c_void_p(tab_dev.data_ptr()), C.POINTER(abi.SlabDesc)), 3, 4096, s) == 0
    torch.cuda.synchronize()
    _close(gc, gg.cpu(), 1e-5, 1e-5, 'reduce_slabs')

    # schedule + adam, three steps, incl. a step across the lr staircase
    sched = abi.Schedule(1e-3, 0.5, 800000.0, 0.5, 0.5, 800000.0, 0.99, 0.9, 0.999, 32)
    n = 10007
    w0, g0 = r.normal(size=n).astype(np.float32), (r.normal(size=n) * 1e-2).astype(np.float32)
    for start in (0.0, 24999.0):
        hc, hg = torch.tensor([start, 0, 0, 0]), torch.tensor([start, 0, 0, 0]).cuda()
        ws = [torch.as_tensor(w0.copy()), torch.as_tensor(w0.copy()).cuda()]
        ms = [torch.zeros(n), torch.zeros(n).cuda()]
        vs = [torch.zeros(n), torch.zeros(n).cuda()]
        gs = [torch.as_tensor(g0), torch.as_tensor(g0).cuda()]
        for it in range(3):
            fake.t3d_schedule_step(fptr(hc), C.byref(sched), None)
            assert hip_lib.t3d_schedule_step(fptr(hg), C.byref(sched), s) == 0
            fake.t3d_adam_tf_step(fptr(ws[0]), fptr(gs[0]), fptr(ms[0]), fptr(vs[0]), n, fptr(hc), 0.9, 0.999, 1e-8, 0.5, None)
            assert hip_lib.t3d_adam_tf_step(fptr(ws[1]), fptr(gs[1]), fptr(ms[1]), fptr(vs[1]), n, fptr(hg), 0.9, 0.999, 1e-8, 0.5, s) == 0
        torch.cuda.synchronize()
        _close(hc, hg.cpu(), 1e-6, 1e-9, 'hyper')
        _close(ws[0], ws[1].cpu(), 1e-6, 1e-7, 'adam w')
        _close(vs[0], vs[1].cpu(), 1e-4, 1e-12, 'adam v')
    assert abs(float(hc[1]) - 5e-4) < 1e-9 and abs(float(hc[2]) - 0.75) < 1e-7      # 25000*32 >= 800000

    # tf.train.MomentumOptimizer (--optimizer momentum, train_semisup.py:226-228): three steps against the oracle's restatement in fp64
    from oracle import ref_torch as R
    hg = torch.tensor([0.0, 0, 0, 0]).cuda()
    wg, ag, gg2 = torch.as_tensor(w0.copy()).cuda(), torch.zeros(n).cuda(), torch.as_tensor(g0).cuda()
    P, acc = {'w': torch.as_tensor(w0.astype(np.float64))}, {'w': torch.zeros(n, dtype=torch.float64)}
    for it in range(3):
        assert hip_lib.t3d_schedule_step(fptr(hg), C.byref(sched), s) == 0
        assert hip_lib.t3d_momentum_step(fptr(wg), fptr(gg2), fptr(ag), n, fptr(hg), 0.9, 0.5, s) == 0
        R.momentum_tf_step(P, {'w': torch.as_tensor(g0.astype(np.float64)) * 0.5}, acc, 1e-3, 0.9)
    torch.cuda.synchronize()
    _close(P['w'].float(), wg.cpu(), 1e-6, 1e-7, 'momentum w')
    _close(acc['w'].float(), ag.cpu(), 1e-6, 1e-9, 'momentum accumulator')
    assert hip_lib.t3d_momentum_step(None, fptr(gg2), fptr(ag), n, fptr(hg), 0.9, 0.5, s) == -1      # T3D_ERR_ARG

    # dropout mask: 0/1 valued, keep fraction, fresh per step
    n = 1 << 20
    m1, m2 = torch.zeros(n, device='cuda'), torch.zeros(n, device='cuda')
    h = torch.tensor([3.0, 0, 0, 0]).cuda()
    assert hip_lib.t3d_dropout_mask(fptr(m1), n, 0.7, 1234, fptr(h), s) == 0
    h2 = torch.tensor([4.0, 0, 0, 0]).cuda()
    assert hip_lib.t3d_dropout_mask(fptr(m2), n, 0.7, 1234, fptr(h2), s) == 0
    torch.cuda.synchronize()
    assert set(m1.unique().tolist()) == {0.0, 1.0}
    assert abs(float(m1.mean()) - 0.7) < 5e-3 and abs(float(m2.mean()) - 0.7) < 5e-3
    assert abs(float((m1 == m2).float().mean()) - (0.49 + 0.09)) < 5e-3


def test_bad_arguments_are_rejected(hip_lib):
    a = abi.PointMlpFwdArgs()
    assert hip_lib.t3d_pointmlp_fwd(C.byref(a), None) == -1                     # T3D_ERR_ARG
    x = torch.zeros(130, 64, device='cuda')
    o = torch.zeros(130, 64, device='cuda')
    a.a = abi.ActSrc(fptr(x), 64, 0, fptr(None), fptr(None), 0, fptr(None), 0)
    a.w, a.y, a.psum, a.psumsq = fptr(x), fptr(o), fptr(o), fptr(o)
    a.M, a.K, a.N, a.rows_per_frustum = 130, 64, 64, 130
    assert hip_lib.t3d_pointmlp_fwd(C.byref(a), None) == -2                     # T3D_ERR_SHAPE


@pytest.mark.parametrize('label_form', [0, 1])
def test_boxpc_rep_fwd_bwd_and_loss(hip_lib, label_form):
    r = np.random.RandomState(31)
    B, rpf, Cc = 3, 256, 4
    M, ld = B * rpf, 12
    pc = r.normal(size=(M, Cc)).astype(np.float32)
    center = r.normal(size=(B, 3)).astype(np.float32)
    dims = (r.normal(size=(B, 3)) * 0.1 + (0 if label_form else 1.0)).astype(np.float32)
    theta = r.uniform(-0.3, 0.3, size=B).astype(np.float32) + (0 if label_form else 1.0)
    ydc, yoc = r.randint(0, 10, size=B).astype(np.int32), r.randint(0, 12, size=B).astype(np.int32)
    saved = {}

    def make(dev):
        t = {k: _mk(dev, v) for k, v in dict(pc=pc, center=center, dims=dims, theta=theta, ydc=ydc, yoc=yoc).items()}
        o = dict(rep=torch.full((M, ld), 7.0, device=dev), box=torch.zeros(B, 7, device=dev))
        a = abi.BoxPcRepArgs(fptr(t['pc']), Cc, Cc, fptr(t['center']), fptr(t['dims']), fptr(t['theta']),
                             iptr(t['ydc'] if label_form else None), iptr(t['yoc'] if label_form else None), fptr(o['rep']), ld,
                             fptr(o['box']), M, rpf)
        a._keep = (t, o)
        saved[dev.type] = (t, o)
        return a, o
    c, g = _run_both(hip_lib, make, 't3d_boxpc_rep')
    _close(c['rep'], g['rep'], 1e-5, 1e-5, 'rep')
    _close(c['box'], g['box'], 1e-6, 1e-6, 'box')

    drep = (r.normal(size=(M, 16)) * 1e-2).astype(np.float32)

    def make_b(dev):
        t, o = saved[dev.type]
        d = _mk(dev, drep)
        box = _mk(dev, saved['cpu'][1]['box'].numpy())
        out = dict(dbox=torch.zeros(B, 7, device=dev))
        a = abi.BoxPcRepBwdArgs(fptr(t['pc']), Cc, fptr(box), fptr(d), 16, 4, fptr(out['dbox']), B, rpf)
        a._keep = (d, box, out)
        return a, out
    c, g = _run_both(hip_lib, make_b, 't3d_boxpc_rep_bwd')
    _close(c['dbox'], g['dbox'], 1e-4, 1e-5, 'dbox')

    Bl = 32
    out9 = r.normal(size=(Bl, 9)).astype(np.float32)
    iou = r.uniform(size=Bl).astype(np.float32)
    dc, ds, da = [(r.normal(size=s) * 0.7).astype(np.float32) for s in ((Bl, 3), (Bl, 3), (Bl,))]
    for conf, gt in ((0, 0), (1, 0), (0, 1)):
        def make_l(dev):
            t = {k: _mk(dev, v) for k, v in dict(o=out9, iou=iou, dc=dc, ds=ds, da=da).items()}
            o = dict(dout=torch.zeros(Bl, 9, device=dev), terms=torch.zeros(Bl, 4, device=dev), loss=torch.zeros(1, device=dev))
            a = abi.BoxPcLossArgs(fptr(t['o']), fptr(t['iou']), fptr(t['dc']), fptr(t['ds']), fptr(t['da']), 0.7, 1.0, 4.0, 0.34,
                                  0.33, 0.33, conf, gt, fptr(o['dout']), fptr(o['terms']), fptr(o['loss']), Bl)
            a._keep = (t, o)
            return a, o
        c, g = _run_both(hip_lib, make_l, 't3d_boxpc_loss')
        for k in c:
            _close(c[k], g[k], 1e-5, 1e-6, 'boxpc_loss ' + k)


def test_stage_c_glue_kernels(hip_lib):
    r = np.random.RandomState(41)
    M, N, Kin = 512, 128, 10
    dz = (r.normal(size=(M, N)) * 1e-2).astype(np.float32)
    y = r.normal(size=(M, N)).astype(np.float32)
    coef = r.normal(size=(3, N)).astype(np.float32)
    w = r.normal(size=(Kin, N)).astype(np.float32)

    def make_n(dev):
        t = {k: _mk(dev, v) for k, v in dict(dz=dz, y=y, coef=coef, w=w).items()}
        o = dict(out=torch.zeros(M, 8, device=dev))
        a = abi.DgradNarrowArgs(abi.DySrc(fptr(t['dz']), fptr(t['y']), fptr(t['coef']), iptr(None), fptr(None)), fptr(t['w']), 4, 6,
                                fptr(o['out']), 8, M, N)
        a._keep = (t, o)
        return a, o
    c, g = _run_both(hip_lib, make_n, 't3d_pointmlp_dgrad_narrow')
    _close(c['out'], g['out'], 1e-4, 1e-4 * float(c['out'].abs().max()), 'dgrad_narrow')

    B = 32
    strong = np.array([3.25], np.float32)
    dims = (1.0 + r.normal(size=(B, 3)) * 0.8).astype(np.float32)
    cls = r.randint(0, 10, size=B)
    oh = np.eye(10, dtype=np.float32)[cls]
    is2d = (r.uniform(size=B) < 0.5).astype(np.int32)
    out9 = r.normal(size=(B, 9)).astype(np.float32)
    for only2d in (0, 1):
        def make_l(dev):
            t = {k: _mk(dev, v) for k, v in dict(strong=strong, dims=dims, oh=oh, is2d=is2d, out9=out9).items()}
            o = dict(d_dims=torch.zeros(B, 3, device=dev), dout9=torch.zeros(B, 9, device=dev), fit=torch.zeros(B, device=dev),
                     terms=torch.zeros(2, device=dev), loss=torch.zeros(1, device=dev))
            a = abi.SemiFinalLossArgs()
            a.strong_loss, a.reg_dims, a.one_hot, a.is_data_2D, a.out9 = fptr(t['strong']), fptr(t['dims']), fptr(t['oh']), iptr(t['is2d']), fptr(t['out9'])
            for i in range(10):
                a.train_classes[i] = int(i in (1, 2, 6, 7, 8))
            a.w_weak, a.w_fit, a.fit_only_2d = 0.1, 1.0, only2d
            a.d_dims, a.dout9, a.fit_prob, a.terms, a.loss, a.B = fptr(o['d_dims']), fptr(o['dout9']), fptr(o['fit']), fptr(o['terms']), fptr(o['loss']), B
            a._keep = (t, o)
            return a, o
        c, g = _run_both(hip_lib, make_l, 't3d_semi_final_loss')
        for k in c:
            _close(c[k], g[k], 1e-5, 1e-6, 'semi_final_loss ' + k)

    box = r.normal(size=(B, 67)).astype(np.float32)
    box[0, 37:67] = -1.5                                   # anchor + res*mean < 1e-5: the max() clamp kills that gradient
    dbox7 = r.normal(size=(B, 7)).astype(np.float32)
    dd = r.normal(size=(B, 3)).astype(np.float32)
    g0, s0 = r.normal(size=(B, 67)).astype(np.float32), r.normal(size=(B, 3)).astype(np.float32)

    def make_a(dev):
        t = {k: _mk(dev, v) for k, v in dict(box=box, dbox7=dbox7, dd=dd).items()}
        o = dict(dbox=_mk(dev, g0.copy()), ds1=_mk(dev, s0.copy()))
        a = abi.AnchorRegBwdArgs(fptr(t['box']), 67, fptr(t['dbox7']), fptr(t['dd']), fptr(o['dbox']), fptr(o['ds1']), B)
        a._keep = (t, o)
        return a, o
    c, g = _run_both(hip_lib, make_a, 't3d_anchor_reg_bwd')
    _close(c['dbox'], g['dbox'], 1e-5, 1e-6, 'anchor_reg_bwd dbox')
    _close(c['ds1'], g['ds1'], 1e-6, 1e-6, 'anchor_reg_bwd ds1')


# ------------------------------------------------------------------------------------------------
# Gram-form backward of a pooled layer (t3d.h K11e)
# ------------------------------------------------------------------------------------------------
def _pool_case(M, K, N, rpf, seed):
    r = np.random.RandomState(seed)
    B = M // rpf
    d = dict(x=r.normal(size=(M, K)).astype(np.float32), sc=(0.5 + r.uniform(size=K)).astype(np.float32),
             sh=(r.normal(size=K) * 0.3).astype(np.float32), w=(r.normal(size=(K, N)) / np.sqrt(K)).astype(np.float32),
             bias=(r.normal(size=N) * 0.1).astype(np.float32), coef=r.normal(size=(3, N)).astype(np.float32),
             dpool=r.normal(size=(B, N)).astype(np.float32))
    d['sc'][::5] *= -1
    d['coef'][1] *= 1e-2
    d['coef'][2] *= 1e-3
    # argmax rows cluster on few points, like PointNet critical points; some channels get no gradient
    hot = r.randint(0, rpf, size=(B, 24))
    d['argidx'] = np.take_along_axis(hot, r.randint(0, 24, size=(B, N)), 1).astype(np.int32)
    d['argidx'][r.uniform(size=(B, N)) < 0.1] = -1
    d['argidx'][0, :N // 2] = 5                                         # one row collects half of a frustum's channels
    return d


def _act_src(t, K):
    return abi.ActSrc(fptr(t['x']), K, 0, fptr(t['sc']), fptr(t['sh']), 1, fptr(None), 0)


@pytest.mark.parametrize('K,N', [(128, 1024), (256, 512), (128, 256), (64, 192)])
def test_pool_bwd_prep(hip_lib, K, N):
    d = _pool_case(256, K, N, 128, K + N)

    def make(dev):
        t = {k: _mk(dev, v) for k, v in d.items()}
        nch = (N + 127) // 128
        o = dict(p=torch.zeros(nch, K, K, device=dev), rowconst=torch.zeros(nch, K, device=dev), wc=torch.zeros(N, K, device=dev))
        a = abi.PoolBwdPrepArgs(fptr(t['w']), fptr(t['bias']), fptr(t['coef']), K, N, fptr(o['p']), fptr(o['rowconst']), fptr(o['wc']))
        a._keep = (t, o)
        return a, o

    c, g = _run_both(hip_lib, make, 't3d_pool_bwd_prep')
    _close(c['p'], g['p'], 1e-4, 1e-5 * float(c['p'].abs().max()), 'P slabs')
    _close(c['rowconst'], g['rowconst'], 1e-4, 1e-5 * float(c['rowconst'].abs().max()), 'rowconst slabs')
    _close(c['wc'], g['wc'], 1e-6, 1e-7, 'wc')


@pytest.mark.parametrize('M,K,N,rpf', [(512, 128, 1024, 256), (256, 256, 512, 128), (1024, 128, 256, 1024)])
def test_pool_sparse_rows(hip_lib, M, K, N, rpf):
    d = _pool_case(M, K, N, rpf, M + K)
    wc = np.ascontiguousarray((d['w'] * d['coef'][0]).T)

    def make(dev):
        t = {k: _mk(dev, v) for k, v in dict(d, wc=wc).items()}
        o = dict(s=torch.full((M, K), 7.0, device=dev))                 # every row must be overwritten
        a = abi.PoolSparseRowsArgs(iptr(t['argidx']), fptr(t['dpool']), fptr(t['wc']), M // rpf, N, K, rpf, fptr(o['s']))
        a._keep = (t, o)
        return a, o

    c, g = _run_both(hip_lib, make, 't3d_pool_sparse_rows')
    _close(c['s'], g['s'], 1e-5, 1e-5 * float(c['s'].abs().max()), 'S')
    # bit-reproducible: fixed summation order whatever the schedule
    _, g2 = _run_both(hip_lib, make, 't3d_pool_sparse_rows')
    assert torch.equal(g['s'], g2['s'])


@pytest.mark.parametrize('M,K,rpf', [(512, 128, 256), (256, 256, 128), (8192, 128, 1024)])
def test_pointmlp_dgrad_gram(hip_lib, gemm_arithmetic, M, K, rpf):
    r = np.random.RandomState(M + K)
    T = M // 128
    x = r.normal(size=(M, K)).astype(np.float32)
    sc, sh = (0.5 + r.uniform(size=K)).astype(np.float32), (r.normal(size=K) * 0.3).astype(np.float32)
    P = (r.normal(size=(K, K)) / np.sqrt(K)).astype(np.float32)
    rc = r.normal(size=K).astype(np.float32) * 0.1
    S = (r.normal(size=(M, K)) * (r.uniform(size=(M, 1)) < 0.1)).astype(np.float32)

    def make(dev):
        t = {k: _mk(dev, v) for k, v in dict(x=x, sc=sc, sh=sh, P=P, rc=rc, S=S).items()}
        o = dict(out=torch.zeros(M, K, device=dev), s1=torch.zeros(T, K, device=dev), s2=torch.zeros(T, K, device=dev))
        a = abi.PointMlpDgradGramArgs()
        a.a, a.p, a.rowconst, a.add_in = _act_src(t, K), fptr(t['P']), fptr(t['rc']), fptr(t['S'])
        a.prev_y, a.prev_scale, a.prev_shift = fptr(t['x']), fptr(t['sc']), fptr(t['sh'])
        a.out, a.psum_dz, a.psum_dzy = fptr(o['out']), fptr(o['s1']), fptr(o['s2'])
        a.M, a.K, a.rows_per_frustum = M, K, rpf
        a._keep = (t, o)
        return a, o

    c, g = _run_both(hip_lib, make, 't3d_pointmlp_dgrad_gram')
    scale = float(c['out'].abs().max())
    _close(c['out'], g['out'], 1e-4, 2e-5 * scale, 'out')
    _close(c['s1'], g['s1'], 1e-3, 1e-3 * scale, 'psum_dz')
    _close(c['s2'], g['s2'], 1e-3, 2e-3 * scale, 'psum_dzy')


@pytest.mark.parametrize('M,K,N,rpf,masked', [(512, 128, 1024, 256, True), (4096, 256, 512, 1024, True), (8192, 128, 256, 1024, True),
                                              (1024, 128, 256, 256, False)])
def test_row_gated_sparse_rows_and_dgrad_gram_equal_the_dense_form(hip_lib, gemm_arithmetic, M, K, N, rpf, masked):
    """row_live / add_live: rows of S without an arg-max hit are neither written nor read (they hold NaN here), the flags
    are exactly the hit rows, and the data gradient is bit-identical to the every-row-written form."""
    d = _pool_case(M, K, N, rpf, M + K + N)
    r = np.random.RandomState(5)
    dev = _dev('cuda')
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    B, T = M // rpf, M // 128
    t = {k: _mk(dev, v) for k, v in dict(d, P=(r.normal(size=(K, K)) / np.sqrt(K)).astype(np.float32),
                                         rc=(r.normal(size=K) * 0.1).astype(np.float32)).items()}
    wc = (t['coef'][0][:, None] * t['w'].t()).contiguous()
    res = []
    for gated in (False, True):
        S = torch.full((M, K), float('nan'), device=dev)
        live = torch.full((M,), 77, dtype=torch.int32, device=dev)
        sp = abi.PoolSparseRowsArgs(iptr(t['argidx']), fptr(t['dpool']), fptr(wc), B, N, K, rpf, fptr(S), iptr(live if gated else None))
        assert hip_lib.t3d_pool_sparse_rows(C.byref(sp), st) == 0
        out, s1, s2 = torch.zeros(M, K, device=dev), torch.zeros(T, K, device=dev), torch.zeros(T, K, device=dev)
        a = abi.PointMlpDgradGramArgs()
        a.a, a.p, a.rowconst, a.add_in, a.add_live = _act_src(t, K), fptr(t['P']), fptr(t['rc']), fptr(S), iptr(live if gated else None)
        if masked:
            a.prev_y, a.prev_scale, a.prev_shift, a.psum_dz, a.psum_dzy = fptr(t['x']), fptr(t['sc']), fptr(t['sh']), fptr(s1), fptr(s2)
        a.out, a.M, a.K, a.rows_per_frustum = fptr(out), M, K, rpf
        assert hip_lib.t3d_pointmlp_dgrad_gram(C.byref(a), st) == 0
        torch.cuda.synchronize()
        res.append((out, s1, s2, S, live))
    want = np.zeros(M, np.int32)
    for b in range(B):
        want[b * rpf + d['argidx'][b][d['argidx'][b] >= 0]] = 1
    assert np.array_equal(res[1][4].cpu().numpy(), want) and 0 < want.sum() < M
    assert bool(torch.isnan(res[1][3][torch.as_tensor(want == 0)]).all()), 'a row without a hit was written'
    assert torch.equal(res[1][3][torch.as_tensor(want == 1)], res[0][3][torch.as_tensor(want == 1)])
    for x, y, what in zip(res[0][:3], res[1][:3], ('out', 'psum_dz', 'psum_dzy')):
        assert torch.equal(x, y) and bool(torch.isfinite(y).all()), what
    a.add_in = fptr(None)
    assert hip_lib.t3d_pointmlp_dgrad_gram(C.byref(a), st) == -1          # T3D_ERR_ARG: flags without the rows


@pytest.mark.parametrize('M,K,rpf', [(1024, 128, 256), (512, 256, 128), (4096, 64, 1024)])
def test_pointmlp_gram_and_act_colsum(hip_lib, gemm_arithmetic, M, K, rpf):
    d = _pool_case(M, K, 64, rpf, M + 3 * K)
    rps, tk, tn = C.c_int(0), C.c_int(0), C.c_int(0)
    assert hip_lib.t3d_wgrad_plan(M, K, K, C.byref(rps), C.byref(tk), C.byref(tn)) == 0
    S = M // rps.value

    def make_g(dev):
        t = {k: _mk(dev, v) for k, v in d.items()}
        o = dict(slabs=torch.zeros(S, K, K, device=dev))
        a = abi.PointMlpGramArgs(_act_src(t, K), fptr(o['slabs']), M, K, rpf, rps.value)
        a._keep = (t, o)
        return a, o

    c, g = _run_both(hip_lib, make_g, 't3d_pointmlp_gram')
    _close(c['slabs'], g['slabs'], 1e-4, 1e-5 * float(c['slabs'].abs().max()), 'gram slabs')
    gs = g['slabs'].sum(0)
    # (x3: the six partial products of G[i][j] and G[j][i] come in another order -- three accumulator sets make the sum order-free)
    assert torch.equal(gs, gs.T.contiguous()), 'Gram matrix must be bitwise symmetric (finish kernel reads it transposed)'

    def make_c(dev):
        t = {k: _mk(dev, v) for k, v in d.items()}
        o = dict(part=torch.zeros(M // 128, K, device=dev))
        a = abi.ActColsumArgs(_act_src(t, K), M, K, rpf, fptr(o['part']))
        a._keep = (t, o)
        return a, o

    c, g = _run_both(hip_lib, make_c, 't3d_act_colsum')
    _close(c['part'], g['part'], 1e-5, 1e-4, 'abar partials')


@pytest.mark.parametrize('M,K,N,rpf', [(512, 128, 1024, 256), (256, 256, 512, 128), (512, 64, 64, 128)])
def test_pool_wgrad_finish(hip_lib, M, K, N, rpf):
    d = _pool_case(M, K, N, rpf, M + N)
    r = np.random.RandomState(5)
    G = r.normal(size=(K, K)).astype(np.float32)
    G = G + G.T
    part = r.normal(size=K).astype(np.float32)

    def make(dev):
        t = {k: _mk(dev, v) for k, v in dict(d, G=G, part=part).items()}
        o = dict(dw=torch.zeros(K, N, device=dev))
        a = abi.PoolWgradFinishArgs()
        a.a, a.argidx, a.dpool, a.coef = _act_src(t, K), iptr(t['argidx']), fptr(t['dpool']), fptr(t['coef'])
        a.w, a.bias, a.g, a.abar = fptr(t['w']), fptr(t['bias']), fptr(t['G']), fptr(t['part'])
        a.B, a.K, a.N, a.rows_per_frustum, a.dw = M // rpf, K, N, rpf, fptr(o['dw'])
        a._keep = (t, o)
        return a, o

    c, g = _run_both(hip_lib, make, 't3d_pool_wgrad_finish')
    _close(c['dw'], g['dw'], 1e-4, 1e-5 * float(c['dw'].abs().max()), 'dw')


def test_pointmlp_fwd_without_y_store_gives_the_same_statistics(hip_lib, gemm_arithmetic):
    M, K, N, rpf = 512, 128, 256, 256
    r = np.random.RandomState(11)
    x, w = r.normal(size=(M, K)).astype(np.float32), (r.normal(size=(K, N)) / np.sqrt(K)).astype(np.float32)
    dev = _dev('cuda')
    outs = []
    for store in (True, False):
        t = dict(x=_mk(dev, x), w=_mk(dev, w))
        o = {k: torch.zeros(M // 128, N, device=dev) for k in ('psum', 'psumsq', 'pmax', 'pmin')}
        o.update(pamax=torch.zeros(M // 128, N, dtype=torch.int32, device=dev), pamin=torch.zeros(M // 128, N, dtype=torch.int32, device=dev))
        y = torch.zeros(M, N, device=dev)
        a = abi.PointMlpFwdArgs()
        a.a = abi.ActSrc(fptr(t['x']), K, 0, fptr(None), fptr(None), 0, fptr(None), 0)
        a.w, a.y, a.psum, a.psumsq = fptr(t['w']), fptr(y if store else None), fptr(o['psum']), fptr(o['psumsq'])
        a.pmax, a.pmin, a.pamax, a.pamin = fptr(o['pmax']), fptr(o['pmin']), iptr(o['pamax']), iptr(o['pamin'])
        a.M, a.K, a.N, a.rows_per_frustum = M, K, N, rpf
        assert hip_lib.t3d_pointmlp_fwd(C.byref(a), C.c_void_p(torch.cuda.current_stream().cuda_stream)) == 0
        torch.cuda.synchronize()
        outs.append({k: v.cpu() for k, v in o.items()})
    for k in outs[0]:
        assert torch.equal(outs[0][k], outs[1][k]), k


def test_gram_form_equals_direct_form_through_the_kernels(hip_lib, gemm_arithmetic):
    """da and dW of a pooled layer: Gram-form kernel chain vs the direct pooled-sparse dgrad / wgrad kernels."""
    M, K, N, rpf = 1024, 128, 512, 256
    B, T = M // rpf, M // 128
    d = _pool_case(M, K, N, rpf, 77)
    dev = _dev('cuda')
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    t = {k: _mk(dev, v) for k, v in d.items()}
    act = _act_src(t, K)
    z = lambda *s, **kw: torch.zeros(*s, device=dev, **kw)
    # forward (stores y for the direct form)
    y, ps, pq = z(M, N), z(T, N), z(T, N)
    f = abi.PointMlpFwdArgs()
    f.a, f.w, f.bias, f.y, f.psum, f.psumsq, f.M, f.K, f.N, f.rows_per_frustum = act, fptr(t['w']), fptr(t['bias']), fptr(y), fptr(ps), fptr(pq), M, K, N, rpf
    assert hip_lib.t3d_pointmlp_fwd(C.byref(f), st) == 0
    # direct form
    dy = abi.DySrc(fptr(None), fptr(y), fptr(t['coef']), iptr(t['argidx']), fptr(t['dpool']))
    out_d, s1d, s2d = z(M, K), z(T, K), z(T, K)
    a = abi.PointMlpDgradArgs()
    a.dy, a.w, a.prev_y, a.prev_scale, a.prev_shift = dy, fptr(t['w']), fptr(t['x']), fptr(t['sc']), fptr(t['sh'])
    a.out, a.psum_dz, a.psum_dzy, a.M, a.K, a.N, a.rows_per_frustum = fptr(out_d), fptr(s1d), fptr(s2d), M, K, N, rpf
    assert hip_lib.t3d_pointmlp_dgrad(C.byref(a), st) == 0
    slab = z(1, K, N)
    wg = abi.PointMlpWgradArgs(act, dy, fptr(slab), M, K, N, rpf, M)
    assert hip_lib.t3d_pointmlp_wgrad(C.byref(wg), st) == 0
    # Gram form
    nch = N // 128
    Ps, rcs, wc, S = z(nch, K, K), z(nch, K), z(N, K), z(M, K)
    assert hip_lib.t3d_pool_bwd_prep(C.byref(abi.PoolBwdPrepArgs(fptr(t['w']), fptr(t['bias']), fptr(t['coef']), K, N, fptr(Ps), fptr(rcs), fptr(wc))), st) == 0
    P, rc = Ps.sum(0), rcs.sum(0)
    assert hip_lib.t3d_pool_sparse_rows(C.byref(abi.PoolSparseRowsArgs(iptr(t['argidx']), fptr(t['dpool']), fptr(wc), B, N, K, rpf, fptr(S))), st) == 0
    out_g, s1g, s2g = z(M, K), z(T, K), z(T, K)
    gg = abi.PointMlpDgradGramArgs()
    gg.a, gg.p, gg.rowconst, gg.add_in, gg.prev_y, gg.prev_scale, gg.prev_shift = act, fptr(P), fptr(rc), fptr(S), fptr(t['x']), fptr(t['sc']), fptr(t['sh'])
    gg.out, gg.psum_dz, gg.psum_dzy, gg.M, gg.K, gg.rows_per_frustum = fptr(out_g), fptr(s1g), fptr(s2g), M, K, rpf
    assert hip_lib.t3d_pointmlp_dgrad_gram(C.byref(gg), st) == 0
    gsl, part, dw = z(1, K, K), z(T, K), z(K, N)
    assert hip_lib.t3d_pointmlp_gram(C.byref(abi.PointMlpGramArgs(act, fptr(gsl), M, K, rpf, M)), st) == 0
    assert hip_lib.t3d_act_colsum(C.byref(abi.ActColsumArgs(act, M, K, rpf, fptr(part))), st) == 0
    fin = abi.PoolWgradFinishArgs()
    fin.a, fin.argidx, fin.dpool, fin.coef, fin.w, fin.bias = act, iptr(t['argidx']), fptr(t['dpool']), fptr(t['coef']), fptr(t['w']), fptr(t['bias'])
    torch.cuda.synchronize()
    abar = part.sum(0)
    fin.g, fin.abar, fin.B, fin.K, fin.N, fin.rows_per_frustum, fin.dw = fptr(gsl), fptr(abar), B, K, N, rpf, fptr(dw)
    assert hip_lib.t3d_pool_wgrad_finish(C.byref(fin), st) == 0
    torch.cuda.synchronize()
    sd, sw = float(out_d.abs().max()), float(slab.abs().max())
    _close(out_d.cpu(), out_g.cpu(), 1e-4, 3e-5 * sd, 'da gram vs direct')
    _close(slab[0].cpu(), dw.cpu(), 1e-4, 3e-5 * sw, 'dW gram vs direct')
    _close(s1d.cpu(), s1g.cpu(), 1e-3, 1e-3 * sd, 'psum_dz gram vs direct')


@pytest.mark.parametrize('M,K,N,rpf,mode', [(1024, 128, 128, 256, 'mask'), (512, 64, 512, 128, 'raw'), (1024, 512, 256, 256, 'mask_addin'),
                                            (65536, 64, 64, 1024, 'mask')])
def test_fused_bwd_equals_separate_dgrad_and_wgrad(hip_lib, gemm_arithmetic, M, K, N, rpf, mode):
    """t3d_pointmlp_bwd is the two kernels in one launch: bit-identical outputs."""
    r = np.random.RandomState(M + K + N)
    dev = _dev('cuda')
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    T = M // 128
    t = {k: _mk(dev, v) for k, v in dict(
        x=r.normal(size=(M, K)).astype(np.float32), sc=(0.5 + r.uniform(size=K)).astype(np.float32),
        sh=(r.normal(size=K) * 0.3).astype(np.float32), w=(r.normal(size=(K, N)) / np.sqrt(N)).astype(np.float32),
        dz=(r.normal(size=(M, N)) * 1e-2).astype(np.float32), y=r.normal(size=(M, N)).astype(np.float32),
        coef=r.normal(size=(3, N)).astype(np.float32), add=(r.normal(size=(M, K)) * 1e-2).astype(np.float32)).items()}
    rps, tk, tn = C.c_int(0), C.c_int(0), C.c_int(0)
    assert hip_lib.t3d_wgrad_plan(M, K, N, C.byref(rps), C.byref(tk), C.byref(tn)) == 0
    S = M // rps.value
    act = _act_src(t, K)
    dy = abi.DySrc(fptr(t['dz']), fptr(t['y']), fptr(t['coef']), iptr(None), fptr(None))
    res = []
    for fused in (False, True):
        out, s1, s2 = torch.zeros(M, K, device=dev), torch.zeros(T, K, device=dev), torch.zeros(T, K, device=dev)
        slabs = torch.zeros(S, K, N, device=dev)
        d = abi.PointMlpDgradArgs()
        d.dy, d.w, d.out = dy, fptr(t['w']), fptr(out)
        if 'addin' in mode:
            d.add_in = fptr(t['add'])
        if 'mask' in mode:
            d.prev_y, d.prev_scale, d.prev_shift, d.psum_dz, d.psum_dzy = fptr(t['x']), fptr(t['sc']), fptr(t['sh']), fptr(s1), fptr(s2)
        d.M, d.K, d.N, d.rows_per_frustum = M, K, N, rpf
        w = abi.PointMlpWgradArgs(act, dy, fptr(slabs), M, K, N, rpf, rps.value)
        if fused:
            assert hip_lib.t3d_pointmlp_bwd(C.byref(d), C.byref(w), st) == 0
        else:
            assert hip_lib.t3d_pointmlp_dgrad(C.byref(d), st) == 0
            assert hip_lib.t3d_pointmlp_wgrad(C.byref(w), st) == 0
        torch.cuda.synchronize()
        res.append((out, s1, s2, slabs))
    # the fused launch takes the one-pass form where the shape and this split allow it (t3d.h, t3d_bwd_plan): dX stays bit-identical
    # (same MFMA step order), the statistics and the slab SUM differ by the fp32 summation order only
    one_pass = rps.value % 128 == 0 and S >= min(256, T) and (T < 256 or rps.value >= 256) and K in (64, 128) and N in (64, 128)
    if one_pass and gemm_arithmetic == 'x3':
        # the one-pass kernel multiplies on the fp32 matrix pipe, the separate launches on the bf16 one with three-term operands
        # (csrc/pointmlp.hip PathX3): same values to fp32 rounding, another summation order
        _close(res[0][0].cpu(), res[1][0].cpu(), 1e-4, 2e-6 * float(res[0][0].abs().max()), 'out (one-pass fp32-MFMA vs x3)')
    else:
        assert torch.equal(res[0][0], res[1][0]), 'out'
    if not one_pass:
        for a, b, what in zip(res[0][1:], res[1][1:], ('psum_dz', 'psum_dzy', 'slabs')):
            assert torch.equal(a, b), what
    else:
        sd = float(res[0][0].abs().max())
        _close(res[0][1].cpu(), res[1][1].cpu(), 1e-4, 2e-5 * sd, 'psum_dz')
        _close(res[0][2].cpu(), res[1][2].cpu(), 1e-4, 1e-4 * sd, 'psum_dzy')
        dw0, dw1 = res[0][3].sum(0), res[1][3].sum(0)
        _close(dw0.cpu(), dw1.cpu(), 1e-4, 2e-5 * float(dw0.abs().max()), 'dW')
    assert float(res[1][3].abs().max()) > 0 and float(res[1][0].abs().max()) > 0


@pytest.mark.parametrize('M,K,N,rpf,mode,rps_', [
    (256, 64, 64, 128, 'mask', -1), (512, 128, 128, 256, 'mask_addin', -1), (1024, 64, 128, 256, 'raw', -1), (768, 128, 64, 128, 'mask', -1),
    (384, 128, 128, 128, 'raw_addin', 128), (65536, 128, 128, 1024, 'mask', -1), (65536, 64, 128, 2048, 'mask_addin', 256),
    (65536, 128, 64, 1024, 'mask', 256), (65536, 64, 64, 1024, 'mask', -1), (1024, 128, 128, 256, 'mask_nostats', -1), (131072, 128, 128, 1024, 'mask_addin', 512)])
def test_fp32_one_pass_backward(hip_lib, fp32_mfma, M, K, N, rpf, mode, rps_):
    """k_pointmlp_bwd1f (the one-pass form t3d_pointmlp_bwd takes for fp32 layers with K, N in {64, 128}) against float64, and its dX
    bit for bit against t3d_pointmlp_dgrad.  rps_: -1 = t3d_bwd_plan's split, > 0 = that many rows per split (whole 128-row tiles,
    >= min(256, M / 128) workgroups, and >= 256 rows per workgroup once M >= 32768: below that the launcher keeps the split form)."""
    g = torch.Generator(device='cpu').manual_seed(M + 3 * K + 7 * N + len(mode))
    dev, T = 'cuda', M // 128
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    dz = (torch.randn(M, N, generator=g) * 1e-2).to(dev)
    y = torch.randn(M, N, generator=g).to(dev)
    coef = torch.randn(3, N, generator=g)
    coef[2] *= 1e-3
    coef = coef.to(dev)
    w_ = (torch.randn(K, N, generator=g) / np.sqrt(N)).to(dev)
    x = torch.randn(M, K, generator=g).to(dev)
    psc, psh = (0.5 + torch.rand(K, generator=g)).to(dev), (torch.randn(K, generator=g) * 0.3).to(dev)
    add = (torch.randn(M, K, generator=g) * 1e-2).to(dev)
    rps, one = C.c_int(0), C.c_int(0)
    assert hip_lib.t3d_bwd_plan(M, K, N, abi.F32, C.byref(rps), C.byref(one)) == 0
    assert one.value == 1 and rps.value % 128 == 0 and M // rps.value >= min(256, T)
    if rps_ > 0:
        rps.value = rps_
        assert M // rps_ >= min(256, T) and (T < 256 or rps_ >= 256)
    masked, stats, addin = 'mask' in mode, 'mask' in mode and 'nostats' not in mode, 'addin' in mode
    dy = abi.DySrc(fptr(dz), fptr(y), fptr(coef), iptr(None), fptr(None))
    outs = []
    for fused in (True, False):
        out = torch.full((M, K), float('nan'), device=dev)
        s1, s2 = torch.full((T, K), float('nan'), device=dev), torch.full((T, K), float('nan'), device=dev)
        slabs = torch.full((M // rps.value, K, N), float('nan'), device=dev)
        d = abi.PointMlpDgradArgs()
        d.dy, d.w, d.out, d.add_in = dy, fptr(w_), fptr(out), fptr(add if addin else None)
        if masked:
            d.prev_y, d.prev_scale, d.prev_shift = fptr(x), fptr(psc), fptr(psh)
        if stats:
            d.psum_dz, d.psum_dzy = fptr(s1), fptr(s2)
        d.M, d.K, d.N, d.rows_per_frustum = M, K, N, rpf
        wa = abi.PointMlpWgradArgs()
        wa.a = abi.ActSrc(fptr(x), K, 0, fptr(psc), fptr(psh), 1, fptr(None), 0) if masked else \
            abi.ActSrc(fptr(x), K, 0, fptr(None), fptr(None), 0, fptr(None), 0)
        wa.dy, wa.slabs = dy, fptr(slabs)
        wa.M, wa.K, wa.N, wa.rows_per_frustum, wa.rows_per_split = M, K, N, rpf, rps.value
        if fused:
            assert hip_lib.t3d_pointmlp_bwd(C.byref(d), C.byref(wa), st) == 0
        else:
            assert hip_lib.t3d_pointmlp_dgrad(C.byref(d), st) == 0
        torch.cuda.synchronize()
        outs.append((out, s1, s2, slabs))
    out, s1, s2, slabs = outs[0]
    assert torch.equal(out, outs[1][0]), 'dX: one-pass vs t3d_pointmlp_dgrad'
    dyv = coef[0].double() * dz.double() + coef[1].double() * y.double() + coef[2].double()
    ref = dyv @ w_.double().t() + (add.double() if addin else 0)
    z = x.double() * psc.double() + psh.double()
    if masked:
        ref = torch.where(z > 0, ref, torch.zeros_like(ref))
    sd = float(ref.abs().max())
    assert float((out.double() - ref).abs().max()) < 1e-5 * sd * max(1.0, np.sqrt(N / 64))
    if stats:
        o = out.double().reshape(T, 128, K)
        assert float((s1.double() - o.sum(1)).abs().max()) < 2e-5 * float(o.abs().sum(1).max() + 1e-9)
        assert float((s2.double() - (o * x.double().reshape(T, 128, K)).sum(1)).abs().max()) < 1e-5 * float(o.abs().sum(1).max() + 1e-9)
    a = torch.relu(z) if masked else x.double()
    dw_ref = a.t() @ dyv
    dw = slabs.double().sum(0)
    assert not bool(torch.isnan(dw).any())
    assert float((dw - dw_ref).abs().max()) < 2e-6 * float(dw_ref.abs().max()) * max(1.0, np.sqrt(M / 512))


def test_fused_pool_stages_equal_the_separate_launches(hip_lib, gemm_arithmetic):
    """t3d_pool_bwd_stage1 / stage2 are the separate K11e launches sharing a grid: bit-identical outputs."""
    M, K, N, rpf = 2048, 128, 512, 512
    B, T = M // rpf, M // 128
    d = _pool_case(M, K, N, rpf, 31)
    dev = _dev('cuda')
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    t = {k: _mk(dev, v) for k, v in d.items()}
    act = _act_src(t, K)
    z = lambda *s: torch.zeros(*s, device=dev)
    rps, tk, tn = C.c_int(0), C.c_int(0), C.c_int(0)
    assert hip_lib.t3d_wgrad_plan(M, K, K, C.byref(rps), C.byref(tk), C.byref(tn)) == 0
    S, nch = M // rps.value, N // 128
    out = []
    for fused in (False, True):
        gsl, part, ps, rcs, wc = z(S, K, K), z(T, K), z(nch, K, K), z(nch, K), z(N, K)
        ga = abi.PointMlpGramArgs(act, fptr(gsl), M, K, rpf, rps.value)
        ca = abi.ActColsumArgs(act, M, K, rpf, fptr(part))
        qa = abi.PoolBwdPrepArgs(fptr(t['w']), fptr(t['bias']), fptr(t['coef']), K, N, fptr(ps), fptr(rcs), fptr(wc))
        if fused:
            assert hip_lib.t3d_pool_bwd_stage1(C.byref(ga), C.byref(ca), C.byref(qa), st) == 0
        else:
            assert hip_lib.t3d_pointmlp_gram(C.byref(ga), st) == 0
            assert hip_lib.t3d_act_colsum(C.byref(ca), st) == 0
            assert hip_lib.t3d_pool_bwd_prep(C.byref(qa), st) == 0
        torch.cuda.synchronize()
        G, abar, P, rc = gsl.sum(0), part.sum(0), ps.sum(0), rcs.sum(0)
        Sm = (torch.randn(M, K, device=dev) * (torch.rand(M, 1, device=dev) < 0.1)).contiguous() if not out else out[0][-1]
        dw, o, s1, s2 = z(K, N), z(M, K), z(T, K), z(T, K)
        f = abi.PoolWgradFinishArgs()
        f.a, f.argidx, f.dpool, f.coef, f.w, f.bias = act, iptr(t['argidx']), fptr(t['dpool']), fptr(t['coef']), fptr(t['w']), fptr(t['bias'])
        f.g, f.abar, f.B, f.K, f.N, f.rows_per_frustum, f.dw = fptr(G), fptr(abar), B, K, N, rpf, fptr(dw)
        dg = abi.PointMlpDgradGramArgs()
        dg.a, dg.p, dg.rowconst, dg.add_in, dg.prev_y, dg.prev_scale, dg.prev_shift = act, fptr(P), fptr(rc), fptr(Sm), fptr(t['x']), fptr(t['sc']), fptr(t['sh'])
        dg.out, dg.psum_dz, dg.psum_dzy, dg.M, dg.K, dg.rows_per_frustum = fptr(o), fptr(s1), fptr(s2), M, K, rpf
        if fused:
            assert hip_lib.t3d_pool_bwd_stage2(C.byref(f), C.byref(dg), st) == 0
        else:
            assert hip_lib.t3d_pool_wgrad_finish(C.byref(f), st) == 0
            assert hip_lib.t3d_pointmlp_dgrad_gram(C.byref(dg), st) == 0
        torch.cuda.synchronize()
        out.append((gsl, part, ps, rcs, wc, dw, o, s1, s2, Sm))
    for a, b, what in zip(out[0], out[1], ('gram', 'abar', 'P', 'rowconst', 'wc', 'dw', 'da', 'psum_dz', 'psum_dzy', 'S')):
        assert torch.equal(a, b), what
    assert float(out[1][5].abs().max()) > 0 and float(out[1][6].abs().max()) > 0


@pytest.mark.parametrize('weigh,first', [(0, 1), (1, 0)])
def test_box_refine_step(hip_lib, weigh, first):
    r = np.random.RandomState(3 + weigh)
    B = 37
    d = dict(out9=r.normal(size=(B, 9)).astype(np.float32), c=r.normal(size=(B, 3)).astype(np.float32),
             s=(1 + r.uniform(size=(B, 3))).astype(np.float32), th=r.normal(size=B).astype(np.float32),
             tot=r.normal(size=(B, 7)).astype(np.float32))

    def make(dev):
        t = {k: _mk(dev, v) for k, v in d.items()}
        o = dict(c=torch.zeros(B, 3, device=dev), s=torch.zeros(B, 3, device=dev), th=torch.zeros(B, device=dev), tot=t['tot'].clone(),
                 fit=torch.zeros(B, device=dev))
        a = abi.BoxRefineStepArgs(fptr(t['out9']), fptr(t['c']), fptr(t['s']), fptr(t['th']), fptr(o['c']), fptr(o['s']), fptr(o['th']),
                                  fptr(o['tot']), fptr(o['fit']), weigh, first, B)
        a._keep = (t, o)
        return a, o

    c, g = _run_both(hip_lib, make, 't3d_box_refine_step')
    for k in c:
        _close(c[k], g[k], 1e-5, 1e-6, 'refine_step ' + k)


@pytest.mark.parametrize('M,K,N,rpf,masked', [(512, 128, 1024, 256, True), (1024, 256, 512, 512, True), (256, 128, 256, 128, False),
                                              (32768, 128, 1024, 1024, True)])
def test_pooled_forward_kernel_equals_generic_kernel(hip_lib, fp32_mfma, M, K, N, rpf, masked):
    """y = NULL on a pooled layer with K in {128, 256} takes the A-resident persistent kernel; with a y buffer the generic
    kernel runs.  Same accumulation and reduction order by construction: statistics and pool partials bit-identical."""
    r = np.random.RandomState(M + K + N)
    dev = _dev('cuda')
    T, B = M // 128, M // rpf
    t = {k: _mk(dev, v) for k, v in dict(
        x=r.normal(size=(M, K)).astype(np.float32), sc=(0.5 + r.uniform(size=K)).astype(np.float32) * np.where(r.uniform(size=K) < 0.2, -1, 1).astype(np.float32),
        sh=(r.normal(size=K) * 0.3).astype(np.float32), w=(r.normal(size=(K, N)) / np.sqrt(K)).astype(np.float32),
        bias=(r.normal(size=N) * 0.1).astype(np.float32), rb=r.normal(size=(B, N)).astype(np.float32),
        mask=(r.uniform(size=M) < 0.4).astype(np.float32)).items()}
    t['mask'][:128] = 0
    outs = []
    for store in (True, False):
        o = {k: torch.zeros(T, N, device=dev) for k in ('psum', 'psumsq', 'pmax', 'pmin')}
        o.update(pamax=torch.zeros(T, N, dtype=torch.int32, device=dev), pamin=torch.zeros(T, N, dtype=torch.int32, device=dev))
        y = torch.zeros(M, N, device=dev)
        a = abi.PointMlpFwdArgs()
        a.a = abi.ActSrc(fptr(t['x']), K, 0, fptr(t['sc']), fptr(t['sh']), 1, fptr(None), 0)
        a.w, a.bias, a.rowbias, a.y = fptr(t['w']), fptr(t['bias']), fptr(t['rb']), fptr(y if store else None)
        a.psum, a.psumsq = fptr(o['psum']), fptr(o['psumsq'])
        if masked:
            a.rowmask = fptr(t['mask'])
        a.pmax, a.pmin, a.pamax, a.pamin = fptr(o['pmax']), fptr(o['pmin']), iptr(o['pamax']), iptr(o['pamin'])
        a.M, a.K, a.N, a.rows_per_frustum = M, K, N, rpf
        assert hip_lib.t3d_pointmlp_fwd(C.byref(a), C.c_void_p(torch.cuda.current_stream().cuda_stream)) == 0
        torch.cuda.synchronize()
        outs.append({k: v.cpu() for k, v in o.items()})
    valid = outs[0]['pamax'] >= 0
    assert torch.equal(outs[0]['pamax'], outs[1]['pamax']) and torch.equal(outs[0]['pamin'], outs[1]['pamin'])
    for k in ('psum', 'psumsq'):
        assert torch.equal(outs[0][k], outs[1][k]), k
    assert torch.equal(outs[0]['pmax'][valid], outs[1]['pmax'][valid]) and torch.equal(outs[0]['pmin'][valid], outs[1]['pmin'][valid])
    assert int(valid.sum()) > 0


def test_fc_dinput_with_fused_pooled_bn_bwd(hip_lib):
    """t3d_fc_dinput with the bn_* fields = t3d_fc_dinput followed by t3d_bn_bwd_finalize (pooled form)."""
    r = np.random.RandomState(21)
    B, N, K = 32, 256, 512
    d = dict(dy=r.normal(size=(B, N)).astype(np.float32), w=(r.normal(size=(K, N)) / 16).astype(np.float32),
             pooled=np.maximum(r.normal(size=(B, K)), 0).astype(np.float32), ysel=r.normal(size=(B, K)).astype(np.float32),
             gamma=(0.5 + r.uniform(size=K)).astype(np.float32), mean=r.normal(size=K).astype(np.float32),
             invstd=(0.5 + r.uniform(size=K)).astype(np.float32), scale=r.normal(size=K).astype(np.float32))

    def make(dev):
        t = {k: _mk(dev, v) for k, v in d.items()}
        o = dict(din=torch.zeros(B, K, device=dev), dpool=torch.zeros(B, K, device=dev), dg=torch.zeros(K, device=dev),
                 db=torch.zeros(K, device=dev), coef=torch.zeros(3, K, device=dev))
        a = abi.FcDinputArgs()
        a.dy, a.N, a.w, a.alpha, a.din, a.ld_din, a.B, a.K = fptr(t['dy']), N, fptr(t['w']), 1.0, fptr(o['din']), K, B, K
        a.bn_pooled, a.bn_ld_pooled, a.bn_ysel, a.bn_dpool, a.bn_count = fptr(t['pooled']), K, fptr(t['ysel']), fptr(o['dpool']), B * 1024
        a.bn_gamma, a.bn_mean, a.bn_invstd, a.bn_scale = fptr(t['gamma']), fptr(t['mean']), fptr(t['invstd']), fptr(t['scale'])
        a.bn_dgamma, a.bn_dbeta, a.bn_coef = fptr(o['dg']), fptr(o['db']), fptr(o['coef'])
        a._keep = (t, o)
        return a, o

    c, g = _run_both(hip_lib, make, 't3d_fc_dinput')
    _close(c['din'], g['din'], 1e-5, 1e-5, 'din')
    _close(c['dpool'], g['dpool'], 1e-5, 1e-5, 'dpool')
    _close(c['dg'], g['dg'], 1e-4, 1e-4, 'dgamma')
    _close(c['db'], g['db'], 1e-4, 1e-4, 'dbeta')
    _close(c['coef'], g['coef'], 1e-4, 1e-6, 'coef')


def test_bn_fwd_finalize_with_fused_pool_pick(hip_lib):
    """t3d_bn_fwd_finalize with the pool_* fields = t3d_bn_fwd_finalize followed by t3d_pool_finalize, bit for bit."""
    r = np.random.RandomState(8)
    B, tpf, N = 6, 4, 256
    T = B * tpf
    dev = _dev('cuda')
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    t = {k: _mk(dev, v) for k, v in dict(
        psum=r.normal(size=(T, N)).astype(np.float32) * 10, psumsq=(100 + 50 * r.uniform(size=(T, N))).astype(np.float32),
        gamma=r.normal(size=N).astype(np.float32), beta=r.normal(size=N).astype(np.float32), decay=np.array([0.5], np.float32),
        pmax=r.normal(size=(T, N)).astype(np.float32), pmin=(r.normal(size=(T, N)) - 3).astype(np.float32),
        pamax=r.randint(-1, 512, size=(T, N)).astype(np.int32), pamin=r.randint(-1, 512, size=(T, N)).astype(np.int32)).items()}
    res = []
    for fused in (False, True):
        o = dict(scale=torch.zeros(N, device=dev), shift=torch.zeros(N, device=dev), mean=torch.zeros(N, device=dev), invstd=torch.zeros(N, device=dev),
                 mm=torch.zeros(N, device=dev), mv=torch.ones(N, device=dev), pooled=torch.zeros(B, N, device=dev),
                 argidx=torch.zeros(B, N, dtype=torch.int32, device=dev), ysel=torch.zeros(B, N, device=dev))
        f = abi.BnFwdFinalizeArgs()
        f.psum, f.psumsq, f.n_tiles, f.count, f.N = fptr(t['psum']), fptr(t['psumsq']), T, T * 128, N
        f.gamma, f.beta, f.moving_mean, f.moving_var, f.decay = fptr(t['gamma']), fptr(t['beta']), fptr(o['mm']), fptr(o['mv']), fptr(t['decay'])
        f.eps, f.is_training, f.unbiased_ema = 1e-3, 1, 1
        f.scale, f.shift, f.mean, f.invstd = fptr(o['scale']), fptr(o['shift']), fptr(o['mean']), fptr(o['invstd'])
        if fused:
            f.pool_pmax, f.pool_pmin, f.pool_pamax, f.pool_pamin = fptr(t['pmax']), fptr(t['pmin']), iptr(t['pamax']), iptr(t['pamin'])
            f.pool_B, f.pool_tiles_per_frustum, f.pooled, f.ld_pooled, f.argidx, f.ysel = B, tpf, fptr(o['pooled']), N, iptr(o['argidx']), fptr(o['ysel'])
        assert hip_lib.t3d_bn_fwd_finalize(C.byref(f), st) == 0
        if not fused:
            q = abi.PoolFinalizeArgs()
            q.scale, q.shift, q.pmax, q.pmin, q.pamax, q.pamin = fptr(o['scale']), fptr(o['shift']), fptr(t['pmax']), fptr(t['pmin']), iptr(t['pamax']), iptr(t['pamin'])
            q.B, q.N, q.tiles_per_frustum, q.pooled, q.ld_pooled, q.argidx, q.ysel = B, N, tpf, fptr(o['pooled']), N, iptr(o['argidx']), fptr(o['ysel'])
            assert hip_lib.t3d_pool_finalize(C.byref(q), st) == 0
        torch.cuda.synchronize()
        res.append(o)
    for k in res[0]:
        assert torch.equal(res[0][k], res[1][k]), k
    assert float(res[1]['pooled'].abs().max()) > 0


@pytest.mark.parametrize('T,tpf,N', [(640, 8, 192), (2048, 16, 128), (2304, 8, 72), (4096, 16, 64), (2048, 16, 256)])
def test_bn_finalizers_many_tiles(hip_lib, T, tpf, N):
    """The finalizers at the tile counts of config 4 (B=128 N=2048: 2048 row tiles) take other block shapes (64 tile groups above 512
    tiles, 4 channels x 256 groups from 1024): forward and dense backward against float64, the fused pool pick against
    t3d_pool_finalize bit for bit."""
    r = np.random.RandomState(T + N)
    dev = _dev('cuda')
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    B, M = T // tpf, T * 128
    psum = (r.normal(size=(T, N)) * 30).astype(np.float32)
    psumsq = (np.abs(r.normal(size=(T, N))) * 128 + psum.astype(np.float64) ** 2 / 128).astype(np.float32)
    gamma, beta = r.normal(size=N).astype(np.float32), r.normal(size=N).astype(np.float32)
    mm, mv = r.normal(size=N).astype(np.float32), (0.5 + r.uniform(size=N)).astype(np.float32)
    pmax = r.normal(size=(T, N)).astype(np.float32)
    pmin = (pmax - np.abs(r.normal(size=(T, N)))).astype(np.float32)
    pamax = (r.randint(0, 128, size=(T, N)) + (np.arange(T) % tpf)[:, None] * 128).astype(np.int32)
    pamin = (r.randint(0, 128, size=(T, N)) + (np.arange(T) % tpf)[:, None] * 128).astype(np.int32)
    dead = r.uniform(size=(T, N)) < 0.3
    pamax[dead] = -1
    pamin[dead] = -1
    t = {k: _mk(dev, v) for k, v in dict(psum=psum, psumsq=psumsq, gamma=gamma, beta=beta, decay=np.array([0.7], np.float32),
                                         pmax=pmax, pmin=pmin, pamax=pamax, pamin=pamin).items()}
    res = []
    for fused in (False, True):
        o = dict(scale=torch.zeros(N, device=dev), shift=torch.zeros(N, device=dev), mean=torch.zeros(N, device=dev),
                 invstd=torch.zeros(N, device=dev), mm=_mk(dev, mm.copy()), mv=_mk(dev, mv.copy()), pooled=torch.zeros(B, N, device=dev),
                 argidx=torch.zeros(B, N, dtype=torch.int32, device=dev), ysel=torch.zeros(B, N, device=dev))
        f = abi.BnFwdFinalizeArgs()
        f.psum, f.psumsq, f.n_tiles, f.count, f.N = fptr(t['psum']), fptr(t['psumsq']), T, M, N
        f.gamma, f.beta, f.moving_mean, f.moving_var, f.decay = fptr(t['gamma']), fptr(t['beta']), fptr(o['mm']), fptr(o['mv']), fptr(t['decay'])
        f.eps, f.is_training, f.unbiased_ema = 1e-3, 1, 1
        f.scale, f.shift, f.mean, f.invstd = fptr(o['scale']), fptr(o['shift']), fptr(o['mean']), fptr(o['invstd'])
        if fused:
            f.pool_pmax, f.pool_pmin, f.pool_pamax, f.pool_pamin = fptr(t['pmax']), fptr(t['pmin']), iptr(t['pamax']), iptr(t['pamin'])
            f.pool_B, f.pool_tiles_per_frustum, f.pooled, f.ld_pooled, f.argidx, f.ysel = B, tpf, fptr(o['pooled']), N, iptr(o['argidx']), fptr(o['ysel'])
        assert hip_lib.t3d_bn_fwd_finalize(C.byref(f), st) == 0
        if not fused:
            q = abi.PoolFinalizeArgs()
            q.scale, q.shift, q.pmax, q.pmin, q.pamax, q.pamin = fptr(o['scale']), fptr(o['shift']), fptr(t['pmax']), fptr(t['pmin']), iptr(t['pamax']), iptr(t['pamin'])
            q.B, q.N, q.tiles_per_frustum, q.pooled, q.ld_pooled, q.argidx, q.ysel = B, N, tpf, fptr(o['pooled']), N, iptr(o['argidx']), fptr(o['ysel'])
            assert hip_lib.t3d_pool_finalize(C.byref(q), st) == 0
        torch.cuda.synchronize()
        res.append(o)
    for k in res[0]:
        assert torch.equal(res[0][k], res[1][k]), k
    o = {k: v.cpu().numpy().astype(np.float64) for k, v in res[1].items()}
    n = float(M)
    mean = psum.astype(np.float64).sum(0) / n
    var = np.maximum(psumsq.astype(np.float64).sum(0) / n - mean * mean, 0.0)
    invstd = 1.0 / np.sqrt(var + 1e-3)
    sc = gamma.astype(np.float64) * invstd
    for k, ref in dict(mean=mean, invstd=invstd, scale=sc, shift=beta - mean * sc, mm=mm * 0.7 + mean * (1 - 0.7),
                       mv=mv * 0.7 + var * n / (n - 1) * (1 - 0.7)).items():
        assert float(np.abs(o[k] - ref).max()) < 2e-6 * float(np.abs(ref).max() + 1), k
    # dense backward
    s1 = r.normal(size=(T, N)).astype(np.float32)
    s2 = r.normal(size=(T, N)).astype(np.float32)
    mean_b, invstd_b = r.normal(size=N).astype(np.float32), (0.5 + r.uniform(size=N)).astype(np.float32)
    tb = {k: _mk(dev, v) for k, v in dict(s1=s1, s2=s2, mean=mean_b, invstd=invstd_b, gamma=gamma).items()}
    ob = dict(coef=torch.zeros(3, N, device=dev), dgamma=torch.zeros(N, device=dev), dbeta=torch.zeros(N, device=dev))
    a = abi.BnBwdFinalizeArgs()
    a.psum_dz, a.psum_dzy, a.n_tiles, a.count, a.N = fptr(tb['s1']), fptr(tb['s2']), T, M, N
    a.gamma, a.mean, a.invstd = fptr(tb['gamma']), fptr(tb['mean']), fptr(tb['invstd'])
    a.dgamma, a.dbeta, a.coef = fptr(ob['dgamma']), fptr(ob['dbeta']), fptr(ob['coef'])
    assert hip_lib.t3d_bn_bwd_finalize(C.byref(a), st) == 0
    torch.cuda.synchronize()
    S1, S2 = s1.astype(np.float64).sum(0), s2.astype(np.float64).sum(0)
    dbeta = S1
    dgamma = invstd_b * (S2 - mean_b.astype(np.float64) * S1)
    c1 = gamma.astype(np.float64) * invstd_b
    k3 = dgamma / n * invstd_b
    ref = dict(dbeta=dbeta, dgamma=dgamma, coef=np.stack([c1, -c1 * k3, c1 * (k3 * mean_b - dbeta / n)]))
    for k, v in ref.items():
        got = ob[k].cpu().numpy().astype(np.float64)
        assert float(np.abs(got - v).max()) < 2e-6 * float(np.abs(v).max() + 1e-3), k


def _sweep_cases(seed, n):
    """Seeded random shapes on the contract of t3d.h (M, rows_per_frustum multiples of 128; widths multiples of 64; K any)."""
    r = np.random.RandomState(seed)
    out = []
    for _ in range(n):
        rpf = int(r.choice([128, 256, 384]))
        M = rpf * int(r.randint(1, 5))
        K = int(r.choice([3, 4, 64, 128, 192, 256, 320, 512]))
        N = int(r.choice([64, 128, 192, 256, 384, 512]))
        out.append((M, K, N, rpf))
    return out


@pytest.mark.parametrize('M,K,N,rpf', _sweep_cases(2024, 14))
def test_pointmlp_shape_sweep_fwd_bwd_against_spec(hip_lib, gemm_arithmetic, M, K, N, rpf):
    """Tile-edge coverage: widths that are not multiples of the 128-wide tile, reduction lengths that are not multiples of
    the 32-deep k-tile, single-tile grids; forward, fused backward (when K % 64 == 0) or weight gradient alone."""
    r = np.random.RandomState(M + 7 * K + 13 * N)
    B, T = M // rpf, M // 128
    ldx = 4 if K <= 4 else K
    d = dict(x=r.normal(size=(M, ldx)).astype(np.float32), sc=(0.5 + r.uniform(size=max(K, 4))).astype(np.float32),
             sh=(r.normal(size=max(K, 4)) * 0.3).astype(np.float32), w=(r.normal(size=(K, N)) / np.sqrt(K)).astype(np.float32),
             bias=(r.normal(size=N) * 0.1).astype(np.float32), dz=(r.normal(size=(M, N)) * 1e-2).astype(np.float32),
             y=r.normal(size=(M, N)).astype(np.float32), coef=r.normal(size=(3, N)).astype(np.float32))
    bn = K > 4

    def act(t):
        return abi.ActSrc(fptr(t['x']), ldx, 0, fptr(t['sc'] if bn else None), fptr(t['sh'] if bn else None), int(bn), fptr(None), 0)

    def make_f(dev):
        t = {k: _mk(dev, v) for k, v in d.items()}
        o = dict(y=torch.zeros(M, N, device=dev), psum=torch.zeros(T, N, device=dev), psumsq=torch.zeros(T, N, device=dev))
        a = abi.PointMlpFwdArgs()
        a.a, a.w, a.bias, a.y, a.psum, a.psumsq = act(t), fptr(t['w']), fptr(t['bias']), fptr(o['y']), fptr(o['psum']), fptr(o['psumsq'])
        a.M, a.K, a.N, a.rows_per_frustum = M, K, N, rpf
        a._keep = (t, o)
        return a, o

    c, g = _run_both(hip_lib, make_f, 't3d_pointmlp_fwd')
    _close(c['y'], g['y'], 2e-5, 3e-5, 'y')
    _close(c['psum'], g['psum'], 1e-4, 3e-3, 'psum')

    rps, tk, tn = C.c_int(0), C.c_int(0), C.c_int(0)
    assert hip_lib.t3d_wgrad_plan(M, K, N, C.byref(rps), C.byref(tk), C.byref(tn)) == 0
    S = M // rps.value

    def make_w(dev):
        t = {k: _mk(dev, v) for k, v in d.items()}
        o = dict(slabs=torch.zeros(S, K, N, device=dev))
        dy = abi.DySrc(fptr(t['dz']), fptr(t['y']), fptr(t['coef']), iptr(None), fptr(None))
        a = abi.PointMlpWgradArgs(act(t), dy, fptr(o['slabs']), M, K, N, rpf, rps.value)
        a._keep = (t, o)
        return a, o

    c, g = _run_both(hip_lib, make_w, 't3d_pointmlp_wgrad')
    _close(c['slabs'].sum(0), g['slabs'].sum(0), 1e-4, 3e-5 * float(c['slabs'].sum(0).abs().max()), 'dW')
    if K % 64:
        return

    def make_d(dev):
        t = {k: _mk(dev, v) for k, v in d.items()}
        o = dict(out=torch.zeros(M, K, device=dev), s1=torch.zeros(T, K, device=dev), s2=torch.zeros(T, K, device=dev))
        a = abi.PointMlpDgradArgs()
        a.dy = abi.DySrc(fptr(t['dz']), fptr(t['y']), fptr(t['coef']), iptr(None), fptr(None))
        a.w, a.out, a.prev_y, a.prev_scale, a.prev_shift = fptr(t['w']), fptr(o['out']), fptr(t['x']), fptr(t['sc']), fptr(t['sh'])
        a.psum_dz, a.psum_dzy, a.M, a.K, a.N, a.rows_per_frustum = fptr(o['s1']), fptr(o['s2']), M, K, N, rpf
        a._keep = (t, o)
        return a, o

    c, g = _run_both(hip_lib, make_d, 't3d_pointmlp_dgrad')
    scale = float(c['out'].abs().max())
    _close(c['out'], g['out'], 1e-4, 3e-5 * scale, 'dX')
    _close(c['s1'], g['s1'], 1e-3, 1e-3 * scale, 'psum_dz')


def _iou_boxes(r, n):
    c1 = r.normal(0, 1.0, size=(n, 3)) + np.array([0, 0, 3.0])
    s1 = r.uniform(0.3, 2.5, size=(n, 3))
    h1 = r.uniform(-np.pi, np.pi, size=n)
    c2 = c1 + r.normal(0, 0.3, size=(n, 3))
    s2 = s1 * r.uniform(0.7, 1.3, size=(n, 3))
    h2 = h1 + r.uniform(0, 0.8, size=n)
    far = r.uniform(size=n) < 0.25                                     # a quarter of the pairs unrelated (many disjoint)
    c2[far] = r.normal(0, 1.5, size=(int(far.sum()), 3)) + np.array([0, 0, 3.0])
    h2[far] = r.uniform(-np.pi, np.pi, size=int(far.sum()))
    # degenerate rows: identical, half turn, shared edge, negative l and w
    c2[0], s2[0], h2[0] = c1[0], s1[0], h1[0]
    c2[1], s2[1], h2[1] = c1[1], s1[1], h1[1] + np.pi
    c1[2], s1[2], h1[2], c2[2], s2[2], h2[2] = [0, 0, 0], [1, 1, 1], 0.0, [1, 0, 0], [1, 1, 1], 0.0
    c2[3], s2[3], h2[3] = c1[3], s1[3] * [-1, -1, 1], h1[3]
    return [a.astype(np.float32) for a in (c1, s1, h1, c2, s2, h2)]


def test_box3d_iou_parameter_and_corner_forms_against_oracle(hip_lib):
    """t3d_box3d_iou / t3d_box3d_iou_corners (fp32, boundary integral) against the oracle's Sutherland-Hodgman restatement of
    box_util.box3d_iou (fp64) on the same boxes.  Tolerance 2e-5 absolute on an IoU in [0,1]."""
    from oracle import ref_box as RB
    r = np.random.RandomState(11)
    n = 1000
    c1, s1, h1, c2, s2, h2 = _iou_boxes(r, n)
    dev = _dev('cuda')
    t = [_mk(dev, a) for a in (c1, s1, h1, c2, s2, h2)]
    i3, i2 = torch.zeros(n, device=dev), torch.zeros(n, device=dev)
    a = abi.Box3dIouArgs(*[fptr(x) for x in t], fptr(i3), fptr(i2), n)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    assert hip_lib.t3d_box3d_iou(C.byref(a), st) == 0
    k1 = np.stack([RB.get_3d_box(np.abs(s1[i]).astype(np.float64), float(h1[i]), c1[i].astype(np.float64)) for i in range(n)])
    k2 = np.stack([RB.get_3d_box(np.abs(s2[i]).astype(np.float64), float(h2[i]), c2[i].astype(np.float64)) for i in range(n)])
    tk1, tk2 = _mk(dev, k1.astype(np.float32)), _mk(dev, k2.astype(np.float32))
    j3, j2 = torch.zeros(n, device=dev), torch.zeros(n, device=dev)
    b = abi.Box3dIouCornersArgs(fptr(tk1), fptr(tk2), fptr(j3), fptr(j2), n)
    assert hip_lib.t3d_box3d_iou_corners(C.byref(b), st) == 0
    torch.cuda.synchronize()
    i3, i2, j3, j2 = [x.cpu().numpy().astype(np.float64) for x in (i3, i2, j3, j2)]
    want = np.array([RB.box3d_iou(k1[i], k2[i]) for i in range(4, n)])
    assert (want[:, 0] > 0.3).sum() > 300 and (want[:, 0] == 0).sum() > 50
    for got3, got2, what in ((i3, i2, 'params'), (j3, j2, 'corners')):
        assert np.abs(got3[4:] - want[:, 0]).max() < 2e-5, what
        assert np.abs(got2[4:] - want[:, 1]).max() < 2e-5, what
        assert abs(got3[0] - 1) < 1e-5 and abs(got3[1] - 1) < 1e-4 and got3[2] < 1e-5 and abs(got3[3] - 1) < 1e-5, (what, got3[:4])
        assert np.all(got3 >= 0) and np.all(got3 <= 1 + 1e-5) and np.all(got3 <= got2 + 1e-6)


def test_box_head_iou_and_strong_loss_summary(hip_lib):
    """t3d_box_head_iou == the IoU outputs of t3d_strong_loss == oracle compute_box3d_iou on the decoded heads."""
    from oracle import ref_box as RB
    from transferable3d_amd.constants import MEAN_DIMS_ARR
    r = np.random.RandomState(5)
    B = 48
    box = r.normal(size=(B, 67)).astype(np.float32) * 0.3
    s1c = r.normal(size=(B, 3)).astype(np.float32) * 0.2
    yc = (s1c + r.normal(size=(B, 3)) * 0.2).astype(np.float32)
    yoc, ydc = r.randint(0, 12, B).astype(np.int32), r.randint(0, 10, B).astype(np.int32)
    yor, ydr = (r.uniform(-1, 1, B) * np.pi / 12).astype(np.float32), (r.normal(size=(B, 3)) * 0.1).astype(np.float32)
    # make half of the predictions agree with the label bins so that the IoUs are not all tiny
    for b in range(0, B, 2):
        box[b, 3 + yoc[b]] = 5.0
        box[b, 27 + ydc[b]] = 5.0

    def make(dev):
        t = {k: _mk(dev, v) for k, v in dict(box=box, s1=s1c, yc=yc, yoc=yoc, yor=yor, ydc=ydc, ydr=ydr).items()}
        o = dict(i2=torch.zeros(B, device=dev), i3=torch.zeros(B, device=dev))
        a = abi.BoxHeadIouArgs(fptr(t['box']), 67, fptr(t['s1']), fptr(t['yc']), iptr(t['yoc']), fptr(t['yor']), iptr(t['ydc']),
                               fptr(t['ydr']), fptr(o['i2']), fptr(o['i3']), B)
        a._keep = (t, o)
        return a, o

    c, g = _run_both(hip_lib, make, 't3d_box_head_iou')
    _close(c['i3'], g['i3'], 0, 2e-5, 'head iou3d')
    _close(c['i2'], g['i2'], 0, 2e-5, 'head iou2d')
    mean = MEAN_DIMS_ARR.astype(np.float64)
    hres = box[:, 15:27].astype(np.float64) * (np.pi / 12)
    sres = box[:, 37:67].astype(np.float64).reshape(B, 10, 3) * mean[None]
    w2, w3 = RB.compute_box3d_iou(box[:, 0:3].astype(np.float64) + s1c, box[:, 3:15], hres, box[:, 27:37], sres, yc.astype(np.float64), yoc,
                                  yor.astype(np.float64), ydc, ydr.astype(np.float64))
    assert np.abs(g['i3'].numpy() - w3).max() < 2e-5 and np.abs(g['i2'].numpy() - w2).max() < 2e-5
    assert (w3 > 0.2).sum() >= 5


def test_seg_head_inline_dropout_equals_the_stored_mask(hip_lib):
    """drop_mask == NULL with a seed: the head draws the keep mask itself; bit-identical to running it on the mask tensor that
    t3d_dropout_mask writes for the same (seed, step)."""
    r = np.random.RandomState(2)
    M, K, rpf = 1024, 128, 256
    B, T = M // rpf, M // 128
    dev = _dev('cuda')
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    t = {k: _mk(dev, v) for k, v in dict(
        y=r.normal(size=(M, K)).astype(np.float32), sc=(0.5 + r.uniform(size=K)).astype(np.float32), sh=(r.normal(size=K) * 0.1).astype(np.float32),
        w=(r.normal(size=(K, 2)) * 0.2).astype(np.float32), b=np.zeros(2, np.float32), lab=(r.uniform(size=M) < 0.3).astype(np.int32),
        is2d=np.zeros(B, np.int32), pc=r.normal(size=(M, 4)).astype(np.float32)).items()}
    hyper = torch.tensor([7.0, 0, 0, 0], device=dev)
    mask = torch.zeros(M, K, device=dev)
    assert hip_lib.t3d_dropout_mask(fptr(mask), M * K, 0.5, 4321, fptr(hyper), st) == 0
    assert 0.45 < float(mask.mean()) < 0.55

    def run(inline):
        o = dict(logits=torch.zeros(M, 2, device=dev), mask=torch.zeros(M, device=dev), part=torch.zeros(T, 8, device=dev),
                 dz=torch.zeros(M, K, device=dev), p1=torch.zeros(T, K, device=dev), p2=torch.zeros(T, K, device=dev),
                 dw=torch.zeros(T, K, 2, device=dev))
        a = abi.SegHeadArgs()
        a.y, a.scale, a.shift, a.keep_prob, a.w, a.bias = fptr(t['y']), fptr(t['sc']), fptr(t['sh']), 0.5, fptr(t['w']), fptr(t['b'])
        a.labels, a.is_data_2D, a.pc, a.ld_pc, a.ce_weight = iptr(t['lab']), iptr(t['is2d']), fptr(t['pc']), 4, 1.0
        a.logits, a.mask, a.part, a.dz, a.psum_dz, a.psum_dzy, a.dw_part = fptr(o['logits']), fptr(o['mask']), fptr(o['part']), fptr(o['dz']), \
            fptr(o['p1']), fptr(o['p2']), fptr(o['dw'])
        a.M, a.K, a.rows_per_frustum, a.B = M, K, rpf, B
        if inline:
            a.drop_seed, a.drop_hyper = 4321, fptr(hyper)
        else:
            a.drop_mask = fptr(mask)
        assert hip_lib.t3d_seg_head(C.byref(a), st) == 0
        torch.cuda.synchronize()
        return {k: v.cpu() for k, v in o.items()}
    a, b = run(True), run(False)
    for k in a:
        assert torch.equal(a[k], b[k]), k
    # the specification draws the same mask
    from fake_t3d import hash_keep_mask
    assert np.array_equal(hash_keep_mask(4321, 7, M * K, 0.5).reshape(M, K), mask.cpu().numpy())


def _x3_frag_planes(hip_lib, w, st):
    """(forward planes, data-gradient planes, plane stride) of one [K, N] matrix through t3d_split_x3_frag (a one-entry device table)."""
    K, N = w.shape
    stride = (K * N + 7) // 8 * 8
    pf = torch.zeros(3 * stride, dtype=torch.bfloat16, device=w.device)
    pd = torch.zeros(3 * stride, dtype=torch.bfloat16, device=w.device)
    raw, nblk = abi.x3_frag_table([(0, K, N)])
    tab = torch.from_numpy(raw).to(w.device)
    assert hip_lib.t3d_split_x3_frag(fptr(w), C.c_void_p(pf.data_ptr()), C.c_void_p(pd.data_ptr()), stride, C.c_void_p(tab.data_ptr()), 1, nblk, st) == 0
    torch.cuda.synchronize()
    return pf, pd, stride


def test_x3_weight_planes_are_exact_and_equal_the_in_kernel_split(hip_lib):
    """t3d_split_x3: every fp32 value is EXACTLY the sum of its three bf16 planes; a forward / data-gradient launch that reads the
    pre-split planes (t3d_pointmlp_fwd_args.w_x3) gives bit for bit what the launch that splits the fp32 matrix while staging it gives."""
    dev, st = _dev('cuda'), C.c_void_p(torch.cuda.current_stream().cuda_stream)
    r = np.random.RandomState(5)
    n = 4099
    src = torch.as_tensor((r.normal(size=n) * np.exp(r.uniform(-20, 20, size=n))).astype(np.float32)).to(dev)
    stride = 4104
    planes = torch.zeros(3 * stride, dtype=torch.bfloat16, device=dev)
    assert hip_lib.t3d_split_x3(fptr(src), C.c_void_p(planes.data_ptr()), n, stride, st) == 0
    torch.cuda.synchronize()
    p = planes.view(3, stride)[:, :n].double().cpu()
    assert torch.equal(p[0] + p[1] + p[2], src.double().cpu())                   # exact
    assert float((p[1].abs() / p[0].abs().clamp_min(1e-300)).max()) <= 2.0 ** -7   # each term below the previous one's last bit
    _planes_equal_in_kernel_split(hip_lib, dev, st, r, 512, 256, 128)


@pytest.mark.parametrize('M,K,N', [(32768, 256, 256), (8192, 64, 512)])
def test_x3_fragment_planes_equal_the_in_kernel_split_on_the_wide_tiles_and_in_the_fused_backward(hip_lib, M, K, N):
    """... at sizes whose launches take the 128-wide forward / data-gradient tiles (M = 32768: two workgroups per CU) and the narrow-input
    wide-output backward, and through the fused t3d_pointmlp_bwd: bit for bit the launch that splits the weights while it stages them."""
    dev, st = _dev('cuda'), C.c_void_p(torch.cuda.current_stream().cuda_stream)
    _planes_equal_in_kernel_split(hip_lib, dev, st, np.random.RandomState(9), M, K, N, fused=True)


def _planes_equal_in_kernel_split(hip_lib, dev, st, r, M, K, N, fused=False):
    rpf = 256
    x, w = _mk(dev, r.normal(size=(M, K)).astype(np.float32)), _mk(dev, (r.normal(size=(K, N)) / 16).astype(np.float32))
    # fragment order (what the kernels take as w_x3): plane p, k-tile rt, 32-wide block nb, lane, eight elements -- and still exact
    pf, pd, fstride = _x3_frag_planes(hip_lib, w, st)
    wc = w.cpu()
    for planes, op in ((pf, wc.t().contiguous()), (pd, wc)):      # Op[n][k] = w[k][n] forward, Op[k][n] = w[k][n] data gradient
        L, R = op.shape                                              # lane index, reduction index
        fr = planes.view(3, fstride)[:, :K * N].double().cpu().sum(0).view(R // 16, L // 32, 2, 32, 8)      # [rt, nb, lane >> 5, lane & 31, j]
        back = fr.permute(1, 3, 0, 2, 4).reshape(L, R)               # [nb * 32 + (lane & 31)][rt * 16 + 8 * (lane >> 5) + j]
        assert torch.equal(back, op.double())
    outs = []
    for pre in (False, True):
        y = torch.zeros(M, N, device=dev)
        p1, p2 = torch.zeros(M // 128, N, device=dev), torch.zeros(M // 128, N, device=dev)
        a = abi.PointMlpFwdArgs()
        a.a = abi.ActSrc(fptr(x), K, 0, fptr(None), fptr(None), 0, fptr(None), 0)
        a.w, a.y, a.psum, a.psumsq = fptr(w), fptr(y), fptr(p1), fptr(p2)
        a.M, a.K, a.N, a.rows_per_frustum = M, K, N, rpf
        if pre:
            a.w_x3, a.w_x3_stride = pf.data_ptr(), fstride
        assert hip_lib.t3d_pointmlp_fwd(C.byref(a), st) == 0
        dz, yv = _mk(dev, r.normal(size=(M, N)).astype(np.float32) * 0 + 1e-2), _mk(dev, np.ones((M, N), np.float32))
        coef = _mk(dev, np.stack([np.ones(N), np.full(N, 0.5), np.zeros(N)]).astype(np.float32))
        out = torch.zeros(M, K, device=dev)
        d = abi.PointMlpDgradArgs()
        d.dy, d.w, d.out = abi.DySrc(fptr(dz), fptr(yv), fptr(coef), iptr(None), fptr(None)), fptr(w), fptr(out)
        d.M, d.K, d.N, d.rows_per_frustum = M, K, N, rpf
        if pre:
            d.w_x3, d.w_x3_stride = pd.data_ptr(), fstride
        assert hip_lib.t3d_pointmlp_dgrad(C.byref(d), st) == 0
        torch.cuda.synchronize()
        res = [y.cpu(), p1.cpu(), out.cpu()]
        if fused:      # data gradient + weight gradient in one launch (t3d_pointmlp_bwd): the data-gradient tiles read the planes
            rps, one = C.c_int(0), C.c_int(0)
            assert hip_lib.t3d_bwd_plan(M, K, N, 0, C.byref(rps), C.byref(one)) == 0
            slabs, out2 = torch.zeros(M // rps.value, K, N, device=dev), torch.zeros(M, K, device=dev)
            d.out = fptr(out2)
            wa = abi.PointMlpWgradArgs()
            wa.a = abi.ActSrc(fptr(x), K, 0, fptr(None), fptr(None), 0, fptr(None), 0)
            wa.dy, wa.slabs = d.dy, fptr(slabs)
            wa.M, wa.K, wa.N, wa.rows_per_frustum, wa.rows_per_split = M, K, N, rpf, rps.value
            assert hip_lib.t3d_pointmlp_bwd(C.byref(d), C.byref(wa), st) == 0
            torch.cuda.synchronize()
            res += [out2.cpu(), slabs.sum(0).cpu()]
        outs.append(res)
    for a_, b_, what in zip(outs[0], outs[1], ('y', 'psum', 'dX', 'dX (fused)', 'dW (fused)')):
        assert torch.equal(a_, b_), what
    assert float(outs[0][0].abs().max()) > 0 and float(outs[0][2].abs().max()) > 0
    if fused:
        assert torch.equal(outs[0][2], outs[0][3]), 'fused data gradient == the stand-alone launch'


@pytest.mark.parametrize('sign', ['positive', 'random'])
def test_x3_weight_gradient_rounding_bias_is_bounded(hip_lib, sign):
    """The bf16 matrix instruction does not round its accumulation to nearest: every v_mfma_f32_32x32x16_bf16 result lies a little
    BELOW the exact value, whatever the signs (a floor-like truncation relative to the magnitude of the accumulator it adds to).  The
    x3 weight gradient therefore carries a systematic negative error the fp32 MFMA (an fma chain, bitwise) does not have; it grows
    with the length of ONE accumulation chain (linearly for same-sign terms, where the accumulator grows with the chain; like
    n^1.5 for random signs), so what bounds it in the product is the row split of the weight gradient (t3d_wgrad_plan: chains of at
    most 1024 rows at M = 32768; the slabs are then summed to nearest by t3d_reduce_slabs).  Measured at M = 32768, K = N = 128
    against fp64, in units of the mean |dW| (round 5, MI355X):
                               plan's split (512-row chains)     ONE 32768-row chain        fp32 MFMA (plan's split)
        same-sign operands     mean -1.8e-8, rms 1.1e-7          mean -6.3e-7, rms 1.6e-6   mean -5e-11, rms 1.1e-7
        random-sign operands   mean -2.8e-7, rms 5.4e-7          mean -1.9e-6, rms 3.8e-6   mean -4e-9,  rms 5.4e-7
    (random signs: the sums cancel, the offsets follow the accumulator's excursions).  What this test pins, at the plan's own split:
    the offset is negative, below the random rounding error of the same gradient (|mean| <= 0.75 rms) and below 1e-6 of the mean |dW|
    (an Adam step divides the gradient by its running magnitude: a 3e-7 relative offset is that fraction of lr); the rms error is the
    fp32-MFMA kernel's; the fp32-MFMA kernel shows no offset; and the split is what bounds it (one long chain is worse).  The fix that
    exists -- every other k-tile multiplied with a negated operand into a second accumulator set, subtracted at the end, so that the
    offsets cancel -- costs 16 VGPRs per 32 x 32 tile: the 128 x 128 weight-gradient tiles (227 VGPRs) have no room for it, and it is
    not built."""
    import os
    dev = 'cuda'
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    torch.manual_seed(3)
    M, K, N, rpf = 32768, 128, 128, 1024
    x = torch.rand(M, K, device=dev) + 0.5 if sign == 'positive' else torch.randn(M, K, device=dev)
    dz = (torch.rand(M, N, device=dev) + 0.5 if sign == 'positive' else torch.randn(M, N, device=dev)) / M
    yv = torch.zeros(M, N, device=dev)
    coef = torch.zeros(3, N, device=dev)
    coef[0] = 1.0
    ref = x.double().t() @ dz.double()
    rps, tk, tn = C.c_int(0), C.c_int(0), C.c_int(0)
    assert hip_lib.t3d_wgrad_plan(M, K, N, C.byref(rps), C.byref(tk), C.byref(tn)) == 0
    assert rps.value <= 1024, 'the plan bounds the accumulation chains of the weight gradient'

    def run(arith, rows_per_split):
        slabs = torch.zeros(M // rows_per_split, K, N, device=dev)
        wa = abi.PointMlpWgradArgs()
        wa.a = abi.ActSrc(fptr(x), K, 0, fptr(None), fptr(None), 0, fptr(None), 0)
        wa.dy, wa.slabs = abi.DySrc(fptr(dz), fptr(yv), fptr(coef), iptr(None), fptr(None)), fptr(slabs)
        wa.M, wa.K, wa.N, wa.rows_per_frustum, wa.rows_per_split, wa.arith = M, K, N, rpf, rows_per_split, arith
        assert hip_lib.t3d_gemm_arithmetic(arith, abi.F32, K, N, 1) == arith
        assert hip_lib.t3d_pointmlp_wgrad(C.byref(wa), st) == 0
        torch.cuda.synchronize()
        dw = slabs[0].clone()
        for i in range(1, slabs.shape[0]):      # t3d_reduce_slabs: fixed order, fp32
            dw += slabs[i]
        e = (dw.double() - ref) / ref.abs().mean()
        return float(e.mean()), float(e.pow(2).mean().sqrt())

    mean_x3, rms_x3 = run(abi.ARITH_BF16X3, rps.value)
    mean_f32, rms_f32 = run(abi.ARITH_FP32_MFMA, rps.value)
    mean_one, rms_one = run(abi.ARITH_BF16X3, M)
    print('x3 wgrad bias (%s): plan split %d rows  x3 mean %+.2e rms %.2e | fp32-MFMA mean %+.2e rms %.2e | one 32768-row chain x3 mean %+.2e rms %.2e'
          % (sign, rps.value, mean_x3, rms_x3, mean_f32, rms_f32, mean_one, rms_one))
    assert mean_x3 < 0 and abs(mean_x3) <= 0.75 * rms_x3 and abs(mean_x3) <= 1e-6, (mean_x3, rms_x3)
    assert rms_x3 <= 1.1 * rms_f32, (rms_x3, rms_f32)
    assert abs(mean_f32) <= 0.05 * rms_f32, (mean_f32, rms_f32)      # the fma chain has no such offset
    assert mean_one < 0 and abs(mean_one) > 2.0 * abs(mean_x3) and abs(mean_one) <= 0.75 * rms_one, (mean_one, mean_x3, rms_one)


@pytest.mark.parametrize('K,N', [(128, 256), (64, 64), (96, 512)])
def test_x3_forward_variants_are_bit_identical_to_the_default(hip_lib, monkeypatch, K, N):
    """Two opt-in forms of the x3 forward kernel reproduce the default bit for bit -- every output, statistics and pool partials
    included: the weights pre-split into three bf16 planes (t3d_split_x3 + w_x3, 16-byte copies instead of the in-kernel split) and
    the producer / consumer workgroups of round 5's experiment (T3D_X3_PC=1: eight waves, three LDS stages), and the eight-wave
    128 x 256 tiles that launches of two or more rounds take by default (T3D_X3_W8; forced here at a small M, N % 256 == 0 only)."""
    dev = 'cuda'
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    torch.manual_seed(7)
    M, rpf = 1024, 256
    T = M // 128
    x = torch.randn(M, K, device=dev)
    sc, sh = torch.rand(K, device=dev) + 0.5, torch.randn(K, device=dev) * 0.1
    w = torch.randn(K, N, device=dev) / K ** 0.5
    bias = torch.randn(N, device=dev) * 0.1
    pf, _, fstride = _x3_frag_planes(hip_lib, w, st)
    res = {}
    modes = ('default', 'presplit', 'producer_consumer') + (('eight_waves', 'presplit_eight_waves') if N % 256 == 0 else ())
    for mode in modes:
        monkeypatch.setenv('T3D_X3_PC', '2' if mode == 'producer_consumer' else '0')
        monkeypatch.setenv('T3D_X3_W8', '2' if mode in ('eight_waves', 'presplit_eight_waves') else '0')
        o = [torch.zeros(M, N, device=dev)] + [torch.zeros(T, N, device=dev) for _ in range(4)] + \
            [torch.zeros(T, N, dtype=torch.int32, device=dev) for _ in range(2)]
        a = abi.PointMlpFwdArgs()
        a.a = abi.ActSrc(fptr(x), K, 0, fptr(sc), fptr(sh), 1, fptr(None), 0)
        a.w, a.bias, a.y, a.psum, a.psumsq = fptr(w), fptr(bias), fptr(o[0]), fptr(o[1]), fptr(o[2])
        a.pmax, a.pmin, a.pamax, a.pamin = fptr(o[3]), fptr(o[4]), iptr(o[5]), iptr(o[6])
        a.M, a.K, a.N, a.rows_per_frustum, a.arith = M, K, N, rpf, abi.ARITH_BF16X3
        if mode in ('presplit', 'presplit_eight_waves'):
            a.w_x3, a.w_x3_stride = C.c_void_p(pf.data_ptr()), fstride
        assert hip_lib.t3d_pointmlp_fwd(C.byref(a), st) == 0
        torch.cuda.synchronize()
        res[mode] = o
    for mode in modes[1:]:
        for u, v in zip(res['default'], res[mode]):
            assert torch.equal(u, v), mode
    ref = torch.relu(x.double() * sc.double() + sh.double()) @ w.double() + bias.double()
    assert float((res['default'][0].double() - ref).abs().max() / ref.abs().max()) < 2e-6

