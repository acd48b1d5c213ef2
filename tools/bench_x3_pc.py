#!/usr/bin/env python3
"""Variants of the x3 GEMM kernels against the default form: bit-identity of every output and time per launch on the layer shapes of
the hot path (M = 32768).  T3D_PC_MODES: comma list of 0 (default), 1 / 2 (producer / consumer waves, T3D_X3_PC), p (the weights
pre-split into three bf16 planes in fragment order, t3d_split_x3_frag + w_x3), w (eight-wave 128 x 256 forward tiles, T3D_X3_W8=2).  T3D_LIB: alternative library; T3D_ONLY=fwd:512x256 one case."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from transferable3d_amd import abi
from transferable3d_amd.abi import fptr, iptr

FWD = [(512, 256, False), (256, 128, False), (128, 256, False), (256, 512, True), (128, 128, False), (128, 1024, True), (64, 128, False), (64, 64, False)]
BWD = [(512, 256), (256, 128), (128, 256), (128, 128), (64, 128), (64, 64)]
MODES = os.environ.get('T3D_PC_MODES', '0,1').split(',')


def timed(fn, R=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(R):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / R * 1e3


def main():
    lib = abi.load(os.environ.get('T3D_LIB'))
    M, rpf = int(os.environ.get('T3D_M', '32768')), 1024
    T = M // 128
    dev = 'cuda'
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    torch.manual_seed(0)
    only = os.environ.get('T3D_ONLY')
    os.environ['T3D_X3'] = '1'
    for K, N, pooled in FWD:
        if only and only != 'fwd:%dx%d' % (K, N):
            continue
        x = torch.randn(M, K, device=dev)
        sc, sh = torch.rand(K, device=dev) + 0.5, torch.randn(K, device=dev) * 0.1
        w = torch.randn(K, N, device=dev) / K ** 0.5
        bias = torch.randn(N, device=dev) * 0.1
        y = torch.zeros(M, N, device=dev)
        p1, p2 = torch.zeros(T, N, device=dev), torch.zeros(T, N, device=dev)
        pm = [torch.zeros(T, N, device=dev) for _ in range(2)] + [torch.zeros(T, N, dtype=torch.int32, device=dev) for _ in range(2)]
        a = abi.PointMlpFwdArgs()
        a.a = abi.ActSrc(fptr(x), K, 0, fptr(sc), fptr(sh), 1, fptr(None), 0)
        a.w, a.bias, a.psum, a.psumsq = fptr(w), fptr(bias), fptr(p1), fptr(p2)
        if pooled:
            a.pmax, a.pmin, a.pamax, a.pamin = fptr(pm[0]), fptr(pm[1]), iptr(pm[2]), iptr(pm[3])
        else:
            a.y = fptr(y)
        a.M, a.K, a.N, a.rows_per_frustum = M, K, N, rpf
        outs, us = {}, {}
        from bench_x3 import frag_planes      # (round 6: `w_x3` = fragment-order planes, t3d_split_x3_frag)
        pf, pd, fstride = frag_planes(lib, w, s) if K % 32 == 0 else (None, None, 0)
        for mode in MODES:
            os.environ['T3D_X3_PC'] = mode if mode in '012' else '0'
            os.environ['T3D_X3_W8'] = '2' if mode == 'w' else '0'
            a.w_x3, a.w_x3_stride = (C.c_void_p(pf.data_ptr()), fstride) if (mode == 'p' and pf is not None) else (None, 0)
            for t_ in [y, p1, p2] + pm:
                t_.zero_()
            rc = lib.t3d_pointmlp_fwd(C.byref(a), s)
            assert rc == 0, rc
            torch.cuda.synchronize()
            outs[mode] = [t_.clone() for t_ in [y, p1, p2] + pm]
            us[mode] = timed(lambda: lib.t3d_pointmlp_fwd(C.byref(a), s))
        same = {m: all(torch.equal(u, v) for u, v in zip(outs[MODES[0]], outs[m])) for m in MODES[1:]}
        fl = 2.0 * M * K * N
        print('fwd %4d -> %4d %s ' % (K, N, 'pool' if pooled else '    ') +
              '  '.join('pc=%s %7.1f us (%5.1f TF/s)' % (m, us[m], fl / us[m] / 1e6) for m in MODES) + '   identical: %s' % same, flush=True)
    for K, N in BWD:
        if only and only != 'bwd:%dx%d' % (K, N):
            continue
        x = torch.randn(M, K, device=dev)
        sc, sh = torch.rand(K, device=dev) + 0.5, torch.randn(K, device=dev) * 0.1
        w = torch.randn(K, N, device=dev) / K ** 0.5
        yv = torch.randn(M, N, device=dev)
        dz = torch.randn(M, N, device=dev) * 1e-2
        coef = torch.randn(3, N, device=dev)
        out = torch.zeros(M, K, device=dev)
        p1, p2 = torch.zeros(T, K, device=dev), torch.zeros(T, K, device=dev)
        rps, one = C.c_int(0), C.c_int(0)
        assert lib.t3d_bwd_plan(M, K, N, 0, C.byref(rps), C.byref(one)) == 0
        slabs = torch.zeros(M // rps.value, K, N, device=dev)
        d = abi.PointMlpDgradArgs()
        d.dy = abi.DySrc(fptr(dz), fptr(yv), fptr(coef), iptr(None), fptr(None))
        d.w, d.out = fptr(w), fptr(out)
        d.prev_y, d.prev_scale, d.prev_shift, d.psum_dz, d.psum_dzy = fptr(x), fptr(sc), fptr(sh), fptr(p1), fptr(p2)
        d.M, d.K, d.N, d.rows_per_frustum = M, K, N, rpf
        wa = abi.PointMlpWgradArgs()
        wa.a = abi.ActSrc(fptr(x), K, 0, fptr(sc), fptr(sh), 1, fptr(None), 0)
        wa.dy, wa.slabs = d.dy, fptr(slabs)
        wa.M, wa.K, wa.N, wa.rows_per_frustum, wa.rows_per_split = M, K, N, rpf, rps.value
        outs, us = {}, {}
        from bench_x3 import frag_planes
        pf, pd, fstride = frag_planes(lib, w, s) if K % 32 == 0 else (None, None, 0)
        for mode in MODES:
            os.environ['T3D_X3_PC'] = mode if mode in '012' else '0'
            d.w_x3, d.w_x3_stride = (C.c_void_p(pd.data_ptr()), fstride) if (mode == 'p' and pd is not None) else (None, 0)
            for t_ in (out, slabs, p1, p2):
                t_.zero_()
            rc = lib.t3d_pointmlp_bwd(C.byref(d), C.byref(wa), s)
            assert rc == 0, rc
            torch.cuda.synchronize()
            outs[mode] = [t_.clone() for t_ in (out, slabs, p1, p2)]
            us[mode] = timed(lambda: lib.t3d_pointmlp_bwd(C.byref(d), C.byref(wa), s))
        same = {m: all(torch.equal(u, v) for u, v in zip(outs[MODES[0]], outs[m])) for m in MODES[1:]}
        fl = 4.0 * M * K * N
        print('bwd %4d -> %4d (one_pass %d) ' % (K, N, one.value) +
              '  '.join('pc=%s %7.1f us (%5.1f TF/s)' % (m, us[m], fl / us[m] / 1e6) for m in MODES) + '   identical: %s' % same, flush=True)
    os.environ.pop('T3D_X3_PC', None)
    os.environ.pop('T3D_X3_W8', None)
    os.environ.pop('T3D_X3', None)


if __name__ == '__main__':
    main()
