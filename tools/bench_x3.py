#!/usr/bin/env python3
"""T3D_X3 (fp32 layers on the bf16 matrix pipe, three bf16 terms per operand) against the fp32-MFMA kernels: error of both against
an fp64 product of the same fp32 operands, and time per launch, on the shapes of the hot path (M = 32768)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from transferable3d_amd import abi
from transferable3d_amd.abi import fptr, iptr

FWD = [(512, 256, False), (256, 128, False), (128, 256, False), (256, 512, True), (128, 128, False), (128, 1024, True), (64, 512, False)]


_junk = None


def timed(fn, R=20):
    global _junk
    if os.environ.get('T3D_BENCH_COLD') == '1':      # every timed launch behind a 1 GB stream through the 256 MB memory-side cache: operands from HBM
        if _junk is None:
            _junk = (torch.zeros(128 << 20, device='cuda'), torch.zeros(128 << 20, device='cuda'))
        tot = 0.0
        for _ in range(R):
            _junk[1].copy_(_junk[0])
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); fn(); e1.record()
            torch.cuda.synchronize()
            tot += e0.elapsed_time(e1)
        return tot / R * 1e3
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


def frag_planes(lib, w, s):
    """(forward planes, data-gradient planes, stride) of one [K, N] matrix in fragment order (t3d_split_x3_frag)"""
    import numpy as np
    K, N = w.shape
    stride = (K * N + 7) // 8 * 8
    pf = torch.zeros(3 * stride, dtype=torch.bfloat16, device=w.device)
    pd = torch.zeros(3 * stride, dtype=torch.bfloat16, device=w.device)
    raw, nblk = abi.x3_frag_table([(0, K, N)])
    tab = torch.from_numpy(raw).to(w.device)
    assert lib.t3d_split_x3_frag(fptr(w), C.c_void_p(pf.data_ptr()), C.c_void_p(pd.data_ptr()), stride, C.c_void_p(tab.data_ptr()), 1, nblk, s) == 0
    torch.cuda.synchronize()
    return pf, pd, stride


def main():
    lib = abi.load(os.environ.get('T3D_LIB'))
    frag = os.environ.get('T3D_BENCH_FRAG', '1') == '1'      # the weights as fragment-order planes (the engine's default); 0: in-kernel split
    M, rpf = int(os.environ.get('T3D_M', '32768')), 1024
    T = M // 128
    dev = 'cuda'
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    torch.manual_seed(0)
    only = os.environ.get('T3D_ONLY')            # "fwd:512x256" / "bwd:128x128": that one case (for counter passes)
    for K, N, pooled in FWD:
        if only and only != 'fwd:%dx%d' % (K, N):
            continue
        pad = int(os.environ.get('T3D_LDPAD', '0'))      # experiment: leading dimension of the input K + pad floats (L2 / HBM channel spread)
        xbuf = torch.randn(M, K + pad, device=dev)
        x = xbuf[:, :K]
        sc, sh = torch.rand(K, device=dev) + 0.5, torch.randn(K, device=dev) * 0.1
        w = torch.randn(K, N, device=dev) / K ** 0.5
        bias = torch.randn(N, device=dev) * 0.1
        y = torch.zeros(M, N, device=dev)
        p1, p2 = torch.zeros(T, N, device=dev), torch.zeros(T, N, device=dev)
        pm = [torch.zeros(T, N, device=dev) for _ in range(2)] + [torch.zeros(T, N, dtype=torch.int32, device=dev) for _ in range(2)]
        a = abi.PointMlpFwdArgs()
        a.a = abi.ActSrc(fptr(xbuf), K + pad, 0, fptr(sc), fptr(sh), 1, fptr(None), 0)
        a.w, a.bias, a.psum, a.psumsq = fptr(w), fptr(bias), fptr(p1), fptr(p2)
        if pooled:
            a.pmax, a.pmin, a.pamax, a.pamin = fptr(pm[0]), fptr(pm[1]), iptr(pm[2]), iptr(pm[3])
        else:
            a.y = fptr(y)
        a.M, a.K, a.N, a.rows_per_frustum = M, K, N, rpf
        if frag and K % 32 == 0:
            keep = frag_planes(lib, w, s)
            a.w_x3, a.w_x3_stride = keep[0].data_ptr(), keep[2]
        act = torch.relu(x.double() * sc.double() + sh.double())
        ref = act[:4096] @ w.double() + bias.double()
        refsum = None
        out = {}
        for mode in ('0', '1'):
            os.environ['T3D_X3'] = mode
            os.environ['T3D_X3_MINKN'] = '1'
            y.zero_(); p1.zero_()
            assert lib.t3d_pointmlp_fwd(C.byref(a), s) == 0
            torch.cuda.synchronize()
            if pooled:      # no y: compare the column sums of the first 32 tiles
                want = (act[:4096] @ w.double() + bias.double()).reshape(32, 128, N).sum(1)
                err = float((p1[:32].double() - want).abs().max() / want.abs().max())
            else:
                err = float((y[:4096].double() - ref).abs().max() / ref.abs().max())
            us = timed(lambda: lib.t3d_pointmlp_fwd(C.byref(a), s))
            out[mode] = (err, us)
        fl = 2.0 * M * K * N
        print('fwd %4d -> %4d %s  fp32-MFMA: err %.2e %7.1f us (%5.1f TF/s)   x3: err %.2e %7.1f us (%5.1f TF/s)   speed-up %.2f' % (
            K, N, 'pool' if pooled else '    ', out['0'][0], out['0'][1], fl / out['0'][1] / 1e6, out['1'][0], out['1'][1], fl / out['1'][1] / 1e6,
            out['0'][1] / out['1'][1]))
    # ---- fused backward (data gradient + weight gradient slabs) ----
    for K, N in [(512, 256), (256, 128), (128, 256), (128, 128), (64, 512), (64, 128), (64, 64)]:
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
        if frag and K % 32 == 0:
            keep = frag_planes(lib, w, s)
            d.w_x3, d.w_x3_stride = keep[1].data_ptr(), keep[2]
        wa = abi.PointMlpWgradArgs()
        wa.a = abi.ActSrc(fptr(x), K, 0, fptr(sc), fptr(sh), 1, fptr(None), 0)
        wa.dy, wa.slabs = d.dy, fptr(slabs)
        wa.M, wa.K, wa.N, wa.rows_per_frustum, wa.rows_per_split = M, K, N, rpf, rps.value
        R0 = 2048
        dy64 = coef[0].double() * dz[:R0].double() + coef[1].double() * yv[:R0].double() + coef[2].double()
        gate = (x[:R0].double() * sc.double() + sh.double()) > 0
        ref_dx = torch.where(gate, dy64 @ w.double().t(), torch.zeros((), dtype=torch.float64, device=dev))
        act64 = torch.relu(x.double() * sc.double() + sh.double())
        dyall = coef[0].double() * dz.double() + coef[1].double() * yv.double() + coef[2].double()
        ref_dw = act64.t() @ dyall
        res = {}
        for mode in ('0', '1'):
            os.environ['T3D_X3'] = mode
            os.environ['T3D_X3_MINKN'] = '1'
            out.zero_(); slabs.zero_()
            bwd = lambda: lib.t3d_pointmlp_bwd(C.byref(d), C.byref(wa), s)
            if os.environ.get('T3D_BENCH_EMPTY_RIDERS') == '1':      # the rider-hosting kernel with an empty rider set: what hosting costs the GEMM by itself
                rs = abi.RiderSet()
                sync = torch.zeros(64, dtype=torch.int32, device=dev)
                rs.n_ops, rs.n_wg, rs.lds_bytes, rs.sync = 0, 0, 0, C.cast(C.c_void_p(sync.data_ptr()), C.POINTER(C.c_uint32))
                bwd = lambda: lib.t3d_pointmlp_bwd_r(C.byref(d), C.byref(wa), C.byref(rs), s)
            assert bwd() == 0
            torch.cuda.synchronize()
            e_dx = float((out[:R0].double() - ref_dx).abs().max() / ref_dx.abs().max())
            e_dw = float((slabs.double().sum(0) - ref_dw).abs().max() / ref_dw.abs().max())
            us = timed(bwd)
            res[mode] = (e_dx, e_dw, us)
        fl = 4.0 * M * K * N
        print('bwd %4d -> %4d (one_pass %d)  fp32-MFMA: dx %.1e dw %.1e %7.1f us (%5.1f TF/s)   x3: dx %.1e dw %.1e %7.1f us (%5.1f TF/s)   speed-up %.2f' % (
            K, N, one.value, res['0'][0], res['0'][1], res['0'][2], fl / res['0'][2] / 1e6, res['1'][0], res['1'][1], res['1'][2], fl / res['1'][2] / 1e6,
            res['0'][2] / res['1'][2]))
    os.environ.pop('T3D_X3', None)
    os.environ.pop('T3D_X3_MINKN', None)


if __name__ == '__main__':
    main()
