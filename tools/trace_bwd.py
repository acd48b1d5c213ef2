#!/usr/bin/env python3
"""Per-workgroup timeline of the fused backward launch (t3d_pointmlp_bwd: weight-gradient tiles first, data-gradient tiles behind them) on
the layer shapes of the hot path; needs a `-DT3D_TRACE -DT3D_TRACE_STRIDE=8` build (tools/build_variant.sh trace8 "...").  Per shape: how
many workgroups of each kind, how long one lives, when the slots turn over, what the last ones to finish are."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import torch
from transferable3d_amd import abi
from transferable3d_amd.abi import fptr, iptr
from bench_x3 import frag_planes

SHAPES = [(512, 256), (256, 128), (128, 256), (128, 128), (64, 512), (64, 128), (64, 64)]


def main():
    lib = abi.load(os.environ.get('T3D_LIB', 'tools/libt3d_trace8.so'))
    lib.t3d_set_trace.argtypes = [C.c_void_p]
    M, rpf, ST = int(os.environ.get('T3D_M', '32768')), 1024, 8
    dev = 'cuda'
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    os.environ['T3D_X3'] = '1'
    os.environ['T3D_X3_MINKN'] = '1'
    for K, N in SHAPES:
        T = M // 128
        x = torch.randn(M, K, device=dev)
        sc, sh = torch.rand(K, device=dev) + 0.5, torch.randn(K, device=dev) * 0.1
        w = torch.randn(K, N, device=dev) / K ** 0.5
        yv, dz, coef = torch.randn(M, N, device=dev), torch.randn(M, N, device=dev) * 1e-2, torch.randn(3, N, device=dev)
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
        if K % 32 == 0:
            keep = frag_planes(lib, w, s)
            d.w_x3, d.w_x3_stride = keep[1].data_ptr(), keep[2]
        wa = abi.PointMlpWgradArgs()
        wa.a = abi.ActSrc(fptr(x), K, 0, fptr(sc), fptr(sh), 1, fptr(None), 0)
        wa.dy, wa.slabs = d.dy, fptr(slabs)
        wa.M, wa.K, wa.N, wa.rows_per_frustum, wa.rows_per_split = M, K, N, rpf, rps.value
        nblk = 8192
        trace = torch.zeros(nblk * ST, dtype=torch.int64, device=dev)
        for _ in range(3):
            assert lib.t3d_pointmlp_bwd(C.byref(d), C.byref(wa), s) == 0
        torch.cuda.synchronize()
        assert lib.t3d_set_trace(C.c_void_p(trace.data_ptr())) == 0
        assert lib.t3d_pointmlp_bwd(C.byref(d), C.byref(wa), s) == 0
        torch.cuda.synchronize()
        assert lib.t3d_set_trace(C.c_void_p(0)) == 0
        tr = trace.cpu().numpy().reshape(nblk, ST)
        tr = tr[tr[:, 0] != 0]
        base = tr[:, 0].min()
        t0, t2, tp = (tr[:, 0] - base) / 100.0, (tr[:, 2] - base) / 100.0, (tr[:, 4] - base) / 100.0
        kind = tr[:, 1]
        cu = (tr[:, 3] >> 32 & 0xf) * 4096 + (tr[:, 3] & 0xff00) // 256
        print('K%-4d N%-4d rows_per_split %d: %d workgroups (%d weight-gradient, %d data-gradient) on %d CUs | last exit %.1f us'
              % (K, N, rps.value, len(tr), (kind == 1).sum(), (kind == 2).sum(), len(np.unique(cu)), t2.max()))
        for k, name in ((1, 'weight-gradient'), (2, 'data-gradient')):
            m = kind == k
            if not m.any():
                continue
            life = t2[m] - t0[m]
            print('        %-15s start %.1f .. %.1f us (p50 %.1f) | life %.1f us (min %.1f max %.1f) | prologue %.2f us | exits %.1f .. %.1f (p50 %.1f)'
                  % (name, t0[m].min(), t0[m].max(), np.median(t0[m]), life.mean(), life.min(), life.max(), np.mean(tp[m] - t0[m]),
                     t2[m].min(), t2[m].max(), np.median(t2[m])))
        # residency: workgroups alive over time (sampled), to see rounds and the tail
        ts = np.linspace(0, t2.max(), 9)[1:-1]
        print('        resident workgroups at ' + ', '.join('%.0f us: %d' % (t, ((t0 <= t) & (t2 > t)).sum()) for t in ts))


if __name__ == '__main__':
    main()
