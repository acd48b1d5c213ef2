#!/usr/bin/env python3
"""Micro-benchmark of the three per-point GEMM kernels on the shapes of the hot path (B=32, N=1024).
Times R back-to-back launches between two events (no per-launch event overhead)."""
import ctypes as C
import sys
import os

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from transferable3d_amd import abi
from transferable3d_amd.abi import fptr, iptr

SHAPES = [('fwd', 128, 1024, True), ('fwd', 512, 256, False), ('fwd', 256, 512, True), ('fwd', 64, 512, False),
          ('fwd', 256, 128, False), ('fwd', 128, 256, False), ('fwd', 128, 128, False), ('fwd', 64, 128, False),
          ('fwd', 64, 64, False), ('fwd', 3, 128, False),
          ('dgrad', 128, 1024, True), ('dgrad', 512, 256, False), ('dgrad', 256, 512, True), ('dgrad', 64, 512, False),
          ('dgrad', 256, 128, False), ('dgrad', 128, 256, False), ('dgrad', 128, 128, False), ('dgrad', 64, 128, False),
          ('dgrad', 64, 64, False),
          ('wgrad', 128, 1024, True), ('wgrad', 512, 256, False), ('wgrad', 256, 512, True), ('wgrad', 64, 512, False),
          ('wgrad', 256, 128, False), ('wgrad', 128, 256, False), ('wgrad', 128, 128, False), ('wgrad', 64, 128, False),
          ('wgrad', 64, 64, False), ('wgrad', 3, 128, False)]


def main():
    lib = abi.load(os.environ.get('T3D_LIB'))
    M, rpf, R = int(os.environ.get('T3D_M', '32768')), 1024, 20     # T3D_M: more row tiles (how much of a launch is ramp / phase lock-step)
    B, T = M // rpf, M // 128
    dev = 'cuda'
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    only = sys.argv[1] if len(sys.argv) > 1 else None
    big = os.environ.get('T3D_BIG')
    tot = {}
    for kind, K, N, pooled in SHAPES:
        if only and kind != only:
            continue
        if big and 2.0 * K * N < 2 * 128 * 256:
            continue
        ldx = 4 if K <= 4 else K
        x = torch.randn(M, ldx, device=dev)
        sc, sh = torch.rand(K, device=dev) + 0.5, torch.randn(K, device=dev) * 0.1
        w = torch.randn(K, N, device=dev) / max(K, 1) ** 0.5
        y = torch.randn(M, N, device=dev)
        dz = torch.randn(M, N, device=dev) * 1e-2
        coef = torch.randn(3, N, device=dev)
        out = torch.zeros(M, max(K, 4), device=dev)
        p1, p2 = torch.zeros(T, max(N, K), device=dev), torch.zeros(T, max(N, K), device=dev)
        pm = [torch.zeros(T, N, device=dev) for _ in range(2)] + [torch.zeros(T, N, dtype=torch.int32, device=dev) for _ in range(2)]
        argidx = torch.randint(0, rpf, (B, N), dtype=torch.int32, device=dev)
        dpool = torch.randn(B, N, device=dev)
        bn = K > 4
        act = abi.ActSrc(fptr(x), ldx, 0, fptr(sc if bn else None), fptr(sh if bn else None), int(bn), fptr(None), 0)
        dy = abi.DySrc(fptr(None), fptr(y), fptr(coef), iptr(argidx), fptr(dpool)) if pooled else \
            abi.DySrc(fptr(dz), fptr(y), fptr(coef), iptr(None), fptr(None))
        if kind == 'fwd':
            a = abi.PointMlpFwdArgs()
            a.a, a.w, a.y, a.psum, a.psumsq = act, fptr(w), fptr(y), fptr(p1), fptr(p2)
            if pooled:
                a.pmax, a.pmin, a.pamax, a.pamin = fptr(pm[0]), fptr(pm[1]), iptr(pm[2]), iptr(pm[3])
            a.M, a.K, a.N, a.rows_per_frustum = M, K, N, rpf
            fn = lib.t3d_pointmlp_fwd
        elif kind == 'dgrad':
            a = abi.PointMlpDgradArgs()
            a.dy, a.w, a.out = dy, fptr(w), fptr(out)
            if not os.environ.get('T3D_DGRAD_RAW'):
                a.prev_y, a.prev_scale, a.prev_shift, a.psum_dz, a.psum_dzy = fptr(x), fptr(sc), fptr(sh), fptr(p1), fptr(p2)
            a.M, a.K, a.N, a.rows_per_frustum = M, K, N, rpf
            fn = lib.t3d_pointmlp_dgrad
        else:
            rps, tk, tn = C.c_int(0), C.c_int(0), C.c_int(0)
            lib.t3d_wgrad_plan(M, K, N, C.byref(rps), C.byref(tk), C.byref(tn))
            slabs = torch.zeros(M // rps.value, K, N, device=dev)
            a = abi.PointMlpWgradArgs()
            a.a, a.dy, a.slabs = act, dy, fptr(slabs)
            a.M, a.K, a.N, a.rows_per_frustum, a.rows_per_split = M, K, N, rpf, rps.value
            fn = lib.t3d_pointmlp_wgrad
        for _ in range(3):
            assert fn(C.byref(a), s) == 0
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(R):
            fn(C.byref(a), s)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / R * 1e3
        fl = 2.0 * M * K * N
        tot[kind] = tot.get(kind, 0) + us
        extra = ' split=%d tile=%dx%d' % (M // rps.value, tk.value, tn.value) if kind == 'wgrad' else ''
        nbytes = 4.0 * M * (K + N)
        print('%-6s K%-4d N%-5d %s %8.1f us  %6.1f TF/s  (ideal %.1f us)  %5.2f TB/s of x+y%s' % (
            kind, K, N, 'pool' if pooled else '    ', us, fl / us / 1e6, fl / 157.3e6, nbytes / us / 1e6, extra))
    print('totals (one of each):', {k: round(v, 1) for k, v in tot.items()})


if __name__ == '__main__':
    main()
