#!/usr/bin/env python3
"""Does running a layer's wgrad beside its dgrad pay?  Serial on one stream vs the two kernels on two streams (eager)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from transferable3d_amd import abi
from transferable3d_amd.abi import fptr, iptr


def main():
    lib = abi.load(os.environ.get('T3D_LIB'))
    M, rpf, R = 32768, 1024, 20
    T = M // 128
    dev = 'cuda'
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    for K, N in [(128, 128), (128, 256), (256, 128), (512, 256), (64, 128), (64, 64)]:
        x = torch.randn(M, K, device=dev)
        sc, sh = torch.rand(K, device=dev) + 0.5, torch.randn(K, device=dev) * 0.1
        w = torch.randn(K, N, device=dev) / K ** 0.5
        y, dz, coef = torch.randn(M, N, device=dev), torch.randn(M, N, device=dev) * 1e-2, torch.randn(3, N, device=dev)
        out, p1, p2 = torch.zeros(M, K, device=dev), torch.zeros(T, K, device=dev), torch.zeros(T, K, device=dev)
        act = abi.ActSrc(fptr(x), K, 0, fptr(sc), fptr(sh), 1, fptr(None), 0)
        dy = abi.DySrc(fptr(dz), fptr(y), fptr(coef), iptr(None), fptr(None))
        d = abi.PointMlpDgradArgs()
        d.dy, d.w, d.out = dy, fptr(w), fptr(out)
        d.prev_y, d.prev_scale, d.prev_shift, d.psum_dz, d.psum_dzy = fptr(x), fptr(sc), fptr(sh), fptr(p1), fptr(p2)
        d.M, d.K, d.N, d.rows_per_frustum = M, K, N, rpf
        rps, tk, tn = C.c_int(0), C.c_int(0), C.c_int(0)
        lib.t3d_wgrad_plan(M, K, N, C.byref(rps), C.byref(tk), C.byref(tn))
        slabs = torch.zeros(M // rps.value, K, N, device=dev)
        g = abi.PointMlpWgradArgs(act, dy, fptr(slabs), M, K, N, rpf, rps.value)

        def run(two):
            a, b = (C.c_void_p(s1.cuda_stream), C.c_void_p(s2.cuda_stream if two else s1.cuda_stream))
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(s1)
            for _ in range(R):
                if two:
                    s2.wait_stream(s1)
                lib.t3d_pointmlp_wgrad(C.byref(g), b)
                lib.t3d_pointmlp_dgrad(C.byref(d), a)
                if two:
                    s1.wait_stream(s2)
            e1.record(s1)
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) / R * 1e3
        run(False); run(True)
        print('K%-4d N%-4d serial %7.1f us   two streams %7.1f us' % (K, N, run(False), run(True)))


if __name__ == '__main__':
    main()
