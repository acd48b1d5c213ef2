#!/usr/bin/env python3
"""How much of a bf16 layer launch is the batch-norm / ReLU (and dy) transform applied while its operands are staged?  The same
forward / fused backward launches with and without the per-channel transform of the input (t3d_act_src.scale == NULL: the loader copies),
M = 262144 (BASELINE configs[4])."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from transferable3d_amd import abi
from transferable3d_amd.abi import fptr, iptr

F = abi.F


def bptr(t):
    return C.cast(C.c_void_p(t.data_ptr()), F)


def timed(fn, R=20):
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
    dev = 'cuda'
    M, rpf = int(os.environ.get('T3D_M', '262144')), 2048
    T = M // 128
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    torch.manual_seed(0)
    for K, N in ((512, 256), (256, 128), (128, 128), (128, 1024)):
        x = torch.randn(M, K, device=dev).bfloat16()
        sc, sh = torch.rand(K, device=dev) + 0.5, torch.randn(K, device=dev) * 0.1
        w = (torch.randn(K, N, device=dev) / K ** 0.5).bfloat16()
        y = torch.zeros(M, N, device=dev).bfloat16()
        p1, p2 = torch.zeros(T, N, device=dev), torch.zeros(T, N, device=dev)
        res = {}
        for mode in ('transform', 'copy'):
            a = abi.PointMlpFwdArgs()
            a.a = abi.ActSrc(bptr(x), K, 0, fptr(sc) if mode == 'transform' else fptr(None), fptr(sh) if mode == 'transform' else fptr(None),
                             1 if mode == 'transform' else 0, fptr(None), 0, abi.BF16)
            a.w, a.y, a.psum, a.psumsq = bptr(w), bptr(y), fptr(p1), fptr(p2)
            a.M, a.K, a.N, a.rows_per_frustum, a.dtype = M, K, N, rpf, abi.BF16
            rc = lib.t3d_pointmlp_fwd(C.byref(a), s)
            assert rc == 0, rc
            res[mode] = timed(lambda: lib.t3d_pointmlp_fwd(C.byref(a), s))
        by = 2.0 * M * (K + N)
        print('fwd bf16 %4d -> %4d   with the transform %7.1f us (%5.0f GB/s)   plain copy %7.1f us (%5.0f GB/s)' %
              (K, N, res['transform'], by / res['transform'] / 1e3, res['copy'], by / res['copy'] / 1e3), flush=True)


if __name__ == '__main__':
    main()
