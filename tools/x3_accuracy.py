#!/usr/bin/env python3
"""Accuracy of the T3D_X3 arithmetic (three bf16 terms per operand, six bf16 MFMAs, fp32 accumulate) against the fp32-MFMA kernels,
both measured against fp64 products of the same fp32 operands: same-sign operands (a rounding BIAS in the accumulation would grow
linearly with the reduction length) and random-sign operands, forward (reduction K) and weight gradient (reduction = rows)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from transferable3d_amd import abi
from transferable3d_amd.abi import fptr, iptr


def main():
    lib = abi.load(os.environ.get('T3D_LIB'))
    dev = 'cuda'
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    torch.manual_seed(1)
    for sign in ('positive', 'random'):
        # forward: M x K times K x N, K = 512 / 1024
        for K in (128, 512, 1024):
            M, N, rpf = 1024, 128, 256
            x = torch.rand(M, K, device=dev) + 0.5 if sign == 'positive' else torch.randn(M, K, device=dev)
            w = (torch.rand(K, N, device=dev) + 0.5 if sign == 'positive' else torch.randn(K, N, device=dev)) / K
            y = torch.zeros(M, N, device=dev)
            p1, p2 = torch.zeros(M // 128, N, device=dev), torch.zeros(M // 128, N, device=dev)
            a = abi.PointMlpFwdArgs()
            a.a = abi.ActSrc(fptr(x), K, 0, fptr(None), fptr(None), 0, fptr(None), 0)
            a.w, a.y, a.psum, a.psumsq = fptr(w), fptr(y), fptr(p1), fptr(p2)
            a.M, a.K, a.N, a.rows_per_frustum = M, K, N, rpf
            ref = x.double() @ w.double()
            out = {}
            for mode in ('0', '1'):
                os.environ['T3D_X3'] = mode
                assert lib.t3d_pointmlp_fwd(C.byref(a), s) == 0
                torch.cuda.synchronize()
                e = (y.double() - ref) / ref.abs().mean()
                out[mode] = (float(e.abs().max()), float(e.mean()), float(e.pow(2).mean().sqrt()))
            print('fwd   %-8s K=%5d   fp32-MFMA: max %.1e mean %+.1e rms %.1e    x3: max %.1e mean %+.1e rms %.1e' % ((sign, K) + out['0'] + out['1']))
        # weight gradient: reduction over rows
        for rows in (512, 4096, 32768):
            M, K, N, rpf = rows, 128, 128, 512 if rows >= 512 else rows
            x = torch.rand(M, K, device=dev) + 0.5 if sign == 'positive' else torch.randn(M, K, device=dev)
            yv = torch.zeros(M, N, device=dev)
            dz = (torch.rand(M, N, device=dev) + 0.5 if sign == 'positive' else torch.randn(M, N, device=dev)) / M
            coef = torch.zeros(3, N, device=dev)
            coef[0] = 1.0
            slabs = torch.zeros(1, K, N, device=dev)
            wa = abi.PointMlpWgradArgs()
            wa.a = abi.ActSrc(fptr(x), K, 0, fptr(None), fptr(None), 0, fptr(None), 0)
            wa.dy, wa.slabs = abi.DySrc(fptr(dz), fptr(yv), fptr(coef), iptr(None), fptr(None)), fptr(slabs)
            wa.M, wa.K, wa.N, wa.rows_per_frustum, wa.rows_per_split = M, K, N, rpf, M
            ref = x.double().t() @ dz.double()
            out = {}
            for mode in ('0', '1'):
                os.environ['T3D_X3'] = mode
                assert lib.t3d_pointmlp_wgrad(C.byref(wa), s) == 0
                torch.cuda.synchronize()
                e = (slabs[0].double() - ref) / ref.abs().mean()
                out[mode] = (float(e.abs().max()), float(e.mean()), float(e.pow(2).mean().sqrt()))
            print('wgrad %-8s rows=%5d fp32-MFMA: max %.1e mean %+.1e rms %.1e    x3: max %.1e mean %+.1e rms %.1e' % ((sign, rows) + out['0'] + out['1']))
    os.environ.pop('T3D_X3', None)


if __name__ == '__main__':
    main()
