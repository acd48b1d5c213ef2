#!/usr/bin/env python3
"""Where a workgroup of the fp32 one-pass backward (k_pointmlp_bwd1f) spends its time (diagnostic build:
tools/build_variant.sh trace "-DT3D_TRACE"): per-workgroup sums, over its tiles, of staging (loop top -> first barrier), MFMA phase
(-> epilogue) and epilogue (-> second barrier), read by thread 0 from the 100 MHz clock.
  T3D_LIB=tools/libt3d_trace.so python tools/trace_bwd1f.py [M]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from transferable3d_amd import abi
from transferable3d_amd.abi import fptr, iptr


def main():
    lib = abi.load(os.environ.get('T3D_LIB', 'tools/libt3d_trace.so'))
    lib.t3d_set_trace.argtypes = [C.c_void_p]
    M = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
    dev, T = 'cuda', M // 128
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for K, N in ((128, 128), (128, 64), (64, 128), (64, 64)):
        dz, y = torch.randn(M, N, device=dev) * 1e-2, torch.randn(M, N, device=dev)
        coef, w = torch.randn(3, N, device=dev), torch.randn(K, N, device=dev) / N ** 0.5
        x = torch.randn(M, K, device=dev)
        psc, psh = torch.rand(K, device=dev) + 0.5, torch.randn(K, device=dev) * 0.3
        out, ps1, ps2 = torch.zeros(M, K, device=dev), torch.zeros(T, K, device=dev), torch.zeros(T, K, device=dev)
        rps, one = C.c_int(0), C.c_int(0)
        assert lib.t3d_bwd_plan(M, K, N, abi.F32, C.byref(rps), C.byref(one)) == 0 and one.value == 1
        nblk = M // rps.value
        slabs = torch.zeros(nblk, K, N, device=dev)
        dy = abi.DySrc(fptr(dz), fptr(y), fptr(coef), iptr(None), fptr(None), abi.F32)
        d = abi.PointMlpDgradArgs()
        d.dy, d.w = dy, fptr(w)
        d.prev_y, d.prev_scale, d.prev_shift, d.out, d.psum_dz, d.psum_dzy = fptr(x), fptr(psc), fptr(psh), fptr(out), fptr(ps1), fptr(ps2)
        d.M, d.K, d.N, d.rows_per_frustum, d.dtype = M, K, N, 1024, abi.F32
        wa = abi.PointMlpWgradArgs()
        wa.a = abi.ActSrc(fptr(x), K, 0, fptr(psc), fptr(psh), 1, fptr(None), 0, abi.F32)
        wa.dy, wa.slabs = dy, fptr(slabs)
        wa.M, wa.K, wa.N, wa.rows_per_frustum, wa.rows_per_split = M, K, N, 1024, rps.value
        trace = torch.zeros(nblk * 4, dtype=torch.int64, device=dev)
        for _ in range(3):
            assert lib.t3d_pointmlp_bwd(C.byref(d), C.byref(wa), s) == 0
        torch.cuda.synchronize()
        assert lib.t3d_set_trace(C.c_void_p(trace.data_ptr())) == 0
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        assert lib.t3d_pointmlp_bwd(C.byref(d), C.byref(wa), s) == 0
        e1.record()
        torch.cuda.synchronize()
        assert lib.t3d_set_trace(C.c_void_p(0)) == 0
        tr = trace.cpu().numpy().reshape(nblk, 4) / 100.0      # us
        nt = rps.value // 64
        print('K%d N%d M%d: launch %.1f us, %d workgroups x %d tiles; per TILE (median over workgroups): staging %.2f us, MFMA phase '
              '%.2f us, epilogue %.2f us; whole workgroup %.1f us (max %.1f); MFMA time at peak %.2f us per tile'
              % (K, N, M, e0.elapsed_time(e1) * 1e3, nblk, nt, np.median(tr[:, 0]) / nt, np.median(tr[:, 1]) / nt,
                 np.median(tr[:, 2]) / nt, np.median(tr[:, 3]), tr[:, 3].max(), 64 * 4.0 * K * N / (157.3e12 / 256) * 1e6))


if __name__ == '__main__':
    main()
