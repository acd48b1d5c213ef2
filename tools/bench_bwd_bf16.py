#!/usr/bin/env python3
"""The fused backward of one dense layer (T3D_BF16, or fp32 with a trailing `f32`), split form vs one-pass form (same arguments,
different row split: t3d_wgrad_plan's split leaves too few workgroups for the one-pass form at the sizes this tool is for):
  python tools/bench_bwd_bf16.py K N [M] [reps] [rows_per_split of the one-pass form, 0 = the plan's] [f32]
Prints us per launch and the algorithmic HBM rate (every tensor once: dz, y, the input, dz_prev out)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from transferable3d_amd import abi
from transferable3d_amd.abi import fptr, iptr


def main():
    K, N = int(sys.argv[1]), int(sys.argv[2])
    M = int(sys.argv[3]) if len(sys.argv) > 3 else 262144
    reps = int(sys.argv[4]) if len(sys.argv) > 4 else 10
    f32 = len(sys.argv) > 6 and sys.argv[6] == 'f32'
    lib = abi.load()
    dev, T, BF = 'cuda', M // 128, (torch.float32 if f32 else torch.bfloat16)
    DT, es = (abi.F32 if f32 else abi.BF16), (4.0 if f32 else 2.0)
    dz = (torch.randn(M, N, device=dev) * 1e-2).to(BF)
    y = torch.randn(M, N, device=dev).to(BF)
    coef = torch.randn(3, N, device=dev)
    w16 = (torch.randn(K, N, device=dev) / N ** 0.5).to(BF)
    prev_y = torch.randn(M, K, device=dev).to(BF)
    psc, psh = torch.rand(K, device=dev) + 0.5, torch.randn(K, device=dev) * 0.3
    out = torch.zeros(M, K, dtype=BF, device=dev)
    ps1, ps2 = torch.zeros(T, K, device=dev), torch.zeros(T, K, device=dev)
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    rps, tk, tn, one = C.c_int(0), C.c_int(0), C.c_int(0), C.c_int(0)
    assert lib.t3d_wgrad_plan(M, K, N, C.byref(rps), C.byref(tk), C.byref(tn)) == 0
    forms = [('split', rps.value)]
    assert lib.t3d_bwd_plan(M, K, N, DT, C.byref(rps), C.byref(one)) == 0
    if one.value:
        forms.append(('one-pass', int(sys.argv[5]) if len(sys.argv) > 5 and int(sys.argv[5]) > 0 else rps.value))
    by = es * (2 * M * N + 2 * M * K)
    res = {}
    for name, r in forms:
        slabs = torch.zeros(M // r, K, N, device=dev)
        dy = abi.DySrc(fptr(dz), fptr(y), fptr(coef), iptr(None), fptr(None), DT)
        d = abi.PointMlpDgradArgs()
        d.dy, d.w, d.add_in = dy, fptr(w16), fptr(None)
        d.prev_y, d.prev_scale, d.prev_shift, d.out, d.psum_dz, d.psum_dzy = fptr(prev_y), fptr(psc), fptr(psh), fptr(out), fptr(ps1), fptr(ps2)
        d.M, d.K, d.N, d.rows_per_frustum, d.dtype = M, K, N, 1024, DT
        wa = abi.PointMlpWgradArgs()
        wa.a = abi.ActSrc(fptr(prev_y), K, 0, fptr(psc), fptr(psh), 1, fptr(None), 0, DT)
        wa.dy, wa.slabs = dy, fptr(slabs)
        wa.M, wa.K, wa.N, wa.rows_per_frustum, wa.rows_per_split = M, K, N, 1024, r
        for _ in range(2):
            assert lib.t3d_pointmlp_bwd(C.byref(d), C.byref(wa), s) == 0
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            assert lib.t3d_pointmlp_bwd(C.byref(d), C.byref(wa), s) == 0
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / reps
        res[name] = (out.float().clone(), slabs.sum(0))
        print('K%d N%d M%d %-8s rows_per_split %6d (%4d workgroups/slabs): %7.1f us per launch, %5.0f GB/s algorithmic (%.2f of 8 TB/s)'
              % (K, N, M, name, r, M // r, us, by / us * 1e-3, by / us * 1e-3 / 8000))
    if len(res) == 2:
        a, b = res['split'], res['one-pass']
        print('   forms agree: dX max diff %.3g (max %.3g), dW max diff %.3g (max %.3g)'
              % (float((a[0] - b[0]).abs().max()), float(a[0].abs().max()), float((a[1] - b[1]).abs().max()), float(a[1].abs().max())))


if __name__ == '__main__':
    main()
