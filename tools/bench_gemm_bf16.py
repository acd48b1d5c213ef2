#!/usr/bin/env python3
"""One T3D_BF16 forward GEMM launched a few times (for rocprofv3 --pmc passes on that kernel alone):
  python tools/bench_gemm_bf16.py K N [M] [reps]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from transferable3d_amd import abi
from transferable3d_amd.abi import fptr


def main():
    K, N = int(sys.argv[1]), int(sys.argv[2])
    M = int(sys.argv[3]) if len(sys.argv) > 3 else 262144
    reps = int(sys.argv[4]) if len(sys.argv) > 4 else 5
    lib = abi.load()
    dev, T = 'cuda', M // 128
    x = torch.randn(M, K, device=dev).to(torch.bfloat16)
    sc, sh = torch.rand(K, device=dev) + 0.5, torch.randn(K, device=dev) * 0.1
    w = (torch.randn(K, N, device=dev) / K ** 0.5).to(torch.bfloat16)
    y = torch.zeros(M, N, device=dev, dtype=torch.bfloat16)
    p1, p2 = torch.zeros(T, N, device=dev), torch.zeros(T, N, device=dev)
    a = abi.PointMlpFwdArgs()
    a.a = abi.ActSrc(fptr(x), K, 0, fptr(sc), fptr(sh), 1, fptr(None), 0, abi.BF16)
    a.w, a.y, a.psum, a.psumsq = fptr(w), fptr(y), fptr(p1), fptr(p2)
    a.M, a.K, a.N, a.rows_per_frustum, a.dtype = M, K, N, 1024, abi.BF16
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for _ in range(2):
        assert lib.t3d_pointmlp_fwd(C.byref(a), s) == 0
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        assert lib.t3d_pointmlp_fwd(C.byref(a), s) == 0
    e1.record()
    torch.cuda.synchronize()
    print('K%d N%d M%d: %.1f us per launch' % (K, N, M, e0.elapsed_time(e1) * 1e3 / reps))


if __name__ == '__main__':
    main()
