#!/usr/bin/env python3
"""Times t3d_pointmlp_fwd on the pooled-layer shapes with y = NULL (A-resident kernel unless T3D_FWD_POOL=0)."""
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
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for K, N in [(128, 1024), (256, 512), (128, 256)]:
        x = torch.randn(M, K, device=dev)
        sc, sh = torch.rand(K, device=dev) + 0.5, torch.randn(K, device=dev) * 0.1
        w = torch.randn(K, N, device=dev) / K ** 0.5
        bias = torch.zeros(N, device=dev)
        mask = (torch.rand(M, device=dev) < 0.4).float()
        o = [torch.zeros(T, N, device=dev) for _ in range(4)] + [torch.zeros(T, N, dtype=torch.int32, device=dev) for _ in range(2)]
        a = abi.PointMlpFwdArgs()
        a.a = abi.ActSrc(fptr(x), K, 0, fptr(sc), fptr(sh), 1, fptr(None), 0)
        a.w, a.bias, a.psum, a.psumsq, a.rowmask = fptr(w), fptr(bias), fptr(o[0]), fptr(o[1]), fptr(mask)
        a.pmax, a.pmin, a.pamax, a.pamin = fptr(o[2]), fptr(o[3]), iptr(o[4]), iptr(o[5])
        a.M, a.K, a.N, a.rows_per_frustum = M, K, N, rpf
        for _ in range(3):
            assert lib.t3d_pointmlp_fwd(C.byref(a), s) == 0
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(R):
            lib.t3d_pointmlp_fwd(C.byref(a), s)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / R * 1e3
        print('fwd pooled K%-4d N%-5d %8.1f us  %6.1f TF/s' % (K, N, us, 2.0 * M * K * N / us / 1e6))


if __name__ == '__main__':
    main()
