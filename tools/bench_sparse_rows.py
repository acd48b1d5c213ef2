#!/usr/bin/env python3
"""Times t3d_pool_sparse_rows on clustered arg-max patterns (a few hot rows per frustum, like PointNet critical points)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from transferable3d_amd import abi
from transferable3d_amd.abi import fptr, iptr


def main():
    lib = abi.load(os.environ.get('T3D_LIB'))
    B, rpf, R = 32, 1024, 20
    M = B * rpf
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    r = np.random.RandomState(0)
    for K, N, hot in [(128, 1024, 280), (256, 512, 64), (128, 256, 40), (128, 1024, 1024)]:
        rows = np.stack([r.choice(rpf, hot, replace=False) for _ in range(B)])
        wts = r.pareto(1.0, size=(B, hot)) + 1e-3                      # heavy-tailed popularity of the critical points
        wts /= wts.sum(1, keepdims=True)
        idx = np.stack([r.choice(hot, N, p=wts[b]) for b in range(B)])
        argidx = torch.as_tensor(np.take_along_axis(rows, idx, 1).astype(np.int32)).cuda()
        dpool, wc, S = torch.randn(B, N, device='cuda'), torch.randn(N, K, device='cuda'), torch.zeros(M, K, device='cuda')
        a = abi.PoolSparseRowsArgs(iptr(argidx), fptr(dpool), fptr(wc), B, N, K, rpf, fptr(S))
        for _ in range(3):
            assert lib.t3d_pool_sparse_rows(C.byref(a), s) == 0
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(R):
            lib.t3d_pool_sparse_rows(C.byref(a), s)
        e1.record()
        torch.cuda.synchronize()
        cnt = np.zeros((B, rpf // 128, 4), int)
        ai = argidx.cpu().numpy()
        for b in range(B):
            np.add.at(cnt[b], (ai[b] // 128, ai[b] & 3), 1)
        print('sparse rows K%-4d N%-5d hot rows %4d  max hits/wave %4d   %7.1f us' % (K, N, hot, cnt.max(), e0.elapsed_time(e1) / R * 1e3))


if __name__ == '__main__':
    main()
