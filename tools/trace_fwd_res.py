#!/usr/bin/env python3
"""Phase clocks of the activation-resident bf16 forward (k_pointmlp_fwd_res), diagnostic build tools/build_variant.sh trace "-DT3D_TRACE":
thread 0 of every workgroup records the 100 MHz wall clock at entry (0), after the panel is staged (1), after the first weight
tile is in LDS (2), after the k-loop of the first column tile (3), after its epilogue arithmetic (4), after the partials (5), after
the y stores were issued (6), at exit (7).   python tools/trace_fwd_res.py K N [M]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from transferable3d_amd import abi
from transferable3d_amd.abi import fptr


def main():
    K, N = int(sys.argv[1]), int(sys.argv[2])
    M = int(sys.argv[3]) if len(sys.argv) > 3 else 262144
    lib = abi.load(os.environ.get('T3D_LIB', 'tools/libt3d_trace.so'))
    lib.t3d_set_trace.argtypes = [C.c_void_p]
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
    for _ in range(3):
        assert lib.t3d_pointmlp_fwd(C.byref(a), s) == 0
    torch.cuda.synchronize()
    trace = torch.zeros(T * 8, dtype=torch.int64, device=dev)
    assert lib.t3d_set_trace(C.c_void_p(trace.data_ptr())) == 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    assert lib.t3d_pointmlp_fwd(C.byref(a), s) == 0
    e1.record()
    torch.cuda.synchronize()
    assert lib.t3d_set_trace(C.c_void_p(0)) == 0
    tr = trace.cpu().numpy().reshape(T, 8)
    tr = tr[tr[:, 0] != 0]
    t = (tr - tr[:, 0].min()) / 100.0
    names = ['entry', 'panel staged', 'W0 in LDS', 'k-loop', 'epilogue math', 'partials', 'y stores issued', 'exit']
    print('K%d N%d M%d: launch %.1f us, %d workgroups traced' % (K, N, M, e0.elapsed_time(e1) * 1e3, len(tr)))
    for i in range(1, 8):
        d = t[:, i] - t[:, i - 1]
        print('  %-16s median %6.2f us   p10 %6.2f   p90 %6.2f' % (names[i], np.median(d), np.percentile(d, 10), np.percentile(d, 90)))
    print('  first tile total  median %6.2f us; workgroup lifetime median %6.2f us; last exit %.1f us' %
          (np.median(t[:, 6] - t[:, 0]), np.median(t[:, 7] - t[:, 0]), t[:, 7].max()))


if __name__ == '__main__':
    main()
