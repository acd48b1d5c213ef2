#!/usr/bin/env python3
"""Per-workgroup timeline of the forward GEMM kernel (diagnostic build: tools/build_variant.sh trace "-DT3D_TRACE").
For each layer shape: dispatch ramp (spread of workgroup start times), workgroups per CU, time in main loop / epilogue,
and how much of the launch the busiest and the idlest CU spend with at least one workgroup resident."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from transferable3d_amd import abi
from transferable3d_amd.abi import fptr, iptr

SHAPES = [(512, 256, False), (256, 512, True), (64, 512, False), (256, 128, False), (128, 256, False), (128, 128, False),
          (64, 128, False), (64, 64, False)]


def main():
    lib = abi.load(os.environ.get('T3D_LIB', 'tools/libt3d_trace.so'))
    lib.t3d_set_trace.argtypes = [C.c_void_p]
    M, rpf = int(os.environ.get('T3D_M', '32768')), 1024
    bf16 = os.environ.get('T3D_DTYPE', 'f32') == 'bf16'      # the T3D_BF16 path: bf16 input / weights / output
    T = M // 128
    dev = 'cuda'
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for K, N, pooled in SHAPES:
        x = torch.randn(M, K, device=dev)
        if bf16:
            x = x.to(torch.bfloat16)
        sc, sh = torch.rand(K, device=dev) + 0.5, torch.randn(K, device=dev) * 0.1
        w = torch.randn(K, N, device=dev) / K ** 0.5
        y = torch.randn(M, N, device=dev)
        if bf16:
            w, y = w.to(torch.bfloat16), y.to(torch.bfloat16)
        p1, p2 = torch.zeros(T, N, device=dev), torch.zeros(T, N, device=dev)
        pm = [torch.zeros(T, N, device=dev) for _ in range(2)] + [torch.zeros(T, N, dtype=torch.int32, device=dev) for _ in range(2)]
        a = abi.PointMlpFwdArgs()
        a.a = abi.ActSrc(fptr(x), K, 0, fptr(sc), fptr(sh), 1, fptr(None), 0, abi.BF16 if bf16 else abi.F32)
        a.dtype = abi.BF16 if bf16 else abi.F32
        a.w, a.y, a.psum, a.psumsq = fptr(w), fptr(y), fptr(p1), fptr(p2)
        if pooled:
            a.pmax, a.pmin, a.pamax, a.pamin = fptr(pm[0]), fptr(pm[1]), iptr(pm[2]), iptr(pm[3])
        a.M, a.K, a.N, a.rows_per_frustum = M, K, N, rpf
        if not bf16 and os.environ.get('T3D_BENCH_FRAG', '1') == '1' and K % 32 == 0:      # the engine's default: fragment-order planes
            from bench_x3 import frag_planes
            keep = frag_planes(lib, w, s)
            a.w_x3, a.w_x3_stride = keep[0].data_ptr(), keep[2]
        nblk = max(8192, (M // 128) * (N // 64))
        ST = int(os.environ.get('T3D_TRACE_STRIDE', '4'))      # 8: a library built with -DT3D_TRACE_STRIDE=8 (slot 4 = x3 prologue done)
        trace = torch.zeros(nblk * ST, dtype=torch.int64, device=dev)
        os.environ['T3D_FWD_POOL'] = '0'
        for _ in range(3):
            assert lib.t3d_pointmlp_fwd(C.byref(a), s) == 0
        torch.cuda.synchronize()
        assert lib.t3d_set_trace(C.c_void_p(trace.data_ptr())) == 0
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        assert lib.t3d_pointmlp_fwd(C.byref(a), s) == 0
        e1.record()
        torch.cuda.synchronize()
        assert lib.t3d_set_trace(C.c_void_p(0)) == 0
        tr = trace.cpu().numpy().reshape(nblk, ST)
        tr = tr[tr[:, 0] != 0]
        t0, t1, t2 = [(tr[:, i] - tr[:, 0].min()) / 100.0 for i in range(3)]     # 100 MHz -> us
        cu = (tr[:, 3] >> 32 & 0xf) * 4096 + (tr[:, 3] & 0xff00) // 256             # xcc, (se, sh, cu) bits of HW_ID
        ids, counts = np.unique(cu, return_counts=True)
        span = np.array([t2[cu == i].max() - t0[cu == i].min() for i in ids])
        print('K%-4d N%-4d %s: event %.1f us | %d workgroups on %d CUs (per CU min %d max %d) | start spread %.1f us (p50 %.1f) | '
              'main loop %.1f us (min %.1f max %.1f) | epilogue %.1f us (max %.1f) | last exit %.1f us | CU busy span min %.1f max %.1f'
              % (K, N, 'pool' if pooled else '    ', e0.elapsed_time(e1) * 1e3, len(tr), len(ids), counts.min(), counts.max(),
                 t0.max(), np.median(t0), (t1 - t0).mean(), (t1 - t0).min(), (t1 - t0).max(), (t2 - t1).mean(), (t2 - t1).max(),
                 t2.max(), span.min(), span.max()))
        if ST >= 8 and (tr[:, 4] != 0).any():
            tp = (tr[:, 4] - tr[:, 0].min()) / 100.0
            print('        prologue (entry -> first k-tile staged) %.2f us (min %.2f max %.2f) | k loop %.2f us (min %.2f max %.2f) | first staged at %.2f, last at %.2f; main loops end %.2f .. %.2f; exits %.2f .. %.2f'
                  % ((tp - t0).mean(), (tp - t0).min(), (tp - t0).max(), (t1 - tp).mean(), (t1 - tp).min(), (t1 - tp).max(), tp.min(), tp.max(),
                     t1.min(), t1.max(), t2.min(), t2.max()))
        # the two workgroups of a CU: does the one dispatched first (lower block index) also finish its main loop first, and by how much
        bidx = np.nonzero(trace.cpu().numpy().reshape(nblk, ST)[:, 0] != 0)[0]
        d_first, d_abs, n_pairs, first_wins = [], [], 0, 0
        for i in ids[counts == 2]:
            m = np.nonzero(cu == i)[0]
            lo, hi = (m[0], m[1]) if bidx[m[0]] < bidx[m[1]] else (m[1], m[0])
            if abs(t0[lo] - t0[hi]) > 2.0:      # (not resident together)
                continue
            n_pairs += 1
            first_wins += t1[lo] < t1[hi]
            d_first.append(t1[hi] - t1[lo]); d_abs.append(abs(t1[hi] - t1[lo]))
        if n_pairs:
            print('        CU pairs resident together: %d | lower block index ends its main loop first in %d | end(hi) - end(lo) mean %.2f us | |difference| mean %.2f max %.2f us'
                  % (n_pairs, first_wins, float(np.mean(d_first)), float(np.mean(d_abs)), float(np.max(d_abs))))
        # second-round workgroups (start well after 0): when do they start relative to the first exits
        late = t0 > 0.5 * t2.max()
        if late.any():
            print('        %d workgroups start after half the launch (first at %.1f us; first exit at %.1f us)' % (late.sum(), t0[late].min(), t2.min()))


if __name__ == '__main__':
    main()
