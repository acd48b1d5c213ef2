#!/usr/bin/env python3
"""Phase clock of the fully-connected forward kernel (diagnostic build: tools/build_variant.sh trace "-DT3D_TRACE").
Marks per workgroup: 0 entry, 1 after the loads + MFMAs of its k-groups, 2 after the cross-wave sum, 3 after the batch statistics,
4 exit.  Before every traced launch the weights are rewritten (as Adam does) and 64 MB are streamed (as the GEMMs do)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from transferable3d_amd import abi
from transferable3d_amd.abi import fptr


def main():
    lib = abi.load(os.environ.get('T3D_LIB', 'tools/libt3d_trace.so'))
    lib.t3d_set_trace_fc.argtypes = [C.c_void_p]
    dev, B = 'cuda', int(os.environ.get('T3D_B', '32'))
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    junk = torch.zeros(16 << 20, device=dev)
    for K, N in ((512, 512), (1024, 512), (256, 256), (256, 67), (128, 3)):
        x = torch.randn(B, K, device=dev)
        w = torch.randn(K, N, device=dev) / K ** 0.5
        bias, gamma, beta = torch.zeros(N, device=dev), torch.ones(N, device=dev), torch.zeros(N, device=dev)
        mm, mv, decay = torch.zeros(N, device=dev), torch.ones(N, device=dev), torch.full((1,), 0.5, device=dev)
        y, out, mean, inv = torch.zeros(B, N, device=dev), torch.zeros(B, N, device=dev), torch.zeros(N, device=dev), torch.zeros(N, device=dev)
        a = abi.FcFwdArgs()
        a.in_, a.ld_in, a.K, a.w, a.bias, a.gamma, a.beta = fptr(x), K, K, fptr(w), fptr(bias), fptr(gamma), fptr(beta)
        a.moving_mean, a.moving_var, a.decay, a.eps, a.is_training, a.unbiased_ema, a.act = fptr(mm), fptr(mv), fptr(decay), 1e-3, 1, 1, 1
        a.keep_prob, a.y, a.out, a.ld_out, a.mean, a.invstd, a.B, a.N = 1.0, fptr(y), fptr(out), N, fptr(mean), fptr(inv), B, N
        nblk = (N + 31) // 32
        res = []
        for rep in range(6):
            trace = torch.zeros(64 * 8, dtype=torch.int64, device=dev)
            w.mul_(1.0)
            junk.add_(1.0)
            torch.cuda.synchronize()
            assert lib.t3d_set_trace_fc(C.c_void_p(trace.data_ptr())) == 0
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            assert lib.t3d_fc_fwd(C.byref(a), s) == 0
            e1.record()
            torch.cuda.synchronize()
            assert lib.t3d_set_trace_fc(C.c_void_p(0)) == 0
            t = trace.cpu().numpy().reshape(64, 8)[:nblk, :5].astype(np.float64) / 100.0
            t0 = t[:, 0].min()
            res.append([e0.elapsed_time(e1) * 1e3, t[:, 0].max() - t0] + [(t[:, i] - t[:, i - 1]).mean() for i in range(1, 5)] + [t[:, 4].max() - t0])
        r = np.median(np.array(res[1:]), 0)
        print('K%-5d N%-4d %2d workgroups: event %.1f us | start spread %.1f | loads+MFMA %.1f | cross-wave sum %.1f | statistics %.1f | '
              'output %.1f | last exit %.1f' % ((K, N, nblk) + tuple(r)))


if __name__ == '__main__':
    main()
