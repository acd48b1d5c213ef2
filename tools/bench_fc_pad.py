#!/usr/bin/env python3
"""Does the row pitch of an FC layer's input matter?  t3d_fc_fwd on x[B, K] with leading dimension K + pad (the pooled features are
allocated with pitch K = 256 / 512 / 1024 floats: 32 rows at a power-of-two pitch).  Weights rewritten and 64 MB streamed before
every timed launch, as in the step.   python tools/bench_fc_pad.py"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from transferable3d_amd import abi
from transferable3d_amd.abi import fptr


def main():
    lib = abi.load()
    dev, B = 'cuda', 32
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    junk = torch.zeros(16 << 20, device=dev)
    for K, N in ((1024, 512), (512, 512), (256, 128)):
        for pad in (0, 16, 32, 64, 96):
            ld = K + pad
            x = torch.randn(B, ld, device=dev)
            w = torch.randn(K, N, device=dev) / K ** 0.5
            out = torch.zeros(B, N, device=dev)
            a = abi.FcFwdArgs()
            a.in_, a.ld_in, a.K, a.w = fptr(x), ld, K, fptr(w)
            a.eps, a.is_training, a.act, a.keep_prob, a.out, a.ld_out, a.B, a.N = 1e-3, 1, 0, 1.0, fptr(out), N, B, N
            ts = []
            for rep in range(12):
                w.mul_(1.0)
                junk.add_(1.0)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                assert lib.t3d_fc_fwd(C.byref(a), s) == 0
                e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) * 1e3)
            print('K%-5d N%-4d pitch K+%-3d: %.1f us (median of %d, event around one launch)' % (K, N, pad, float(np.median(ts[2:])), len(ts) - 2))


if __name__ == '__main__':
    main()
