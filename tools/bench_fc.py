#!/usr/bin/env python3
"""Time per launch of the fully-connected entry points (t3d_fc_fwd / t3d_fc_bwd / t3d_fc_dinput) at T3D_B rows, in a stream of
back-to-back launches with the weights rewritten and 64 MB streamed between them (as in the step).  T3D_FC16=0/1 selects the form for
more than 32 rows."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from transferable3d_amd import abi
from transferable3d_amd.abi import fptr


def timed(fn, between, R=20):
    for _ in range(3):
        between(); fn()
    torch.cuda.synchronize()
    tot = 0.0
    for _ in range(R):
        between()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record()
        torch.cuda.synchronize()
        tot += e0.elapsed_time(e1)
    return tot / R * 1e3


def main():
    lib = abi.load(os.environ.get('T3D_LIB'))
    dev, B = 'cuda', int(os.environ.get('T3D_B', '128'))
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    junk = torch.zeros(16 << 20, device=dev)
    for K, N in ((512, 512), (1024, 512), (256, 256), (512, 256), (256, 128), (256, 67)):
        x = torch.randn(B, K, device=dev)
        w = torch.randn(K, N, device=dev) / K ** 0.5
        bias, gamma, beta = torch.zeros(N, device=dev), torch.ones(N, device=dev), torch.zeros(N, device=dev)
        mm, mv, decay = torch.zeros(N, device=dev), torch.ones(N, device=dev), torch.full((1,), 0.5, device=dev)
        y, out, mean, inv = torch.zeros(B, N, device=dev), torch.zeros(B, N, device=dev), torch.zeros(N, device=dev), torch.zeros(N, device=dev)
        a = abi.FcFwdArgs()
        a.in_, a.ld_in, a.K, a.w, a.bias, a.gamma, a.beta = fptr(x), K, K, fptr(w), fptr(bias), fptr(gamma), fptr(beta)
        a.moving_mean, a.moving_var, a.decay, a.eps, a.is_training, a.unbiased_ema, a.act = fptr(mm), fptr(mv), fptr(decay), 1e-3, 1, 1, 1
        a.keep_prob, a.y, a.out, a.ld_out, a.mean, a.invstd, a.B, a.N = 1.0, fptr(y), fptr(out), N, fptr(mean), fptr(inv), B, N

        def between():
            w.mul_(1.0); junk.add_(1.0)
        t_f = timed(lambda: lib.t3d_fc_fwd(C.byref(a), s), between)
        # backward of the same layer: dout given, dW wanted
        dout, dy, dw = torch.randn(B, N, device=dev), torch.zeros(B, N, device=dev), torch.zeros(K, N, device=dev)
        dg, db, dbias = torch.zeros(N, device=dev), torch.zeros(N, device=dev), torch.zeros(N, device=dev)
        b = abi.FcBwdArgs()
        b.dout, b.ld_dout, b.y, b.gamma, b.beta, b.mean, b.invstd = fptr(dout), N, fptr(y), fptr(gamma), fptr(beta), fptr(mean), fptr(inv)
        b.bn_training, b.act, b.keep_prob, b.in_, b.ld_in, b.K = 1, 1, 1.0, fptr(x), K, K
        b.dy, b.dw, b.dgamma, b.dbeta, b.dbias, b.B, b.N = fptr(dy), fptr(dw), fptr(dg), fptr(db), fptr(dbias), B, N
        t_b = timed(lambda: lib.t3d_fc_bwd(C.byref(b), s), between)
        din = torch.zeros(B, K, device=dev)
        d = abi.FcDinputArgs()
        d.dy, d.N, d.w, d.alpha, d.din, d.ld_din, d.B, d.K = fptr(dy), N, fptr(w), 1.0, fptr(din), K, B, K
        t_d = timed(lambda: lib.t3d_fc_dinput(C.byref(d), s), between)
        print('B %3d  %4d -> %4d   fwd %6.1f us   bwd (dout given, dW) %6.1f us   dinput %6.1f us' % (B, K, N, t_f, t_b, t_d), flush=True)


if __name__ == '__main__':
    main()
