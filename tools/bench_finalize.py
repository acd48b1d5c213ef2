#!/usr/bin/env python3
"""Back-to-back timing of t3d_bn_fwd_finalize / t3d_bn_bwd_finalize (dense) at a given tile count, per channel count:
  [T3D_FIN_WIDE=0] python tools/bench_finalize.py [n_tiles] [reps]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from transferable3d_amd import abi
from transferable3d_amd.abi import fptr


def main():
    T = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 50
    lib = abi.load()
    dev = 'cuda'
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for N in (64, 128, 256, 512, 1024):
        z = lambda *s: torch.zeros(*s, device=dev)
        psum, psumsq = torch.randn(T, N, device=dev), torch.rand(T, N, device=dev) * 128 + 200
        f = abi.BnFwdFinalizeArgs()
        keep = [z(N) + 1, z(N), z(N), z(N) + 1, z(1) + 0.5, z(N), z(N), z(N), z(N)]
        f.psum, f.psumsq, f.n_tiles, f.count, f.N = fptr(psum), fptr(psumsq), T, T * 128, N
        f.gamma, f.beta, f.moving_mean, f.moving_var, f.decay = [fptr(k) for k in keep[:5]]
        f.eps, f.is_training, f.unbiased_ema = 1e-3, 1, 1
        f.scale, f.shift, f.mean, f.invstd = [fptr(k) for k in keep[5:]]
        b = abi.BnBwdFinalizeArgs()
        kb = [z(3, N), z(N), z(N)]
        b.psum_dz, b.psum_dzy, b.n_tiles, b.count, b.N = fptr(psum), fptr(psumsq), T, T * 128, N
        b.gamma, b.mean, b.invstd = fptr(keep[0]), fptr(keep[7]), fptr(keep[8])
        b.coef, b.dgamma, b.dbeta = [fptr(k) for k in kb]
        out = []
        for fn, a in ((lib.t3d_bn_fwd_finalize, f), (lib.t3d_bn_bwd_finalize, b)):
            for _ in range(3):
                assert fn(C.byref(a), st) == 0
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                assert fn(C.byref(a), st) == 0
            e1.record()
            torch.cuda.synchronize()
            out.append(e0.elapsed_time(e1) * 1e3 / reps)
        print('n_tiles %d N %4d: fwd finalize %.2f us, bwd finalize %.2f us per launch (back to back)' % (T, N, out[0], out[1]))


if __name__ == '__main__':
    main()
