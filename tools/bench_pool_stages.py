#!/usr/bin/env python3
"""The Gram-form backward of the max-pooled layers at the headline size (M = 32768), launch by launch: Gram slabs, column sums and
P / rowconst preparation alone and as stage 1; weight-gradient assembly and the data gradient alone and as stage 2; the slab reduction
+ sparse rows (pool_bwd_mid).  us per launch, both arithmetics."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from transferable3d_amd import abi
from transferable3d_amd.abi import fptr, iptr


def timed(fn, R=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(R):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / R * 1e3


def main():
    lib = abi.load(os.environ.get('T3D_LIB'))
    dev = 'cuda'
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    M, rpf = 32768, 1024
    B, T = M // rpf, M // 128
    torch.manual_seed(0)
    for K, N in ((128, 1024), (128, 256), (256, 512)):
        x = torch.randn(M, K, device=dev)
        sc, sh = torch.rand(K, device=dev) + 0.5, torch.randn(K, device=dev) * 0.1
        w = torch.randn(K, N, device=dev) / K ** 0.5
        bias = torch.randn(N, device=dev) * 0.1
        coef = torch.randn(3, N, device=dev)
        argidx = torch.randint(0, rpf, (B, N), dtype=torch.int32, device=dev)
        dpool = torch.randn(B, N, device=dev)
        act = abi.ActSrc(fptr(x), K, 0, fptr(sc), fptr(sh), 1, fptr(None), 0)
        rps, tk, tn = C.c_int(0), C.c_int(0), C.c_int(0)
        assert lib.t3d_wgrad_plan(M, K, K, C.byref(rps), C.byref(tk), C.byref(tn)) == 0
        S, nch = M // rps.value, N // 128
        z = lambda *sh_: torch.zeros(*sh_, device=dev)
        for arith, name in ((abi.ARITH_BF16X3, 'bf16x3'), (abi.ARITH_FP32_MFMA, 'fp32_mfma')):
            gsl, part, ps, rcs, wc = z(S, K, K), z(T, K), z(nch, K, K), z(nch, K), z(N, K)
            ga = abi.PointMlpGramArgs(act, fptr(gsl), M, K, rpf, rps.value, arith)
            ca = abi.ActColsumArgs(act, M, K, rpf, fptr(part))
            qa = abi.PoolBwdPrepArgs(fptr(w), fptr(bias), fptr(coef), K, N, fptr(ps), fptr(rcs), fptr(wc))
            t_g = timed(lambda: lib.t3d_pointmlp_gram(C.byref(ga), s))
            t_c = timed(lambda: lib.t3d_act_colsum(C.byref(ca), s))
            t_q = timed(lambda: lib.t3d_pool_bwd_prep(C.byref(qa), s))
            t_1 = timed(lambda: lib.t3d_pool_bwd_stage1(C.byref(ga), C.byref(ca), C.byref(qa), s))
            G, abar, P, rc = gsl.sum(0), part.sum(0), ps.sum(0), rcs.sum(0)
            Sm = z(M, K)
            dw, o, s1, s2 = z(K, N), z(M, K), z(T, K), z(T, K)
            f = abi.PoolWgradFinishArgs()
            f.a, f.argidx, f.dpool, f.coef, f.w, f.bias = act, iptr(argidx), fptr(dpool), fptr(coef), fptr(w), fptr(bias)
            f.g, f.abar, f.B, f.K, f.N, f.rows_per_frustum, f.dw = fptr(G), fptr(abar), B, K, N, rpf, fptr(dw)
            dg = abi.PointMlpDgradGramArgs()
            dg.a, dg.p, dg.rowconst, dg.add_in, dg.prev_y, dg.prev_scale, dg.prev_shift = act, fptr(P), fptr(rc), fptr(Sm), fptr(x), fptr(sc), fptr(sh)
            dg.out, dg.psum_dz, dg.psum_dzy, dg.M, dg.K, dg.rows_per_frustum, dg.arith = fptr(o), fptr(s1), fptr(s2), M, K, rpf, arith
            t_f = timed(lambda: lib.t3d_pool_wgrad_finish(C.byref(f), s))
            t_d = timed(lambda: lib.t3d_pointmlp_dgrad_gram(C.byref(dg), s))
            t_2 = timed(lambda: lib.t3d_pool_bwd_stage2(C.byref(f), C.byref(dg), s))
            print('K %3d N %4d %-9s rows_per_split %4d (%3d slabs) | gram %5.1f  colsum %5.1f  prep %5.1f  -> stage1 %5.1f | wgrad_finish %5.1f  dgrad_gram %5.1f -> stage2 %5.1f   us per launch'
                  % (K, N, name, rps.value, S, t_g, t_c, t_q, t_1, t_f, t_d, t_2), flush=True)


if __name__ == '__main__':
    main()
