#!/usr/bin/env python3
"""t3d_seg_head (conv10 + soft-max cross-entropy + mask statistics + the gradient into conv9, one launch) taken apart by its own
arguments: inference, training forward only, training with the backward, with the dropout mask generated in the kernel / read from
memory / absent.  us per launch and the algorithmic bytes over that time, fp32 at M = 32768 and bf16 at M = 262144."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from transferable3d_amd import abi
from transferable3d_amd.abi import fptr, iptr


def timed(fn, R=40):
    for _ in range(5):
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
    for dtype, M, rpf in (('f32', 32768, 1024), ('bf16', 262144, 2048), ('f32', 65536, 2048)):
        torch.manual_seed(0)
        K, B, T = 128, M // rpf, M // 128
        td = torch.bfloat16 if dtype == 'bf16' else torch.float32
        es = 2 if dtype == 'bf16' else 4
        y = torch.randn(M, K, device=dev).to(td)
        sc, sh = torch.rand(K, device=dev) + 0.5, torch.randn(K, device=dev) * 0.1
        w, b = torch.randn(K, 2, device=dev) * 0.1, torch.zeros(2, device=dev)
        pc = torch.randn(M, 4, device=dev)
        lab = (torch.rand(M, device=dev) < 0.3).int()
        is2d = torch.zeros(B, dtype=torch.int32, device=dev)
        mask_in = (torch.rand(M, K, device=dev) < 0.5).float()
        hyper = torch.tensor([3.0, 0, 0, 0], device=dev)
        logits, mask, part = torch.zeros(M, 2, device=dev), torch.zeros(M, device=dev), torch.zeros(T, 8, device=dev)
        dz = torch.zeros(M, K, device=dev).to(td)
        p1, p2, dwp = torch.zeros(T, K, device=dev), torch.zeros(T, K, device=dev), torch.zeros(T, K, 2, device=dev)
        for name, train, bwd, drop in (('inference', 0, 0, 'none'), ('train fwd, no dropout', 1, 0, 'none'), ('train fwd+bwd, no dropout', 1, 1, 'none'),
                                       ('train fwd+bwd, mask from memory', 1, 1, 'mem'), ('train fwd+bwd, mask generated (the step)', 1, 1, 'gen')):
            a = abi.SegHeadArgs()
            a.y, a.scale, a.shift, a.w, a.bias = C.cast(C.c_void_p(y.data_ptr()), abi.F), fptr(sc), fptr(sh), fptr(w), fptr(b)
            a.pc, a.ld_pc, a.ce_weight = fptr(pc), 4, 1.0
            a.logits, a.mask, a.part = fptr(logits), fptr(mask), fptr(part)
            a.M, a.K, a.rows_per_frustum, a.B, a.dtype = M, K, rpf, B, abi.BF16 if dtype == 'bf16' else abi.F32
            a.keep_prob = 1.0
            if train:
                a.labels, a.is_data_2D = iptr(lab), iptr(is2d)
            if bwd:
                a.dz, a.psum_dz, a.psum_dzy, a.dw_part = C.cast(C.c_void_p(dz.data_ptr()), abi.F), fptr(p1), fptr(p2), fptr(dwp)
            if drop == 'mem':
                a.drop_mask, a.keep_prob = fptr(mask_in), 0.5
            elif drop == 'gen':
                a.drop_seed, a.drop_hyper, a.keep_prob = 4321, fptr(hyper), 0.5
            assert lib.t3d_seg_head(C.byref(a), s) == 0
            us = timed(lambda: lib.t3d_seg_head(C.byref(a), s))
            by = M * K * es * (2 if bwd else 1) + (M * K * 4 if drop == 'mem' else 0)
            print('%-5s M %6d  %-42s %7.1f us   %6.0f GB/s' % (dtype, M, name, us, by / us / 1e3), flush=True)


if __name__ == '__main__':
    main()
