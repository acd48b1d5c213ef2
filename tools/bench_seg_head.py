#!/usr/bin/env python3
"""Where the segmentation head's time goes: the kernel with / without its backward half, with stored / drawn / no dropout mask."""
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
    M, K, rpf = 32768, 128, 1024
    B, T = M // rpf, M // 128
    dev = 'cuda'
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    y = torch.randn(M, K, device=dev)
    sc, sh = torch.rand(K, device=dev) + 0.5, torch.randn(K, device=dev) * 0.1
    w, b = torch.randn(K, 2, device=dev) * 0.2, torch.zeros(2, device=dev)
    lab = (torch.rand(M, device=dev) < 0.3).int()
    is2d = torch.zeros(B, dtype=torch.int32, device=dev)
    pc = torch.randn(M, 4, device=dev)
    mask = (torch.rand(M, K, device=dev) < 0.5).float()
    hyper = torch.tensor([3.0, 0, 0, 0], device=dev)
    o = dict(logits=torch.zeros(M, 2, device=dev), mask=torch.zeros(M, device=dev), part=torch.zeros(T, 8, device=dev),
             dz=torch.zeros(M, K, device=dev), p1=torch.zeros(T, K, device=dev), p2=torch.zeros(T, K, device=dev), dw=torch.zeros(T, K, 2, device=dev))
    for name, bwd, drop in (('fwd+bwd, drawn mask', 1, 'gen'), ('fwd+bwd, stored mask', 1, 'mem'), ('fwd+bwd, no dropout', 1, None),
                            ('fwd only (labels), drawn mask', 0, 'gen'), ('inference (no labels)', -1, None)):
        a = abi.SegHeadArgs()
        a.y, a.scale, a.shift, a.keep_prob, a.w, a.bias = fptr(y), fptr(sc), fptr(sh), 0.5 if drop else 1.0, fptr(w), fptr(b)
        a.labels, a.is_data_2D, a.pc, a.ld_pc, a.ce_weight = iptr(lab if bwd >= 0 else None), iptr(is2d), fptr(pc), 4, 1.0
        a.logits, a.mask, a.part = fptr(o['logits']), fptr(o['mask']), fptr(o['part'])
        if bwd == 1:
            a.dz, a.psum_dz, a.psum_dzy, a.dw_part = fptr(o['dz']), fptr(o['p1']), fptr(o['p2']), fptr(o['dw'])
        a.M, a.K, a.rows_per_frustum, a.B = M, K, rpf, B
        if drop == 'gen':
            a.drop_seed, a.drop_hyper = 99, fptr(hyper)
        elif drop == 'mem':
            a.drop_mask = fptr(mask)
        for _ in range(3):
            assert lib.t3d_seg_head(C.byref(a), st) == 0
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            lib.t3d_seg_head(C.byref(a), st)
        e1.record()
        torch.cuda.synchronize()
        print('%-34s %6.1f us' % (name, e0.elapsed_time(e1) / 20 * 1e3))


if __name__ == '__main__':
    main()
