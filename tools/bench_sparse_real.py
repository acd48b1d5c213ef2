#!/usr/bin/env python3
"""Times t3d_pool_sparse_rows (row-gated) on the arg-max patterns of the bench workload's three pooled layers, for the library in
$T3D_LIB (default: the in-tree one; ablation builds: tools/build_variant.sh noscan "-DT3D_ABL_SR_NOSCAN" noadd "-DT3D_ABL_SR_NOADD").
The patterns come from one forward + backward of SEMI_MODEL A at B=32, N=1024 with the in-tree library."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from transferable3d_amd import abi
from transferable3d_amd.abi import fptr, iptr
from transferable3d_amd.config import make_parser
from transferable3d_amd.engine import Runtime
from transferable3d_amd.nets import Graph, SemiModelA
from transferable3d_amd.synthetic import make_batch


def main():
    B, N, Cc = 32, 1024, 4
    g = Graph(B, N, Cc, rt=Runtime(), seed=0)
    c = make_parser().parse_special_args(['--SEMI_MODEL', 'A', '--WEAK_WEIGHT_REPROJECTION', '0', '--WEAK_WEIGHT_SURFACE', '0'])
    m = SemiModelA(g, c)
    m.emit_forward(g.fwd, True, True); m.emit_backward(g.bwd); g.finalize()
    m.inputs.load(make_batch(B, N, Cc, seed=1234))
    g.fwd.run(); g.bwd.run(); torch.cuda.synchronize()
    libs = [('in-tree', g.rt.lib)] + [(os.path.basename(p), abi.load(p)) for p in sys.argv[1:]]
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    R = 50
    for name, L in (('seg conv5', m.seg.L5), ('tnet conv3', m.tnet.T3), ('box conv4', m.box.B4)):
        S, live = torch.zeros_like(L.S), torch.zeros(L.M, dtype=torch.int32, device='cuda')
        a = abi.PoolSparseRowsArgs(iptr(L.argidx), fptr(L.dpool), fptr(L.wc), B, L.N, L.K, g.rpf, fptr(S), iptr(live))
        line = '%-11s K%-4d N%-5d' % (name, L.K, L.N)
        for lname, lib in libs:
            for _ in range(3):
                assert lib.t3d_pool_sparse_rows(C.byref(a), s) == 0
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(R):
                lib.t3d_pool_sparse_rows(C.byref(a), s)
            e1.record()
            torch.cuda.synchronize()
            line += '  %s %6.1f us' % (lname, e0.elapsed_time(e1) / R * 1e3)
        print(line)


if __name__ == '__main__':
    main()
