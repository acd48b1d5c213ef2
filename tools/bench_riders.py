#!/usr/bin/env python3
"""Diagnostic: what a rider set costs.  FC chain 256 -> 512 -> 512 -> 256 -> 64 at B = 32 as (a) four stand-alone launches, (b) ONE rider
set of its own (t3d_run_riders: in-launch barriers), (c) the set inside a 32768 x 128 x 128 forward GEMM launch, (d) that GEMM alone."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import torch
from transferable3d_amd import abi, schedule
from transferable3d_amd.abi import fptr
from transferable3d_amd.engine import Runtime
from test_riders_gpu import _fc_chain

lib = abi.load()
rt = Runtime(lib=lib)
dev = torch.device('cuda')
s = rt.stream()
ops, outs, keep = _fc_chain(dev, 32, (256, 512, 512, 256, 64), 7)
sets = schedule.RiderSets(rt)
rs = sets.make([('t3d_fc_fwd', a) for a in ops])
rs1 = [sets.make([('t3d_fc_fwd', a)]) for a in ops]
M = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
K = int(sys.argv[2]) if len(sys.argv) > 2 else 128
N = int(sys.argv[3]) if len(sys.argv) > 3 else 128
r = np.random.RandomState(5)
t = lambda a: torch.as_tensor(np.ascontiguousarray(a, dtype=np.float32)).to(dev)
xg, wg = t(r.randn(M, K)), t(r.randn(K, N) / 11.0)
yg, ps, pq = t(np.zeros((M, N))), t(np.zeros((M // 128, N))), t(np.zeros((M // 128, N)))
fa = abi.PointMlpFwdArgs()
fa.a = abi.ActSrc(fptr(xg), K, 0, None, None, 0, None, 0, abi.F32)
fa.w, fa.y, fa.psum, fa.psumsq, fa.M, fa.K, fa.N, fa.rows_per_frustum, fa.dtype = fptr(wg), fptr(yg), fptr(ps), fptr(pq), M, K, N, 1024, abi.F32


def timed(fn, reps=200):
    g = torch.cuda.CUDAGraph()
    st = torch.cuda.Stream()
    fn_s = lambda: fn(C.c_void_p(st.cuda_stream))
    with torch.cuda.stream(st):
        for _ in range(3):
            fn_s()
    torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=st):
        for _ in range(20):
            fn_s()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps // 20):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (reps // 20 * 20)


def four(st):
    for a in ops:
        lib.t3d_fc_fwd(C.byref(a), st)


def four_sets(st):
    for x in rs1:
        lib.t3d_run_riders(C.byref(x), st)


print('GEMM %dx%dx%d' % (M, K, N))
print('(a) four stand-alone FC launches        %7.1f us' % timed(four))
print("(a') four one-op rider sets (256 thr)     %7.1f us" % timed(four_sets))
print('(b) one rider set, in-launch barriers   %7.1f us' % timed(lambda st: lib.t3d_run_riders(C.byref(rs), st)))
print('(d) the GEMM alone                      %7.1f us' % timed(lambda st: lib.t3d_pointmlp_fwd(C.byref(fa), st)))
print('(c) the GEMM with the set riding        %7.1f us' % timed(lambda st: lib.t3d_pointmlp_fwd_r(C.byref(fa), C.byref(rs), st)))
print("(c') the GEMM with ONE op riding         %7.1f us" % timed(lambda st: lib.t3d_pointmlp_fwd_r(C.byref(fa), C.byref(rs1[1]), st)))
# the `_r` kernel with a rider that does next to nothing (one block of a 256-element column sum): what the rider form itself costs the GEMM
cs_psum, cs_coef, cs_out = t(np.zeros((8, 32))), t(np.ones((3, 32))), t(np.zeros((8, 32)))
cs = abi.DyColsumArgs(fptr(cs_psum), fptr(cs_psum), fptr(cs_coef), 8, 32, 1, 128, 1.0, fptr(cs_out))
rs0 = sets.make([('t3d_dy_colsum', cs)])
print("(c0) the GEMM's rider form, trivial rider    %7.1f us" % timed(lambda st: lib.t3d_pointmlp_fwd_r(C.byref(fa), C.byref(rs0), st)))
print('timeouts', sets.timeouts())
