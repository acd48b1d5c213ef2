#!/usr/bin/env python3
"""What clock and power does the chip hold under the hot path's GEMM launches?  Runs one launch shape (T3D_ONLY-style: fwd 512 -> 256 by
default, or the whole step with --step) back to back for a few seconds while rocm-smi is polled from a child process."""
import ctypes as C
import os
import subprocess
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
from transferable3d_amd import abi
from transferable3d_amd.abi import fptr


def poll(out, stop):
    while not stop.is_set():
        try:
            r = subprocess.run(['/opt/rocm/bin/rocm-smi', '--showclocks', '--showpower', '--showtemp', '--csv'], capture_output=True, text=True, timeout=20)
            out.append((time.time(), r.stdout.strip()))
        except Exception as e:      # noqa
            out.append((time.time(), 'ERR %r' % (e,)))
        time.sleep(0.2)


def main():
    lib = abi.load(os.environ.get('T3D_LIB'))
    from bench_x3 import frag_planes
    M, K, N = 32768, 512, 256
    dev = 'cuda'
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    x = torch.randn(M, K, device=dev)
    sc, sh = torch.rand(K, device=dev) + 0.5, torch.randn(K, device=dev) * 0.1
    w = torch.randn(K, N, device=dev) / K ** 0.5
    y = torch.zeros(M, N, device=dev)
    p1, p2 = torch.zeros(M // 128, N, device=dev), torch.zeros(M // 128, N, device=dev)
    a = abi.PointMlpFwdArgs()
    a.a = abi.ActSrc(fptr(x), K, 0, fptr(sc), fptr(sh), 1, fptr(None), 0)
    a.w, a.bias, a.psum, a.psumsq, a.y = fptr(w), fptr(torch.zeros(N, device=dev)), fptr(p1), fptr(p2), fptr(y)
    a.M, a.K, a.N, a.rows_per_frustum = M, K, N, 1024
    keep = frag_planes(lib, w, s)
    a.w_x3, a.w_x3_stride = keep[0].data_ptr(), keep[2]
    os.environ['T3D_X3'] = '1'
    idle = []
    stop = threading.Event()
    th = threading.Thread(target=poll, args=(idle, stop)); th.start(); time.sleep(1.0); stop.set(); th.join()
    print('--- idle (columns: junction C, memory C, fclk, level, mclk, level, sclk, level, socclk, level, package W)'); print(idle[-1][1] if idle else 'no sample')
    # the other regimes, for comparison: the same layer on the fp32 MFMA (T3D_X3=0), a latency-bound narrow forward, a pure HBM stream
    xs = torch.randn(M, 128, device=dev)
    ws = torch.randn(128, 128, device=dev) / 11.0
    ys = torch.zeros(M, 128, device=dev)
    a2 = abi.PointMlpFwdArgs()
    a2.a = abi.ActSrc(fptr(xs), 128, 0, fptr(sc[:128].contiguous()), fptr(sh[:128].contiguous()), 1, fptr(None), 0)
    p3, p4 = torch.zeros(M // 128, 128, device=dev), torch.zeros(M // 128, 128, device=dev)
    a2.w, a2.bias, a2.psum, a2.psumsq, a2.y = fptr(ws), fptr(torch.zeros(128, device=dev)), fptr(p3), fptr(p4), fptr(ys)
    a2.M, a2.K, a2.N, a2.rows_per_frustum = M, 128, 128, 1024
    keep2 = frag_planes(lib, ws, s)
    a2.w_x3, a2.w_x3_stride = keep2[0].data_ptr(), keep2[2]
    big = torch.zeros(64 << 20, device=dev)
    big2 = torch.zeros(64 << 20, device=dev)

    def case_x3():
        os.environ['T3D_X3'] = '1'; lib.t3d_pointmlp_fwd(C.byref(a), s)

    def case_f32():
        os.environ['T3D_X3'] = '0'; lib.t3d_pointmlp_fwd(C.byref(a), s)

    def case_narrow():
        os.environ['T3D_X3'] = '1'; lib.t3d_pointmlp_fwd(C.byref(a2), s)

    def case_copy():
        big2.copy_(big)

    for secs, label, fn in ((4.0, 'forward 512 -> 256, three-term bf16 (the default)', case_x3), (4.0, 'forward 512 -> 256, fp32 MFMA (T3D_X3=0)', case_f32),
                            (4.0, 'forward 128 -> 128, three-term bf16', case_narrow), (4.0, '256 MB device-to-device copy', case_copy)):
        out, stop = [], threading.Event()
        th = threading.Thread(target=poll, args=(out, stop)); th.start()
        t0 = time.time(); n = 0
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        while time.time() - t0 < secs:
            for _ in range(200):
                fn()
            n += 200
            torch.cuda.synchronize()
        e1.record(); torch.cuda.synchronize()
        stop.set(); th.join()
        print('--- %s: %d launches, %.1f us each (incl. host gaps)' % (label, n, e0.elapsed_time(e1) * 1e3 / n))
        for t, txt in out[1:-1][:4]:
            row = [l for l in txt.splitlines() if l.startswith('card0')]
            print('  t=%.2f s  %s' % (t - t0, row[0] if row else txt))


if __name__ == '__main__':
    main()
