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
    print('--- idle'); print(idle[-1][1] if idle else 'no sample')
    for secs, label in ((4.0, 'back-to-back launches'),):
        out, stop = [], threading.Event()
        th = threading.Thread(target=poll, args=(out, stop)); th.start()
        t0 = time.time(); n = 0
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        while time.time() - t0 < secs:
            for _ in range(200):
                lib.t3d_pointmlp_fwd(C.byref(a), s)
            n += 200
            torch.cuda.synchronize()
        e1.record(); torch.cuda.synchronize()
        stop.set(); th.join()
        print('--- %s: %d launches, %.1f us each (incl. host gaps)' % (label, n, e0.elapsed_time(e1) * 1e3 / n))
        for t, txt in out[1:-1][:6]:
            print('t=%.2f' % (t - t0)); print(txt)


if __name__ == '__main__':
    main()
