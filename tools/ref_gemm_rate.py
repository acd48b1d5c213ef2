#!/usr/bin/env python3
"""Diagnosis only: what the vendor fp32 GEMM (torch.matmul -> rocBLAS / hipBLASLt) reaches on the layer shapes, as a yardstick
for the hand-written kernels (which also fuse batch-norm / ReLU / statistics / pooling)."""
import torch

torch.backends.cuda.matmul.allow_tf32 = False
M = 32768
for K, N in [(128, 1024), (512, 256), (256, 512), (128, 128), (64, 64), (4096, 4096)]:
    m = M if K != 4096 else 4096
    a, b = torch.randn(m, K, device='cuda'), torch.randn(K, N, device='cuda')
    for _ in range(3):
        c = a @ b
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        c = a @ b
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    print('matmul fp32 M%-6d K%-5d N%-5d %8.1f us  %6.1f TF/s' % (m, K, N, us, 2.0 * m * K * N / us / 1e6))
