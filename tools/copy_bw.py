import torch
dev='cuda'
for mb in (8, 16, 32, 64, 128, 256, 512):
    n = mb << 18
    a = torch.zeros(n, device=dev); b = torch.zeros(n, device=dev)
    for _ in range(5): b.copy_(a)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    R = 200
    e0.record()
    for _ in range(R): b.copy_(a)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / R * 1e3
    print('copy %4d MB: %7.1f us  %.2f TB/s (read + write)' % (mb, us, 2 * mb * 1.048576 / us))
    # read-only: sum
    for _ in range(3): a.sum()
    torch.cuda.synchronize()
    e0.record()
    for _ in range(R): a.sum()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / R * 1e3
    print('sum  %4d MB: %7.1f us  %.2f TB/s (read)' % (mb, us, mb * 1.048576 / us))
