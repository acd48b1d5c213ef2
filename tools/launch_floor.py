import torch, time
x = torch.zeros(64, device='cuda')
s = torch.cuda.Stream()
for n in (100, 400):
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        for _ in range(n):
            x.add_(1.0)
    for _ in range(3): g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20): g.replay()
    torch.cuda.synchronize()
    print('graph nodes', n, 'per node us', (time.perf_counter() - t0) / 20 / n * 1e6)
# eager back-to-back
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(2000): x.add_(1.0)
torch.cuda.synchronize(); print('eager per launch us', (time.perf_counter() - t0) / 2000 * 1e6)
