"""python tools/micro/grid_barrier.py  (on the GPU box; builds tools/micro/libgb.so with hipcc if missing)."""
import ctypes, os, subprocess, time
import torch
here = os.path.dirname(os.path.abspath(__file__))
so = os.path.join(here, 'libgb.so')
if not os.path.exists(so):
    subprocess.check_call(['hipcc', '--offload-arch=gfx950', '-O3', '-shared', '-fPIC', os.path.join(here, 'grid_barrier.hip'), '-o', so])
lib = ctypes.CDLL(so)
vp = ctypes.c_void_p
lib.mb_stage.argtypes = [vp, vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, vp]
lib.mb_chain.argtypes = [vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, vp, ctypes.c_int, vp, ctypes.c_int]
s = torch.cuda.Stream()
big = torch.zeros(64 << 20, device='cuda')       # 256 MB: flushes L2 / MALL between replays when touched


def timed(fn, reps=50):
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(s):
        fn(s.cuda_stream)
    torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=s):
        fn(s.cuda_stream)
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        g.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6


for G, width, stride, mode in [(G, w, 1, m) for (G, w) in ((16, 4096), (32, 1024), (16, 65536)) for m in (0, 1, 2)]:
    n = 12
    buf = torch.zeros(2 * G * width, device='cuda')
    cnt = torch.zeros(4, dtype=torch.int32, device='cuda')

    def launches(st):
        for k in range(n):
            a = buf.data_ptr() + (k & 1) * G * width * 4
            b = buf.data_ptr() + ((k + 1) & 1) * G * width * 4
            assert lib.mb_stage(a, b, G, width, stride, st) == 0

    def chain(st):
        assert lib.mb_chain(buf.data_ptr(), G, width, n, cnt.data_ptr(), stride, st, mode) == 0

    def one(st):
        assert lib.mb_stage(buf.data_ptr(), buf.data_ptr() + G * width * 4, G, width, stride, st) == 0

    buf.zero_(); torch.cuda.synchronize()
    with torch.cuda.stream(s):
        chain(s.cuda_stream)
    torch.cuda.synchronize()
    ok = bool((buf[:G * width] == float(n)).all()) if n % 2 == 0 else bool((buf[G * width:] == float(n)).all())
    t_l, t_c, t_1 = timed(launches), timed(chain), timed(one)
    print(f'G={G} width={width} mode={mode}: {n} launches {t_l:.1f} us ({t_l / n:.2f}/stage), one chain launch {t_c:.1f} us '
          f'(stage+barrier {(t_c - t_1) / (n - 1):.2f} us), single launch {t_1:.1f} us, chain result correct: {ok}')

# ---- two independent chains of small dependent stages: 2 x n launches vs n launches that each carry both chains' blocks ----
for G, width in ((16, 4096), (8, 1024)):
    n = 12
    bufA, bufB = torch.zeros(2 * G * width, device='cuda'), torch.zeros(2 * G * width, device='cuda')
    both = torch.zeros(2 * 2 * G * width, device='cuda')

    def separate(st):
        for k in range(n):
            for buf in (bufA, bufB):
                a = buf.data_ptr() + (k & 1) * G * width * 4
                b = buf.data_ptr() + ((k + 1) & 1) * G * width * 4
                assert lib.mb_stage(a, b, G, width, 1, st) == 0

    def paired(st):
        for k in range(n):
            a = both.data_ptr() + (k & 1) * 2 * G * width * 4
            b = both.data_ptr() + ((k + 1) & 1) * 2 * G * width * 4
            assert lib.mb_stage(a, b, 2 * G, width, 1, st) == 0

    t_s, t_p = timed(separate), timed(paired)
    print(f'two chains of {n} stages, G={G} width={width}: {2 * n} launches {t_s:.1f} us, {n} paired launches {t_p:.1f} us '
          f'-> {(t_s - t_p) / n:.2f} us saved per pair')
