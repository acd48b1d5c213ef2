#!/usr/bin/env python3
"""Headline benchmark: frustums/sec of one SEMI_MODEL A training step (seg PointNet + T-Net + box PointNet,
forward + backward + TF-form Adam) on synthetic B=32, N=1024, C=4 frustum batches resident in HBM
(BASELINE.json configs[1]), one process per GPU, data-parallel gradient all-reduce over RCCL for N > 1.

    python bench.py --gpus 1 --steps 50 --warmup 10
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0 (contract in the task description), with `roofline` for the dominant kernel
(per-launch HIP-event timing on the launch stream) and `cpu_baseline` (the oracle's torch-CPU restatement of
the same step timed on this host's cores; baseline, not target).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# the host driver of this pool supports dmabuf IPC only: without it RCCL's buffer exchange between the ranks fails with
# hipIpcGetMemHandle: invalid argument.  Must be in the environment before the HIP runtime starts.
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')

import numpy as np   # noqa: E402
import torch         # noqa: E402

HBM_PEAK_TBS = 8.0                 # MI355X_MICROARCH.md: HBM3E spec peak (about 6.3 TB/s achievable)
MFMA_F32_PEAK_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense fp32 matrix peak
MFMA_BF16_SUSTAINED_TFLOPS = 1860.0      # measured (tools/micro/mfma_rate.hip): 32 cycles per 32x32x16 bf16 MFMA per SIMD at 1.85 GHz under load
MFMA_BF16_PEAK_TFLOPS = 2500.0    # MI355X_MICROARCH.md: dense bf16 MFMA peak (~2.5 PFLOP/s; the 5 PF headline includes 2:1 sparsity)
SPLIT_GFLOP_PER_FRUSTUM = 3.638   # fwd+bwd, split-conv6 count (the algorithm actually run)
DENSE_GFLOP_PER_FRUSTUM = 6.856   # as-written dense concat count (reported for reference only)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=100)
    ap.add_argument('--warmup', type=int, default=20)
    ap.add_argument('--batch_size', type=int, default=32)
    ap.add_argument('--num_point', type=int, default=1024)
    ap.add_argument('--num_channel', type=int, default=4)
    ap.add_argument('--workload', choices=['A', 'boxpc', 'F'], default='A',
                    help='A = BASELINE configs[1] (the metric); boxpc / F = configs[2] / configs[3], informational')
    ap.add_argument('--no_graph', action='store_true', help='eager launches instead of hipGraph replay')
    ap.add_argument('--no_cpu_baseline', action='store_true')
    ap.add_argument('--no_other_configs', action='store_true',
                    help='skip the informational side runs of configs[2..4] (child processes, <= 20 steps each)')
    ap.add_argument('--cpu_steps', type=int, default=5)
    ap.add_argument('--cpu_one_thread', type=int, default=1, help='also time one step of the CPU baseline on a single thread')
    ap.add_argument('--dtype', choices=['f32', 'bf16'], default='f32',
                    help='f32: exact-fp32 MFMA path (configs[1..3]); bf16: bf16 storage + bf16 MFMA, fp32 accumulate (configs[4])')
    ap.add_argument('--cpu_threads', type=int, default=32,
                    help='host threads for the CPU baseline (torch CPU ops stop scaling / oversubscribe beyond this)')
    ap.add_argument('--profile_steps', type=int, default=5)
    ap.add_argument('--gemm_detail', action='store_true', help='per-shape GEMM timings on stderr')
    ap.add_argument('--call_detail', action='store_true', help='every launch of the step in order with its time, on stderr')
    return ap.parse_args()


LIB = None
CALLS = {}


def _null(ptr):
    return not bool(ptr)


def gemm_label_and_flops(name, a):
    """(label, flops) -- see gemm_work."""
    lab, fl, _ = gemm_work(name, a)
    return lab, fl


def gemm_work(name, a):
    """Mirror of the template dispatch in csrc/pointmlp.hip -> (rocprof kernel name, algorithmic FLOPs, algorithmic HBM
    bytes) of one launch.

    Algorithmic bytes: every tensor the LAUNCH needs counted ONCE, however many of the
    launch's workgroup kinds read it, and results in their final form (dW = K*N floats, not the per-split slabs):
      forward          4(MK + KN [+ MN if y is stored])
      fused backward   4(2MN [dz, y] + MK [the input: dW operand == ReLU mask / BN-bwd partial source of dX] + MK [dz_prev out]
                         [+ MK add_in] + 2KN [w in, dW out])          -- the judge's 4M(2N + 2K) plus the weight terms
      dgrad alone      4(2MN + KN + MK out [+ MK prev_y] [+ MK add_in]);   wgrad alone 4(MK + 2MN + KN)
      Gram forms       the same with K in place of N (the [M,N] tensor does not exist)."""
    import ctypes
    es = lambda dt: 2.0 if dt == 1 else 4.0       # bytes per element of a [M, C] layer tensor (t3d.h T3D_BF16 / T3D_F32)
    # fp32 layers run on the bf16 matrix pipe with three-term operands (csrc/pointmlp.hip PathX3: k_..._x3<...>) when the launch struct
    # asks for it AND the launcher's own rule takes it: the library says which (t3d_gemm_arithmetic), nothing here reads the environment
    x3f = lambda st, K, N: LIB.t3d_gemm_arithmetic(st.arith, 0, K, N, 0) == 2
    x3b = lambda st, K, N, kind=1: LIB.t3d_gemm_arithmetic(st.arith, 0, K, N, kind) == 2      # kind: t3d.h T3D_GEMM_* (the launcher asked)
    tag = lambda label: label.replace('<', '_x3<', 1)
    if name == 't3d_pool_bwd_stage1':
        gl, gf, gb = gemm_work('t3d_pointmlp_gram', a[0])
        q = a[2]
        # the activation panel is read once for G = a'a and the column sums; w, coef in; G, abar, P, rowconst, wc out
        by = es(a[0].a.dtype) * a[0].M * a[0].K + 4.0 * (2 * q.K * q.K + 2 * q.K * q.N + 2 * q.K)
        g0 = a[0]
        if g0.a.dtype == 1 and g0.K in (128, 256) and g0.rows_per_split % 128 == 0 and g0.M // g0.rows_per_split >= min(256, g0.M // 128) \
                and os.environ.get('T3D_GRAM1', '1') != '0':
            return 'k_pool_bwd_stage1_h<%d,%d>' % (g0.K, 128 if g0.K == 128 else 64), gf + 2.0 * q.K * q.K * q.N, by      # one-pass form
        lab = 'k_pool_bwd_stage1<%s>' % gl[gl.index('<') + 1:gl.index(',')]
        return (tag(lab) if g0.a.dtype == 0 and x3b(g0, g0.K, g0.K, 4) else lab), gf + 2.0 * q.K * q.K * q.N, by
    if name == 't3d_pool_bwd_stage2':
        dl, df, db = gemm_work('t3d_pointmlp_dgrad_gram', a[1])
        f = a[0]
        # + w in, dW out, G in, the B*N arg-max rows of the input
        by = db + 4.0 * (2 * f.K * f.N + f.K * f.K) + es(f.a.dtype) * f.B * f.N * f.K
        dm = int(os.environ.get('T3D_DGRAM1', '1'))
        if a[1].dtype == 1 and os.environ.get('T3D_GRAM1', '1') != '0' and ((dm >= 1 and f.K == 256) or (dm == 2 and f.K == 128)):
            return 'k_pool_bwd_stage2_h<%d,%d>' % (f.K, 128 if f.K == 128 else 64), df + 2.0 * f.K * f.K * f.N, by      # one-pass form
        lab = 'k_pool_bwd_stage2<%s>' % dl[dl.index('<') + 1:-1]
        return (tag(lab) if a[1].dtype == 0 and x3b(a[1], f.K, f.K, 5) else lab), df + 2.0 * f.K * f.K * f.N, by
    if name == 't3d_pointmlp_bwd':
        d, w = a
        dl, df, _ = gemm_work('t3d_pointmlp_dgrad', d)
        wl, wf, _ = gemm_work('t3d_pointmlp_wgrad', w)
        M, K, N = d.M, d.K, d.N
        by = es(d.dtype) * (2 * M * N + M * K * (2 + (0 if _null(d.add_in) else 1))) + 4.0 * 2 * K * N
        if d.dtype == 1 and w.a.dtype == 1 and ((K in (64, 128) and N in (64, 128)) or (K, N) in ((256, 128), (128, 256))) and w.rows_per_split % 128 == 0 and \
                M // w.rows_per_split >= min(256, M // 128) and os.environ.get('T3D_BWD1', '1') != '0':
            return 'k_pointmlp_bwd1<%d,%d,%d>' % (K, N, 64 if 256 in (K, N) else 128), df + wf, by      # one-pass form
        if d.dtype == 0 and w.a.dtype == 0 and K in (64, 128) and N in (64, 128) and w.rows_per_split % 128 == 0 and \
                M // w.rows_per_split >= min(256, M // 128) and os.environ.get('T3D_BWD1F', '1') != '0' and \
                (M // 128 < 256 or w.rows_per_split >= 256 or os.environ.get('T3D_BWD1F') == '2'):
            return 'k_pointmlp_bwd1f<%d,%d>' % (K, N), df + wf, by      # fp32 one-pass form (not taken at the headline size)
        lab = 'k_pointmlp_bwd<%s,%s>' % (dl[dl.index('<') + 1:-1], wl[wl.index('<') + 1:-1])
        return (tag(lab) if d.dtype == 0 and w.a.dtype == 0 and x3b(d, K, N, 1) else lab), df + wf, by
    if name == 't3d_pointmlp_dgrad_gram':
        # (the sparse arg-max rows S behind add_live are read only where a row received a hit -- a data-dependent few percent of
        # the rows: not counted; a dense add_in is a full pass)
        dense_add = (not _null(a.add_in)) and _null(a.add_live)
        by = es(a.dtype) * a.M * a.K * (2 + (0 if _null(a.prev_y) else 1) + (1 if dense_add else 0)) + 4.0 * a.K * a.K
        lab = 'k_pointmlp_dgrad_gram<%d>' % (128 if a.K % 128 == 0 and (a.M // 128) * (a.K // 128) >= 512 else 64)
        return (tag(lab) if a.dtype == 0 and x3b(a, a.K, a.K, 5) else lab), 2.0 * a.M * a.K * a.K, by
    if name == 't3d_pointmlp_gram':
        rps, tk, tn = ctypes.c_int(0), ctypes.c_int(0), ctypes.c_int(0)
        LIB.t3d_wgrad_plan(a.M, a.K, a.K, ctypes.byref(rps), ctypes.byref(tk), ctypes.byref(tn))
        t = tk.value if (rps.value == a.rows_per_split and tk.value == tn.value) else 64
        x3g = a.a.dtype == 0 and x3b(a, a.K, a.K, 4)
        if x3g:
            t = 64      # (the x3 Gram kernels exist for 64 x 64 tiles only: three accumulator sets keep G bitwise symmetric)
        lab = 'k_pointmlp_gram<%d,%d>' % (t, t)
        return (tag(lab) if x3g else lab), 2.0 * a.M * a.K * a.K, es(a.a.dtype) * a.M * a.K + 4.0 * a.K * a.K
    flops = 2.0 * a.M * a.K * a.N
    if name == 't3d_pointmlp_fwd':
        by = es(a.a.dtype) * a.M * a.K + es(a.dtype) * (a.K * a.N + (0 if _null(a.y) else a.M * a.N))
        if a.a.dtype == 0 and a.K <= 4 and not _null(a.y) and _null(a.pmax) and _null(a.rowbias) and a.N in (64, 128) \
                and os.environ.get('T3D_FWD_TINYK', '1') != '0':
            return 'k_pointmlp_fwd_tinyk<%d>' % a.N, flops, by      # first layer of a net: the register kernel (bf16 and fp32)
        if a.dtype == 1 and a.a.dtype == 1 and _null(a.a.sub) and (a.K in (64, 128) or (a.K == 256 and os.environ.get('T3D_FWD_RES', '1') == '2')) and os.environ.get('T3D_FWD_RES', '1') != '0':
            return 'k_pointmlp_fwd_res<%d,%d>' % (128 if a.N % 128 == 0 else 64, a.K // 64), flops, by      # activation-resident bf16 forward
        if a.dtype == 0 and a.a.dtype == 0 and _null(a.a.sub) and a.K % 16 == 0 and x3f(a, a.K, a.N):
            w8 = int(os.environ.get('T3D_X3_W8', '1'))      # eight-wave 128 x 256 tiles (csrc/pointmlp.hip t3d_x3_fwd)
            if w8 and a.N % 256 == 0 and (w8 == 2 or (a.M // 128) * (a.N // 256) >= 512):
                return 'k_pointmlp_fwd_w8_x3<256>', flops, by
            return 'k_pointmlp_fwd_x3<%d>' % (128 if a.N % 128 == 0 and (a.M // 128) * (a.N // 128) >= 512 else 64), flops, by
        if a.dtype == 0 and os.environ.get('T3D_FWD_POOL', '1') != '0' and _null(a.y) and not _null(a.pmax) and _null(a.a.sub) and a.K == 128 and \
                a.N % 128 == 0 and a.N >= 256:
            return 'k_pointmlp_fwd_pool<128,32,8>', flops, by
        return 'k_pointmlp_fwd<%d>' % (128 if a.N % 128 == 0 and (a.M // 128) * (a.N // 128) >= 512 else 64), flops, by
    if name == 't3d_pointmlp_dgrad':
        by = es(a.dtype) * (2 * a.M * a.N + a.K * a.N + a.M * a.K * (1 + (0 if _null(a.prev_y) else 1) + (0 if _null(a.add_in) else 1)))
        lab = 'k_pointmlp_dgrad<%d>' % (128 if a.K % 128 == 0 and (a.M // 128) * (a.K // 128) >= 512 else 64)
        return (tag(lab) if a.dtype == 0 and not _null(a.dy.dz) and x3b(a, a.K, a.N, 2) else lab), flops, by
    rps, tk, tn = ctypes.c_int(0), ctypes.c_int(0), ctypes.c_int(0)
    LIB.t3d_wgrad_plan(a.M, a.K, a.N, ctypes.byref(rps), ctypes.byref(tk), ctypes.byref(tn))
    if rps.value != a.rows_per_split:
        tk.value, tn.value = (128 if a.K > 64 else 64), (128 if a.N % 128 == 0 else 64)
    by = es(a.a.dtype) * a.M * a.K + es(a.dy.dtype) * 2 * a.M * a.N + 4.0 * a.K * a.N
    if a.K <= 4 and a.a.dtype == 0 and a.dy.dtype == 0 and not _null(a.dy.dz) and a.N in (64, 128) and os.environ.get('T3D_WGRAD_TINYK', '1') != '0':
        return 'k_pointmlp_wgrad_tinyk<%d>' % a.N, flops, by      # first layer of a net, fp32: the register kernel
    lab = 'k_pointmlp_wgrad<%d,%d>' % (tk.value, tn.value)
    return (tag(lab) if a.dy.dtype == 0 and a.a.dtype == 0 and _null(a.a.sub) and not _null(a.dy.dz) and a.K % tk.value == 0 and x3b(a, a.K, a.N, 3) else lab), flops, by


def gemm_arithmetic_report(rt, plans, dtype):
    """What the per-point GEMM launches of the step multiply with, as the LIBRARY answers for each launch struct of the plans
    (t3d_gemm_arithmetic on the struct's `arith` request, element type and shape) -- not what the environment of this process says."""
    from transferable3d_amd import abi
    counts, other = {}, []
    for plan in plans:
        for name, _, arg in plan.calls:
            base = name[:-2] if name.endswith('_r') else name
            if not base.startswith(('t3d_pointmlp', 't3d_pool_bwd_stage')) or base == 't3d_pointmlp_dgrad_narrow' or arg is None:
                continue
            st = (arg[1] if base == 't3d_pool_bwd_stage2' else arg[0]) if isinstance(arg, tuple) else arg
            K, N = st.K, getattr(st, 'N', st.K)
            dt = st.dtype if hasattr(st, 'dtype') else (st.dy.dtype if hasattr(st, 'dy') else st.a.dtype)
            kind = {'t3d_pointmlp_fwd': 0, 't3d_pointmlp_bwd': 1, 't3d_pointmlp_dgrad': 2, 't3d_pointmlp_wgrad': 3, 't3d_pointmlp_gram': 4,
                    't3d_pool_bwd_stage1': 4, 't3d_pointmlp_dgrad_gram': 5, 't3d_pool_bwd_stage2': 5}.get(base, 1)      # t3d.h T3D_GEMM_*
            took = abi.ARITH_NAMES[LIB.t3d_gemm_arithmetic(st.arith, dt, K, N, kind)]
            counts[took] = counts.get(took, 0) + 1
            if dt == 0 and took != rt.gemm_arithmetic:
                other.append('%s %dx%d' % (base[4:], K, N))
    text = {'bf16': 'bf16 operands, fp32 accumulate',
            'fp32_mfma': 'fp32 MFMA (v_mfma_f32_32x32x2_f32)',
            'bf16x3': 'fp32 operands as three exact bf16 terms, six bf16 MFMAs per multiply-add, fp32 accumulate (errors vs fp64 at or below '
                      'the fp32-MFMA kernels: profiles/r04_x3_accuracy.log)'}
    return {'requested': 'bf16' if dtype == 'bf16' else rt.gemm_arithmetic, 'launches_by_arithmetic': counts,
            'launches_not_on_the_requested_arithmetic': other,
            'reported_by': 't3d_gemm_arithmetic (the library, per launch struct)',
            'description': text['bf16' if dtype == 'bf16' else rt.gemm_arithmetic]}


def profile_kernels(plans, steps, repeat=4):
    """Per-launch HIP events (torch events on the stream the kernels are launched on), eager replay.  Every launch of the
    step is issued `repeat` times back to back between its two events and the interval divided by `repeat`: a single eager
    launch of a 5-30 us kernel is bounded by the host's submission time (~5 us per launch from Python), not by the kernel.
    (Repeating is harmless here: this pass runs after the timed region and after the loss was read.)"""
    acc = {}
    detail = {}
    global CALLS
    CALLS = {}
    for _ in range(steps):
        evs = []
        for plan in plans:
            s = plan.rt.stream()
            for name, call, arg in plan.calls:
                if name.startswith('__'):          # lane-join marker, not a launch
                    continue
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _r in range(repeat):
                    rc = call(s)
                    assert rc == 0, (name, rc)
                e1.record()
                evs.append((name, arg, e0, e1))
        # a GEMM launch that hosts riders (`_r`): also time the SAME kernel body on the same arguments without them (the plain
        # entry point), so that the hosted duration -- which is what the step pays and what rocprofv3 reports for k_..._r -- can be
        # told apart from the GEMM's own quality
        alone = {}
        for ci, (name, arg, _, _) in enumerate(evs):
            if not name.endswith('_r') or arg is None:
                continue
            import ctypes
            fn = getattr(LIB, name[:-2])
            refs = [ctypes.byref(a) for a in (arg if isinstance(arg, tuple) else (arg,))]
            s0 = plans[0].rt.stream()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _r in range(repeat):
                assert fn(*refs, s0) == 0
            e1.record()
            alone[ci] = (e0, e1)
        torch.cuda.synchronize()
        for ci, (name, arg, e0, e1) in enumerate(evs):
            dt = e0.elapsed_time(e1) * 1e-3 / repeat
            CALLS.setdefault(ci, [name, arg, 0.0])[2] += dt / steps
            # `_r`: the same GEMM with small launches of an independent chain riding in it (schedule.py); kernel name k_..._r<...>
            base = name[:-2] if name.endswith('_r') else name
            label, flops, nbytes = (gemm_work(base, arg) if base.startswith(('t3d_pointmlp', 't3d_pool_bwd_stage')) and
                                    base != 't3d_pointmlp_dgrad_narrow' else (name, 0.0, 0.0))
            if base != name and '<' in label and label.split('<')[0] in ('k_pointmlp_fwd', 'k_pointmlp_bwd', 'k_pointmlp_wgrad',
                                                                        'k_pool_bwd_stage1', 'k_pool_bwd_stage2', 'k_pointmlp_fwd_x3', 'k_pointmlp_fwd_w8_x3',
                                                                        'k_pointmlp_bwd_x3', 'k_pool_bwd_stage1_x3', 'k_pool_bwd_stage2_x3'):
                label = label.replace('<', '_r<', 1)
            d = acc.setdefault(label, [0.0, 0, 0.0, 0.0, 0.0])
            d[0] += dt
            d[1] += 1
            d[2] += flops
            d[3] += nbytes
            d[4] += (alone[ci][0].elapsed_time(alone[ci][1]) * 1e-3 / repeat) if ci in alone else dt      # the GEMM alone
            if flops:
                a0 = (arg[1] if base == 't3d_pool_bwd_stage2' else arg[0]) if isinstance(arg, tuple) else arg
                dd = detail.setdefault('%s M%d K%d N%d' % (label, a0.M, a0.K, getattr(a0, 'N', a0.K)), [0.0, 0, flops, nbytes])
                dd[0] += dt
                dd[1] += 1
    return acc, detail


from transferable3d_amd.build import lib_source_hash   # noqa: E402  (shared with tools/pmc_traffic.py)


def pmc_traffic(label, args):
    """HBM bytes per launch of `label` from a committed PMC summary (profiles/*pmc_traffic*.json, written by tools/pmc_traffic.py;
    counters cannot be read from inside the process being timed).  Only a summary taken for THIS workload, problem size and dtype
    counts (the newest by file name); `stale` tells whether the kernels' sources changed since it was taken."""
    import glob
    want = {'workload': args.workload, 'B': args.batch_size, 'N': args.num_point, 'C': args.num_channel, 'dtype': args.dtype}
    for path in sorted(glob.glob(os.path.join(ROOT, 'profiles', '*pmc_traffic*.json')), reverse=True):
        try:
            with open(path) as fh:
                z = json.load(fh)
            meta = z.get('_meta', {})
            if any(meta.get(k) != v for k, v in want.items()):
                continue
            return z['kernels'][label]['bytes_per_launch'], meta.get('lib_source_hash') != lib_source_hash()
        except (OSError, KeyError, ValueError):
            continue
    return None, None


def cpu_model():
    try:
        with open('/proc/cpuinfo') as fh:
            for line in fh:
                if line.startswith('model name'):
                    return line.split(':', 1)[1].strip()
    except OSError:
        pass
    return 'unknown'


def cpu_baseline(args, batch):
    """The oracle's torch-CPU fp32 restatement of the identical step (fwd + bwd + TF-form Adam) on this host's cores: 3 warm-up +
    `--cpu_steps` (>= 5) timed steps on `--cpu_threads` threads, then one warm-up + three timed steps (median) on ONE thread.  TF1 itself
    cannot run, so kind = "port".  Baseline, not target."""
    from oracle import ref_torch as R
    C = args.num_channel
    c = R.default_config()
    batch = dict(batch)
    batch['dropout_masks'] = {'inst_seg/dp1': (np.random.RandomState(1).uniform(size=(args.batch_size, args.num_point, 128)) < 0.5)
                              .astype(np.float32)}

    def timed(threads, warm, n):
        torch.set_num_threads(threads)
        P = R.init_params(np.random.RandomState(0), R.layer_table(C, 'A'), dtype=torch.float32)
        names = R.trainable_names(P)
        m = {k: torch.zeros_like(P[k]) for k in names}
        v = {k: torch.zeros_like(P[k]) for k in names}
        times = []
        for it in range(warm + n):
            t0 = time.perf_counter()
            _, _, grads, ema = R.model_a_forward_backward(P, batch, c, dtype=torch.float32)
            R.adam_tf_step(P, grads, m, v, it + 1, 1e-3)
            for k, val in ema.items():
                P[k] = val.detach()
            times.append(time.perf_counter() - t0)
        return float(np.median(times[warm:]))

    threads = min(os.cpu_count(), args.cpu_threads)
    t_multi = timed(threads, 3, max(args.cpu_steps, 5))
    t_one = timed(1, 1, 3) if args.cpu_one_thread else None      # median of 3 timed steps
    return {'value': args.batch_size / t_multi, 'unit': 'frustums/s', 'cores': threads, 'kind': 'port',
            'host_cpu': cpu_model(), 'host_logical_cpus': os.cpu_count(),
            'one_thread_value': (args.batch_size / t_one) if t_one else None,
            'sample': '%d timed steps (median; after 3 warm-up) of the same B=%d N=%d C=%d fp32 fwd+bwd+Adam step on %d threads, '
                      'and 3 timed steps (median; after 1 warm-up) on one thread; torch-CPU restatement of the reference graph '
                      '(oracle/ref_torch.py; TF1 not installable)' % (max(args.cpu_steps, 5), args.batch_size, args.num_point, C, threads)}


def spawn_ranks(args):
    """`python bench.py --gpus N` without a launcher: start N ranks as CHILD processes (torch.distributed.run) before this process has
    touched the GPU, hand their one JSON line through, exit with their code.  (A process that has initialised the GPU must never be
    replaced by exec on this pool; a child is fine.)"""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus), '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, T3D_BENCH_CHILD='1')
    return subprocess.call(cmd, env=env)


class Watchdog:
    """Exits the process (a plain exit: no exec, no re-launch) with a message if the guarded section takes longer than `seconds` --
    a collective that never completes must end the run with an explanation, not hang the node until the driver's limit.  `on_fire`
    may print what is already known first."""

    def __init__(self, seconds, what, on_fire=None, code=3):
        import threading
        self.what, self.seconds, self.on_fire, self.code = what, seconds, on_fire, code
        self.t = threading.Timer(seconds, self._fire)
        self.t.daemon = True

    def _fire(self):
        sys.stderr.write('bench.py: %s did not finish within %d s (rank %s of %s) -- giving up\n' %
                         (self.what, self.seconds, os.environ.get('RANK', '0'), os.environ.get('WORLD_SIZE', '1')))
        sys.stderr.flush()
        code = self.code
        try:
            if self.on_fire is not None:
                code = self.on_fire()
        finally:
            os._exit(code)

    def __enter__(self):
        self.t.start()
        return self

    def __exit__(self, *exc):
        self.t.cancel()
        return False


def other_configs(args):
    """BASELINE.json configs[2], configs[3] (one replica), configs[4] (one replica) and the reference's own problem size: <= 20 timed
    steps each in a child process of this script (own Runtime, own hipGraph), reported beside the headline -- informational, never
    `value`."""
    import subprocess
    # ... and the reference's own problem size (train_semisup.py:34-36,61: 2048 points, RGB on): B=32 N=2048 C=6 fp32
    # ... and the headline workload as the software-pipelined program (step.PipelinedStep: the seg forward of step k+1 beside the
    # T-Net / box chain of step k; bit-identical, opt-in: T3D_PIPELINE=1)
    runs = [('boxpc', 'f32', [], {}), ('F', 'f32', [], {}), ('A', 'bf16', ['--batch_size', '128', '--num_point', '2048'], {}),
            ('A', 'f32', ['--num_point', '2048', '--num_channel', '6'], {}), ('A', 'f32', [], {'T3D_PIPELINE': '1'})]
    out = []
    for wl, dt, extra, env_over in runs:
        cmd = [sys.executable, os.path.abspath(__file__), '--workload', wl, '--dtype', dt, '--steps', '20', '--warmup', '5',
               '--profile_steps', '2', '--no_cpu_baseline', '--no_other_configs'] + extra
        try:
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=240,
                               env=dict(os.environ, WORLD_SIZE='1', RANK='0', LOCAL_RANK='0', **env_over))
            d = json.loads(r.stdout.strip().split('\n')[-1])
            rf = d.get('roofline') or {}
            out.append({'workload': d['config']['workload'] + (' [software-pipelined program]' if d['config'].get('pipelined') else ''),
                        'dtype': d['dtype'], 'value': d['value'], 'unit': d['unit'],
                        'ms_per_step': d['ms_per_step'], 'steps': d['steps'],
                        'rider_barrier_timeouts': ((d['config'].get('schedule') or {}).get('rider_barrier_timeouts', 0)),
                        'roofline': {'kernel': rf.get('kernel'), 'bound': rf.get('bound'), 'frac': rf.get('frac'),
                                     'avg_launch_us': rf.get('avg_launch_us')}})
        except Exception as e:      # a failed side run must not cost the headline line
            out.append({'workload': wl, 'dtype': dt, 'error': repr(e)[:200]})
    return out


def main():
    args = parse()
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        sys.exit(spawn_ranks(args))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    assert torch.cuda.is_available(), 'bench.py needs an MI355X (no CPU fallback on the product path)'
    local_rank %= torch.cuda.device_count()      # (T3D_DIST_BACKEND=gloo runs: several ranks may share the one GPU of a test box)
    torch.cuda.set_device(local_rank)
    dist = None
    # T3D_FORCE_DIST=1: take the multi-rank code path (RCCL init, all-reduce between the two graphs) with a single rank, so
    # that it can be exercised on a 1-GPU box
    use_dist = world > 1 or os.environ.get('T3D_FORCE_DIST', '0') == '1'
    saved_stdout = None
    if use_dist:
        # RCCL prints a version banner on the C-level stdout; the contract is ONE JSON line there.  Route fd 1 to stderr for the
        # duration of the run and put it back just before the line is printed.
        sys.stdout.flush()
        saved_stdout = os.dup(1)
        os.dup2(2, 1)
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29511')
        backend = os.environ.get('T3D_DIST_BACKEND', 'nccl')       # 'gloo': the multi-rank flow on a box with fewer GPUs than ranks
        import datetime
        tmo = datetime.timedelta(seconds=int(os.environ.get('T3D_DIST_TIMEOUT_S', '120')))
        with Watchdog(int(tmo.total_seconds()) + 30, 'the rendezvous of the %d ranks (init_process_group, %s)' % (world, backend)):
            if backend == 'nccl':
                dist.init_process_group('nccl', rank=rank, world_size=world, device_id=torch.device('cuda', local_rank), timeout=tmo)
            else:
                dist.init_process_group(backend, rank=rank, world_size=world, timeout=tmo)
        assert dist.get_world_size() == world and (world == args.gpus or os.environ.get('T3D_FORCE_DIST', '0') == '1'), \
            'process group of %d ranks, --gpus %d' % (dist.get_world_size(), args.gpus)
        # the first collective (communicator set-up over xGMI) under a watchdog of its own: it either completes or the run ends with a
        # message -- it must not be able to hang the one scaling run the driver makes
        with Watchdog(int(os.environ.get('T3D_FIRST_COLLECTIVE_TIMEOUT_S', '180')), 'the first all-reduce of the %d ranks (%s)' % (world, backend)):
            probe = torch.ones(1024, device='cuda')
            dist.all_reduce(probe)
            torch.cuda.synchronize()
            assert float(probe[0].item()) == float(world), 'first all-reduce returned %r, expected %d' % (float(probe[0].item()), world)

    from transferable3d_amd.engine import Runtime
    from transferable3d_amd.step import build_training_step
    from transferable3d_amd.synthetic import make_batch

    B, N, C = args.batch_size, args.num_point, args.num_channel
    rt = Runtime()
    global LIB
    LIB = rt.lib
    from transferable3d_amd import abi as _abi
    lib_hash = _abi.source_hash_of(rt.lib)      # (abi.load has already refused a library that was not built from these sources)
    desc = {'A': 'seg-PointNet + T-Net + box-est fwd+bwd+Adam (SEMI_MODEL A)', 'boxpc': 'Box-PC Fit net fwd+bwd+Adam (train_boxpc.py path)',
            'F': 'SEMI_MODEL F stage c (frozen seg + Box-PC branch, var_list optimiser) fwd+bwd+Adam'}[args.workload]
    # the SAME step object the drivers run and tests/test_step_gpu.py checks against the oracle trajectory: device schedules, the
    # seg head's in-kernel dropout, forward, backward, TF-form Adam; data parallel: gradient buckets all-reduced beside the backward
    # T3D_PIPELINE=1 (single replica, SEMI_MODEL A, fp32): the software-pipelined step (step.PipelinedStep: the seg forward of step
    # k+1 beside the T-Net / box chain of step k; bit-identical to the one-step-at-a-time program).  Measured SLOWER on one MI355X
    # (1.501 vs 1.486 ms, same-box A/B gpurun_out/r03/ab_v5): off by default.
    pipelined = (world == 1 and not use_dist and args.workload == 'A' and args.dtype == 'f32' and
                 os.environ.get('T3D_PIPELINE', '0') == '1')
    if pipelined:
        from transferable3d_amd.step import build_pipelined_step
        trainstep, ctxs = build_pipelined_step(rt, B, N, C, use_hip_graph=not args.no_graph, inline_dropout=True, dropout_seed=1234, seed=0)
        g, model = ctxs[0]['g'], ctxs[0]['model']
        batch = make_batch(B, N, C, seed=1234 + rank)
        for cx in ctxs:                             # the same synthetic batch every step: both contexts hold it
            cx['model'].inputs.load(batch)
        loss_buf = lambda: trainstep.loss(trainstep.n_runs - 1)
    else:
        # Several ranks over RCCL: the HEADLINE is the library's default for more than one rank -- the gradient all-reduce issued by the
        # host between graph segments, the plain use of the collective library (step.TrainStep: the one-graph form is the default on
        # ONE rank only, where it has been run; no round has had two GPUs).  The step with its all-reduce captured in one graph is an
        # informational second leg (config.dp.modes), <= 20 steps, under a watchdog; it never replaces the headline figures.
        # T3D_DP_ONE_GRAPH=1 makes the one-graph form the (only) headline program, as on one rank.
        safe_first = use_dist and os.environ.get('T3D_DIST_BACKEND', 'nccl') == 'nccl' and 'T3D_DP_ONE_GRAPH' not in os.environ and world > 1
        g, model, trainstep, loss_t = build_training_step(
            rt, args.workload, B, N, C, world=world, rank=rank, process_group=dist.group.WORLD if use_dist else None,
            force_dist=use_dist and world == 1, flat_allreduce=os.environ.get('T3D_DP_FLAT', '1') == '1',
            use_hip_graph=not args.no_graph, inline_dropout=True, dropout_seed=1234, seed=0, dtype=args.dtype)
        loss_buf = lambda: loss_t
        batch = make_batch(B, N, C, seed=1234 + rank, boxpc=args.workload == 'boxpc')  # per-rank shard (weak scaling)
        if args.workload == 'F':
            batch['is_data_2D'][::2] = 1
        model.inputs.load(batch)
    torch.cuda.synchronize()
    use_graph = trainstep.want_graph
    step = trainstep.run
    # multi-rank: the capture of the step (with its collective inside on RCCL), the warm-up and the timed region run under a watchdog
    # too -- a replica that never arrives must end every rank with a message, not hang the node until the driver's limit
    run_guard = Watchdog(int(os.environ.get('T3D_RUN_TIMEOUT_S', '300')), 'the capture / warm-up / timed steps of the %d ranks' % world) \
        if dist is not None else None
    if run_guard is not None:
        run_guard.__enter__()
    step()                     # eager: loads the code objects, sets the communicator up (outside any capture)
    step()                     # captures the hipGraph segment(s) and replays
    torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    # the metric is B / MEDIAN step time.  Events on the launch stream every `chunk` steps (ten chunks over the timed
    # region) give the distribution; the wall clock over exactly K steps between the barriers gives the mean (second field).
    chunk = max(1, args.steps // 10)
    marks = [torch.cuda.Event(enable_timing=True)]
    t0 = time.perf_counter()
    marks[0].record()
    for i in range(args.steps):
        step()
        if (i + 1) % chunk == 0 or i + 1 == args.steps:
            marks.append(torch.cuda.Event(enable_timing=True))
            marks[-1].record()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    bounds = list(range(chunk, args.steps, chunk)) + [args.steps]
    per_step = [marks[k + 1].elapsed_time(marks[k + 2]) * 1e-3 / (bounds[k + 1] - bounds[k]) for k in range(len(bounds) - 1)]
    per_step.insert(0, marks[0].elapsed_time(marks[1]) * 1e-3 / bounds[0])
    median_step = float(np.median(per_step))
    if dist is not None:
        t = torch.tensor([elapsed, median_step], device='cuda', dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, median_step = float(t[0].item()), float(t[1].item())
    if run_guard is not None:
        run_guard.__exit__(None, None, None)
    if trainstep.dist:          # five more steps (every rank) with events around the bucket waits: the exposed all-reduce time
        trainstep.time_waits = True
        for _ in range(5):
            step()
        torch.cuda.synchronize()
        trainstep.time_waits = False
    loss = float(loss_buf().item())
    assert np.isfinite(loss), 'non-finite loss'
    # a rider barrier that ever gave up waiting (t3d.h t3d_rider_set.sync) would mean wrong results: fail loudly
    rider_timeouts = trainstep.rider_timeouts()
    assert rider_timeouts == 0, 'a rider barrier timed out: the results of the step are invalid'

    roofline = None
    cpu = None
    if rank == 0 and args.profile_steps > 0:
        # per-kernel timing for the roofline object (eager, per-launch events on the launch stream)
        # the launches of the program that was timed (the scheduled one: step.TrainStep._overlap), not of the plans it was made from
        if pipelined:
            trainstep.run(last=True)      # (untimed) finish the step whose forward the last timed call started: no forward is pending
            torch.cuda.synchronize()
            plans = trainstep.profile_plans()
        else:
            prog = trainstep.cache[True]['prog'] if True in trainstep.cache else []
            plans = [x for kind, x in prog if kind == 'run'] if (prog and all(kind == 'run' for kind, _ in prog)) else [g.pre, g.fwd, g.bwd, g.opt]
        acc, detail = profile_kernels(plans, args.profile_steps)
        total = sum(v[0] for v in acc.values())
        dom = max((k for k in acc if k.startswith(('k_pointmlp', 'k_pool_bwd_stage'))), key=lambda k: acc[k][0])
        tsec, n, fl, nby, tsec_alone = acc[dom]
        # The roofline that bounds the dominant kernel, from its ALGORITHMIC work (gemm_work: every tensor once): the larger of
        # FLOPs / MFMA peak and bytes / HBM peak.  fp32 configs: MFMA; bf16 config 4: HBM.
        mfma_peak = MFMA_F32_PEAK_TFLOPS if args.dtype == 'f32' else MFMA_BF16_PEAK_TFLOPS
        mfma_bound = fl / (mfma_peak * 1e12) >= nby / (HBM_PEAK_TBS * 1e12)
        achieved = fl / tsec / 1e12
        hbm_achieved = nby / tsec / 1e9
        is_gemm = lambda k: k.startswith(('k_pointmlp', 'k_pool_bwd_stage'))
        gemm_t = sum(v[0] for k, v in acc.items() if is_gemm(k))
        gemm_t_alone = sum(v[4] for k, v in acc.items() if is_gemm(k))
        gemm_f = sum(v[2] for k, v in acc.items() if is_gemm(k))
        gemm_b = sum(v[3] for k, v in acc.items() if is_gemm(k))
        traffic, stale = pmc_traffic(dom, args)
        if traffic is not None and nby / n > 1.02 * traffic and not stale:
            sys.stderr.write('WARNING: algorithmic bytes per launch of %s (%.1f MB) exceed the PMC-measured HBM traffic (%.1f MB): the '
                             'byte model over-counts or profiles/pmc_traffic.json is stale\n' % (dom, nby / n / 1e6, traffic / 1e6))
        step_flops = SPLIT_GFLOP_PER_FRUSTUM * (N / 1024.0) * 1e9 * B        # per-point terms dominate: scale with N
        emulated = args.dtype == 'f32' and '_x3' in dom.split('<')[0]
        roofline = {'bound': 'mfma' if mfma_bound else 'hbm', 'kernel': dom,
                    # fp32 layers on the bf16 matrix pipe: every operand is the exact sum of three bf16 terms, six bf16 MFMAs with fp32
                    # accumulation per fp32 multiply-add (csrc/pointmlp.hip PathX3; results agree with the fp32-MFMA kernels to fp32
                    # rounding, tests/ run both).  `peak` stays the fp32-MFMA peak the fp32 FLOP count is priced against; the
                    # emulation's own ceiling is the dense bf16 peak / 6.
                    'emulated': 'bf16x3' if emulated else None,
                    'emulated_ceiling_tflops': (MFMA_BF16_PEAK_TFLOPS / 6.0) if emulated else None,
                    'frac_of_emulated_ceiling': (achieved / (MFMA_BF16_PEAK_TFLOPS / 6.0)) if emulated else None,
                    # what the bf16 matrix pipe SUSTAINS on this chip: 32 cycles per v_mfma_f32_32x32x16_bf16 at the 1.85 GHz it holds
                    # under load = 1.86 PFLOP/s (tools/micro/mfma_rate.hip, docs/EXPERIMENTS.md round 5), six products per multiply-add
                    'emulated_ceiling_sustained_tflops': (MFMA_BF16_SUSTAINED_TFLOPS / 6.0) if emulated else None,
                    'frac_of_emulated_ceiling_sustained': (achieved / (MFMA_BF16_SUSTAINED_TFLOPS / 6.0)) if emulated else None,
                    'achieved': achieved if mfma_bound else hbm_achieved,
                    'peak': mfma_peak if mfma_bound else HBM_PEAK_TBS * 1e3,
                    'unit': 'TFLOP/s' if mfma_bound else 'GB/s',
                    'frac': achieved / mfma_peak if mfma_bound else hbm_achieved / (HBM_PEAK_TBS * 1e3),
                    'mfma': {'achieved_tflops': achieved, 'frac': achieved / mfma_peak},
                    'hbm': {'achieved_gbs': hbm_achieved, 'frac': hbm_achieved / (HBM_PEAK_TBS * 1e3),
                            'algorithmic_bytes_per_launch': nby / n},
                    'traffic': traffic,
                    'traffic_unit': 'HBM bytes per launch, rocprofv3 PMC passes of this workload (profiles/pmc_traffic.json)',
                    'traffic_over_algorithmic': (traffic / (nby / n)) if traffic else None,
                    'traffic_summary_predates_this_build': stale,
                    'avg_launch_us': tsec / n * 1e6, 'launches_per_step': n // args.profile_steps,
                    # `_r` kernels: the launch also runs small ops of the independent chain (schedule.py) in its first workgroups and
                    # lasts as long as the longer of the two; the same GEMM on the same arguments WITHOUT riders:
                    'gemm_alone': ({'avg_launch_us': tsec_alone / n * 1e6, 'achieved': fl / tsec_alone / 1e12,
                                    'frac': fl / tsec_alone / 1e12 / mfma_peak} if dom.split('<')[0].endswith('_r') else None),
                    'flops_per_launch': fl / n,
                    'all_gemm_kernels': {'achieved': gemm_f / gemm_t / 1e12, 'frac': gemm_f / gemm_t / 1e12 / mfma_peak,
                                         'hbm_achieved_gbs': gemm_b / gemm_t / 1e9, 'hbm_frac': gemm_b / gemm_t / 1e9 / (HBM_PEAK_TBS * 1e3),
                                         'share_of_step_kernel_time': gemm_t / total,
                                         'frac_without_riders': gemm_f / gemm_t_alone / 1e12 / mfma_peak},
                    'whole_step': None if args.workload != 'A' else {'gflop_per_frustum_split': step_flops / B / 1e9,
                                   # what the GEMM kernels actually execute (the Gram-form backward of the pooled layers needs
                                   # fewer FLOPs than the split count the roofline figure is quoted on)
                                   'gflop_per_frustum_executed': gemm_f / args.profile_steps / B / 1e9,
                                   'achieved_executed': gemm_f / args.profile_steps / median_step / 1e12,
                                   'achieved': step_flops / median_step / 1e12,
                                   'frac': step_flops / median_step / 1e12 / mfma_peak,
                                   'algorithmic_gemm_bytes_per_step': gemm_b / args.profile_steps,
                                   'hbm_frac': gemm_b / args.profile_steps / median_step / (HBM_PEAK_TBS * 1e12)},
                    'per_kernel_us_per_step': {k: v[0] / args.profile_steps * 1e6 for k, v in sorted(acc.items(), key=lambda kv: -kv[1][0])}}
        if args.call_detail:
            for ci in sorted(CALLS):
                name, arg, t_ = CALLS[ci]
                arg = arg[0] if isinstance(arg, tuple) else arg
                dims = ' '.join('%s=%d' % (f, getattr(arg, f)) for f in ('M', 'K', 'N', 'B') if arg is not None and hasattr(arg, f))
                sys.stderr.write('%3d %-28s %-28s %8.1f us\n' % (ci, name, dims, t_ * 1e6))
        if args.gemm_detail:
            for k, (t_, n_, f_, b_) in sorted(detail.items(), key=lambda kv: -kv[1][0]):
                sys.stderr.write('%-52s x%d  %8.1f us  %6.1f TF/s  %6.0f GB/s algorithmic (%.2f of the HBM peak; %.1f us at 6.3 TB/s)\n'
                                 % (k, n_ // args.profile_steps, t_ / n_ * 1e6, f_ / (t_ / n_) / 1e12, b_ / (t_ / n_) / 1e9,
                                    b_ / (t_ / n_) / 1e9 / (HBM_PEAK_TBS * 1e3), b_ / 6.3e6))
    others = None
    default_workload = args.workload == 'A' and args.dtype == 'f32' and (B, N, C) == (32, 1024, 4)
    if rank == 0 and world == 1 and default_workload and not args.no_other_configs:
        others = other_configs(args)
    if rank == 0 and world == 1 and not args.no_cpu_baseline and args.workload == 'A':
        cpu = cpu_baseline(args, batch)

    if rank == 0:
        value = B * world / median_step
        out = {'metric': 'frustums/sec fwd+bwd', 'value': value, 'unit': 'frustums/s', 'n_gpus': world, 'steps': args.steps,
               'warmup': args.warmup, 'ms_per_step': median_step * 1e3,
               'timing': {'value_from': 'median over %d chunks of %d step(s), HIP events on the launch stream, max over ranks (SURVEY 8d)' % (len(per_step), chunk),
                          'ms_per_step_mean': elapsed / args.steps * 1e3, 'value_mean': B * world * args.steps / elapsed,
                          'ms_per_step_min': min(per_step) * 1e3, 'ms_per_step_max': max(per_step) * 1e3},
               'lib_source_hash': lib_hash, 'higher_is_better': True,
               'scaling': 'weak', 'vs_baseline': None, 'dtype': args.dtype, 'data': 'synthetic',
               'config': {'workload': '%s, B=%d N=%d C=%d %s per GPU, dp%d' % (desc, B, N, C, 'fp32' if args.dtype == 'f32' else 'bf16', world),
                          'global_batch': B * world, 'hipgraph': use_graph, 'graph_segments_per_step': trainstep.n_graph_segments(),
                          'gradient_buckets': len(g.buckets) if trainstep.dist else 0,
                          'launches_per_step': (trainstep.n_launches() if pipelined else
                                                sum(len(x) for kind, x in trainstep.cache[True]['prog'] if kind == 'run')
                                                if True in trainstep.cache else len(g.pre) + len(g.fwd) + len(g.bwd) + len(g.opt)),
                          'pipelined': pipelined,
                          # how the fp32 per-point GEMMs multiply (storage, element-wise work, accumulation and statistics are fp32 either way)
                          'gemm_arithmetic': gemm_arithmetic_report(rt, [g.fwd, g.bwd], args.dtype),
                          'schedule': (dict({k: v for k, v in trainstep.schedule_report.items() if k != 'lines'},
                                            rider_barrier_timeouts=rider_timeouts) if trainstep.schedule_report else None),
                          'final_loss': loss},
               'roofline': roofline, 'cpu_baseline': cpu, 'other_configs': others}
        if trainstep.dist:
            out['config']['dp'] = trainstep.dp_report()
    def emit_and_leave(what='the final barrier', code=4):
        """Watchdog exit behind the timed region: the headline line is complete -- print it with a `hang` field saying what did not
        finish, then leave with a NON-ZERO code (a hung collective must not read as a clean run; only the purely informational
        second data-parallel leg leaves with 0)."""
        if rank == 0:
            out['hang'] = '%s did not finish; the headline figures were complete before it' % what
            if saved_stdout is not None:
                os.dup2(saved_stdout, 1)
            os.write(1, (json.dumps(out) + '\n').encode())
        return code

    flat_default = os.environ.get('T3D_DP_FLAT', '1') == '1'
    if (trainstep.dist and not pipelined and os.environ.get('T3D_DP_BOTH_MODES', '1') == '1'):      # (world 1: T3D_FORCE_DIST=1)
        # One driver run compares two forms of the data-parallel step: <= 20 timed steps of a second form on a second step object,
        # reported beside the default's.  RCCL default = the step and its one flat all-reduce captured in ONE graph; second form = the
        # same collective issued by the host between graph segments (the default of rounds 2-4).  gloo (no capture): flat vs bucketed.
        # Under a watchdog that prints the headline line (already complete) and leaves if this extra leg does not finish.
        bucketing = 'flat' if flat_default else 'bucketed'
        if safe_first and not trainstep.one_graph and not args.no_graph:
            name_default, name_alt = bucketing + ', host-issued', bucketing + ', one graph'
            alt_kw = dict(flat_allreduce=flat_default, one_graph=True)
        elif trainstep.one_graph:
            name_default, name_alt = bucketing + ', one graph', bucketing + ', host-issued'
            alt_kw = dict(flat_allreduce=flat_default, one_graph=False)
        else:
            name_default, name_alt = bucketing + ', host-issued', ('bucketed' if flat_default else 'flat') + ', host-issued'
            alt_kw = dict(flat_allreduce=not flat_default, one_graph=False)
        if rank == 0:
            out['config']['dp']['modes'] = {name_default: {'ms_per_step': median_step * 1e3,
                                                           'exposed_allreduce_us': out['config']['dp'].get('exposed_allreduce_us_per_step')}}

        def give_up():
            if rank == 0:
                out['config']['dp']['modes'][name_alt] = 'did not finish'
            return emit_and_leave('the %s data-parallel mode (informational leg)' % name_alt, code=0)

        with Watchdog(int(os.environ.get('T3D_DP_ALT_TIMEOUT_S', '150')), 'the %s data-parallel mode (informational leg)' % name_alt, give_up):
            dist.barrier()
            g2, model2, step2, _ = build_training_step(
                rt, args.workload, B, N, C, world=world, rank=rank, process_group=dist.group.WORLD, force_dist=world == 1,
                use_hip_graph=not args.no_graph, inline_dropout=True, dropout_seed=1234, seed=0, dtype=args.dtype, **alt_kw)
            model2.inputs.load(batch)
            n_alt = min(20, args.steps)
            for _ in range(2 + 5):
                step2.run()
            torch.cuda.synchronize()
            dist.barrier()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(n_alt):
                step2.run()
            torch.cuda.synchronize()
            dist.barrier()
            torch.cuda.synchronize()
            t_alt = torch.tensor([time.perf_counter() - t1], device='cuda', dtype=torch.float64)
            dist.all_reduce(t_alt, op=dist.ReduceOp.MAX)
            step2.time_waits = True
            for _ in range(5):
                step2.run()
            torch.cuda.synchronize()
            rep2 = step2.dp_report()
            if rank == 0:
                alt_ms = float(t_alt.item()) / n_alt * 1e3
                out['config']['dp']['modes'][name_alt] = {'ms_per_step': alt_ms, 'steps': n_alt,
                                                          'exposed_allreduce_us': rep2.get('exposed_allreduce_us_per_step'),
                                                          'mode': rep2.get('mode')}
                # the headline mode is fixed before anything is timed (the default program of step.TrainStep for this world size and
                # backend); the other leg is reported, never promoted -- a best-of-two would be a biased estimator
                out['config']['dp']['modes'][name_default]['ms_per_step_mean'] = out['timing']['ms_per_step_mean']
                out['config']['dp']['headline_mode'] = name_default
    if dist is not None:
        with Watchdog(120, 'the final barrier', emit_and_leave):
            dist.barrier()                        # rank 0 profiled its kernels meanwhile: every rank leaves together
            torch.cuda.synchronize()
            dist.destroy_process_group()
    if saved_stdout is not None:
        import ctypes
        sys.stdout.flush()
        ctypes.CDLL(None).fflush(None)          # whatever C stdio still holds goes to stderr too
        os.dup2(saved_stdout, 1)
        os.close(saved_stdout)
    if rank == 0:
        print(json.dumps(out))
        sys.stdout.flush()


if __name__ == '__main__':
    main()
