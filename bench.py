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

import numpy as np   # noqa: E402
import torch         # noqa: E402

HBM_PEAK_TBS = 8.0                 # MI355X_MICROARCH.md: HBM3E spec peak (about 6.3 TB/s achievable)
MFMA_F32_PEAK_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense fp32 matrix peak
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
    ap.add_argument('--cpu_steps', type=int, default=2)
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
    bytes) of one launch.  Bytes: every operand and result once, fp32: fwd 4(MK + KN [+ MN if y is stored]);
    dgrad 4(2MN + KN + MK out [+ MK prev_y] [+ MK add_in]); wgrad 4(MK + 2MN + slabs); Gram forms with K in place of N."""
    import ctypes
    if name == 't3d_pool_bwd_stage1':
        gl, gf, gb = gemm_work('t3d_pointmlp_gram', a[0])
        q = a[2]
        nch = (q.N + 127) // 128
        by = gb + 4.0 * a[1].M * a[1].K + 4.0 * (2 * q.K * q.N + nch * q.K * q.K)
        return 'k_pool_bwd_stage1<%s>' % gl[gl.index('<') + 1:gl.index(',')], gf + 2.0 * q.K * q.K * q.N, by
    if name == 't3d_pool_bwd_stage2':
        dl, df, db = gemm_work('t3d_pointmlp_dgrad_gram', a[1])
        f = a[0]
        by = db + 4.0 * (2 * f.K * f.N + f.K * f.K + f.B * f.N * f.K)
        return 'k_pool_bwd_stage2<%s>' % dl[dl.index('<') + 1:-1], df + 2.0 * f.K * f.K * f.N, by
    if name == 't3d_pointmlp_bwd':
        d, w = a
        dl, df, db = gemm_work('t3d_pointmlp_dgrad', d)
        wl, wf, wb = gemm_work('t3d_pointmlp_wgrad', w)
        return 'k_pointmlp_bwd<%s,%s>' % (dl[dl.index('<') + 1:-1], wl[wl.index('<') + 1:-1]), df + wf, db + wb
    if name == 't3d_pointmlp_dgrad_gram':
        by = 4.0 * a.M * a.K * (2 + (0 if _null(a.add_in) else 1) + (0 if _null(a.prev_y) else 1)) + 4.0 * a.K * a.K
        return 'k_pointmlp_dgrad_gram<%d>' % (128 if a.K % 128 == 0 and (a.M // 128) * (a.K // 128) >= 512 else 64), 2.0 * a.M * a.K * a.K, by
    if name == 't3d_pointmlp_gram':
        rps, tk, tn = ctypes.c_int(0), ctypes.c_int(0), ctypes.c_int(0)
        LIB.t3d_wgrad_plan(a.M, a.K, a.K, ctypes.byref(rps), ctypes.byref(tk), ctypes.byref(tn))
        t = tk.value if (rps.value == a.rows_per_split and tk.value == tn.value) else 64
        return 'k_pointmlp_gram<%d,%d>' % (t, t), 2.0 * a.M * a.K * a.K, 4.0 * a.M * a.K + 4.0 * (a.M // a.rows_per_split) * a.K * a.K
    flops = 2.0 * a.M * a.K * a.N
    if name == 't3d_pointmlp_fwd':
        by = 4.0 * (a.M * a.K + a.K * a.N + (0 if _null(a.y) else a.M * a.N))
        if os.environ.get('T3D_FWD_POOL', '1') != '0' and _null(a.y) and not _null(a.pmax) and _null(a.a.sub) and a.K == 128 and \
                a.N % 128 == 0 and a.N >= 256:
            return 'k_pointmlp_fwd_pool<128,32,8>', flops, by
        return 'k_pointmlp_fwd<%d>' % (128 if a.N % 128 == 0 and (a.M // 128) * (a.N // 128) >= 512 else 64), flops, by
    if name == 't3d_pointmlp_dgrad':
        by = 4.0 * (2 * a.M * a.N + a.K * a.N + a.M * a.K * (1 + (0 if _null(a.prev_y) else 1) + (0 if _null(a.add_in) else 1)))
        return 'k_pointmlp_dgrad<%d>' % (128 if a.K % 128 == 0 and (a.M // 128) * (a.K // 128) >= 512 else 64), flops, by
    rps, tk, tn = ctypes.c_int(0), ctypes.c_int(0), ctypes.c_int(0)
    LIB.t3d_wgrad_plan(a.M, a.K, a.N, ctypes.byref(rps), ctypes.byref(tk), ctypes.byref(tn))
    if rps.value != a.rows_per_split:
        tk.value, tn.value = (128 if a.K > 64 else 64), (128 if a.N % 128 == 0 else 64)
    by = 4.0 * (a.M * a.K + 2 * a.M * a.N + (a.M // a.rows_per_split) * a.K * a.N)
    return 'k_pointmlp_wgrad<%d,%d>' % (tk.value, tn.value), flops, by


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
        torch.cuda.synchronize()
        for ci, (name, arg, e0, e1) in enumerate(evs):
            dt = e0.elapsed_time(e1) * 1e-3 / repeat
            CALLS.setdefault(ci, [name, arg, 0.0])[2] += dt / steps
            label, flops, nbytes = (gemm_work(name, arg) if name.startswith(('t3d_pointmlp', 't3d_pool_bwd_stage')) and
                                    name != 't3d_pointmlp_dgrad_narrow' else (name, 0.0, 0.0))
            d = acc.setdefault(label, [0.0, 0, 0.0, 0.0])
            d[0] += dt
            d[1] += 1
            d[2] += flops
            d[3] += nbytes
            if flops:
                a0 = (arg[1] if name == 't3d_pool_bwd_stage2' else arg[0]) if isinstance(arg, tuple) else arg
                dd = detail.setdefault('%s M%d K%d N%d' % (label, a0.M, a0.K, getattr(a0, 'N', a0.K)), [0.0, 0, flops])
                dd[0] += dt
                dd[1] += 1
    return acc, detail


def pmc_traffic(label):
    """HBM bytes per launch of `label` from the committed PMC summary (tools/pmc_traffic.py; counters cannot be read from inside
    the process being timed).  None when the summary is absent or was taken at another problem size."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'profiles', 'pmc_traffic.json')
    try:
        with open(path) as fh:
            return json.load(fh)['kernels'][label]['bytes_per_launch']
    except (OSError, KeyError, ValueError):
        return None


def cpu_baseline(args, batch):
    """The oracle's torch-CPU fp32 restatement of the identical step (fwd + bwd + TF-form Adam), all host
    threads.  TF1 itself cannot run, so kind = "port"."""
    from oracle import ref_torch as R
    torch.set_num_threads(min(os.cpu_count(), args.cpu_threads))
    C = args.num_channel
    P = R.init_params(np.random.RandomState(0), R.layer_table(C, 'A'), dtype=torch.float32)
    c = R.default_config()
    names = R.trainable_names(P)
    m = {k: torch.zeros_like(P[k]) for k in names}
    v = {k: torch.zeros_like(P[k]) for k in names}
    batch = dict(batch)
    batch['dropout_masks'] = {'inst_seg/dp1': (np.random.RandomState(1).uniform(size=(args.batch_size, args.num_point, 128)) < 0.5)
                              .astype(np.float32)}
    times = []
    for it in range(args.cpu_steps + 1):
        t0 = time.perf_counter()
        _, _, grads, ema = R.model_a_forward_backward(P, batch, c, dtype=torch.float32)
        R.adam_tf_step(P, grads, m, v, it + 1, 1e-3)
        for k, val in ema.items():
            P[k] = val.detach()
        times.append(time.perf_counter() - t0)
    t = float(np.median(times[1:]))
    return {'value': args.batch_size / t, 'unit': 'frustums/s', 'cores': torch.get_num_threads(), 'kind': 'port',
            'sample': '%d timed steps (after 1 warm-up) of the same B=%d N=%d C=%d fp32 fwd+bwd+Adam step, torch-CPU '
                      'restatement of the reference graph (TF1 not installable)' % (args.cpu_steps, args.batch_size,
                                                                                     args.num_point, C)}


def main():
    args = parse()
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
        if backend == 'nccl':
            dist.init_process_group('nccl', rank=rank, world_size=world, device_id=torch.device('cuda', local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    from transferable3d_amd.config import make_parser
    from transferable3d_amd.engine import Runtime, Plan
    from transferable3d_amd.nets import BoxPCModel, Graph, SemiModelA, SemiModelF, make_schedule
    from transferable3d_amd.synthetic import make_batch

    B, N, C = args.batch_size, args.num_point, args.num_channel
    rt = Runtime()
    global LIB
    LIB = rt.lib
    g = Graph(B, N, C, rt=rt, seed=0)              # identical initial weights on every rank
    g.inline_dropout, g.dropout_seed = True, 1234 + rank      # the seg head draws its dropout mask in its own kernel
    prefixes = None
    if args.workload == 'A':
        c = make_parser().parse_special_args(['--SEMI_MODEL', 'A', '--WEAK_WEIGHT_REPROJECTION', '0', '--WEAK_WEIGHT_SURFACE', '0'])
        model = SemiModelA(g, c)
        loss_buf = lambda: model.loss_op.loss
        desc = 'seg-PointNet + T-Net + box-est fwd+bwd+Adam (SEMI_MODEL A)'
    elif args.workload == 'boxpc':
        c = make_parser().parse_special_args(['--BOX_PC_MASK_REPRESENTATION', 'A', '--BOXPC_WEIGHT_DELTA', '4'])
        model = BoxPCModel(g, c, False)
        loss_buf = lambda: model.loss_op.loss
        desc = 'Box-PC Fit net fwd+bwd+Adam (train_boxpc.py path)'
    else:
        c = make_parser().parse_special_args(
            ['--SEMI_MODEL', 'F', '--BOX_PC_MASK_REPRESENTATION', 'A', '--WEAK_WEIGHT_INTRACLASSVAR', '2', '--WEAK_WEIGHT_REPROJECTION', '0',
             '--SEMI_MULTIPLIER_FOR_WEAK_LOSS', '0.05', '--SEMI_BOXPC_FIT_ONLY_ON_2D_CLS', '1', '--SEMI_WEIGHT_BOXPC_FIT_LOSS', '1'])
        model = SemiModelF(g, c, use_one_hot=True, train_classes=[i in (1, 2, 6, 7, 8) for i in range(10)])
        loss_buf = lambda: model.loss
        prefixes = list(SemiModelF.VAR_LIST)
        desc = 'SEMI_MODEL F stage c (frozen seg + Box-PC branch, var_list optimiser) fwd+bwd+Adam'
    sched = make_schedule(B * world)
    g.emit_schedule(g.pre, sched)
    g.emit_dropout_masks(g.pre, seed=1234 + rank)
    model.emit_forward(g.fwd, True, True)
    model.emit_backward(g.bwd)
    g.emit_adam(g.opt, prefixes=prefixes, grad_scale=1.0 / world)
    g.finalize()
    batch = make_batch(B, N, C, seed=1234 + rank, boxpc=args.workload == 'boxpc')  # per-rank shard (weak scaling)
    if args.workload == 'F':
        batch['is_data_2D'][::2] = 1
    model.inputs.load(batch)
    torch.cuda.synchronize()

    nparam = g.vars.used
    flat_grads = g.vars.grads[:nparam]

    def run_compute():
        g.pre.run()
        g.fwd.run()
        g.bwd.run()

    use_graph = not args.no_graph
    if use_graph:
        # warm the kernels once eagerly, then capture the step into hipGraphs
        run_compute()
        if use_dist:
            dist.all_reduce(flat_grads)        # communicator set-up happens here, outside any capture
        g.opt.run()
        torch.cuda.synchronize()
        s = torch.cuda.Stream()
        one_graph = False
        if use_dist and os.environ.get('T3D_DP_ONE_GRAPH', '0') == '1':
            # the gradient all-reduce captured between backward and Adam: one replay per step (RCCL supports stream capture)
            try:
                g1 = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g1, stream=s, capture_error_mode='thread_local'):
                    run_compute()
                    dist.all_reduce(flat_grads)
                    g.opt.run()
                one_graph = True
            except RuntimeError as err:
                sys.stderr.write('capture with the all-reduce inside failed (%s); two graphs around an eager all-reduce\n' % err)
                torch.cuda.synchronize()
        try:
            if one_graph:
                raise StopIteration
            g1 = torch.cuda.CUDAGraph()
            # thread_local: calls of other threads (the RCCL watchdog) during capture must not invalidate it
            with torch.cuda.graph(g1, stream=s, capture_error_mode='thread_local'):
                run_compute()
                if not use_dist:
                    g.opt.run()
            g2 = None
            if use_dist:
                g2 = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g2, stream=s, capture_error_mode='thread_local'):
                    g.opt.run()
        except StopIteration:
            pass
        except RuntimeError as err:          # a capture that the runtime refuses must not cost the measurement: eager launches
            sys.stderr.write('hipGraph capture failed (%s); falling back to eager launches\n' % err)
            torch.cuda.synchronize()
            use_graph = False

    def step():
        if use_graph:
            g1.replay()
            if use_dist and not one_graph:
                dist.all_reduce(flat_grads)
                g2.replay()
        else:
            run_compute()
            if use_dist:
                dist.all_reduce(flat_grads)
            g.opt.run()

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], device='cuda', dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    loss = float(loss_buf().item())
    assert np.isfinite(loss), 'non-finite loss'

    roofline = None
    cpu = None
    if rank == 0 and args.profile_steps > 0:
        # per-kernel timing for the roofline object (eager, per-launch events on the launch stream)
        acc, detail = profile_kernels([g.pre, g.fwd, g.bwd, g.opt], args.profile_steps)
        total = sum(v[0] for v in acc.values())
        dom = max((k for k in acc if k.startswith(('k_pointmlp', 'k_pool_bwd_stage'))), key=lambda k: acc[k][0])
        tsec, n, fl, nby = acc[dom]
        # the roofline that bounds the dominant kernel: the larger of (algorithmic FLOPs / MFMA peak) and (algorithmic bytes / HBM peak)
        mfma_bound = fl / (MFMA_F32_PEAK_TFLOPS * 1e12) >= nby / (HBM_PEAK_TBS * 1e12)
        achieved = fl / tsec / 1e12
        hbm_achieved = nby / tsec / 1e9
        gemm_t = sum(v[0] for k, v in acc.items() if k.startswith(('k_pointmlp', 'k_pool_bwd_stage')))
        gemm_f = sum(v[2] for k, v in acc.items() if k.startswith(('k_pointmlp', 'k_pool_bwd_stage')))
        roofline = {'bound': 'mfma' if mfma_bound else 'hbm', 'kernel': dom,
                    'achieved': achieved if mfma_bound else hbm_achieved,
                    'peak': MFMA_F32_PEAK_TFLOPS if mfma_bound else HBM_PEAK_TBS * 1e3,
                    'unit': 'TFLOP/s' if mfma_bound else 'GB/s',
                    'frac': achieved / MFMA_F32_PEAK_TFLOPS if mfma_bound else hbm_achieved / (HBM_PEAK_TBS * 1e3),
                    'mfma': {'achieved_tflops': achieved, 'frac': achieved / MFMA_F32_PEAK_TFLOPS},
                    'hbm': {'achieved_gbs': hbm_achieved, 'frac': hbm_achieved / (HBM_PEAK_TBS * 1e3),
                            'algorithmic_bytes_per_launch': nby / n},
                    'traffic': pmc_traffic(dom),
                    'traffic_unit': 'HBM bytes per launch, rocprofv3 PMC passes of this workload (profiles/pmc_traffic.json)',
                    'avg_launch_us': tsec / n * 1e6, 'launches_per_step': n // args.profile_steps,
                    'flops_per_launch': fl / n,
                    'all_gemm_kernels': {'achieved': gemm_f / gemm_t / 1e12, 'frac': gemm_f / gemm_t / 1e12 / MFMA_F32_PEAK_TFLOPS,
                                         'share_of_step_kernel_time': gemm_t / total},
                    'whole_step': None if args.workload != 'A' else {'gflop_per_frustum_split': SPLIT_GFLOP_PER_FRUSTUM,
                                   # what the GEMM kernels actually execute (the Gram-form backward of the pooled layers needs
                                   # fewer FLOPs than the split count the roofline figure is quoted on)
                                   'gflop_per_frustum_executed': gemm_f / args.profile_steps / B / 1e9,
                                   'achieved_executed': gemm_f / args.profile_steps * args.steps / elapsed / 1e12,
                                   'achieved': SPLIT_GFLOP_PER_FRUSTUM * B * args.steps / elapsed / 1e3,
                                   'frac': SPLIT_GFLOP_PER_FRUSTUM * B * args.steps / elapsed / 1e3 / MFMA_F32_PEAK_TFLOPS},
                    'per_kernel_us_per_step': {k: v[0] / args.profile_steps * 1e6 for k, v in sorted(acc.items(), key=lambda kv: -kv[1][0])}}
        if args.call_detail:
            for ci in sorted(CALLS):
                name, arg, t_ = CALLS[ci]
                arg = arg[0] if isinstance(arg, tuple) else arg
                dims = ' '.join('%s=%d' % (f, getattr(arg, f)) for f in ('M', 'K', 'N', 'B') if arg is not None and hasattr(arg, f))
                sys.stderr.write('%3d %-28s %-28s %8.1f us\n' % (ci, name, dims, t_ * 1e6))
        if args.gemm_detail:
            for k, (t_, n_, f_) in sorted(detail.items(), key=lambda kv: -kv[1][0]):
                sys.stderr.write('%-44s x%d  %8.1f us  %6.1f TF/s\n' % (k, n_ // args.profile_steps, t_ / n_ * 1e6, f_ / (t_ / n_) / 1e12))
    if rank == 0 and world == 1 and not args.no_cpu_baseline and args.workload == 'A':
        cpu = cpu_baseline(args, batch)

    if rank == 0:
        value = B * world * args.steps / elapsed
        out = {'metric': 'frustums/sec fwd+bwd', 'value': value, 'unit': 'frustums/s', 'n_gpus': world, 'steps': args.steps,
               'warmup': args.warmup, 'ms_per_step': elapsed / args.steps * 1e3, 'higher_is_better': True,
               'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
               'config': {'workload': '%s, B=%d N=%d C=%d fp32 per GPU, dp%d' % (desc, B, N, C, world), 'global_batch': B * world, 'hipgraph': use_graph,
                          'launches_per_step': len(g.pre) + len(g.fwd) + len(g.bwd) + len(g.opt), 'final_loss': loss},
               'roofline': roofline, 'cpu_baseline': cpu}
    if dist is not None:
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
