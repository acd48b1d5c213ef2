"""GPU: the step object bench.py times (transferable3d_amd.step.build_training_step -> TrainStep: device schedules, in-kernel
dropout, forward, backward, TF-form Adam, hipGraph replay) against an oracle trajectory, step by step; the data-parallel program on
a one-rank RCCL group against the single-replica step, bit for bit."""
import os
import socket

import numpy as np
import pytest
import torch

from model_check import trajectory_check
from transferable3d_amd.engine import Runtime

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('workload', ['A', 'boxpc', 'F'])
def test_five_replayed_steps_follow_the_oracle(hip_lib, workload):
    """Step 0 runs eagerly, step 1 captures the hipGraph, steps 1-4 are replays of it: loss, heads, every gradient tensor, moving
    statistics (EMA chaining with the device-side bn_decay), Adam moments and post-step weights (device-side lr, step counter,
    var_list of stage c) of EVERY step against the oracle."""
    rep = trajectory_check(Runtime(lib=hip_lib), workload, steps=5, B=8, N=256, use_hip_graph=True, verbose=True)
    assert rep[-1]['graph_segments'] == 1
    assert all(r['weight_entries_checked'] > 1000 for r in rep[:-1])


@pytest.mark.parametrize('workload', ['A', 'boxpc', 'F'])
def test_three_replayed_steps_at_the_headline_size_follow_the_oracle(hip_lib, gemm_arithmetic, workload):
    """BASELINE configs[1], [2], [3] at their own size: B=32, N=1024, C=4 (the plans differ from the small cases': tile widths,
    weight-gradient splits, which launches host riders), under both GEMM arithmetics of the fp32 path (the default three-term bf16
    form and the fp32-MFMA kernels)."""
    rep = trajectory_check(Runtime(lib=hip_lib), workload, steps=3, B=32, N=1024, use_hip_graph=True, verbose=True)
    assert rep[-1]['graph_segments'] == 1


def test_three_replayed_steps_at_the_reference_problem_size_follow_the_oracle(hip_lib, gemm_arithmetic):
    """The reference's own defaults (train_semisup.py:34-36,61: --num_point 2048, batch 32, RGB channels on: C = 6): M = 65536 rows --
    six-channel first layers (K = 6: neither the register kernels of K <= 4 nor a whole x3 k-tile) and the plan rules that switch at
    65536 rows (the one-pass fp32 backward of the <= 128-channel layers, 128-wide tiles where they give >= 512 workgroups)."""
    rep = trajectory_check(Runtime(lib=hip_lib), 'A', steps=3, B=32, N=2048, C=6, use_hip_graph=True, verbose=True)
    assert rep[-1]['graph_segments'] == 1


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_data_parallel_step_on_one_rank_rccl_equals_the_single_replica_step(hip_lib):
    """The multi-rank program (three gradient buckets all-reduced on RCCL beside the rest of the backward, one hipGraph segment
    between two collectives, the buckets' Adam launches in the last one) with ONE rank must reproduce the single-replica step bit for bit -- weights after 4
    steps (3 of them graph replays) and the loss; so must the flat variant (one all-reduce between backward and Adam)."""
    import torch.distributed as dist
    from transferable3d_amd.step import build_training_step
    from transferable3d_amd.synthetic import make_batch
    torch.cuda.set_device(0)
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    dist.init_process_group('nccl', rank=0, world_size=1, init_method='tcp://127.0.0.1:%d' % _free_port(),
                            device_id=torch.device('cuda', 0))
    try:
        B, N, C = 8, 256, 4
        out = {}
        for mode in ('single', 'bucketed', 'flat', 'one_graph_inline', 'one_graph_overlapped'):
            pg = None if mode == 'single' else dist.group.WORLD
            os.environ['T3D_DP_ONE_GRAPH'] = {'one_graph_inline': '1', 'one_graph_overlapped': '2'}.get(mode, '0')
            g, model, step, loss = build_training_step(Runtime(lib=hip_lib), 'A', B, N, C, process_group=pg, force_dist=mode != 'single',
                                                       flat_allreduce=mode == 'flat', seed=3, use_hip_graph=True)
            for k in range(4):
                model.inputs.load(make_batch(B, N, C, seed=70 + k))
                step.run()
            torch.cuda.synchronize()
            out[mode] = (g.vars.params[:g.vars.used].clone(), float(loss), step.n_graph_segments(),
                         sum(1 for kind, _ in step.cache[True]['prog'] if kind == 'run'))
        # program: one run of launches for a single replica; four around three collectives (the buckets' Adam launches share the
        # last); two around one.  Graphs: the FIRST run only (the later ones are launched kernel by kernel: a graph launch costs more
        # on the GPU than the host saves there), or ONE with the collectives captured inside (in line, or on RCCL's stream as a branch
        # of the graph beside the rest of the backward)
        modes = ('single', 'bucketed', 'flat', 'one_graph_inline', 'one_graph_overlapped')
        assert [out[m][3] for m in modes] == [1, 4, 2, 4, 4]
        assert [out[m][2] for m in modes] == [1, 1, 1, 1, 1]
        for m in out:
            assert torch.equal(out['single'][0], out[m][0]) and out['single'][1] == out[m][1], m
    finally:
        os.environ.pop('T3D_DP_ONE_GRAPH', None)
        dist.destroy_process_group()


def test_default_rccl_step_is_one_graph_and_equals_the_host_issued_and_the_single_replica_steps(hip_lib):
    """On RCCL the data-parallel step is ONE captured graph with the (flat) gradient all-reduce inside it -- the default, no
    environment variable.  For the three workloads, at one rank: weights, Adam moments, moving statistics and the loss after 5 steps
    (3 of them replays of that graph) are bit-identical to the host-issued program (collective between two graph segments: the
    default of rounds 2-4) and to the single-replica step."""
    import torch.distributed as dist
    from transferable3d_amd.step import build_training_step
    from transferable3d_amd.synthetic import make_batch
    torch.cuda.set_device(0)
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    assert 'T3D_DP_ONE_GRAPH' not in os.environ
    dist.init_process_group('nccl', rank=0, world_size=1, init_method='tcp://127.0.0.1:%d' % _free_port(),
                            device_id=torch.device('cuda', 0))
    try:
        B, N, C = 8, 256, 4
        for workload in ('A', 'boxpc', 'F'):
            out = {}
            for mode in ('single', 'default', 'host_issued'):
                pg = None if mode == 'single' else dist.group.WORLD
                g, model, step, loss = build_training_step(Runtime(lib=hip_lib), workload, B, N, C, process_group=pg, force_dist=mode != 'single',
                                                           flat_allreduce=True, seed=5, use_hip_graph=True,
                                                           one_graph=False if mode == 'host_issued' else None)
                for k in range(5):
                    b = make_batch(B, N, C, seed=90 + k, boxpc=workload == 'boxpc')
                    if workload == 'F':
                        b['is_data_2D'][::2] = 1
                    model.inputs.load(b)
                    step.run()
                torch.cuda.synchronize()
                v = g.vars
                out[mode] = (v.params[:v.used].clone(), v.adam_m[:v.used].clone(), v.adam_v[:v.used].clone(), v.state[:v.state_used].clone(),
                             float(loss), step.one_graph, [kind for kind, _ in step.cache[True]['cprog']], step.dp_report()['mode'])
            assert out['default'][5] and not out['host_issued'][5] and not out['single'][5]
            assert out['default'][6] == ['run_graph'], out['default'][6]                       # the whole step: one replayed graph
            assert out['host_issued'][6] == ['run', 'allreduce', 'wait', 'run'], out['host_issued'][6]
            assert 'ONE graph' in out['default'][7] and 'host-issued' in out['host_issued'][7]
            for m in ('default', 'host_issued'):
                for i in range(4):
                    assert torch.equal(out['single'][i], out[m][i]), (workload, m, i)
                assert out['single'][4] == out[m][4], (workload, m)
    finally:
        dist.destroy_process_group()


def test_paired_small_launches_are_bit_identical(hip_lib):
    """nets.pair_small_launches: the box / T-Net backward and the seg-net backward interleaved so that their small launches share
    launches (t3d_small_pair): 11 launches fewer per step, weights after four hipGraph-replayed steps bit-identical to the unpaired
    plan."""
    from transferable3d_amd import nets
    from transferable3d_amd.engine import Runtime
    from transferable3d_amd.step import build_training_step
    from transferable3d_amd.synthetic import make_batch
    B, N, C = 32, 1024, 4
    batch = make_batch(B, N, C, seed=21)
    res = {}
    keep, keep_ov = nets.PAIR_SMALL, nets.OVERLAP
    nets.OVERLAP = False            # (the step scheduler supersedes the pairing of the backward plan: tests/test_riders_gpu.py)
    try:
        for on in (False, True):
            nets.PAIR_SMALL = on
            g, model, step, loss = build_training_step(Runtime(lib=hip_lib), 'A', B, N, C, seed=4)
            model.inputs.load(batch)
            losses = []
            for _ in range(4):
                step.run()
                losses.append(float(loss))
            torch.cuda.synchronize()
            names = [n for n, _, _ in g.bwd.calls if n.startswith('t3d')]
            res[on] = (losses, g.vars.params[:g.vars.used].clone(), len(names), names.count('t3d_small_pair'))
    finally:
        nets.PAIR_SMALL, nets.OVERLAP = keep, keep_ov
    assert res[True][3] >= 8 and res[False][3] == 0 and res[True][2] == res[False][2] - res[True][3]
    assert res[True][0] == res[False][0] and torch.equal(res[True][1], res[False][1])
