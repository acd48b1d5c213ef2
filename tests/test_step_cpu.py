"""CPU (NumPy specification library): the step object the drivers and bench.py run (transferable3d_amd.step) -- several
consecutive steps against the oracle, and the data-parallel program (bucket markers -> segments + collectives) on a one-rank gloo
group against the single-replica step, bit for bit."""
import socket

import numpy as np
import pytest
import torch

from fake_t3d import FakeLib
from model_check import trajectory_check
from transferable3d_amd.engine import Runtime


@pytest.mark.parametrize('workload,B,N', [('A', 4, 128), ('boxpc', 4, 128), ('F', 4, 128)])
def test_three_steps_of_the_timed_step_follow_the_oracle(workload, B, N):
    rep = trajectory_check(Runtime(device='cpu', lib=FakeLib()), workload, steps=3, B=B, N=N)
    assert [r['step'] for r in rep[:-1]] == [0, 1, 2]
    assert all(r['weight_entries_checked'] > 1000 for r in rep[:-1])


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_bucketed_data_parallel_program_equals_the_single_replica_step():
    """force_dist on a one-rank gloo group: the backward is cut at the three bucket markers, each bucket is all-reduced (a sum over
    one rank) and has its own Adam launch -- the weights after 2 steps equal the unbucketed single-replica step bit for bit, and so
    does ONE flat all-reduce."""
    import torch.distributed as dist
    from transferable3d_amd.step import build_training_step
    from transferable3d_amd.synthetic import make_batch
    dist.init_process_group('gloo', rank=0, world_size=1, init_method='tcp://127.0.0.1:%d' % _free_port())
    try:
        B, N, C = 4, 128, 4
        out = {}
        for mode in ('single', 'bucketed', 'flat'):
            rt = Runtime(device='cpu', lib=FakeLib())
            pg = None if mode == 'single' else dist.group.WORLD
            g, model, step, loss = build_training_step(rt, 'A', B, N, C, process_group=pg, force_dist=mode != 'single',
                                                       flat_allreduce=mode == 'flat', seed=3)
            for k in range(2):
                model.inputs.load(make_batch(B, N, C, seed=70 + k))
                step.run()
            prog = step.cache[True]['prog']
            kinds = [kind for kind, _ in prog]
            if mode == 'single':
                assert kinds == ['run']
            elif mode == 'bucketed':
                assert kinds == ['run', 'allreduce', 'run', 'allreduce', 'run', 'allreduce', 'wait', 'wait', 'wait', 'run']
                assert len(g.buckets) == 3 and sum(n for b in g.buckets for _, n in b) == g.vars.used
                # box + T-Net first (ready before the seg net's backward starts), the seg net's first layers last
                assert g.buckets[0][0][0] == g.vars.offset('tnet/conv-reg1-stage1/weights')
                assert g.buckets[2][0][0] == g.vars.offset('inst_seg/conv1/weights')
            else:
                assert kinds == ['run', 'allreduce', 'wait', 'run']
            out[mode] = (g.vars.params[:g.vars.used].clone(), float(loss))
        assert torch.equal(out['single'][0], out['bucketed'][0]) and torch.equal(out['single'][0], out['flat'][0])
        assert out['single'][1] == out['bucketed'][1] == out['flat'][1]
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('workload', ['A', 'boxpc', 'F'])
def test_pooled_gradient_tensors_leave_the_step_bit_identical(workload, monkeypatch):
    """engine.DZ_POOL (T3D_DZ_POOL=1, an opt-in experiment): the [M, N] gradient tensors between the per-point layers of a net come from a
    small pool of that net -- a buffer is handed on once the launches of the layer that reads it are recorded.  Same launches, same
    arguments apart from where those tensors live: weights, Adam moments and moving statistics after two steps equal the default's bit
    for bit, and the pool really shares buffers."""
    from transferable3d_amd import engine
    from transferable3d_amd.step import build_training_step
    from transferable3d_amd.synthetic import make_batch
    B, N, C = 4, 128, 4
    out = {}
    for pooled in (False, True):
        monkeypatch.setattr(engine, 'DZ_POOL', pooled)
        rt = Runtime(device='cpu', lib=FakeLib())
        g, model, step, loss = build_training_step(rt, workload, B, N, C, seed=5)
        for k in range(2):
            model.inputs.load(make_batch(B, N, C, seed=40 + k))
            step.run()
        out[pooled] = (g.vars.params.clone(), g.vars.adam_m.clone(), g.vars.adam_v.clone(), g.vars.state.clone())
        pools = getattr(g, '_dz_pools', {})
        assert bool(pools) == pooled
        if pooled:
            import gc
            lays = [o for o in gc.get_objects() if isinstance(o, engine.PointLayer) and o.g is g and o.dz is not None]
            assert len({l.dz.data_ptr() for l in lays}) < len(lays)
    for a, b in zip(out[False], out[True]):
        assert torch.equal(a, b)
