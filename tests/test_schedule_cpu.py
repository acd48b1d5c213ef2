"""CPU (NumPy specification library): the step scheduler (transferable3d_amd/schedule.py) -- the alignment keeps each chain's order and
emits every op exactly once; the scheduled step (riders, pairs) equals the unscheduled one bit for bit over several steps; what cannot
ride is refused by t3d_riders_plan."""
import ctypes as C

import numpy as np
import pytest
import torch

from fake_t3d import FakeLib
from transferable3d_amd import abi, nets, schedule
from transferable3d_amd.engine import Runtime
from transferable3d_amd.step import build_training_step
from transferable3d_amd.synthetic import make_batch


class _O:
    def __init__(self, us, host=False, ride=False, pair=False):
        self.us, self.host, self.ride, self.pair = us, host, ride, pair


def _check_alignment(S, T, steps):
    seen = ([], [])
    for st in steps:
        if st[0] == 'solo':
            seen[st[1]].append(st[2])
        elif st[0] == 'host':
            _, ch, p, q0, q1 = st
            assert (S, T)[ch][p].host and all((S, T)[1 - ch][q].ride for q in range(q0, q1)) and q1 > q0
            seen[ch].append(p)
            seen[1 - ch].extend(range(q0, q1))
        else:
            assert S[st[1]].pair and T[st[2]].pair
            seen[0].append(st[1])
            seen[1].append(st[2])
    return seen


def test_alignment_keeps_chain_order_and_hides_small_ops_under_gemms():
    r = np.random.RandomState(0)
    for trial in range(20):
        def chain(n):
            out = []
            for _ in range(n):
                if r.rand() < 0.4:
                    out.append(_O(float(r.uniform(10, 150)), host=r.rand() < 0.8))
                else:
                    out.append(_O(float(r.uniform(3, 12)), ride=r.rand() < 0.8, pair=r.rand() < 0.5))
            return out
        S, T = chain(int(r.randint(1, 14))), chain(int(r.randint(1, 30)))
        steps, total = schedule.align(S, T)
        seen = _check_alignment(S, T, steps)
        # every op once, each chain in its own order (each step advances its chains monotonically)
        assert seen[0] == list(range(len(S))) and seen[1] == list(range(len(T)))
        serial = sum(o.us for o in S + T)
        assert total <= serial + 1e-9
    # a long GEMM of one chain next to a run of small ops of the other: the run rides, whole
    S = [_O(100.0, host=True)]
    T = [_O(5.0, ride=True) for _ in range(4)]
    steps, total = schedule.align(S, T)
    assert steps == [('host', 0, 0, 0, 4)] and total < 100.0 + 20.0
    # without in-launch barriers (max_run = 1) only one of them can
    steps, _ = schedule.align(S, T, max_run=1)
    assert sum(1 for st in steps if st[0] == 'host') == 1 and sum(1 for st in steps if st[0] == 'solo') == 3


@pytest.mark.parametrize('workload,B,N', [('A', 4, 128), ('F', 4, 128)])
def test_scheduled_step_equals_the_unscheduled_step_on_the_specification_library(workload, B, N):
    out = {}
    keep = nets.OVERLAP
    try:
        for on in (False, True):
            nets.OVERLAP = on
            rt = Runtime(device='cpu', lib=FakeLib())
            g, model, step, loss = build_training_step(rt, workload, B, N, 4, seed=3)
            losses = []
            for k in range(3):
                model.inputs.load(make_batch(B, N, 4, seed=70 + k))
                step.run()
                losses.append(float(loss))
            names = [c[0] for kind, x in step.cache[True]['prog'] if kind == 'run' for c in x.calls if c[0].startswith('t3d')]
            out[on] = (losses, g.vars.params[:g.vars.used].clone(), g.vars.state[:g.vars.state_used].clone(), names, step.schedule_report)
    finally:
        nets.OVERLAP = keep
    assert out[False][4] is None and out[True][4]['hosted'] >= (3 if workload == 'A' else 1) and out[True][4]['rider_ops'] >= (3 if workload == 'A' else 1)
    assert any(n.endswith('_r') for n in out[True][3]) and not any(n.endswith('_r') for n in out[False][3])
    assert len(out[True][3]) < len(out[False][3])                      # fewer launches
    assert out[True][0] == out[False][0]
    assert torch.equal(out[True][1], out[False][1]) and torch.equal(out[True][2], out[False][2])


def test_riders_plan_refuses_what_cannot_ride():
    lib = FakeLib()
    rs = abi.RiderSet()
    a = abi.FcFwdArgs()
    a.B, a.K, a.N = 64, 128, 128                                        # B > 32: the 4-row-block form does not ride
    rs.ops[0], rs.n_ops = schedule.small_op('t3d_fc_fwd', a), 1
    assert lib.t3d_riders_plan(C.byref(rs)) == -2
    a.B = 32
    rs.ops[0] = schedule.small_op('t3d_fc_fwd', a)
    assert lib.t3d_riders_plan(C.byref(rs)) == 0 and rs.n_wg >= 1
    rs.ops[0].kind = 9
    assert lib.t3d_riders_plan(C.byref(rs)) == -1
    rs.n_ops = abi.RIDER_MAX_OPS + 1
    assert lib.t3d_riders_plan(C.byref(rs)) == -1


def test_a_timed_out_rider_barrier_stops_the_training_loop():
    """A set's time-out word (t3d.h t3d_rider_set.sync, written by csrc/rider_dev.h when a barrier gives up waiting) must never go
    unnoticed: TrainStep.run looks every `rider_check_every` steps, check_riders() on request (the drivers: before a checkpoint)."""
    rt = Runtime(device='cpu', lib=FakeLib())
    g, model, step, loss = build_training_step(rt, 'A', 4, 128, 4, seed=3)
    step.rider_check_every = 2
    for k in range(4):
        model.inputs.load(make_batch(4, 128, 4, seed=70 + k))
        step.run()
    assert step._sets is not None and step._sets.used >= 3 and step.rider_timeouts() == 0
    step.check_riders()
    step._sets.sets[1][2][-2] = 1                 # what the device writes on a time-out
    assert step.rider_timeouts() == 1
    with pytest.raises(schedule.RiderBarrierTimeout):
        step.check_riders()
    model.inputs.load(make_batch(4, 128, 4, seed=99))
    step.run()                                    # step 5: not a check step
    with pytest.raises(schedule.RiderBarrierTimeout):
        step.run()                                # step 6: is


def test_session_check_riders_reaches_every_compiled_step():
    from transferable3d_amd import api, semisup_v1_sunrgbd as M
    from transferable3d_amd.step import workload_flags
    c = workload_flags('A')
    B, N = 4, 128
    with api.Graph(rt=Runtime(device='cpu', lib=FakeLib()), seed=1).as_default() as gph:
        pls = M.placeholder_inputs(B, N, 4)
        pred, end_points = M.get_semi_model(pls[0], pls[1], pls[2], pls[3], True, use_one_hot=False, c=c)
        loss = M.get_semi_loss(pred, pls[4:], end_points, c=c)
        train_op = api.AdamOptimizer().minimize(loss)
        sess = api.Session()
        batch = make_batch(B, N, 4, seed=5)
        feed = {pl: batch[pl.field] for pl in pls if getattr(pl, 'field', None) in batch}
        sess.run([loss, train_op], feed_dict=feed)
        sess.check_riders()
        impl = sess.steps['train'].impl
        assert impl._sets is not None and impl._sets.used >= 1
        impl._sets.sets[0][2][-2] = 1
        with pytest.raises(schedule.RiderBarrierTimeout):
            sess.check_riders()
