"""CPU (NumPy specification library): the software-pipelined step (transferable3d_amd.step.PipelinedStep -- the seg forward of step
k+1 runs beside the T-Net / box chain of step k, two contexts over one variable store) equals the one-step-at-a-time program bit for
bit: losses of every step, weights, moving statistics, Adam moments."""
import torch

from fake_t3d import FakeLib
from transferable3d_amd.engine import Runtime
from transferable3d_amd.step import build_pipelined_step, build_training_step
from transferable3d_amd.synthetic import make_batch


def test_pipelined_step_equals_the_sequential_program():
    B, N, C, K = 4, 128, 4, 5
    batches = [make_batch(B, N, C, seed=70 + k) for k in range(K)]
    g, model, step, loss = build_training_step(Runtime(device='cpu', lib=FakeLib()), 'A', B, N, C, seed=3)
    seq = []
    for k in range(K):
        model.inputs.load(batches[k])
        step.run()
        seq.append(float(loss))
    vs0 = g.vars
    ps, ctxs = build_pipelined_step(Runtime(device='cpu', lib=FakeLib()), B, N, C, seed=3)
    vs1 = ctxs[0]['g'].vars
    assert ctxs[1]['g'].vars is vs1 and vs1.used == vs0.used                   # one variable store, same variables
    got = []
    ps.inputs(0).load(batches[0])
    for k in range(K):
        if k + 1 < K:
            ps.inputs(k + 1).load(batches[k + 1])       # the next step's inputs are in place before this step runs
        ps.run(last=(k == K - 1))
        got.append(float(ps.loss(k)))
    assert got == seq, (got, seq)
    for a, b in ((vs0.params, vs1.params), (vs0.adam_m, vs1.adam_m), (vs0.adam_v, vs1.adam_v)):
        assert torch.equal(a[:vs0.used], b[:vs0.used])
    assert torch.equal(vs0.state[:vs0.state_used], vs1.state[:vs0.state_used])     # moving statistics too (run(last=True) starts no forward)
    # the step counters of the two contexts interleave: context 0 ran steps 0, 2, 4; context 1 steps 1, 3
    assert float(ctxs[0]['g'].hyper[0]) == 5.0 and float(ctxs[1]['g'].hyper[0]) == 4.0
    rep = ps.schedule_report
    assert rep['hosted'] >= 5 and ps.rider_timeouts() == 0
