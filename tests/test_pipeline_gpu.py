"""GPU: the software-pipelined step (transferable3d_amd.step.PipelinedStep) on the HIP path, hipGraph replay included, equals the
one-step-at-a-time program bit for bit under BOTH GEMM arithmetics of the fp32 path -- at the headline size, where the rider
schedule and the plan rules are the ones bench.py times (T3D_PIPELINE=1 / `other_configs`)."""
import pytest
import torch

from transferable3d_amd.engine import Runtime
from transferable3d_amd.step import build_pipelined_step, build_training_step
from transferable3d_amd.synthetic import make_batch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('B,N', [(8, 256), (32, 1024)])
def test_pipelined_step_equals_the_one_step_program_bit_for_bit(hip_lib, gemm_arithmetic, B, N):
    C, K = 4, 6                                           # step 0 eager, step 1 captures, steps 2.. replay (both programs)
    batches = [make_batch(B, N, C, seed=170 + k) for k in range(K)]
    g, model, step, loss = build_training_step(Runtime(lib=hip_lib), 'A', B, N, C, seed=3, use_hip_graph=True)
    seq = []
    for k in range(K):
        model.inputs.load(batches[k])
        step.run()
        seq.append(float(loss))
    torch.cuda.synchronize()
    vs0 = g.vars
    ps, ctxs = build_pipelined_step(Runtime(lib=hip_lib), B, N, C, seed=3, use_hip_graph=True)
    vs1 = ctxs[0]['g'].vars
    got = []
    ps.inputs(0).load(batches[0])
    for k in range(K):
        if k + 1 < K:
            ps.inputs(k + 1).load(batches[k + 1])
        ps.run(last=(k == K - 1))
        got.append(float(ps.loss(k)))
    torch.cuda.synchronize()
    assert got == seq, (got, seq)
    for a, b in ((vs0.params, vs1.params), (vs0.adam_m, vs1.adam_m), (vs0.adam_v, vs1.adam_v)):
        assert torch.equal(a[:vs0.used], b[:vs0.used])
    assert torch.equal(vs0.state[:vs0.state_used], vs1.state[:vs0.state_used])
    assert ps.rider_timeouts() == 0 and step.rider_timeouts() == 0
    assert ps.n_graph_segments() == 1
