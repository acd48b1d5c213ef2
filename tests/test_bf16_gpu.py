"""GPU: BASELINE.json configs[4] -- the bf16 path (bf16 storage of every [M, C] layer tensor, v_mfma_f32_32x32x16_bf16 with fp32
accumulation; statistics, weights, weight gradients, optimiser in fp32) against the fp64 oracle.

Tolerances, stated up front.  bf16 keeps 8 significant bits (relative rounding 2^-9 = 2e-3); every layer rounds its operands (the
activated input and the weights) and its stored output once, and the gradients pass through as many roundings again.  Against the
fp64 oracle the test therefore asks for
  * forward heads (logits, centres, box parameters, pooled features) within 3e-2 * max(1, |ref|_max) absolute, the loss within 2e-2
    relative;
  * every gradient tensor within 1e-1 relative L2 (flip-aware: the oracle differentiates the ReLU / arg-max branches the kernels
    took), the median tensor within 3e-2, all gradients together within 5e-2;
  * moving statistics within 2e-2 * max(1, |ref|_max);
and, independently of any tolerance, that the bf16 and fp32 paths of the SAME library agree with each other better than either
bound (the bf16 error is rounding, not a different function)."""
import numpy as np
import pytest
import torch

from model_check import FWD_KEYS, grad_errors, product_decisions
from oracle import ref_torch as R
from transferable3d_amd.engine import Runtime
from transferable3d_amd.nets import Graph, SemiModelA
from transferable3d_amd.synthetic import make_batch

pytestmark = pytest.mark.gpu

FWD_TOL, LOSS_TOL, GRAD_PER, GRAD_MED, GRAD_GLOB, EMA_TOL = 3e-2, 2e-2, 1e-1, 3e-2, 5e-2, 2e-2


def run(rt, batch, P, c, dtype):
    B, N, C = batch['pc'].shape
    g = Graph(B, N, C, rt=rt, dtype=dtype)
    m = SemiModelA(g, c)
    g.vars.load_state_dict({k: v.detach().cpu().numpy() for k, v in P.items()})
    g.hyper[2] = 0.5
    m.emit_forward(g.fwd, True, True)
    m.emit_backward(g.bwd)
    g.finalize()
    m.inputs.load(batch)
    g.fwd.run()
    g.bwd.run()
    torch.cuda.synchronize()
    return g, m


@pytest.mark.parametrize('B,N,seed', [(4, 256, 1), (8, 512, 2)])
def test_bf16_model_a_step_matches_oracle(hip_lib, B, N, seed):
    C = 4
    batch = make_batch(B, N, C, seed=seed, dropout_scopes={'inst_seg/dp1': ((B, N, 128), 0.5)})
    P = R.init_params(np.random.RandomState(7 + seed), R.layer_table(C, 'A'))
    c = R.default_config()
    rt = Runtime(lib=hip_lib)
    g, m = run(rt, batch, P, c, 'bf16')
    assert m.seg.L2.y.dtype == torch.bfloat16 and m.seg.L9.dz.dtype == torch.bfloat16 and m.seg.L5.y is None
    loss, ep, grads, ema = R.model_a_forward_backward(P, batch, c, forced=product_decisions(m))
    e = m.end_points()
    worst = {}
    for k in FWD_KEYS:
        ref = ep[k].detach().numpy()
        err = float(np.abs(e[k].float().cpu().numpy().reshape(ref.shape) - ref).max() / max(1.0, np.abs(ref).max()))
        worst[k] = err
        assert err < FWD_TOL, (k, err)
    lm, lr = float(e['loss'].cpu()), float(loss)
    assert abs(lm - lr) < LOSS_TOL * abs(lr), (lm, lr)
    per, glob = grad_errors(g, {k: v.numpy() for k, v in grads.items()})
    top = sorted(per.items(), key=lambda kv: -kv[1])[:4]
    med = float(np.median(list(per.values())))
    print('bf16 vs oracle: fwd', worst, 'loss', (lm, lr), 'grad worst', top, 'median', med, 'global', glob,
          'flips', sum(ep['__flips__'].values()))
    assert top[0][1] < GRAD_PER and med < GRAD_MED and glob < GRAD_GLOB, (top, med, glob)
    for k, v in ema.items():
        mine = g.vars.get(k).detach().cpu().numpy()
        assert np.abs(mine - v.detach().numpy()).max() < EMA_TOL * max(1.0, float(v.abs().max())), k
    # the same library in fp32 on the same inputs: the two paths compute the same function
    g32, m32 = run(rt, batch, P, c, 'f32')
    e32 = m32.end_points()
    for k in FWD_KEYS:
        d = float((e[k].float() - e32[k].float()).abs().max() / max(1.0, float(e32[k].abs().max())))
        assert d < FWD_TOL, (k, d)
    num = float((g.vars.grads[:g.vars.used] - g32.vars.grads[:g32.vars.used]).norm())
    assert num < GRAD_GLOB * float(g32.vars.grads[:g32.vars.used].norm())


def test_bf16_config4_problem_size_trains(hip_lib):
    """B = 128, N = 2048 (configs[4]) through the step object bench.py times: 6 steps (5 of them hipGraph replays) with a finite,
    decreasing loss on a fixed batch, a second identical run bit-identical (no atomics), batch-norm'd activations of unit variance."""
    from transferable3d_amd.step import build_training_step
    B, N, C = 128, 2048, 4
    batch = make_batch(B, N, C, seed=77)
    losses = []
    for rep in range(2):
        g, model, step, loss = build_training_step(Runtime(lib=hip_lib), 'A', B, N, C, dtype='bf16', seed=5)
        model.inputs.load(batch)
        cur = []
        for k in range(6):
            step.run()
            cur.append(float(loss))
        torch.cuda.synchronize()
        losses.append(cur)
        if rep == 0:
            L = model.seg.L7
            z = L.y.float() * L.scale + L.shift
            assert float(z.mean(0).abs().max()) < 2e-2 and float((z.var(0, unbiased=False) - 1).abs().max()) < 5e-2
            w1 = g.vars.params[:g.vars.used].clone()
        else:
            assert torch.equal(w1, g.vars.params[:g.vars.used])
    assert losses[0] == losses[1] and all(np.isfinite(losses[0])) and losses[0][-1] < losses[0][0], losses
