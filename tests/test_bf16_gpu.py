"""GPU: BASELINE.json configs[4] -- the bf16 path (bf16 storage of every [M, C] layer tensor, v_mfma_f32_32x32x16_bf16 with fp32
accumulation; statistics, weights, weight gradients, optimiser in fp32).

What "parity" means here, stated up front.  bf16 keeps 8 significant bits (relative rounding 2^-9 = 2e-3).  Batch-norm re-centres
every layer's pre-activations on zero, so a fraction of the ReLU gates flips under that noise and the perturbation grows by about
1.5x per layer (tools/bf16_error_growth.py: 3e-3 after conv1, 7e-2 after conv9; the heads' batch-norm over B = 4-8 rows amplifies
it further).  Against the plain fp64 oracle a bf16 run is therefore ~10 % off in the logits at the test sizes -- that is what bf16
training is, not a kernel property, and it would hide real bugs.  The oracle therefore EMULATES the path
(oracle.ref_torch.Ctx.bf16): fp64 arithmetic with the bf16 roundings injected exactly where the kernels round -- both operands of
every per-point GEMM and every stored raw layer output; pooled layers, the conv10 logits, the fully-connected heads, statistics
and losses unrounded -- and differentiates the ReLU / arg-max / mask branches the kernels took.  What is left between the two is
the fp32 accumulation order (an element in ~4000 lands on the other side of a bf16 rounding boundary and differs by one spacing)
and the backward pass's own roundings (dz and dy to bf16: unbiased, averaged out by the sums over rows).  Bounds:
Once a few elements differ, the layers behind them see inputs that differ by ~1e-3 and their own roundings decorrelate, so deep in
the net the emulation can only be as close as a fraction of the bf16 noise itself (measured: logits 1-4e-2 against 6e-1 to the
plain fp64 oracle).  Hence two kinds of bounds:
  * LAYER-WISE, tight: the stored bf16 outputs of the first two layers of the seg net (inputs identical up to batch-norm
    statistics): at most 2 % of the elements differ, none by more than four bf16 spacings (plus one operand element rounded the other
    way); the T3D_BF16 GEMM kernels one by one
    against torch matmuls of the same rounded operands: tests/test_kernels_bf16_gpu.py (one rounding of the output, nothing else);
  * END TO END: forward heads within 6e-2 * max(1, |ref|_max), loss within 1e-2 relative, moving statistics within 1e-2 *
    max(1, |ref|_max); every gradient tensor that carries at least 1 % of the gradient norm within 5e-2 relative L2, the median
    tensor within 2.5e-2, all gradients together within 3.5e-2; EVERY tensor within 2.5e-1 or absolutely negligible; a four-step
    trajectory (every tensor within 4e-2 there, Adam on the master weights, the bf16 copy).
The distance to the un-emulated fp64 oracle is printed, not asserted."""
import numpy as np
import pytest
import torch

from model_check import FWD_KEYS, grad_errors, product_decisions
from oracle import ref_torch as R
from transferable3d_amd.engine import Runtime
from transferable3d_amd.nets import Graph, SemiModelA
from transferable3d_amd.synthetic import make_batch

pytestmark = pytest.mark.gpu

FWD_TOL, LOSS_TOL, GRAD_PER, GRAD_MED, GRAD_GLOB, EMA_TOL = 6e-2, 1e-2, 5e-2, 2.5e-2, 3.5e-2, 1e-2      # (round 2: 1.2e-1, 4e-2, 5e-2; measured 3.6e-2, 1.2e-2, 2.3e-2)


def check_bf16_decisions(margins, what):
    """model_check.check_decision_margins with the bf16 bounds: at most 2 % of a site's decisions taken over from the kernels; their
    margins by the oracle's own numbers within FOUR bf16 spacings of the tensor's magnitude in the layers in front of the pooled global
    feature (seg conv1-5: inputs identical up to batch-norm statistics; measured <= 0.2), within SIXTEEN behind it (conv6 adds a
    1024-term per-frustum product of pooled features that each carry bf16-level noise, and batch-norm re-amplifies it; everything
    behind the mask inherits that: measured up to 6.9 on fresh weights, 11.3 after three Adam steps)."""
    from model_check import check_decision_margins
    front = lambda site: site.startswith('inst_seg/conv') and site.split('#')[0][-1] in '12345' and 'conv10' not in site
    a = {k: v for k, v in margins.items() if front(k)}
    b = {k: v for k, v in margins.items() if not front(k)}
    out = dict(check_decision_margins(a, flip_frac=0.02, flip_floor=8, margin_tol=4 * 2.0 ** -8, what=what))
    out.update(check_decision_margins(b, flip_frac=0.02, flip_floor=8, margin_tol=16 * 2.0 ** -8, what=what))
    return out


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


@pytest.mark.parametrize('B,N,seed', [(4, 256, 1), (8, 512, 2), (32, 512, 3)])
def test_bf16_model_a_step_matches_the_bf16_emulating_oracle(hip_lib, B, N, seed):
    C = 4
    batch = make_batch(B, N, C, seed=seed, dropout_scopes={'inst_seg/dp1': ((B, N, 128), 0.5)})
    P = R.init_params(np.random.RandomState(7 + seed), R.layer_table(C, 'A'))
    c = R.default_config()
    rt = Runtime(lib=hip_lib)
    g, m = run(rt, batch, P, c, 'bf16')
    assert m.seg.L2.y.dtype == torch.bfloat16 and m.seg.L9.dz.dtype == torch.bfloat16 and m.seg.L5.y is None
    forced = product_decisions(m)
    forced['bf16'], forced['keep_raw'] = True, {}
    loss, ep, grads, ema = R.model_a_forward_backward(P, batch, c, forced=forced)
    # layer-wise: the stored outputs of the first layers, element by element
    for lay in (m.seg.L1, m.seg.L2, m.tnet.T1):
        ref = forced['keep_raw'][lay.scope].reshape(lay.M, lay.N)
        got = lay.y.double().cpu()
        big = torch.maximum(ref.abs(), got.abs()).clamp(min=1e-30)
        spacing = 2.0 ** (torch.floor(torch.log2(big)) - 7)                # bf16 spacing at the element's magnitude
        diff = (got - ref).abs()
        frac = float((diff > 0).double().mean())
        # + fp32 accumulation noise on elements near zero + ONE operand element rounded the other way (the kernel forms
        # x - centroid / relu(bn(y)) in fp32, the oracle in fp64: one bf16 spacing of an operand of magnitude <= 6 times a weight)
        allowed = 4 * spacing + 1e-5 * float(ref.abs().max()) + 2.0 ** -7 * 6.0 * float(lay.w.abs().max())
        assert frac < 0.02 and bool((diff <= allowed).all()), (lay.scope, frac, float((diff / allowed).max()))
    e = m.end_points()
    worst = {}
    for k in FWD_KEYS:
        ref = ep[k].detach().numpy()
        err = float(np.abs(e[k].float().cpu().numpy().reshape(ref.shape) - ref).max() / max(1.0, np.abs(ref).max()))
        worst[k] = err
    lm, lr = float(e['loss'].cpu()), float(loss)
    per, glob = grad_errors(g, {k: v.numpy() for k, v in grads.items()})
    top = sorted(per.items(), key=lambda kv: -kv[1])[:4]
    med = float(np.median(list(per.values())))
    # for the record: the same run against the plain fp64 oracle (what bf16 costs at this size; see the module docstring)
    loss64, ep64, _, _ = R.model_a_forward_backward(P, batch, c, want_grads=False)
    d64 = {k: float(np.abs(e[k].float().cpu().numpy().reshape(ep64[k].shape) - ep64[k].detach().numpy()).max()) for k in FWD_KEYS}
    print('bf16 vs emulating oracle: fwd', worst, 'loss', (lm, lr), 'grad worst', top, 'median', med, 'global', glob,
          'decisions differing', sum(ep['__flips__'].values()), '| vs plain fp64 oracle: max abs', d64, 'loss', float(loss64))
    mg = ep['__margins__']
    print('forced decisions: site (flips / elements, oracle margin in bf16 spacings of the tensor scale):',
          {k: ('%d/%d' % (v[1], v[0]), round(v[2] / (2.0 ** -8 * max(1.0, v[3])), 2)) for k, v in mg.items() if v[1]})
    check_bf16_decisions(mg, 'bf16')
    for k, err in worst.items():
        assert err < FWD_TOL, (k, err)
    assert abs(lm - lr) < LOSS_TOL * abs(lr), (lm, lr)
    total = np.sqrt(sum(float(np.linalg.norm(v.numpy())) ** 2 for v in grads.values()))
    big = {k: v for k, v in per.items() if float(np.linalg.norm(grads[k].numpy())) >= 1e-2 * total}
    assert max(big.values()) < GRAD_PER and med < GRAD_MED and glob < GRAD_GLOB, (sorted(big.items(), key=lambda kv: -kv[1])[:3], med, glob)
    # ... and EVERY tensor, however small its share of the gradient norm: within 2.5e-1 relative, or -- a tensor whose terms cancel
    # (the beta gradient of a pooled layer over 8 frustums: relative error 0.42 on 0.2 % of the gradient norm) -- off by less than
    # 5e-3 of the whole gradient's norm in absolute terms.  A wrongly wired small tensor is off by its own norm and fails both.
    print('all gradient tensors: worst', top[0], 'of', len(per))
    for k, e in per.items():
        nk = float(np.linalg.norm(grads[k].numpy()))
        assert e < 2.5e-1 or e * nk < 5e-3 * total, (k, e, nk / total)
    for k, v in ema.items():
        mine = g.vars.get(k).detach().cpu().numpy()
        assert np.abs(mine - v.detach().numpy()).max() < EMA_TOL * max(1.0, float(v.abs().max())), k


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


@pytest.mark.parametrize('workload', ['boxpc', 'F'])
def test_bf16_stage_b_and_c_steps_stay_close_to_fp32_and_train(hip_lib, workload):
    """The Box-PC Fit net (configs[2]) and SEMI_MODEL F (configs[3]) with dtype = bf16, through the step object: not an oracle
    comparison (the emulating oracle above covers the shared per-point kernels on model A; every T3D_BF16 kernel has its own test)
    but a guard on the wiring of those two graphs -- same weights and batch as an fp32 step: first loss within 3 % (Box-PC) / 8 % (stage c: its fit term -log(0.01 + p) of an untrained
    Box-PC branch amplifies the bf16 noise of the logits: 8.10 vs 8.50 measured), loss finite and
    decreasing over 8 steps on a fixed batch, a second bf16 run bit-identical."""
    from transferable3d_amd.step import build_training_step
    B, N, C = 32, 1024, 4
    runs = {}
    for dtype, rep in (('f32', 0), ('bf16', 0), ('bf16', 1)):
        g, model, step, loss = build_training_step(Runtime(lib=hip_lib), workload, B, N, C, dtype=dtype, seed=11)
        model.inputs.load(make_batch(B, N, C, seed=5, boxpc=(workload == 'boxpc')))
        cur = []
        for k in range(8):
            step.run()
            cur.append(float(loss))
        torch.cuda.synchronize()
        runs[(dtype, rep)] = (cur, g.vars.params[:g.vars.used].clone())
    f32, b0, b1 = runs[('f32', 0)], runs[('bf16', 0)], runs[('bf16', 1)]
    assert all(np.isfinite(b0[0])) and b0[0][-1] < b0[0][0], b0[0]
    assert abs(b0[0][0] - f32[0][0]) < (3e-2 if workload == 'boxpc' else 8e-2) * abs(f32[0][0]), (b0[0][0], f32[0][0])
    assert b0[0] == b1[0] and torch.equal(b0[1], b1[1])


def test_bf16_four_step_trajectory_adam_master_weights_and_the_bf16_copy(hip_lib):
    """Four consecutive bf16 steps of the step object (the capture and two hipGraph replays among them), each checked from the state
    the product held before it: (1) its gradients against the bf16-emulating oracle along the product's branches -- EVERY tensor (no
    share-of-the-norm filter), per tensor / median / global; (2) Adam: moments and fp32 MASTER weights are the TF-form update of the
    previous state with the PRODUCT's gradient (fp32 rounding only: this is the optimiser's arithmetic, independent of how noisy the
    bf16 gradient is); (3) the bf16 copy the GEMMs read during the step equals the master weights the step started from, rounded once
    (t3d_cast_bf16 runs first in the forward plan); (4) moving statistics against the oracle."""
    from fake_t3d import hash_keep_mask
    from model_check import trajectory_batch
    from transferable3d_amd.step import build_training_step
    B, N, C = 8, 256, 4
    P0 = R.init_params(np.random.RandomState(41), R.layer_table(C, 'A'))
    c = R.default_config()
    rt = Runtime(lib=hip_lib)
    g, model, step, loss_buf = build_training_step(rt, 'A', B, N, C, dtype='bf16', inline_dropout=True, dropout_seed=1234,
                                                   state_dict={k: v.detach().cpu().numpy() for k, v in P0.items()})
    vs = g.vars
    names = [k for k, (off, shape, tr) in vs.index.items() if tr]

    def flat(buf, k):
        off, shape, _ = vs.index[k]
        return torch.as_tensor(buf[off:off + int(np.prod(shape))].detach().float().cpu().numpy().astype(np.float64)).reshape(shape)

    worst = dict(grad=0.0, med=0.0, w=0.0)
    for k in range(4):
        batch = trajectory_batch('A', B, N, C, seed=900 + k)
        Pk = {n_: torch.as_tensor(v.astype(np.float64)) for n_, v in vs.state_dict().items()}
        m0 = {n_: flat(vs.adam_m, n_) for n_ in names}
        v0 = {n_: flat(vs.adam_v, n_) for n_ in names}
        model.inputs.load(batch)
        torch.cuda.synchronize()
        step.run()
        torch.cuda.synchronize()
        # (3) the bf16 copy used by this step's GEMMs == bf16(master weights before the step)
        for n_ in names:
            if n_.endswith('/weights') and 'conv' in n_ and 'conv10' not in n_:
                off, shape, _ = vs.index[n_]
                cp = vs.params16[off:off + int(np.prod(shape))].float().cpu()
                ref = Pk[n_].to(torch.float32).to(torch.bfloat16).float().reshape(-1)
                assert torch.equal(cp, ref), ('bf16 copy', n_, k)
        masks = {'inst_seg/dp1': hash_keep_mask((g.dropout_seed + 0x5EED) & 0xffffffff, k + 1, B * N * 128, 0.5).reshape(B, N, 128)}
        ob = dict(batch)
        ob['dropout_masks'] = masks
        forced = product_decisions(model)
        forced['bf16'] = True
        bn_d, lr = R.bn_decay(k, B), R.learning_rate(k, B)
        loss, ep, grads, ema = R.model_a_forward_backward(Pk, ob, c, bn_decay_val=bn_d, forced=forced)
        check_bf16_decisions(ep['__margins__'], 'bf16 step %d' % k)
        assert abs(float(loss_buf) - float(loss)) < 2e-2 * max(1.0, abs(float(loss))), (k, float(loss_buf), float(loss))
        # (1) every gradient tensor
        per, glob = grad_errors(g, {n_: v.numpy() for n_, v in grads.items()})
        top = sorted(per.items(), key=lambda kv: -kv[1])[:3]
        med = float(np.median(list(per.values())))
        assert top[0][1] < 4e-2 and med < GRAD_MED and glob < GRAD_GLOB, (k, top, med, glob)      # EVERY tensor <= 4e-2 (measured 1.6e-2)
        worst['grad'], worst['med'] = max(worst['grad'], top[0][1]), max(worst['med'], med)
        # (2) Adam on the PRODUCT's gradient
        gp = {n_: flat(vs.grads, n_) for n_ in grads}
        Pn = {n_: Pk[n_].clone() for n_ in grads}
        m1 = {n_: m0[n_].clone() for n_ in grads}
        v1 = {n_: v0[n_].clone() for n_ in grads}
        R.adam_tf_step(Pn, gp, m1, v1, k + 1, lr)
        for n_ in grads:
            mscale, vscale = max(float(m1[n_].abs().max()), 1e-30), max(float(v1[n_].max()), 1e-60)
            assert float((flat(vs.adam_m, n_) - m1[n_]).abs().max()) < 1e-5 * mscale, ('adam m', n_, k)
            assert float((flat(vs.adam_v, n_) - v1[n_]).abs().max()) < 5e-5 * vscale, ('adam v', n_, k)      # g*g, then the blend: two fp32 roundings
            sel = torch.sqrt(v1[n_]) > 1e-2 * np.sqrt(vscale)            # entries whose second moment is above fp32 noise
            if bool(sel.any()):
                werr = float((flat(vs.params, n_) - Pn[n_]).abs()[sel].max())
                assert werr < 2e-5, ('master weights', n_, k, werr)
                worst['w'] = max(worst['w'], werr)
        # (4) moving statistics
        for n_, v in ema.items():
            mine = vs.get(n_).detach().cpu().numpy()
            assert np.abs(mine - v.detach().numpy()).max() < EMA_TOL * max(1.0, float(v.abs().max())), (n_, k)
    print('bf16 trajectory: worst per-tensor gradient error %.3g, worst median %.3g, worst master-weight error %.3g' %
          (worst['grad'], worst['med'], worst['w']))
