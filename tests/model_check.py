"""Shared helpers: build the product's SEMI_MODEL A plan on a runtime, run it, and compare against the
oracle / golden vectors.  Used with the NumPy spec library on CPU and with the HIP library on the GPU."""
import os

import numpy as np
import torch

from oracle import ref_torch as R
from transferable3d_amd.nets import Graph, SemiModelA

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')

FWD_KEYS = ('logits', 'stage1_center', 'center', 'box_params', 'feats_lv1', 'mask_xyz_mean')
TERM_ORDER = ('mask', 'center', 'stage1', 'hcls', 'hres', 'scls', 'sres', 'corner')


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name))
    batch = {k[6:]: z[k] for k in z.files if k.startswith('batch/')}
    batch['dropout_masks'] = {k[5:]: z[k] for k in z.files if k.startswith('mask/')}
    C = batch['pc'].shape[-1]
    P = R.init_params(np.random.RandomState(int(z['param_seed'])), R.layer_table(C, 'A'))
    return batch, P, z


def run_model_a(rt, batch, P, c, bn_decay=0.5, train=True, is_training=True):
    """One fwd(+bwd) of the product plan; returns (graph, model)."""
    B, N, C = batch['pc'].shape
    g = Graph(B, N, C, rt=rt)
    m = SemiModelA(g, c)
    g.vars.load_state_dict({k: v.detach().cpu().numpy() for k, v in P.items()})
    g.hyper[2] = bn_decay
    m.emit_forward(g.fwd, is_training, True)
    if train:
        m.emit_backward(g.bwd)
    g.finalize()
    m.inputs.load(batch)
    g.fwd.run()
    if train:
        g.bwd.run()
    if rt.device.type == 'cuda':
        torch.cuda.synchronize()
    return g, m


def grad_errors(g, ref_grads):
    """Per-tensor relative L2 errors and the global relative L2 error of the plan's gradients."""
    per, num, den = {}, 0.0, 0.0
    gscale = max(float(np.abs(np.asarray(r)).max()) for r in ref_grads.values())
    for k, r in ref_grads.items():
        mine = g.vars.grad(k).detach().cpu().numpy().astype(np.float64)
        r = np.asarray(r, dtype=np.float64).reshape(mine.shape)
        e, n = np.linalg.norm(mine - r), np.linalg.norm(r)
        if n > 1e-9:
            per[k] = e / n
        else:
            # analytically-zero gradients (a bias or beta feeding a batch-norm): rounding noise only
            assert np.abs(mine).max() < 1e-5 * gscale, (k, float(np.abs(mine).max()), gscale)
        num += e * e
        den += n * n
    return per, float(np.sqrt(num / den))


def iou_summary_check(e, ep, batch, mine_prefix, ref_prefix, tol=1e-4):
    """end_points iou2ds / iou3ds (get_iou_summary, semisup_v1_sunrgbd.py:236-246) against the oracle's compute_box3d_iou on the
    oracle's own heads."""
    from oracle import ref_box as RB
    h = lambda k: ep[ref_prefix + k].detach().numpy().astype(np.float64)
    i2, i3 = RB.compute_box3d_iou(h('center'), h('heading_scores'), h('heading_residuals'), h('size_scores'), h('size_residuals'),
                                  np.asarray(batch['y_center'], np.float64), np.asarray(batch['y_orient_cls']),
                                  np.asarray(batch['y_orient_reg'], np.float64), np.asarray(batch['y_dims_cls']),
                                  np.asarray(batch['y_dims_reg'], np.float64))
    assert np.abs(e[mine_prefix + 'iou3ds'].detach().cpu().numpy() - i3).max() < tol, mine_prefix + 'iou3ds'
    assert np.abs(e[mine_prefix + 'iou2ds'].detach().cpu().numpy() - i2).max() < tol, mine_prefix + 'iou2ds'
    return i3


def check_against_oracle(g, m, batch, P, c, fwd_atol=1e-4, grad_median_tol=1e-4):
    """Forward tensors within `fwd_atol` (BASELINE.json: fp32 outputs within 1e-4 of the reference
    restatement); gradients: median per-tensor relative L2 error <= 1e-4 and global <= 1e-2.  The two-level
    gradient bound is deliberate: a ReLU whose pre-activation is within fp32 rounding of zero can legitimately
    flip between the fp32 path and the fp64 oracle, moving single elements of a few tensors (observed and
    analysed in DESIGN.md); a systematic error moves the median."""
    loss, ep, grads, ema = R.model_a_forward_backward(P, batch, c)
    e = m.end_points()
    out = {}
    for k in FWD_KEYS:
        ref = ep[k].detach().numpy()
        mine = e[k].detach().cpu().numpy().reshape(ref.shape)
        out[k] = float(np.abs(mine - ref).max())
        assert out[k] < fwd_atol * max(1.0, float(np.abs(ref).max())), (k, out[k], float(np.abs(ref).max()))
    terms = e['loss_terms'].detach().cpu().numpy()
    for i, name in enumerate(TERM_ORDER):
        ref = ep['loss_terms'][name].detach().numpy()
        assert np.abs(terms[:, i] - ref).max() < fwd_atol * max(1.0, np.abs(ref).max()), name
    lref = float(loss.detach())
    lmine = float(e['loss'].detach().cpu())
    assert abs(lmine - lref) < 1e-4 * max(1.0, abs(lref)), (lmine, lref)
    _, dims, theta = ep['S_pred_box_reg']
    assert np.abs(e['S_dims'].detach().cpu().numpy() - dims.detach().numpy()).max() < fwd_atol
    assert np.abs(e['S_theta'].detach().cpu().numpy() - theta.detach().numpy()).max() < fwd_atol
    iou_summary_check(e, ep, batch, '', '')
    per, glob = grad_errors(g, {k: v.numpy() for k, v in grads.items()})
    med = float(np.median(list(per.values())))
    if grad_median_tol is not None:
        assert med < grad_median_tol, ('median per-tensor grad error', med)
    assert glob < 1e-2, ('global grad error', glob, sorted(per.items(), key=lambda kv: -kv[1])[:5])
    for k, v in ema.items():
        mine = g.vars.get(k).detach().cpu().numpy()
        assert np.abs(mine - v.detach().numpy()).max() < 1e-4 * max(1.0, float(v.abs().max())), k
    return dict(fwd=out, grad_median=med, grad_global=glob, loss=(lmine, lref))


def check_config0_single_frustum_forward(rt, seed=5):
    """BASELINE.json configs[0]: SEMI_MODEL A forward on ONE synthetic frustum (N=1024, C=4), inference-mode batch-norm
    (moving statistics; a training-mode FC batch-norm over one row is degenerate).  Outputs within 1e-4 of the oracle."""
    import numpy as np
    from oracle import ref_torch as R
    from transferable3d_amd.synthetic import make_batch
    B, N, C = 1, 1024, 4
    batch = make_batch(B, N, C, seed=seed)
    rng = np.random.RandomState(seed)
    P = R.init_params(rng, R.layer_table(C, 'A'))
    for k in P:                                     # non-trivial moving statistics, as after training
        if k.endswith('moving_mean'):
            P[k] = torch.as_tensor(rng.normal(size=tuple(P[k].shape)) * 0.2)
        elif k.endswith('moving_variance'):
            P[k] = torch.as_tensor(0.5 + rng.uniform(size=tuple(P[k].shape)))
    c = R.default_config()
    g, m = run_model_a(rt, batch, P, c, train=False, is_training=False)
    loss, ep, _, _ = R.model_a_forward_backward(P, batch, c, is_training=False, want_grads=False)
    e = m.end_points()
    worst = {}
    for k in FWD_KEYS:
        ref = ep[k].detach().numpy()
        mine = e[k].detach().cpu().numpy().reshape(ref.shape)
        worst[k] = float(np.abs(mine - ref).max() / max(1.0, np.abs(ref).max()))
        assert worst[k] < 1e-4, (k, worst[k])
    assert abs(float(e['loss'].cpu()) - float(loss)) < 1e-4 * max(1.0, abs(float(loss)))
    return worst


def check_stage_c_inference(rt, refine):
    """test_semisup.py:61-262 on synthetic frustums: the inference graph of SEMI_MODEL F (inference-mode batch-norm, no
    dropout, `--refine` Box-PC refinement steps), the F2_ heads and the detection score, against the oracle."""
    from transferable3d_amd import test_semisup as TS
    from transferable3d_amd.synthetic import make_batch
    B, N, C = 4, 256, 4
    FLAGS = TS.build_flags(['--semi_type', 'F', '--use_one_hot', '--num_point', str(N), '--num_channels', str(C), '--batch_size', str(B),
                            '--refine', str(refine), '--pred_prefix', 'F2_', '--use_boxpc_fit_prob' if refine else '--synthetic'])
    rng = np.random.RandomState(3)
    P = R.stage_c_params(rng, C)
    for k in P:                                     # non-trivial moving statistics
        if k.endswith('moving_mean'):
            P[k] = torch.as_tensor(rng.normal(size=tuple(P[k].shape)) * 0.2)
        elif k.endswith('moving_variance'):
            P[k] = torch.as_tensor(0.5 + rng.uniform(size=tuple(P[k].shape)))
    sess, ops = TS.get_model(FLAGS, B, N, C, rt=rt,
                             state_dict={k: v.numpy() for k, v in P.items()})
    batches = [make_batch(B, N, C, seed=40 + i) for i in range(2)]
    pc, oh = np.concatenate([b['pc'] for b in batches]), np.concatenate([b['one_hot_vec'] for b in batches])
    seg, centers, hcls, hres, scls, sres, scores = TS.inference(sess, ops, pc, oh, B, prefix='F2_', use_boxpc_fit_prob=bool(refine))
    c = R.default_config(SEMI_REFINE_USING_BOXPC_DELTA_NUM=refine)
    for i, b in enumerate(batches):
        pred, ep = R.stage_c_inference(P, b, c, refine)
        sl = slice(i * B, (i + 1) * B)
        num = lambda t: t.detach().numpy()
        assert np.abs(centers[sl] - num(ep['F2_center'])).max() < 1e-4
        assert np.array_equal(hcls[sl], np.argmax(num(ep['F2_heading_scores']), 1))
        assert np.array_equal(scls[sl], np.argmax(num(ep['F2_size_scores']), 1))
        hr = num(ep['F2_heading_residuals'])[np.arange(B), hcls[sl]]
        sr = num(ep['F2_size_residuals'])[np.arange(B), scls[sl]]
        assert np.abs(hres[sl] - hr).max() < 1e-4 and np.abs(sres[sl] - sr).max() < 1e-4
        ref_scores = R.inference_scores(num(ep['logits']), num(ep['F2_heading_scores']), num(ep['F2_size_scores']),
                                        num(ep['boxpc_fit_prob']) if refine else None)
        assert np.abs(scores[sl] - ref_scores).max() < 1e-3
        assert (seg[sl] == np.argmax(num(ep['logits']), 2)).mean() > 0.999


def check_golden_boxpc(rt):
    """Committed vectors of the Box-PC Fit net step (tests/golden/boxpc_B4_N256.npz, make_fixtures.boxpc)."""
    import test_boxpc_cpu as TB
    from transferable3d_amd.synthetic import make_batch
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'boxpc_B4_N256.npz'))
    B, N, C = 4, 256, 4
    batch = make_batch(B, N, C, seed=3, boxpc=True, dropout_scopes=TB.SCOPES(B))
    P = R.init_params(np.random.RandomState(5), R.layer_table(C, 'boxpc'))
    g, m = TB.run_boxpc(rt, batch, P, R.default_config(BOXPC_WEIGHT_DELTA=4.0))
    e = m.end_points()
    assert np.abs(e['boxpc_out'].detach().cpu().numpy() - z['out/boxpc_out']).max() < 1e-4
    assert abs(float(e['loss'].cpu()) - float(z['out/loss'])) < 1e-4 * float(z['out/loss'])
    rel = [abs(float(g.vars.grad(k[9:]).norm()) - float(z[k])) / max(float(z[k]), 1e-12) for k in z.files
           if k.startswith('gradnorm/') and float(z[k]) > 1e-9]
    assert np.median(rel) < 1e-4 and max(rel) < 5e-2, (np.median(rel), max(rel))


def check_golden_stage_c(rt):
    """Committed vectors of the stage-c step and of the inference graph with two Box-PC refinement steps
    (tests/golden/stage_c_B4_N256.npz, make_fixtures.stage_c)."""
    import test_stage_c_cpu as TC
    from transferable3d_amd import test_semisup as TS
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'stage_c_B4_N256.npz'))
    B, N, C = 4, 256, 4
    P, batch, c = TC.stage_c_params(C, 21), TC.stage_c_batch(B, N, C, 22, 2), TC.stage_c_config()
    g, m = TC.run_stage_c(rt, batch, P, c)
    e = m.end_points()
    num = lambda t: t.detach().cpu().numpy()
    assert np.abs(num(e['F_center']) - z['out/F_center']).max() < 1e-4
    assert np.abs(num(e['boxpc_fit_prob']) - z['out/boxpc_fit_prob']).max() < 1e-4
    assert abs(float(num(e['loss'])) - float(z['out/loss'])) < 1e-4 * float(z['out/loss'])
    rel = [abs(float(g.vars.grad(k[9:]).norm()) - float(z[k])) / max(float(z[k]), 1e-12) for k in z.files
           if k.startswith('gradnorm/') and float(z[k]) > 1e-9]
    assert np.median(rel) < 2e-4 and max(rel) < 5e-2, (np.median(rel), max(rel))
    refine = int(z['infer/refine'])
    FLAGS = TS.build_flags(['--semi_type', 'F', '--use_one_hot', '--num_point', str(N), '--num_channels', str(C), '--batch_size', str(B),
                            '--refine', str(refine), '--use_boxpc_fit_prob'])
    sess, ops = TS.get_model(FLAGS, B, N, C, rt=rt, state_dict={k: v.numpy() for k, v in P.items()})
    _, centers, hcls, hres, scls, sres, scores = TS.inference(sess, ops, batch['pc'], batch['one_hot_vec'], B, prefix='F2_',
                                                              use_boxpc_fit_prob=True)
    assert np.abs(centers - z['infer/F2_center']).max() < 1e-4
    assert np.abs(hres - z['infer/F2_heading_residuals'][np.arange(B), hcls]).max() < 1e-4
    assert np.abs(sres - z['infer/F2_size_residuals'][np.arange(B), scls]).max() < 1e-4
    assert np.abs(scores - z['infer/score']).max() < 1e-3
