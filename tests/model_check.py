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


def product_decisions(model):
    """The piecewise-linear decisions the product's step actually took, in the form oracle.ref_torch.Ctx.forced_gates /
    forced_argmax wants them: for every dense per-point layer the sign of its ReLU input (the kernels evaluate
    fmaf(y, scale, shift) > 0, i.e. the sign of the exact product-sum, reproduced here in fp64), for every max-pooled layer the
    arg-max row per (frustum, channel) (-1 where the pooled value is 0), for every fully-connected layer with a (leaky) ReLU the
    sign of its output; and the hard segmentation mask.  Lets the gradient checks bound EVERY tensor near 1e-4 instead of tolerating
    ReLU-boundary flips."""
    from transferable3d_amd.engine import FcLayer, PointLayer
    gates, argmax, seen = {}, {}, set()

    def walk(o, depth=0):
        if id(o) in seen or depth > 3:
            return
        seen.add(id(o))
        if isinstance(o, PointLayer):
            if getattr(o, 'src', None) is None:
                return                                         # never emitted (a net built but not on this graph's path)
            B, Np = o.g.B, o.g.rpf
            key = getattr(o, 'decision_scope', o.scope)     # a re-used scope's further evaluations: oracle.SharedParams
            if o.pool:
                argmax[key] = o.argidx.detach().cpu().numpy().reshape(B, o.N).astype(np.int64)
            if o.y is not None:
                z = o.y.double() * o.scale.double() + o.shift.double()
                gates[key] = (z > 0).cpu().numpy().reshape(B, Np, o.N)
            return
        if isinstance(o, FcLayer):
            if o.act in ('relu', 'leaky_relu') and getattr(o, 'x', None) is not None:
                # the sign of the output -- except where a dropout that follows the activation (box_pc_mask_model/fc1, fc2, the
                # box_refine layers) zeroed it: there the gate cannot be read off `out` and does not matter (the element's value and
                # gradient are 0 either way), so it is left to the oracle (-1).  (Recomputing the activation's input from the stored
                # raw output would need gamma / beta as they were BEFORE this step's Adam update.)
                gate = (o.out > 0).to(torch.int8)
                if getattr(o, 'use_drop', False) and o.is_training:
                    gate = torch.where(o.drop_mask > 0, gate, torch.full_like(gate, -1))
                gates[getattr(o, 'decision_scope', o.scope)] = gate.cpu().numpy()
            return
        if isinstance(o, (list, tuple)):
            for v in o:
                walk(v, depth + 1)
            return
        if hasattr(o, '__dict__') and type(o).__module__.startswith('transferable3d_amd'):
            for v in vars(o).values():
                walk(v, depth + 1)
    walk(model)
    out = {'gates': gates, 'argmax': argmax}
    seg = getattr(model, 'seg', None)
    if seg is not None and getattr(seg, 'mask', None) is not None:
        # the hard segmentation mask (semisup_models.py:150: logit0 < logit1) is the same kind of decision: a point whose two logits
        # agree to fp32 rounding may fall on either side, which moves the masked centroid and everything behind it
        out['mask'] = seg.mask.detach().cpu().numpy().reshape(seg.g.B, seg.g.rpf)
    weak, lo = getattr(model, 'weak', None), getattr(model, 'loss_op', None)
    if weak is not None and lo is not None:
        # the weak surface / reprojection losses are piecewise functions of the predicted box with a kink per point and face (|.|,
        # the minimum over six faces, min / max over eight corners, the Huber bands): a forward difference of 3e-5 in the box centre
        # moved d surface / d centre_y of one sample from +0.002 to -0.023 (tools/diag_weak*.py).  The oracle therefore evaluates
        # those losses AT the box the product predicted (value substituted, gradient straight through to its own box).
        out['S_box'] = (lo.center.detach().double().cpu(), lo.reg_dims.detach().double().cpu(), lo.reg_theta.detach().double().cpu())
    return out


def check_decision_margins(margins, flip_frac=1e-4, flip_floor=2, margin_tol=1e-4, what=''):
    """Bound what the product may dictate to the oracle (oracle.ref_torch.Ctx.margins): per forced site -- a layer's ReLU gates, a
    pooled layer's arg-max rows, the hard segmentation mask -- at most `flip_frac` of the elements (never fewer than `flip_floor`
    allowed: one flip in a 32 x 128 FC layer is already 2.4e-4) decided differently from the oracle's own evaluation, and every such
    decision within `margin_tol` * max(1, tensor magnitude) of the boundary BY THE ORACLE'S OWN NUMBERS.  fp32 defaults (observed on
    MI355X: 4-50 flips per step in all, margins <= 3e-6); the bf16 tests pass 2 % and four bf16 spacings.  A kernel that gated,
    pooled or masked wrongly on a systematic subset fails here instead of dragging the oracle's gradient with it."""
    worst = {}
    for site, (numel, flips, margin, scale) in margins.items():
        assert flips <= max(flip_floor, flip_frac * numel), (what, site, 'decisions differing from the oracle', flips, 'of', numel)
        assert margin <= margin_tol * max(1.0, scale), (what, site, 'oracle margin of a forced decision', margin, 'tensor scale', scale)
        if flips:
            worst[site] = (flips, margin)
    return worst


MED_TOL = {'fp32_mfma': 5e-5, 'bf16x3': 1e-4}      # median per-tensor gradient error allowed, by GEMM arithmetic of the fp32 path


def tight_grad_check(g, ref_grads, per_tol=1e-3, med_tol=None, glob_tol=1e-4, what=''):
    """Every gradient tensor within `per_tol` (relative L2), the median within `med_tol`, all together within `glob_tol` -- the
    bounds an fp32 implementation meets against the fp64 restatement once both differentiate the same ReLU / arg-max branch
    (product_decisions); a wrong kernel is off by orders of magnitude more, and a 1 % error in ONE small tensor fails.
    Observed on MI355X (profiles/r02_gpu_tests.log): worst tensor 2e-5 ... 3.4e-4 (a 512-float beta gradient whose terms cancel,
    B = 4), median 7e-6 ... 1.4e-5, global 8e-6 ... 2.4e-5, with up to 50 ReLU decisions differing from the fp64 run.
    `med_tol` follows the GEMM arithmetic the graph's runtime was built with (MED_TOL): 5e-5 for the fp32-MFMA kernels (an fma
    chain), 1e-4 for the three-term bf16 form.  From profiles/r05_x3_traj_errors.log (tools/x3_traj_errors.py: workloads F, A and
    boxpc x 4 parameter seeds x 5 steps x both arithmetics = 59 / 60 step checks): the typical step sits at 1e-5 under both; the
    worst step medians are 2.6e-5 (fp32 MFMA) and 5.7e-5 (three-term bf16) -- one step each in which a batch sum of eight nearly
    cancelling terms (tnet fc3 biases) carries the error of everything above it.  The per-tensor and global bounds are common."""
    if med_tol is None:
        med_tol = MED_TOL.get(getattr(getattr(g, 'rt', None), 'gemm_arithmetic', 'bf16x3'), 1e-4)
    per, glob = grad_errors(g, ref_grads)
    worst = sorted(per.items(), key=lambda kv: -kv[1])[:4]
    med = float(np.median(list(per.values())))
    assert worst[0][1] < per_tol, (what, 'per-tensor gradient error', worst)
    assert med < med_tol, (what, 'median per-tensor gradient error', med)
    assert glob < glob_tol, (what, 'global gradient error', glob, worst)
    return dict(grad_max=worst[0], grad_median=med, grad_global=glob)


def grad_errors(g, ref_grads):
    """Per-tensor relative L2 errors and the global relative L2 error of the plan's gradients."""
    per, num, den = {}, 0.0, 0.0
    gscale = max(float(np.abs(np.asarray(r)).max()) for r in ref_grads.values())
    for k, r in ref_grads.items():
        mine = g.vars.grad(k).detach().cpu().numpy().astype(np.float64)
        r = np.asarray(r, dtype=np.float64).reshape(mine.shape)
        e, n = np.linalg.norm(mine - r), np.linalg.norm(r)
        if n > 1e-9:
            # A tensor all of whose entries are <= 1e-3 of the step's largest gradient entry AND agree with the oracle to 1e-6 of that
            # entry sits at the fp32 noise floor of the sums it is made of (its terms cancel: e.g. the beta of box_est/conv-reg4 in
            # stage c, 511 analytically-zero entries and one of 5e-4 beside gammas of 0.4 -- 1.8e-7 of rounding noise per entry reads
            # as 2.7e-3 "relative").  Its relative error says nothing; the absolute agreement is the check (observed on MI355X,
            # tests/test_off_recipe_gpu.py, stage c with norm_box2D, step 1).
            if np.abs(r).max() <= 1e-3 * gscale and np.abs(mine - r).max() <= 1e-6 * gscale:
                per[k] = min(e / n, 1e-6)
            else:
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


def check_against_oracle(g, m, batch, P, c, fwd_atol=1e-4, grad_median_tol=1e-4, flip_aware=True, bn_decay=0.5):
    """Forward tensors within `fwd_atol` (BASELINE.json: fp32 outputs within 1e-4 of the reference restatement).
    Gradients, flip_aware (default): the oracle differentiates the branch of every ReLU / max-pool the product actually took
    (product_decisions) and EVERY tensor is bounded near 1e-4 (tight_grad_check).  flip_aware=False is the former two-level bound
    (median per-tensor <= grad_median_tol, global <= 1e-2) for callers that have no product graph to read decisions from."""
    forced = product_decisions(m) if flip_aware else None
    loss, ep, grads, ema = R.model_a_forward_backward(P, batch, c, bn_decay_val=bn_decay, forced=forced)
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
    if flip_aware:
        gres = tight_grad_check(g, {k: v.numpy() for k, v in grads.items()}, what='model A')
        gres['flips'] = {k: v for k, v in ep['__flips__'].items() if v}
        gres['forced'] = check_decision_margins(ep['__margins__'], what='model A')
    else:
        per, glob = grad_errors(g, {k: v.numpy() for k, v in grads.items()})
        med = float(np.median(list(per.values())))
        if grad_median_tol is not None:
            assert med < grad_median_tol, ('median per-tensor grad error', med)
        assert glob < 1e-2, ('global grad error', glob, sorted(per.items(), key=lambda kv: -kv[1])[:5])
        gres = dict(grad_median=med, grad_global=glob)
    for k, v in ema.items():
        mine = g.vars.get(k).detach().cpu().numpy()
        assert np.abs(mine - v.detach().numpy()).max() < 1e-4 * max(1.0, float(v.abs().max())), k
    return dict(fwd=out, loss=(lmine, lref), **gres)


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


def check_stage_c_inference(rt, refine, use_oracle_mask=False, mask_pc_for_boxpc=False):
    """test_semisup.py:61-262 on synthetic frustums: the inference graph of SEMI_MODEL F (inference-mode batch-norm, no
    dropout, `--refine` Box-PC refinement steps), the F2_ heads and the detection score, against the oracle.  use_oracle_mask
    (test_semisup.py:61,75, semisup_v1_sunrgbd.py:161-162) and --mask_pc_for_boxpc (test_semisup.py:103-105) on both sides."""
    from transferable3d_amd import test_semisup as TS
    from transferable3d_amd.synthetic import make_batch
    B, N, C = 4, 256, 4
    FLAGS = TS.build_flags(['--semi_type', 'F', '--use_one_hot', '--num_point', str(N), '--num_channels', str(C), '--batch_size', str(B),
                            '--refine', str(refine), '--pred_prefix', 'F2_', '--use_boxpc_fit_prob' if refine else '--synthetic'] +
                           (['--mask_pc_for_boxpc'] if mask_pc_for_boxpc else []))
    rng = np.random.RandomState(3)
    P = R.stage_c_params(rng, C)
    for k in P:                                     # non-trivial moving statistics
        if k.endswith('moving_mean'):
            P[k] = torch.as_tensor(rng.normal(size=tuple(P[k].shape)) * 0.2)
        elif k.endswith('moving_variance'):
            P[k] = torch.as_tensor(0.5 + rng.uniform(size=tuple(P[k].shape)))
    sess, ops = TS.get_model(FLAGS, B, N, C, rt=rt,
                             state_dict={k: v.numpy() for k, v in P.items()}, use_oracle_mask=use_oracle_mask)
    batches = [make_batch(B, N, C, seed=40 + i) for i in range(2)]
    pc, oh = np.concatenate([b['pc'] for b in batches]), np.concatenate([b['one_hot_vec'] for b in batches])
    seg_gt = np.concatenate([b['y_seg'] for b in batches])
    seg, centers, hcls, hres, scls, sres, scores = TS.inference(sess, ops, pc, oh, B, prefix='F2_', use_boxpc_fit_prob=bool(refine),
                                                                oracle_mask=seg_gt if use_oracle_mask else None)
    if use_oracle_mask:
        assert np.array_equal(seg, seg_gt)                      # argmax(stack([1 - m, m])) = m
    c = R.default_config(SEMI_REFINE_USING_BOXPC_DELTA_NUM=refine)
    for i, b in enumerate(batches):
        pred, ep = R.stage_c_inference(P, b, c, refine, use_oracle_mask=use_oracle_mask, mask_pc_for_boxpc=mask_pc_for_boxpc)
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


# ------------------------------------------------------------------------------------------------------------------------------
# the TIMED step against an oracle trajectory: transferable3d_amd.step.build_training_step is what bench.py
# runs; here the same object is replayed for several steps and every step is checked against the oracle
# ------------------------------------------------------------------------------------------------------------------------------
def _stage_c_like_params(C, seed, norm_box2D=False):
    import test_stage_c_cpu as TC
    return TC.stage_c_params(C, seed, norm_box2D=norm_box2D)


def trajectory_batch(workload, B, N, C, seed):
    from transferable3d_amd.synthetic import make_batch
    b = make_batch(B, N, C, seed=seed, boxpc=workload == 'boxpc')
    if workload == 'F':
        b['is_data_2D'][::2] = 1
    return b


def _oracle_step(workload, P, batch, c, bn_decay_val, forced):
    from transferable3d_amd.step import STAGE_C_TRAIN_CLASSES
    if workload == 'A':
        return R.model_a_forward_backward(P, batch, c, bn_decay_val=bn_decay_val, forced=forced)
    if workload == 'boxpc':
        return R.boxpc_forward_backward(P, batch, c, bn_decay_val=bn_decay_val, forced=forced)
    return R.stage_c_forward_backward(P, batch, c, STAGE_C_TRAIN_CLASSES, bn_decay_val=bn_decay_val, forced=forced)


def _fwd_pairs(workload, e, ep):
    if workload == 'A':
        return [(k, e[k], ep[k]) for k in FWD_KEYS]
    if workload == 'boxpc':
        return [('boxpc_out', e['boxpc_out'], ep['boxpc_out'])]
    pairs = [(mine, e[mine], ref) for mine, ref in (
        ('logits', ep['logits']), ('stage1_center', ep['stage1_center']), ('feats_lv1', ep['feats_lv1']),
        ('F_box_params', ep['F_box_params']), ('F_center', ep['F_center']), ('boxpc_out', ep['boxpc_out']),
        ('boxpc_fit_prob', ep['boxpc_fit_prob']), ('F_dims', ep['F_pred_box_reg'][1]), ('F_theta', ep['F_pred_box_reg'][2]))]
    if len(ep.get('boxpc_outs', ())) > 1:      # several refinement steps in the training graph: the last evaluation and the totals behind F2_
        pairs += [('boxpc_out_last', e['boxpc_out_last'], ep['boxpc_outs'][-1]), ('total_delta', e['total_delta'], ep['total_delta'])]
    return pairs


def oracle_config(workload):
    if workload == 'A':
        return R.default_config()
    if workload == 'boxpc':
        return R.default_config(BOXPC_WEIGHT_DELTA=4.0)
    import test_stage_c_cpu as TC
    return TC.stage_c_config()


def trajectory_check(rt, workload='A', steps=4, B=8, N=256, C=4, use_hip_graph=None, process_group=None, force_dist=False,
                     flat_allreduce=False, param_seed=31, fwd_tol=1e-4, weight_tol=2e-5, verbose=False, config_over=None):
    """Runs `steps` consecutive steps of the step object bench.py times (pre: device schedules + dropout masks; forward with the
    seg head's in-kernel dropout; backward; TF-form Adam; hipGraph replay from the second step on) and checks EVERY step against
    the oracle started from the state the product held before that step (weights, moving statistics, Adam moments, step counter):
    forward heads and loss <= fwd_tol, every gradient tensor (flip-aware, tight_grad_check), the moving statistics, the Adam
    moments, and the post-step weights <= weight_tol (entries whose second moment is above fp32 noise; conv / FC biases and betas
    that feed a training-mode batch-norm are analytically gradient-free and excluded, as in tests/test_api_cpu.py).  The dropout
    masks are the ones the kernels drew (read back, or recomputed from the generator's specification for the in-kernel mask).
    Teacher-forced per step on purpose: Adam's first steps move every weight by ~lr*sign(g), so a free-running fp64 trajectory
    diverges from ANY fp32 implementation in the entries whose gradient is rounding noise -- that says nothing about the kernels.
    Returns per-step diagnostics (incl. the free-running loss curve of the product)."""
    from fake_t3d import hash_keep_mask
    from transferable3d_amd.step import build_training_step, workload_flags
    over = dict(config_over or {})          # off-recipe flags, set on the product's FLAGS and on the oracle's config alike
    nb = bool(over.get('USE_NORMALIZED_BOX2D_AS_FEATS', False))
    if workload == 'A':
        P0 = R.init_params(np.random.RandomState(param_seed), R.layer_table(C, 'A', norm_box2D=nb))
    elif workload == 'boxpc':
        P0 = R.init_params(np.random.RandomState(param_seed), R.layer_table(C, 'boxpc'))
    else:
        P0 = _stage_c_like_params(C, param_seed, norm_box2D=nb)
    c = oracle_config(workload)
    flags = None
    if over:
        flags = workload_flags(workload)
        for k_, v_ in over.items():
            assert hasattr(flags, k_) and hasattr(c, k_), k_
            setattr(flags, k_, v_)
            setattr(c, k_, v_)
    world = process_group.size() if process_group is not None else 1
    g, model, step, loss_buf = build_training_step(rt, workload, B, N, C, world=world, process_group=process_group,
                                                   force_dist=force_dist, flat_allreduce=flat_allreduce,
                                                   use_hip_graph=use_hip_graph, inline_dropout=True, dropout_seed=1234,
                                                   state_dict={k: v.detach().cpu().numpy() for k, v in P0.items()}, c=flags)
    vs = g.vars
    names = [k for k, (off, shape, tr) in vs.index.items() if tr]
    sync = (lambda: torch.cuda.synchronize()) if rt.device.type == 'cuda' else (lambda: None)

    def moments():
        out = {}
        for k in names:
            off, shape, _ = vs.index[k]
            n = int(np.prod(shape))
            out[k] = (torch.as_tensor(vs.adam_m[off:off + n].detach().cpu().numpy().astype(np.float64)).reshape(shape),
                      torch.as_tensor(vs.adam_v[off:off + n].detach().cpu().numpy().astype(np.float64)).reshape(shape))
        return out

    report = []
    for k in range(steps):
        batch = trajectory_batch(workload, B, N, C, seed=500 + k)
        Pk = {n_: torch.as_tensor(v.astype(np.float64)) for n_, v in vs.state_dict().items()}
        mv = moments()
        model.inputs.load(batch)
        sync()
        step.run()
        sync()
        assert float(g.hyper[0]) == k + 1                           # the device step counter (`batch` of train_semisup.py:214)
        # the masks the kernels drew
        masks = {scope: t.detach().cpu().numpy() for scope, (t, keep) in g.dropout_masks.items()}
        seg_scope = {'A': 'inst_seg/dp1', 'F': 'class_agnostic/inst_seg/dp1'}.get(workload)
        if seg_scope is not None:
            masks[seg_scope] = hash_keep_mask((g.dropout_seed + 0x5EED) & 0xffffffff, k + 1, B * N * 128, 0.5).reshape(B, N, 128)
        ob = dict(batch)
        ob['dropout_masks'] = masks
        bn_d, lr = R.bn_decay(k, B * world), R.learning_rate(k, B * world)
        assert abs(float(g.hyper[1]) - lr) < 1e-9 and abs(float(g.hyper[2]) - bn_d) < 1e-6
        loss, ep, grads, ema = _oracle_step(workload, Pk, ob, c, bn_d, product_decisions(model))
        e = model.end_points()
        worst_fwd = 0.0
        for name, mine, ref in _fwd_pairs(workload, e, ep):
            r = ref.detach().numpy()
            err = float(np.abs(mine.detach().cpu().numpy().reshape(r.shape) - r).max() / max(1.0, np.abs(r).max()))
            assert err < fwd_tol, (workload, 'step', k, name, err)
            worst_fwd = max(worst_fwd, err)
        lmine, lref = float(loss_buf.detach().cpu()), float(loss.detach())
        assert abs(lmine - lref) < fwd_tol * max(1.0, abs(lref)), (workload, 'step', k, 'loss', lmine, lref)
        # gradients of this step (world == 1: the buffer holds them as the backward wrote them)
        gres = tight_grad_check(g, {n_: v.numpy() for n_, v in grads.items()}, what='%s step %d' % (workload, k)) if world == 1 else {}
        check_decision_margins(ep['__margins__'], what='%s step %d' % (workload, k))
        for n_, v in ema.items():
            mine = vs.get(n_).detach().cpu().numpy()
            assert np.abs(mine - v.detach().numpy()).max() < 1e-5 * max(1.0, float(v.abs().max())), (workload, 'step', k, n_)
        # Adam: moments and weights from the oracle's gradient and the product's previous moments
        Pn = {n_: Pk[n_].clone() for n_ in grads}
        m_ = {n_: mv[n_][0].clone() for n_ in grads}
        v_ = {n_: mv[n_][1].clone() for n_ in grads}
        R.adam_tf_step(Pn, grads, m_, v_, k + 1, lr)
        mv1 = moments()
        checked = total = 0
        worst_w = 0.0
        for n_ in grads:
            if n_.endswith('/biases') and (n_[:-7] + '/bn/gamma') in Pk:
                continue
            if n_.endswith('/bn/beta') and float(grads[n_].abs().max()) < 1e-12:
                continue
            mref, vref = m_[n_].numpy(), v_[n_].numpy()
            mscale, vscale = max(np.abs(mref).max(), 1e-30), max(vref.max(), 1e-60)
            assert np.abs(mv1[n_][0].numpy() - mref).max() < 3e-4 * mscale, (workload, 'step', k, 'adam m', n_)
            assert np.abs(mv1[n_][1].numpy() - vref).max() < 6e-4 * vscale, (workload, 'step', k, 'adam v', n_)
            got = vs.get(n_).detach().cpu().numpy().astype(np.float64).reshape(Pn[n_].shape)
            sel = np.sqrt(vref) > 1e-2 * np.sqrt(vscale)
            total += sel.size
            if sel.any():
                werr = float(np.abs(got - Pn[n_].numpy())[sel].max())
                assert werr < weight_tol, (workload, 'step', k, 'weights', n_, werr)
                worst_w = max(worst_w, werr)
                checked += int(sel.sum())
            # nothing moves more than Adam can move it
            assert float(np.abs(got - Pk[n_].numpy()).max()) < 3.2 * lr + 1e-7, (workload, 'step', k, 'step size', n_)
        # variables outside the var_list do not move at all
        for n_ in names:
            if n_ not in grads:
                assert np.array_equal(vs.get(n_).detach().cpu().numpy(), Pk[n_].numpy().astype(np.float32)), (workload, k, n_)
        report.append(dict(step=k, loss=lmine, loss_ref=lref, fwd=worst_fwd, weights=worst_w, weight_entries_checked=checked,
                           weight_entries=total, flips={a: b for a, b in ep['__flips__'].items() if b}, **gres))
        if verbose:
            print(report[-1])
    report.append(dict(graph_segments=step.n_graph_segments(), launches=step.n_launches()))
    return report
