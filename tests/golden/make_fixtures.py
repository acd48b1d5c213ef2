"""Generates tests/golden/*.npz from the oracle (run in the build container:
`python tests/golden/make_fixtures.py`).  The reference itself cannot run anywhere (TensorFlow 1 is not
installable), so the vectors come from the cross-checked fp64 restatement in oracle/.

Fixture layout: batch/* inputs, mask/* dropout keep masks, param_seed (weights are re-drawn from
np.random.RandomState(param_seed) by oracle.ref_torch.init_params to keep the file small), out/* forward
tensors and losses, grad/* gradients of the small tensors, gradnorm/* = L2 norm of every gradient."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import ref_torch as R                      # noqa: E402
from transferable3d_amd.synthetic import make_batch    # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def model_a(B=4, N=128, C=4, seed=11, pseed=5):
    batch = make_batch(B, N, C, seed=seed, dropout_scopes={'inst_seg/dp1': ((B, N, 128), 0.5)})
    P = R.init_params(np.random.RandomState(pseed), R.layer_table(C, 'A'))
    c = R.default_config()
    loss, ep, grads, ema = R.model_a_forward_backward(P, batch, c)
    out = {'param_seed': np.int64(pseed)}
    for k, v in batch.items():
        if k != 'dropout_masks':
            out['batch/' + k] = v
    out['mask/inst_seg/dp1'] = batch['dropout_masks']['inst_seg/dp1']
    for k in ('logits', 'stage1_center', 'center', 'box_params', 'feats_lv1', 'mask_xyz_mean'):
        out['out/' + k] = ep[k].detach().numpy()
    out['out/loss'] = np.float64(loss.detach())
    for k, v in ep['loss_terms'].items():
        out['out/term_' + k] = v.detach().numpy()
    for k, g in grads.items():
        out['gradnorm/' + k] = np.float64(g.norm())
        if g.numel() <= 4096:
            out['grad/' + k] = g.numpy()
    out['grad/box_est/fc1/weights'] = grads['box_est/fc1/weights'].numpy()[:8]
    for k, v in ema.items():
        out['ema/' + k] = v.detach().numpy()
    np.savez_compressed(os.path.join(HERE, 'model_a_B%d_N%d.npz' % (B, N)), **out)


sys.path.insert(0, os.path.join(ROOT, 'tests'))


def boxpc(B=4, N=256, C=4):
    """Box-PC Fit net step (config 2) on the inputs of tests/test_boxpc_cpu.py: outputs, loss, gradient norms."""
    import test_boxpc_cpu as TB
    batch = make_batch(B, N, C, seed=3, boxpc=True, dropout_scopes=TB.SCOPES(B))
    P = R.init_params(np.random.RandomState(5), R.layer_table(C, 'boxpc'))
    c = R.default_config(BOXPC_WEIGHT_DELTA=4.0)
    loss, ep, grads, _ = R.boxpc_forward_backward(P, batch, c)
    out = {'out/loss': np.float64(loss.detach()), 'out/boxpc_out': ep['boxpc_out'].detach().numpy()}
    for k, g in grads.items():
        out['gradnorm/' + k] = np.float64(g.norm())
    np.savez_compressed(os.path.join(HERE, 'boxpc_B%d_N%d.npz' % (B, N)), **out)


def stage_c(B=4, N=256, C=4, refine=2):
    """Stage-c step (config 3) on the inputs of tests/test_stage_c_cpu.py, and the inference graph with `refine` Box-PC
    refinement steps (test_semisup.py) on the same weights."""
    import test_stage_c_cpu as TC
    P = TC.stage_c_params(C, 21)
    batch = TC.stage_c_batch(B, N, C, 22, 2)
    c = TC.stage_c_config()
    loss, ep, grads, _ = R.stage_c_forward_backward(P, batch, c, TC.TRAIN_CLASSES)
    out = {'out/loss': np.float64(loss.detach()), 'out/boxpc_fit_prob': ep['boxpc_fit_prob'].detach().numpy(),
           'out/F_center': ep['F_center'].detach().numpy()}
    for k, g in grads.items():
        out['gradnorm/' + k] = np.float64(g.norm())
    _, epi = R.stage_c_inference(P, batch, c, refine)
    out['infer/refine'] = np.int64(refine)
    for k in ('F2_center', 'F2_heading_residuals', 'F2_size_residuals', 'boxpc_fit_prob'):
        out['infer/' + k] = epi[k].detach().numpy()
    out['infer/score'] = R.inference_scores(epi['logits'].detach().numpy(), epi['F2_heading_scores'].detach().numpy(),
                                            epi['F2_size_scores'].detach().numpy(), epi['boxpc_fit_prob'].detach().numpy())
    np.savez_compressed(os.path.join(HERE, 'stage_c_B%d_N%d.npz' % (B, N)), **out)


if __name__ == '__main__':
    model_a()
    boxpc()
    stage_c()
    print('fixtures written to', HERE)
