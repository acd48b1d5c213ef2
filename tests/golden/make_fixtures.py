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


if __name__ == '__main__':
    model_a()
    print('fixtures written to', HERE)
