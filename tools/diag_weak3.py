import sys
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import numpy as np, torch
from oracle import ref_torch as R
from transferable3d_amd import abi
from transferable3d_amd.engine import Runtime
from test_weak_cpu import weak_case, run_model_a
from model_check import product_decisions
lib = abi.load()
batch = weak_case(seed=3)
P = R.init_params(np.random.RandomState(5), R.layer_table(4, 'A'))
c = R.default_config(WEAK_WEIGHT_SURFACE=1.0)
g, m = run_model_a(Runtime(lib=lib), batch, P, c)
torch.cuda.synchronize()
forced = product_decisions(m)
# oracle with access to intermediates
names = R.trainable_names(P)
Pl = {k: (v.detach().double().requires_grad_(k in names)) for k, v in P.items()}
masks = {k: torch.as_tensor(v) for k, v in batch.get('dropout_masks', {}).items()}
ctx = R.Ctx(Pl, is_training=True, bn_decay=0.5, dropout_masks=masks)
R._apply_forced(ctx, forced)
pc = torch.as_tensor(batch['pc'], dtype=torch.float64); oh = torch.as_tensor(batch['one_hot_vec'], dtype=torch.float64)
pred, ep = R.get_semi_model_backbone(ctx, pc, oh, False)
ep['weak_inputs'] = {k: torch.as_tensor(batch[k], dtype=torch.float64) for k in ('Rtilt', 'K', 'rot_frust', 'box2D', 'img_dim')}
loss = R.get_semi_loss_backbone(pred, R._labels_to_torch(batch, torch.float64), ep, c)
print([k for k in ep.keys()][:40])
box_out = pred[1] if not isinstance(pred[1], tuple) else None
gb, gs1, gl = torch.autograd.grad(loss, [ep['box_params'], ep['stage1_center'], ep['logits']], retain_graph=True, allow_unused=True)
lo = m.loss_op
print('dbox err', float((lo.dbox.double().cpu() - gb).abs().max()), float(gb.abs().max()))
print('dstage1 prod\n', lo.dstage1.cpu().numpy(), '\noracle d/dstage1 (direct + through box input translation)\n', gs1.numpy())
d = (lo.dbox.double().cpu() - gb)
idx = (d.abs() > 1e-4).nonzero()
print('dbox mismatches (row, col, prod, ref):', [(int(i), int(j), float(lo.dbox[i, j]), float(gb[i, j])) for i, j in idx][:20])
print('dbox7 prod', m.weak.dbox7.cpu().numpy())
gc, gd, gt = torch.autograd.grad(loss, list(ep['S_pred_box_reg']), retain_graph=True, allow_unused=True)
print('oracle d/dS: center', gc.numpy(), 'dims', None if gd is None else gd.numpy(), 'theta', gt.numpy())
