import sys
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import numpy as np, torch
from oracle import ref_torch as R, ref_weak as W
from transferable3d_amd import abi
from transferable3d_amd.engine import Runtime
from test_weak_cpu import weak_case, run_model_a
from model_check import product_decisions, grad_errors
lib = abi.load()
batch = weak_case(seed=3)
P = R.init_params(np.random.RandomState(5), R.layer_table(4, 'A'))
c = R.default_config(WEAK_WEIGHT_REPROJECTION=0.01, WEAK_WEIGHT_SURFACE=1.0, SEMI_MULTIPLIER_FOR_WEAK_LOSS=1.0)
g, m = run_model_a(Runtime(lib=lib), batch, P, c)
torch.cuda.synchronize()
lo, wk = m.loss_op, m.weak
f64 = lambda t: t.detach().double().cpu()
center, dims, theta = f64(lo.center).requires_grad_(True), f64(lo.reg_dims).requires_grad_(True), f64(lo.reg_theta).requires_grad_(True)
tb = lambda k: torch.as_tensor(batch[k], dtype=torch.float64)
r = W.get_reprojection_loss((center, dims, theta), tb('box2D'), tb('Rtilt'), tb('K'), tb('img_dim'), tb('rot_frust'), False, 10., 1.5, True, False, 'huber', [True]*3)
soft = torch.softmax(f64(m.seg.logits).reshape(4, -1, 2), -1)[:, :, 1].requires_grad_(True)
s = W.get_surface_loss((center, dims, theta), tb('pc')[:, :, :3], soft, 0., 0.9, [True, False, True])
is2d = tb('is_data_2D')
l = (is2d * (0.01 * r + 1.0 * s)).mean()
r_keep = r
gs = torch.autograd.grad(l, [center, dims, theta, soft], retain_graph=True)
g7 = torch.cat([gs[0], gs[1], gs[2][:, None]], 1)
print('reproj prod', f64(wk.reproj).numpy(), 'ref', r.detach().numpy())
print('surf prod', f64(wk.surface).numpy(), 'ref', s.detach().numpy())
print('dbox7 prod\n', f64(wk.dbox7).numpy(), '\nref\n', g7.numpy())
print('dsoft err', float((f64(wk.dsoft).reshape(4, -1) - gs[3]).abs().max()), float(gs[3].abs().max()))
forced = product_decisions(m)
loss, ep, grads, ema = R.model_a_forward_backward(P, batch, c, forced=forced)
per, glob = grad_errors(g, {k: v.numpy() for k, v in grads.items()})
print(sorted(per.items(), key=lambda kv: -kv[1])[:12], glob)
print('S box prod', f64(lo.center).numpy(), 'ref', ep['S_pred_box_reg'][0].detach().numpy())
# sensitivity: the oracle's own d loss / d box at ITS box vs at the product's box
co, do, to = [t.detach().clone().requires_grad_(True) for t in ep['S_pred_box_reg']]
so = ep['soft_mask'].detach().clone().requires_grad_(True)
r2 = W.get_reprojection_loss((co, do, to), tb('box2D'), tb('Rtilt'), tb('K'), tb('img_dim'), tb('rot_frust'), False, 10., 1.5, True, False, 'huber', [True]*3)
s2 = W.get_surface_loss((co, do, to), tb('pc')[:, :, :3], so, 0., 0.9, [True, False, True])
l2 = (is2d * (0.01 * r2 + 1.0 * s2)).mean()
g2 = torch.autograd.grad(l2, [co, do, to])
g72 = torch.cat([g2[0], g2[1], g2[2][:, None]], 1)
print('dbox7 at oracle box - at product box:\n', (g72 - g7).numpy())
print('box diff', float((co - center).abs().max()), float((do - dims).abs().max()), float((to - theta).abs().max()))
# split: reprojection part only
gr_o = torch.autograd.grad((is2d * 0.01 * r2).mean(), [co, do, to]); gr_p = torch.autograd.grad((is2d * 0.01 * r).mean(), [center, dims, theta])
print('reproj part diff', max(float((a - b).abs().max()) for a, b in zip(gr_o, gr_p)), 'surface part diff', float((g72 - g7).abs().max()))
