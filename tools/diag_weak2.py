import sys
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import numpy as np, torch
from oracle import ref_torch as R
from transferable3d_amd import abi
from transferable3d_amd.engine import Runtime
from test_weak_cpu import weak_case, run_model_a
from model_check import product_decisions, grad_errors
lib = abi.load()
batch = weak_case(seed=3)
P = R.init_params(np.random.RandomState(5), R.layer_table(4, 'A'))
for over in (dict(), dict(WEAK_WEIGHT_REPROJECTION=0.01), dict(WEAK_WEIGHT_SURFACE=1.0), dict(WEAK_WEIGHT_REPROJECTION=0.01, WEAK_WEIGHT_SURFACE=1.0)):
    c = R.default_config(**over)
    g, m = run_model_a(Runtime(lib=lib), batch, P, c)
    torch.cuda.synchronize()
    forced = product_decisions(m)
    loss, ep, grads, ema = R.model_a_forward_backward(P, batch, c, forced=forced)
    per, glob = grad_errors(g, {k: v.numpy() for k, v in grads.items()})
    top = sorted(per.items(), key=lambda kv: -kv[1])[:5]
    print(over, 'loss', float(m.loss_op.loss), float(loss), 'glob', glob, [(k, round(v, 5)) for k, v in top], 'flips', {k: v for k, v in ep['__flips__'].items() if v})
