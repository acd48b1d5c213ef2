#!/usr/bin/env python3
"""Where the bf16 path's error comes from: SEMI_MODEL A on the same batch and weights in fp32 and in bf16 (same library), relative
L2 / max error of every stored layer output, the heads and the gradients.  Run on the GPU box."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests')]
from oracle import ref_torch as R                                  # noqa: E402  (diagnosis tool: same status as tests/)
from transferable3d_amd.engine import PointLayer, Runtime           # noqa: E402
from transferable3d_amd.nets import Graph, SemiModelA               # noqa: E402
from transferable3d_amd.synthetic import make_batch                 # noqa: E402


def run(rt, batch, P, c, dtype):
    B, N, C = batch['pc'].shape
    g = Graph(B, N, C, rt=rt, dtype=dtype)
    m = SemiModelA(g, c)
    g.vars.load_state_dict({k: v.detach().cpu().numpy() for k, v in P.items()})
    m.emit_forward(g.fwd, True, True)
    m.emit_backward(g.bwd)
    g.finalize()
    m.inputs.load(batch)
    g.fwd.run()
    g.bwd.run()
    torch.cuda.synchronize()
    return g, m


def layers(m):
    out = []
    for net in (m.seg, m.tnet, m.box):
        for v in vars(net).values():
            if isinstance(v, PointLayer) and v.y is not None:
                out.append(v)
    return out


def main():
    B, N, C = int(sys.argv[1]) if len(sys.argv) > 1 else 8, int(sys.argv[2]) if len(sys.argv) > 2 else 512, 4
    batch = make_batch(B, N, C, seed=2, dropout_scopes={'inst_seg/dp1': ((B, N, 128), 0.5)})
    P = R.init_params(np.random.RandomState(9), R.layer_table(C, 'A'))
    c = R.default_config()
    rt = Runtime()
    g32, m32 = run(rt, batch, P, c, 'f32')
    g16, m16 = run(rt, batch, P, c, 'bf16')
    rel = lambda a, b: float((a.float() - b.float()).norm() / b.float().norm())
    for l32, l16 in zip(layers(m32), layers(m16)):
        z32, z16 = l32.y.float() * l32.scale + l32.shift, l16.y.float() * l16.scale + l16.shift
        print('%-32s y rel L2 %.2e   bn(y) rel L2 %.2e   dz rel L2 %s' % (
            l32.scope, rel(l16.y, l32.y), rel(z16, z32), '%.2e' % rel(l16.dz, l32.dz) if l32.dz is not None else '-'))
    e32, e16 = m32.end_points(), m16.end_points()
    for k in ('logits', 'stage1_center', 'center', 'box_params', 'feats_lv1', 'loss'):
        a, b = e16[k].float(), e32[k].float()
        print('%-14s rel L2 %.2e   max abs %.3e   (|ref| max %.3f, rms %.3f)' % (k, rel(a, b), float((a - b).abs().max()),
                                                                               float(b.abs().max()), float(b.pow(2).mean().sqrt())))
    n = g32.vars.used
    print('all gradients rel L2 %.2e' % rel(g16.vars.grads[:n], g32.vars.grads[:n]))
    worst = []
    for name, (off, shape, tr) in g32.vars.index.items():
        if tr:
            k = int(np.prod(shape))
            a, b = g16.vars.grads[off:off + k], g32.vars.grads[off:off + k]
            if float(b.norm()) > 1e-9:
                worst.append((rel(a, b), name))
    worst.sort(reverse=True)
    print('per-tensor gradient rel L2: worst', worst[:5], 'median %.2e' % float(np.median([w[0] for w in worst])))


if __name__ == '__main__':
    main()
