#!/usr/bin/env python3
"""Gradient-error statistics (tests/model_check.tight_grad_check: per-tensor relative L2 against the flip-aware fp64 oracle) of both
GEMM arithmetics of the fp32 path along teacher-forced trajectories: T3D_X3=0 (fp32 MFMA) vs T3D_X3=1 (three-term bf16, default).
    python tools/x3_traj_errors.py [workload ...]      -> one line per (workload, seed, step, arithmetic)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests'))
sys.path.insert(0, ROOT)
import model_check as MC                                    # noqa: E402
from transferable3d_amd import abi                         # noqa: E402
from transferable3d_amd.engine import Runtime              # noqa: E402

lib = abi.load()
orig, orig_m = MC.tight_grad_check, MC.check_decision_margins
state = {}


def loose(g, ref, per_tol=1e-3, med_tol=5e-5, glob_tol=1e-4, what=''):
    r = orig(g, ref, per_tol=1.0, med_tol=1.0, glob_tol=1.0, what=what)
    state['line'] = '%-14s seed %d T3D_X3=%s median %.2e global %.2e worst %s %.2e' % (
        what, state['seed'], os.environ.get('T3D_X3'), r['grad_median'], r['grad_global'], r['grad_max'][0], r['grad_max'][1])
    return r


def margins(m, **kw):
    w = orig_m(m, **kw)
    print(state['line'], '| flipped decisions:', {k: (v[0], '%.1e' % v[1]) for k, v in w.items()})
    return w


MC.tight_grad_check, MC.check_decision_margins = loose, margins
for wl in (sys.argv[1:] or ['F', 'A', 'boxpc']):
    for seed in (31, 32, 33, 34):
        for mode in ('0', '1'):
            os.environ['T3D_X3'] = mode
            state['seed'] = seed
            MC.trajectory_check(Runtime(lib=lib), wl, steps=5, B=8, N=256, use_hip_graph=True, param_seed=seed)
