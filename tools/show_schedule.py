#!/usr/bin/env python3
"""The rider schedule of the headline step as the step object built it: which GEMM launch hosts which small ops of the other chain
(schedule.overlap_chains report lines), then bench.py --call_detail style per-launch times (eager, per-launch events)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from transferable3d_amd.engine import Runtime
from transferable3d_amd.step import build_training_step
from transferable3d_amd.synthetic import make_batch

rt = Runtime()
B, N, C = 32, 1024, 4
g, model, step, loss = build_training_step(rt, 'A', B, N, C, use_hip_graph=True, inline_dropout=True, dropout_seed=1234, seed=0)
model.inputs.load(make_batch(B, N, C, seed=1234))
for _ in range(5):
    step.run()
torch.cuda.synchronize()
rep = step.schedule_report
print({k: v for k, v in rep.items() if k != 'lines'})
for line in rep['lines']:
    print(line)
