"""CPU: the multi-rank flow of bench.py that can be exercised without a GPU -- the rank launcher and the watchdogs.  bench.py itself has
no CPU path (the product path fails loudly without an MI355X), so what is checked here is exactly that: `--gpus N` starts N ranks as
child processes, every rank refuses to run without a GPU, the parent hands the failure through with a non-zero code and no JSON line, and
nothing hangs.  (The N > 1 measurement itself is the driver's; no round has had more than one GPU.)"""
import os
import subprocess
import sys
import time

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(torch.cuda.is_available(), reason='a GPU box runs the real thing')
def test_gpus_n_spawns_ranks_that_fail_loudly_without_a_gpu_and_the_parent_hands_the_code_through():
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '1', '--warmup', '0'],
                       capture_output=True, text=True, timeout=300, env=dict(os.environ, OMP_NUM_THREADS='1'))
    assert r.returncode != 0
    assert 'needs an MI355X' in r.stderr                      # every rank's own message, through torch.distributed.run
    assert not any(l.startswith('{') for l in r.stdout.splitlines())      # no JSON line from a run that measured nothing
    assert time.time() - t0 < 280


def test_watchdog_ends_the_process_with_its_code_and_a_message():
    code = ('import sys, time; sys.path.insert(0, %r); import bench\n'
            'with bench.Watchdog(1, "the guarded section", code=7):\n'
            '    time.sleep(30)\n' % ROOT)
    t0 = time.time()
    r = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=120)
    assert r.returncode == 7 and 'the guarded section did not finish within 1 s' in r.stderr and time.time() - t0 < 60


def test_watchdog_on_fire_prints_what_is_known_and_chooses_the_code():
    code = ('import sys, time, os; sys.path.insert(0, %r); import bench\n'
            'def known():\n'
            '    os.write(1, b\'{"metric": "partial", "hang": "leg two"}\\n\'); return 0\n'
            'with bench.Watchdog(1, "the informational leg", known):\n'
            '    time.sleep(30)\n' % ROOT)
    r = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and r.stdout.strip() == '{"metric": "partial", "hang": "leg two"}'
