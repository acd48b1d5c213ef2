"""CPU: the train_semisup command line runs end to end on synthetic frustums (spec library), the loss goes down,
and the data-parallel path (world_size 2, gloo) produces the mean of the two replicas' gradients and keeps the
replicas' weights identical."""
import os
import socket
import sys

import numpy as np
import torch
import torch.multiprocessing as mp

from fake_t3d import FakeLib
from transferable3d_amd.engine import Runtime
from transferable3d_amd.train_semisup import build_flags, train

ARGS = ['--SEMI_MODEL', 'A', '--WEAK_WEIGHT_REPROJECTION', '0', '--WEAK_WEIGHT_SURFACE', '0', '--num_point', '128',
        '--batch_size', '4', '--num_channels', '4', '--max_epoch', '2', '--steps_per_epoch', '6', '--synthetic']


def test_cli_trains_and_loss_decreases(tmp_path):
    # every step sees a fresh synthetic batch (and fresh dropout draws): a learning rate that moves the loss well beyond that noise
    FLAGS = build_flags(ARGS + ['--log_dir', str(tmp_path), '--learning_rate', '0.005'])
    logs = []
    _, last = train(FLAGS, rt=Runtime(device='cpu', lib=FakeLib()), log=logs.append)
    epochs = [l for l in logs if l.startswith('**** EPOCH')]
    first = float(epochs[0].split('mean loss: ')[1].split()[0])
    assert len(epochs) == 2 and last < first, (first, last)
    assert os.path.exists(os.path.join(str(tmp_path), 'model_epoch_0.npz'))
    sd = np.load(os.path.join(str(tmp_path), 'model_epoch_0.npz'))
    assert 'inst_seg/conv1/weights' in sd.files and sd['inst_seg/conv1/weights'].shape == (1, 4, 1, 64)
    assert 'box_est/fc3/biases' in sd.files and 'tnet/conv-reg1-stage1/bn/moving_variance' in sd.files


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _dp_worker(rank, world, port, tmp, q, flat=False):
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR='127.0.0.1',
                      MASTER_PORT=str(port), T3D_DP_FLAT='1' if flat else '0')
    torch.set_num_threads(2)
    from fake_t3d import FakeLib as FL
    FLAGS = build_flags(ARGS[:-5] + ['--max_epoch', '1', '--steps_per_epoch', '1', '--synthetic', '--log_dir',
                                     os.path.join(tmp, 'r%d' % rank)])
    sd, _ = train(FLAGS, rt=Runtime(device='cpu', lib=FL()), log=lambda *_: None)
    q.put((rank, {k: v for k, v in sd.items() if k.endswith('weights') or k.endswith('gamma')}))


def _run_dp(tmp, world, flat):
    port = _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_dp_worker, args=(r, world, port, tmp, q, flat)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return res


def test_data_parallel_world_size_2_gloo(tmp_path):
    world = 2
    res = _run_dp(str(tmp_path / 'bucketed'), world, flat=False)
    # replicas stay bit-identical (same init, same all-reduced gradients, same Adam)
    for k in res[0]:
        assert np.array_equal(res[0][k], res[1][k]), k
    # three gradient buckets reduced beside the backward (box + T-Net | seg conv10..6 | seg conv5..1), Adam per bucket
    # == ONE flat all-reduce between the backward and one Adam launch, bit for bit
    flat = _run_dp(str(tmp_path / 'flat'), world, flat=True)
    for k in res[0]:
        assert np.array_equal(res[0][k], flat[0][k]), k

    # reference: average of the two replicas' single-process gradients at step 0
    from model_check import run_model_a
    from oracle import ref_torch as R
    from transferable3d_amd.nets import Graph, SemiModelA
    from transferable3d_amd.synthetic import make_batch
    rt = Runtime(device='cpu', lib=FakeLib())
    grads = []
    for rank in range(world):
        g = Graph(4, 128, 4, rt=rt, seed=0)
        g.inline_dropout, g.dropout_seed = True, 1234 + rank       # as the driver: the seg head draws its own mask from (seed, step)
        m = SemiModelA(g, R.default_config())
        m.emit_forward(g.fwd, True, True)
        m.emit_backward(g.bwd)
        g.finalize()
        g.hyper[0] = 1.0                     # masks are drawn after the schedule kernel bumped the step to 1
        m.inputs.load(make_batch(4, 128, 4, seed=0 * 1000003 + 0 * world + rank))
        g.fwd.run()
        g.bwd.run()
        grads.append(g.vars.grads[:g.vars.used].clone())
        w0 = g.vars.params[:g.vars.used].clone()
        off = g.vars.offset('box_est/fc3/weights')
    mean_grad = ((grads[0] + grads[1]) / world).double().numpy()
    # one TF-form Adam step from the common initial weights with the MEAN gradient (what the all-reduce must produce)
    b1, b2, eps, lr = 0.9, 0.999, 1e-8, 1e-3
    lr_t = lr * np.sqrt(1 - b2) / (1 - b1)
    w1 = w0.double().numpy() - lr_t * ((1 - b1) * mean_grad) / (np.sqrt((1 - b2) * mean_grad ** 2) + eps)
    checked = 0
    for k, got in res[0].items():
        o = g.vars.offset(k)
        n = got.size
        gk = np.abs(mean_grad[o:o + n])
        sel = gk > 1e-2 * gk.max()           # Adam's first step is ~lr*sign(g): compare where g is above fp32 noise
        if sel.sum() == 0:
            continue
        assert np.abs(got.reshape(-1).astype(np.float64) - w1[o:o + n])[sel].max() < 2e-5, k
        checked += int(sel.sum())
    assert checked > 10000


def test_three_stage_recipe_on_synthetic_data(tmp_path):
    """README.md:58-99 in miniature: stage a -> stage b -> stage c restoring both checkpoints by scope prefix."""
    from transferable3d_amd import train_boxpc, train_semisup_adv
    rt = lambda: Runtime(device='cpu', lib=FakeLib())
    small = ['--num_point', '128', '--batch_size', '4', '--num_channels', '4', '--max_epoch', '1', '--steps_per_epoch', '2', '--synthetic']
    a_dir, b_dir, c_dir = [str(tmp_path / d) for d in 'abc']
    sd_a, _ = train(build_flags(['--SEMI_MODEL', 'A', '--WEAK_WEIGHT_REPROJECTION', '0', '--WEAK_WEIGHT_SURFACE', '0', '--log_dir', a_dir] + small),
                    rt=rt(), log=lambda *_: None)
    sd_b, _ = train_boxpc.train(train_boxpc.build_flags(['--BOX_PC_MASK_REPRESENTATION', 'A', '--BOXPC_WEIGHT_DELTA', '4', '--log_dir', b_dir] + small),
                                rt=rt(), log=lambda *_: None)
    logs = []
    flags_c = train_semisup_adv.build_flags(
        ['--SEMI_MODEL', 'F', '--BOX_PC_MASK_REPRESENTATION', 'A', '--use_one_hot', '--SEMI_TRAIN_BOX_TRAIN_CLASS_AG_TNET', '1',
         '--SEMI_TRAIN_BOX_TRAIN_CLASS_AG_BOX', '1', '--SEMI_BOXPC_FIT_ONLY_ON_2D_CLS', '1', '--WEAK_WEIGHT_INTRACLASSVAR', '2',
         '--WEAK_WEIGHT_REPROJECTION', '0', '--SEMI_MULTIPLIER_FOR_WEAK_LOSS', '0.05', '--SUNRGBD_SEMI_TEST_CLS', 'table', 'sofa', 'dresser',
         'night_stand', 'bookshelf', '--init_class_ag_path', os.path.join(a_dir, 'model_epoch_0.npz'), '--init_boxpc_path',
         os.path.join(b_dir, 'model_epoch_0.npz'), '--log_dir', c_dir] + small)
    sd_c, loss_c = train_semisup_adv.train(flags_c, rt=rt(), log=logs.append)
    # every tensor of both checkpoints (weights, biases, BN beta/gamma and moving statistics) found its prefixed twin
    assert 'restored %d class_agnostic variables' % len(sd_a) in logs, logs[:3]
    assert 'restored %d D_boxpc_branch variables' % len(sd_b) in logs, logs[:3]
    assert np.isfinite(loss_c)
    # the frozen branches still equal the checkpoints they were restored from
    assert np.array_equal(sd_c['D_boxpc_branch/box_pc_mask_model/fc1/weights'], sd_b['box_pc_mask_model/fc1/weights'])
    assert np.array_equal(sd_c['class_agnostic/inst_seg/conv4/weights'], sd_a['inst_seg/conv4/weights'])
    assert not np.array_equal(sd_c['class_agnostic/tnet/fc1-stage1/weights'], sd_a['tnet/fc1-stage1/weights'])


def test_cli_trains_from_a_device_resident_dataset(tmp_path):
    """--device_data: nothing is fed, every batch is assembled by t3d_batch_assemble inside the step."""
    logs = []
    flags = build_flags(['--SEMI_MODEL', 'A', '--WEAK_WEIGHT_REPROJECTION', '0', '--WEAK_WEIGHT_SURFACE', '0', '--num_point', '128',
                         '--batch_size', '4', '--num_channels', '4', '--max_epoch', '2', '--steps_per_epoch', '12', '--device_data', '24',
                         '--log_dir', str(tmp_path)])
    _, loss = train(flags, rt=Runtime(device='cpu', lib=FakeLib()), log=logs.append)
    ep = [l for l in logs if 'EPOCH' in l]
    assert len(ep) == 2 and 'assembled on the device' in ep[0]
    l0, l1 = [float(l.split('mean loss: ')[1].split()[0]) for l in ep]
    assert np.isfinite(l1) and l1 < l0


def test_stage_c_cli_from_a_device_resident_dataset_alternates_weak_and_strong_batches(tmp_path):
    from transferable3d_amd import train_semisup_adv
    logs = []
    flags = train_semisup_adv.build_flags(
        ['--SEMI_MODEL', 'F', '--BOX_PC_MASK_REPRESENTATION', 'A', '--use_one_hot', '--SEMI_TRAIN_BOX_TRAIN_CLASS_AG_TNET', '1',
         '--SEMI_TRAIN_BOX_TRAIN_CLASS_AG_BOX', '1', '--SEMI_BOXPC_FIT_ONLY_ON_2D_CLS', '1', '--WEAK_WEIGHT_INTRACLASSVAR', '2',
         '--WEAK_WEIGHT_REPROJECTION', '0', '--SEMI_MULTIPLIER_FOR_WEAK_LOSS', '0.05', '--SUNRGBD_SEMI_TEST_CLS', 'table', 'sofa', 'dresser',
         'night_stand', 'bookshelf', '--num_point', '128', '--batch_size', '4', '--num_channels', '4', '--max_epoch', '1',
         '--steps_per_epoch', '4', '--device_data', '40', '--log_dir', str(tmp_path)])
    sd, loss = train_semisup_adv.train(flags, rt=Runtime(device='cpu', lib=FakeLib()), log=logs.append)
    assert np.isfinite(loss) and any('assembled on the device' in l for l in logs)


def test_train_boxpc_from_a_device_resident_dataset(tmp_path):
    """train_boxpc --device_data: t3d_batch_assemble + t3d_boxpc_perturb make every (points, perturbed box, IoU / delta targets)
    sample inside the step (the sampling law itself is checked in test_dataset_cpu.py / test_dataset_gpu.py)."""
    from transferable3d_amd import train_boxpc
    logs = []
    flags = train_boxpc.build_flags(['--BOX_PC_MASK_REPRESENTATION', 'A', '--BOXPC_WEIGHT_DELTA', '4', '--num_point', '128', '--batch_size', '8',
                                     '--num_channels', '4', '--max_epoch', '2', '--steps_per_epoch', '15', '--device_data', '32',
                                     '--log_dir', str(tmp_path)])
    _, loss = train_boxpc.train(flags, rt=Runtime(device='cpu', lib=FakeLib()), log=logs.append)
    ep = [l for l in logs if 'EPOCH' in l]
    assert len(ep) == 2 and 'samples made on the device' in ep[0]
    l0, l1 = [float(l.split('mean loss: ')[1].split()[0]) for l in ep]
    assert np.isfinite(l0) and np.isfinite(l1) and 0 < l1 < 20


def test_cli_evaluates_after_every_epoch_in_the_training_graph(tmp_path):
    """--eval_batches: eval_one_epoch (train_semisup.py:436-545) on held-out frustums with is_training fed False."""
    logs = []
    flags = build_flags(['--SEMI_MODEL', 'A', '--WEAK_WEIGHT_REPROJECTION', '0', '--WEAK_WEIGHT_SURFACE', '0', '--num_point', '128',
                         '--batch_size', '4', '--num_channels', '4', '--max_epoch', '2', '--steps_per_epoch', '3', '--synthetic',
                         '--eval_batches', '2', '--log_dir', str(tmp_path)])
    train(flags, rt=Runtime(device='cpu', lib=FakeLib()), log=logs.append)
    ev = [l for l in logs if str(l).startswith('eval ')]
    assert len(ev) == 10 and sum('EVALUATION' in str(l) for l in logs) == 2
    acc = [float(l.split(': ')[1]) for l in ev if l.startswith('eval accuracy')]
    assert all(0.0 <= a <= 1.0 for a in acc) and all(np.isfinite(float(l.split(': ')[1].split()[0])) for l in ev if 'mean loss' in l)


def test_cli_evaluates_on_device_assembled_held_out_frustums(tmp_path):
    logs = []
    flags = build_flags(['--SEMI_MODEL', 'A', '--WEAK_WEIGHT_REPROJECTION', '0', '--WEAK_WEIGHT_SURFACE', '0', '--num_point', '128',
                         '--batch_size', '4', '--num_channels', '4', '--max_epoch', '1', '--steps_per_epoch', '4', '--device_data', '16',
                         '--eval_batches', '2', '--log_dir', str(tmp_path)])
    train(flags, rt=Runtime(device='cpu', lib=FakeLib()), log=logs.append)
    ev = {l.split(':')[0]: l for l in logs if str(l).startswith('eval ')}
    assert sorted(ev) == ['eval accuracy', 'eval avg class acc', 'eval box IoU (ground/3D)     ', 'eval mIoU', 'eval mean loss']
    assert np.isfinite(float(ev['eval mean loss'].split(': ')[1]))


def test_train_boxpc_reports_the_reference_statistics_and_evaluates(tmp_path):
    from transferable3d_amd import train_boxpc
    logs = []
    flags = train_boxpc.build_flags(['--BOX_PC_MASK_REPRESENTATION', 'A', '--BOXPC_WEIGHT_DELTA', '4', '--num_point', '128', '--batch_size', '8',
                                     '--num_channels', '4', '--max_epoch', '1', '--steps_per_epoch', '10', '--device_data', '32',
                                     '--eval_batches', '2', '--log_dir', str(tmp_path)])
    train_boxpc.train(flags, rt=Runtime(device='cpu', lib=FakeLib()), log=logs.append)
    text = '\n'.join(str(l) for l in logs)
    assert text.count('Classname   Prec Recall  F1   Supp') == 2 and text.count('Before After ') == 4      # train + eval reports
    assert 'EVALUATION' in text and 'eval mean loss' in text


def test_stage_c_cli_evaluates_intermediate_and_refined_boxes(tmp_path):
    from transferable3d_amd import train_semisup_adv
    logs = []
    flags = train_semisup_adv.build_flags(
        ['--SEMI_MODEL', 'F', '--BOX_PC_MASK_REPRESENTATION', 'A', '--use_one_hot', '--SEMI_BOXPC_FIT_ONLY_ON_2D_CLS', '1',
         '--WEAK_WEIGHT_INTRACLASSVAR', '2', '--WEAK_WEIGHT_REPROJECTION', '0', '--SEMI_MULTIPLIER_FOR_WEAK_LOSS', '0.05', '--num_point', '128',
         '--batch_size', '4', '--num_channels', '4', '--max_epoch', '1', '--steps_per_epoch', '2', '--device_data', '24', '--eval_batches', '2',
         '--log_dir', str(tmp_path)])
    train_semisup_adv.train(flags, rt=Runtime(device='cpu', lib=FakeLib()), log=logs.append)
    text = '\n'.join(str(l) for l in logs)
    assert 'intermediate (F_)' in text and 'refined by the Box-PC deltas (F2_)' in text and text.count('Mean AP:') == 2
    assert 'class-agnostic heads' in text
