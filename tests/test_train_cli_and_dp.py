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


def _mean_gradient_adam_reference(world):
    """(state after one TF-form Adam step on the MEAN of the replicas' step-0 gradients, the last graph): what any world size must give."""
    from oracle import ref_torch as R
    from transferable3d_amd.nets import Graph, SemiModelA
    from transferable3d_amd.synthetic import make_batch
    rt = Runtime(device='cpu', lib=FakeLib())
    acc, w0, g = None, None, None
    for rank in range(world):
        g = Graph(4, 128, 4, rt=rt, seed=0)
        g.inline_dropout, g.dropout_seed = True, 1234 + rank
        m = SemiModelA(g, R.default_config())
        m.emit_forward(g.fwd, True, True)
        m.emit_backward(g.bwd)
        g.finalize()
        g.hyper[0] = 1.0
        m.inputs.load(make_batch(4, 128, 4, seed=0 * 1000003 + 0 * world + rank))      # the driver's slice of step 0: step * world + rank
        g.fwd.run()
        g.bwd.run()
        gr = g.vars.grads[:g.vars.used].double().numpy()
        acc = gr if acc is None else acc + gr
        w0 = g.vars.params[:g.vars.used].double().numpy()
    mean_grad = acc / world
    b1, b2, eps, lr = 0.9, 0.999, 1e-8, 1e-3
    lr_t = lr * np.sqrt(1 - b2) / (1 - b1)
    return w0 - lr_t * ((1 - b1) * mean_grad) / (np.sqrt((1 - b2) * mean_grad ** 2) + eps), mean_grad, g


def _check_world(tmp_path, world):
    res = _run_dp(str(tmp_path / ('w%d' % world)), world, flat=True)
    assert sorted(res) == list(range(world))
    for r in range(1, world):              # every replica holds the same weights, bit for bit
        for k in res[0]:
            assert np.array_equal(res[0][k], res[r][k]), (r, k)
    # rank 0 alone writes the checkpoint (train_semisup.py:316-318 runs on one GPU; here: one writer)
    assert os.path.exists(os.path.join(str(tmp_path / ('w%d' % world)), 'r0', 'model_epoch_0.npz'))
    for r in range(1, world):
        assert not os.path.exists(os.path.join(str(tmp_path / ('w%d' % world)), 'r%d' % r, 'model_epoch_0.npz')), r
    # == one TF-form Adam step on the mean of the `world` replicas' gradients (grad_scale = 1 / world), each replica on its own slice
    w1, mean_grad, g = _mean_gradient_adam_reference(world)
    checked = 0
    for k, got in res[0].items():
        o, n = g.vars.offset(k), got.size
        gk = np.abs(mean_grad[o:o + n])
        sel = gk > 1e-2 * gk.max()
        if sel.sum() == 0:
            continue
        assert np.abs(got.reshape(-1).astype(np.float64) - w1[o:o + n])[sel].max() < 2e-5, k
        checked += int(sel.sum())
    assert checked > 10000


def test_data_parallel_world_size_4_gloo(tmp_path):
    """Four replicas (gloo, the NumPy specification library): identical weights on every rank, equal to the TF-form Adam step on the mean
    of the four replicas' gradients; disjoint data slices (step * world + rank); one checkpoint writer.  Unmeasured on hardware: no round
    has had more than one GPU."""
    _check_world(tmp_path, 4)


def test_data_parallel_world_size_8_gloo(tmp_path):
    """... and eight, the node size BASELINE.json's metric is quoted at."""
    _check_world(tmp_path, 8)


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


# ---- stage c (SEMI_MODEL F, BASELINE configs[3] = the 8-GPU config) under data parallelism -----------------------------------------
STAGE_C_ARGS = ['--SEMI_MODEL', 'F', '--BOX_PC_MASK_REPRESENTATION', 'A', '--use_one_hot', '--SEMI_TRAIN_BOX_TRAIN_CLASS_AG_TNET', '1',
                '--SEMI_TRAIN_BOX_TRAIN_CLASS_AG_BOX', '1', '--SEMI_BOXPC_FIT_ONLY_ON_2D_CLS', '1', '--WEAK_WEIGHT_INTRACLASSVAR', '2',
                '--WEAK_WEIGHT_REPROJECTION', '0', '--SEMI_MULTIPLIER_FOR_WEAK_LOSS', '0.05', '--SUNRGBD_SEMI_TEST_CLS', 'table', 'sofa',
                'dresser', 'night_stand', 'bookshelf', '--num_point', '128', '--batch_size', '4', '--num_channels', '4', '--max_epoch', '1',
                '--steps_per_epoch', '1', '--synthetic']


def _stage_c_worker(rank, world, port, tmp, q, same_batches):
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    torch.set_num_threads(2)
    from fake_t3d import FakeLib as FL
    from transferable3d_amd import api, train_semisup_adv as T
    if same_batches:
        # every replica sees rank 0's batches and dropout draws: the mean of identical gradients is that gradient, exactly
        orig_mb, orig_sess, orig_rs = T.make_batch, api.Session, np.random.RandomState
        T.make_batch = lambda B, N, C, seed, **kw: orig_mb(B, N, C, seed=(seed - rank) // world, **kw)      # (--seed 0: seed = step * world + rank)
        api.Session = lambda *a, **kw: orig_sess(*a, **dict(kw, dropout_seed=1234))

        class _NP:      # train(): `np.random.RandomState(step * world + rank)` picks the classes of the ALTERNATE_BATCH batches
            def __getattr__(self, name):
                return getattr(np, name)

            class random:
                RandomState = staticmethod(lambda seed: orig_rs((seed - rank) // world))
        T.np = _NP()
    sd, _ = T.train(T.build_flags(STAGE_C_ARGS + ['--log_dir', os.path.join(tmp, 'r%d' % rank)]), rt=Runtime(device='cpu', lib=FL()),
                    log=lambda *_: None)
    q.put((rank, sd))


def _run_stage_c(tmp, world, same_batches):
    port = _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_stage_c_worker, args=(r, world, port, tmp, q, same_batches)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=600) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return res


def test_stage_c_data_parallel_world_size_2_gloo(tmp_path):
    """train_semisup_adv on two ranks (gloo): the var_list range (train_semisup_adv.py:415-422) is one gradient bucket; the replicas
    stay bit-identical, the frozen branches (class-agnostic seg net, D_boxpc_branch) are untouched, and with identical batches on
    both ranks the run reproduces the single-replica run bit for bit (sum of two equal gradients x 1/2 is exact)."""
    from transferable3d_amd import train_semisup_adv as T
    world = 2
    trained = lambda k: k.startswith(('class_dependent', 'class_agnostic/tnet', 'class_agnostic/box'))
    learned = lambda k: trained(k) and k.endswith(('weights', 'biases', 'gamma', 'beta'))
    init, _ = T.train(T.build_flags(STAGE_C_ARGS[:-5] + ['--max_epoch', '0', '--steps_per_epoch', '1', '--synthetic', '--log_dir',
                                                         str(tmp_path / 'init')]), rt=Runtime(device='cpu', lib=FakeLib()), log=lambda *_: None)
    single, _ = T.train(T.build_flags(STAGE_C_ARGS + ['--log_dir', str(tmp_path / 'single')]), rt=Runtime(device='cpu', lib=FakeLib()),
                        log=lambda *_: None)
    res = _run_stage_c(str(tmp_path / 'dp'), world, same_batches=False)
    n_frozen = n_moved = 0
    for k in res[0]:
        assert np.array_equal(res[0][k], res[1][k]) or not learned(k) and 'moving_' in k, k      # (moving statistics are per replica)
        if not trained(k) and not k.endswith(('moving_mean', 'moving_variance')):
            assert np.array_equal(res[0][k], init[k]), k                                           # frozen: bit-unchanged
            n_frozen += 1
        elif learned(k) and res[0][k].size > 8 and not np.array_equal(res[0][k], init[k]):
            n_moved += 1
    assert n_frozen > 40 and n_moved > 20, (n_frozen, n_moved)
    assert any(learned(k) and not np.array_equal(res[0][k], single[k]) for k in res[0])          # other batches than the single run's
    # identical batches and dropout draws on both ranks == the single-replica run, bit for bit
    same = _run_stage_c(str(tmp_path / 'dp_same'), world, same_batches=True)
    for r in range(world):
        for k in single:
            assert np.array_equal(same[r][k], single[k]), (r, k)


def test_momentum_optimizer_follows_the_tf_update(tmp_path):
    """--optimizer momentum (train_semisup.py:226-228): accum = momentum * accum + g; w -= lr * accum, checked on two consecutive
    steps of the compiled training step against the gradients the step itself left in the gradient buffer (oracle.momentum_tf_step)."""
    from oracle import ref_torch as R
    from transferable3d_amd import api, semisup_v1_sunrgbd as M
    from transferable3d_amd.step import workload_flags
    from transferable3d_amd.synthetic import make_batch
    B, N, C = 4, 128, 4
    c = workload_flags('A')
    c.learning_rate, c.momentum, c.optimizer, c.decay_step, c.decay_rate = 0.01, 0.9, 'momentum', 800000, 0.5
    with api.Graph(rt=Runtime(device='cpu', lib=FakeLib()), seed=1).as_default() as g:
        pls = M.placeholder_inputs(B, N, C)
        pred, ep = M.get_semi_model(pls[0], pls[1], pls[2], pls[3], True, use_one_hot=False, c=c)
        loss = M.get_semi_loss(pred, pls[4:], ep, c=c)
        opt = api.make_optimizer(c)
        assert isinstance(opt, api.MomentumOptimizer)
        train_op = opt.minimize(loss)
        sess = api.Session()
        vs = g.vars
        names = [k for k, (off, shape, tr) in vs.index.items() if tr]
        snap = lambda buf: {k: buf[vs.index[k][0]:vs.index[k][0] + int(np.prod(vs.index[k][1]))].clone().double() for k in names}
        P, acc = snap(vs.params), {k: torch.zeros_like(v) for k, v in snap(vs.params).items()}
        for step in range(2):
            batch = make_batch(B, N, C, seed=5 + step)
            sess.run([loss, train_op], feed_dict={pl: batch[pl.field] for pl in pls if getattr(pl, 'field', None) in batch})
            R.momentum_tf_step(P, snap(vs.grads), acc, 0.01, 0.9)
            got, slot = snap(vs.params), snap(vs.adam_m)
            for k in names:
                scale = max(1.0, float(P[k].abs().max()))
                assert float((got[k] - P[k]).abs().max()) < 2e-6 * scale, (step, k)
                assert float((slot[k] - acc[k]).abs().max()) < 2e-6 * max(1.0, float(acc[k].abs().max())), (step, k)
        assert any(float(acc[k].abs().max()) > 0 for k in names)
    # the command line takes the flag, and the checkpoint carries TF's slot name
    FLAGS = build_flags(ARGS[:-5] + ['--max_epoch', '1', '--steps_per_epoch', '2', '--synthetic', '--optimizer', 'momentum', '--ckpt_format', 'tf',
                                     '--log_dir', str(tmp_path)])
    train(FLAGS, rt=Runtime(device='cpu', lib=FakeLib()), log=lambda *_: None)
    from transferable3d_amd.tf_checkpoint import load_state
    sd = load_state(os.path.join(str(tmp_path), 'model_epoch_0.ckpt'))
    assert 'inst_seg/conv1/weights/Momentum' in sd and 'inst_seg/conv1/weights/Adam' not in sd and 'beta1_power' not in sd
    assert np.abs(sd['box_est/fc3/weights/Momentum']).max() > 0
