"""CPU: SEMI_MODEL A with the weak reprojection / surface losses switched on -- the reference's DEFAULT flags
(models/config.py:112-113: WEAK_WEIGHT_REPROJECTION 0.01, WEAK_WEIGHT_SURFACE 1.0; recipe a of its README zeroes both).

On the specification library t3d_weak_loss is the oracle's own restatement, so these tests check the WIRING around it: the weak term
in the loss, its gradient into the box head / T-Net centre through the anchor->reg conversion, into the segmentation net through the
soft mask (second run of the fused seg head), and the builder API feeding the camera placeholders.  The HIP kernels themselves are
compared with the oracle in tests/test_weak_gpu.py and, end to end, in tests/test_model_gpu.py."""
import numpy as np
import torch

from fake_t3d import FakeLib
from model_check import check_against_oracle
from oracle import ref_torch as R
from transferable3d_amd import api, semisup_v1_sunrgbd as MODEL
from transferable3d_amd.config import make_parser
from transferable3d_amd.engine import Runtime
from transferable3d_amd.nets import Graph, SemiModelA
from transferable3d_amd.synthetic import make_batch


def weak_case(B=4, N=128, C=4, seed=3):
    batch = make_batch(B, N, C, seed=seed, dropout_scopes={'inst_seg/dp1': ((B, N, 128), 0.5)})
    batch['is_data_2D'] = np.array(([1, 0, 1, 1] * B)[:B], np.int32)      # weak losses act on the 2-D-label samples only
    return batch


def run_model_a(rt, batch, P, c):
    B, N, C = batch['pc'].shape
    g = Graph(B, N, C, rt=rt)
    m = SemiModelA(g, c)
    g.vars.load_state_dict({k: v.detach().cpu().numpy() for k, v in P.items()})
    g.hyper[2] = 0.5
    m.emit_forward(g.fwd, True, True)
    m.emit_backward(g.bwd)
    g.finalize()
    m.inputs.load(batch)
    g.fwd.run()
    g.bwd.run()
    return g, m


def check_weak_model(rt, c_over, seed=3):
    batch = weak_case(seed=seed)
    P = R.init_params(np.random.RandomState(5), R.layer_table(4, 'A'))
    c = R.default_config(**c_over)
    g, m = run_model_a(rt, batch, P, c)
    res = check_against_oracle(g, m, batch, P, c)
    e = m.end_points()
    _, ep, grads, _ = R.model_a_forward_backward(P, batch, c)
    for mine, ref in (('reprojection_loss', 'reprojection_loss'), ('surface_loss', 'surface_loss')):
        r = ep[ref].detach().numpy()
        assert np.abs(e[mine].detach().cpu().numpy() - r).max() < 2e-4 * max(1.0, np.abs(r).max()), mine
    # the weak term really is in the graph: it moves the loss and sends gradient into the seg net of a 2-D-only sample
    c0 = R.default_config()
    l0, _, g0, _ = R.model_a_forward_backward(P, batch, c0)
    assert abs(float(res['loss'][1]) - float(l0.detach())) > 1e-3 * float(l0.detach())
    # the surface loss reaches the seg net through the soft mask; the reprojection loss only the box / T-Net side
    k = 'inst_seg/conv9/weights' if c.WEAK_WEIGHT_SURFACE != 0 else 'box_est/fc3/weights'
    assert float((grads[k] - g0[k]).abs().max()) > 1e-6 * float(g0[k].abs().max())
    return res


def test_model_a_with_default_weak_weights_matches_oracle():
    check_weak_model(Runtime(lib=FakeLib(), device='cpu'),
                     dict(WEAK_WEIGHT_REPROJECTION=0.01, WEAK_WEIGHT_SURFACE=1.0, SEMI_MULTIPLIER_FOR_WEAK_LOSS=1.0))


def test_model_a_weak_loss_variants_match_oracle():
    rt = Runtime(lib=FakeLib(), device='cpu')
    check_weak_model(rt, dict(WEAK_WEIGHT_REPROJECTION=0.02, WEAK_WEIGHT_SURFACE=0.0, SEMI_MULTIPLIER_FOR_WEAK_LOSS=0.5,
                              WEAK_REPROJECTION_USE_SOFTMAX_PROJ=True, WEAK_REPROJECTION_CLIP_LOWERB_LOSS=False), seed=4)
    check_weak_model(rt, dict(WEAK_WEIGHT_REPROJECTION=0.0, WEAK_WEIGHT_SURFACE=2.0, SEMI_MULTIPLIER_FOR_WEAK_LOSS=0.1,
                              WEAK_TRAIN_BOX_W_SURFACE=[True, True, True], WEAK_SURFACE_MARGIN=0.05), seed=5)


def test_builder_api_with_the_reference_default_flags():
    """train_semisup.py's call sequence with the flag parser's defaults (weak losses ON): the camera placeholders are fed, the loss
    carries the weak term, a training step runs."""
    B, N, C = 4, 128, 4
    FLAGS = make_parser().parse_special_args(['--SEMI_MODEL', 'A'])
    assert FLAGS.WEAK_WEIGHT_REPROJECTION == 0.01 and FLAGS.WEAK_WEIGHT_SURFACE == 1.0
    batch = weak_case(B, N, C, seed=8)
    with api.Graph(rt=Runtime(device='cpu', lib=FakeLib()), seed=3).as_default() as g:
        pls = MODEL.placeholder_inputs(B, N, C)
        pc_pl, bg_pc_pl, img_pl, one_hot_vec_pl, y_seg_pl, y_centers_pl, y_orient_cls_pl, y_orient_reg_pl, y_dims_cls_pl, \
            y_dims_reg_pl, R0_rect_pl, P_pl, Rtilt_pl, K_pl, rot_frust_pl, box2D_pl, img_dim_pl, is_data_2D_pl = pls
        pred, end_points = MODEL.get_semi_model(pc_pl, bg_pc_pl, img_pl, one_hot_vec_pl, True, use_one_hot=False,
                                                norm_box2D=None, bn_decay=None, c=FLAGS)
        labels = (y_seg_pl, y_centers_pl, y_orient_cls_pl, y_orient_reg_pl, y_dims_cls_pl, y_dims_reg_pl, R0_rect_pl, P_pl,
                  Rtilt_pl, K_pl, rot_frust_pl, box2D_pl, img_dim_pl, is_data_2D_pl)
        semi_loss = MODEL.get_semi_loss(pred, labels, end_points, c=FLAGS)
        train_op = api.AdamOptimizer(1e-3).minimize(semi_loss)
        sess = api.Session()
        P0 = {k: torch.tensor(v, dtype=torch.float64) for k, v in g.vars.state_dict().items()}
        feed = {pc_pl: batch['pc'], one_hot_vec_pl: batch['one_hot_vec'], y_seg_pl: batch['y_seg'],
                y_centers_pl: batch['y_center'], y_orient_cls_pl: batch['y_orient_cls'], y_orient_reg_pl: batch['y_orient_reg'],
                y_dims_cls_pl: batch['y_dims_cls'], y_dims_reg_pl: batch['y_dims_reg'], is_data_2D_pl: batch['is_data_2D'],
                Rtilt_pl: batch['Rtilt'], K_pl: batch['K'], rot_frust_pl: batch['rot_frust'], box2D_pl: batch['box2D'],
                img_dim_pl: batch['img_dim'], 'inst_seg/dp1': batch['dropout_masks']['inst_seg/dp1']}
        loss_val, reproj, surf, _ = sess.run([semi_loss, end_points['reproj_loss'], end_points['surface_loss'], train_op], feed_dict=feed)
        P1 = g.vars.state_dict()
    c = R.default_config(WEAK_WEIGHT_REPROJECTION=FLAGS.WEAK_WEIGHT_REPROJECTION, WEAK_WEIGHT_SURFACE=FLAGS.WEAK_WEIGHT_SURFACE,
                         SEMI_MULTIPLIER_FOR_WEAK_LOSS=FLAGS.SEMI_MULTIPLIER_FOR_WEAK_LOSS)
    loss, ep, grads, _ = R.model_a_forward_backward(P0, batch, c, bn_decay_val=R.bn_decay(0, B))
    assert abs(float(loss_val) - float(loss)) < 1e-4 * float(loss)
    assert np.abs(reproj - ep['reprojection_loss'].detach().numpy()).max() < 2e-4 * float(ep['reprojection_loss'].abs().max())
    assert np.abs(surf - ep['surface_loss'].detach().numpy()).max() < 2e-4 * max(1.0, float(ep['surface_loss'].abs().max()))
    moved = max(float(np.abs(P1[k].astype(np.float64).reshape(P0[k].shape) - P0[k].numpy()).max()) for k in R.trainable_names(P0))
    assert moved > 1e-4      # Adam stepped


def test_train_cli_with_the_default_flags_host_fed_and_device_assembled(tmp_path):
    """`train_semisup.py --SEMI_MODEL A` without zeroing the weak weights (the reference's defaults): synthetic batches fed from the
    host, and a device-resident data set whose calibration / 2-D boxes t3d_batch_assemble copies into the batch slots."""
    from transferable3d_amd.train_semisup import build_flags, train
    for i, extra in enumerate((['--synthetic', '--steps_per_epoch', '3'], ['--device_data', '40'])):
        FLAGS = build_flags(['--SEMI_MODEL', 'A', '--num_point', '128', '--batch_size', '4', '--num_channels', '4', '--max_epoch', '1',
                             '--eval_batches', '1', '--log_dir', str(tmp_path / str(i))] + extra)
        assert FLAGS.WEAK_WEIGHT_REPROJECTION == 0.01 and FLAGS.WEAK_WEIGHT_SURFACE == 1.0
        logs = []
        _, last = train(FLAGS, rt=Runtime(device='cpu', lib=FakeLib()), log=logs.append)
        assert np.isfinite(last)
        ev = [l for l in logs if l.startswith('eval mean loss')]
        assert ev and np.isfinite(float(ev[0].split(':')[1]))


def test_stage_c_with_reprojection_and_inactive_volume_matches_oracle():
    """get_semi_loss_final with the reprojection loss of the refined box (the default weight 0.01; all samples, and the
    ..._ONLY_ON_2D_CLS form) and the inactive-volume loss switched on (semisup_v1_sunrgbd.py:345-392)."""
    from test_stage_c_cpu import check_stage_c, run_stage_c, stage_c_batch, stage_c_config, stage_c_params
    B, N, C = 6, 256, 4
    batch = stage_c_batch(B, N, C, seed=2, n2d=3)
    P = stage_c_params(C, 1)
    for over in (dict(WEAK_WEIGHT_REPROJECTION=0.01),
                 dict(WEAK_WEIGHT_REPROJECTION=0.02, WEAK_REPROJECTION_ONLY_ON_2D_CLS=True, WEAK_WEIGHT_INACTIVE_VOLUME=1.0,
                      WEAK_INACTIVE_VOL_LOSS_MARGINS=[10.0, 0.5, 3.0, 0.2, 0.2, 1.0, 0.8, 0.3, 1.2, 0.6])):
        c = stage_c_config()
        for k, v in over.items():
            setattr(c, k, v)
        g, m = run_stage_c(Runtime(device='cpu', lib=FakeLib()), batch, P, c)
        check_stage_c(g, m, batch, P, c)
        c0 = stage_c_config()
        l0 = R.stage_c_forward_backward(P, batch, c0, [i in (1, 2, 6, 7, 8) for i in range(10)], want_grads=False)[0]
        assert abs(float(m.loss) - float(l0)) > 1e-3 * float(l0)      # the weak terms are in the loss


def test_weak_losses_evaluated_for_their_summaries_at_zero_weight():
    """The reference evaluates the reprojection / surface losses for its `Weak_Loss/...` summaries whatever their weights
    (semisup_v1_sunrgbd.py:270-293); recipe a zeroes both weights.  With `c.WEAK_LOSS_SUMMARIES` (the drivers set it with --weak_loss_summaries) the product
    evaluates them too: the values are the oracle's, the loss and EVERY gradient are bit for bit those of the graph without them,
    and no backward launch is added."""
    rt = Runtime(lib=FakeLib(), device='cpu')
    batch = weak_case(seed=6)
    P = R.init_params(np.random.RandomState(5), R.layer_table(4, 'A'))
    c0 = R.default_config(WEAK_WEIGHT_REPROJECTION=0.0, WEAK_WEIGHT_SURFACE=0.0)
    c1 = R.default_config(WEAK_WEIGHT_REPROJECTION=0.0, WEAK_WEIGHT_SURFACE=0.0)
    c1.WEAK_LOSS_SUMMARIES = True
    g0, m0 = run_model_a(rt, batch, P, c0)
    g1, m1 = run_model_a(rt, batch, P, c1)
    assert m0.weak is None and m1.weak is not None
    assert [n for n, _, _ in g1.fwd.calls].count('t3d_weak_loss') == 1 and len(g1.fwd) == len(g0.fwd) + 1
    assert [n for n, _, _ in g1.bwd.calls] == [n for n, _, _ in g0.bwd.calls]            # nothing added to the backward
    assert float(m1.loss_op.loss) == float(m0.loss_op.loss)
    assert torch.equal(g1.vars.grads[:g1.vars.used], g0.vars.grads[:g0.vars.used])
    # the values: the oracle's losses for non-zero weights do not depend on the weights
    cw = R.default_config(WEAK_WEIGHT_REPROJECTION=0.01, WEAK_WEIGHT_SURFACE=1.0)
    _, ep, _, _ = R.model_a_forward_backward(P, batch, cw)
    e = m1.end_points()
    for k in ('reprojection_loss', 'surface_loss'):
        r = ep[k].detach().numpy()
        assert np.abs(e[k].detach().cpu().numpy() - r).max() < 2e-4 * max(1.0, np.abs(r).max()), k
        assert np.abs(r).max() > 0
