"""CPU (NumPy specification library): the branches of the reference that no published recipe switches on but the flag parser offers
-- USE_NORMALIZED_BOX2D_AS_FEATS (semisup_models.py:194-195, 251-252; semisup_v1_sunrgbd.py:97,145), checked as whole training steps
against the oracle with the flag set on both sides."""
import numpy as np
import pytest

from fake_t3d import FakeLib
from model_check import trajectory_check
from transferable3d_amd.engine import Runtime


SHAPE = (4, 128)          # (B, N) of the trajectory tests; the GPU module runs them at (8, 256)


def _runtime():
    """tests/test_off_recipe_gpu.py replaces this factory with the HIP library."""
    return Runtime(device='cpu', lib=FakeLib())


@pytest.mark.parametrize('workload', ['A', 'F'])
def test_norm_box2d_features_follow_the_oracle(workload):
    rep = trajectory_check(_runtime(), workload, steps=2, B=SHAPE[0], N=SHAPE[1],
                           config_over={'USE_NORMALIZED_BOX2D_AS_FEATS': True})
    assert all(r['weight_entries_checked'] > 1000 for r in rep[:-1])


def test_norm_box2d_changes_the_graph():
    """The four columns are really read: the T-Net / box net FC1 weights grow by 4 rows and the heads move when box2D moves."""
    from transferable3d_amd.step import build_training_step, workload_flags
    from transferable3d_amd.synthetic import make_batch
    rt = _runtime()
    flags = workload_flags('A')
    flags.USE_NORMALIZED_BOX2D_AS_FEATS = True
    g, model, step, loss = build_training_step(rt, 'A', 4, 128, 4, c=flags, use_hip_graph=False)
    sd = g.vars.state_dict()
    assert sd['tnet/fc1-stage1/weights'].shape[0] == 256 + 4 and sd['box_est/fc1/weights'].shape[0] == 512 + 4
    b = make_batch(4, 128, 4, seed=3)
    model.inputs.load(b)
    model.emit_forward  # noqa: B018  (the plans were emitted by build_training_step)
    g.fwd.run()
    s1 = model.tnet.stage1_center.detach().cpu().numpy().copy()
    b2 = dict(b)
    b2['box2D'] = b['box2D'] * 0.5
    model.inputs.load(b2)
    g.fwd.run()
    assert np.abs(model.tnet.stage1_center.detach().cpu().numpy() - s1).max() > 1e-6


def test_tf_normalize_2D_bboxes_handle_and_host_form():
    from transferable3d_amd import api, tf_util, semisup_v1_sunrgbd as MODEL
    with api.Graph(rt=_runtime()).as_default():
        pls = MODEL.placeholder_inputs(4, 128, 4)
        h = tf_util.tf_normalize_2D_bboxes(pls[15], pls[16])
        assert tuple(h.shape) == (4, 4)
    box = np.array([[10., 20., 110., 220.]], np.float32)
    dim = np.array([[530., 730.]], np.float32)
    np.testing.assert_allclose(tf_util.tf_normalize_2D_bboxes(box, dim), [[10 / 730., 20 / 530., 110 / 730., 220 / 530.]], rtol=1e-6)


def test_reference_call_sequence_with_norm_box2d_and_one_hot():
    """train_semisup.py:208-249 as the reference writes it, USE_NORMALIZED_BOX2D_AS_FEATS on and use_one_hot on: FC1 of the T-Net /
    box net reads [pooled | one_hot | norm_box2D]; forward heads and loss against the oracle with the same variables."""
    import torch
    from oracle import ref_torch as R
    from model_check import product_decisions
    from transferable3d_amd import api, tf_util, semisup_v1_sunrgbd as MODEL
    from transferable3d_amd.config import make_parser
    from transferable3d_amd.synthetic import make_batch
    B, N, C = 4, 128, 4
    FLAGS = make_parser().parse_special_args(['--SEMI_MODEL', 'A', '--WEAK_WEIGHT_REPROJECTION', '0', '--WEAK_WEIGHT_SURFACE', '0',
                                              '--USE_NORMALIZED_BOX2D_AS_FEATS', '1'])
    batch = make_batch(B, N, C, seed=8, dropout_scopes={'inst_seg/dp1': ((B, N, 128), 0.5)})
    with api.Graph(rt=_runtime(), seed=3).as_default() as g:
        pls = MODEL.placeholder_inputs(B, N, C)
        norm_box2D = tf_util.tf_normalize_2D_bboxes(pls[15], pls[16])
        pred, end_points = MODEL.get_semi_model(pls[0], pls[1], pls[2], pls[3], True, use_one_hot=True, norm_box2D=norm_box2D,
                                                bn_decay=None, c=FLAGS)
        labels = tuple(pls[4:10]) + tuple(pls[10:])
        semi_loss = MODEL.get_semi_loss(pred, labels, end_points, c=FLAGS)
        sess = api.Session()
        P0 = {k: torch.tensor(v, dtype=torch.float64) for k, v in g.vars.state_dict().items()}
        assert tuple(P0['tnet/fc1-stage1/weights'].shape) == (256 + 10 + 4, 256)
        assert tuple(P0['box_est/fc1/weights'].shape) == (512 + 10 + 4, 512)
        feed = {pls[0]: batch['pc'], pls[3]: batch['one_hot_vec'], pls[4]: batch['y_seg'], pls[5]: batch['y_center'],
                pls[6]: batch['y_orient_cls'], pls[7]: batch['y_orient_reg'], pls[8]: batch['y_dims_cls'], pls[9]: batch['y_dims_reg'],
                pls[15]: batch['box2D'], pls[16]: batch['img_dim'], pls[17]: batch['is_data_2D'],
                'inst_seg/dp1': batch['dropout_masks']['inst_seg/dp1']}
        loss_val, center_val, s1_val = sess.run([semi_loss, end_points['center'], end_points['stage1_center']], feed_dict=feed)
        forced = product_decisions(g.assembly)
    c = R.default_config(USE_NORMALIZED_BOX2D_AS_FEATS=True)
    loss, ep, _, _ = R.model_a_forward_backward(P0, batch, c, bn_decay_val=R.bn_decay(0, B), use_one_hot=True, forced=forced,
                                                want_grads=False)
    assert abs(float(loss_val) - float(loss)) < 1e-4 * max(1.0, float(loss))
    assert np.abs(center_val - ep['center'].detach().numpy()).max() < 1e-4
    assert np.abs(s1_val - ep['stage1_center'].detach().numpy()).max() < 1e-4
    # and the flag off drops the handle, as semisup_v1_sunrgbd.py:97 does
    FLAGS.USE_NORMALIZED_BOX2D_AS_FEATS = False
    with api.Graph(rt=_runtime(), seed=3).as_default() as g2:
        pls = MODEL.placeholder_inputs(B, N, C)
        MODEL.get_semi_model(pls[0], pls[1], pls[2], pls[3], True, use_one_hot=True,
                             norm_box2D=tf_util.tf_normalize_2D_bboxes(pls[15], pls[16]), bn_decay=None, c=FLAGS)
        assert g2.vars.state_dict()['tnet/fc1-stage1/weights'].shape[0] == 256 + 10


# ---- Box-PC: BOXPC_WEIGH_DELTA_PRED_BY_CLS_CONF / BOXPC_STOP_GRAD_OF_CLS_VIA_DELTA / BOXPC_DELTA_LOSS_TYPE (boxpc_sunrgbd.py:73-92,150-164)
BOXPC_VARIANTS = [
    {'BOXPC_WEIGH_DELTA_PRED_BY_CLS_CONF': True},
    {'BOXPC_WEIGH_DELTA_PRED_BY_CLS_CONF': True, 'BOXPC_STOP_GRAD_OF_CLS_VIA_DELTA': False},
    {'BOXPC_WEIGH_DELTA_LOSS_BY_CLS_CONF': True, 'BOXPC_STOP_GRAD_OF_CLS_VIA_DELTA': False},
    {'BOXPC_WEIGH_DELTA_PRED_BY_CLS_CONF': True, 'BOXPC_WEIGH_DELTA_LOSS_BY_CLS_CONF': True, 'BOXPC_STOP_GRAD_OF_CLS_VIA_DELTA': False,
     'BOXPC_DELTA_LOSS_TYPE': 'mse'},
]


@pytest.mark.parametrize('over', BOXPC_VARIANTS, ids=lambda o: '+'.join(sorted(k[6:] for k in o)))
def test_boxpc_delta_weighting_variants_follow_the_oracle(over):
    rep = trajectory_check(_runtime(), 'boxpc', steps=2, B=SHAPE[0], N=SHAPE[1], config_over=over)
    assert all(r['weight_entries_checked'] > 1000 for r in rep[:-1])


# ---- stage c: SEMI_REFINE_USING_BOXPC_DELTA_NUM > 1 in the TRAINING graph (train_semisup_adv.py:362-399) ----------------------------
STAGE_C_VARIANTS = [
    {'SEMI_REFINE_USING_BOXPC_DELTA_NUM': 3},                                                   # loss as with one step; F2_ heads after three
    {'SEMI_REFINE_USING_BOXPC_DELTA_NUM': 2, 'SEMI_BOXPC_MIN_FIT_LOSS_AFT_REFINE': True},       # the gradient runs back through the step
    {'SEMI_REFINE_USING_BOXPC_DELTA_NUM': 3, 'SEMI_BOXPC_MIN_FIT_LOSS_AFT_REFINE': True, 'SEMI_WEIGH_BOXPC_DELTA_DURING_TEST': True,
     'BOXPC_WEIGH_DELTA_PRED_BY_CLS_CONF': True, 'BOXPC_STOP_GRAD_OF_CLS_VIA_DELTA': False},
]


@pytest.mark.parametrize('over', STAGE_C_VARIANTS, ids=lambda o: '+'.join('%s=%s' % (k.split('_')[-1], v) for k, v in sorted(o.items())))
def test_stage_c_training_graph_with_several_refinement_steps(over):
    rep = trajectory_check(_runtime(), 'F', steps=2, B=SHAPE[0], N=SHAPE[1], config_over=over)
    assert all(r['weight_entries_checked'] > 1000 for r in rep[:-1])
