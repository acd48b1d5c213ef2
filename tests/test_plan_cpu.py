"""CPU: the host-side launch plan (engine.py / nets.py), executed on the NumPy executable specification of
the C ABI (tests/fake_t3d.py), reproduces the oracle's forward, loss, gradients and EMA updates.  This
validates the fused schedule (lazy batch-norm apply, split conv6, pooled-sparse gradients, colsum tricks)
independently of the HIP kernels, which are checked against the same specification on the GPU."""
import numpy as np
import pytest
import torch

from fake_t3d import FakeLib
from model_check import check_against_oracle, load_golden, run_model_a
from oracle import ref_torch as R
from transferable3d_amd.engine import Runtime
from transferable3d_amd.synthetic import make_batch


def _rt():
    return Runtime(device='cpu', lib=FakeLib())


@pytest.mark.parametrize('B,N,seed', [(2, 128, 3), (4, 256, 1)])
def test_model_a_plan_matches_oracle(B, N, seed):
    C = 4
    batch = make_batch(B, N, C, seed=seed, dropout_scopes={'inst_seg/dp1': ((B, N, 128), 0.5)})
    P = R.init_params(np.random.RandomState(7 + seed), R.layer_table(C, 'A'))
    c = R.default_config()
    g, m = run_model_a(_rt(), batch, P, c)
    res = check_against_oracle(g, m, batch, P, c)
    assert res['grad_median'] < 1e-4


def test_model_a_plan_matches_golden_vectors():
    batch, P, z = load_golden('model_a_B4_N128.npz')
    g, m = run_model_a(_rt(), batch, P, R.default_config())
    e = m.end_points()
    for k in ('logits', 'stage1_center', 'center', 'box_params', 'feats_lv1', 'mask_xyz_mean'):
        ref = z['out/' + k]
        assert np.abs(e[k].numpy().reshape(ref.shape) - ref).max() < 1e-4, k
    assert abs(float(e['loss']) - float(z['out/loss'])) < 1e-4
    gmax = max(float(np.abs(z[k]).max()) for k in z.files if k.startswith('grad/'))
    for k in z.files:
        if k.startswith('grad/') and k != 'grad/box_est/fc1/weights':
            ref = z[k]
            mine = g.vars.grad(k[5:]).numpy().reshape(ref.shape)
            assert np.abs(mine - ref).max() < 2e-3 * max(np.abs(ref).max(), 1e-2 * gmax), k


def test_is_data_2D_masks_the_loss_and_gradients():
    B, N, C = 2, 128, 4
    batch = make_batch(B, N, C, seed=5, dropout_scopes={'inst_seg/dp1': ((B, N, 128), 0.5)})
    batch['is_data_2D'][:] = 1
    P = R.init_params(np.random.RandomState(2), R.layer_table(C, 'A'))
    g, m = run_model_a(_rt(), batch, P, R.default_config())
    assert float(m.end_points()['loss']) == 0.0
    assert float(g.vars.grads.abs().max()) == 0.0


def test_all_points_background_gives_empty_mask_and_zero_centroid():
    """mask_count = 0 -> mean = sum/ max(count,1) = 0 and the masked max-pools return 0
    (semisup_models.py:158, 185-188)."""
    B, N, C = 2, 128, 4
    batch = make_batch(B, N, C, seed=6, dropout_scopes={'inst_seg/dp1': ((B, N, 128), 0.5)})
    P = R.init_params(np.random.RandomState(3), R.layer_table(C, 'A'))
    P['inst_seg/conv10/biases'] = torch.tensor([50.0, -50.0], dtype=torch.float64)     # logit0 >> logit1
    c = R.default_config()
    g, m = run_model_a(_rt(), batch, P, c)
    e = m.end_points()
    assert float(e['mask'].sum()) == 0.0
    assert float(e['mask_xyz_mean'].abs().max()) == 0.0
    assert float(e['feats_lv1'].abs().max()) == 0.0
    # saturated logits make single-point CE gradients O(1): one fp32 ReLU-boundary flip moves the whole seg
    # chain by ~3e-3, so only the global bound is meaningful here
    check_against_oracle(g, m, batch, P, c, grad_median_tol=None)


def test_config0_single_frustum_forward_plumbing():
    from model_check import check_config0_single_frustum_forward
    check_config0_single_frustum_forward(Runtime(device='cpu', lib=FakeLib()))


@pytest.mark.parametrize('C', [3, 6])
def test_model_a_with_three_and_six_channel_point_clouds(C):
    """The reference's default is NUM_CHANNELS = 6 (xyz + rgb; 3 with --no_rgb, train_semisup.py:67-75): the point cloud's rows are
    padded to a multiple of four floats in HBM (16-byte operand loads), the [1,C] first convolution sees exactly C channels."""
    B, N = 4, 128
    batch = make_batch(B, N, C, seed=6, dropout_scopes={'inst_seg/dp1': ((B, N, 128), 0.5)})
    P = R.init_params(np.random.RandomState(13), R.layer_table(C, 'A'))
    c = R.default_config()
    g, m = run_model_a(_rt(), batch, P, c)
    assert g.ldpc == (4 if C == 3 else 8) and m.inputs.pc.shape == (B * N, g.ldpc)
    check_against_oracle(g, m, batch, P, c)
