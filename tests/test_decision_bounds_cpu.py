"""CPU: the flip-aware gradient checks hand the product's ReLU / arg-max / mask decisions to the oracle -- and bound them
(model_check.check_decision_margins on oracle.ref_torch.Ctx.margins): few flips, each within rounding of the boundary by the oracle's
own numbers.  A product that decides wrongly on 1 % of a layer's elements, pools a row that is not the maximum, or masks points whose
logits are not tied must FAIL that check instead of steering the oracle's gradient."""
import numpy as np
import pytest
import torch

from fake_t3d import FakeLib
from model_check import check_against_oracle, check_decision_margins, product_decisions, run_model_a
from oracle import ref_torch as R
from transferable3d_amd.engine import Runtime
from transferable3d_amd.synthetic import make_batch


@pytest.fixture(scope='module')
def case():
    B, N, C = 4, 128, 4
    P = R.init_params(np.random.RandomState(11), R.layer_table(C, 'A'))
    c = R.default_config()
    batch = make_batch(B, N, C, seed=3, dropout_scopes={'inst_seg/dp1': ((B, N, 128), 0.5)})
    g, m = run_model_a(Runtime(device='cpu', lib=FakeLib()), batch, P, c)
    return g, m, batch, P, c


def _margins(case, forced):
    _, _, batch, P, c = case
    _, ep, _, _ = R.model_a_forward_backward(P, batch, c, bn_decay_val=0.5, forced=forced, want_grads=False)
    return ep['__margins__']


def test_honest_decisions_pass_and_are_recorded_per_site(case):
    g, m, batch, P, c = case
    res = check_against_oracle(g, m, batch, P, c)                  # includes check_decision_margins
    mg = _margins(case, product_decisions(m))
    assert 'mask' in mg and 'inst_seg/conv5#argmax' in mg and 'inst_seg/conv2' in mg and 'box_est/fc1' in mg
    for site, (numel, flips, margin, scale) in mg.items():
        assert numel > 0 and flips <= 2 and margin <= 1e-4 * max(1.0, scale), (site, flips, margin)
    assert isinstance(res['forced'], dict)


def test_one_percent_of_wrong_gates_fail(case):
    forced = product_decisions(case[1])
    r = np.random.RandomState(0)
    gate = np.array(forced['gates']['inst_seg/conv3'], copy=True)
    flip = r.rand(*gate.shape) < 0.01
    gate[flip] = ~gate[flip] if gate.dtype == bool else 1 - gate[flip]
    forced['gates']['inst_seg/conv3'] = gate
    mg = _margins(case, forced)
    assert mg['inst_seg/conv3'][1] >= 0.005 * gate.size
    with pytest.raises(AssertionError, match='inst_seg/conv3'):
        check_decision_margins(mg)
    # a handful of flips is allowed by COUNT, but not when they are far from the boundary
    gate = np.array(product_decisions(case[1])['gates']['inst_seg/conv3'], copy=True)
    gate.reshape(-1)[:2] = ~gate.reshape(-1)[:2] if gate.dtype == bool else 1 - gate.reshape(-1)[:2]
    forced['gates']['inst_seg/conv3'] = gate
    mg = _margins(case, forced)
    assert mg['inst_seg/conv3'][1] == 2 and mg['inst_seg/conv3'][2] > 1e-3
    with pytest.raises(AssertionError, match='oracle margin'):
        check_decision_margins(mg)


def test_a_pooled_row_that_is_not_the_maximum_fails(case):
    forced = product_decisions(case[1])
    idx = np.array(forced['argmax']['box_est/conv-reg4'], copy=True)
    idx[0, :3] = (idx[0, :3] + 17) % 128                          # three channels of one frustum take some other row
    forced['argmax']['box_est/conv-reg4'] = idx
    mg = _margins(case, forced)
    with pytest.raises(AssertionError, match='conv-reg4#argmax'):
        check_decision_margins(mg)


def test_masking_points_whose_logits_are_not_tied_fails(case):
    forced = product_decisions(case[1])
    mask = np.array(forced['mask'], copy=True)
    mask[1, :2] = 1 - mask[1, :2]
    forced['mask'] = mask
    mg = _margins(case, forced)
    assert mg['mask'][1] == 2
    with pytest.raises(AssertionError, match='mask'):
        check_decision_margins(mg)
