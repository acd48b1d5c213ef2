"""GPU: the API-surface tests of tests/test_api_cpu.py -- the reference's call sequences of train_semisup.py / train_boxpc.py /
train_semisup_adv.py (placeholder_inputs -> get_semi_model / get_model -> get_semi_loss / get_loss -> AdamOptimizer.minimize ->
Session.run with a feed_dict), the fed is_training placeholder, and the operator wrappers of tf_util called one by one -- re-run
with the HIP library on the MI355X (the step inside Session.run is the hipGraph-captured transferable3d_amd.step.TrainStep)."""
import pytest

import test_api_cpu as T
from transferable3d_amd.engine import Runtime

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _hip_runtime(hip_lib, monkeypatch):
    monkeypatch.setattr(T, '_runtime', lambda: Runtime(lib=hip_lib))


def test_reference_call_sequence_runs_a_training_step():
    T.test_reference_call_sequence_runs_a_training_step()


def test_forward_only_fetch_compiles_the_inference_plan():
    T.test_forward_only_fetch_compiles_the_inference_plan()


def test_boxpc_reference_call_sequence():
    T.test_boxpc_reference_call_sequence()


def test_stage_c_reference_call_sequence_with_var_list():
    T.test_stage_c_reference_call_sequence_with_var_list()


def test_is_training_placeholder_selects_train_and_eval_schedules_of_one_graph():
    T.test_is_training_placeholder_selects_train_and_eval_schedules_of_one_graph()


@pytest.mark.parametrize('is_training', [True, False])
def test_operator_wrappers_called_one_by_one(is_training):
    T.test_operator_wrappers_called_one_by_one(is_training)


@pytest.mark.parametrize('use_one_hot', [False, True])
def test_operator_surface_takes_the_reference_inst_seg_call_sequence(use_one_hot):
    T.test_operator_surface_takes_the_reference_inst_seg_call_sequence(use_one_hot)
