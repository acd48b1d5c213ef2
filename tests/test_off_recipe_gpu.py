"""GPU: tests/test_off_recipe_cpu.py -- the flag-selected branches outside the published recipes (norm_box2D features, the Box-PC
delta weightings / mse loss, several Box-PC refinement steps inside the stage-c TRAINING graph) -- re-run with the HIP library on the
MI355X, every step against the oracle."""
import pytest

import test_off_recipe_cpu as T
from transferable3d_amd.engine import Runtime

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _hip_runtime(hip_lib, monkeypatch):
    monkeypatch.setattr(T, '_runtime', lambda: Runtime(lib=hip_lib))
    monkeypatch.setattr(T, 'SHAPE', (8, 256))


@pytest.mark.parametrize('workload', ['A', 'F'])
def test_norm_box2d_features_follow_the_oracle(workload):
    T.test_norm_box2d_features_follow_the_oracle(workload)


def test_norm_box2d_changes_the_graph():
    T.test_norm_box2d_changes_the_graph()


def test_reference_call_sequence_with_norm_box2d_and_one_hot():
    T.test_reference_call_sequence_with_norm_box2d_and_one_hot()


@pytest.mark.parametrize('over', T.BOXPC_VARIANTS, ids=lambda o: '+'.join(sorted(k[6:] for k in o)))
def test_boxpc_delta_weighting_variants_follow_the_oracle(over):
    T.test_boxpc_delta_weighting_variants_follow_the_oracle(over)


@pytest.mark.parametrize('over', T.STAGE_C_VARIANTS, ids=lambda o: '+'.join('%s=%s' % (k.split('_')[-1], v) for k, v in sorted(o.items())))
def test_stage_c_training_graph_with_several_refinement_steps(over):
    T.test_stage_c_training_graph_with_several_refinement_steps(over)
