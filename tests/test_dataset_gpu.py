"""GPU: t3d_batch_assemble against the oracle restatement (explicit draws), against the NumPy specification (generated
draws: same hash generator), and the statistics of the generated draws."""
import numpy as np
import pytest

from fake_t3d import FakeLib
from test_dataset_cpu import assemble, check_against_oracle, check_generated_draws
from transferable3d_amd.dataset import synthetic_frustums
from transferable3d_amd.engine import Runtime

pytestmark = pytest.mark.gpu


def test_batch_assembly_matches_the_reference_restatement(hip_lib):
    for flags in ((True, True, True), (False, False, False), (True, False, True)):
        check_against_oracle(Runtime(lib=hip_lib), flags)


def test_generated_draws_and_permutation_walk(hip_lib):
    check_generated_draws(Runtime(lib=hip_lib))


def test_generated_batch_equals_the_specification(hip_lib):
    """Same counter-based generator on both sides: identical resampling indices and flips; the Box-Muller shift differs
    only by the rounding of logf/cosf."""
    B, N, Cc = 8, 512, 6
    host = synthetic_frustums(40, num_channel=6, seed=9, min_points=100, max_points=700)
    for step in (0, 5):
        c, _ = assemble(Runtime(device='cpu', lib=FakeLib()), host, B, N, Cc, step=step, seed=11)
        g, _ = assemble(Runtime(lib=hip_lib), host, B, N, Cc, step=step, seed=11)
        assert np.array_equal(c['y_seg'], g['y_seg']) and np.array_equal(c['y_dims_cls'], g['y_dims_cls'])
        assert np.array_equal(c['pc'][:, :, 3:], g['pc'][:, :, 3:])            # copied channels: same source points
        assert np.abs(c['pc'] - g['pc']).max() < 1e-4 and np.abs(c['y_center'] - g['y_center']).max() < 1e-4
        assert np.array_equal(c['y_orient_cls'], g['y_orient_cls']) and np.abs(c['y_orient_reg'] - g['y_orient_reg']).max() < 1e-5


def test_device_batch_against_the_batch_the_reference_assembled(hip_lib):
    """The kernel on the reference's own frustum file, slots and np.random draws (tests/golden/reference_vectors.npz)."""
    from test_reference_vectors import check_device_batch_against_the_reference_batch
    check_device_batch_against_the_reference_batch(Runtime(lib=hip_lib))


def test_alternate_batch_sampling(hip_lib):
    from test_dataset_cpu import check_alternate_batch
    check_alternate_batch(Runtime(lib=hip_lib))


def test_boxpc_perturb_against_the_reference_restatement(hip_lib):
    from test_dataset_cpu import check_boxpc_perturb_against_oracle
    check_boxpc_perturb_against_oracle(hip_lib, 'cuda')


def test_boxpc_perturb_generated_draws_equal_the_specification(hip_lib):
    """Same hash generator on both sides: the device picks the candidates the NumPy specification picks (an IoU within rounding
    of a bound could flip one acceptance; none does for this seed)."""
    from test_dataset_cpu import check_boxpc_perturb_generated
    base, t, o = check_boxpc_perturb_generated(hip_lib, 'cuda')
    _, tc, oc = check_boxpc_perturb_generated(FakeLib(), 'cpu')
    assert np.abs(o['iou'].numpy() - oc['iou'].numpy()).max() < 2e-5
    assert np.abs(o['dc'].numpy() - oc['dc'].numpy()).max() < 1e-6 and np.abs(o['da'].numpy() - oc['da'].numpy()).max() < 1e-6
    assert np.array_equal(t['ocls'].numpy(), tc['ocls'].numpy())


def test_equal_samples_per_class_sampler(hip_lib):
    from test_dataset_cpu import check_equal_class_sampler
    check_equal_class_sampler(Runtime(lib=hip_lib))
