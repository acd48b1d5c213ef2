import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'tests')):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def hip_lib():
    """The real HIP library; GPU tests fail loudly if it is missing (no fallback)."""
    from transferable3d_amd import abi
    return abi.load()


@pytest.fixture
def fp32_mfma(monkeypatch):
    """The fp32-MFMA GEMM kernels (v_mfma_f32_32x32x2_f32) instead of the default three-term bf16 form (csrc/pointmlp.hip PathX3):
    for tests that pin relations BETWEEN fp32-MFMA kernels (one-pass vs split form bit for bit, ...).  T3D_X3 is what a launch
    struct WITHOUT a request (`arith` = T3D_ARITH_AUTO: the kernel tests' structs) follows, read at every such launch, and the
    default of `engine.Runtime(gemm_arithmetic=...)`, read once when a runtime is built."""
    monkeypatch.setenv('T3D_X3', '0')


@pytest.fixture(params=['x3', 'fp32_mfma'])
def gemm_arithmetic(request, monkeypatch):
    """Runs a test under both GEMM arithmetics of the fp32 path: 'x3' (default: bf16 matrix pipe, three bf16 terms per operand, six
    products) and 'fp32_mfma' -- through T3D_X3, i.e. the default of every `Runtime` the test builds afterwards (which then puts
    the request into its launch structs) and the choice of request-less structs."""
    monkeypatch.setenv('T3D_X3', '1' if request.param == 'x3' else '0')
    return request.param
