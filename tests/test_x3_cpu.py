"""CPU: what the three-term bf16 arithmetic of the fp32 GEMM kernels (csrc/pointmlp.hip PathX3, oracle/ref_x3.py) guarantees --
the split of an fp32 value into three bf16 values is EXACT, and the six products the kernels form reproduce a product of fp32
values to 2^-22 relative: the tolerance behind running fp32 layers on the bf16 matrix pipe."""
import numpy as np

from oracle import ref_x3 as X


def _values(r, n):
    return (r.normal(size=n) * np.exp(r.uniform(-30, 30, size=n))).astype(np.float32)


def test_split_into_three_bf16_terms_is_exact():
    r = np.random.RandomState(0)
    x = np.concatenate([_values(r, 200000), np.float32([0.0, -0.0, 1.0, -1.0, 1e-30, 3e38, 2.0 ** -100, 1 + 2.0 ** -23, 1 - 2.0 ** -24])])
    h, m, l = X.split3(x)
    for t in (h, m, l):
        assert np.all(t.view(np.uint32) & 0xFFFF == 0)                 # each term is a bf16 value
    assert np.array_equal(h.astype(np.float64) + m.astype(np.float64) + l.astype(np.float64), x.astype(np.float64))
    nz = h != 0
    assert np.all(np.abs(m[nz]) <= np.abs(h[nz]) * 2.0 ** -8 * 1.0001) and np.all(np.abs(l[nz]) <= np.abs(h[nz]) * 2.0 ** -16 * 1.0001)


def test_six_products_reproduce_the_fp32_product():
    r = np.random.RandomState(1)
    x, y = _values(r, 100000), _values(r, 100000)
    xh, xm, xl = (t.astype(np.float64) for t in X.split3(x))
    yh, ym, yl = (t.astype(np.float64) for t in X.split3(y))
    six = xl * yh + xh * yl + xm * ym + xm * yh + xh * ym + xh * yh
    exact = x.astype(np.float64) * y.astype(np.float64)
    rel = np.abs(six - exact) / np.abs(exact)
    assert rel.max() < 2.0 ** -22, rel.max()              # dropped: m l + l m + l l < 3 * 2^-24 |x y|
    # a product of two bf16 values has 16 significant bits: exact in fp32, so the hardware's fp32 accumulation adds exact terms
    p = (xh * yh).astype(np.float32).astype(np.float64)
    assert np.array_equal(p, xh * yh)


def test_matrix_product_error_is_that_of_fp32_accumulation():
    r = np.random.RandomState(2)
    a, b = r.normal(size=(64, 512)).astype(np.float32), (r.normal(size=(512, 48)) / 22).astype(np.float32)
    ref = a.astype(np.float64) @ b.astype(np.float64)
    x3 = X.matmul_x3(a, b)
    f32 = (a @ b).astype(np.float64)                      # an fp32 product + accumulation
    scale = np.abs(ref).max()
    assert np.abs(x3 - ref).max() < 2e-7 * scale          # the emulation's own error (fp64 accumulation): the dropped terms only
    assert np.abs(x3 - ref).max() < np.abs(f32 - ref).max()      # ... smaller than what fp32 accumulation alone costs
