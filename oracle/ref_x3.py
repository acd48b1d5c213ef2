"""TEST INFRASTRUCTURE ONLY -- NumPy restatement of the three-term bf16 arithmetic the fp32 GEMM kernels run on (csrc/pointmlp.hip
PathX3): x = h + m + l with h = bf16(x), m = bf16(x - h), l = bf16(x - h - m) (round to nearest even, subtractions in fp32), and
x * y ~ hh + hm + mh + hl + lh + mm.  Nothing in the reference corresponds to it (TensorFlow multiplies in fp32); it exists so that
the CPU suite can state what the emulation guarantees: the split is exact, and the six products reproduce the fp32 product to
2^-22 of |x y| -- the dropped terms (ml, lm, ll) are below 2^-23 |x y|."""
import numpy as np


def bf16_round(x):
    """fp32 -> nearest bf16 (ties to even), returned as fp32 (v_cvt_pk_bf16_f32)."""
    u = np.asarray(x, np.float32).view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
    return (r & 0xFFFFFFFF).astype(np.uint32).view(np.float32)


def split3(x):
    x = np.asarray(x, np.float32)
    h = bf16_round(x)
    r1 = (x - h).astype(np.float32)
    m = bf16_round(r1)
    r2 = (r1 - m).astype(np.float32)
    return h, m, bf16_round(r2)


def matmul_x3(a, b):
    """a [M,K] @ b [K,N] with the kernels' six products, accumulated in fp64 here (the hardware accumulates in fp32: the
    difference is the ordinary fp32 accumulation error both arithmetics share)."""
    ah, am, al = (t.astype(np.float64) for t in split3(a))
    bh, bm, bl = (t.astype(np.float64) for t in split3(b))
    return al @ bh + ah @ bl + am @ bm + am @ bh + ah @ bm + ah @ bh
