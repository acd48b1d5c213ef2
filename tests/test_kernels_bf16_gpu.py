"""GPU: the T3D_BF16 GEMM kernels (bf16 storage, v_mfma_f32_32x32x16_bf16, fp32 accumulation) against stock-torch matmuls of the
SAME bf16-rounded operands in fp64.  What is checked is the kernels' index arithmetic -- the two LDS image formats, the
transposing fragment reads, the accumulator -> row/column mapping of the typed epilogues -- so the bound is tight: the only error
left is the fp32 accumulation order (1e-5 relative to the operand scale) plus ONE bf16 rounding of each stored output
(2^-9 = 2e-3 relative)."""
import ctypes as C

import numpy as np
import pytest
import torch

from transferable3d_amd import abi
from transferable3d_amd.abi import fptr, iptr

pytestmark = pytest.mark.gpu
DEV = 'cuda'
BF = torch.bfloat16


def stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def rb(t):
    """round to bf16, back to fp64"""
    return t.to(BF).double()


def close_bf16(got, ref, what, acc_rel=2e-5, opmax=0.0):
    """got: stored bf16 tensor; ref: fp64 value before the output rounding.  opmax = max|A| * max|B| of the two operands: the
    kernel forms an operand element in fp32 and rounds it to bf16, the reference in fp64 -- about one element in 2^16 lands on the
    other side of a bf16 rounding boundary and then differs by one bf16 spacing (2^-7 relative); allow two such elements per
    output.  (A wrong index anywhere moves outputs by the operand scale itself.)"""
    got, ref = got.double().cpu(), ref.double().cpu()
    scale = float(ref.abs().max())
    tol = 2.0 ** -8 * ref.abs() + acc_rel * scale + 2 * 2.0 ** -7 * opmax + 1e-30
    bad = (got - ref).abs() > tol
    assert not bool(bad.any()), (what, int(bad.sum()), float((got - ref).abs().max()), scale)


@pytest.mark.parametrize('M,K,N,rpf,xbf,pool', [(512, 64, 64, 256, True, False), (256, 4, 64, 128, False, False),
                                                (256, 3, 128, 128, False, False), (512, 128, 256, 256, True, True),
                                                (256, 64, 512, 128, True, False), (384, 512, 256, 128, True, False),
                                                (65536, 128, 128, 1024, True, False),
                                                # more row tiles than persistent workgroups (536 > 512): the resident forward walks a second tile
                                                # with its panel / weight / bias prefetched (k_pointmlp_fwd_res, round 3)
                                                (68608, 128, 256, 1024, True, False), (68608, 128, 256, 1024, True, True),
                                                (68608, 64, 128, 1024, True, False)])
def test_bf16_forward(hip_lib, M, K, N, rpf, xbf, pool):
    g = torch.Generator(device='cpu').manual_seed(M + K + N)
    ldx = 4 if K <= 4 else K
    x32 = torch.randn(M, ldx, generator=g)
    x = (x32.to(BF) if xbf else x32).to(DEV)
    w32 = torch.randn(K, N, generator=g) / np.sqrt(K)
    w16 = w32.to(BF).to(DEV)
    bias = (torch.randn(N, generator=g) * 0.1).to(DEV)
    sc = (0.5 + torch.rand(K, generator=g)).to(DEV)
    sh = (torch.randn(K, generator=g) * 0.2).to(DEV)
    B, T = M // rpf, M // 128
    sub = torch.randn(B, 3, generator=g).to(DEV)
    mask = (torch.rand(M, generator=g) < 0.4).float().to(DEV)
    bn = K > 4
    y = torch.zeros(M, N, dtype=BF, device=DEV)
    psum, psumsq = torch.zeros(T, N, device=DEV), torch.zeros(T, N, device=DEV)
    a = abi.PointMlpFwdArgs()
    a.a = abi.ActSrc(fptr(x), ldx, 0, fptr(sc if bn else None), fptr(sh if bn else None), int(bn), fptr(sub if K == 3 else None), 3,
                     abi.BF16 if xbf else abi.F32)
    a.w, a.bias, a.psum, a.psumsq = fptr(w16), fptr(bias), fptr(psum), fptr(psumsq)
    if pool:
        pmax, pmin = torch.zeros(T, N, device=DEV), torch.zeros(T, N, device=DEV)
        pamax, pamin = torch.zeros(T, N, dtype=torch.int32, device=DEV), torch.zeros(T, N, dtype=torch.int32, device=DEV)
        a.pmax, a.pmin, a.pamax, a.pamin, a.rowmask = fptr(pmax), fptr(pmin), iptr(pamax), iptr(pamin), fptr(mask)
    else:
        a.y = fptr(y)
    a.M, a.K, a.N, a.rows_per_frustum, a.dtype = M, K, N, rpf, abi.BF16
    assert hip_lib.t3d_pointmlp_fwd(C.byref(a), stream()) == 0
    torch.cuda.synchronize()
    act = x.double()[:, :K]
    if bn:
        act = torch.relu(act * sc.double() + sh.double())
    if K == 3:
        act = act - sub.double().repeat_interleave(rpf, 0)
    ref = rb(act) @ w16.double() + bias.double()
    if not pool:
        close_bf16(y, ref, 'y', opmax=float(act.abs().max() * w16.double().abs().max()))
        ys = y.double().reshape(T, 128, N)
        assert float((psum.double() - ys.sum(1)).abs().max()) < 1e-4 * float(ys.abs().sum(1).max() + 1)
        assert float((psumsq.double() - (ys * ys).sum(1)).abs().max()) < 1e-4 * float((ys * ys).sum(1).max() + 1)
    else:
        rt = ref.reshape(T, 128, N)
        assert float((psum.double() - rt.sum(1)).abs().max()) < 1e-4 * float(rt.abs().sum(1).max() + 1)
        keep = mask.reshape(T, 128, 1) > 0
        mx = torch.where(keep, rt, torch.full_like(rt, -1e30)).max(1).values
        has = keep.any(1).expand_as(mx)
        assert float((pmax.double() - mx)[has].abs().max()) < 1e-4 * float(rt.abs().max())
        assert bool(((pamax >= 0) == has).all())


@pytest.mark.parametrize('M,K,N,rpf,addin,rps_', [
    (512, 64, 128, 256, False, 0), (256, 64, 512, 128, True, 0), (384, 512, 256, 128, False, 0), (256, 256, 128, 128, True, 0),
    (65536, 128, 128, 1024, False, 0),
    # the one-pass form (t3d_bwd_plan's split, or any split of whole 128-row tiles that leaves >= min(256, M/128) workgroups)
    (512, 64, 64, 256, True, -1), (1024, 128, 64, 256, False, -1), (768, 64, 128, 128, True, -1), (65536, 128, 128, 1024, True, -1),
    (131072, 128, 128, 2048, False, 512), (65536, 64, 64, 2048, True, 256), (32768, 128, 64, 1024, False, 128),
    (512, 256, 128, 256, True, -1), (1024, 128, 256, 256, False, -1), (65536, 256, 128, 1024, False, 256), (65536, 128, 256, 2048, True, 256)])
def test_bf16_fused_backward(hip_lib, M, K, N, rpf, addin, rps_):
    """t3d_pointmlp_bwd: dX = dy . W^T with the ReLU mask / batch-norm-backward partials of the producing layer, dW = a^T dy as
    row-split slabs; dy = c0*dz + c1*y + c2 formed from bf16 dz, y while loading.  rps_: 0 = t3d_wgrad_plan's split (the split
    form unless the shape is one-pass eligible and small), -1 = t3d_bwd_plan's, > 0 = that many rows per split."""
    g = torch.Generator(device='cpu').manual_seed(M + K + N + 1)
    T = M // 128
    dz = (torch.randn(M, N, generator=g) * 1e-2).to(BF).to(DEV)
    y = torch.randn(M, N, generator=g).to(BF).to(DEV)
    coef = torch.randn(3, N, generator=g)
    coef[2] *= 1e-3
    coef = coef.to(DEV)
    w32 = torch.randn(K, N, generator=g) / np.sqrt(N)
    w16 = w32.to(BF).to(DEV)
    prev_y = torch.randn(M, K, generator=g).to(BF).to(DEV)
    psc = (0.5 + torch.rand(K, generator=g)).to(DEV)
    psh = (torch.randn(K, generator=g) * 0.3).to(DEV)
    add = (torch.randn(M, K, generator=g) * 1e-2).to(BF).to(DEV)
    out = torch.zeros(M, K, dtype=BF, device=DEV)
    ps1, ps2 = torch.zeros(T, K, device=DEV), torch.zeros(T, K, device=DEV)
    rps, tk, tn = C.c_int(0), C.c_int(0), C.c_int(0)
    if rps_ == 0:
        assert hip_lib.t3d_wgrad_plan(M, K, N, C.byref(rps), C.byref(tk), C.byref(tn)) == 0
    elif rps_ < 0:
        one = C.c_int(0)
        assert hip_lib.t3d_bwd_plan(M, K, N, abi.BF16, C.byref(rps), C.byref(one)) == 0
        assert one.value == 1 and rps.value % 128 == 0 and M // rps.value >= min(256, M // 128)
    else:
        rps.value = rps_
    ns = M // rps.value
    slabs = torch.full((ns, K, N), float('nan'), device=DEV)
    dy = abi.DySrc(fptr(dz), fptr(y), fptr(coef), iptr(None), fptr(None), abi.BF16)
    d = abi.PointMlpDgradArgs()
    d.dy, d.w, d.add_in = dy, fptr(w16), fptr(add if addin else None)
    d.prev_y, d.prev_scale, d.prev_shift, d.out, d.psum_dz, d.psum_dzy = fptr(prev_y), fptr(psc), fptr(psh), fptr(out), fptr(ps1), fptr(ps2)
    d.M, d.K, d.N, d.rows_per_frustum, d.dtype = M, K, N, rpf, abi.BF16
    wa = abi.PointMlpWgradArgs()
    wa.a = abi.ActSrc(fptr(prev_y), K, 0, fptr(psc), fptr(psh), 1, fptr(None), 0, abi.BF16)
    wa.dy, wa.slabs = dy, fptr(slabs)
    wa.M, wa.K, wa.N, wa.rows_per_frustum, wa.rows_per_split = M, K, N, rpf, rps.value
    assert hip_lib.t3d_pointmlp_bwd(C.byref(d), C.byref(wa), stream()) == 0
    torch.cuda.synchronize()
    dyv = rb(coef[0].double() * dz.double() + coef[1].double() * y.double() + coef[2].double())
    ref = dyv @ w16.double().t() + (add.double() if addin else 0)
    z = prev_y.double() * psc.double() + psh.double()
    refm = torch.where(z > 0, ref, torch.zeros_like(ref))
    close_bf16(out, refm, 'dX', opmax=float(dyv.abs().max() * w16.double().abs().max()))
    o = out.double().reshape(T, 128, K)
    assert float((ps1.double() - o.sum(1)).abs().max()) < 1e-4 * float(o.abs().sum(1).max() + 1e-9)
    assert float((ps2.double() - (o * prev_y.double().reshape(T, 128, K)).sum(1)).abs().max()) < 1e-4 * float(o.abs().sum(1).max() + 1e-9)
    a = rb(torch.relu(z))
    dw_ref = a.t() @ dyv
    dw = slabs.double().sum(0)
    flip = 2 * 2.0 ** -7 * float(a.abs().max() * dyv.abs().max()) * max(1.0, M / 65536.0 * 4)      # see close_bf16
    assert float((dw - dw_ref).abs().max()) < 2e-5 * float(dw_ref.abs().max()) * max(1.0, np.sqrt(M / 512)) + flip, \
        (float((dw - dw_ref).abs().max()), float(dw_ref.abs().max()))


@pytest.mark.parametrize('M,K,rpf', [(512, 128, 256), (1024, 256, 128), (512, 64, 128)])
def test_bf16_gram_forms(hip_lib, M, K, rpf):
    """G = a^T a slabs (both operands through the transposing LDS reads) and out = act(a) . P + rowconst with the dgrad epilogue."""
    g = torch.Generator(device='cpu').manual_seed(M + K)
    T = M // 128
    x = torch.randn(M, K, generator=g).to(BF).to(DEV)
    sc = (0.5 + torch.rand(K, generator=g)).to(DEV)
    sh = (torch.randn(K, generator=g) * 0.2).to(DEV)
    src = abi.ActSrc(fptr(x), K, 0, fptr(sc), fptr(sh), 1, fptr(None), 0, abi.BF16)
    rps, tk, tn = C.c_int(0), C.c_int(0), C.c_int(0)
    assert hip_lib.t3d_wgrad_plan(M, K, K, C.byref(rps), C.byref(tk), C.byref(tn)) == 0
    slabs = torch.zeros(M // rps.value, K, K, device=DEV)
    ga = abi.PointMlpGramArgs(src, fptr(slabs), M, K, rpf, rps.value)
    assert hip_lib.t3d_pointmlp_gram(C.byref(ga), stream()) == 0
    a = rb(torch.relu(x.double() * sc.double() + sh.double()))
    G = a.t() @ a
    torch.cuda.synchronize()
    assert float((slabs.double().sum(0) - G).abs().max()) < 2e-5 * float(G.abs().max())
    P = (torch.randn(K, K, generator=g) / np.sqrt(K)).to(DEV)
    rc = torch.randn(K, generator=g).to(DEV) * 0.1
    prev_y = torch.randn(M, K, generator=g).to(BF).to(DEV)
    out = torch.zeros(M, K, dtype=BF, device=DEV)
    ps1, ps2 = torch.zeros(T, K, device=DEV), torch.zeros(T, K, device=DEV)
    d = abi.PointMlpDgradGramArgs()
    d.a, d.p, d.rowconst = src, fptr(P), fptr(rc)
    d.prev_y, d.prev_scale, d.prev_shift, d.out, d.psum_dz, d.psum_dzy = fptr(prev_y), fptr(sc), fptr(sh), fptr(out), fptr(ps1), fptr(ps2)
    d.M, d.K, d.rows_per_frustum, d.dtype = M, K, rpf, abi.BF16
    assert hip_lib.t3d_pointmlp_dgrad_gram(C.byref(d), stream()) == 0
    torch.cuda.synchronize()
    ref = a @ rb(P) + rc.double()
    ref = torch.where(prev_y.double() * sc.double() + sh.double() > 0, ref, torch.zeros_like(ref))
    close_bf16(out, ref, 'dX gram', opmax=float(a.abs().max() * P.abs().max()))


@pytest.mark.parametrize('M,K,N,rpf', [(4096, 128, 256, 512), (2048, 256, 512, 256), (65536, 128, 1024, 2048), (32768, 256, 512, 1024)])
def test_bf16_one_pass_gram_stages(hip_lib, M, K, N, rpf):
    """The one-pass forms of the Gram-form backward (t3d_gram_plan's split; the layer input is also the producer's raw output):
    stage 1 = Gram slabs + per-tile column sums from ONE pass over the input (+ the P / rowconst / wc job), stage 2 = dW assembly +
    out = (act(a) . P + rowconst + sparse rows) masked with the producer's batch-norm-backward partials.  GEMM results against fp64
    on the same rounded operands; the small jobs (their 256-thread bodies run two to a 512-thread block) bit-identical to the
    stand-alone launches."""
    g = torch.Generator(device='cpu').manual_seed(M + K + N)
    B, T = M // rpf, M // 128
    x = torch.randn(M, K, generator=g).to(BF).to(DEV)
    sc = (0.5 + torch.rand(K, generator=g)).to(DEV)
    sh = (torch.randn(K, generator=g) * 0.2).to(DEV)
    src = abi.ActSrc(fptr(x), K, 0, fptr(sc), fptr(sh), 1, fptr(None), 0, abi.BF16)
    w = (torch.randn(K, N, generator=g) / np.sqrt(K)).to(DEV)
    bias = (torch.randn(N, generator=g) * 0.1).to(DEV)
    coef = torch.randn(3, N, generator=g).to(DEV)
    argidx = torch.randint(-1, rpf, (B, N), generator=g, dtype=torch.int32).to(DEV)
    dpool = torch.randn(B, N, generator=g).to(DEV)
    rps, one = C.c_int(0), C.c_int(0)
    assert hip_lib.t3d_gram_plan(M, K, abi.BF16, C.byref(rps), C.byref(one)) == 0
    assert one.value == 1 and rps.value % 128 == 0 and M // rps.value >= min(256, T)
    S_, nch = M // rps.value, (N + 127) // 128
    z = lambda *s: torch.zeros(*s, device=DEV)
    # ---- stage 1 ----
    gsl, part, ps, rcs, wc = torch.full((S_, K, K), float('nan'), device=DEV), z(T, K), z(nch, K, K), z(nch, K), z(N, K)
    ga = abi.PointMlpGramArgs(src, fptr(gsl), M, K, rpf, rps.value)
    ca = abi.ActColsumArgs(src, M, K, rpf, fptr(part))
    qa = abi.PoolBwdPrepArgs(fptr(w), fptr(bias), fptr(coef), K, N, fptr(ps), fptr(rcs), fptr(wc))
    assert hip_lib.t3d_pool_bwd_stage1(C.byref(ga), C.byref(ca), C.byref(qa), stream()) == 0
    torch.cuda.synchronize()
    a32 = torch.relu(x.float() * sc + sh)
    a = rb(a32)
    G = a.t() @ a
    assert float((gsl.double().sum(0) - G).abs().max()) < 2e-5 * float(G.abs().max()) * max(1.0, np.sqrt(M / 4096))
    col = a32.double().reshape(T, 128, K).sum(1)
    assert float((part.double() - col).abs().max()) < 1e-5 * float(col.abs().max())
    gsl2 = torch.zeros(S_, K, K, device=DEV)                      # stand-alone Gram launch: the same kernel body
    ga2 = abi.PointMlpGramArgs(src, fptr(gsl2), M, K, rpf, rps.value)
    assert hip_lib.t3d_pointmlp_gram(C.byref(ga2), stream()) == 0
    ps2, rcs2, wc2 = z(nch, K, K), z(nch, K), z(N, K)
    qa2 = abi.PoolBwdPrepArgs(fptr(w), fptr(bias), fptr(coef), K, N, fptr(ps2), fptr(rcs2), fptr(wc2))
    assert hip_lib.t3d_pool_bwd_prep(C.byref(qa2), stream()) == 0
    torch.cuda.synchronize()
    assert torch.equal(gsl, gsl2) and torch.equal(ps, ps2) and torch.equal(rcs, rcs2) and torch.equal(wc, wc2)
    # ---- stage 2 ----
    Gr, abar, P, rc = gsl.sum(0), part.sum(0), ps.sum(0), rcs.sum(0)
    live = (torch.rand(M, generator=g) < 0.1).to(torch.int32).to(DEV)
    Sm = torch.randn(M, K, generator=g).to(DEV)                   # rows without the flag must never be read: leave them non-zero
    out = torch.zeros(M, K, dtype=BF, device=DEV)
    dw, s1, s2 = z(K, N), z(T, K), z(T, K)
    f = abi.PoolWgradFinishArgs()
    f.a, f.argidx, f.dpool, f.coef, f.w, f.bias = src, iptr(argidx), fptr(dpool), fptr(coef), fptr(w), fptr(bias)
    f.g, f.abar, f.B, f.K, f.N, f.rows_per_frustum, f.dw = fptr(Gr), fptr(abar), B, K, N, rpf, fptr(dw)
    d = abi.PointMlpDgradGramArgs()
    d.a, d.p, d.rowconst, d.add_in, d.add_live = src, fptr(P), fptr(rc), fptr(Sm), iptr(live)
    d.prev_y, d.prev_scale, d.prev_shift, d.out, d.psum_dz, d.psum_dzy = fptr(x), fptr(sc), fptr(sh), fptr(out), fptr(s1), fptr(s2)
    d.M, d.K, d.rows_per_frustum, d.dtype = M, K, rpf, abi.BF16
    assert hip_lib.t3d_pool_bwd_stage2(C.byref(f), C.byref(d), stream()) == 0
    torch.cuda.synchronize()
    ref = a @ rb(P) + rc.double() + Sm.double() * live.double().unsqueeze(1)
    ref = torch.where(x.double() * sc.double() + sh.double() > 0, ref, torch.zeros_like(ref))
    close_bf16(out, ref, 'dX gram one-pass', opmax=float(a.abs().max() * P.abs().max()))
    o = out.double().reshape(T, 128, K)
    assert float((s1.double() - o.sum(1)).abs().max()) < 1e-4 * float(o.abs().sum(1).max() + 1e-9)
    assert float((s2.double() - (o * x.double().reshape(T, 128, K)).sum(1)).abs().max()) < 1e-4 * float(o.abs().sum(1).max() + 1e-9)
    dw2 = z(K, N)
    f.dw = fptr(dw2)
    assert hip_lib.t3d_pool_wgrad_finish(C.byref(f), stream()) == 0
    out2, t1, t2 = torch.zeros(M, K, dtype=BF, device=DEV), z(T, K), z(T, K)      # stand-alone dgrad_gram: the same kernel body either way
    d.out, d.psum_dz, d.psum_dzy = fptr(out2), fptr(t1), fptr(t2)
    assert hip_lib.t3d_pointmlp_dgrad_gram(C.byref(d), stream()) == 0
    torch.cuda.synchronize()
    assert torch.equal(dw, dw2) and float(dw.abs().max()) > 0
    assert torch.equal(out, out2) and torch.equal(s1, t1) and torch.equal(s2, t2)
