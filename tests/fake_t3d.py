"""TEST INFRASTRUCTURE ONLY -- NumPy executable specification of include/t3d.h.

Each function takes the same ctypes argument struct as the HIP entry point it mirrors and operates
on HOST memory, so that (a) the host-side step plan can be exercised on CPU tensors in this GPU-less
container and checked end to end against the oracle, and (b) every HIP kernel can be checked on the
GPU against the same inputs.  The product package never imports this module: `abi.load()` is the only
way the product obtains a library object and it raises when libt3d.so is missing.
"""
import ctypes as C
import math
import os

import numpy as np

from transferable3d_amd import abi
from transferable3d_amd.constants import MEAN_DIMS_ARR, NUM_HEADING_BIN as NH, NUM_SIZE_CLUSTER as NS

MEAN32 = MEAN_DIMS_ARR.astype(np.float32)
BINS32 = (np.arange(NH) * (2.0 * np.pi / 12.0)).astype(np.float32)


def hash_keep_mask(seed, step, n, keep):
    """The keep mask of k_dropout_mask (bn_optim.hip) / the inline generator of k_seg_head: element i from (seed, step, i)."""
    M64 = (1 << 64) - 1
    key = np.uint64(((seed << 32) ^ ((step * 0x9E3779B97F4A7C15) & M64)) & M64)
    with np.errstate(over='ignore'):
        x = key + np.arange(n, dtype=np.uint64) * np.uint64(0xD6E8FEB86659FD93)
        x = x ^ (x >> np.uint64(33)); x = x * np.uint64(0xff51afd7ed558ccd)
        x = x ^ (x >> np.uint64(33)); x = x * np.uint64(0xc4ceb9fe1a85ec53)
        x = x ^ (x >> np.uint64(33))
    r = (x >> np.uint64(16)).astype(np.uint32)
    u = (r >> np.uint32(8)).astype(np.float32) * np.float32(1.0 / 16777216.0)
    return (u < np.float32(keep)).astype(np.float32)


def _ccw(q):
    a2 = np.sum(q[:, 0] * np.roll(q[:, 1], -1) - q[:, 1] * np.roll(q[:, 0], -1))
    return q if a2 >= 0 else q[[0, 3, 2, 1]]


def _boundary_inside(P, Q, closed):
    """1/2 sum cross(a', b') over the parts of P's edges inside the convex CCW quad Q (boxgeom_dev.h)."""
    acc = 0.0
    for i in range(4):
        a, b = P[i], P[(i + 1) % 4]
        t0, t1, alive = 0.0, 1.0, True
        for e in range(4):
            q0, q1 = Q[e], Q[(e + 1) % 4]
            ex, ez = q1 - q0
            da = ex * (a[1] - q0[1]) - ez * (a[0] - q0[0])
            db = ex * (b[1] - q0[1]) - ez * (b[0] - q0[0])
            tol = 1e-5 * (ex * ex + ez * ez) + 1e-12
            ina, inb = (da >= -tol, db >= -tol) if closed else (da > tol, db > tol)
            if closed and abs(da) <= tol and abs(db) <= tol and (b[0] - a[0]) * ex + (b[1] - a[1]) * ez <= 0:
                alive = False                 # collinear edges pointing opposite ways: the rectangles only touch
            if not (ina or inb):
                alive = False
            if ina != inb:
                t = min(max(da / (da - db), 0.0), 1.0)
                if inb:
                    t0 = max(t0, t)
                else:
                    t1 = min(t1, t)
        if alive and t1 > t0:
            p, q = a + t0 * (b - a), a + t1 * (b - a)
            acc += 0.5 * (p[0] * q[1] - p[1] * q[0])
    return acc


def iou_from_quads(P, Q, ymax1, ymin1, ymax2, ymin2, vol1, vol2):
    P, Q = _ccw(np.asarray(P, np.float64)), _ccw(np.asarray(Q, np.float64))
    area = lambda q: 0.5 * abs(np.sum(q[:, 0] * np.roll(q[:, 1], -1) - q[:, 1] * np.roll(q[:, 0], -1)))
    o = P[0].copy()
    inter = max(_boundary_inside(P - o, Q - o, True) + _boundary_inside(Q - o, P - o, False), 0.0)
    iou2 = inter / (area(P) + area(Q) - inter)
    iv = inter * max(0.0, min(ymax1, ymax2) - max(ymin1, ymin2))
    return iv / (vol1 + vol2 - iv), iou2


def box3d_iou_spec(c1, s1, h1, c2, s2, h2):
    """t3d_box3d_iou for one pair in float64: (iou3d, iou2d)."""
    def rect(c, s, h):
        hl, hw = 0.5 * abs(s[0]), 0.5 * abs(s[1])
        lx, lz = np.array([hl, -hl, -hl, hl]), np.array([hw, hw, -hw, -hw])
        cs, sn = math.cos(h), math.sin(h)
        return np.stack([cs * lx + sn * lz + c[0], -sn * lx + cs * lz + c[2]], 1)
    c1, s1, c2, s2 = [np.asarray(v, np.float64) for v in (c1, s1, c2, s2)]
    a1, a2 = abs(s1[0] * s1[1]), abs(s2[0] * s2[1])
    hh1, hh2 = 0.5 * s1[2], 0.5 * s2[2]          # signed: a negative h inverts the height range -> iou3d = 0 (boxgeom_dev.h)
    return iou_from_quads(rect(c1, s1, float(h1)), rect(c2, s2, float(h2)), c1[1] + hh1, c1[1] - hh1, c2[1] + hh2, c2[1] - hh2,
                          a1 * abs(s1[2]), a2 * abs(s2[2]))


def arr(ptr, *shape):
    if not ptr:
        return None
    n = int(np.prod(shape))
    return np.ctypeslib.as_array(ptr, shape=(n,)).reshape(shape)


class AbiSizeError(ValueError):
    """What the library answers with T3D_ERR_ABI: an argument struct whose `struct_size` is not the size this side knows."""


def _struct(a):
    p = a._obj if hasattr(a, '_obj') else a.contents
    if hasattr(type(p), 'struct_size') and p.struct_size != C.sizeof(type(p)):
        raise AbiSizeError('%s: struct_size %d, expected %d' % (type(p).__name__, p.struct_size, C.sizeof(type(p))))
    return p


def _act(src, M, K, rpf):
    x = arr(src.x, M, src.ldx)[:, src.coff:src.coff + K].astype(np.float64)
    if src.scale:
        x = x * arr(src.scale, K).astype(np.float64) + arr(src.shift, K).astype(np.float64)
    if src.relu:
        x = np.maximum(x, 0.0)
    if src.sub:
        B = M // rpf
        sub = arr(src.sub, B, src.sub_ld)[:, :K].astype(np.float64)
        x = x - np.repeat(sub, rpf, axis=0)
    return x


def _dy(src, M, N, rpf):
    y = arr(src.y, M, N).astype(np.float64)
    coef = arr(src.coef, 3, N).astype(np.float64)
    if src.dz:
        dz = arr(src.dz, M, N).astype(np.float64)
    else:
        B = M // rpf
        ai = arr(src.argidx, B, N)
        dp = arr(src.dpool, B, N).astype(np.float64)
        dz = np.zeros((B, rpf, N))
        bb, cc = np.nonzero(ai >= 0)
        dz[bb, ai[bb, cc], cc] = dp[bb, cc]
        dz = dz.reshape(M, N)
    return coef[0] * dz + coef[1] * y + coef[2]


class FakeLib:
    """Drop-in for the ctypes library object (same call signatures, host pointers)."""

    def t3d_abi_version(self):
        return 3

    def t3d_gemm_arithmetic(self, arith, dtype, K, N, kind):
        """The specification library has one arithmetic (float64 products rounded once): it reports what the product's rule would take
        (kind: t3d.h T3D_GEMM_FWD / _BWD / _DGRAD / _WGRAD / _GRAM / _DGRAD_GRAM)."""
        if dtype == abi.BF16:
            return abi.ARITH_BF16
        if arith == abi.ARITH_FP32_MFMA or (arith == abi.ARITH_AUTO and os.environ.get('T3D_X3', '1') == '0'):
            return abi.ARITH_FP32_MFMA
        narrow = os.environ.get('T3D_X3_BWD_NARROW', '1') != '0'
        bwd_rule = lambda k, n: (n <= 4 * k or k >= 128 or narrow) and k <= 4096 and n <= 4096
        whole = K % 64 == 0 and (K <= 64 or K % 128 == 0)
        if kind == 0:
            ok = K % 16 == 0 and K <= 4096 and N <= 4096
        elif kind == 1:
            ok = N % 16 == 0 and whole and bwd_rule(K, N)
        elif kind == 2:
            ok = N % 16 == 0 and bwd_rule(K, N)
        elif kind == 3:
            ok = whole and bwd_rule(K, N)
        elif kind == 4:
            ok = bwd_rule(K, K)
        elif kind == 5:
            ok = K % 16 == 0 and bwd_rule(K, K)
        else:
            return -1
        return abi.ARITH_BF16X3 if ok else abi.ARITH_FP32_MFMA

    def t3d_source_hash(self, out, cap):
        return -1          # (the specification library is not a build of csrc/)

    def t3d_pointmlp_fwd(self, a, stream):
        p = _struct(a)
        M, K, N, rpf = p.M, p.K, p.N, p.rows_per_frustum
        x = _act(p.a, M, K, rpf)
        y = x @ arr(p.w, K, N).astype(np.float64)
        if p.bias:
            y = y + arr(p.bias, N)
        if p.rowbias:
            y = y + np.repeat(arr(p.rowbias, M // rpf, N).astype(np.float64), rpf, axis=0)
        y32 = y.astype(np.float32)
        if p.y:
            arr(p.y, M, N)[:] = y32
        T = M // 128
        yt = y32.astype(np.float64).reshape(T, 128, N)
        arr(p.psum, T, N)[:] = yt.sum(1)
        arr(p.psumsq, T, N)[:] = (yt * yt).sum(1)
        if p.pmax:
            keep = np.ones(M, bool) if not p.rowmask else arr(p.rowmask, M) != 0
            kt = keep.reshape(T, 128)
            tile_in_frustum = (np.arange(T) * 128) % rpf
            ymax = np.where(kt[:, :, None], y32.reshape(T, 128, N), -np.inf)
            ymin = np.where(kt[:, :, None], y32.reshape(T, 128, N), np.inf)
            anyk = kt.any(1)
            amax = ymax.argmax(1) + tile_in_frustum[:, None]
            amin = ymin.argmin(1) + tile_in_frustum[:, None]
            arr(p.pmax, T, N)[:] = ymax.max(1)
            arr(p.pmin, T, N)[:] = ymin.min(1)
            arr(p.pamax, T, N)[:] = np.where(anyk[:, None], amax, -1)
            arr(p.pamin, T, N)[:] = np.where(anyk[:, None], amin, -1)
        return 0

    def t3d_bn_fwd_finalize(self, a, stream):
        p = _struct(a)
        N = p.N
        g, b = arr(p.gamma, N).astype(np.float64), arr(p.beta, N).astype(np.float64)
        mm, mv = arr(p.moving_mean, N), arr(p.moving_var, N)
        if p.is_training:
            s = arr(p.psum, p.n_tiles, N).astype(np.float64).sum(0)
            ss = arr(p.psumsq, p.n_tiles, N).astype(np.float64).sum(0)
            n = float(p.count)
            mean = s / n
            var = np.maximum(ss / n - mean * mean, 0.0)
            d = float(arr(p.decay, 1)[0])
            var_ema = var * (n / max(n - 1, 1)) if p.unbiased_ema else var
            mm[:] = mm.astype(np.float64) * d + mean * (1 - d)
            mv[:] = mv.astype(np.float64) * d + var_ema * (1 - d)
        else:
            mean, var = mm.astype(np.float64), mv.astype(np.float64)
        invstd = 1.0 / np.sqrt(var + p.eps)
        arr(p.scale, N)[:] = g * invstd
        arr(p.shift, N)[:] = b - mean * g * invstd
        arr(p.mean, N)[:] = mean
        arr(p.invstd, N)[:] = invstd
        if p.pool_pmax:
            q = abi.PoolFinalizeArgs()
            q.scale, q.shift, q.pmax, q.pmin, q.pamax, q.pamin = p.scale, p.shift, p.pool_pmax, p.pool_pmin, p.pool_pamax, p.pool_pamin
            q.B, q.N, q.tiles_per_frustum = p.pool_B, N, p.pool_tiles_per_frustum
            q.pooled, q.ld_pooled, q.argidx, q.ysel = p.pooled, p.ld_pooled, p.argidx, p.ysel
            return self.t3d_pool_finalize(C.byref(q), stream)
        return 0

    def t3d_pool_finalize(self, a, stream):
        p = _struct(a)
        B, N, tpf = p.B, p.N, p.tiles_per_frustum
        sc, sh = arr(p.scale, N), arr(p.shift, N)
        pmax, pmin = arr(p.pmax, B, tpf, N), arr(p.pmin, B, tpf, N)
        pamax, pamin = arr(p.pamax, B, tpf, N), arr(p.pamin, B, tpf, N)
        vmax = np.where(pamax >= 0, pmax, -np.inf)
        vmin = np.where(pamin >= 0, pmin, np.inf)
        tmax, tmin = vmax.argmax(1), vmin.argmin(1)
        bmax = np.take_along_axis(vmax, tmax[:, None, :], 1)[:, 0]
        bmin = np.take_along_axis(vmin, tmin[:, None, :], 1)[:, 0]
        amax = np.take_along_axis(pamax, tmax[:, None, :], 1)[:, 0]
        amin = np.take_along_axis(pamin, tmin[:, None, :], 1)[:, 0]
        use_max = sc >= 0
        best = np.where(use_max, bmax, bmin)
        arg = np.where(use_max, amax, amin)
        valid = arg >= 0
        with np.errstate(invalid='ignore'):
            out = np.where(valid, np.maximum(np.where(valid, best, 0.0).astype(np.float32) * sc + sh, 0.0), 0.0).astype(np.float32)
        live = out > 0
        arr(p.pooled, B, p.ld_pooled)[:, :N] = out
        arr(p.argidx, B, N)[:] = np.where(live, arg, -1)
        arr(p.ysel, B, N)[:] = np.where(live, np.where(valid, best, 0.0), 0.0)
        return 0

    def t3d_pointmlp_dgrad(self, a, stream):
        p = _struct(a)
        M, K, N, rpf = p.M, p.K, p.N, p.rows_per_frustum
        dy = _dy(p.dy, M, N, rpf)
        da = dy @ arr(p.w, K, N).astype(np.float64).T
        if p.add_in:
            da = da + arr(p.add_in, M, K)
        if p.prev_y:
            yp = arr(p.prev_y, M, K).astype(np.float64)
            z = yp * arr(p.prev_scale, K) + arr(p.prev_shift, K)
            da = np.where(z > 0, da, 0.0)
        out32 = da.astype(np.float32)
        arr(p.out, M, K)[:] = out32
        if p.psum_dz:
            T = M // 128
            o = out32.astype(np.float64).reshape(T, 128, K)
            arr(p.psum_dz, T, K)[:] = o.sum(1)
            arr(p.psum_dzy, T, K)[:] = (o * yp.reshape(T, 128, K)).sum(1)
        return 0

    # ---- Gram-form backward of a pooled layer (t3d.h K11e) -------------------------------------------------
    def t3d_pool_bwd_prep(self, a, stream):
        p = _struct(a)
        K, N = p.K, p.N
        w = arr(p.w, K, N).astype(np.float64)
        c = arr(p.coef, 3, N).astype(np.float64)
        bias = arr(p.bias, N).astype(np.float64) if p.bias else np.zeros(N)
        nch = (N + 127) // 128
        ps, rs = arr(p.p_slabs, nch, K, K), arr(p.rc_slabs, nch, K)
        for ch in range(nch):
            sl = slice(ch * 128, min(N, (ch + 1) * 128))
            ps[ch] = (w[:, sl] * c[1, sl]) @ w[:, sl].T
            rs[ch] = w[:, sl] @ (bias[sl] * c[1, sl] + c[2, sl])
        if p.wc:
            arr(p.wc, N, K)[:] = (w * c[0]).T
        return 0

    def t3d_pool_sparse_rows(self, a, stream):
        p = _struct(a)
        B, N, K, rpf = p.B, p.N, p.K, p.rows_per_frustum
        ai, dp = arr(p.argidx, B, N), arr(p.dpool, B, N).astype(np.float64)
        wc = arr(p.wc, N, K).astype(np.float64)
        s = np.zeros((B * rpf, K))
        for b in range(B):
            for n in np.nonzero(ai[b] >= 0)[0]:
                s[b * rpf + ai[b, n]] += dp[b, n] * wc[n]
        if p.row_live:                 # rows without a hit are not written: poison them so that a reader that ignores the flags shows
            live = np.zeros(B * rpf, np.int32)
            for b in range(B):
                live[b * rpf + ai[b][ai[b] >= 0]] = 1
            arr(p.row_live, B * rpf)[:] = live
            s[live == 0] = np.nan
        arr(p.s, B * rpf, K)[:] = s
        return 0

    def t3d_pointmlp_dgrad_gram(self, a, stream):
        p = _struct(a)
        M, K, rpf = p.M, p.K, p.rows_per_frustum
        da = _act(p.a, M, K, rpf) @ arr(p.p, K, K).astype(np.float64)
        if p.rowconst:
            da = da + arr(p.rowconst, K)
        if p.add_in and p.add_live:
            da = da + np.where(arr(p.add_live, M)[:, None] != 0, arr(p.add_in, M, K), 0.0)
        elif p.add_in:
            da = da + arr(p.add_in, M, K)
        elif p.add_live:
            return -1
        if p.prev_y:
            yp = arr(p.prev_y, M, K).astype(np.float64)
            da = np.where(yp * arr(p.prev_scale, K) + arr(p.prev_shift, K) > 0, da, 0.0)
        out32 = da.astype(np.float32)
        arr(p.out, M, K)[:] = out32
        if p.psum_dz:
            T = M // 128
            o = out32.astype(np.float64).reshape(T, 128, K)
            arr(p.psum_dz, T, K)[:] = o.sum(1)
            arr(p.psum_dzy, T, K)[:] = (o * yp.reshape(T, 128, K)).sum(1)
        return 0

    def t3d_pointmlp_gram(self, a, stream):
        p = _struct(a)
        M, K, rps = p.M, p.K, p.rows_per_split
        x = _act(p.a, M, K, p.rows_per_frustum)
        slabs = arr(p.slabs, M // rps, K, K)
        for s in range(M // rps):
            slabs[s] = x[s * rps:(s + 1) * rps].T @ x[s * rps:(s + 1) * rps]
        return 0

    def t3d_act_colsum(self, a, stream):
        p = _struct(a)
        M, K = p.M, p.K
        arr(p.part, M // 128, K)[:] = _act(p.a, M, K, p.rows_per_frustum).reshape(M // 128, 128, K).sum(1)
        return 0

    def t3d_pool_wgrad_finish(self, a, stream):
        p = _struct(a)
        B, K, N, rpf = p.B, p.K, p.N, p.rows_per_frustum
        x = _act(p.a, B * rpf, K, rpf)
        ai, dp = arr(p.argidx, B, N), arr(p.dpool, B, N).astype(np.float64)
        c = arr(p.coef, 3, N).astype(np.float64)
        w = arr(p.w, K, N).astype(np.float64)
        bias = arr(p.bias, N).astype(np.float64) if p.bias else np.zeros(N)
        g = arr(p.g, K, K).astype(np.float64)
        abar = arr(p.abar, K).astype(np.float64)
        rows = x[(np.arange(B)[:, None] * rpf + np.maximum(ai, 0)).reshape(-1)].reshape(B, N, K)
        gat = (np.where(ai >= 0, dp, 0.0)[:, :, None] * rows).sum(0).T                        # [K,N]
        arr(p.dw, K, N)[:] = c[1] * (g @ w + np.outer(abar, bias)) + np.outer(abar, c[2]) + c[0] * gat
        return 0

    def t3d_pointmlp_wgrad(self, a, stream):
        p = _struct(a)
        M, K, N, rpf, rps = p.M, p.K, p.N, p.rows_per_frustum, p.rows_per_split
        x = _act(p.a, M, K, rpf)
        dy = _dy(p.dy, M, N, rpf)
        S = M // rps
        slabs = arr(p.slabs, S, K, N)
        for s in range(S):
            slabs[s] = x[s * rps:(s + 1) * rps].T @ dy[s * rps:(s + 1) * rps]
        return 0

    def t3d_batch_assemble(self, a, stream):
        """Vectorised restatement of csrc/data.hip (the hash generator included, in uint64 NumPy arithmetic)."""
        p = _struct(a)
        B, N, Cc, Cs = p.B, p.N, p.C, p.C_src
        step = int(arr(p.hyper, 1)[0]) if p.hyper else 0
        M64 = (1 << 64) - 1

        def mix(x):
            x = np.asarray(x, dtype=np.uint64)
            with np.errstate(over='ignore'):
                x = x ^ (x >> np.uint64(33)); x = x * np.uint64(0xff51afd7ed558ccd)
                x = x ^ (x >> np.uint64(33)); x = x * np.uint64(0xc4ceb9fe1a85ec53)
                x = x ^ (x >> np.uint64(33))
            return (x >> np.uint64(16)).astype(np.uint32)

        def u01(r):
            return ((r >> np.uint32(8)).astype(np.float32) + np.float32(0.5)) * np.float32(1.0 / 16777216.0)
        key = ((p.seed << 32) ^ ((step * 0x9E3779B97F4A7C15) & M64)) & M64
        off = np.ctypeslib.as_array(p.offsets, shape=(1 << 30,))
        from_first = True
        if p.sample2:
            from_first = step % 2 == 0
            lst = arr(p.sample, p.sample_len) if from_first else arr(p.sample2, p.sample2_len)
            sample = lst[((step // 2) * B + np.arange(B)) % len(lst)]
        elif p.sample_len > 0:
            perm = arr(p.sample, p.sample_len)
            sample = perm[(step * B + np.arange(B)) % p.sample_len]
        else:
            sample = arr(p.sample, B)
        F = int(sample.max()) + 1
        total = int(off[F])
        pts, seg = arr(p.points, total, Cs), arr(p.seg, total)
        fang, bc, head, size, cls = arr(p.frustum_angle, F), arr(p.box_center, F, 3), arr(p.heading, F), arr(p.size, F, 3), arr(p.cls, F)
        ld = p.ld_pc if p.ld_pc > 0 else Cc
        pc_full = arr(p.pc, B, N, ld)
        pc_full[:, :, Cc:] = 0
        pc, yseg = pc_full[:, :, :Cc], arr(p.y_seg, B, N)
        PI = np.float32(np.pi)
        for b in range(B):
            f = int(sample[b])
            rot = PI * np.float32(0.5) + fang[f]
            c, s_ = (np.cos(rot), np.sin(rot)) if p.rotate_to_center else (np.float32(1), np.float32(0))
            cen = np.array([bc[f, 0] * c - bc[f, 2] * s_, bc[f, 1], bc[f, 0] * s_ + bc[f, 2] * c], np.float32)
            heading = head[f] - rot if p.rotate_to_center else head[f]
            if p.sample2:
                is2d = bool(from_first)
            elif p.frustum_is_2D:
                is2d = bool(np.ctypeslib.as_array(p.frustum_is_2D, shape=(1 << 30,))[f])
            else:
                is2d = bool(arr(p.slot_is_2D, B)[b]) if p.slot_is_2D else False
            if p.aug:
                flip, rn, hu = arr(p.aug, B, 3)[b]
            else:
                kb = (key + (b + 1) * 0xA24BAED4963EE407) & M64
                flip = np.float32(1.0) if u01(mix((kb + 1) & M64)) > 0.5 else np.float32(0.0)
                rn = np.sqrt(np.float32(-2.0) * np.log(u01(mix((kb + 2) & M64)))) * np.cos(np.float32(2.0) * PI * u01(mix((kb + 3) & M64)))
                hu = u01(mix((kb + 4) & M64))
            flipx = np.float32(1)
            if p.random_flip and not is2d and flip != 0:
                flipx, cen[0], heading = np.float32(-1), -cen[0], PI - heading
            shift = hs = np.float32(0)
            if p.random_shift and not is2d:
                dist = np.sqrt(cen[0] * cen[0] + cen[1] * cen[1])
                shift = np.float32(min(max(rn * dist * np.float32(0.05), dist * np.float32(0.8)), dist * np.float32(1.2)))
                hs = np.float32(hu * np.float32(0.4) - np.float32(0.2))
                cen[2] += shift
                cen[1] += hs
            lo, cnt = int(off[f]), int(off[f + 1] - off[f])
            if p.choice:
                ch = arr(p.choice, B, N)[b].astype(np.int64)
            else:
                idx = (key + (np.uint64(b * N) + np.arange(N, dtype=np.uint64) + np.uint64(17)) * np.uint64(0xD6E8FEB86659FD93))
                ch = ((mix(idx).astype(np.uint64) * np.uint64(cnt)) >> np.uint64(32)).astype(np.int64)
            src = pts[lo + ch]
            pc[b, :, 0] = (src[:, 0] * c - src[:, 2] * s_) * flipx
            pc[b, :, 1] = src[:, 1] + hs
            pc[b, :, 2] = src[:, 0] * s_ + src[:, 2] * c + shift
            pc[b, :, 3:] = src[:, 3:Cc]
            yseg[b] = seg[lo + ch]
            two_pi, per = np.float32(2) * PI, np.float32(2) * PI / np.float32(12)
            ang = np.fmod(np.float32(heading), two_pi)
            ang = ang + two_pi if ang < 0 else ang
            sh = np.fmod(ang + per * np.float32(0.5), two_pi)
            cid = min(int(sh / per), 11)
            arr(p.y_orient_cls, B)[b] = cid
            arr(p.y_orient_reg, B)[b] = sh - (np.float32(cid) * per + per * np.float32(0.5))
            arr(p.y_dims_cls, B)[b] = cls[f]
            arr(p.y_center, B, 3)[b] = cen
            arr(p.y_dims_reg, B, 3)[b] = size[f] - MEAN32[cls[f]]
            oh = arr(p.one_hot, B, 10)
            oh[b] = 0
            oh[b, cls[f]] = 1
            if p.rot_angle:
                arr(p.rot_angle, B)[b] = rot
            if p.cam_rtilt:
                F_ = len(cls)
                arr(p.Rtilt, B, 9)[b] = arr(p.cam_rtilt, F_, 9)[f]
                arr(p.K, B, 9)[b] = arr(p.cam_k, F_, 9)[f]
                arr(p.box2D, B, 4)[b] = arr(p.cam_box2d, F_, 4)[f]
                arr(p.img_dim, B, 2)[b] = arr(p.cam_img_dim, F_, 2)[f]
            if is2d:                        # get_classes2D: every 3-D label is zero
                yseg[b] = 0
                arr(p.y_orient_cls, B)[b] = 0
                arr(p.y_orient_reg, B)[b] = 0
                arr(p.y_dims_cls, B)[b] = 0
                arr(p.y_center, B, 3)[b] = 0
                arr(p.y_dims_reg, B, 3)[b] = 0
            if p.is_data_2D:
                arr(p.is_data_2D, B)[b] = 1 if is2d else 0
        return 0

    def t3d_sample_equal_classes(self, a, stream):
        """csrc/data.hip k_sample_equal_classes (the hash generator included)."""
        p = _struct(a)
        B = p.B
        step = int(arr(p.hyper, 1)[0])
        M64 = (1 << 64) - 1

        def mix(x):
            x = np.asarray(x, dtype=np.uint64)
            with np.errstate(over='ignore'):
                x = x ^ (x >> np.uint64(33)); x = x * np.uint64(0xff51afd7ed558ccd)
                x = x ^ (x >> np.uint64(33)); x = x * np.uint64(0xc4ceb9fe1a85ec53)
                x = x ^ (x >> np.uint64(33))
            return (x >> np.uint64(16)).astype(np.uint32)

        def u01(r):
            return ((r >> np.uint32(8)).astype(np.float32) + np.float32(0.5)) * np.float32(1.0 / 16777216.0)
        two = p.set[1].n_groups > 0
        which = step % 2 if two else 0
        g = p.set[which]
        key = ((((p.seed << 32) ^ ((step * 0x9E3779B97F4A7C15) & M64)) & M64) + 0x632BE59BD9B4E019) & M64
        flag = 1 if (two and which == 0) else 0
        pu = arr(p.prob_draw, 1)[0] if p.prob_draw else u01(mix((key + 7) & M64))
        out = arr(p.sample, B)
        if not (pu < np.float32(p.equal_prob)):
            s_set = step // 2 if two else step
            perm = arr(g.perm, g.perm_len)
            for b in range(B):
                out[b] = perm[(s_set * B + b) % g.perm_len]
        else:
            n = g.n_groups
            keys = arr(p.order_draws, n).copy() if p.order_draws else \
                np.array([u01(mix((key + (i + 1) * 0xA24BAED4963EE407) & M64)) for i in range(n)], np.float32)
            rank = [sum(1 for j in range(n) if keys[j] < keys[i] or (keys[j] == keys[i] and j < i)) for i in range(n)]
            sizes = [B // n + (1 if rank[i] < B % n else 0) for i in range(n)]
            off = arr(g.offsets, n + 1)
            b = 0
            for i in range(n):
                lo, ln = int(off[i]), int(off[i + 1] - off[i])
                mem = arr(g.members, int(off[n]))
                for _ in range(sizes[i]):
                    u = arr(p.member_draws, B)[b] if p.member_draws else u01(mix((key + (b + 1) * 0xD6E8FEB86659FD93) & M64))
                    out[b] = mem[lo + min(int(np.float32(u) * np.float32(ln)), ln - 1)]
                    b += 1
        if p.is_data_2D:
            arr(p.is_data_2D, B)[:] = flag
        return 0

    @staticmethod
    def perturb_candidate_draws(seed, step, B, max_rounds):
        """The uniforms csrc/data.hip (k_boxpc_perturb) generates: (fit_draw [B], cand_draws [B, max_rounds*64, 7])."""
        M64 = (1 << 64) - 1

        def mix(x):
            x = np.asarray(x, dtype=np.uint64)
            with np.errstate(over='ignore'):
                x = x ^ (x >> np.uint64(33)); x = x * np.uint64(0xff51afd7ed558ccd)
                x = x ^ (x >> np.uint64(33)); x = x * np.uint64(0xc4ceb9fe1a85ec53)
                x = x ^ (x >> np.uint64(33))
            return (x >> np.uint64(16)).astype(np.uint32)

        def u01(r):
            return ((r >> np.uint32(8)).astype(np.float32) + np.float32(0.5)) * np.float32(1.0 / 16777216.0)
        T = max_rounds * 64
        fit, cand = np.zeros(B, np.float32), np.zeros((B, T, 7), np.float32)
        idx = (np.arange(T, dtype=object)[:, None] * 8 + np.arange(7, dtype=object)[None, :] + 101)
        for b in range(B):
            key = ((((seed << 32) ^ ((step * 0x9E3779B97F4A7C15) & M64)) & M64) + (b + 1) * 0xC2B2AE3D27D4EB4F) & M64
            fit[b] = u01(mix((key + 11) & M64))
            ks = np.array([[(key + int(v) * 0xD6E8FEB86659FD93) & M64 for v in row] for row in idx], dtype=np.uint64)
            cand[b] = u01(mix(ks))
        return fit, cand

    def t3d_boxpc_perturb(self, a, stream):
        """csrc/data.hip k_boxpc_perturb in float32 (the same operation order), first accepted candidate of the stream."""
        p = _struct(a)
        B, T = p.B, p.max_rounds * 64
        f32 = np.float32
        if p.fit_draw and p.cand_draws:
            fit_u, cand = arr(p.fit_draw, B), arr(p.cand_draws, B, T, 7)
        else:
            step = int(arr(p.hyper, 1)[0])
            gf, gc = self.perturb_candidate_draws(p.seed, step, B, p.max_rounds)
            fit_u = arr(p.fit_draw, B) if p.fit_draw else gf
            cand = arr(p.cand_draws, B, T, 7) if p.cand_draws else gc
        per = f32(2.0) * f32(np.pi) / f32(12.0)
        two_pi = f32(2.0) * f32(np.pi)
        for b in range(B):
            cls = int(arr(p.dims_cls, B)[b])
            c = arr(p.center, B, 3)[b].astype(f32).copy()
            size = (MEAN32[cls] + arr(p.dims_reg, B, 3)[b]).astype(f32)
            heading = f32(arr(p.orient_cls, B)[b]) * per + f32(arr(p.orient_reg, B)[b])
            fit = fit_u[b] < f32(p.proportion_fit)
            lo, hi = (f32(p.fit_lo), f32(p.fit_hi)) if fit else (f32(p.nofit_lo), f32(p.nofit_hi))
            scale = f32(1.0) - f32(0.5) * (lo + hi)
            cp, sp, ap = f32(p.center_perturbation) * scale, f32(p.size_perturbation) * scale, f32(p.angle_perturbation) * scale
            chosen = None
            for t in range(T):
                u = cand[b, t].astype(f32)
                dc = (f32(2.0) * u[0:3] - f32(1.0)) * cp
                ds = size * ((f32(2.0) * u[3:6] - f32(1.0)) * sp)
                da = u[6] * ap
                iou, _ = box3d_iou_spec(c, size, heading, c + dc, size + ds, heading + da)
                iou = f32(iou)
                if iou > lo and iou < hi:
                    chosen = (dc, ds, da, iou)
                    break
                if t == (p.max_rounds - 1) * 64:       # the fallback: lane 0 of the last round
                    fallback = (dc, ds, da, iou)
            dc, ds, da, iou = chosen if chosen is not None else fallback
            a_ = np.fmod(heading + da, two_pi)
            if a_ < 0:
                a_ += two_pi
            sh = np.fmod(f32(a_) + per * f32(0.5), two_pi)
            cid = min(int(sh / per), 11)
            arr(p.orient_cls, B)[b] = cid
            arr(p.orient_reg, B)[b] = sh - (f32(cid) * per + per * f32(0.5))
            arr(p.center, B, 3)[b] = c + dc
            arr(p.dims_reg, B, 3)[b] = (size + ds) - MEAN32[cls]
            arr(p.y_center_delta, B, 3)[b] = dc
            arr(p.y_dims_delta, B, 3)[b] = ds
            arr(p.y_orient_delta, B)[b] = da
            arr(p.y_box_iou, B)[b] = iou
        return 0

    def t3d_box_refine_step(self, a, stream):
        p = _struct(a)
        B = p.B
        o = arr(p.out9, B, 9).astype(np.float64)
        z = o[:, 7:9] - o[:, 7:9].max(1, keepdims=True)
        pfit = np.exp(z[:, 1]) / np.exp(z).sum(1)
        w = (1.0 - pfit) ** int(p.weigh_by_conf)
        if p.fit_prob:
            arr(p.fit_prob, B)[:] = pfit
        d = o[:, :7] * w[:, None]
        cin, din, tin = arr(p.center_in, B, 3).astype(np.float64), arr(p.dims_in, B, 3).astype(np.float64), arr(p.theta_in, B).astype(np.float64)
        arr(p.center_out, B, 3)[:] = cin - d[:, 0:3]
        arr(p.dims_out, B, 3)[:] = din - d[:, 3:6]
        arr(p.theta_out, B)[:] = tin - d[:, 6]
        tot = arr(p.total, B, 7)
        tot[:] = d if p.first else tot + d
        return 0

    def t3d_box_refine_step_bwd(self, a, stream):
        p = _struct(a)
        B = p.B
        tot = arr(p.dbox_rep, B, 7).astype(np.float64)
        if p.carry:
            tot = tot + arr(p.carry, B, 7)
        arr(p.tot_out, B, 7)[:] = tot
        if not p.out9:
            return 0
        o = arr(p.out9, B, 9).astype(np.float64)
        z = o[:, 7:9] - o[:, 7:9].max(1, keepdims=True)
        pf = np.exp(z[:, 1]) / np.exp(z).sum(1)
        q, n = 1.0 - pf, int(p.weigh_by_conf)
        w = q ** n
        g = arr(p.dout9, B, 9)
        g[:, :7] = -w[:, None] * tot
        t = np.zeros(B)
        if p.grad_via_conf and n > 0:
            dot = -(tot * o[:, :7]).sum(1)
            t = dot * (-n * q ** (n - 1)) * pf * q
        g[:, 7], g[:, 8] = -t, t
        return 0

    def t3d_pool_bwd_mid(self, slab_base, grad_base, table, n, mx, sparse, stream):
        return self.t3d_reduce_slabs(slab_base, grad_base, table, n, mx, stream) or self.t3d_pool_sparse_rows(sparse, stream)

    def t3d_pool_bwd_stage1(self, g, c, q, stream):
        return self.t3d_pointmlp_gram(g, stream) or self.t3d_act_colsum(c, stream) or self.t3d_pool_bwd_prep(q, stream)

    def t3d_pool_bwd_stage2(self, f, d, stream):
        return self.t3d_pool_wgrad_finish(f, stream) or self.t3d_pointmlp_dgrad_gram(d, stream)

    def t3d_pointmlp_bwd(self, d, w, stream):
        return self.t3d_pointmlp_dgrad(d, stream) or self.t3d_pointmlp_wgrad(w, stream)

    def t3d_gram_plan(self, M, K, dtype, rps, one):
        one._obj.value = 0
        if dtype == 1 and K in (128, 256) and M % 128 == 0:
            tiles, per = M // 128, 1
            while tiles // per > 256 and tiles % (per * 2) == 0:
                per *= 2
            if dtype == 1 or tiles < 256 or 128 * per >= 256:
                rps._obj.value, one._obj.value = 128 * per, 1
                return 0
        tk, tn = C.c_int(0), C.c_int(0)
        return self.t3d_wgrad_plan(M, K, K, rps, C.byref(tk), C.byref(tn))

    def t3d_bwd_plan(self, M, K, N, dtype, rps, one):
        one._obj.value = 0
        narrow = K in (64, 128) and N in (64, 128)
        if ((dtype == 1 and (narrow or (K, N) in ((256, 128), (128, 256)))) or (dtype == 0 and narrow)) and M % 128 == 0:
            tiles, per = M // 128, 1
            while tiles // per > 256 and tiles % (per * 2) == 0:
                per *= 2
            if dtype == 1 or tiles < 256 or 128 * per >= 256:
                rps._obj.value, one._obj.value = 128 * per, 1
                return 0
        tk, tn = C.c_int(0), C.c_int(0)
        return self.t3d_wgrad_plan(M, K, N, rps, C.byref(tk), C.byref(tn))

    def t3d_wgrad_plan(self, M, K, N, rps, tk, tn):
        cap = max(64, (1 << 21) // (K * N))
        cap = max(1, min(cap, M // 128))
        ck, cn, tiles = 64, 64, ((K + 63) // 64) * (N // 64)
        for a_, b_ in ((128, 128), (128, 64), (64, 128), (64, 64)):
            if (a_ == 128 and K <= 64) or N % b_:
                continue
            t = ((K + a_ - 1) // a_) * (N // b_)
            if t * cap >= 512:
                ck, cn, tiles = a_, b_, t
                break
        want = min((512 + tiles - 1) // tiles, cap)
        s = 1
        while s * 2 <= want and M % (s * 2) == 0 and (M // (s * 2)) % 32 == 0:
            s *= 2
        rps._obj.value, tk._obj.value, tn._obj.value = M // s, ck, cn
        return 0

    def t3d_bn_bwd_finalize(self, a, stream):
        p = _struct(a)
        N = p.N
        if p.psum_dz:
            s1 = arr(p.psum_dz, p.n_tiles, N).astype(np.float64).sum(0)
            s2 = arr(p.psum_dzy, p.n_tiles, N).astype(np.float64).sum(0)
        else:
            B = p.B
            live = arr(p.pooled, B, p.ld_pooled)[:, :N] > 0
            g = (arr(p.dpool_in, B, p.ld_dpool_in)[:, :N] * live).astype(np.float32)
            arr(p.dpool, B, N)[:] = g
            s1 = g.astype(np.float64).sum(0)
            s2 = (g.astype(np.float64) * arr(p.ysel, B, N)).sum(0)
        coef = arr(p.coef, 3, N)
        if p.frozen:
            coef[0] = arr(p.scale, N)
            coef[1] = 0
            coef[2] = 0
            return 0
        mean, invstd = arr(p.mean, N).astype(np.float64), arr(p.invstd, N).astype(np.float64)
        gamma, n = arr(p.gamma, N).astype(np.float64), float(p.count)
        dbeta = s1
        dgamma = invstd * (s2 - mean * s1)
        if p.dbeta:
            arr(p.dbeta, N)[:] = dbeta
        if p.dgamma:
            arr(p.dgamma, N)[:] = dgamma
        c1 = gamma * invstd
        k3 = dgamma / n * invstd
        coef[0] = c1
        coef[1] = -c1 * k3
        coef[2] = c1 * (k3 * mean - dbeta / n)
        return 0

    def t3d_dy_colsum(self, a, stream):
        p = _struct(a)
        B, N, tpf = p.B, p.N, p.tiles_per_frustum
        sdz = arr(p.psum_dz, B, tpf, N).astype(np.float64).sum(1)
        sy = arr(p.psum_y, B, tpf, N).astype(np.float64).sum(1)
        coef = arr(p.coef, 3, N).astype(np.float64)
        arr(p.out, B, N)[:] = p.alpha * (coef[0] * sdz + coef[1] * sy + coef[2] * p.rows_per_frustum)
        return 0

    # ---- FC --------------------------------------------------------------------------------------
    @staticmethod
    def _fc_in(p):
        x = arr(p.in_, p.B, p.ld_in)[:, :p.K].astype(np.float64)
        if p.K2 > 0:
            x = np.concatenate([x, arr(p.in2, p.B, p.ld_in2)[:, :p.K2].astype(np.float64)], 1)
        return x

    @staticmethod
    def _actf(z, act, alpha):
        if act == abi.ACT_RELU:
            return np.maximum(z, 0)
        if act == abi.ACT_LEAKY_RELU:
            return np.where(z > 0, z, alpha * z)
        if act == abi.ACT_TANH:
            return np.tanh(z)
        return z

    @staticmethod
    def _actd(z, act, alpha):
        if act == abi.ACT_RELU:
            return (z > 0).astype(np.float64)
        if act == abi.ACT_LEAKY_RELU:
            return np.where(z > 0, 1.0, alpha)
        if act == abi.ACT_TANH:
            return 1 - np.tanh(z) ** 2
        return np.ones_like(z)

    def t3d_fc_fwd(self, a, stream):
        p = _struct(a)
        B, N = p.B, p.N
        x = self._fc_in(p)
        y = x @ arr(p.w, p.K + p.K2, N).astype(np.float64) if p.w else x.copy()      # w == NULL: identity (standalone BN / dropout)
        if p.bias:
            y = y + arr(p.bias, N)
        y = y.astype(np.float32).astype(np.float64)
        if p.y:
            arr(p.y, B, N)[:] = y
        z = y
        if p.gamma:
            mm, mv = arr(p.moving_mean, N), arr(p.moving_var, N)
            if p.is_training:
                mean = y.mean(0)
                var = ((y - mean) ** 2).mean(0)
                d = float(arr(p.decay, 1)[0])
                var_ema = var * (B / max(B - 1, 1)) if p.unbiased_ema else var
                mm[:] = mm * d + mean * (1 - d)
                mv[:] = mv * d + var_ema * (1 - d)
            else:
                mean, var = mm.astype(np.float64), mv.astype(np.float64)
            invstd = 1 / np.sqrt(var + p.eps)
            arr(p.mean, N)[:] = mean
            arr(p.invstd, N)[:] = invstd
            z = (y - mean) * invstd * arr(p.gamma, N) + arr(p.beta, N)
        z = self._actf(z, p.act, p.leaky_alpha)
        if p.drop_mask:
            z = z * arr(p.drop_mask, B, N) / p.keep_prob
        if p.add_in:
            z[:, :p.add_n] += arr(p.add_in, B, p.ld_add)[:, :p.add_n]
        arr(p.out, B, p.ld_out)[:, :N] = z
        return 0

    def t3d_act_dropout(self, a, stream):
        p = _struct(a)
        v = _act(p.a, p.M, p.K, p.rows_per_frustum)
        if p.mask:
            v = v * arr(p.mask, p.M, p.K) / (p.keep_prob if p.keep_prob < 1.0 else 1.0)
        arr(p.out, p.M, p.K)[:] = v
        return 0

    def t3d_fc_bwd(self, a, stream):
        p = _struct(a)
        B, N = p.B, p.N
        if p.dout:
            g = arr(p.dout, B, p.ld_dout)[:, :N].astype(np.float64)
        else:
            wn = arr(p.w_next, N, p.N_next).astype(np.float64)
            g = arr(p.dy_next, B, p.N_next).astype(np.float64) @ wn.T
        y = arr(p.y, B, N).astype(np.float64) if p.y else np.zeros((B, N))
        bn = bool(p.gamma)
        if bn:
            mean, invstd = arr(p.mean, N).astype(np.float64), arr(p.invstd, N).astype(np.float64)
            gam = arr(p.gamma, N).astype(np.float64)
            xh = (y - mean) * invstd
            z = xh * gam + arr(p.beta, N)
        else:
            xh = y
            z = y
        if p.drop_mask:
            g = g * arr(p.drop_mask, B, N) / p.keep_prob
        dz = g * self._actd(z, p.act, p.leaky_alpha)
        dbias = np.zeros(N)
        if bn and p.bn_training:
            dbeta, dgamma = dz.sum(0), (dz * xh).sum(0)
            if p.dbeta:
                arr(p.dbeta, N)[:] = dbeta
            if p.dgamma:
                arr(p.dgamma, N)[:] = dgamma
            dy = gam * invstd * (dz - dbeta / B - xh * dgamma / B)
        elif bn:
            dy = gam * invstd * dz
        else:
            dy = dz
            dbias = dz.sum(0)
        if p.dbias:
            arr(p.dbias, N)[:] = dbias
        dy32 = dy.astype(np.float32)
        arr(p.dy, B, N)[:] = dy32
        if p.dw:
            arr(p.dw, p.K + p.K2, N)[:] = self._fc_in(p).T @ dy32.astype(np.float64)
        return 0

    def t3d_fc_dinput(self, a, stream):
        p = _struct(a)
        v = p.alpha * (arr(p.dy, p.B, p.N).astype(np.float64) @ arr(p.w, p.K, p.N).astype(np.float64).T)
        if p.add_in:
            v = v + arr(p.add_in, p.B, p.ld_add)[:, :p.K]
        arr(p.din, p.B, p.ld_din)[:, :p.K] = v
        if p.bn_coef:                  # fused pooled-form batch-norm backward statistics on din's columns
            q = abi.BnBwdFinalizeArgs()
            q.dpool_in, q.ld_dpool_in, q.pooled, q.ld_pooled, q.ysel, q.dpool, q.B = p.din, p.ld_din, p.bn_pooled, p.bn_ld_pooled, \
                p.bn_ysel, p.bn_dpool, p.B
            q.count, q.N, q.gamma, q.mean, q.invstd, q.scale, q.frozen = p.bn_count, p.K, p.bn_gamma, p.bn_mean, p.bn_invstd, \
                p.bn_scale, p.bn_frozen
            q.dgamma, q.dbeta, q.coef = p.bn_dgamma, p.bn_dbeta, p.bn_coef
            return self.t3d_bn_bwd_finalize(C.byref(q), stream)
        return 0

    # ---- heads -----------------------------------------------------------------------------------
    def t3d_seg_head(self, a, stream):
        p = _struct(a)
        M, K, rpf, B = p.M, p.K, p.rows_per_frustum, p.B
        T = M // 128
        y = arr(p.y, M, K).astype(np.float64)
        z = y * arr(p.scale, K) + arr(p.shift, K)
        keep = np.full((M, K), 1.0)
        if p.drop_mask:
            keep = arr(p.drop_mask, M, K).astype(np.float64) / p.keep_prob
        elif p.drop_hyper and p.keep_prob < 1.0:
            keep = hash_keep_mask(p.drop_seed, int(arr(p.drop_hyper, 1)[0]), M * K, p.keep_prob).reshape(M, K).astype(np.float64) / p.keep_prob
        d = np.maximum(z, 0) * keep
        w = arr(p.w, K, 2).astype(np.float64)
        logits = (d @ w + arr(p.bias, 2)).astype(np.float32)
        if p.oracle_mask:      # semisup_v1_sunrgbd.py:161-162
            om = arr(p.oracle_mask, M).astype(np.float32)
            logits = np.stack([1 - om, om], 1)
        arr(p.logits, M, 2)[:] = logits
        mask = (logits[:, 0] < logits[:, 1]).astype(np.float32)
        arr(p.mask, M)[:] = mask
        part = arr(p.part, T, 8)
        part[:] = 0
        xyz = arr(p.pc, M, p.ld_pc)[:, :3].astype(np.float64)
        part[:, 1] = mask.reshape(T, 128).sum(1)
        part[:, 2:5] = (mask[:, None] * xyz).reshape(T, 128, 3).sum(1)
        if p.labels:
            lab = arr(p.labels, M).astype(np.int64)
            l64 = logits.astype(np.float64)
            mx = l64.max(1, keepdims=True)
            lse = mx[:, 0] + np.log(np.exp(l64 - mx).sum(1))
            ce = lse - l64[np.arange(M), lab]
            part[:, 0] = ce.reshape(T, 128).sum(1)
            part[:, 7] = ((l64[:, 1] > l64[:, 0]).astype(np.int64) == lab).reshape(T, 128).sum(1)
            if p.dz:
                is2d = arr(p.is_data_2D, B)
                wb = np.repeat(p.ce_weight * (1 - is2d) / (B * rpf), rpf)
                g = (np.exp(l64 - lse[:, None]) - np.eye(2)[lab]) * wb[:, None]
                if p.dsoft:      # d loss / d soft_mask from t3d_weak_loss: soft = softmax(logits)[:,1]
                    p1 = np.exp(l64[:, 1] - lse)
                    gs = arr(p.dsoft, M).astype(np.float64) * p1 * (1 - p1)
                    g = g + np.stack([-gs, gs], 1)
                if p.oracle_mask:
                    g = np.zeros_like(g)
                part[:, 5:7] = g.reshape(T, 128, 2).sum(1)
                arr(p.dw_part, T, K, 2)[:] = np.einsum('tik,tij->tkj', d.reshape(T, 128, K), g.reshape(T, 128, 2))
                dz = np.where(z > 0, (g @ w.T) * keep, 0.0).astype(np.float32)
                arr(p.dz, M, K)[:] = dz
                dz64 = dz.astype(np.float64).reshape(T, 128, K)
                arr(p.psum_dz, T, K)[:] = dz64.sum(1)
                arr(p.psum_dzy, T, K)[:] = (dz64 * y.reshape(T, 128, K)).sum(1)
        return 0

    def t3d_weak_loss(self, a, stream):
        """Specification of the weak box losses: the torch-autograd restatement itself (tests/ may use oracle/)."""
        import torch
        from oracle import ref_weak as W
        p = _struct(a)
        B, N = p.B, p.N
        t64 = lambda x, *shape: torch.as_tensor(np.array(arr(x, *shape)), dtype=torch.float64)
        center, dims, theta = t64(p.center, B, 3).requires_grad_(True), t64(p.reg_dims, B, 3).requires_grad_(True), \
            t64(p.reg_theta, B).requires_grad_(True)
        is2d = torch.as_tensor(np.array(arr(p.is_data_2D, B)), dtype=torch.float64) if p.is_data_2D else torch.ones(B, dtype=torch.float64)
        box = (center, dims, theta)
        reproj = torch.zeros(B, dtype=torch.float64)
        surf = torch.zeros(B, dtype=torch.float64)
        soft = None
        if p.Rtilt:
            reproj = W.get_reprojection_loss(box, t64(p.box2D, B, 4), t64(p.Rtilt, B, 3, 3), t64(p.K, B, 3, 3), t64(p.img_dim, B, 2),
                                             t64(p.rot_frust, B), bool(p.use_softmax_proj), p.softmax_scale, p.dilate,
                                             bool(p.clip_lower_b_loss), bool(p.clip_pred_box), 'mse' if p.loss_mse else 'huber',
                                             [bool(x) for x in p.train_box_reproj])
        if p.pc:
            soft = torch.softmax(t64(p.logits, B, N, 2), -1)[:, :, 1].detach().requires_grad_(True)
            pc = t64(p.pc, B * N, p.ld_pc)[:, :3].reshape(B, N, 3)
            surf = W.get_surface_loss(box, pc, soft, p.surface_margin, p.surface_scale_dims, [bool(x) for x in p.train_box_surface])
        add = is2d * p.multiplier * (p.w_reproj * reproj + p.w_surface * surf)
        lossv = add.mean()
        if p.w_inactive != 0:
            cls = torch.as_tensor(np.array(arr(p.one_hot, B, 10)).argmax(1))
            iv = W.get_inactive_volume_loss_v1(dims, cls, [bool(x) for x in p.inactive_train],
                                               torch.as_tensor([float(x) for x in p.inactive_margins], dtype=torch.float64))
            lossv = lossv + p.multiplier * p.w_inactive * iv
            if p.inactive:
                arr(p.inactive, 1)[0] = float(iv.detach())
        ins = [center, dims, theta] + ([soft] if soft is not None else [])
        gs = torch.autograd.grad(lossv, ins, allow_unused=True) if lossv.requires_grad else [None] * len(ins)
        z = lambda g, like: (torch.zeros_like(like) if g is None else g).detach().numpy()
        arr(p.dbox7, B, 7)[:] = np.concatenate([z(gs[0], center), z(gs[1], dims), z(gs[2], theta)[:, None]], 1)
        if p.dsoft:
            arr(p.dsoft, B * N)[:] = z(gs[3], soft).reshape(-1) if soft is not None else 0.0
        if p.reproj:
            arr(p.reproj, B)[:] = reproj.detach().numpy()
        if p.surface:
            arr(p.surface, B)[:] = surf.detach().numpy()
        if p.total_losses:
            arr(p.total_losses, B)[:] += add.detach().numpy()
        arr(p.loss, 1)[0] += float(lossv.detach())
        return 0

    def t3d_seg_finalize(self, a, stream):
        p = _struct(a)
        B, tpf, rpf, K = p.B, p.tiles_per_frustum, p.rows_per_frustum, p.K
        part = arr(p.part, B, tpf, 8).astype(np.float64)
        s = part.sum(1)
        arr(p.mask_xyz_mean, B, 3)[:] = s[:, 2:5] / np.maximum(s[:, 1:2], 1.0)
        if p.seg_loss:
            arr(p.seg_loss, B)[:] = s[:, 0] / rpf
        if p.dw:
            arr(p.dw, K, 2)[:] = arr(p.dw_part, B * tpf, K, 2).astype(np.float64).sum(0)
        if p.dbias:
            arr(p.dbias, 2)[:] = part[:, :, 5:7].sum((0, 1))
        if p.n_correct:
            arr(p.n_correct, 1)[0] = part[:, :, 7].sum()
        return 0

    def t3d_strong_loss(self, a, stream):
        """Analytic restatement in float64 of the loss kernel (forward and hand-derived backward)."""
        p = _struct(a)
        B = p.B
        W = p.wts
        box = arr(p.box, B, p.ld_box)[:, :67].astype(np.float64)
        s1 = arr(p.stage1_center, B, 3).astype(np.float64)
        yc = arr(p.y_center, B, 3).astype(np.float64)
        yoc, yor = arr(p.y_orient_cls, B), arr(p.y_orient_reg, B).astype(np.float64)
        ydc, ydr = arr(p.y_dims_cls, B), arr(p.y_dims_reg, B, 3).astype(np.float64)
        w3d = (1 - arr(p.is_data_2D, B)).astype(np.float64)
        norm = 1.0 / (w3d.sum() + 1e-3) if p.normalize_by_3d_count else 1.0 / B
        seg = arr(p.seg_loss, B).astype(np.float64) if p.seg_loss else np.zeros(B)
        dbox, ds1 = arr(p.dbox, B, 67), arr(p.dstage1, B, 3)
        terms, tot = arr(p.terms, B, 8), arr(p.total_losses, B)
        mean = MEAN32.astype(np.float64)
        bins = BINS32.astype(np.float64)
        sx = np.array([1, 1, -1, -1, 1, 1, -1, -1.])
        sy = np.array([1, 1, 1, 1, -1, -1, -1, -1.])
        sz = np.array([1, -1, -1, 1, 1, -1, -1, 1.])

        def hub(e, d):
            q = min(abs(e), d)
            return 0.5 * q * q + d * (abs(e) - q)

        def ce(z, lab, g, gsc):
            m = z.max()
            lse = m + math.log(np.exp(z - m).sum())
            g += gsc * (np.exp(z - lse) - np.eye(len(z))[lab])
            return lse - z[lab]

        total_sum = 0.0
        for b in range(B):
            o = box[b]
            g = np.zeros(67)
            gc, gs1 = np.zeros(3), np.zeros(3)
            gs = w3d[b] * norm
            bm = W.box_multiplier
            cen = o[0:3] + s1[b]
            j, k = int(yoc[b]), int(ydc[b])
            dist = np.linalg.norm(yc[b] - cen)
            l_c = hub(dist, 2.0)
            if dist > 0:
                gc += gs * bm * W.center * min(dist, 2.0) / dist * (cen - yc[b])
            dist = np.linalg.norm(yc[b] - s1[b])
            l_s1 = hub(dist, 1.0)
            if dist > 0:
                gs1 += gs * bm * W.tnet_center * min(dist, 1.0) / dist * (s1[b] - yc[b])
            l_hc = ce(o[3:3 + NH], j, g[3:3 + NH], gs * bm * W.orient_cls)
            hrn = o[3 + NH + j]
            eh = hrn - yor[b] / (np.pi / NH)
            l_hr = hub(eh, 1.0)
            g[3 + NH + j] += gs * bm * W.orient_reg * np.clip(eh, -1, 1)
            l_sc = ce(o[3 + 2 * NH:3 + 2 * NH + NS], k, g[3 + 2 * NH:3 + 2 * NH + NS], gs * bm * W.dims_cls)
            so = 3 + 2 * NH + NS + 3 * k
            srn = o[so:so + 3]
            dsv = srn - ydr[b] / mean[k]
            sd = np.linalg.norm(dsv)
            l_sr = hub(sd, 1.0)
            if sd > 0:
                g[so:so + 3] += gs * bm * W.dims_reg * min(sd, 1.0) / sd * dsv
            th = bins[j] + hrn * (np.pi / NH)
            c, s = math.cos(th), math.sin(th)
            size = mean[k] + 2 * srn * mean[k]
            hl = bins[j] + yor[b]
            gl = mean[k] + ydr[b]
            l_co = 0.0
            gth = 0.0
            gsize = np.zeros(3)
            for i in range(8):
                x, y, z = sx[i] * size[0] / 2, sy[i] * size[2] / 2, sz[i] * size[1] / 2
                cp = np.array([c * x + s * z, y, -s * x + c * z]) + cen
                xg, yg, zg = sx[i] * gl[0] / 2, sy[i] * gl[2] / 2, sz[i] * gl[1] / 2
                tg = []
                for ang in (hl, hl + np.pi):
                    ca, sa = math.cos(ang), math.sin(ang)
                    tg.append(np.array([ca * xg + sa * zg, yg, -sa * xg + ca * zg]) + yc[b])
                d1, d2 = np.linalg.norm(cp - tg[0]), np.linalg.norm(cp - tg[1])
                t, dm = (tg[0], d1) if d1 <= d2 else (tg[1], d2)
                l_co += hub(dm, 1.0) / 8
                if dm > 0:
                    gv = gs * W.corner / 8 * min(dm, 1.0) / dm * (cp - t)
                    gc += gv
                    gth += gv[0] * (-s * x + c * z) + gv[2] * (-c * x - s * z)
                    gsize[0] += (gv[0] * c - gv[2] * s) * sx[i] / 2
                    gsize[2] += gv[1] * sy[i] / 2
                    gsize[1] += (gv[0] * s + gv[2] * c) * sz[i] / 2
            g[3 + NH + j] += gth * (np.pi / NH)
            g[so:so + 3] += gsize * 2 * mean[k]
            g[0:3] = gc
            box_l = bm * (W.center * l_c + W.orient_cls * l_hc + W.dims_cls * l_sc + W.orient_reg * l_hr
                          + W.dims_reg * l_sr + W.tnet_center * l_s1) + W.corner * l_co
            t_b = w3d[b] * (W.cross_entropy * seg[b] + box_l)
            total_sum += t_b
            dbox[b] = g
            ds1[b] = gc + gs1
            arr(p.center, B, 3)[b] = cen
            terms[b] = [seg[b], l_c, l_s1, l_hc, l_hr, l_sc, l_sr, l_co]
            tot[b] = t_b
            js = int(np.argmax(o[3:3 + NH]))
            ks = int(np.argmax(o[3 + 2 * NH:3 + 2 * NH + NS]))
            so2 = 3 + 2 * NH + NS + 3 * ks
            arr(p.reg_dims, B, 3)[b] = np.maximum(mean[ks] + o[so2:so2 + 3] * mean[ks], 1e-5)
            arr(p.reg_theta, B)[b] = bins[js] + o[3 + NH + js] * (np.pi / NH)
            if p.iou3d:
                sp = mean[ks] + o[so2:so2 + 3] * mean[ks]
                i3, i2 = box3d_iou_spec(cen, sp, arr(p.reg_theta, B)[b], yc[b], mean[k] + ydr[b], bins[j] + yor[b])
                arr(p.iou3d, B)[b], arr(p.iou2d, B)[b] = i3, i2
        arr(p.loss, 1)[0] = total_sum * norm
        return 0

    # ---- 3-D IoU (K13) ----------------------------------------------------------------------------------
    def t3d_box3d_iou(self, a, stream):
        p = _struct(a)
        n = p.n
        c1, s1, h1 = arr(p.center1, n, 3), arr(p.size1, n, 3), arr(p.heading1, n)
        c2, s2, h2 = arr(p.center2, n, 3), arr(p.size2, n, 3), arr(p.heading2, n)
        for i in range(n):
            i3, i2 = box3d_iou_spec(c1[i], s1[i], h1[i], c2[i], s2[i], h2[i])
            arr(p.iou3d, n)[i] = i3
            if p.iou2d:
                arr(p.iou2d, n)[i] = i2
        return 0

    def t3d_box3d_iou_corners(self, a, stream):
        p = _struct(a)
        n = p.n
        k1, k2 = arr(p.corners1, n, 8, 3).astype(np.float64), arr(p.corners2, n, 8, 3).astype(np.float64)
        vol = lambda k: np.linalg.norm(k[0] - k[1]) * np.linalg.norm(k[1] - k[2]) * np.linalg.norm(k[0] - k[4])
        for i in range(n):
            P, Q = k1[i][[3, 2, 1, 0]][:, [0, 2]], k2[i][[3, 2, 1, 0]][:, [0, 2]]
            i3, i2 = iou_from_quads(P, Q, k1[i][0, 1], k1[i][4, 1], k2[i][0, 1], k2[i][4, 1], vol(k1[i]), vol(k2[i]))
            arr(p.iou3d, n)[i] = i3
            if p.iou2d:
                arr(p.iou2d, n)[i] = i2
        return 0

    def t3d_box_head_iou(self, a, stream):
        p = _struct(a)
        B = p.B
        box = arr(p.box, B, p.ld_box)[:, :67].astype(np.float64)
        s1 = arr(p.stage1_center, B, 3).astype(np.float64) if p.stage1_center else np.zeros((B, 3))
        yc, yoc, yor = arr(p.y_center, B, 3).astype(np.float64), arr(p.y_orient_cls, B), arr(p.y_orient_reg, B).astype(np.float64)
        ydc, ydr = arr(p.y_dims_cls, B), arr(p.y_dims_reg, B, 3).astype(np.float64)
        mean, bins = MEAN32.astype(np.float64), BINS32.astype(np.float64)
        for b in range(B):
            o = box[b]
            js, ks = int(np.argmax(o[3:3 + NH])), int(np.argmax(o[3 + 2 * NH:3 + 2 * NH + NS]))
            so = 3 + 2 * NH + NS + 3 * ks
            i3, i2 = box3d_iou_spec(o[0:3] + s1[b], mean[ks] + o[so:so + 3] * mean[ks], bins[js] + o[3 + NH + js] * (np.pi / NH),
                                    yc[b], mean[int(ydc[b])] + ydr[b], bins[int(yoc[b])] + yor[b])
            arr(p.iou3d, B)[b], arr(p.iou2d, B)[b] = i3, i2
        return 0

    # ---- Box-PC --------------------------------------------------------------------------------------
    @staticmethod
    def _box(p, B):
        center = arr(p.center, B, 3).astype(np.float64)
        if p.y_dims_cls:
            k, j = arr(p.y_dims_cls, B), arr(p.y_orient_cls, B)
            dims = np.maximum(MEAN32.astype(np.float64)[k] + arr(p.dims, B, 3), 1e-5)
            theta = BINS32.astype(np.float64)[j] + arr(p.theta, B)
        else:
            dims, theta = arr(p.dims, B, 3).astype(np.float64), arr(p.theta, B).astype(np.float64)
        return center, dims, theta

    def t3d_boxpc_rep(self, a, stream):
        p = _struct(a)
        M, rpf, Cc = p.M, p.rows_per_frustum, p.C
        B = M // rpf
        center, dims, theta = self._box(p, B)
        if p.box_out:
            arr(p.box_out, B, 7)[:] = np.concatenate([center, dims, theta[:, None]], 1)
        pc = arr(p.pc, M, p.ld_pc).astype(np.float64)
        if p.rowmask:          # test_semisup.py:103-105
            pc = pc * arr(p.rowmask, M).astype(np.float64)[:, None]
        rep = arr(p.rep, M, p.ld_rep)
        rep[:] = 0
        rep[:, :Cc] = pc[:, :Cc]
        t = pc[:, :3] - np.repeat(center, rpf, 0)
        c, s = np.repeat(np.cos(theta), rpf), np.repeat(np.sin(theta), rpf)
        l, w, h = [np.repeat(dims[:, i], rpf) for i in range(3)]
        u, q = c * t[:, 0] - s * t[:, 2], s * t[:, 0] + c * t[:, 2]
        rep[:, Cc:Cc + 6] = np.stack([l / 2 - u, l / 2 + u, h / 2 - t[:, 1], h / 2 + t[:, 1], w / 2 - q, w / 2 + q], 1)
        return 0

    def t3d_boxpc_rep_bwd(self, a, stream):
        p = _struct(a)
        B, rpf = p.B, p.rows_per_frustum
        M = B * rpf
        box = arr(p.box, B, 7).astype(np.float64)
        g = arr(p.drep, M, p.ld_drep)[:, p.coff:p.coff + 6].astype(np.float64).reshape(B, rpf, 6)
        pc = arr(p.pc, M, p.ld_pc).astype(np.float64).reshape(B, rpf, -1)
        c, s = np.cos(box[:, 6])[:, None], np.sin(box[:, 6])[:, None]
        tx, tz = pc[:, :, 0] - box[:, 0:1], pc[:, :, 2] - box[:, 2:3]
        u, q = c * tx - s * tz, s * tx + c * tz
        du, dv, dq = g[:, :, 1] - g[:, :, 0], g[:, :, 3] - g[:, :, 2], g[:, :, 5] - g[:, :, 4]
        out = arr(p.dbox, B, 7)
        out[:, 0] = -(du * c + dq * s).sum(1)
        out[:, 1] = -dv.sum(1)
        out[:, 2] = -(-du * s + dq * c).sum(1)
        out[:, 3] = 0.5 * (g[:, :, 0] + g[:, :, 1]).sum(1)
        out[:, 4] = 0.5 * (g[:, :, 4] + g[:, :, 5]).sum(1)
        out[:, 5] = 0.5 * (g[:, :, 2] + g[:, :, 3]).sum(1)
        out[:, 6] = (-du * q + dq * u).sum(1)
        return 0

    def t3d_boxpc_loss(self, a, stream):
        p = _struct(a)
        B = p.B
        o = arr(p.out, B, 9).astype(np.float64)
        iou = arr(p.y_box_iou, B).astype(np.float64)
        cls = (iou > p.fit_bound).astype(np.int64)
        lg = o[:, 7:9]
        mx = lg.max(1, keepdims=True)
        lse = mx[:, 0] + np.log(np.exp(lg - mx).sum(1))
        ce = lse - lg[np.arange(B), cls]
        prob = np.exp(lg - lse[:, None])
        wl = np.ones(B)
        if p.weigh_by_cls_gt:
            wl = 1 - iou
        if p.weigh_by_cls_conf:
            wl = 1 - prob[:, 1]
        if p.delta_loss_mse:
            hub, hubd = (lambda e: e * e), (lambda e: 2 * e)
        else:
            hub = lambda e: 0.5 * np.minimum(np.abs(e), 1) ** 2 + (np.abs(e) - np.minimum(np.abs(e), 1))
            hubd = lambda e: np.clip(e, -1, 1)
        wp = 1 - prob[:, 1] if p.weigh_pred_by_cls_conf else np.ones(B)
        ec, es, ea = o[:, 0:3] * wp[:, None] - arr(p.y_center_delta, B, 3), o[:, 3:6] * wp[:, None] - arr(p.y_dims_delta, B, 3), \
            o[:, 6] * wp - arr(p.y_orient_delta, B)
        unw = p.w_center * hub(ec).mean(1) + p.w_size * hub(es).mean(1) + p.w_angle * hub(ea)
        delta = wl * unw
        total = p.w_cls * ce + p.w_delta * delta
        g = arr(p.dout, B, 9)
        g[:, 7:9] = p.w_cls * (prob - np.eye(2)[cls]) / B
        gc, gs, ga = p.w_center * hubd(ec) / 3, p.w_size * hubd(es) / 3, p.w_angle * hubd(ea)
        g[:, 0:3] = p.w_delta * (wl * wp)[:, None] * gc / B
        g[:, 3:6] = p.w_delta * (wl * wp)[:, None] * gs / B
        g[:, 6] = p.w_delta * wl * wp * ga / B
        if p.grad_cls_via_delta:
            dp1 = np.zeros(B)
            if p.weigh_by_cls_conf:
                dp1 -= unw
            if p.weigh_pred_by_cls_conf:
                dp1 -= wl * ((gc * o[:, 0:3]).sum(1) + (gs * o[:, 3:6]).sum(1) + ga * o[:, 6])
            t = p.w_delta * dp1 * prob[:, 1] * prob[:, 0] / B
            g[:, 7] -= t
            g[:, 8] += t
        arr(p.terms, B, 4)[:] = np.stack([ce, delta, prob[:, 1], total], 1)
        arr(p.loss, 1)[0] = total.mean()
        return 0

    def t3d_box2d_feats(self, a, stream):
        p = _struct(a)
        B, oh = p.B, p.n_oh
        out = arr(p.out, B, oh + 4)
        if oh:
            out[:, :oh] = arr(p.one_hot, B, oh)
        box, dim = arr(p.box2D, B, 4), arr(p.img_dim, B, 2)
        rows, cols = dim[:, 0], dim[:, 1]
        out[:, oh:] = np.stack([box[:, 0] / cols, box[:, 1] / rows, box[:, 2] / cols, box[:, 3] / rows], 1)
        return 0

    # ---- stage-c glue --------------------------------------------------------------------------------
    def t3d_pointmlp_dgrad_narrow(self, a, stream):
        p = _struct(a)
        dy = _dy(p.dy, p.M, p.N, p.M)
        K = p.k0 + p.kn
        w = arr(p.w, K, p.N).astype(np.float64)[p.k0:]
        arr(p.out, p.M, p.ld_out)[:, :p.kn] = dy @ w.T
        return 0

    def t3d_semi_final_loss(self, a, stream):
        p = _struct(a)
        B = p.B
        dims = arr(p.reg_dims, B, 3).astype(np.float64)
        cls = arr(p.one_hot, B, 10).argmax(1)
        tc = [bool(p.train_classes[i]) for i in range(10)]
        T = sum(tc)
        gd = np.zeros((B, 3))
        intra = 0.0
        hub = lambda e: 0.5 * np.minimum(np.abs(e), 1) ** 2 + (np.abs(e) - np.minimum(np.abs(e), 1))
        if p.w_weak != 0 and T > 0:
            for i in range(10):
                sel = cls == i
                if not tc[i] or sel.sum() == 0:
                    continue
                e = dims[sel] - dims[sel].mean(0)
                intra += hub(e).mean() / T
                gd[sel] = p.w_weak * np.clip(e, -1, 1) / (3 * sel.sum() * T)
        arr(p.d_dims, B, 3)[:] = gd
        o = arr(p.out9, B, 9).astype(np.float64)
        lg = o[:, 7:9]
        mx = lg.max(1, keepdims=True)
        lse = mx[:, 0] + np.log(np.exp(lg - mx).sum(1))
        pf = np.exp(lg[:, 1] - lse)
        m = arr(p.is_data_2D, B).astype(np.float64) if p.fit_only_2d else np.ones(B)
        fit = (-np.log(0.01 + pf) * m).mean()
        dp = -p.w_fit * m / ((0.01 + pf) * B)
        g = arr(p.dout9, B, 9)
        g[:] = 0
        g[:, 7] = -dp * pf * (1 - pf)
        g[:, 8] = dp * pf * (1 - pf)
        arr(p.fit_prob, B)[:] = pf
        arr(p.terms, 2)[:] = [intra, fit]
        arr(p.loss, 1)[0] = float(arr(p.strong_loss, 1)[0]) + p.w_weak * intra + p.w_fit * fit
        return 0

    def t3d_small_pair(self, a, b, stream):
        """Two independent small launches in one: by specification the two stand-alone calls."""
        names = {1: ('t3d_bn_bwd_finalize', 'bn_bwd'), 2: ('t3d_fc_bwd', 'fc_bwd'), 3: ('t3d_fc_dinput', 'fc_dinput'),
                 4: ('t3d_dy_colsum', 'dy_colsum')}
        for op in (_struct(a), _struct(b)):
            fn, field = names[op.kind]
            rc = getattr(self, fn)(C.byref(getattr(op.u, field)), stream)
            if rc:
                return rc
        return 0

    # ---- riders (t3d.h: small ops of one chain inside a GEMM launch of an independent chain): by specification the stand-alone
    # calls of the set's ops in order, then the GEMM -- in either order, the two do not depend on each other
    _RIDER = {1: ('t3d_bn_bwd_finalize', 'bn_bwd'), 2: ('t3d_fc_bwd', 'fc_bwd'), 3: ('t3d_fc_dinput', 'fc_dinput'),
              4: ('t3d_dy_colsum', 'dy_colsum'), 5: ('t3d_bn_fwd_finalize', 'bn_fwd'), 6: ('t3d_fc_fwd', 'fc_fwd'),
              7: ('t3d_pool_bwd_mid', 'mid')}

    def t3d_riders_plan(self, r):
        rs = _struct(r)
        if rs.n_ops <= 0 or rs.n_ops > abi.RIDER_MAX_OPS:
            return -1
        for k in range(rs.n_ops):
            o = rs.ops[k]
            if o.kind not in self._RIDER:
                return -1
            u = getattr(o.u, self._RIDER[o.kind][1])
            if o.kind == 7:
                if rs.n_ops != 1:
                    return -1
                if 128 * 128 * 4 + (4 * u.sparse.N + 128) * 4 > 76 * 1024:
                    return -2
                continue
            if o.kind in (2, 3, 6) and u.B > 32:
                return -2
            if (o.kind == 5 and u.n_tiles > 512) or (o.kind == 1 and u.psum_dz and u.n_tiles > 512):
                return -2
        rs.n_wg, rs.lds_bytes = 1, 0
        return 0

    # (specification library: every fp32 GEMM launch "hosts" -- the set simply runs beside it)
    def t3d_pointmlp_fwd_hosts_riders(self, a):
        return int(_struct(a).dtype == 0)

    def t3d_pointmlp_wgrad_hosts_riders(self, a):
        p = _struct(a)
        return int(p.dy.dtype == 0 and p.K <= 64 and p.N <= 128 and bool(p.dy.dz))

    def t3d_pointmlp_bwd_hosts_riders(self, d, w):
        return int(_struct(d).dtype == 0)

    def t3d_pool_bwd_stage1_hosts_riders(self, g, c, q):
        return int(_struct(g).a.dtype == 0)

    def t3d_pool_bwd_stage2_hosts_riders(self, f, d):
        return int(_struct(d).dtype == 0)

    def t3d_run_riders(self, r, stream):
        if not r:
            return 0
        rs = _struct(r)
        for k in range(rs.n_ops):
            o = rs.ops[k]
            fn, field = self._RIDER[o.kind]
            if o.kind == 7:
                m = o.u.mid
                rc = self.t3d_pool_bwd_mid(m.slab_base, m.grad_base, m.table_dev, m.n_tensors, m.max_numel, C.byref(m.sparse), stream)
            else:
                rc = getattr(self, fn)(C.byref(getattr(o.u, field)), stream)
            if rc:
                return rc
        return 0

    def t3d_pointmlp_fwd_r(self, a, r, stream):
        return self.t3d_run_riders(r, stream) or self.t3d_pointmlp_fwd(a, stream)

    def t3d_pointmlp_wgrad_r(self, a, r, stream):
        return self.t3d_run_riders(r, stream) or self.t3d_pointmlp_wgrad(a, stream)

    def t3d_pointmlp_bwd_r(self, d, w, r, stream):
        return self.t3d_pointmlp_bwd(d, w, stream) or self.t3d_run_riders(r, stream)

    def t3d_pool_bwd_stage1_r(self, g, c, q, r, stream):
        return self.t3d_pool_bwd_stage1(g, c, q, stream) or self.t3d_run_riders(r, stream)

    def t3d_pool_bwd_stage2_r(self, f, d, r, stream):
        return self.t3d_run_riders(r, stream) or self.t3d_pool_bwd_stage2(f, d, stream)

    def t3d_anchor_reg_bwd(self, a, stream):
        p = _struct(a)
        B = p.B
        o = arr(p.box, B, p.ld_box)[:, :67].astype(np.float64)
        g, ds1 = arr(p.dbox, B, 67), arr(p.dstage1, B, 3)
        js, ks = o[:, 3:15].argmax(1), o[:, 27:37].argmax(1)
        gc, gd, gt = np.zeros((B, 3)), np.zeros((B, 3)), np.zeros(B)
        if p.dbox7:
            q = arr(p.dbox7, B, 7).astype(np.float64)
            gc, gd, gt = q[:, 0:3].copy(), q[:, 3:6].copy(), q[:, 6].copy()
        if p.d_dims:
            gd = gd + arr(p.d_dims, B, 3)
        mean = MEAN32.astype(np.float64)
        for b in range(B):
            g[b, 0:3] += gc[b]
            ds1[b] += gc[b]
            k = ks[b]
            raw = mean[k] + o[b, 37 + 3 * k:40 + 3 * k] * mean[k]
            g[b, 37 + 3 * k:40 + 3 * k] += np.where(raw > 1e-5, gd[b] * mean[k], 0.0)
            g[b, 15 + js[b]] += gt[b] * (np.pi / 12)
        return 0

    # ---- optimiser -------------------------------------------------------------------------------
    def t3d_reduce_slabs(self, slab_base, grad_base, table, n_tensors, max_numel, stream):
        for i in range(n_tensors):
            d = table[i]
            sl = np.ctypeslib.as_array(slab_base, shape=(d.slab_off + d.n_slabs * d.numel,))[d.slab_off:]
            g = np.ctypeslib.as_array(grad_base, shape=(d.grad_off + d.numel,))[d.grad_off:]
            g[:] = sl.reshape(d.n_slabs, d.numel).astype(np.float64).sum(0)
        return 0

    def t3d_schedule_step(self, hyper, s, stream):
        h = arr(hyper, 4)
        s = _struct(s)
        step = float(h[0]) + s.step_offset
        seen = step * s.batch_size
        lr = s.base_lr * s.lr_decay_rate ** math.floor(seen / s.lr_decay_step)
        bnd = min(s.bn_decay_clip, 1 - s.bn_init_decay * s.bn_decay_rate ** math.floor(seen / s.bn_decay_step))
        t = step + 1
        h[1], h[2] = lr, bnd
        h[3] = lr * math.sqrt(1 - s.beta2 ** t) / (1 - s.beta1 ** t)
        h[0] = t
        return 0

    def t3d_adam_tf_step(self, params, grads, m, v, n, hyper, b1, b2, eps, gscale, stream):
        w, g, mm, vv = arr(params, n), arr(grads, n), arr(m, n), arr(v, n)
        lr_t = arr(hyper, 4)[3]
        gi = g * np.float32(gscale)
        mm[:] = np.float32(b1) * mm + np.float32(1 - b1) * gi
        vv[:] = np.float32(b2) * vv + np.float32(1 - b2) * gi * gi
        w[:] = w - lr_t * mm / (np.sqrt(vv) + np.float32(eps))
        return 0

    def t3d_split_x3(self, src, planes, n, stride, stream):
        return 0          # (the specification library multiplies in fp64: no operand planes)

    def t3d_split_x3_frag(self, params, planes_fwd, planes_dgrad, stride, table, n, n_blocks, stream):
        return 0          # (same: the launch structs' w_x3 is ignored)

    def t3d_momentum_step(self, params, grads, accum, n, hyper, momentum, gscale, stream):
        w, g, a = arr(params, n), arr(grads, n), arr(accum, n)
        lr = arr(hyper, 4)[1]
        a[:] = np.float32(momentum) * a + g * np.float32(gscale)
        w[:] = w - lr * a
        return 0

    def t3d_dropout_mask(self, mask, n, keep, seed, hyper, stream):
        arr(mask, n)[:] = hash_keep_mask(seed, int(arr(hyper, 4)[0]), n, keep)
        return 0
