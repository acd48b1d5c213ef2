"""Host-side engine: runtime, variable store, launch plan and the fused layer building blocks.

This is the analogue of the reference's TF-1 graph + session (train_semisup.py:204-277): model
builders record kernel launches into a static `Plan` whose argument structs point at pre-allocated
HBM buffers; a step is one replay of the plan (captured into a hipGraph on the GPU).  PyTorch is used
only to own device memory / streams / the RCCL process group.

Per-point tensors are never materialised in their normalised form: a layer writes its RAW conv
output y plus per-tile statistics, and every consumer applies `relu(y*scale+shift)` while loading
(t3d_act_src).  Gradients likewise travel as dz (grad w.r.t. the batch-norm output) + three
per-channel coefficients (t3d_dy_src).
"""
import contextlib
import ctypes as C
import math
import os

import numpy as np
import torch

from . import abi
from .abi import fptr, iptr
from .constants import BN_EPS

TILE = abi.TILE_ROWS


class Runtime:
    """Owns the library handle and the device.  `lib` defaults to the HIP library (no fallback)."""

    def __init__(self, device=None, lib=None, gemm_arithmetic=None):
        """`gemm_arithmetic`: arithmetic of the fp32 per-point GEMM launches of every plan built on this runtime -- 'bf16x3' (default:
        three exact bf16 terms per operand on the bf16 matrix pipe) or 'fp32_mfma' (t3d.h T3D_ARITH_*).  It is fixed HERE, travels in
        the launch structs, and is what `describe_gemm_arithmetic` reports; the environment (T3D_X3=0 -> 'fp32_mfma') only supplies
        the default, once."""
        self.lib = lib if lib is not None else abi.load()
        if gemm_arithmetic is None:
            gemm_arithmetic = 'fp32_mfma' if os.environ.get('T3D_X3', '1') == '0' else 'bf16x3'
        self.gemm_arithmetic = gemm_arithmetic
        self.arith = abi.ARITH_BY_NAME[gemm_arithmetic]
        if device is None:
            if not torch.cuda.is_available():
                raise abi.T3DError('no GPU visible: the product path runs on an MI355X only')
            device = torch.device('cuda', torch.cuda.current_device())
        self.device = torch.device(device)
        self.allocs = []     # every buffer lives as long as the runtime (launch structs hold raw pointers)
        self._side = None

    def side_stream(self):
        """Second HIP stream for work off the critical path (weight gradients): see Plan.side()."""
        if self._side is None:
            self._side = torch.cuda.Stream(self.device)
        return self._side

    def stream(self):
        if self.device.type == 'cuda':
            return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        return C.c_void_p(0)

    def zeros(self, *shape, dtype=torch.float32):
        t = torch.zeros(*shape, dtype=dtype, device=self.device)
        self.allocs.append(t)
        return t

    def full(self, shape, val, dtype=torch.float32):
        t = torch.full(shape, val, dtype=dtype, device=self.device)
        self.allocs.append(t)
        return t


# Two-stream execution of a plan (see Plan.side).  Measured on MI355X / ROCm 7.2 with the step captured into a hipGraph:
# forking the weight gradients into a parallel graph branch is SLOWER than the linear graph (2.04 ms with a fork per
# layer, 1.86 ms with two batched forks, 1.82 ms linear) -- cross-stream graph edges cost more than the overlap returns --
# so it is off unless T3D_SIDE_STREAM=1.
SIDE_STREAM = os.environ.get('T3D_SIDE_STREAM', '0') == '1'
# dgrad + wgrad of a dense layer in one launch (t3d_pointmlp_bwd); T3D_FUSE_BWD=0 restores the two launches
FUSE_BWD = os.environ.get('T3D_FUSE_BWD', '1') != '0'
# the fully-connected backward chains of the box / T-Net (a few workgroups each, ~100 us per step of dependent launches) on the
# second stream beside the segmentation net's backward GEMMs, which do not depend on them
FC_SIDE = os.environ.get('T3D_FC_SIDE', '0') == '1'
# the sparse rows S are written / read only where a row received an arg-max hit (row flags beside S)
SPARSE_GATED = os.environ.get('T3D_SPARSE_GATED', '1') != '0'
# EXPERIMENT (round 6, off by default: T3D_DZ_POOL=1): the [M, N] gradient tensors dz between the per-point layers of a net from a small pool
# of that net instead of one allocation per layer -- a dz lives from the launch that writes it (the next layer's data gradient) to the launch
# that reads it (its own layer's backward), so two to four buffers serve a net.  The idea: one allocation per layer puts 277 MB of write-once
# tensors through the chip's 256 MB memory-side cache in every backward pass, and the narrow backward launches take 24 us with their operands
# in that cache against 35 us without (tools/bench_x3.py, T3D_BENCH_COLD=1).  Measured: 1.1257 against 1.1271 ms per step (four same-box
# pairs: noise).  The two widest dz of a net are alive together, so the pool only goes from 277 to 217 MB, and the forward activations
# (277 MB) exceed the cache by themselves.  Never on when launches of one net may run out of their recorded order (weight gradients on a
# second stream, un-fused backward).
DZ_POOL = os.environ.get('T3D_DZ_POOL', '0') == '1' and not SIDE_STREAM and not FC_SIDE and FUSE_BWD


class Plan:
    """Ordered list of kernel launches with frozen argument structs.

    Launches recorded inside `with plan.side():` are off the critical path (the weight gradients: nothing but the
    optimiser consumes them).  On the GPU they go to a second HIP stream, forked from the main stream at the point
    at the next `plan.flush()` and joined at `plan.join()` / the end of the plan, so that under hipGraph capture they
    become a parallel branch of the graph.  On one stream (the default, see SIDE_STREAM) the recorded order is a valid
    serial schedule."""

    JOIN, FLUSH = '__join__', '__flush__'
    MARK = '__mark__'                             # chain boundaries for schedule.overlap_chains (step.TrainStep); no-ops when run
    BUCKET, WAIT = '__bucket__', '__wait__'       # data-parallel markers (step.TrainStep): gradient bucket i complete / needed

    def __init__(self, rt):
        self.rt = rt
        self.calls = []      # (name, callable(stream) -> rc, argument struct or None)
        self.lanes = []      # 0 = main stream, 1 = side stream, parallel to `calls`
        self.keep = []       # keep-alive for structs / tensors
        self._lane = 0
        self.two_streams = False   # this plan's side lane really runs on the second stream (FC_SIDE schedule)

    @contextlib.contextmanager
    def side(self):
        prev, self._lane = self._lane, 1
        try:
            yield self
        finally:
            self._lane = prev

    def add(self, name, args):
        fn = getattr(self.rt.lib, name)
        ref = C.byref(args)
        self.keep.append(args)
        self.calls.append((name, lambda s, fn=fn, ref=ref: fn(ref, s), args))
        self.lanes.append(self._lane)

    def add_raw(self, name, thunk, *keep):
        self.keep.extend(keep)
        self.calls.append((name, thunk, None))
        self.lanes.append(self._lane)

    def mark(self, tag):
        """Names a point of the plan (the boundaries of the two independent chains the step scheduler interleaves)."""
        self.calls.append((Plan.MARK, lambda s: 0, tag))
        self.lanes.append(0)

    def join(self):
        """Everything recorded so far (both lanes) completes before what follows."""
        self.calls.append((Plan.JOIN, lambda s: 0, None))
        self.lanes.append(0)

    def flush(self):
        """Fork point: the side-lane launches recorded so far are issued here, behind one cross-stream dependency
        (every graph edge between streams costs microseconds, so the side work is forked in a few large batches
        rather than launch by launch)."""
        self.calls.append((Plan.FLUSH, lambda s: 0, None))
        self.lanes.append(0)

    def run(self):
        if self.rt.device.type == 'cuda' and any(self.lanes) and (SIDE_STREAM or self.two_streams):
            return self._run_two_streams()
        s = self.rt.stream()
        for name, call, _ in self.calls:
            rc = call(s)
            if rc != 0:
                abi.check(rc, name)

    def _run_two_streams(self):
        main, side = torch.cuda.current_stream(self.rt.device), self.rt.side_stream()
        ms, ss = C.c_void_p(main.cuda_stream), C.c_void_p(side.cuda_stream)
        deferred, pending = [], False

        def issue_side():
            nonlocal pending
            if deferred:
                side.wait_stream(main)           # fork: the batch sees everything recorded on main so far
                for name, call in deferred:
                    rc = call(ss)
                    if rc != 0:
                        abi.check(rc, name)
                del deferred[:]
                pending = True

        for (name, call, _), lane in zip(self.calls, self.lanes):
            if name == Plan.FLUSH:
                issue_side()
            elif name == Plan.JOIN:
                issue_side()
                if pending:
                    main.wait_stream(side)
                pending = False
            elif lane == 1:
                deferred.append((name, call))
            else:
                rc = call(ms)
                if rc != 0:
                    abi.check(rc, name)
        issue_side()
        if pending:
            main.wait_stream(side)

    def bucket_ready(self, i):
        """Every gradient of bucket i (nets.Graph.buckets[i]) has been written by the launches recorded so far: a data-parallel
        step issues the bucket's all-reduce here, beside what follows.  No-op for a single replica."""
        self.calls.append((Plan.BUCKET, lambda s: 0, i))
        self.lanes.append(0)

    def bucket_wait(self, i):
        """What follows (the bucket's Adam launch) needs bucket i's all-reduced gradients."""
        self.calls.append((Plan.WAIT, lambda s: 0, i))
        self.lanes.append(0)

    def __len__(self):
        return sum(1 for name, _, _ in self.calls if not name.startswith('__'))


# (round 4, 8-byte pieces: 1.276 ms per B=32 N=1024 step with the planes, 1.257 ms with the split in the kernels.  Round 5 rebuilt the
# copy with 16-byte pieces -- one load and one ds_write_b128 per eight elements, a third of the staging pass's vector instructions
# gone (csrc/pointmlp.hip StagerX3W) -- bit-identical again, and per launch at M = 32768 (tools/bench_x3_pc.py, modes 0,p): forward
# 512 -> 256 57.9 -> 61.3 us, 128 -> 1024 61.3 -> 58.0, 256 -> 512 58.3 -> 57.7, 128 -> 256 20.1 -> 18.8, 128 -> 128 13.3 -> 12.4;
# fused backward 512 -> 256 117 -> 127, 128 -> 256 41.0 -> 42.9, 64 -> 64 9.3 -> 9.9: the count of vector instructions is not what
# bounds these kernels.  Off by default.)
# Round 6: the planes are written in MFMA-FRAGMENT order (t3d_split_x3_frag, one launch per step for every layer, a forward and a
# data-gradient arrangement) and the kernels read the weight operand of every MFMA straight from them -- no LDS image, no conversion, no
# ds_write, no fragment ds_read for that operand.  On by default (T3D_X3_PRESPLIT=0: the in-kernel split of rounds 4-5).
X3_PRESPLIT = os.environ.get('T3D_X3', '1') != '0' and os.environ.get('T3D_X3_PRESPLIT', '1') == '1'


class VarStore:
    """TF-named variables as views of flat fp32 buffers (params | grads | adam m | adam v, and a
    separate flat buffer for the non-trainable moving statistics).  Names follow the reference's
    checkpoint layout: <scope>/weights, /biases, /bn/{beta,gamma,moving_mean,
    moving_variance}."""

    def __init__(self, rt, capacity=4 << 20, state_capacity=1 << 16, seed=0):
        self.rt = rt
        self.params = rt.zeros(capacity)
        self.grads = rt.zeros(capacity)
        self.adam_m = rt.zeros(capacity)
        self.adam_v = rt.zeros(capacity)
        self.state = rt.zeros(state_capacity)
        self.params16 = None       # bf16 copy of `params` for the T3D_BF16 GEMMs (enable_bf16; refreshed by t3d_cast_bf16 every step)
        self.used = 0
        self.state_used = 0
        self.index = {}        # name -> (offset, shape, trainable)
        self.rng = np.random.RandomState(seed)

    def _alloc(self, name, shape, trainable, init):
        if name in self.index:
            off, shp, tr = self.index[name]
            assert tuple(shp) == tuple(shape) and tr == trainable, 'variable %s re-declared with a different shape' % name
            return self.get(name)
        n = int(np.prod(shape))
        n_pad = (n + 3) // 4 * 4            # keep every view 16-byte aligned
        if trainable:
            off = self.used
            assert off + n_pad <= self.params.numel(), 'VarStore capacity exceeded'
            self.used += n_pad
            buf = self.params
        else:
            off = self.state_used
            assert off + n_pad <= self.state.numel()
            self.state_used += n_pad
            buf = self.state
        self.index[name] = (off, tuple(shape), trainable)
        view = buf[off:off + n].view(*shape)
        view.copy_(torch.as_tensor(np.asarray(init, dtype=np.float32).reshape(shape)))
        return view

    def enable_bf16(self):
        if self.params16 is None:
            self.params16 = self.rt.zeros(self.params.numel(), dtype=torch.bfloat16)
        return self.params16

    def enable_x3(self):
        """Three bf16 planes of `params` (x = h + m + l exactly; t3d_split_x3, refreshed once per step by Graph.emit_cast_weights): what
        the fp32 GEMM kernels on the bf16 matrix pipe (csrc/pointmlp.hip PathX3) read as the weight operand instead of splitting the
        fp32 matrix again in every tile that stages it."""
        if getattr(self, 'params_x3', None) is None:
            self.params_x3 = self.rt.zeros(3 * self.params.numel(), dtype=torch.bfloat16)
        return self.params_x3

    def x3_ptr(self, w):
        """(device address of plane 0, elements between planes) of the x3 twin of a contiguous view `w` of the fp32 parameter buffer;
        (None, 0) if `w` does not live there."""
        off = (w.data_ptr() - self.params.data_ptr()) // 4
        if not (0 <= off and off + w.numel() <= self.params.numel() and w.is_contiguous()) or off % 4:
            return None, 0
        return self.enable_x3().data_ptr() + 2 * off, self.params.numel()

    def x3_frag(self, w, K, N):
        """Registers the [K, N] view `w` of the parameter buffer for t3d_split_x3_frag (both arrangements) and returns
        ((forward planes address, plane stride), (data-gradient planes address, plane stride)); ((None, 0), (None, 0)) if the view
        cannot take part (not in the buffer, offset not a multiple of 8 elements, K or N not a multiple of 32)."""
        off = (w.data_ptr() - self.params.data_ptr()) // 4
        none = ((None, 0), (None, 0))
        if not (0 <= off and off + w.numel() <= self.params.numel() and w.is_contiguous()) or off % 8 or K % 32 or N % 32 or w.numel() != K * N:
            return none
        n = self.params.numel()
        stride = (n + 7) // 8 * 8
        if getattr(self, 'x3_frag_planes', None) is None:
            self.x3_frag_planes = (self.rt.zeros(3 * stride, dtype=torch.bfloat16), self.rt.zeros(3 * stride, dtype=torch.bfloat16))
            self.x3_frag_entries, self.x3_frag_stride, self.x3_frag_table = {}, stride, None
        self.x3_frag_entries[off] = (K, N)
        self.x3_frag_table = None            # rebuilt by frag_table()
        pf, pd = self.x3_frag_planes
        return (pf.data_ptr() + 2 * off, stride), (pd.data_ptr() + 2 * off, stride)

    def frag_table(self):
        """(device table of t3d_x3_frag_entry, number of entries, number of workgroups) of every registered matrix; None if there is none."""
        ent = getattr(self, 'x3_frag_entries', None)
        if not ent:
            return None
        if self.x3_frag_table is None:
            raw, blocks = abi.x3_frag_table([(off, K, N) for off, (K, N) in sorted(ent.items())])
            self.x3_frag_table = (torch.from_numpy(raw).to(self.rt.device), len(ent), blocks)
        return self.x3_frag_table

    def bf16_view(self, w):
        """The bf16 twin of a view `w` of the fp32 parameter buffer (same element offset, same shape)."""
        off = (w.data_ptr() - self.params.data_ptr()) // 4
        assert 0 <= off and off + w.numel() <= self.params.numel() and w.is_contiguous()
        return self.enable_bf16()[off:off + w.numel()].view(*w.shape)

    def get(self, name):
        off, shape, trainable = self.index[name]
        buf = self.params if trainable else self.state
        return buf[off:off + int(np.prod(shape))].view(*shape)

    def grad(self, name):
        off, shape, trainable = self.index[name]
        assert trainable
        return self.grads[off:off + int(np.prod(shape))].view(*shape)

    def offset(self, name):
        return self.index[name][0]

    def xavier(self, name, shape, fan_in, fan_out):
        """tf.contrib.layers.xavier_initializer (uniform), tf_util.py:1183."""
        if name in self.index:
            return self.get(name)
        lim = math.sqrt(6.0 / (fan_in + fan_out))
        return self._alloc(name, shape, True, self.rng.uniform(-lim, lim, size=shape))

    def const(self, name, shape, val, trainable=True):
        return self._alloc(name, shape, trainable, np.full(shape, val, np.float32))

    # ---- state dict in the reference's naming --------------------------------------------------
    def state_dict(self):
        return {k: self.get(k).detach().cpu().numpy().copy() for k in self.index}

    def load_state_dict(self, sd, strict=True):
        for k, v in sd.items():
            if k not in self.index:
                if strict:
                    raise KeyError(k)
                continue
            self.get(k).copy_(torch.as_tensor(np.asarray(v, dtype=np.float32)).reshape(self.index[k][1]))

    def trainable_ranges(self, prefixes=None):
        """Contiguous [offset, n) ranges of the trainable variables whose name starts with one of
        `prefixes` (regex-prefix semantics of tf.get_collection(scope=...), train_semisup_adv.py:415-422)."""
        items = sorted((off, int(np.prod(shape)), name) for name, (off, shape, tr) in self.index.items() if tr)
        out = []
        for off, n, name in items:
            if prefixes is not None and not any(name.startswith(p) for p in prefixes):
                continue
            n_pad = (n + 3) // 4 * 4
            if out and out[-1][0] + out[-1][1] == off:
                out[-1][1] += n_pad
            else:
                out.append([off, n_pad])
        return [(o, n) for o, n in out]


class Workspace:
    """Bump allocator for wgrad slabs + the device-side slab table of t3d_reduce_slabs."""

    def __init__(self, rt):
        self.rt = rt
        self.entries = []      # (slab_off, grad_off, numel, n_slabs)
        self.total = 0
        self.buf = None
        self.table = None

    def reserve(self, grad_off, numel, n_slabs):
        off = self.total
        self.entries.append((off, grad_off, numel, n_slabs))
        self.total += numel * n_slabs
        return off

    def finalize(self):
        self.buf = self.rt.zeros(max(self.total, 4))
        n = len(self.entries)
        host = (abi.SlabDesc * max(n, 1))()
        for i, (so, go, ne, ns) in enumerate(self.entries):
            host[i] = abi.SlabDesc(so, go, ne, ns)
        raw = np.frombuffer(bytes(host), dtype=np.uint8).copy()
        self.table = torch.as_tensor(raw).to(self.rt.device)
        return self


# ------------------------------------------------------------------------------------------------
# lazy per-point activation handle
# ------------------------------------------------------------------------------------------------
class ActSpec:
    """a[m,k] = relu?(x[m, coff+k]*scale[k] + shift[k]) - sub[b(m),k]  (t3d_act_src)."""

    def __init__(self, x, ldx, K, coff=0, scale=None, shift=None, relu=False, sub=None, sub_ld=0, producer=None):
        self.x, self.ldx, self.K, self.coff = x, ldx, K, coff
        self.scale, self.shift, self.relu, self.sub, self.sub_ld = scale, shift, relu, sub, sub_ld
        self.producer = producer     # PointLayer that owns x (None for raw inputs)
        self.dtype = abi.BF16 if x.dtype == torch.bfloat16 else abi.F32      # element type of x: the raw inputs are always fp32

    def struct(self):
        return abi.ActSrc(fptr(self.x), self.ldx, self.coff, fptr(self.scale), fptr(self.shift), int(self.relu),
                          fptr(self.sub), self.sub_ld, self.dtype)


def wgrad_rows_per_split(lib, M, K, N):
    """Row split of a weight gradient, from the library's own policy (t3d_wgrad_plan)."""
    rps, tk, tn = C.c_int(0), C.c_int(0), C.c_int(0)
    abi.check(lib.t3d_wgrad_plan(M, K, N, C.byref(rps), C.byref(tk), C.byref(tn)), 't3d_wgrad_plan')
    return rps.value


def bwd_rows_per_split(lib, M, K, N, dtype):
    """Row split of the fused backward (t3d_bwd_plan): the one-pass form's split where the library takes that form."""
    rps, one = C.c_int(0), C.c_int(0)
    abi.check(lib.t3d_bwd_plan(M, K, N, dtype, C.byref(rps), C.byref(one)), 't3d_bwd_plan')
    return rps.value


def gram_rows_per_split(lib, M, K, dtype):
    """Row split of the Gram slabs of a pooled layer's backward (t3d_gram_plan)."""
    rps, one = C.c_int(0), C.c_int(0)
    abi.check(lib.t3d_gram_plan(M, K, dtype, C.byref(rps), C.byref(one)), 't3d_gram_plan')
    return rps.value


class PointLayer:
    """tf_util.conv2d 1x1 (+ batch_norm + ReLU) over M = B*N point rows (tf_util.py:1258-1323)."""

    def __init__(self, g, scope, K, N, w=None, bias=None, kernel_1xD=False, w_name=None, w_row0=0, pool=False, gram=None, bn=True,
                 n_alloc=None):
        """bn=False: tf_util.conv2d(bn=False) -- no batch-norm variables, the forward emits the GEMM only (forward-only surface:
        the hot path has no such layer except conv10, which lives in the segmentation head).  n_alloc: padded column count of the
        GEMM (N < 64 outputs: the weights are copied into a zero-padded [K, n_alloc] matrix in front of the launch)."""
        self.g, self.scope, self.K, self.N = g, scope, K, N
        self.bn = bn
        # pooled layers: Gram-form backward (t3d.h K11e) -- the [M,N] output is never stored
        self.gram = (pool and os.environ.get('T3D_POOL_GRAM', '1') != '0') if gram is None else (gram and pool)
        self.w_name, self.w_row0 = w_name, w_row0
        rt, vs = g.rt, g.vars
        M, T = g.M, g.M // TILE
        self.M, self.T = M, T
        if w is None:
            shape = (1, K, 1, N) if kernel_1xD else (1, 1, K, N)
            fan_in, fan_out = (K, K * N) if kernel_1xD else (K, N)
            w = vs.xavier(scope + '/weights', shape, fan_in, fan_out).view(K, N)
            self.w_name = scope + '/weights'
            self.w_row0 = 0
        self.w = w
        self.dt, self.adt = g.dt, g.adt                  # abi.F32 / abi.BF16 and the torch dtype of the [M, C] layer tensors
        # what the GEMM kernels read as `w`: the fp32 weights, or their bf16 copy (refreshed every step: Graph.emit_cast_weights)
        self.w_mm = vs.bf16_view(w) if self.dt == abi.BF16 else w
        # fp32 layers on the three-term bf16 path: the weights pre-split once per step (T3D_X3_PRESPLIT=0: split in every tile)
        # (vs.x3_frag_enabled = False: a program that does not refresh the planes between EVERY optimiser launch and the next use of its
        # weights -- step.PipelinedStep, whose two chains update their halves of the variables at different times -- keeps the in-kernel split)
        self.w_x3, self.w_x3_d = vs.x3_frag(w, K, N) if (self.dt == abi.F32 and X3_PRESPLIT and rt.device.type == 'cuda' and rt.arith != abi.ARITH_FP32_MFMA and
                                                       getattr(vs, 'x3_frag_enabled', True)) else ((None, 0), (None, 0))
        assert not (self.dt == abi.BF16 and pool and not self.gram), "bf16: max-pooled layers take the Gram-form backward"
        self.bias = bias if bias is not None else vs.const(scope + '/biases', (N,), 0.0)
        if bn:
            self.gamma = vs.const(scope + '/bn/gamma', (N,), 1.0)
            self.beta = vs.const(scope + '/bn/beta', (N,), 0.0)
            self.mm = vs.const(scope + '/bn/moving_mean', (N,), 0.0, trainable=False)
            self.mv = vs.const(scope + '/bn/moving_variance', (N,), 1.0, trainable=False)
        self.NA = n_alloc or N               # columns of the GEMM (>= N, multiple of 64)
        if self.NA != N:
            assert not pool and self.dt == abi.F32
            self.w_pad, self.b_pad = rt.zeros(K, self.NA), rt.zeros(self.NA)
        N = self.NA
        self.y = None if self.gram else rt.zeros(M, N, dtype=self.adt)
        self.psum, self.psumsq = rt.zeros(T, N), rt.zeros(T, N)
        self.scale, self.shift = rt.zeros(N), rt.zeros(N)
        self.mean, self.invstd = rt.zeros(N), rt.zeros(N)
        self.pool = pool
        if pool:
            B = g.B
            self.pmax, self.pmin = rt.zeros(T, N), rt.zeros(T, N)
            self.pamax, self.pamin = rt.zeros(T, N, dtype=torch.int32), rt.zeros(T, N, dtype=torch.int32)
            self.pooled, self.argidx, self.ysel = rt.zeros(B, N), rt.zeros(B, N, dtype=torch.int32), rt.zeros(B, N)
        self.src = None
        self.dz = None
        self.coef = None

    # ---- forward -------------------------------------------------------------------------------
    def fwd(self, plan, src, is_training, rowbias=None, rowmask=None):
        g, rt = self.g, self.g.rt
        assert src.K == self.K
        pool = self.pool
        self.src, self.rowmask, self.is_training = src, rowmask, is_training
        a = abi.PointMlpFwdArgs()
        a.arith = self.g.rt.arith
        a.a = src.struct()
        a.w, a.bias, a.rowbias, a.y = fptr(self.w_mm), fptr(self.bias), fptr(rowbias), fptr(self.y)
        a.dtype = self.dt
        a.w_x3, a.w_x3_stride = self.w_x3 if self.NA == self.N else (None, 0)
        if self.NA != self.N:                 # fewer than 64 output channels: zero-padded copy of the weights, refreshed per run
            wp, bp, w, b, n = self.w_pad, self.b_pad, self.w, self.bias, self.N
            plan.add_raw('pad_weights', lambda s: (wp[:, :n].copy_(w), bp[:n].copy_(b), 0)[2])
            a.w, a.bias = fptr(wp), fptr(bp)
        a.psum, a.psumsq = fptr(self.psum), fptr(self.psumsq)
        if pool:
            a.rowmask = fptr(rowmask)
            a.pmax, a.pmin, a.pamax, a.pamin = fptr(self.pmax), fptr(self.pmin), iptr(self.pamax), iptr(self.pamin)
        a.M, a.K, a.N, a.rows_per_frustum = self.M, self.K, self.NA, g.rpf
        plan.add('t3d_pointmlp_fwd', a)
        self._fwd_args, self._fin_args = a, None
        if not self.bn:
            self.out = ActSpec(self.y, self.NA, self.N, 0, None, None, False, producer=self)
            return self.out
        f = abi.BnFwdFinalizeArgs()
        f.psum, f.psumsq, f.n_tiles, f.count, f.N = fptr(self.psum), fptr(self.psumsq), self.T, self.M, self.N
        f.gamma, f.beta, f.moving_mean, f.moving_var = fptr(self.gamma), fptr(self.beta), fptr(self.mm), fptr(self.mv)
        f.decay, f.eps, f.is_training, f.unbiased_ema = fptr(g.bn_decay_ptr), BN_EPS, int(is_training), int(g.unbiased_ema)
        f.scale, f.shift, f.mean, f.invstd = fptr(self.scale), fptr(self.shift), fptr(self.mean), fptr(self.invstd)
        if pool:                      # the max-pool pick of the same channels rides in the finalize launch (K3)
            f.pool_pmax, f.pool_pmin, f.pool_pamax, f.pool_pamin = fptr(self.pmax), fptr(self.pmin), iptr(self.pamax), iptr(self.pamin)
            f.pool_B, f.pool_tiles_per_frustum = g.B, g.rpf // TILE
            f.pooled, f.ld_pooled, f.argidx, f.ysel = fptr(self.pooled), self.N, iptr(self.argidx), fptr(self.ysel)
        plan.add('t3d_bn_fwd_finalize', f)
        self._fin_args = f
        self.out = None if self.gram else ActSpec(self.y, self.N, self.N, 0, self.scale, self.shift, True, producer=self)
        return self.out

    def enable_pool(self, rowmask=None):
        """Turn an already-emitted batch-normed layer into one that ALSO yields its global max-pool over the points: the recorded
        argument structs of its GEMM (per-tile max / min / arg partials in the epilogue) and of its finalize (the pick) are
        completed in place.  This is how tf_util.max_pool2d(net, [num_point, 1]) after an ordinary tf_util.conv2d gets the fused
        form (the reference decides at the max_pool2d call, semisup_models.py:96; the launch schedule is not captured before the
        first Session.run, so the structs are still open)."""
        g, rt = self.g, self.g.rt
        assert self.bn and self._fin_args is not None and self.NA == self.N, 'enable_pool: a batch-normed layer with >= 64 channels'
        if not self.pool:
            T, N, B = self.T, self.N, g.B
            self.pmax, self.pmin = rt.zeros(T, N), rt.zeros(T, N)
            self.pamax, self.pamin = rt.zeros(T, N, dtype=torch.int32), rt.zeros(T, N, dtype=torch.int32)
            self.pooled, self.argidx, self.ysel = rt.zeros(B, N), rt.zeros(B, N, dtype=torch.int32), rt.zeros(B, N)
            self.pool = True
        a, f = self._fwd_args, self._fin_args
        self.rowmask = rowmask
        a.rowmask = fptr(rowmask)
        a.pmax, a.pmin, a.pamax, a.pamin = fptr(self.pmax), fptr(self.pmin), iptr(self.pamax), iptr(self.pamin)
        f.pool_pmax, f.pool_pmin, f.pool_pamax, f.pool_pamin = fptr(self.pmax), fptr(self.pmin), iptr(self.pamax), iptr(self.pamin)
        f.pool_B, f.pool_tiles_per_frustum = g.B, g.rpf // TILE
        f.pooled, f.ld_pooled, f.argidx, f.ysel = fptr(self.pooled), self.N, iptr(self.argidx), fptr(self.ysel)
        return self.pooled

    # ---- backward ------------------------------------------------------------------------------
    def _ensure_bwd_buffers(self):
        rt = self.g.rt
        if self.coef is None:
            self.coef = rt.zeros(3, self.N)
            if self.pool:
                self.dpool = rt.zeros(self.g.B, self.N)
                if self.gram:
                    K, N = self.K, self.N
                    self.wc, self.S = rt.zeros(N, K), rt.zeros(self.M, K)
                    self.S_live = rt.zeros(self.M, dtype=torch.int32) if SPARSE_GATED else None
                    assert self.S_live is not None or self.dt == abi.F32, "bf16: the fp32 sparse rows S need their row flags"
            else:
                self.dz = self._acquire_dz()
                self.psum_dz, self.psum_dzy = rt.zeros(self.T, self.N), rt.zeros(self.T, self.N)

    def _acquire_dz(self):
        """[M, N] view of a pooled buffer of this layer's net (DZ_POOL above), or an allocation of its own."""
        g, rt = self.g, self.g.rt
        if not (DZ_POOL and getattr(g, 'pool_dz', True)):
            return rt.zeros(self.M, self.N, dtype=self.adt)
        pools = g.__dict__.setdefault('_dz_pools', {})
        free = pools.setdefault((self.scope.rsplit('/', 1)[0], self.adt), [])
        need = self.M * self.N
        pick = None
        for i in range(len(free) - 1, -1, -1):      # the most recently released buffer that is large enough (still in the cache)
            if free[i].numel() >= need:
                pick = free.pop(i)
                break
        if pick is None:
            pick = rt.zeros(need, dtype=self.adt)
        self._dz_buf = pick
        return pick[:need].view(self.M, self.N)

    def _release_dz(self):
        """This layer's backward launches are recorded: nothing recorded after them reads its dz."""
        buf = getattr(self, '_dz_buf', None)
        if buf is not None:
            self._dz_buf = None
            self.g.__dict__.setdefault('_dz_pools', {}).setdefault((self.scope.rsplit('/', 1)[0], self.adt), []).append(buf)

    def dy_struct(self):
        if self.pool:
            return abi.DySrc(fptr(None), fptr(self.y), fptr(self.coef), iptr(self.argidx), fptr(self.dpool), self.dt)
        return abi.DySrc(fptr(self.dz), fptr(self.y), fptr(self.coef), iptr(None), fptr(None), self.dt)

    def bn_bwd(self, plan, dpool_in=None, ld_dpool_in=0, param_grads=True):
        """dgamma/dbeta + the three dy coefficients.  `frozen` nets (eval-mode BN) use scale only."""
        g, vs = self.g, self.g.vars
        self._ensure_bwd_buffers()
        if self.pool and getattr(self, '_bn_bwd_fused_in', None) is plan:
            return                    # already done inside the t3d_fc_dinput launch that produced dpool_in
        a = abi.BnBwdFinalizeArgs()
        if self.pool:
            a.dpool_in, a.ld_dpool_in = fptr(dpool_in), ld_dpool_in
            a.pooled, a.ld_pooled, a.ysel, a.dpool, a.B = fptr(self.pooled), self.N, fptr(self.ysel), fptr(self.dpool), g.B
        else:
            a.psum_dz, a.psum_dzy, a.n_tiles = fptr(self.psum_dz), fptr(self.psum_dzy), self.T
        a.count, a.N = self.M, self.N
        a.gamma, a.mean, a.invstd, a.scale = fptr(self.gamma), fptr(self.mean), fptr(self.invstd), fptr(self.scale)
        a.frozen = int(not self.is_training)
        if param_grads and self.is_training:
            a.dgamma, a.dbeta = fptr(vs.grad(self.scope + '/bn/gamma')), fptr(vs.grad(self.scope + '/bn/beta'))
        a.coef = fptr(self.coef)
        plan.add('t3d_bn_bwd_finalize', a)

    def _gram_reduce(self, plan, regions):
        """regions: [(name, numel, n_slabs)] -> (slab views, reduced views); ONE t3d_reduce_slabs launch sums them."""
        rt = self.g.rt
        soff, ooff, table = {}, {}, []
        st = ot = 0
        for name, numel, ns in regions:
            soff[name], ooff[name] = st, ot
            table.append(abi.SlabDesc(st, ot, numel, ns))
            st += numel * ns
            ot += numel
        slabs, out = rt.zeros(st), rt.zeros(ot)
        desc = (abi.SlabDesc * len(table))(*table)
        tab = torch.as_tensor(np.frombuffer(bytes(desc), dtype=np.uint8).copy()).to(rt.device)
        rt.allocs.append(tab)
        lib, n, mx = rt.lib, len(table), max(r[1] for r in regions)

        def emit(sparse=None):
            tp = lambda: C.cast(C.c_void_p(tab.data_ptr()), C.POINTER(abi.SlabDesc))
            if sparse is None:
                plan.add_raw('t3d_reduce_slabs', lambda s: lib.t3d_reduce_slabs(fptr(slabs), fptr(out), tp(), n, mx, s))
            else:
                ref = C.byref(sparse)
                # (the same arguments as one struct: what the step scheduler needs to let this launch ride in a GEMM, schedule.py)
                marg = abi.PoolBwdMidArgs(fptr(slabs), fptr(out), tp(), n, mx, sparse)
                plan.keep.append(sparse)
                plan.calls.append(('t3d_pool_bwd_mid', lambda s: lib.t3d_pool_bwd_mid(fptr(slabs), fptr(out), tp(), n, mx, ref, s), marg))
                plan.lanes.append(plan._lane)
        return ({k: slabs[v:] for k, v in soff.items()}, {name: out[ooff[name]:ooff[name] + numel] for name, numel, _ in regions},
                emit)

    def _wgrad_gram(self, plan):
        """dW of a pooled layer from the K x K Gram matrix of its input (t3d.h K11e): gram + column sums of the input,
        one slab reduction, then the assembly kernel.  Independent of the dgrad chain."""
        g, rt, K, N = self.g, self.g.rt, self.K, self.N
        rps = gram_rows_per_split(rt.lib, self.M, K, self.src.dtype)
        regions = [('G', K * K, self.M // rps), ('abar', K, self.T)]
        merged = not SIDE_STREAM            # one stream: P / rowconst ride in the same slab reduction (one launch less)
        if merged:
            nch = (N + 127) // 128
            regions += [('P', K * K, nch), ('rowconst', K, nch)]
        sl, red, emit_reduce = self._gram_reduce(plan, regions)
        a = abi.PointMlpGramArgs()
        a.arith = self.g.rt.arith
        a.a, a.slabs = self.src.struct(), fptr(sl['G'])
        a.M, a.K, a.rows_per_frustum, a.rows_per_split = self.M, K, g.rpf, rps
        plan.add('t3d_pointmlp_gram', a)
        c = abi.ActColsumArgs()
        c.a, c.M, c.K, c.rows_per_frustum, c.part = self.src.struct(), self.M, K, g.rpf, fptr(sl['abar'])
        plan.add('t3d_act_colsum', c)
        if merged:
            self._emit_prep(plan, sl)
            self._gram_prepped = (plan, red)
        emit_reduce()
        f = abi.PoolWgradFinishArgs()
        f.a, f.argidx, f.dpool, f.coef = self.src.struct(), iptr(self.argidx), fptr(self.dpool), fptr(self.coef)
        f.w, f.bias, f.g, f.abar = fptr(self.w), fptr(self.bias), fptr(red['G']), fptr(red['abar'])
        f.B, f.K, f.N, f.rows_per_frustum = g.B, K, N, g.rpf
        goff = g.vars.offset(self.w_name) + self.w_row0 * N
        f.dw = fptr(g.vars.grads[goff:goff + K * N])
        plan.add('t3d_pool_wgrad_finish', f)
        self._gram_keep_w = (sl, red)

    def _dgrad_gram(self, plan):
        """Input gradient of a pooled layer: P / rowconst / wc, the sparse argmax rows, then the act(a).P GEMM with
        the dgrad epilogue."""
        g, K, N = self.g, self.K, self.N
        prev = self.src.producer
        assert prev is not None and not prev.pool, 'pooled layers follow a dense per-point layer in every reference net'
        prev._ensure_bwd_buffers()
        prepped = getattr(self, '_gram_prepped', (None, None))
        if prepped[0] is plan:
            red = prepped[1]
        else:
            nch = (N + 127) // 128
            sl, red, emit_reduce = self._gram_reduce(plan, [('P', K * K, nch), ('rowconst', K, nch)])
            self._emit_prep(plan, sl)
            emit_reduce()
            self._gram_keep_d = (sl, red)
        r = abi.PoolSparseRowsArgs(iptr(self.argidx), fptr(self.dpool), fptr(self.wc), g.B, N, K, g.rpf, fptr(self.S),
                                   iptr(self.S_live))
        plan.add('t3d_pool_sparse_rows', r)
        a = abi.PointMlpDgradGramArgs()
        a.arith = self.g.rt.arith
        a.a, a.p, a.rowconst, a.add_in = self.src.struct(), fptr(red['P']), fptr(red['rowconst']), fptr(self.S)
        a.add_live = iptr(self.S_live)
        a.prev_y, a.prev_scale, a.prev_shift = fptr(prev.y), fptr(prev.scale), fptr(prev.shift)
        a.out, a.psum_dz, a.psum_dzy = fptr(prev.dz), fptr(prev.psum_dz), fptr(prev.psum_dzy)
        a.M, a.K, a.rows_per_frustum, a.dtype = self.M, K, g.rpf, self.dt
        plan.add('t3d_pointmlp_dgrad_gram', a)

    def _bwd_pair_gram(self, plan):
        """Both gradients of a pooled layer in five launches: stage 1 (Gram slabs + column sums + P/rowconst slabs),
        one slab reduction, the sparse argmax rows, stage 2 (dW assembly + input-gradient GEMM)."""
        g, rt, K, N = self.g, self.g.rt, self.K, self.N
        lib = rt.lib
        prev = self.src.producer
        assert prev is not None and not prev.pool
        prev._ensure_bwd_buffers()
        rps = gram_rows_per_split(lib, self.M, K, self.src.dtype)
        nch = (N + 127) // 128
        sl, red, emit_reduce = self._gram_reduce(plan, [('G', K * K, self.M // rps), ('abar', K, self.T), ('P', K * K, nch),
                                                        ('rowconst', K, nch)])
        ga = abi.PointMlpGramArgs()
        ga.arith = self.g.rt.arith
        ga.a, ga.slabs = self.src.struct(), fptr(sl['G'])
        ga.M, ga.K, ga.rows_per_frustum, ga.rows_per_split = self.M, K, g.rpf, rps
        ca = abi.ActColsumArgs()
        ca.a, ca.M, ca.K, ca.rows_per_frustum, ca.part = self.src.struct(), self.M, K, g.rpf, fptr(sl['abar'])
        qa = abi.PoolBwdPrepArgs(fptr(self.w), fptr(self.bias), fptr(self.coef), K, N, fptr(sl['P']), fptr(sl['rowconst']),
                                 fptr(self.wc))
        fn1, r1 = lib.t3d_pool_bwd_stage1, (C.byref(ga), C.byref(ca), C.byref(qa))
        plan.keep.extend([ga, ca, qa, sl, red])
        plan.calls.append(('t3d_pool_bwd_stage1', lambda s: fn1(r1[0], r1[1], r1[2], s), (ga, ca, qa)))
        plan.lanes.append(0)
        r = abi.PoolSparseRowsArgs(iptr(self.argidx), fptr(self.dpool), fptr(self.wc), g.B, N, K, g.rpf, fptr(self.S),
                                   iptr(self.S_live))
        emit_reduce(sparse=r)           # slab reduction + sparse rows share a launch (both wait only for stage 1)
        f = abi.PoolWgradFinishArgs()
        f.a, f.argidx, f.dpool, f.coef = self.src.struct(), iptr(self.argidx), fptr(self.dpool), fptr(self.coef)
        f.w, f.bias, f.g, f.abar = fptr(self.w), fptr(self.bias), fptr(red['G']), fptr(red['abar'])
        f.B, f.K, f.N, f.rows_per_frustum = g.B, K, N, g.rpf
        goff = g.vars.offset(self.w_name) + self.w_row0 * N
        f.dw = fptr(g.vars.grads[goff:goff + K * N])
        d = abi.PointMlpDgradGramArgs()
        d.arith = self.g.rt.arith
        d.a, d.p, d.rowconst, d.add_in = self.src.struct(), fptr(red['P']), fptr(red['rowconst']), fptr(self.S)
        d.add_live = iptr(self.S_live)
        d.prev_y, d.prev_scale, d.prev_shift = fptr(prev.y), fptr(prev.scale), fptr(prev.shift)
        d.out, d.psum_dz, d.psum_dzy = fptr(prev.dz), fptr(prev.psum_dz), fptr(prev.psum_dzy)
        d.M, d.K, d.rows_per_frustum, d.dtype = self.M, K, g.rpf, self.dt
        fn2, r2 = lib.t3d_pool_bwd_stage2, (C.byref(f), C.byref(d))
        plan.keep.extend([f, d])
        plan.calls.append(('t3d_pool_bwd_stage2', lambda s: fn2(r2[0], r2[1], s), (f, d)))
        plan.lanes.append(0)

    def _emit_prep(self, plan, sl):
        q = abi.PoolBwdPrepArgs(fptr(self.w), fptr(self.bias), fptr(self.coef), self.K, self.N, fptr(sl['P']),
                                fptr(sl['rowconst']), fptr(self.wc))
        plan.add('t3d_pool_bwd_prep', q)

    def wgrad(self, plan):
        """Weight gradient: off the critical path, recorded on the plan's side lane.  (Called on its own for the first layer of a net: the
        last reader of its dz.)"""
        with plan.side():
            self._wgrad(plan)
        self._release_dz()

    def _wgrad_args(self, fused=False):
        g = self.g
        if fused and self.src.dtype == self.dt:
            rps = bwd_rows_per_split(g.rt.lib, self.M, self.K, self.N, self.dt)
        else:
            rps = wgrad_rows_per_split(g.rt.lib, self.M, self.K, self.N)
        n_slabs = self.M // rps
        goff = g.vars.offset(self.w_name) + self.w_row0 * self.N
        soff = g.ws.reserve(goff, self.K * self.N, n_slabs)
        a = abi.PointMlpWgradArgs()
        a.arith = self.g.rt.arith
        a.a, a.dy = self.src.struct(), self.dy_struct()
        a.M, a.K, a.N, a.rows_per_frustum, a.rows_per_split = self.M, self.K, self.N, g.rpf, rps
        g.deferred_slab_ptrs.append((a, 'slabs', soff))
        return a

    def _wgrad(self, plan):
        if self.gram:
            return self._wgrad_gram(plan)
        plan.add('t3d_pointmlp_wgrad', self._wgrad_args())

    def _dgrad_args(self, out_raw=None, add_in=None):
        prev = self.src.producer if out_raw is None else None
        a = abi.PointMlpDgradArgs()
        a.arith = self.g.rt.arith
        a.dy, a.w, a.add_in, a.dtype = self.dy_struct(), fptr(self.w_mm), fptr(add_in), self.dt
        a.w_x3, a.w_x3_stride = self.w_x3_d if self.NA == self.N else (None, 0)      # (the data-gradient arrangement of the fragment planes)
        if prev is not None:
            prev._ensure_bwd_buffers()
            assert not prev.pool
            a.prev_y, a.prev_scale, a.prev_shift = fptr(prev.y), fptr(prev.scale), fptr(prev.shift)
            a.out, a.psum_dz, a.psum_dzy = fptr(prev.dz), fptr(prev.psum_dz), fptr(prev.psum_dzy)
        else:
            a.out = fptr(out_raw)
        a.M, a.K, a.N, a.rows_per_frustum = self.M, self.K, self.N, self.g.rpf
        return a

    def dgrad(self, plan, out_raw=None, add_in=None):
        """Input gradient.  If the input's producer is a PointLayer, writes its dz (ReLU-masked) and
        batch-norm-backward partials; otherwise writes the raw gradient into `out_raw`."""
        if self.gram:
            assert out_raw is None and add_in is None
            self._dgrad_gram(plan)
        else:
            plan.add('t3d_pointmlp_dgrad', self._dgrad_args(out_raw, add_in))
        self._release_dz()      # (bwd_pair records the weight gradient BEFORE it calls this; alone: a frozen net's layer, no weight gradient)

    def bwd_pair(self, plan, out_raw=None, add_in=None):
        """Weight gradient + input gradient.  Dense layers: ONE launch (t3d_pointmlp_bwd); pooled layers: the Gram path."""
        if self.gram and FUSE_BWD and not SIDE_STREAM:
            assert out_raw is None and add_in is None
            return self._bwd_pair_gram(plan)
        if self.gram or self.pool or not FUSE_BWD:
            with plan.side():
                self._wgrad(plan)
            return self.dgrad(plan, out_raw=out_raw, add_in=add_in)
        d, w = self._dgrad_args(out_raw, add_in), self._wgrad_args(fused=True)
        fn, dref, wref = self.g.rt.lib.t3d_pointmlp_bwd, C.byref(d), C.byref(w)
        plan.keep.extend([d, w])
        plan.calls.append(('t3d_pointmlp_bwd', lambda s: fn(dref, wref, s), (d, w)))
        plan.lanes.append(0)
        self._release_dz()

    def dy_colsum(self, plan, alpha=1.0):
        """[B,N] per-frustum column sums of dy (dense layers only)."""
        g = self.g
        out = g.rt.zeros(g.B, self.N)
        a = abi.DyColsumArgs(fptr(self.psum_dz), fptr(self.psum), fptr(self.coef), g.B, self.N, g.rpf // TILE, g.rpf,
                             alpha, fptr(out))
        plan.add('t3d_dy_colsum', a)
        return out


class FcLayer:
    """tf_util.fully_connected (+ batch_norm over the batch + activation) + following dropout."""

    IDENTITY = 'identity'      # w=IDENTITY: no matmul (a standalone batch_norm_for_fc / dropout node on a [B,N] tensor)

    def __init__(self, g, scope, K, N, bn=True, act='relu', K2=0, w=None, bias='own', keep_prob=None, drop_scope=None, bn_scope=None):
        self.g, self.scope, self.K, self.K2, self.N, self.bn, self.act = g, scope, K, K2, N, bn, act
        rt, vs = g.rt, g.vars
        B = g.B
        bn_scope = bn_scope or scope + '/bn'
        if isinstance(w, str) and w == FcLayer.IDENTITY:
            assert K == N and K2 == 0
            w, self.w_grad = None, None
        elif w is None:
            w = vs.xavier(scope + '/weights', (K + K2, N), K + K2, N)
            self.w_grad = vs.grad(scope + '/weights')
        else:
            self.w_grad = None
        self.w = w
        if isinstance(bias, str):
            self.bias = vs.const(scope + '/biases', (N,), 0.0)
            self.bias_grad = vs.grad(scope + '/biases')
        else:
            self.bias, self.bias_grad = bias, None
        if bn:
            self.gamma = vs.const(bn_scope + '/gamma', (N,), 1.0)
            self.beta = vs.const(bn_scope + '/beta', (N,), 0.0)
            self.mm = vs.const(bn_scope + '/moving_mean', (N,), 0.0, trainable=False)
            self.mv = vs.const(bn_scope + '/moving_variance', (N,), 1.0, trainable=False)
            self.mean, self.invstd = rt.zeros(N), rt.zeros(N)
        self.y = rt.zeros(B, N)
        self.out = rt.zeros(B, N)
        self.dy = None
        self.keep_prob = keep_prob
        self.drop_scope = drop_scope
        self.drop_mask = rt.full((B, N), 1.0) if keep_prob is not None else None
        if keep_prob is not None:
            g.dropout_masks[drop_scope] = (self.drop_mask, keep_prob)

    def fwd(self, plan, x, ld_in, is_training, in2=None, ld_in2=0, add_in=None, ld_add=0, add_n=0):
        g = self.g
        self.x, self.ld_in, self.in2, self.ld_in2, self.is_training = x, ld_in, in2, ld_in2, is_training
        a = abi.FcFwdArgs()
        a.in_, a.ld_in, a.K, a.in2, a.ld_in2, a.K2 = fptr(x), ld_in, self.K, fptr(in2), ld_in2, self.K2
        a.w, a.bias = fptr(self.w), fptr(self.bias)
        if self.bn:
            a.gamma, a.beta, a.moving_mean, a.moving_var = fptr(self.gamma), fptr(self.beta), fptr(self.mm), fptr(self.mv)
            a.mean, a.invstd = fptr(self.mean), fptr(self.invstd)
        a.decay, a.eps, a.is_training, a.unbiased_ema = fptr(g.bn_decay_ptr), BN_EPS, int(is_training), int(g.unbiased_ema)
        a.act, a.leaky_alpha = abi.ACT_BY_NAME[self.act], 0.2
        self.use_drop = self.drop_mask is not None and is_training
        if self.use_drop:
            a.drop_mask, a.keep_prob = fptr(self.drop_mask), self.keep_prob
        a.add_in, a.ld_add, a.add_n = fptr(add_in), ld_add, add_n
        a.y, a.out, a.ld_out, a.B, a.N = fptr(self.y), fptr(self.out), self.N, g.B, self.N
        plan.add('t3d_fc_fwd', a)
        return self.out

    def bwd(self, plan, dout=None, ld_dout=0, nxt=None, param_grads=True):
        g, vs = self.g, self.g.vars
        if self.dy is None:
            self.dy = g.rt.zeros(g.B, self.N)
        a = abi.FcBwdArgs()
        if dout is not None:
            a.dout, a.ld_dout = fptr(dout), ld_dout
        else:
            a.dy_next, a.w_next, a.N_next = fptr(nxt.dy), fptr(nxt.w), nxt.N
        a.in_, a.ld_in, a.K, a.in2, a.ld_in2, a.K2 = fptr(self.x), self.ld_in, self.K, fptr(self.in2), self.ld_in2, self.K2
        a.y, a.out, a.ld_out = fptr(self.y), fptr(self.out), self.N
        if self.bn:
            a.gamma, a.beta, a.mean, a.invstd = fptr(self.gamma), fptr(self.beta), fptr(self.mean), fptr(self.invstd)
        a.bn_training = int(self.is_training)
        a.act, a.leaky_alpha = abi.ACT_BY_NAME[self.act], 0.2
        if self.use_drop:
            a.drop_mask, a.keep_prob = fptr(self.drop_mask), self.keep_prob
        a.dy = fptr(self.dy)
        if param_grads:
            a.dw = fptr(self.w_grad) if self.w_grad is not None else fptr(None)
            a.dbias = fptr(self.bias_grad)
            if self.bn and self.is_training:
                a.dgamma, a.dbeta = fptr(vs.grad(self.scope + '/bn/gamma')), fptr(vs.grad(self.scope + '/bn/beta'))
        a.B, a.N = g.B, self.N
        plan.add('t3d_fc_bwd', a)

    def dinput(self, plan, K=None, alpha=1.0, add_in=None, ld_add=0, bn_bwd_of=None, param_grads=True):
        """[B,K] gradient w.r.t. the first K input columns.  `bn_bwd_of`: the max-pooled PointLayer whose pooled feature
        these columns are -- its batch-norm-backward finalize (pooled form) then runs inside this launch and the layer's
        later bn_bwd(plan, dpool_in=<this result>) call is a no-op."""
        g = self.g
        K = K or self.K
        out = g.rt.zeros(g.B, K)
        a = abi.FcDinputArgs()
        a.dy, a.N, a.w, a.add_in, a.ld_add, a.alpha, a.din, a.ld_din, a.B, a.K = fptr(self.dy), self.N, fptr(self.w), fptr(add_in), \
            ld_add, alpha, fptr(out), K, g.B, K
        L = bn_bwd_of
        if L is not None and FUSE_BWD:
            assert L.pool and L.N == K
            L._ensure_bwd_buffers()
            vs = g.vars
            a.bn_pooled, a.bn_ld_pooled, a.bn_ysel, a.bn_dpool, a.bn_count = fptr(L.pooled), L.N, fptr(L.ysel), fptr(L.dpool), L.M
            a.bn_gamma, a.bn_mean, a.bn_invstd, a.bn_scale = fptr(L.gamma), fptr(L.mean), fptr(L.invstd), fptr(L.scale)
            a.bn_frozen = int(not L.is_training)
            if param_grads and L.is_training:
                a.bn_dgamma, a.bn_dbeta = fptr(vs.grad(L.scope + '/bn/gamma')), fptr(vs.grad(L.scope + '/bn/beta'))
            a.bn_coef = fptr(L.coef)
            L._bn_bwd_fused_in = plan
        plan.add('t3d_fc_dinput', a)
        return out
