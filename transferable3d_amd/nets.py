"""Graph context + the sub-networks of the hot path as fused launch schedules.

Each class mirrors one builder of the reference's `semisup_models.py` and emits, into the graph's
plans, the kernel launches of its forward pass and (separately) of its backward pass:
  InstSegNet  <- v1_inst_seg (semisup_models.py:69-139) + subtract_points_mean (145-162)
  TNet        <- v1_tnet (164-202)
  BoxEstNet   <- subtract_1st_stage_center (204-209) + v1_box_est (215-291)
  StrongLoss  <- get_strong_loss (semisup_v1_sunrgbd.py:423-553) + anchor->reg (tf_util.py:1001-1041)
"""
import ctypes as C
import os

import numpy as np
import torch

from . import abi
from .abi import fptr, iptr
from .constants import BOX_OUT_DIMS, NUM_CLASS
from .engine import ActSpec, FcLayer, Plan, PointLayer, Runtime, VarStore, Workspace, TILE


class Graph:
    """Static launch graph for one (batch_size, num_point, num_channel) problem; the analogue of the
    reference's tf.Graph + tf.Session (train_semisup.py:204-277)."""

    def __init__(self, batch_size, num_point, num_channel, rt=None, seed=0, unbiased_ema=True, vars=None, dtype='f32'):
        """dtype: 'f32' = exact-fp32 MFMA path (BASELINE configs[1..3]); 'bf16' = bf16 activation storage + bf16 MFMA with fp32
        accumulation, statistics, parameters and optimiser (configs[4])."""
        assert dtype in ('f32', 'bf16')
        self.dtype = dtype
        self.dt = abi.DTYPE_BY_NAME[dtype]                                   # t3d.h T3D_F32 / T3D_BF16
        self.adt = torch.bfloat16 if dtype == 'bf16' else torch.float32      # storage of the [M, C] layer tensors
        if num_point % TILE:
            raise abi.T3DError('num_point must be a multiple of %d' % TILE)
        self.rt = rt or Runtime()
        self.B, self.rpf, self.C = batch_size, num_point, num_channel
        self.ldpc = (num_channel + 3) // 4 * 4     # row stride of the point cloud in HBM: 16-byte rows for the float4 operand loads (C = 6 -> 8)
        self.M = batch_size * num_point
        self.vars = vars or VarStore(self.rt, seed=seed)
        self.ws = Workspace(self.rt)
        self.unbiased_ema = unbiased_ema
        self.hyper = self.rt.zeros(4)              # step, lr, bn_decay, adam lr_t
        self.hyper[2] = 0.5
        self.bn_decay_ptr = self.hyper[2:3]
        self.dropout_masks = {}                    # scope -> (mask tensor, keep_prob)
        # the seg head's [M,128] keep mask generated inside the head kernel instead of stored (production drivers); parity tests
        # inject explicit masks and leave this off
        self.inline_dropout, self.dropout_seed = False, 1234
        self.deferred_slab_ptrs = []
        # data parallelism: gradient buckets [[(offset, n), ...], ...] in the order the backward completes them (declare_bucket)
        self.dp_buckets = False                    # set before emit_backward: cut the backward into buckets (step.TrainStep)
        self.buckets = []
        self.trained_prefixes = None               # var_list of the optimiser (None: everything)
        self._slabs_reduced = 0
        self.fwd = Plan(self.rt)
        self.bwd = Plan(self.rt)
        self.opt = Plan(self.rt)
        self.pre = Plan(self.rt)                   # schedule + dropout mask generation
        self.finalized = False

    def finalize(self):
        self.ws.finalize()
        base = self.ws.buf.data_ptr()
        for a, field, soff in self.deferred_slab_ptrs:
            setattr(a, field, C.cast(C.c_void_p(base + 4 * soff), abi.F))
        self.finalized = True
        return self

    # ---- optimiser ---------------------------------------------------------------------------------
    def emit_reduce_slabs(self, plan):
        """Sums the weight-gradient slabs recorded since the previous call into the gradient buffer (fixed order, no atomics).
        Must follow the wgrads it covers; pointers are resolved at run time (post-finalize)."""
        ws, vs, lib = self.ws, self.vars, self.rt.lib
        plan.join()                       # the weight gradients were recorded on the side lane
        i0, i1 = self._slabs_reduced, len(ws.entries)
        self._slabs_reduced = i1
        if i1 == i0:
            return
        mx = max(e[2] for e in ws.entries[i0:i1])
        stride = C.sizeof(abi.SlabDesc)

        def thunk(s):
            return lib.t3d_reduce_slabs(fptr(ws.buf), fptr(vs.grads),
                                        C.cast(C.c_void_p(ws.table.data_ptr() + i0 * stride), C.POINTER(abi.SlabDesc)), i1 - i0, mx, s)
        plan.add_raw('t3d_reduce_slabs', thunk)

    def emit_cast_weights(self, plan):
        """bf16 path: refresh the bf16 copy of the weights the GEMM kernels read (the optimiser updates the fp32 master copy).
        fp32 path on the bf16 matrix pipe: refresh the three bf16 planes of the weights (t3d_split_x3), if any layer asked for them."""
        vs, lib = self.vars, self.rt.lib
        if self.dt != abi.BF16:
            from .engine import X3_PRESPLIT
            if (X3_PRESPLIT and self.rt.device.type == 'cuda' and self.rt.arith != abi.ARITH_FP32_MFMA and getattr(vs, 'x3_frag_enabled', True)
                    and hasattr(lib, 't3d_split_x3_frag')):
                def thunk(s):      # (the table is read when the launch is issued: every layer registered by then is in it)
                    if not getattr(vs, 'x3_frag_entries', None):
                        return 0
                    t, n, nblk = vs.frag_table()
                    pf, pd = vs.x3_frag_planes
                    return lib.t3d_split_x3_frag(fptr(vs.params), C.c_void_p(pf.data_ptr()), C.c_void_p(pd.data_ptr()), vs.x3_frag_stride,
                                                 C.c_void_p(t.data_ptr()), n, nblk, s)
                plan.add_raw('t3d_split_x3_frag', thunk)
            return
        p16 = vs.enable_bf16()
        plan.add_raw('t3d_cast_bf16', lambda s: lib.t3d_cast_bf16(fptr(vs.params), C.c_void_p(p16.data_ptr()), vs.used, s))

    def default_bucket(self):
        """Every trained range: the one bucket of a step whose backward declares none."""
        return self.vars.trainable_ranges(self.trained_prefixes)

    def declare_bucket(self, plan, prefixes):
        """Data-parallel gradient bucket: the trainable variables under `prefixes` are complete once the launches recorded so far
        (and one slab reduction emitted here) have run.  No-op unless `dp_buckets` is set."""
        if not self.dp_buckets:
            return
        self.emit_reduce_slabs(plan)
        self.buckets.append(self.vars.trainable_ranges(prefixes))
        plan.bucket_ready(len(self.buckets) - 1)

    def emit_batch_assemble(self, plan, dataset, inputs, seed=0, **aug):
        """The input pipeline as a launch of the step: batch slot b of step s takes frustum perm[(s*B + b) % F] of the
        HBM-resident data set (dataset.DeviceFrustumSet) -- recorded before the schedule kernel advances the step counter."""
        equal_prob = aug.pop('equal_class_prob', 0.0)
        if equal_prob > 0.0:
            # class-balanced composition (equal_samples_per_class): a sampler launch writes the frustum index of every slot, the
            # assembly then runs in its explicit-sample mode
            slots = self.rt.zeros(self.B, dtype=torch.int32)
            alternate = bool(aug.pop('alternate', False))
            plan.add('t3d_sample_equal_classes', dataset.sample_equal_args(
                self.hyper, self.B, slots, is_data_2D=inputs.is_data_2D if alternate else None, seed=seed ^ 0x2545F491,
                equal_prob=equal_prob, alternate=alternate))
            a = dataset.assemble_args(inputs, self.hyper, self.B, self.rpf, self.C, seed=seed, sample=slots, **aug)
            if alternate:
                a.is_data_2D = iptr(None)          # the sampler launch knows which list the step drew from
                a.slot_is_2D = iptr(inputs.is_data_2D)
            plan.add('t3d_batch_assemble', a)
            return
        a = dataset.assemble_args(inputs, self.hyper, self.B, self.rpf, self.C, seed=seed, **aug)
        plan.add('t3d_batch_assemble', a)

    def emit_boxpc_perturb(self, plan, inputs, c, seed=0):
        """Box-PC Fit training samples on the device (box_pc_fit_dataset.py:211-244): after the batch is assembled, perturb every
        label box until its IoU falls inside the fit / no-fit bounds."""
        from .dataset import boxpc_perturb_args
        plan.add('t3d_boxpc_perturb', boxpc_perturb_args(inputs, self.hyper, self.B, c, seed=seed))

    def emit_schedule(self, plan, sched):
        lib, hyper = self.rt.lib, self.hyper
        plan.add_raw('t3d_schedule_step', lambda s: lib.t3d_schedule_step(fptr(hyper), C.byref(sched), s), sched)

    def emit_dropout_masks(self, plan, seed=1234):
        lib, hyper = self.rt.lib, self.hyper
        for i, (scope, (mask, keep)) in enumerate(sorted(self.dropout_masks.items())):
            plan.add_raw('t3d_dropout_mask',
                         lambda s, mask=mask, keep=keep, sd=seed + 7919 * i: lib.t3d_dropout_mask(
                             fptr(mask), mask.numel(), keep, sd, fptr(hyper), s))

    def emit_adam(self, plan, prefixes=None, beta1=0.9, beta2=0.999, eps=1e-8, grad_scale=1.0):
        """TF-form Adam over the trained ranges.  With gradient buckets declared by the backward (data parallel): one launch per
        bucket range behind that bucket's wait marker, in the order the buckets complete."""
        lib, vs, hyper = self.rt.lib, self.vars, self.hyper
        vs.optimizer_kind = 'adam'

        def adam(off, n):
            plan.add_raw('t3d_adam_tf_step',
                         lambda s, off=off, n=n: lib.t3d_adam_tf_step(
                             fptr(vs.params[off:]), fptr(vs.grads[off:]), fptr(vs.adam_m[off:]), fptr(vs.adam_v[off:]),
                             n, fptr(hyper), beta1, beta2, eps, grad_scale, s))
        self._emit_optimizer(plan, prefixes, adam)

    def emit_momentum(self, plan, prefixes=None, momentum=0.9, grad_scale=1.0):
        """tf.train.MomentumOptimizer (`--optimizer momentum`, train_semisup.py:226-228) over the trained ranges; the accumulator
        (TF slot `Momentum`) lives in the first moment buffer."""
        lib, vs, hyper = self.rt.lib, self.vars, self.hyper
        vs.optimizer_kind = 'momentum'

        def mom(off, n):
            plan.add_raw('t3d_momentum_step',
                         lambda s, off=off, n=n: lib.t3d_momentum_step(
                             fptr(vs.params[off:]), fptr(vs.grads[off:]), fptr(vs.adam_m[off:]), n, fptr(hyper), momentum, grad_scale, s))
        self._emit_optimizer(plan, prefixes, mom)

    def _emit_optimizer(self, plan, prefixes, launch):
        vs = self.vars
        self.trained_prefixes = prefixes
        ranges = vs.trainable_ranges(prefixes)
        if self.buckets:
            covered = sorted(r for b in self.buckets for r in b)
            assert sum(n for _, n in covered) == sum(n for _, n in ranges) and \
                all(any(o >= ro and o + n <= ro + rn for ro, rn in ranges) for o, n in covered), \
                'gradient buckets must partition the trained variables'
            for i, b in enumerate(self.buckets):
                plan.bucket_wait(i)
                for off, n in b:
                    launch(off, n)
            return
        for off, n in ranges:
            launch(off, n)


def make_schedule(batch_size, base_lr=1e-3, decay_step=800000, decay_rate=0.5, bn_init_decay=0.5,
                  bn_decay_rate=0.5, bn_decay_clip=0.99, beta1=0.9, beta2=0.999):
    """train_semisup.py:53-67,127-145 defaults."""
    return abi.Schedule(base_lr, decay_rate, float(decay_step), bn_init_decay, bn_decay_rate, float(decay_step),
                        bn_decay_clip, beta1, beta2, batch_size, 0)


class InstSegNet:
    def __init__(self, g, scope, use_one_hot):
        self.g, self.scope = g, scope
        C_, vs = g.C, g.vars
        s = scope + '/'
        self.L1 = PointLayer(g, s + 'conv1', C_, 64, kernel_1xD=True)
        self.L2 = PointLayer(g, s + 'conv2', 64, 64)
        self.L3 = PointLayer(g, s + 'conv3', 64, 64)
        self.L4 = PointLayer(g, s + 'conv4', 64, 128)
        self.L5 = PointLayer(g, s + 'conv5', 128, 1024, pool=True)
        # conv6 reads [point_feat(64) | global(1024) | one_hot]; split into a per-point K=64 GEMM and a
        # per-frustum FC on the global feature (identical math, tile+concat never materialised).
        self.oh = NUM_CLASS if use_one_hot else 0
        k6 = 64 + 1024 + self.oh
        w6 = vs.xavier(s + 'conv6/weights', (1, 1, k6, 512), k6, 512).view(k6, 512)
        self.L6 = PointLayer(g, s + 'conv6', 64, 512, w=w6[0:64], w_name=s + 'conv6/weights', w_row0=0)
        self.G6 = FcLayer(g, s + 'conv6/global', 1024, 512, bn=False, act=None, K2=self.oh, w=w6[64:], bias=None)
        self.G6.w_grad = vs.grad(s + 'conv6/weights').view(k6, 512)[64:]
        self.L7 = PointLayer(g, s + 'conv7', 512, 256)
        self.L8 = PointLayer(g, s + 'conv8', 256, 128)
        self.L9 = PointLayer(g, s + 'conv9', 128, 128)
        self.w10 = vs.xavier(s + 'conv10/weights', (1, 1, 128, 2), 128, 2).view(128, 2)
        self.b10 = vs.const(s + 'conv10/biases', (2,), 0.0)
        rt, M, T, B = g.rt, g.M, g.M // TILE, g.B
        self.drop_mask = None
        if not g.inline_dropout:
            self.drop_mask = rt.full((M, 128), 1.0)
            g.dropout_masks[s + 'dp1'] = (self.drop_mask, 0.5)
        self.logits, self.mask = rt.zeros(M, 2), rt.zeros(M)
        self.part = rt.zeros(T, 8)
        self.mask_xyz_mean, self.seg_loss, self.n_correct = rt.zeros(B, 3), rt.zeros(B), rt.zeros(1)

    def fwd(self, plan, pc, one_hot, labels, is_data_2D, is_training, train_seg, ce_weight=1.0, oracle_mask=None):
        """`train_seg`: emit the seg-loss backward (dz of conv9 etc.) inside the head kernel.  `oracle_mask` ([B,N] int32, the
        y_seg buffer): the head's logits become stack([1 - m, m]) (semisup_v1_sunrgbd.py:161-162)."""
        g, rt = self.g, self.g.rt
        M, T = g.M, g.M // TILE
        a = ActSpec(pc, g.ldpc, g.C)
        a = self.L1.fwd(plan, a, is_training)
        a = self.L2.fwd(plan, a, is_training)
        a3 = self.L3.fwd(plan, a, is_training)
        a = self.L4.fwd(plan, a3, is_training)
        self.L5.fwd(plan, a, is_training)
        rb = self.G6.fwd(plan, self.L5.pooled, 1024, is_training, in2=one_hot if self.oh else None, ld_in2=NUM_CLASS)
        a = self.L6.fwd(plan, a3, is_training, rowbias=rb)
        a = self.L7.fwd(plan, a, is_training)
        a = self.L8.fwd(plan, a, is_training)
        self.L9.fwd(plan, a, is_training)
        self.train_seg = train_seg
        h = abi.SegHeadArgs()
        L9 = self.L9
        h.y, h.scale, h.shift = fptr(L9.y), fptr(L9.scale), fptr(L9.shift)
        if is_training:
            h.drop_mask, h.keep_prob = fptr(self.drop_mask), 0.5
            if self.drop_mask is None:
                h.drop_seed, h.drop_hyper = (g.dropout_seed + 0x5EED) & 0xffffffff, fptr(g.hyper)
        h.w, h.bias = fptr(self.w10), fptr(self.b10)
        h.labels, h.is_data_2D = iptr(labels), iptr(is_data_2D)
        h.pc, h.ld_pc, h.ce_weight = fptr(pc), g.ldpc, ce_weight
        h.logits, h.mask, h.part = fptr(self.logits), fptr(self.mask), fptr(self.part)
        if train_seg:
            L9._ensure_bwd_buffers()
            h.dz, h.psum_dz, h.psum_dzy = fptr(L9.dz), fptr(L9.psum_dz), fptr(L9.psum_dzy)
            # per-tile conv10 weight-gradient partials live in the slab workspace: summed by t3d_reduce_slabs
            soff = g.ws.reserve(g.vars.offset(self.scope + '/conv10/weights'), 256, T)
            g.deferred_slab_ptrs.append((h, 'dw_part', soff))
        h.M, h.K, h.rows_per_frustum, h.B, h.dtype = M, 128, g.rpf, g.B, g.dt
        h.oracle_mask = iptr(oracle_mask)
        plan.add('t3d_seg_head', h)
        f = abi.SegFinalizeArgs()
        f.part, f.B, f.tiles_per_frustum, f.rows_per_frustum, f.K = fptr(self.part), g.B, g.rpf // TILE, g.rpf, 128
        f.mask_xyz_mean, f.seg_loss, f.n_correct = fptr(self.mask_xyz_mean), fptr(self.seg_loss), fptr(self.n_correct)
        if train_seg:
            f.dbias = fptr(g.vars.grad(self.scope + '/conv10/biases'))
        plan.add('t3d_seg_finalize', f)
        self._head, self._head_slab, self._finalize = h, (soff if train_seg else None), f
        return self.logits

    def emit_head_again(self, plan, dsoft):
        """Second run of the fused head and its finalize with d loss / d soft_mask added to the logit gradients (weak surface loss,
        nets.WeakLoss): same forward values, conv9's gradient / partials / conv10 gradients rewritten."""
        assert self.train_seg, 'the soft-mask gradient needs the training backward of the seg head'
        h = abi.SegHeadArgs()
        C.memmove(C.byref(h), C.byref(self._head), C.sizeof(h))
        h.dsoft = fptr(dsoft)
        self.g.deferred_slab_ptrs.append((h, 'dw_part', self._head_slab))
        plan.add('t3d_seg_head', h)
        plan.add('t3d_seg_finalize', self._finalize)

    def bwd(self, plan, part=None):
        """part None: everything; 0 / 1 / 2: conv9..conv7 | conv6, conv5 | conv4..conv1 (the FC_SIDE schedule interleaves the
        other nets between them)."""
        assert self.train_seg
        g = self.g
        L = self
        if part in (None, 0):
            for lay in (L.L9, L.L8, L.L7):
                lay.bn_bwd(plan)
                lay.bwd_pair(plan)
        if part in (None, 1):
            self._bwd_mid(plan)
        if part in (None, 2):
            self._bwd_tail(plan)

    def _bwd_mid(self, plan):
        g, L = self.g, self
        L.L6.bn_bwd(plan)
        colsum6 = L.L6.dy_colsum(plan)                                   # [B,512]
        L.G6.bwd(plan, dout=colsum6, ld_dout=512)                        # dW6[64:], no bias
        dg5 = L.G6.dinput(plan, K=1024, bn_bwd_of=L.L5)                   # [B,1024] + conv5's BN-bwd finalize
        self.da3_part = g.rt.zeros(g.M, 64, dtype=g.adt)
        L.L6.bwd_pair(plan, out_raw=self.da3_part)
        L.L5.bn_bwd(plan, dpool_in=dg5, ld_dpool_in=1024)
        L.L5.bwd_pair(plan)

    def _bwd_tail(self, plan):
        L = self
        L.L4.bn_bwd(plan)
        L.L4.bwd_pair(plan, add_in=self.da3_part)
        for lay in (L.L3, L.L2):
            lay.bn_bwd(plan)
            lay.bwd_pair(plan)
        L.L1.bn_bwd(plan)
        L.L1.wgrad(plan)


class TNet:
    def __init__(self, g, scope, use_one_hot, box2d=False):
        self.g, self.scope = g, scope
        s = scope + '/'
        oh = (NUM_CLASS if use_one_hot else 0) + (4 if box2d else 0)      # fc1 reads [pooled | one_hot | norm_box2D] (semisup_models.py:192-195)
        self.oh = oh
        self.T1 = PointLayer(g, s + 'conv-reg1-stage1', 3, 128)
        self.T2 = PointLayer(g, s + 'conv-reg2-stage1', 128, 128)
        self.T3 = PointLayer(g, s + 'conv-reg3-stage1', 128, 256, pool=True)
        self.F1 = FcLayer(g, s + 'fc1-stage1', 256, 256, K2=oh)
        self.F2 = FcLayer(g, s + 'fc2-stage1', 256, 128)
        self.F3 = FcLayer(g, s + 'fc3-stage1', 128, 3, bn=False, act=None)

    def fwd(self, plan, pc, mask, mask_xyz_mean, one_hot, is_training, ld_oh=NUM_CLASS):
        """`one_hot` = the extra FC-input block [B, ld_oh]: the one-hot vector, or ExtraFeats' [one_hot | norm_box2D]."""
        g = self.g
        a = ActSpec(pc, g.ldpc, 3, sub=mask_xyz_mean, sub_ld=3)
        a = self.T1.fwd(plan, a, is_training)
        a = self.T2.fwd(plan, a, is_training)
        self.T3.fwd(plan, a, is_training, rowmask=mask)
        x = self.F1.fwd(plan, self.T3.pooled, 256, is_training, in2=one_hot if self.oh else None, ld_in2=ld_oh)
        x = self.F2.fwd(plan, x, 256, is_training)
        self.stage1_center = self.F3.fwd(plan, x, 128, is_training, add_in=mask_xyz_mean, ld_add=3, add_n=3)
        return self.stage1_center

    def bwd_fc(self, plan, dstage1):
        self.F3.bwd(plan, dout=dstage1, ld_dout=3)
        self.F2.bwd(plan, nxt=self.F3)
        self.F1.bwd(plan, nxt=self.F2)
        return self.F1.dinput(plan, K=256, bn_bwd_of=self.T3)

    def bwd(self, plan, dstage1):
        self.bwd_convs(plan, self.bwd_fc(plan, dstage1))

    def bwd_convs(self, plan, dft):
        self.T3.bn_bwd(plan, dpool_in=dft, ld_dpool_in=256)
        self.T3.bwd_pair(plan)
        self.T2.bn_bwd(plan)
        self.T2.bwd_pair(plan)
        self.T1.bn_bwd(plan)
        self.T1.wgrad(plan)


class BoxEstNet:
    def __init__(self, g, scope, use_one_hot, box2d=False):
        self.g, self.scope = g, scope
        s = scope + '/'
        oh = (NUM_CLASS if use_one_hot else 0) + (4 if box2d else 0)      # fc1 reads [pooled | one_hot | norm_box2D] (semisup_models.py:249-252)
        self.oh = oh
        self.B1 = PointLayer(g, s + 'conv-reg1', 3, 128)
        self.B2 = PointLayer(g, s + 'conv-reg2', 128, 128)
        self.B3 = PointLayer(g, s + 'conv-reg3', 128, 256)
        self.B4 = PointLayer(g, s + 'conv-reg4', 256, 512, pool=True)
        self.G1 = FcLayer(g, s + 'fc1', 512, 512, K2=oh)
        self.G2 = FcLayer(g, s + 'fc2', 512, 256)
        self.G3 = FcLayer(g, s + 'fc3', 256, BOX_OUT_DIMS, bn=False, act=None)

    def fwd(self, plan, pc, mask, stage1_center, one_hot, is_training, ld_oh=NUM_CLASS):
        self.fwd_convs(plan, pc, mask, stage1_center, is_training)
        return self.fwd_heads(plan, one_hot, is_training, ld_oh)

    def fwd_convs(self, plan, pc, mask, stage1_center, is_training):
        g = self.g
        a = ActSpec(pc, g.ldpc, 3, sub=stage1_center, sub_ld=3)
        a = self.B1.fwd(plan, a, is_training)
        a = self.B2.fwd(plan, a, is_training)
        a = self.B3.fwd(plan, a, is_training)
        self.B4.fwd(plan, a, is_training, rowmask=mask)
        self.feats_lv1 = self.B4.pooled
        return self.feats_lv1

    def fwd_heads(self, plan, one_hot, is_training, ld_oh=NUM_CLASS):
        """fc1-fc3 on the pooled feature (semisup_models.py:249-262).  Separate from the convolutions: in SEMI_MODEL F these heads feed
        end points only (W_pred_box), so the stage-c step runs them as a chain of its own beside the refinement branch."""
        self.feats_lv2 = self.G1.fwd(plan, self.feats_lv1, 512, is_training, in2=one_hot if self.oh else None,
                                     ld_in2=ld_oh)
        self.feats_lv3 = self.G2.fwd(plan, self.feats_lv2, 512, is_training)
        self.box_params = self.G3.fwd(plan, self.feats_lv3, 256, is_training)
        return self.box_params

    def bwd_convs(self, plan, dfeats, ld, dstage1_in):
        """Backward of the per-point stack given d(feats_lv1); returns the total d(stage1_center)."""
        self.B4.bn_bwd(plan, dpool_in=dfeats, ld_dpool_in=ld)
        self.B4.bwd_pair(plan)
        for lay in (self.B3, self.B2):
            lay.bn_bwd(plan)
            lay.bwd_pair(plan)
        self.B1.bn_bwd(plan)
        self.B1.wgrad(plan)
        # input = xyz - stage1_center  =>  d stage1_center -= sum_n dX = (per-frustum colsum of dy1) . W1^T
        colsum1 = self.B1.dy_colsum(plan)                                # [B,128]
        g = self.g
        out = g.rt.zeros(g.B, 3)
        a = abi.FcDinputArgs(fptr(colsum1), 128, fptr(self.B1.w), fptr(dstage1_in), 3, -1.0, fptr(out), 3, g.B, 3)
        plan.add('t3d_fc_dinput', a)
        return out

    def bwd_fc(self, plan, dbox):
        self.G3.bwd(plan, dout=dbox, ld_dout=BOX_OUT_DIMS)
        self.G2.bwd(plan, nxt=self.G3)
        self.G1.bwd(plan, nxt=self.G2)
        return self.G1.dinput(plan, K=512, bn_bwd_of=self.B4)

    def bwd(self, plan, dbox, dstage1_in):
        return self.bwd_convs(plan, self.bwd_fc(plan, dbox), 512, dstage1_in)


PAIR_SMALL = os.environ.get('T3D_PAIR', '1') != '0'
OVERLAP = os.environ.get('T3D_OVERLAP', '1') != '0'      # schedule.py: riders + pairs over [T-Net/box fwd, loss, bwd] || [seg bwd]
PAIR_POLICY = int(os.environ.get('T3D_PAIR_POLICY', '1'))      # both heads large: 1 = first chain first, 2 = second chain first (no
#                                                                 measurable difference: 1.5764 vs 1.5759 ms, three same-box A/B pairs)


def pair_small_launches(plan, i0, i1, i2):
    """Interleave two INDEPENDENT chains of a plan -- calls [i0, i1) and [i1, i2) -- so that their small launches (batch-norm backward
    finalizers, FC head kernels, column sums: 16-64 workgroups each, bound by the kernel boundary, not by their work) share
    launches pairwise (t3d_small_pair, csrc/pair.hip).  Each chain keeps its own order; a chain waits at a small launch until the
    other one has reached one too.  Results are bit-identical to the unpaired plan (same kernels, same arguments)."""
    lib = plan.rt.lib
    if not hasattr(lib, 't3d_small_pair'):
        return 0
    kinds = abi.SMALL_KIND
    A = list(zip(plan.calls[i0:i1], plan.lanes[i0:i1]))
    S = list(zip(plan.calls[i1:i2], plan.lanes[i1:i2]))

    def small(entry):
        (name, _, arg), lane = entry
        if lane != 0 or name not in kinds or arg is None:      # a side-lane op keeps its own queue order (Plan._run_two_streams)
            return False
        if name == 't3d_bn_bwd_finalize' and arg.psum_dz and arg.n_tiles > 512:      # the 64-group form runs alone
            return False
        return True

    out, n_pairs = [], 0
    while A and S:
        if small(A[0]) and small(S[0]):
            ((na, _, aa), _), ((nb, _, ab), _) = A.pop(0), S.pop(0)
            oa, ob = abi.SmallOp(), abi.SmallOp()
            for o, n_, a_ in ((oa, na, aa), (ob, nb, ab)):
                o.kind = kinds[n_]
                C.memmove(C.byref(o.u), C.byref(a_), C.sizeof(a_))
            fn, ra, rb = lib.t3d_small_pair, C.byref(oa), C.byref(ob)
            plan.keep.extend([oa, ob])
            out.append((('t3d_small_pair', (lambda s, fn=fn, ra=ra, rb=rb: fn(ra, rb, s)), (oa, ob)), 0))
            n_pairs += 1
        elif not small(A[0]) and not (PAIR_POLICY == 2 and not small(S[0])):
            out.append(A.pop(0))
        else:
            out.append(S.pop(0))
    out += A + S
    plan.calls[i0:i2] = [c for c, _ in out]
    plan.lanes[i0:i2] = [l for _, l in out]
    return n_pairs


class StrongLoss:
    def __init__(self, g):
        self.g = g
        rt, B = g.rt, g.B
        self.dbox, self.dstage1 = rt.zeros(B, BOX_OUT_DIMS), rt.zeros(B, 3)
        self.terms, self.total_losses, self.loss = rt.zeros(B, 8), rt.zeros(B), rt.zeros(1)
        self.center, self.reg_dims, self.reg_theta = rt.zeros(B, 3), rt.zeros(B, 3), rt.zeros(B)
        self.iou2d, self.iou3d = rt.zeros(B), rt.zeros(B)        # get_iou_summary (semisup_v1_sunrgbd.py:236-246)

    def emit(self, plan, box, stage1_center, seg_loss, labels, c, normalize_by_3d_count=False):
        y_center, y_orient_cls, y_orient_reg, y_dims_cls, y_dims_reg, is_data_2D = labels
        a = abi.StrongLossArgs()
        a.box, a.ld_box, a.stage1_center, a.seg_loss = fptr(box), BOX_OUT_DIMS, fptr(stage1_center), fptr(seg_loss)
        a.y_center, a.y_orient_cls, a.y_orient_reg = fptr(y_center), iptr(y_orient_cls), fptr(y_orient_reg)
        a.y_dims_cls, a.y_dims_reg, a.is_data_2D = iptr(y_dims_cls), fptr(y_dims_reg), iptr(is_data_2D)
        a.wts = abi.StrongWeights(c.STRONG_WEIGHT_CENTER, c.STRONG_WEIGHT_ORIENT_CLS, c.STRONG_WEIGHT_ORIENT_REG,
                                  c.STRONG_WEIGHT_DIMS_CLS, c.STRONG_WEIGHT_DIMS_REG, c.STRONG_WEIGHT_TNET_CENTER,
                                  c.STRONG_WEIGHT_CORNER, c.STRONG_BOX_MULTIPLER, c.STRONG_WEIGHT_CROSS_ENTROPY)
        a.normalize_by_3d_count = int(normalize_by_3d_count)
        a.dbox, a.dstage1, a.terms = fptr(self.dbox), fptr(self.dstage1), fptr(self.terms)
        a.total_losses, a.loss = fptr(self.total_losses), fptr(self.loss)
        a.center, a.reg_dims, a.reg_theta, a.B = fptr(self.center), fptr(self.reg_dims), fptr(self.reg_theta), self.g.B
        a.iou2d, a.iou3d = fptr(self.iou2d), fptr(self.iou3d)
        plan.add('t3d_strong_loss', a)


WEAK_DEFAULTS = dict(       # models/config.py:119-162 of the reference (for configs built without the flag parser)
    WEAK_TRAIN_BOX_W_REPROJECTION=[True, True, True], WEAK_TRAIN_BOX_W_SURFACE=[True, False, True],
    WEAK_REPROJECTION_USE_SOFTMAX_PROJ=False, WEAK_REPROJECTION_SOFTMAX_SCALE=10., WEAK_REPROJECTION_CLIP_LOWERB_LOSS=True,
    WEAK_REPROJECTION_CLIP_PRED_BOX=False, WEAK_REPROJECTION_LOSS_TYPE='huber', WEAK_REPROJECTION_DILATE_FACTOR=1.5,
    WEAK_SURFACE_MARGIN=0., WEAK_SURFACE_LOSS_SCALE_DIMS=0.9)


class WeakLoss:
    """The weak reprojection + surface losses of get_semi_loss_backbone (semisup_v1_sunrgbd.py:270-311) on the box the strong loss
    converted (S_pred_box_reg): `emit` adds them to the loss in the forward plan; `emit_backward` routes their gradients -- into the
    box head and the T-Net centre through the anchor->reg conversion, and into the segmentation logits through the soft mask (a
    second run of the fused seg head, which rewrites conv9's gradient, and of its finalize)."""

    def __init__(self, g):
        self.g = g
        rt, B, N = g.rt, g.B, g.rpf
        self.part, self.dsoft = rt.zeros(B, N // TILE, 8), rt.zeros(B * N)
        self.reproj, self.surface, self.dbox7 = rt.zeros(B), rt.zeros(B), rt.zeros(B, 7)
        self.inactive = rt.zeros(1)

    @staticmethod
    def active(c):
        return getattr(c, 'WEAK_WEIGHT_REPROJECTION', 0) != 0 or getattr(c, 'WEAK_WEIGHT_SURFACE', 0) != 0

    @staticmethod
    def wanted(c):
        """Evaluate the two losses of get_semi_loss_backbone: when a weight is non-zero, or -- as the reference always does for its
        `Weak_Loss/...` summaries (semisup_v1_sunrgbd.py:270-293) -- when `c.WEAK_LOSS_SUMMARIES` asks for their values at zero weight
        (the drivers do; the kernel then adds exactly 0 to the loss and writes zero gradients, and no backward launch is emitted)."""
        return WeakLoss.active(c) or bool(getattr(c, 'WEAK_LOSS_SUMMARIES', False))

    @staticmethod
    def active_final(c):
        return getattr(c, 'WEAK_WEIGHT_REPROJECTION', 0) != 0 or getattr(c, 'WEAK_WEIGHT_INACTIVE_VOLUME', 0) != 0

    def emit(self, plan, loss_op, seg, x, c, final=None):
        """final = None: get_semi_loss_backbone (both losses on the 2-D-label samples, in/out loss_op.loss / total_losses).
        final = (loss buffer, train_classes): get_semi_loss_final (semisup_v1_sunrgbd.py:345-392) -- the reprojection loss of the
        refined box on every sample (or the 2-D-label ones, WEAK_REPROJECTION_ONLY_ON_2D_CLS) and the inactive-volume loss, added
        to the stage-c loss; no surface loss there."""
        g = self.g
        f = lambda k: getattr(c, k, WEAK_DEFAULTS[k])
        a = abi.WeakLossArgs()
        a.center, a.reg_dims, a.reg_theta = fptr(loss_op.center), fptr(loss_op.reg_dims), fptr(loss_op.reg_theta)
        if final is None:
            a.pc, a.ld_pc, a.logits = fptr(x.pc), g.ldpc, fptr(seg.logits)
        a.Rtilt, a.K, a.rot_frust, a.box2D, a.img_dim = fptr(x.Rtilt), fptr(x.K), fptr(x.rot_frust), fptr(x.box2D), fptr(x.img_dim)
        if final is None or getattr(c, 'WEAK_REPROJECTION_ONLY_ON_2D_CLS', False):
            a.is_data_2D = iptr(x.is_data_2D)
        a.w_reproj = float(c.WEAK_WEIGHT_REPROJECTION)
        a.w_surface = float(c.WEAK_WEIGHT_SURFACE) if final is None else 0.0
        a.multiplier = float(c.SEMI_MULTIPLIER_FOR_WEAK_LOSS)
        a.use_softmax_proj, a.softmax_scale = int(bool(f('WEAK_REPROJECTION_USE_SOFTMAX_PROJ'))), float(f('WEAK_REPROJECTION_SOFTMAX_SCALE'))
        a.dilate = float(f('WEAK_REPROJECTION_DILATE_FACTOR'))
        a.clip_lower_b_loss, a.clip_pred_box = int(bool(f('WEAK_REPROJECTION_CLIP_LOWERB_LOSS'))), int(bool(f('WEAK_REPROJECTION_CLIP_PRED_BOX')))
        lt = f('WEAK_REPROJECTION_LOSS_TYPE')
        if lt not in ('huber', 'mse'):
            raise Exception('Not implemented: %s' % lt)                      # weak_losses.py:33-34
        a.loss_mse = int(lt == 'mse')
        a.train_box_reproj = (C.c_int32 * 3)(*[int(bool(v)) for v in f('WEAK_TRAIN_BOX_W_REPROJECTION')])
        a.train_box_surface = (C.c_int32 * 3)(*[int(bool(v)) for v in f('WEAK_TRAIN_BOX_W_SURFACE')])
        a.surface_margin, a.surface_scale_dims = float(f('WEAK_SURFACE_MARGIN')), float(f('WEAK_SURFACE_LOSS_SCALE_DIMS'))
        a.surf_part, a.dsoft = fptr(self.part), fptr(self.dsoft)
        a.reproj, a.surface, a.dbox7 = fptr(self.reproj), fptr(self.surface), fptr(self.dbox7)
        a.B, a.N = g.B, g.rpf
        if final is None:
            a.total_losses, a.loss = fptr(loss_op.total_losses), fptr(loss_op.loss)
        else:
            loss_buf, train_classes = final
            a.loss = fptr(loss_buf)
            a.w_inactive = float(getattr(c, 'WEAK_WEIGHT_INACTIVE_VOLUME', 0))
            if a.w_inactive != 0:
                margins = list(c.WEAK_INACTIVE_VOL_LOSS_MARGINS)
                assert len(margins) == 10                                      # semisup_v1_sunrgbd.py:350
                a.one_hot, a.inactive = fptr(x.one_hot_vec), fptr(self.inactive)
                a.inactive_margins = (C.c_float * 10)(*[float(v) for v in margins])
                a.inactive_train = (C.c_int32 * 10)(*[int(bool(v)) for v in train_classes])
        self.surface_on = a.w_surface != 0
        plan.add('t3d_weak_loss', a)

    def emit_backward(self, plan, loss_op, seg, box_head_out):
        a = abi.AnchorRegBwdArgs()
        a.box, a.ld_box, a.dbox7, a.dbox, a.dstage1, a.B = fptr(box_head_out), BOX_OUT_DIMS, fptr(self.dbox7), fptr(loss_op.dbox), \
            fptr(loss_op.dstage1), self.g.B
        plan.add('t3d_anchor_reg_bwd', a)
        if self.surface_on:
            seg.emit_head_again(plan, self.dsoft)


def emit_box_head_iou(g, plan, box, stage1_center, labels, iou2d, iou3d):
    """compute_box3d_iou on raw box heads (roi_seg_box3d_dataset.py:103-140) as its own launch: the `W_` summary of stage c."""
    y_center, y_orient_cls, y_orient_reg, y_dims_cls, y_dims_reg, _ = labels
    a = abi.BoxHeadIouArgs(fptr(box), BOX_OUT_DIMS, fptr(stage1_center), fptr(y_center), iptr(y_orient_cls), fptr(y_orient_reg),
                           iptr(y_dims_cls), fptr(y_dims_reg), fptr(iou2d), fptr(iou3d), g.B)
    plan.add('t3d_box_head_iou', a)


class ExtraFeats:
    """[one_hot | norm_box2D] per frustum, the extra FC-input block of the T-Net and the box net under
    USE_NORMALIZED_BOX2D_AS_FEATS (semisup_v1_sunrgbd.py:97,145; train_semisup.py:240 norm_box2D =
    tf_util.tf_normalize_2D_bboxes(box2D_pl, img_dim_pl)); one tiny launch at the head of the forward (t3d_box2d_feats)."""

    def __init__(self, g, n_oh):
        self.g, self.n_oh, self.ld = g, n_oh, n_oh + 4
        self.buf = g.rt.zeros(g.B, self.ld)

    def emit(self, plan, x):
        a = abi.Box2dFeatsArgs(fptr(x.one_hot_vec) if self.n_oh else None, self.n_oh, fptr(x.box2D), fptr(x.img_dim), fptr(self.buf),
                               self.g.B)
        plan.add('t3d_box2d_feats', a)
        return self.buf, self.ld


class Inputs:
    """Device-resident feed buffers in the reference's batch layout (roi_semi_dataset.py:531-534;
    semisup_v1_sunrgbd.placeholder_inputs 37-67).  `load(batch)` copies a NumPy batch in (H2D)."""

    FIELDS = [('pc', torch.float32, lambda B, N, C: (B * N, C)), ('one_hot_vec', torch.float32, lambda B, N, C: (B, NUM_CLASS)),
              ('y_seg', torch.int32, lambda B, N, C: (B * N,)), ('y_center', torch.float32, lambda B, N, C: (B, 3)),
              ('y_orient_cls', torch.int32, lambda B, N, C: (B,)), ('y_orient_reg', torch.float32, lambda B, N, C: (B,)),
              ('y_dims_cls', torch.int32, lambda B, N, C: (B,)), ('y_dims_reg', torch.float32, lambda B, N, C: (B, 3)),
              ('is_data_2D', torch.int32, lambda B, N, C: (B,)),
              ('y_box_iou', torch.float32, lambda B, N, C: (B,)), ('y_center_delta', torch.float32, lambda B, N, C: (B, 3)),
              ('y_dims_delta', torch.float32, lambda B, N, C: (B, 3)), ('y_orient_delta', torch.float32, lambda B, N, C: (B,)),
              # camera side of the weak losses (semisup_v1_sunrgbd.py:52-62)
              ('Rtilt', torch.float32, lambda B, N, C: (B, 9)), ('K', torch.float32, lambda B, N, C: (B, 9)),
              ('rot_frust', torch.float32, lambda B, N, C: (B,)), ('box2D', torch.float32, lambda B, N, C: (B, 4)),
              ('img_dim', torch.float32, lambda B, N, C: (B, 2))]

    def __init__(self, g):
        self.g = g
        for name, dt, shp in self.FIELDS:
            setattr(self, name, g.rt.zeros(*shp(g.B, g.rpf, g.ldpc if name == 'pc' else g.C), dtype=dt))

    def load(self, batch):
        for name, dt, _ in self.FIELDS:
            if name in batch:
                t = getattr(self, name)
                src = torch.as_tensor(np.ascontiguousarray(batch[name])).to(dt)
                if name == 'pc':                      # rows padded to g.ldpc floats; the padding stays zero
                    t[:, :self.g.C].copy_(src.reshape(-1, self.g.C))
                else:
                    t.copy_(src.reshape(t.shape))
        for scope, m in batch.get('dropout_masks', {}).items():
            if scope in self.g.dropout_masks:
                t = self.g.dropout_masks[scope][0]
                t.copy_(torch.as_tensor(np.ascontiguousarray(m)).to(torch.float32).reshape(t.shape))


class ModelAssembly:
    """Orchestrates whichever sub-networks have been built on a graph (seg -> T-Net -> box -> loss), forward and
    backward.  The reference wires these in get_semi_model_backbone (semisup_v1_sunrgbd.py:81-130) and lets TF
    autodiff derive the backward; here both directions are explicit launch schedules."""

    def __init__(self, g, c, inputs=None, use_one_hot=False):
        self.g, self.c, self.use_one_hot = g, c, use_one_hot
        self.inputs = inputs or Inputs(g)
        self.seg = self.tnet = self.box = self.loss_op = None
        self.weak = None
        self.box2d = bool(getattr(c, 'USE_NORMALIZED_BOX2D_AS_FEATS', False))
        self.extra = ExtraFeats(g, NUM_CLASS if use_one_hot else 0) if self.box2d else None

    def emit_forward(self, plan, is_training, with_loss):
        g, x, c = self.g, self.inputs, self.c
        g.emit_cast_weights(plan)
        labels = x.y_seg if with_loss else None
        train_seg = with_loss and is_training
        oh, ld_oh = x.one_hot_vec, NUM_CLASS
        self.seg.fwd(plan, x.pc, oh, labels, x.is_data_2D, is_training, train_seg, ce_weight=c.STRONG_WEIGHT_CROSS_ENTROPY)
        if self.tnet is None:
            return
        if self.extra is not None:       # the seg net never sees norm_box2D (semisup_models.py:69)
            oh, ld_oh = self.extra.emit(plan, x)
        if train_seg:
            plan.mark('T_begin')        # from here to the seg net's backward: the chain the seg backward does not depend on (schedule.py)
        s1 = self.tnet.fwd(plan, x.pc, self.seg.mask, self.seg.mask_xyz_mean, oh, is_training, ld_oh=ld_oh)
        if self.box is None:
            return
        box = self.box.fwd(plan, x.pc, self.seg.mask, s1, oh, is_training, ld_oh=ld_oh)
        if self.loss_op is None:
            self.loss_op = StrongLoss(g)
        if with_loss:
            lab = (x.y_center, x.y_orient_cls, x.y_orient_reg, x.y_dims_cls, x.y_dims_reg, x.is_data_2D)
            self.loss_op.emit(plan, box, s1, self.seg.seg_loss, lab, c)
            if WeakLoss.wanted(c):       # reprojection / surface losses with non-zero weight (the reference's DEFAULT flags), or for their summaries
                if self.weak is None:
                    self.weak = WeakLoss(g)
                self.weak.emit(plan, self.loss_op, self.seg, x, c)
                self.weak_trains = WeakLoss.active(c)

    def emit_backward(self, plan):
        from .engine import FC_SIDE
        if self.weak is not None and getattr(self, 'weak_trains', True):        # weak-loss gradients join dbox / dstage1 and conv9's gradient before anything reads them
            self.weak.emit_backward(plan, self.loss_op, self.seg, self.box.box_params)
        if FC_SIDE and self.seg.train_seg:
            plan.two_streams = True
            with plan.side():
                dfb = self.box.bwd_fc(plan, self.loss_op.dbox)
            plan.flush()                                    # fork: box FC chain || seg conv9..conv7
            self.seg.bwd(plan, part=0)
            plan.join()
            ds1 = self.box.bwd_convs(plan, dfb, 512, self.loss_op.dstage1)
            with plan.side():
                dft = self.tnet.bwd_fc(plan, ds1)
            plan.flush()                                    # fork: T-Net FC chain || seg conv6, conv5
            self.seg.bwd(plan, part=1)
            plan.join()
            self.tnet.bwd_convs(plan, dft)
            self.seg.bwd(plan, part=2)
            self.g.emit_reduce_slabs(plan)
            return
        i0 = len(plan.calls)
        head_reduce = None
        if getattr(self.g, 'split_opt', False):
            # pipelined step: the slabs recorded by the FORWARD (conv10's weight-gradient partials, written by the seg head) belong to
            # the seg net's chain: their reduction is recorded now (before the T-Net / box entries join the table) and placed behind
            # `S_begin` below
            from .engine import Plan as _Plan
            head_reduce = _Plan(self.g.rt)
            self.g.emit_reduce_slabs(head_reduce)
        ds1 = self.box.bwd(plan, self.loss_op.dbox, self.loss_op.dstage1)
        plan.flush()                     # box-net weight gradients run beside the T-Net / seg-net dgrad chain
        self.tnet.bwd(plan, ds1)
        plan.flush()
        i1 = len(plan.calls)
        # data parallel: the box / T-Net gradients are complete here and nothing the seg net computes touches
        # them (semisup_models.py:150-151), so their all-reduce runs beside the seg net's backward; the seg net's own gradients
        # go in two buckets, conv10..conv6 (2.9 MB, ready after conv6) and conv5..conv1 (0.6 MB, the exposed tail)
        g, sp = self.g, self.seg.scope + '/'
        g.declare_bucket(plan, [self.tnet.scope + '/', self.box.scope + '/'])
        # single replica, no weak-loss gradient into the seg net: the seg backward is independent of everything since `T_begin`
        # (semisup_models.py:150-151) -- the step scheduler interleaves the two chains (schedule.py); T3D_OVERLAP=0: the pairing below
        overlap = OVERLAP and not g.dp_buckets and self.seg.train_seg and (self.weak is None or not getattr(self, 'weak_trains', True))
        if overlap and getattr(g, 'split_opt', False):
            g.emit_reduce_slabs(plan)      # software-pipelined step (step.PipelinedStep): the T chain ends with ITS slab reduction + Adam
        if overlap:
            plan.mark('S_begin')
        if head_reduce is not None:
            assert overlap, 'split_opt needs the two-chain backward'
            for cl, ln in zip(head_reduce.calls, head_reduce.lanes):
                if not cl[0].startswith('__'):
                    plan.calls.append(cl)
                    plan.lanes.append(ln)
        self.seg.bwd(plan, part=0)
        self.seg.bwd(plan, part=1)
        g.declare_bucket(plan, [sp + 'conv%d/' % i for i in (6, 7, 8, 9, 10)])
        self.seg.bwd(plan, part=2)
        if overlap:
            plan.mark('S_end')
        elif PAIR_SMALL and not g.dp_buckets and self.seg.train_seg:
            # single replica: the box / T-Net backward [i0, i1) and the seg-net backward [i1, here) are independent
            # (semisup_models.py:150-151): their small launches pair up (data parallel keeps the chains apart -- the first bucket's
            # all-reduce is to start as early as possible)
            self.n_pairs = pair_small_launches(plan, i0, i1, len(plan.calls))
        if g.dp_buckets:
            g.declare_bucket(plan, [sp + 'conv%d/' % i for i in (1, 2, 3, 4, 5)])
        else:
            g.emit_reduce_slabs(plan)

    def end_points(self):
        g = self.g
        B, N = g.B, g.rpf
        ep = {'logits': self.seg.logits.view(B, N, 2), 'mask': self.seg.mask.view(B, N),
              'mask_xyz_mean': self.seg.mask_xyz_mean, 'seg_global_feat': self.seg.L5.pooled}
        if self.tnet is not None:
            ep.update({'stage1_center': self.tnet.stage1_center, 'tnet_feats': self.tnet.T3.pooled})
        if self.box is not None:
            ep.update({'box_params': self.box.box_params, 'feats_lv1': self.box.feats_lv1,
                       'feats_lv2': self.box.feats_lv2, 'feats_lv3': self.box.feats_lv3})
        if self.loss_op is not None:
            ep.update({'center': self.loss_op.center, 'loss_terms': self.loss_op.terms,
                       'total_losses': self.loss_op.total_losses, 'loss': self.loss_op.loss,
                       'S_dims': self.loss_op.reg_dims, 'S_theta': self.loss_op.reg_theta,
                       'iou2ds': self.loss_op.iou2d, 'iou3ds': self.loss_op.iou3d})
        if self.weak is not None:
            ep.update({'reprojection_loss': self.weak.reproj, 'surface_loss': self.weak.surface})
        return ep


class SemiModelA(ModelAssembly):
    """SEMI_MODEL A (get_semi_model_backbone + get_semi_loss_backbone, semisup_v1_sunrgbd.py:81-130,
    256-321): seg PointNet -> masked centroid -> T-Net -> box PointNet -> strong loss, forward and
    backward, as one static launch schedule."""

    def __init__(self, g, c, use_one_hot=False, scope_prefix='', inputs=None):
        ModelAssembly.__init__(self, g, c, inputs, use_one_hot)
        self.seg = InstSegNet(g, scope_prefix + 'inst_seg', use_one_hot)
        self.tnet = TNet(g, scope_prefix + 'tnet', use_one_hot, box2d=self.box2d)
        self.box = BoxEstNet(g, scope_prefix + 'box_est', use_one_hot, box2d=self.box2d)
        self.loss_op = StrongLoss(g)


class BoxPCNet:
    """Box-PC Fit network, representation 'A' (combined_box_pc_mask_features_model, semisup_models.py:326-398, under
    the literal scope `box_pc_mask_model`; boxpc_sunrgbd.get_model 56-100)."""

    def __init__(self, g, scope_prefix='', use_one_hot=False):
        self.g = g
        s = scope_prefix + 'box_pc_mask_model/'
        self.scope = s
        C_ = g.C
        self.ld_rep = (C_ + 6 + 3) // 4 * 4
        self.rep = g.rt.zeros(g.M, self.ld_rep)
        self.box7 = g.rt.zeros(g.B, 7)
        oh = NUM_CLASS if use_one_hot else 0
        self.oh = oh
        self.P1 = PointLayer(g, s + 'conv-reg1', C_ + 6, 128, kernel_1xD=True)
        self.P2 = PointLayer(g, s + 'conv-reg2', 128, 128)
        self.P3 = PointLayer(g, s + 'conv-reg3', 128, 256)
        self.P4 = PointLayer(g, s + 'conv-reg4', 256, 512, pool=True)
        self.F1 = FcLayer(g, s + 'fc1', 512, 512, K2=oh, keep_prob=0.7, drop_scope=s + 'dp1')
        self.F2 = FcLayer(g, s + 'fc2', 512, 256, keep_prob=0.7, drop_scope=s + 'dp2')
        self.F3 = FcLayer(g, s + 'fc3', 256, 9, bn=False, act=None)

    def fwd(self, plan, pc, center, dims, theta, one_hot, is_training, y_dims_cls=None, y_orient_cls=None, rowmask=None):
        """`rowmask` ([M] 0/1): the net sees pc * mask (--mask_pc_for_boxpc, test_semisup.py:103-105)."""
        g = self.g
        a = abi.BoxPcRepArgs(fptr(pc), g.ldpc, g.C, fptr(center), fptr(dims), fptr(theta), iptr(y_dims_cls), iptr(y_orient_cls),
                             fptr(self.rep), self.ld_rep, fptr(self.box7), g.M, g.rpf, fptr(rowmask))
        plan.add('t3d_boxpc_rep', a)
        x = ActSpec(self.rep, self.ld_rep, g.C + 6)
        x = self.P1.fwd(plan, x, is_training)
        x = self.P2.fwd(plan, x, is_training)
        x = self.P3.fwd(plan, x, is_training)
        self.P4.fwd(plan, x, is_training)
        f = self.F1.fwd(plan, self.P4.pooled, 512, is_training, in2=one_hot if self.oh else None, ld_in2=NUM_CLASS)
        f = self.F2.fwd(plan, f, 512, is_training)
        self.out = self.F3.fwd(plan, f, 256, is_training)
        return self.out

    def bwd(self, plan, dout, param_grads=True):
        self.F3.bwd(plan, dout=dout, ld_dout=9, param_grads=param_grads)
        self.F2.bwd(plan, nxt=self.F3, param_grads=param_grads)
        self.F1.bwd(plan, nxt=self.F2, param_grads=param_grads)
        dfeat = self.F1.dinput(plan, K=512, bn_bwd_of=self.P4, param_grads=param_grads)
        self.P4.bn_bwd(plan, dpool_in=dfeat, ld_dpool_in=512, param_grads=param_grads)
        if param_grads:
            self.P4.bwd_pair(plan)
        else:
            self.P4.dgrad(plan)
        for lay in (self.P3, self.P2):
            lay.bn_bwd(plan, param_grads=param_grads)
            if param_grads:
                lay.bwd_pair(plan)
            else:
                lay.dgrad(plan)
        self.P1.bn_bwd(plan, param_grads=param_grads)
        if param_grads:
            self.P1.wgrad(plan)


class BoxPCLoss:
    def __init__(self, g):
        self.g = g
        rt, B = g.rt, g.B
        self.dout, self.terms, self.loss = rt.zeros(B, 9), rt.zeros(B, 4), rt.zeros(1)

    def emit(self, plan, out, x, c):
        a = abi.BoxPcLossArgs(fptr(out), fptr(x.y_box_iou), fptr(x.y_center_delta), fptr(x.y_dims_delta), fptr(x.y_orient_delta),
                              c.BOXPC_FIT_BOUNDS[0], c.BOXPC_WEIGHT_CLS, c.BOXPC_WEIGHT_DELTA, c.BOXPC_WEIGHT_DELTA_CENTER_PERCENT,
                              c.BOXPC_WEIGHT_DELTA_SIZE_PERCENT, c.BOXPC_WEIGHT_DELTA_ANGLE_PERCENT,
                              int(c.BOXPC_WEIGH_DELTA_LOSS_BY_CLS_CONF), int(c.BOXPC_WEIGH_DELTA_LOSS_BY_CLS_GT),
                              fptr(self.dout), fptr(self.terms), fptr(self.loss), self.g.B,
                              int(bool(c.BOXPC_WEIGH_DELTA_PRED_BY_CLS_CONF)), int(not c.BOXPC_STOP_GRAD_OF_CLS_VIA_DELTA),
                              int(c.BOXPC_DELTA_LOSS_TYPE == 'mse'))
        plan.add('t3d_boxpc_loss', a)


class BoxPCModel:
    """Stage-b training graph (train_boxpc.py:219-261): GT box (label form) + point cloud -> Box-PC net -> loss."""

    def __init__(self, g, c, use_one_hot=False, inputs=None):
        assert c.BOX_PC_MASK_REPRESENTATION in ('A', ''), 'representation B is in no published recipe (out of scope)'
        assert c.BOXPC_DELTA_LOSS_TYPE in ('huber', 'mse'), c.BOXPC_DELTA_LOSS_TYPE
        assert not (c.BOXPC_WEIGH_DELTA_LOSS_BY_CLS_CONF and c.BOXPC_WEIGH_DELTA_LOSS_BY_CLS_GT)       # boxpc_sunrgbd.py:166
        self.g, self.c = g, c
        self.inputs = inputs or Inputs(g)
        self.net = BoxPCNet(g, '', use_one_hot)
        self.loss_op = BoxPCLoss(g)

    def emit_forward(self, plan, is_training, with_loss):
        x = self.inputs
        self.g.emit_cast_weights(plan)
        out = self.net.fwd(plan, x.pc, x.y_center, x.y_dims_reg, x.y_orient_reg, x.one_hot_vec, is_training,
                           y_dims_cls=x.y_dims_cls, y_orient_cls=x.y_orient_cls)
        if with_loss:
            self.loss_op.emit(plan, out, x, self.c)

    def emit_backward(self, plan):
        self.net.bwd(plan, self.loss_op.dout)
        self.g.emit_reduce_slabs(plan)

    def end_points(self):
        n = self.net
        return {'boxpc_out': n.out, 'box_pc_rep': n.rep, 'loss': self.loss_op.loss, 'terms': self.loss_op.terms,
                'boxpc_fit_logits': n.out[:, 7:9], 'boxpc_delta_center': n.out[:, 0:3], 'boxpc_delta_size': n.out[:, 3:6],
                'boxpc_delta_angle': n.out[:, 6], 'feats_lv1': n.P4.pooled}


class SemiModelF:
    """SEMI_MODEL F, stage c (get_semi_model_final semisup_v1_sunrgbd.py:132-230 + the Box-PC glue of
    train_semisup_adv.py:331-411 + get_semi_loss_final 323-421): class-agnostic seg / T-Net / box nets, the
    class-dependent box_refine MLP, the FROZEN Box-PC Fit net applied to the refined box, strong + intraclass + fit loss.

    Backward follows the var_list of train_semisup_adv.py:415-422: nothing for the seg net, box_est/fc1-3 receive no
    gradient (the loss never reads the class-agnostic heads); the Box-PC net back-propagates data gradients only."""

    def __init__(self, g, c, use_one_hot=True, train_classes=None, inputs=None, oracle_mask=False, mask_pc_for_boxpc=False):
        """oracle_mask: the seg logits are replaced by stack([1 - y_seg, y_seg]) (semisup_v1_sunrgbd.py:161-162; test_semisup.py:75);
        mask_pc_for_boxpc: the Box-PC net of the inference graph sees pc * mask (test_semisup.py:103-105)."""
        self.g, self.c = g, c
        self.inputs = inputs or Inputs(g)
        self.oracle_mask, self.mask_pc_for_boxpc = bool(oracle_mask), bool(mask_pc_for_boxpc)
        p = 'class_agnostic/'
        self.seg = InstSegNet(g, p + 'inst_seg', False)
        self.box2d = bool(getattr(c, 'USE_NORMALIZED_BOX2D_AS_FEATS', False))      # semisup_v1_sunrgbd.py:145,168,176
        self.extra = ExtraFeats(g, 0) if self.box2d else None
        self.tnet = TNet(g, p + 'tnet', False, box2d=self.box2d)
        self.box = BoxEstNet(g, p + 'box_est', False, box2d=self.box2d)
        q = 'class_dependent/box_refine/'
        oh = NUM_CLASS if use_one_hot else 0
        self.oh = oh
        act = 'leaky_relu' if c.SEMI_ADV_LEAKY_RELU else 'relu'
        last = 'tanh' if c.SEMI_ADV_TANH_FOR_LAST_LAYER_OF_G else act
        dp = c.SEMI_ADV_DROPOUTS_FOR_G
        self.R0 = FcLayer(g, q + 'fc0', 512, 512, act=act, K2=oh, keep_prob=dp, drop_scope=q + 'dp0')
        self.R1 = FcLayer(g, q + 'fc1', 512, 256, act=last, keep_prob=dp, drop_scope=q + 'dp1')
        self.R2 = FcLayer(g, q + 'fc2', 256, BOX_OUT_DIMS, bn=False, act=None)
        self.loss_op = StrongLoss(g)
        self.boxpc = BoxPCNet(g, 'D_boxpc_branch/', False)
        # SEMI_REFINE_USING_BOXPC_DELTA_NUM > 1 in the TRAINING graph (train_semisup_adv.py:362-386): one evaluation of the frozen
        # Box-PC net per refinement step -- the same variables (reuse=True), own activations, because with
        # SEMI_BOXPC_MIN_FIT_LOSS_AFT_REFINE the fit loss reads the LAST evaluation and its gradient runs back through every step
        self.refine_train = max(1, int(c.SEMI_REFINE_USING_BOXPC_DELTA_NUM))
        self.boxpc_nets = [self.boxpc] + [BoxPCNet(g, 'D_boxpc_branch/', False) for _ in range(self.refine_train - 1)]
        for i, net in enumerate(self.boxpc_nets[1:], 1):      # names under which tests read the decisions of the i-th further evaluation
            for lay in (net.P1, net.P2, net.P3, net.P4, net.F1, net.F2, net.F3):
                lay.decision_scope = lay.scope.replace('D_boxpc_branch/', 'D_boxpc_branch@%d/' % i, 1)
        self.loss_eval = self.refine_train - 1 if c.SEMI_BOXPC_MIN_FIT_LOSS_AFT_REFINE else 0
        self.refine_w = int(bool(c.SEMI_WEIGH_BOXPC_DELTA_DURING_TEST)) + int(bool(c.BOXPC_WEIGH_DELTA_PRED_BY_CLS_CONF))
        self.train_classes = list(train_classes) if train_classes is not None else [True] * NUM_CLASS
        rt, B = g.rt, g.B
        self.d_dims, self.dout9, self.fit_prob = rt.zeros(B, 3), rt.zeros(B, 9), rt.zeros(B)
        self.terms, self.loss = rt.zeros(2), rt.zeros(1)
        self.W_iou2d, self.W_iou3d = rt.zeros(B), rt.zeros(B)      # get_iou_summary(W_pred_box, ..., 'W_') (semisup_v1_sunrgbd.py:414)
        self.drep, self.dbox7 = rt.zeros(g.M, 8), rt.zeros(B, 7)
        self.carry, self.dout9_chain = rt.zeros(B, 7), rt.zeros(B, 9)
        self.weak = None
        # inference graph (test_semisup.py:95-149): iterated Box-PC refinement of the F_ box
        self.refine_num = None
        self.cur_center, self.cur_dims, self.cur_theta, self.total_delta = rt.zeros(B, 3), rt.zeros(B, 3), rt.zeros(B), rt.zeros(B, 7)

    def emit_forward(self, plan, is_training, with_loss, train=False):
        """`train`: a backward plan follows (api.Session / step.build_training_step say so).  Then the class-agnostic box heads
        fc1-fc3 and the W_ IoU summary -- they feed end points only, nothing of the loss or the backward reads them -- are emitted
        at the END of the backward plan as a chain of their own (`S_begin` / `S_end`), and everything from the refinement branch on
        is the other chain (`T_begin`): schedule.overlap_chains lets the three FC launches ride in the Box-PC net's GEMM launches."""
        if not is_training and self.refine_num is not None:
            return self.emit_forward_inference(plan, self.refine_num)
        g, x, c = self.g, self.inputs, self.c
        g.emit_cast_weights(plan)
        self.seg.fwd(plan, x.pc, x.one_hot_vec, x.y_seg if with_loss else None, x.is_data_2D, is_training, False,
                     ce_weight=c.STRONG_WEIGHT_CROSS_ENTROPY, oracle_mask=x.y_seg if self.oracle_mask else None)
        ex, ld_ex = self.extra.emit(plan, x) if self.extra is not None else (x.one_hot_vec, NUM_CLASS)
        s1 = self.tnet.fwd(plan, x.pc, self.seg.mask, self.seg.mask_xyz_mean, ex, is_training, ld_oh=ld_ex)
        self.box.fwd_convs(plan, x.pc, self.seg.mask, s1, is_training)
        self._deferred = None
        if train and OVERLAP and is_training and not g.dp_buckets:
            self._deferred = (ex, ld_ex, is_training, with_loss, s1)
            plan.mark('T_begin')
        else:
            self.box.fwd_heads(plan, ex, is_training, ld_oh=ld_ex)
        f = self.R0.fwd(plan, self.box.feats_lv1, 512, is_training, in2=x.one_hot_vec if self.oh else None, ld_in2=NUM_CLASS)
        f = self.R1.fwd(plan, f, 512, is_training)
        self.F_out = self.R2.fwd(plan, f, 256, is_training)
        lab = (x.y_center, x.y_orient_cls, x.y_orient_reg, x.y_dims_cls, x.y_dims_reg, x.is_data_2D)
        self.loss_op.emit(plan, self.F_out, s1, self.seg.seg_loss, lab, c, normalize_by_3d_count=True)
        if with_loss and self._deferred is None:
            emit_box_head_iou(g, plan, self.box.box_params, s1, lab, self.W_iou2d, self.W_iou3d)
        # frozen Box-PC net on F_pred_box_reg (is_training_D = False: eval-mode batch-norm, no dropout)
        lo = self.loss_op
        src = (lo.center, lo.reg_dims, lo.reg_theta)
        cur = (self.cur_center, self.cur_dims, self.cur_theta)
        self.outs9 = []
        for i, net in enumerate(self.boxpc_nets):
            self.outs9.append(net.fwd(plan, x.pc, src[0], src[1], src[2], x.one_hot_vec, False))
            if self.refine_train > 1:      # box <- box - w * delta(box, pc); the totals give the F2_ heads (train_semisup_adv.py:376-399)
                r = abi.BoxRefineStepArgs(fptr(self.outs9[i]), fptr(src[0]), fptr(src[1]), fptr(src[2]), fptr(cur[0]), fptr(cur[1]),
                                          fptr(cur[2]), fptr(self.total_delta), None, self.refine_w, int(i == 0), g.B)
                plan.add('t3d_box_refine_step', r)
                src = cur
        out9 = self.outs9[self.loss_eval]
        a = abi.SemiFinalLossArgs()
        a.strong_loss, a.reg_dims, a.one_hot, a.is_data_2D, a.out9 = fptr(lo.loss), fptr(lo.reg_dims), fptr(x.one_hot_vec), \
            iptr(x.is_data_2D), fptr(out9)
        for i in range(NUM_CLASS):
            a.train_classes[i] = int(bool(self.train_classes[i]))
        a.w_weak = float(c.SEMI_MULTIPLIER_FOR_WEAK_LOSS * c.WEAK_WEIGHT_INTRACLASSVAR)
        a.w_fit, a.fit_only_2d = float(c.SEMI_WEIGHT_BOXPC_FIT_LOSS), int(bool(c.SEMI_BOXPC_FIT_ONLY_ON_2D_CLS))
        a.d_dims, a.dout9, a.fit_prob, a.terms, a.loss, a.B = fptr(self.d_dims), fptr(self.dout9), fptr(self.fit_prob), \
            fptr(self.terms), fptr(self.loss), g.B
        plan.add('t3d_semi_final_loss', a)
        if with_loss and WeakLoss.active_final(c):      # reprojection of the refined box / inactive volume (semisup_v1_sunrgbd.py:345-392)
            if self.weak is None:
                self.weak = WeakLoss(g)
            self.weak.emit(plan, lo, self.seg, x, c, final=(self.loss, getattr(self, 'inactive_train_classes', self.train_classes)))

    def emit_forward_inference(self, plan, refine_num):
        """test_semisup.py:61-149: every net in inference mode; the F_ box in regression form is refined `refine_num`
        times, box <- box - w * delta(box, pc), with the deltas accumulated in total_delta (the F2_ heads are the F_ heads
        minus the totals).  The strong-loss kernel runs for its anchor->reg outputs only (labels are whatever the label
        buffers hold; its loss values are not part of this graph)."""
        g, x, c = self.g, self.inputs, self.c
        g.emit_cast_weights(plan)
        self.seg.fwd(plan, x.pc, x.one_hot_vec, None, x.is_data_2D, False, False, ce_weight=c.STRONG_WEIGHT_CROSS_ENTROPY,
                     oracle_mask=x.y_seg if self.oracle_mask else None)
        ex, ld_ex = self.extra.emit(plan, x) if self.extra is not None else (x.one_hot_vec, NUM_CLASS)
        s1 = self.tnet.fwd(plan, x.pc, self.seg.mask, self.seg.mask_xyz_mean, ex, False, ld_oh=ld_ex)
        self.box.fwd(plan, x.pc, self.seg.mask, s1, ex, False, ld_oh=ld_ex)
        f = self.R0.fwd(plan, self.box.feats_lv1, 512, False, in2=x.one_hot_vec if self.oh else None, ld_in2=NUM_CLASS)
        f = self.R1.fwd(plan, f, 512, False)
        self.F_out = self.R2.fwd(plan, f, 256, False)
        lab = (x.y_center, x.y_orient_cls, x.y_orient_reg, x.y_dims_cls, x.y_dims_reg, x.is_data_2D)
        lo = self.loss_op
        lo.emit(plan, self.F_out, s1, self.seg.seg_loss, lab, c, normalize_by_3d_count=True)
        src = (lo.center, lo.reg_dims, lo.reg_theta)
        cur = (self.cur_center, self.cur_dims, self.cur_theta)
        for i in range(int(refine_num)):
            # (mask = argmax(logits) = the head's hard mask: both are 0 on a tie)
            out9 = self.boxpc.fwd(plan, x.pc, src[0], src[1], src[2], x.one_hot_vec, False,
                                  rowmask=self.seg.mask if self.mask_pc_for_boxpc else None)
            a = abi.BoxRefineStepArgs(fptr(out9), fptr(src[0]), fptr(src[1]), fptr(src[2]), fptr(cur[0]), fptr(cur[1]), fptr(cur[2]),
                                      fptr(self.total_delta), fptr(self.fit_prob),
                                      int(bool(c.SEMI_WEIGH_BOXPC_DELTA_DURING_TEST)) + int(bool(c.BOXPC_WEIGH_DELTA_PRED_BY_CLS_CONF)),
                                      int(i == 0), g.B)
            plan.add('t3d_box_refine_step', a)
            src = cur

    def emit_backward(self, plan):
        g, x = self.g, self.inputs
        dout, carry = self.dout9, None
        for i in range(self.loss_eval, -1, -1):       # the evaluation the fit loss read, then back through the refinement steps before it
            bp = self.boxpc_nets[i]
            bp.bwd(plan, dout, param_grads=False)
            n = abi.DgradNarrowArgs(bp.P1.dy_struct(), fptr(bp.P1.w), g.C, 6, fptr(self.drep), 8, g.M, 128)
            plan.add('t3d_pointmlp_dgrad_narrow', n)
            r = abi.BoxPcRepBwdArgs(fptr(x.pc), g.ldpc, fptr(bp.box7), fptr(self.drep), 8, 0, fptr(self.dbox7), g.B, g.rpf)
            plan.add('t3d_boxpc_rep_bwd', r)
            if i > 0:        # box_i = box_{i-1} - w * delta_{i-1}: gradient into evaluation i-1's deltas, and past it (carry)
                q = abi.BoxRefineStepBwdArgs(fptr(self.outs9[i - 1]), fptr(self.dbox7), fptr(carry), fptr(self.carry),
                                             fptr(self.dout9_chain), self.refine_w, int(not self.c.BOXPC_STOP_GRAD_OF_CLS_VIA_DELTA), g.B)
                plan.add('t3d_box_refine_step_bwd', q)
                dout, carry = self.dout9_chain, self.carry
            elif carry is not None:                      # total gradient w.r.t. the unrefined F_ box
                q = abi.BoxRefineStepBwdArgs(None, fptr(self.dbox7), fptr(carry), fptr(self.dbox7), None, 0, 0, g.B)
                plan.add('t3d_box_refine_step_bwd', q)
        lo = self.loss_op
        q = abi.AnchorRegBwdArgs(fptr(self.F_out), BOX_OUT_DIMS, fptr(self.dbox7), fptr(self.d_dims), fptr(lo.dbox),
                                 fptr(lo.dstage1), g.B)
        plan.add('t3d_anchor_reg_bwd', q)
        if self.weak is not None:
            q2 = abi.AnchorRegBwdArgs(fptr(self.F_out), BOX_OUT_DIMS, fptr(self.weak.dbox7), fptr(None), fptr(lo.dbox),
                                      fptr(lo.dstage1), g.B)
            plan.add('t3d_anchor_reg_bwd', q2)
        self.R2.bwd(plan, dout=lo.dbox, ld_dout=BOX_OUT_DIMS)
        self.R1.bwd(plan, nxt=self.R2)
        self.R0.bwd(plan, nxt=self.R1)
        dfeat = self.R0.dinput(plan, K=512, bn_bwd_of=self.box.B4)
        ds1 = self.box.bwd_convs(plan, dfeat, 512, lo.dstage1)
        self.tnet.bwd(plan, ds1)
        if getattr(self, '_deferred', None) is not None:      # the W_ heads, as the second chain (emit_forward)
            ex, ld_ex, is_training, with_loss, s1 = self._deferred
            plan.mark('S_begin')
            self.box.fwd_heads(plan, ex, is_training, ld_oh=ld_ex)
            if with_loss:
                lab = (x.y_center, x.y_orient_cls, x.y_orient_reg, x.y_dims_cls, x.y_dims_reg, x.is_data_2D)
                emit_box_head_iou(g, plan, self.box.box_params, s1, lab, self.W_iou2d, self.W_iou3d)
            plan.mark('S_end')
        g.emit_reduce_slabs(plan)

    VAR_LIST = ('class_dependent', 'class_agnostic/tnet', 'class_agnostic/box')

    def end_points(self):
        g = self.g
        B, N = g.B, g.rpf
        lo = self.loss_op
        return {'logits': self.seg.logits.view(B, N, 2), 'stage1_center': self.tnet.stage1_center,
                'feats_lv1': self.box.feats_lv1, 'box_params': self.box.box_params, 'F_box_params': self.F_out,
                'F_center': lo.center, 'F_dims': lo.reg_dims, 'F_theta': lo.reg_theta, 'boxpc_fit_prob': self.fit_prob,
                'boxpc_out': self.boxpc.F3.out, 'boxpc_out_last': self.boxpc_nets[-1].F3.out, 'loss': self.loss, 'strong_loss': lo.loss, 'terms': self.terms,
                'loss_terms': lo.terms, 'iou2ds': lo.iou2d, 'iou3ds': lo.iou3d, 'W_iou2ds': self.W_iou2d, 'W_iou3ds': self.W_iou3d,
                'total_delta': self.total_delta, 'refined_center': self.cur_center,
                'refined_dims': self.cur_dims, 'refined_theta': self.cur_theta}
