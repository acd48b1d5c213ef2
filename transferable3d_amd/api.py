"""Graph / placeholder / session layer that the reference-named builder modules sit on.

The reference drives TensorFlow 1 like this (train_semisup.py:204-277, 405-411):

    with tf.Graph().as_default():
        pls = MODEL.placeholder_inputs(B, N, C)
        pred, end_points = MODEL.get_semi_model(...)
        loss = MODEL.get_semi_loss(pred, labels, end_points, c=FLAGS)
        train_op = tf.train.AdamOptimizer(lr).minimize(loss, global_step=batch)
        sess = tf.Session(); sess.run(init)
        sess.run([loss, train_op], feed_dict={pc_pl: batch, ...})

The same sequence works here with `api.Graph`, `api.AdamOptimizer`, `api.Session`.  Builders only RECORD which
sub-networks exist and allocate their HBM buffers; the fused launch schedule is emitted on the first
`Session.run`, pruned by what is fetched (a fetch list without the train op compiles the forward-only plan,
exactly like TF prunes its graph), and on the GPU the schedule is captured into a hipGraph and replayed.

`is_training` may be a Python bool, or -- as in the reference (train_semisup.py:210 `is_training_pl = tf.placeholder(tf.bool,
shape=())`) -- a placeholder made by `is_training_placeholder()` and fed per `Session.run`: the session keeps one launch schedule
per (train op fetched?, is_training) combination (batch statistics + EMA update + dropout vs moving statistics), all over the same
variables and buffers, so `eval_one_epoch` runs in the graph that trains.
"""
import contextlib

import numpy as np
import torch

from . import abi
from .engine import Runtime, VarStore
from .nets import Graph as _EngineGraph, Inputs, ModelAssembly, make_schedule

_stack = []


class Shape(tuple):
    """tf.TensorShape as far as the reference uses it: indexable, and `.as_list()` (semisup_models.py:72)."""

    def as_list(self):
        return list(self)


class Tensor:
    """Handle of a device buffer that holds a value after `Session.run` (the analogue of a tf.Tensor)."""

    def __init__(self, ctx, buf=None, shape=None, name=None, producer=None, field=None):
        self.ctx, self.buf, self.name, self.producer, self.field = ctx, buf, name, producer, field
        self._shape = Shape(shape) if shape is not None else (Shape(buf.shape) if buf is not None else None)

    def get_shape(self):
        return self._shape

    shape = property(get_shape)

    def numpy(self):
        a = self.buf.detach().cpu().numpy()
        if self._shape and a.size != int(np.prod(self._shape)):          # rows padded in HBM (the point cloud at C = 3 or 6)
            a = a.reshape(-1, a.shape[-1])[:, :self._shape[-1]]
        return a.reshape(self._shape)

    def __repr__(self):
        return 'Tensor(%s, shape=%s)' % (self.name, self._shape)


class Placeholder(Tensor):
    """Feedable device buffer; `field` names the slot of nets.Inputs it aliases (None: accepted and ignored)."""


class LazyPoints(Tensor):
    """xyz columns of the point cloud minus a per-frustum vector -- never materialised: the subtraction is folded
    into the first layer's operand load (semisup_models.py:159, 207)."""

    def __init__(self, ctx, pc, ncols, sub=None):
        Tensor.__init__(self, ctx, None, (pc.shape[0], pc.shape[1], ncols), 'lazy_points')
        self.pc, self.ncols, self.sub = pc, ncols, sub


class Graph:
    """tf.Graph analogue: owns the runtime, the variable store and the recorded model assembly."""

    def __init__(self, rt=None, seed=0, vars=None, inline_dropout=False, dtype=None):
        """inline_dropout: the segmentation head draws its dropout mask inside its kernel (no [M,128] mask tensor, no mask launch); a
        graph built this way cannot be fed an explicit mask for that scope.  dtype: 'f32' (default; env T3D_DTYPE) or 'bf16' -- the
        element type of the per-point layer tensors and of the GEMM operands (t3d.h T3D_BF16: BASELINE configs[4])."""
        import os
        self._rt, self.seed, self._vars, self.inline_dropout = rt, seed, vars, inline_dropout
        self.dtype = dtype or os.environ.get('T3D_DTYPE', 'f32')
        assert self.dtype in ('f32', 'bf16'), self.dtype
        self.engine = None          # nets.Graph, created by the first placeholder_inputs
        self.inputs = None
        self.assembly = None
        self.is_training = None
        self.loss = None
        self.train_op = None
        self.compiled = {}
        self.dataset = None         # dataset.DeviceFrustumSet: batches assembled on the device (Graph.use_device_dataset)
        self.dataset_opts = {}

    @contextlib.contextmanager
    def as_default(self):
        _stack.append(self)
        try:
            yield self
        finally:
            _stack.pop()

    @property
    def rt(self):
        if self._rt is None:
            self._rt = Runtime()
        return self._rt

    def ensure_engine(self, batch_size, num_point, num_channel):
        if self.engine is None:
            vs = self._vars or VarStore(self.rt, seed=self.seed)
            self.engine = _EngineGraph(batch_size, num_point, num_channel, rt=self.rt, seed=self.seed, vars=vs, dtype=self.dtype)
            self.engine.inline_dropout = self.inline_dropout
            self.inputs = Inputs(self.engine)
        else:
            e = self.engine
            assert (e.B, e.rpf, e.C) == (batch_size, num_point, num_channel), 'one problem size per graph'
        return self.engine

    @property
    def vars(self):
        return self.engine.vars

    @property
    def hyper(self):
        """Device schedule state [global step, lr, bn_decay, Adam lr_t] (the reference's `batch` variable lives in [0])."""
        return self.engine.hyper

    def use_device_dataset(self, dataset, seed=0, **aug):
        """Feed the training plan from a data set resident in HBM: every Session.run of the train op assembles its own
        batch on the device (t3d_batch_assemble), nothing needs to be fed.  `boxpc_perturb=FLAGS` (the BOXPC_* flags) adds the
        Box-PC Fit sampler (t3d_boxpc_perturb) behind it."""
        self.dataset, self.dataset_opts = dataset, dict(seed=seed, **aug)

    def ensure_assembly(self, c, use_one_hot=False):
        if self.assembly is None:
            self.assembly = ModelAssembly(self.engine, c, self.inputs, use_one_hot)
        return self.assembly


# ---- the handful of tf.* graph functions the reference's sub-network bodies use between tf_util calls ------------------------------
# (semisup_models.py:69-139: variable_scope, expand_dims, concat, tile, squeeze; 150-162, 184-185: multiply by the mask).  They build
# LAZY handles -- a tiled global feature or a point-tensor/global concat never exists in memory; tf_util.conv2d recognises the handle
# and emits the split form (per-point GEMM + one per-frustum FC as a row bias).
class GlobalVec(Tensor):
    """(B, 1, 1, C) per-frustum vector made of column blocks [(device buffer [B, >= n], n)] (pooled feature [+ one-hot])."""

    def __init__(self, ctx, parts, name='global_vec'):
        Tensor.__init__(self, ctx, None, (ctx.engine.B, 1, 1, sum(n for _, n in parts)), name)
        self.parts = parts

    def numpy(self):
        return np.concatenate([b.detach().float().cpu().numpy()[:, :n] for b, n in self.parts], axis=1).reshape(self._shape)


class TiledGlobal(Tensor):
    """tf.tile(global, [1, num_point, 1, 1]): (B, N, 1, C), never materialised."""

    def __init__(self, ctx, vec):
        Tensor.__init__(self, ctx, None, (ctx.engine.B, ctx.engine.rpf, 1, vec.shape[-1]), 'tiled_global')
        self.vec = vec


class ConcatPointGlobal(Tensor):
    """tf.concat(axis=3, values=[point tensor, tiled global]) (semisup_models.py:107-108), never materialised."""

    def __init__(self, ctx, point, tiled):
        Tensor.__init__(self, ctx, None, (ctx.engine.B, ctx.engine.rpf, 1, point.shape[-1] + tiled.shape[-1]), 'concat_point_global')
        self.point, self.tiled = point, tiled


class _VariableScope:
    def __init__(self, name):
        self.name = name


@contextlib.contextmanager
def variable_scope(name_or_scope, *args, **kwargs):
    """tf.variable_scope: prefixes the `scope` of the tf_util layers built inside (variables <outer>/<scope>/weights, ...)."""
    g = get_default_graph()
    name = name_or_scope.name if isinstance(name_or_scope, _VariableScope) else (name_or_scope or '')
    stack = g.__dict__.setdefault('_scope_stack', [])
    stack.append(name)
    try:
        yield _VariableScope('/'.join(x for x in stack if x))
    finally:
        stack.pop()


def scoped(scope):
    """`scope` under the variable scopes currently open on the default graph."""
    stack = get_default_graph().__dict__.get('_scope_stack', [])
    return '/'.join([x for x in stack if x] + [scope])


def _as_global_vec(ctx, t):
    if isinstance(t, GlobalVec):
        return t
    n = t.shape[-1]
    if t.buf is None or int(np.prod(t.shape)) != ctx.engine.B * n:
        raise NotImplementedError('expected a per-frustum [B, ..., C] tensor, got %r' % (t,))
    return GlobalVec(ctx, [(t.buf, n)], t.name)


def expand_dims(t, axis=-1):
    """tf.expand_dims on a handle: a pure shape change (the point cloud as a one-channel image [B,N,C,1]; a [B,C] vector as
    [B,1,C] / [B,1,1,C])."""
    shp = list(t.shape)
    ax = axis if axis >= 0 else len(shp) + 1 + axis
    shp.insert(ax, 1)
    import copy
    out = copy.copy(t)
    out._shape = Shape(shp)
    return out


def squeeze(t, axis=None):
    shp = [d for i, d in enumerate(t.shape) if not ((axis is None and d == 1) or (axis is not None and i in [a % len(t.shape) for a in axis]))]
    import copy
    out = copy.copy(t)
    out._shape = Shape(shp)
    return out


def tile(t, multiples):
    ctx = get_default_graph()
    if list(multiples) != [1, ctx.engine.rpf, 1, 1] or len(t.shape) != 4 or t.shape[1] != 1:
        raise NotImplementedError('tile: only a [B,1,1,C] global feature over the num_point axis (semisup_models.py:107)')
    return TiledGlobal(ctx, _as_global_vec(ctx, t))


def concat(values=None, axis=None, **kw):
    """tf.concat(values, axis) / tf.concat(axis=, values=) for the two forms of semisup_models.py:101-108: per-frustum vectors along
    the channel axis, and [point tensor, tiled global] along the channel axis."""
    if isinstance(values, int) and isinstance(axis, (list, tuple)):      # TF 0.x argument order
        values, axis = axis, values
    ctx = get_default_graph()
    vals = list(values)
    if axis not in (3, -1) or len(vals) != 2:
        raise NotImplementedError('concat: two tensors along the channel axis')
    a, b = vals
    if isinstance(b, TiledGlobal):
        return ConcatPointGlobal(ctx, a, b)
    va, vb = _as_global_vec(ctx, a), _as_global_vec(ctx, b)
    return GlobalVec(ctx, va.parts + vb.parts)


class MaskedPoints:
    """tf.multiply(net, mask): a per-point tensor times the hard [B,N,1,1] mask.  Never materialised -- the only consumer the
    reference has for it is the max-pool right behind it (semisup_models.py:184-188, 240-244), which takes `mask` as its row mask.
    A type of its own so that every OTHER consumer (conv2d, dropout, a fetch) fails instead of silently reading the unmasked
    activations."""

    def __init__(self, points, mask_buf):
        self.points, self.rowmask = points, mask_buf
        self.layer, self.ctx = getattr(points, 'layer', None), points.ctx
        self.shape = points.shape

    def get_shape(self):
        return self.points.get_shape()

    def numpy(self):
        raise NotImplementedError('a masked per-point tensor (api.multiply) can only feed tf_util.max_pool2d')


def multiply(x, y):
    """tf.multiply(net, mask) in front of a max-pool (semisup_models.py:184-185, 240-241): the mask becomes the pooled layer's row
    mask (the masked tensor itself is never written)."""
    if hasattr(x, 'spec') and getattr(y, 'buf', None) is not None and int(np.prod(y.shape)) == x.ctx.engine.M:
        return MaskedPoints(x, y.buf)
    raise NotImplementedError('multiply: a per-point tensor times the [B,N,1,1] mask')


class nn:
    """tf.nn.* activation functions as the `activation_fn` argument of the tf_util layers (tf_util.py:1321-1322)."""

    @staticmethod
    def relu(x):
        raise NotImplementedError('api.nn.relu is an activation_fn marker for the tf_util layers')

    @staticmethod
    def leaky_relu(x, alpha=0.2):
        raise NotImplementedError('api.nn.leaky_relu is an activation_fn marker for the tf_util layers')

    @staticmethod
    def tanh(x):
        raise NotImplementedError('api.nn.tanh is an activation_fn marker for the tf_util layers')


tanh = nn.tanh


def activation_name(fn):
    """'relu' / 'leaky_relu' / 'tanh' / None from the reference's activation_fn (a callable of this module's nn namespace, a torch /
    numpy function of the same name, or already a string)."""
    if fn is None or isinstance(fn, str):
        return fn
    if fn in (nn.relu, nn.leaky_relu, nn.tanh):
        return {nn.relu: 'relu', nn.leaky_relu: 'leaky_relu', nn.tanh: 'tanh'}[fn]
    name = getattr(fn, '__name__', '')
    if name in ('relu', 'leaky_relu', 'tanh'):
        return name
    raise NotImplementedError('activation_fn %r: tf.nn.relu, tf.nn.leaky_relu, tf.tanh or None at every call site of the reference' % (fn,))


def get_default_graph():
    if not _stack:
        _stack.append(Graph())
    return _stack[-1]


def reset_default_graph():
    del _stack[:]


class BoolPlaceholder(Placeholder):
    """`tf.placeholder(tf.bool, shape=())` for is_training: truthy at build time (graphs are assembled in their training form), its
    fed value picks the schedule at run time."""

    def __init__(self, ctx, name='is_training'):
        Placeholder.__init__(self, ctx, None, (), name, field=None)

    def __bool__(self):
        return True


def is_training_placeholder(name='is_training'):
    return BoolPlaceholder(get_default_graph(), name)


def placeholder(field, shape, dtype=torch.float32, name=None):
    g = get_default_graph()
    buf = getattr(g.inputs, field, None) if field else None
    return Placeholder(g, buf, shape, name or field, field=field)


class TrainOp:
    def __init__(self, ctx, loss, var_prefixes, sched):
        self.ctx, self.loss, self.var_prefixes, self.sched = ctx, loss, var_prefixes, sched


class AdamOptimizer:
    """tf.train.AdamOptimizer (train_semisup.py:230) with the reference's staircase schedules
    (train_semisup.py:127-145) evaluated on the device each step."""

    def __init__(self, learning_rate=1e-3, beta1=0.9, beta2=0.999, epsilon=1e-8, decay_step=800000, decay_rate=0.5,
                 world_size=1):
        self.lr, self.b1, self.b2, self.eps = learning_rate, beta1, beta2, epsilon
        self.decay_step, self.decay_rate, self.world_size = decay_step, decay_rate, world_size

    def minimize(self, loss, global_step=None, var_list=None):
        """var_list: iterable of scope prefixes (tf.get_collection(scope=...) regex-prefix semantics,
        train_semisup_adv.py:415-422) or None for every trainable variable."""
        ctx = loss.ctx
        if ctx.assembly is None:
            raise NotImplementedError(
                'minimize: the loss was built from operator-level tf_util calls, which are a forward-only surface (launches go into '
                'the forward schedule as they are called); the training step -- backward + Adam -- is emitted for graphs built from '
                'the fused sub-networks (semisup_models.v1_inst_seg / v1_tnet / v1_box_est, semisup_v1_sunrgbd.get_semi_model, '
                'boxpc_sunrgbd.get_model), whose variables and outputs are the same')
        sched = make_schedule(ctx.engine.B * self.world_size, self.lr, self.decay_step, self.decay_rate, beta1=self.b1,
                              beta2=self.b2)
        ctx.train_op = TrainOp(ctx, loss, list(var_list) if var_list is not None else None, sched)
        ctx.optimizer = self
        return ctx.train_op


class MomentumOptimizer(AdamOptimizer):
    """tf.train.MomentumOptimizer(learning_rate, momentum=MOMENTUM) of `--optimizer momentum` (train_semisup.py:226-228,
    train_boxpc.py:247, train_semisup_adv.py:296): accum = momentum * accum + g; w -= lr * accum, with the same staircase schedules."""

    def __init__(self, learning_rate=1e-3, momentum=0.9, decay_step=800000, decay_rate=0.5, world_size=1):
        AdamOptimizer.__init__(self, learning_rate, decay_step=decay_step, decay_rate=decay_rate, world_size=world_size)
        self.momentum = float(momentum)


def make_optimizer(FLAGS, world_size=1):
    """The optimiser the reference's drivers pick from --optimizer / --momentum (train_semisup.py:226-231)."""
    kind = getattr(FLAGS, 'optimizer', 'adam')
    if kind == 'adam':
        return AdamOptimizer(FLAGS.learning_rate, decay_step=FLAGS.decay_step, decay_rate=FLAGS.decay_rate, world_size=world_size)
    if kind == 'momentum':
        return MomentumOptimizer(FLAGS.learning_rate, momentum=getattr(FLAGS, 'momentum', 0.9), decay_step=FLAGS.decay_step,
                                 decay_rate=FLAGS.decay_rate, world_size=world_size)
    raise ValueError('--optimizer %r: adam or momentum (train_semisup.py:39)' % (kind,))


def init_data_parallel(rt=None, gpu=0):
    """One process per GPU: reads WORLD_SIZE / RANK / LOCAL_RANK (torch.distributed.run), joins the process group -- RCCL
    on the GPUs, gloo with a CPU runtime -- and returns (world, rank, process group or None).  The three training drivers call this
    where the reference selects its one GPU (train_semisup.py:31,205)."""
    import os
    import torch
    world, rank = int(os.environ.get('WORLD_SIZE', '1')), int(os.environ.get('RANK', '0'))
    if world > 1:
        import datetime
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')      # dmabuf IPC (RCCL between the ranks of one node)
        on_gpu = torch.cuda.is_available() and rt is None
        if on_gpu:
            torch.cuda.set_device(int(os.environ.get('LOCAL_RANK', '0')) % torch.cuda.device_count())
        if not dist.is_initialized():
            dist.init_process_group('nccl' if on_gpu else 'gloo',
                                    timeout=datetime.timedelta(seconds=int(os.environ.get('T3D_DIST_TIMEOUT_S', '600'))))
        return world, rank, dist.group.WORLD
    if rt is None and torch.cuda.is_available():
        torch.cuda.set_device(gpu)
    return 1, 0, None


def finish_data_parallel(world):
    if world > 1:
        import torch.distributed as dist
        if dist.is_initialized():
            dist.destroy_process_group()


class Session:
    """tf.Session analogue.  run(fetches, feed_dict): H2D copies of the fed arrays, one replay of the compiled
    launch schedule, D2H of the fetched tensors."""

    def __init__(self, graph=None, use_hip_graph=None, dropout_seed=1234, process_group=None, force_dist=None):
        """process_group: data-parallel replicas (one process per GPU); force_dist (default: env T3D_FORCE_DIST=1): take the
        multi-rank path -- bucketed backward, collectives -- even when the group has one rank."""
        import os
        self.g = graph or get_default_graph()
        self.use_hip_graph = use_hip_graph
        self.dropout_seed = dropout_seed
        self.pg = process_group
        self.force_dist = (os.environ.get('T3D_FORCE_DIST', '0') == '1') if force_dist is None else bool(force_dist)
        self.dp_flat = os.environ.get('T3D_DP_FLAT', '1') == '1'      # one all-reduce between backward and Adam (default since round 3: the
        #                                                                scheduled backward of step.TrainStep._overlap has no early bucket); 0: three buckets
        self.steps = {}

    # -- compilation ------------------------------------------------------------------------------------
    def _compile(self, train, is_training=None):
        g, e = self.g, self.g.engine
        if is_training is None or not isinstance(g.is_training, BoolPlaceholder):
            is_training = bool(g.is_training)
        key = ('train' if train else 'infer') + ('' if is_training else '_eval')
        if key in self.steps:
            return self.steps[key]
        from .engine import Plan
        if g.assembly is None:
            # operator-level graph (tf_util.conv2d / fully_connected / dropout / batch_norm_for_* called one by one): the launches
            # were recorded into the engine's forward plan as the ops were built; a run = the dropout-mask generators + that plan
            if 'ops' not in self.steps:
                pre = Plan(e.rt)
                e.emit_cast_weights(pre)      # (the GEMM kernels' copies of the weights: bf16 twin / three bf16 planes)
                e.emit_dropout_masks(pre, seed=self.dropout_seed)
                if not e.finalized:
                    e.finalize()
                self.steps['ops'] = _Step(self, pre, e.fwd, None, None, False)
            return self.steps['ops']
        # the training schedule reserves the weight-gradient slabs: it is compiled first, whatever is run first
        if not train and g.train_op is not None and not e.finalized:
            self._compile(True, True)
        assert not (train and e.finalized), 'the training step must be compiled before any forward-only schedule of the graph'
        asm = g.assembly
        pre, fwd, bwd, opt = Plan(e.rt), Plan(e.rt), Plan(e.rt), Plan(e.rt)
        e.dropout_seed = self.dropout_seed
        with_loss = g.loss is not None
        if train:
            top = g.train_op
            if g.dataset is not None:
                opts = dict(g.dataset_opts)
                perturb = opts.pop('boxpc_perturb', None)
                e.emit_batch_assemble(pre, g.dataset, g.inputs, **opts)
                if perturb is not None:        # Box-PC Fit samples: the label box becomes a perturbed box with an IoU target
                    e.emit_boxpc_perturb(pre, g.inputs, perturb, seed=opts.get('seed', 0) ^ 0x5bd1e995)
            e.emit_schedule(pre, top.sched)
            e.emit_dropout_masks(pre, seed=self.dropout_seed)
        if train and type(asm).__name__ == 'SemiModelF':
            asm.emit_forward(fwd, is_training, with_loss, train=True)      # (the W_ heads become a chain of their own: nets.SemiModelF)
        else:
            asm.emit_forward(fwd, is_training, with_loss)
        if train:
            # data parallel: the backward declares its gradient buckets (all-reduce of a finished bucket beside the rest)
            e.dp_buckets = self.pg is not None and (self.pg.size() > 1 or self.force_dist) and not self.dp_flat
            asm.emit_backward(bwd)
            o = g.optimizer
            world = self.pg.size() if self.pg is not None else 1
            if isinstance(o, MomentumOptimizer):
                e.emit_momentum(opt, prefixes=top.var_prefixes, momentum=o.momentum, grad_scale=1.0 / world)
            else:
                e.emit_adam(opt, prefixes=top.var_prefixes, beta1=o.b1, beta2=o.b2, eps=o.eps, grad_scale=1.0 / world)
        if not e.finalized:
            e.finalize()
        step = _Step(self, pre, fwd, bwd, opt, train)
        self.steps[key] = step
        g.compiled[key] = True
        return step

    def check_riders(self):
        """Raises schedule.RiderBarrierTimeout if a rider barrier of any compiled step ever timed out (step.TrainStep.check_riders;
        run() looks every T3D_RIDER_CHECK_EVERY steps by itself, the drivers call this before they write a checkpoint)."""
        for st in self.steps.values():
            st.impl.check_riders()

    def run(self, fetches, feed_dict=None):
        single = not isinstance(fetches, (list, tuple))
        flist = [fetches] if single else list(fetches)
        train = any(isinstance(f, TrainOp) for f in flist)
        batch, masks = {}, {}
        mode = None
        for pl, val in (feed_dict or {}).items():
            if isinstance(pl, BoolPlaceholder):
                mode = bool(val)
            elif isinstance(pl, Placeholder):
                if pl.field is not None:
                    batch[pl.field] = np.asarray(val)
            elif isinstance(pl, str):                      # dropout-mask injection for parity tests: scope -> mask
                masks[pl] = np.asarray(val)
        if masks:
            unknown = [k for k in masks if k not in self.g.engine.dropout_masks]
            if unknown:
                raise ValueError('no stored dropout mask for scope(s) %s (a graph built with inline_dropout=True draws the segmentation '
                                 'head mask inside its kernel)' % unknown)
            batch['dropout_masks'] = masks
        if isinstance(self.g.is_training, BoolPlaceholder) and mode is None:
            if not train:
                raise ValueError('You must feed a value for placeholder tensor %r' % self.g.is_training.name)
            mode = True
        step = self._compile(train, mode)
        self.g.inputs.load(batch)
        step.run(skip_mask_generation=bool(masks))
        out = []
        for f in flist:
            if isinstance(f, TrainOp):
                out.append(None)
            elif isinstance(f, Tensor):
                out.append(f.numpy())
            elif isinstance(f, (tuple, list)):
                out.append(type(f)(x.numpy() for x in f))
            else:
                raise TypeError('cannot fetch %r' % (f,))
        return out[0] if single else out


class _Step:
    """pre (batch assembly, schedules, dropout masks) -> fwd -> bwd -> [bucketed all-reduce] -> adam: step.TrainStep, the object
    bench.py times; hipGraph-captured on the GPU (also when dropout masks are fed: that variant is captured without the mask
    generator launches)."""

    def __init__(self, sess, pre, fwd, bwd, opt, train):
        from .step import TrainStep
        self.sess, self.pre, self.fwd, self.bwd, self.opt, self.train = sess, pre, fwd, bwd, opt, train
        e = sess.g.engine
        self.impl = TrainStep(e, pre if (train or len(pre)) else None, fwd, bwd if train else None, opt if train else None, process_group=sess.pg,
                              use_hip_graph=sess.use_hip_graph, force_dist=sess.force_dist)

    @property
    def n_runs(self):
        return self.impl.n_runs

    def run(self, skip_mask_generation=False):
        self.impl.run(generate_masks=not skip_mask_generation)
