"""Launch schedule of two INDEPENDENT dependency chains of a step: which launches share a launch.

The reference hands its graph to TensorFlow's executor, which runs whatever is independent concurrently (sess.run,
train_semisup.py:405-411).  Here a step is a static list of launches, and at B = 32 a quarter of it is spent in small launches (1-64
workgroups: batch-norm finalizers, the FC heads of the T-Net / box net and their backward, column sums) that occupy a few of the 256
CUs for the 4-10 us a dependent kernel boundary costs.  One independence is large enough to matter: nothing the T-Net / box net
compute -- forward, loss, backward -- reaches the segmentation net's backward (`mask` is a hard comparison,
semisup_models.py:150-151), so

    S = the segmentation net's backward                     (GEMM launches with a finalizer between any two)
    T = T-Net forward, box-net forward, loss, box-net backward, T-Net backward   (GEMMs, and runs of up to nine small launches)

can be interleaved freely.  A small launch of one chain then RIDES in a GEMM launch of the other (csrc/rider_dev.h: the first
workgroups of the launch run the small ops -- a whole run of them, with a barrier among those workgroups between dependent ops -- and
the GEMM's tiles follow), two small launches of different chains share one launch (t3d_small_pair), and what finds no partner runs
alone.  The step takes the sum of its launches, so the assignment is a sequence alignment: a dynamic programme over (ops of S
consumed, ops of T consumed) minimising the summed launch time under a duration model; each chain keeps its own order.

Results are bit-identical to the unscheduled step (same kernels' bodies, same arguments; tests/test_schedule_cpu.py on the
specification library, tests/test_riders_gpu.py on the device)."""
import ctypes as C
import os

import numpy as np
import torch

from . import abi

HOST_FN = {'t3d_pointmlp_fwd': 't3d_pointmlp_fwd_r', 't3d_pointmlp_bwd': 't3d_pointmlp_bwd_r',
           't3d_pool_bwd_stage1': 't3d_pool_bwd_stage1_r', 't3d_pool_bwd_stage2': 't3d_pool_bwd_stage2_r',
           't3d_pointmlp_wgrad': 't3d_pointmlp_wgrad_r'}
MAX_RUN = min(int(os.environ.get('T3D_RIDER_RUN', '10')), abi.RIDER_MAX_OPS)          # longest run of dependent small ops in one rider set (1: no in-launch barriers)
# measured (tools/bench_riders.py, four dependent FC ops, 29 us as four launches): 41 us as one set of its own, 52 us inside a GEMM
# launch that fills the chip; a 87 us two-round GEMM launch hosting them takes 96 us (the riders' slots push 16 of its tiles into a
# third round), ONE riding op is free
# (round 4, hosts on the three-term bf16 path: re-swept on the whole step, same box: 1.85 -> 1.262 ms, 1.5 -> 1.244, 1.3 -> 1.248)
RIDER_SLOWDOWN = float(os.environ.get('T3D_RIDER_SLOW', '1.5' if os.environ.get('T3D_X3', '1') != '0' else '1.85'))      # a rider run vs the stand-alone launches of its ops
HOST_STRETCH = float(os.environ.get('T3D_RIDER_STRETCH', '0.2'))     # what a hosted run adds to its host, per us of the run ...
HOST_STRETCH_MAX = float(os.environ.get('T3D_RIDER_STRETCH_MAX', '8.0'))   # ... at most (us): displaced tiles wait for ONE slot to free
RIDER_SLOT_SHARE = 16.0 / 512.0                                       # 16 rider workgroups of 512 resident slots
HOST_OVERHEAD_US = float(os.environ.get('T3D_RIDER_COST', '0.5'))
# a run may take at most this fraction of its host's own time: riders that outlast their GEMM turn the launch into a latency chain
# with 500 idle workgroup slots (and make the GEMM kernel's measured duration that of its riders)
RIDER_MAX_FRAC = float(os.environ.get('T3D_RIDER_MAXFRAC', '1.0'))
WIDE_SHARE = float(os.environ.get('T3D_RIDER_WIDE', '0.5'))          # of a wide rider's stand-alone time that its host launch grows by (<= 0: never host one)


def _first(arg):
    return arg[0] if isinstance(arg, tuple) else arg


def is_host(lib, name, arg):
    """Does this GEMM launch have a rider form?  Asked of the library itself (t3d_*_hosts_riders: the `_r` launcher's own dispatch
    run without launching), so the kernel-selection rules -- tile widths, register kernels, one-pass forms, T3D_X3 -- live in ONE
    place.  Where a launch has no rider form the `_r` entry point would run the set as its own launch first: slower, never wrong."""
    if name not in HOST_FN or arg is None:
        return False
    fn = getattr(lib, name + '_hosts_riders', None)
    if fn is None:
        return False
    args = arg if isinstance(arg, tuple) else (arg,)
    return fn(*[C.byref(a) for a in args]) == 1


def small_op(name, arg, depends=0):
    o = abi.SmallOp()
    o.kind, o.depends = abi.RIDER_KIND[name], int(depends)
    C.memmove(C.byref(o.u), C.byref(arg), C.sizeof(arg))
    return o


def can_ride(lib, name, arg):
    if name not in abi.RIDER_KIND or arg is None or isinstance(arg, tuple) or not hasattr(lib, 't3d_riders_plan'):
        return False
    if name in abi.RIDER_WIDE and WIDE_SHARE <= 0:
        return False
    rs = abi.RiderSet()
    rs.ops[0], rs.n_ops = small_op(name, arg), 1
    return lib.t3d_riders_plan(C.byref(rs)) == 0


def can_pair(name, arg):
    if name not in abi.SMALL_KIND or arg is None:
        return False
    return not (name == 't3d_bn_bwd_finalize' and arg.psum_dz and arg.n_tiles > 512)      # the 64-group form runs alone


def _x3_requested(a):
    """The launch struct asks for the three-term bf16 arithmetic (engine.Runtime.gemm_arithmetic; a struct without the request -- the
    kernel tests' -- follows the library's default and its T3D_X3 experiment switch)."""
    req = getattr(a, 'arith', abi.ARITH_AUTO) if a is not None else abi.ARITH_AUTO
    return req == abi.ARITH_BF16X3 or (req == abi.ARITH_AUTO and os.environ.get('T3D_X3', '1') != '0')


# ---- duration model (us on one MI355X, from bench.py --call_detail at B=32 N=1024; only the ORDER of magnitude steers the alignment) ----
def est_us(name, arg):
    a = _first(arg) if arg is not None else None
    # fp32 GEMMs on the bf16 matrix pipe (csrc/pointmlp.hip PathX3, the default) are 1.1-1.5x faster than the fp32-MFMA kernels
    x3 = _x3_requested(a) and getattr(a, 'dtype', 0) == abi.F32
    if name == 't3d_pointmlp_fwd':
        if a.K <= 4:
            return 5.0                                            # register kernel, bound by its output store
        return 6.0 + 2.0 * a.M * a.K * a.N / (float(os.environ.get('T3D_SCHED_FWD_RATE', '1.6e8')) if x3 else 1.0e8)
    if name == 't3d_pointmlp_bwd':
        return 6.0 + 4.0 * a.M * a.K * a.N / (float(os.environ.get('T3D_SCHED_BWD_RATE', '1.25e8')) if x3 else 0.95e8)
    if name == 't3d_pointmlp_wgrad':
        if a.K <= 4:
            return 7.0
        return 6.0 + 2.0 * a.M * max(a.K, 8) * a.N / 0.5e8
    if name == 't3d_pointmlp_dgrad':
        return 6.0 + 2.0 * a.M * a.K * a.N / 0.9e8
    if name == 't3d_pool_bwd_stage1':
        return 6.0 + 2.0 * a.M * a.K * a.K / 0.9e8
    if name == 't3d_pool_bwd_stage2':
        f = a
        return 8.0 + 2.0 * arg[1].M * f.K * f.K / (1.0e8 if _x3_requested(arg[1]) else 0.85e8)
    if name in ('t3d_fc_fwd', 't3d_fc_bwd', 't3d_fc_dinput'):      # 7.4 (256 x 256) ... 9.9 (512 x 512) fwd, 11-14 bwd
        rb = max(1.0, a.B / 32.0)
        kn = float(a.K) * a.N * rb
        return (8.0 + 2.5e-5 * kn) if name == 't3d_fc_bwd' else (5.5 + 1.7e-5 * kn)
    if name == 't3d_bn_fwd_finalize':
        return 6.3 if a.pool_pmax else 4.0
    if name == 't3d_pool_bwd_mid' and a is not None and hasattr(a, 'sparse'):      # 32.6 (N 512, K 256), 24.7 (1024, 128), 18.8 (256, 128)
        return 15.0 + 1.1e-4 * a.sparse.N * a.sparse.K * max(1.0, a.sparse.B / 32.0)
    return {'t3d_bn_fwd_finalize': 4.0, 't3d_bn_bwd_finalize': 4.3, 't3d_dy_colsum': 4.0, 't3d_strong_loss': 12.0,
            't3d_pool_bwd_mid': 25.0, 't3d_small_pair': 10.0, 't3d_seg_head': 13.0, 't3d_seg_finalize': 4.0}.get(name, 6.0)


class _Op:
    __slots__ = ('call', 'name', 'arg', 'host', 'ride', 'pair', 'us', 'wide')

    def __init__(self, lib, call):
        self.call = call
        self.name, _, self.arg = call
        self.host = is_host(lib, self.name, self.arg)
        self.ride = can_ride(lib, self.name, self.arg)
        self.pair = can_pair(self.name, self.arg)
        self.us = est_us(self.name, self.arg)
        self.wide = self.name in abi.RIDER_WIDE


def align(S, T, max_run=None):
    """[('solo', chain, i) | ('host', chain, i, j0, j1) | ('pair', i, j)] of minimal summed time: `host` = op i of `chain`
    (0 = S, 1 = T) carries ops [j0, j1) of the OTHER chain as riders."""
    max_run = MAX_RUN if max_run is None else max_run
    nS, nT = len(S), len(T)
    INF = float('inf')
    cost = np.full((nS + 1, nT + 1), INF)
    back = {}
    cost[0, 0] = 0.0
    chains = (S, T)

    def relax(i, j, c, step):
        if c < cost[i, j]:
            cost[i, j] = c
            back[(i, j)] = step

    for i in range(nS + 1):
        for j in range(nT + 1):
            c0 = cost[i, j]
            if c0 == INF:
                continue
            pos = (i, j)
            for ch in (0, 1):
                me, other = chains[ch], chains[1 - ch]
                p, q = pos[ch], pos[1 - ch]
                if p >= len(me):
                    continue
                x = me[p]
                nxt = (lambda dp, dq: (p + dp, q + dq) if ch == 0 else (q + dq, p + dp))
                relax(*nxt(1, 0), c0 + x.us, ('solo', ch, p, pos))
                if x.host:
                    run = 0.0
                    for r in range(1, max_run + 1):
                        if q + r > len(other) or not other[q + r - 1].ride:
                            break
                        if getattr(other[q + r - 1], 'wide', False):
                            # a WIDE rider (t3d_pool_bwd_mid: hundreds of latency-bound workgroups) is alone in its set; it is real
                            # work for the chip: the launch takes the host's time plus a share of the rider's
                            if r == 1:
                                relax(*nxt(1, 1), c0 + x.us + WIDE_SHARE * other[q].us + HOST_OVERHEAD_US, ('host', ch, p, q, q + 1, pos))
                            break
                        run += other[q + r - 1].us
                        rh = run * (RIDER_SLOWDOWN if r > 1 else 1.2)
                        if rh > RIDER_MAX_FRAC * x.us and r > 1:
                            break
                        # what the riders' slots cost the host: their share of the chip for the run's duration, plus the tiles of
                        # a one- or two-round launch that slip into another round (bounded by a fraction of the run)
                        stretch = (RIDER_SLOT_SHARE * rh + min(HOST_STRETCH * rh, HOST_STRETCH_MAX)) if r > 1 else 0.0
                        relax(*nxt(1, r), c0 + max(x.us + stretch, rh) + HOST_OVERHEAD_US, ('host', ch, p, q, q + r, pos))
            if i < nS and j < nT and S[i].pair and T[j].pair:
                relax(i + 1, j + 1, c0 + max(S[i].us, T[j].us) + 0.5, ('pair', i, j, pos))
    steps, pos = [], (nS, nT)
    while pos != (0, 0):
        st = back[pos]
        steps.append(st[:-1])
        pos = st[-1]
    steps.reverse()
    return steps, float(cost[nS, nT])


class RiderBarrierTimeout(RuntimeError):
    """A rider barrier gave up waiting (csrc/rider_dev.h: bounded spin): the ops behind it ran on incomplete inputs, so every
    result of the affected steps -- weights, Adam moments, moving statistics -- is invalid."""


class RiderSets:
    """Owner of the barrier words of a program's rider sets: ONE pooled device buffer (a row of 2 * RIDER_MAX_OPS + 2 words per set,
    the set's timeout word at [-2]), so that the training loop can look at every timeout word with one 4-byte read."""
    POOL = 128

    def __init__(self, rt):
        self.rt = rt
        self.sets = []       # (RiderSet struct | None, keep-alive ...)
        self.width = 2 * abi.RIDER_MAX_OPS + 2
        self.pools, self.used = [], 0

    def _row(self):
        if self.used == len(self.pools) * self.POOL:
            self.pools.append(torch.zeros((self.POOL, self.width), dtype=torch.int32, device=self.rt.device))
        row = self.pools[-1][self.used % self.POOL]
        self.used += 1
        return row

    def make(self, ops):
        """ops: [(name, arg struct)] in chain order -> abi.RiderSet (ops by value, zeroed barrier words on the device)."""
        lib, n = self.rt.lib, len(ops)
        rs = abi.RiderSet()
        for k, (name, arg) in enumerate(ops):
            rs.ops[k] = small_op(name, arg, depends=1 if k > 0 else 0)
        rs.n_ops = n
        abi.check(lib.t3d_riders_plan(C.byref(rs)), 't3d_riders_plan')
        sync = self._row()
        rs.sync = C.cast(C.c_void_p(sync.data_ptr()), C.POINTER(C.c_uint32))
        self.sets.append((rs, None, sync, None))
        return rs

    def timeouts(self):
        """Number of sets whose barrier ever gave up waiting (must be 0; t3d.h t3d_rider_set.sync).  One device reduction + one
        4-byte read per pool of 128 sets (a step has ~20)."""
        n = 0
        for k, pool in enumerate(self.pools):
            rows = min(self.POOL, self.used - k * self.POOL)
            n += int((pool[:rows, self.width - 2] != 0).sum().item())
        return n

    def check(self):
        n = self.timeouts()
        if n:
            raise RiderBarrierTimeout(
                '%d rider set(s) report a barrier time-out (csrc/rider_dev.h): the small ops behind that barrier ran before their '
                'inputs were complete, so the weights / Adam moments / moving statistics written since the last check are invalid.  '
                'Restore the last checkpoint; T3D_OVERLAP=0 runs the step without riders.' % n)


def _host_call(lib, op, rs):
    fn, ref = getattr(lib, HOST_FN[op.name]), C.byref(rs)
    if isinstance(op.arg, tuple):
        refs = tuple(C.byref(a) for a in op.arg)
        if len(refs) == 2:
            thunk = lambda s, fn=fn, r=refs, ref=ref: fn(r[0], r[1], ref, s)
        else:
            thunk = lambda s, fn=fn, r=refs, ref=ref: fn(r[0], r[1], r[2], ref, s)
    else:
        aref = C.byref(op.arg)
        thunk = lambda s, fn=fn, aref=aref, ref=ref: fn(aref, ref, s)
    return (HOST_FN[op.name], thunk, op.arg)


def overlap_chains(rt, S_calls, T_calls, sets, max_run=None):
    """Merged call list of two independent chains (lists of Plan call tuples); `sets`: RiderSets that keeps the device tables alive.
    Returns (calls, report)."""
    lib = rt.lib
    S = [_Op(lib, c) for c in S_calls]
    T = [_Op(lib, c) for c in T_calls]
    steps, total = align(S, T, max_run)
    chains = (S, T)
    out, rep = [], {'serial_us': sum(o.us for o in S + T), 'scheduled_us': total, 'hosted': 0, 'rider_ops': 0, 'pairs': 0, 'solo': 0,
                    'lines': []}
    short = lambda o: '%s%s[%.0f]' % ('ST'[o in T and o not in S], o.name[4:], o.us)
    for st in steps:
        if st[0] == 'solo':
            out.append(chains[st[1]][st[2]].call)
            rep['solo'] += 1
            rep['lines'].append(short(chains[st[1]][st[2]]))
        elif st[0] == 'host':
            _, ch, p, q0, q1 = st
            riders = chains[1 - ch][q0:q1]
            rs = sets.make([(o.name, o.arg) for o in riders])
            out.append(_host_call(lib, chains[ch][p], rs))
            rep['hosted'] += 1
            rep['rider_ops'] += len(riders)
            rep['lines'].append('%s  <- %s' % (short(chains[ch][p]), ' , '.join(short(o) for o in riders)))
        else:
            _, i, j = st
            oa, ob = abi.SmallOp(), abi.SmallOp()
            for o, x in ((oa, S[i]), (ob, T[j])):
                o.kind = abi.SMALL_KIND[x.name]
                C.memmove(C.byref(o.u), C.byref(x.arg), C.sizeof(x.arg))
            fn, ra, rb = lib.t3d_small_pair, C.byref(oa), C.byref(ob)
            sets.sets.append((None, oa, ob, None))      # keep-alive
            out.append(('t3d_small_pair', (lambda s, fn=fn, ra=ra, rb=rb: fn(ra, rb, s)), (oa, ob)))
            rep['pairs'] += 1
            rep['lines'].append('%s || %s' % (short(S[i]), short(T[j])))
    return out, rep
