"""One training / inference step as the drivers run it: pre (batch assembly, schedules, dropout masks) -> forward -> backward
[-> gradient all-reduce, bucketed and overlapped with the rest of the backward] -> Adam, captured into hipGraphs on the GPU.

This is the ONE step object behind `api.Session.run`, `bench.py` and the trajectory parity tests, so that what is timed is what is
parity-tested.  It replaces the reference's `sess.run([..., train_op], feed_dict)` (train_semisup.py:405-411).

Data parallelism: one process per GPU; the only collective is the sum of the gradients.  The backward plan carries
`bucket ready` markers (engine.Plan.bucket_ready): the step is cut into hipGraph segments at those markers, and after each
segment the bucket's all-reduce is issued asynchronously on RCCL's own stream, so that it runs beside the next segment of the
backward (the seg-net's gradients do not depend on the box / T-Net gradients: semisup_models.py:150-151 blocks that gradient).
The optimiser plan carries `bucket wait` markers in front of each bucket's Adam launch.  With one rank the markers are ignored and
the whole step is ONE graph.
"""
import os

import torch

from . import abi
from .engine import Plan


OPTIMIZER_CALLS = ('t3d_adam_tf_step', 't3d_momentum_step')


class TrainStep:
    def __init__(self, engine, pre, fwd, bwd=None, opt=None, process_group=None, use_hip_graph=None, force_dist=False,
                 one_graph=None):
        """engine: nets.Graph; pre/fwd/bwd/opt: engine.Plan (bwd/opt None for a forward-only step).
        process_group: torch.distributed group (None: single replica).  force_dist: take the multi-rank code path (segments,
        collectives) even with one rank -- how a 1-GPU box exercises it."""
        self.e, self.rt = engine, engine.rt
        self.pre, self.fwd, self.bwd, self.opt = pre, fwd, bwd, opt
        self.train = bwd is not None
        self.pg = process_group
        self.world = process_group.size() if process_group is not None else 1
        self.dist = self.train and process_group is not None and (self.world > 1 or force_dist)
        self.on_gpu = self.rt.device.type == 'cuda'
        want = use_hip_graph if use_hip_graph is not None else self.on_gpu
        self.want_graph = bool(want) and self.on_gpu
        # Data parallel on RCCL: the whole step, the gradient all-reduce included, is ONE captured graph (RCCL collectives can be
        # stream-captured; nothing is host-issued between the backward and the optimiser, and a replay costs what the single
        # replica's does).  gloo cannot be captured and keeps the host-issued program, as does T3D_DP_ONE_GRAPH=0.  With ONE bucket
        # the collective is captured in line on the launch stream (there is nothing to run beside it); with several, on RCCL's own
        # stream, which joins the capture through the event edges torch records around a collective (T3D_DP_ONE_GRAPH=1 / 2 force
        # a form).
        # By default only with ONE rank (force_dist: how a 1-GPU box exercises the data-parallel program, and the only place the
        # captured collective has ever run): a capture that succeeds says nothing about a replay that hangs on a second GPU, so with
        # several ranks the host-issued collectives stay the default until the one-graph form has run there (T3D_DP_ONE_GRAPH=1 / 2 or
        # one_graph=True opt in; the ranks then AGREE on the outcome of the capture, see _capture).
        env = os.environ.get('T3D_DP_ONE_GRAPH', '')
        if one_graph is None:
            if env in ('0', '1', '2'):
                one_graph = env != '0'
            else:
                one_graph = self.dist and self.on_gpu and self._backend() == 'nccl' and self.world == 1
        self.one_graph = bool(one_graph) and self.dist and self.on_gpu
        self.one_graph_form = env if env in ('1', '2') else None      # None: by the number of buckets
        self.cache = {}            # generate_masks -> list of program items with captured graphs
        self.n_runs = 0
        self._capture_stream = None
        self.time_waits = False      # bench.py: events around the bucket waits of the host-issued program
        self._wait_events = []
        self._sets = None            # schedule.RiderSets of the scheduled program (device op tables, barrier words)
        self.schedule_report = None
        # A rider barrier that gives up waiting (csrc/rider_dev.h) leaves wrong weights behind: the training loop looks at the
        # sets' time-out words every `rider_check_every` steps (one 4-byte read) and raises schedule.RiderBarrierTimeout; the
        # drivers also call check_riders() before they write a checkpoint.  0: only on request.
        self.rider_check_every = int(os.environ.get('T3D_RIDER_CHECK_EVERY', '64'))

    def _backend(self):
        try:
            import torch.distributed as dist
            return str(dist.get_backend(self.pg)).lower()
        except Exception:      # (no process group, or a stand-in object in a test)
            return ''

    # ---- program --------------------------------------------------------------------------------------------------------------
    def _buckets(self):
        """[[(off, n), ...], ...] -- bucket i of the backward plan's markers; one bucket with every trained range if none."""
        b = list(getattr(self.e, 'buckets', []))
        return b if b else [self.e.default_bucket()]

    def _program(self, generate_masks):
        """[('run', Plan) | ('allreduce', i) | ('wait', i)] in issue order."""
        plans = ([self.pre] if self.pre is not None else []) + [self.fwd] + ([self.bwd, self.opt] if self.train else [])
        calls, lanes, two = [], [], False
        for p in plans:
            for (name, call, arg), lane in zip(p.calls, p.lanes):
                if name == 't3d_dropout_mask' and not generate_masks:
                    continue
                calls.append((name, call, arg))
                lanes.append(lane)
            calls.append((Plan.JOIN, lambda s: 0, None))      # plans are serial with respect to each other
            lanes.append(0)
            two = two or p.two_streams
        if self.dist and not any(c[0] == Plan.BUCKET for c in calls):
            # a backward plan without markers: ONE bucket (every trained range), reduced between the backward and the optimiser
            k = next((i for i, c in enumerate(calls) if c[0] in OPTIMIZER_CALLS), len(calls))
            calls[k:k] = [(Plan.BUCKET, lambda s: 0, 0), (Plan.WAIT, lambda s: 0, 0)]
            lanes[k:k] = [0, 0]
        if not two:       # (data parallel: the flat form -- no bucket marker inside the backward -- takes the scheduled program too)
            calls, lanes = self._overlap(calls, lanes)
            two = two or getattr(self, 'two_stream', False)
        prog, cur = [], Plan(self.rt)
        cur.two_streams = two

        def close():
            nonlocal cur
            if len(cur):
                prog.append(('run', cur))
            cur = Plan(self.rt)
            cur.two_streams = two

        for (name, call, arg), lane in zip(calls, lanes):
            if name in (Plan.BUCKET, Plan.WAIT):
                if self.dist:
                    close()
                    prog.append(('allreduce' if name == Plan.BUCKET else 'wait', arg))
                continue
            cur.calls.append((name, call, arg))
            cur.lanes.append(lane)
        close()
        # The optimiser plan waits for a bucket right before that bucket's Adam launch; as host-issued program that made a graph
        # segment per bucket (≈23 us of graph launch each, and Adam is 7 us in all).  The waits move ahead of the first Adam segment
        # (a wait only adds a dependency earlier) and the Adam launches share ONE segment: [... allreduce, wait, wait, wait, run].
        k = len(prog)
        while k > 0 and (prog[k - 1][0] == 'wait' or (prog[k - 1][0] == 'run' and
                                                      all(c[0] in OPTIMIZER_CALLS + (Plan.JOIN,) for c in prog[k - 1][1].calls))):
            k -= 1
        tail = prog[k:]
        if sum(1 for kind, _ in tail if kind == 'run') > 1:
            merged = Plan(self.rt)
            merged.two_streams = two
            for kind, x in tail:
                if kind == 'run':
                    merged.calls.extend(x.calls)
                    merged.lanes.extend(x.lanes)
            prog = prog[:k] + [t for t in tail if t[0] == 'wait'] + [('run', merged)]
        return prog

    def _overlap(self, calls, lanes):
        """Interleave the two independent chains the plans marked (nets.ModelAssembly: `T_begin` in the forward plan, `S_begin` /
        `S_end` around the segmentation net's backward) -- schedule.overlap_chains.  Single replica, single stream."""
        from . import schedule
        tags = {c[2]: i for i, c in enumerate(calls) if c[0] == Plan.MARK}
        if not all(t in tags for t in ('T_begin', 'S_begin', 'S_end')) or not hasattr(self.rt.lib, 't3d_pointmlp_bwd_r'):
            return calls, lanes
        tb, sb, se = tags['T_begin'], tags['S_begin'], tags['S_end']
        if not tb < sb < se:
            return calls, lanes
        if self._shares_gpu():
            # the rider barrier's liveness rests on all <= 32 rider workgroups of a set being resident together; with another
            # process's kernels on the same chip (several ranks on one GPU) that is no longer this process's to guarantee
            return calls, lanes
        real = lambda cs: [c for c in cs if not c[0].startswith('__')]
        if any(c[0] in (Plan.BUCKET, Plan.WAIT) for c in calls[tb:se]):
            return calls, lanes
        S_calls, T_calls = real(calls[sb:se]), real(calls[tb:sb])
        if self._two_stream_overlap(S_calls + T_calls):
            # Large batches (BASELINE configs[4]: bf16, B=128 N=2048) -- no launch of these chains has a rider form (bf16 kernels; FC
            # ops of 128 rows, finalizers over 2048 tiles cannot ride) but every GEMM launch is several rounds of workgroups, so a
            # small launch on a SECOND queue finds a free slot within a fraction of the GEMM beside it (at B=32 a GEMM launch is one
            # round and the other queue waits the whole launch: DESIGN.md section 5).  Chain S on the side stream between ONE fork
            # and ONE join, chain T on the main stream; in the captured graph the two are parallel branches.  Same kernels, same
            # arguments, disjoint outputs: bit-identical to the serial program.
            self.two_stream = True
            # (which chain is issued first does not matter -- T3D_TWO_STREAM_FIRST=T, and an alternating issue order, eager or
            # captured: 3.25-3.29 ms in every arrangement, docs/EXPERIMENTS.md round 5; the second queue's first kernel starts a few
            # hundred us after the fork either way and the step's length is the sum of both chains' work less a fixed overlap)
            first, second = (T_calls, S_calls) if os.environ.get('T3D_TWO_STREAM_FIRST', 'S') == 'T' else (S_calls, T_calls)
            lanes2 = [0] * tb + [1] * len(first) + [0] + [0] * len(second) + [0]
            merged = first + [(Plan.FLUSH, lambda s: 0, None)] + second + [(Plan.JOIN, lambda s: 0, None)]
            calls = calls[:tb] + merged + calls[se + 1:]
            lanes2 += [0] * (len(calls) - len(lanes2))
            self.schedule_report = {'mode': 'two streams', 'S_launches': len(S_calls), 'T_launches': len(T_calls), 'hosted': 0, 'rider_ops': 0,
                                    'pairs': 0, 'solo': len(S_calls) + len(T_calls)}
            return calls, lanes2
        if self._sets is None:
            self._sets = schedule.RiderSets(self.rt)
        merged, self.schedule_report = schedule.overlap_chains(self.rt, S_calls, T_calls, self._sets)
        # (the launches outside the two chains keep their lanes; inside, the scheduled order is a single-stream program)
        return calls[:tb] + merged + calls[se + 1:], lanes[:tb] + [0] * len(merged) + lanes[se + 1:]

    def _two_stream_overlap(self, chain_calls):
        """T3D_OVERLAP_STREAMS: 1 = the two chains on two streams, 0 = riders (the scheduled single stream); default: two streams for
        large batches.  Measured, same box: B=32 N=1024 fp32 1.217 ms with riders vs 1.248 ms on two streams; B=128 N=2048 bf16
        3.49 ms serial vs 3.26 ms on two streams."""
        mode = os.environ.get('T3D_OVERLAP_STREAMS', '')
        if mode in ('0', '1'):
            return mode == '1' and self.on_gpu
        if not self.on_gpu:
            return False
        # from B x N = 131072 rows on the FC ops (more than 32 rows) and the finalizers (more than 512 row tiles) cannot ride whatever the
        # arithmetic, and the GEMM launches are four and more rounds deep
        return self.e.M >= 4 * 32768

    def _shares_gpu(self):
        """More ranks on this node than GPUs (two ranks on one GPU over gloo: tests, bench.py --gpus 2 on a one-GPU box).
        T3D_RIDERS_SHARED_GPU=1 keeps the riders anyway."""
        if not self.on_gpu or os.environ.get('T3D_RIDERS_SHARED_GPU', '0') == '1':
            return False
        local_ranks = int(os.environ.get('LOCAL_WORLD_SIZE', self.world))      # (one node: SURVEY 8e)
        return local_ranks > torch.cuda.device_count()

    def rider_timeouts(self):
        return self._sets.timeouts() if self._sets is not None else 0

    def check_riders(self):
        """Raises schedule.RiderBarrierTimeout if a rider barrier of this step's program ever timed out (drains the stream)."""
        if self._sets is not None:
            self._sets.check()

    # ---- execution ------------------------------------------------------------------------------------------------------------
    def _allreduce(self, i, async_op):
        import torch.distributed as dist
        g = self.e.vars.grads
        works = []
        for off, n in self._buckets()[i]:
            works.append(dist.all_reduce(g[off:off + n], group=self.pg, async_op=async_op))
        return works

    def _run_program(self, prog, graphs=None):
        pending = {}
        for k, (kind, x) in enumerate(prog):
            if kind == 'run':
                if graphs is not None and graphs.get(k) is not None:
                    graphs[k].replay()
                else:
                    x.run()
            elif kind == 'allreduce':
                pending[x] = self._allreduce(x, async_op=self.on_gpu)
            else:
                ev = None
                if self.time_waits and self.on_gpu:      # how long the launch stream stands still for this bucket (exposed all-reduce)
                    ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                    ev[0].record()
                for w in pending.pop(x, []):
                    if w is not None:
                        w.wait()           # GPU: the current stream waits for the collective; the host does not block
                if ev is not None:
                    ev[1].record()
                    self._wait_events.append(ev)
        for ws in pending.values():
            for w in ws:
                if w is not None:
                    w.wait()

    def _capture(self, prog):
        torch.cuda.synchronize()
        if self._capture_stream is None:
            self._capture_stream = torch.cuda.Stream()
        s = self._capture_stream
        graphs = {}
        if self.dist and self.one_graph:
            # ONE graph per step with the collectives captured inside (RCCL supports stream capture).  T3D_DP_ONE_GRAPH=1: in line
            # on the capture stream (no overlap); =2: asynchronously on RCCL's stream, which joins the capture through the event
            # edges torch records around a collective, so that the replayed graph has the bucket's all-reduce on a branch beside the
            # rest of the backward (no host work and no graph-launch gap between the segments).
            overlap = (self.one_graph_form == '2') if self.one_graph_form else len(self._buckets()) > 1
            g, err = None, None
            try:
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=s, capture_error_mode='thread_local'):
                    if overlap:
                        self._run_program(prog)
                    else:
                        for kind, x in prog:
                            if kind == 'run':
                                x.run()
                            elif kind == 'allreduce':
                                self._allreduce(x, async_op=False)
            except Exception as e:      # a collective this RCCL build cannot capture: the host-issued program still works
                err = e
                torch.cuda.synchronize()
            # The ranks agree on the outcome: one rank replaying a graph while another issues its collectives from the host would
            # pair different operations (a hang at best).  One host-issued all-reduce (MIN) of a success flag; every rank falls back
            # if any rank's capture failed.
            ok = err is None
            if self.world > 1:
                import torch.distributed as dist
                flag = torch.tensor([1 if ok else 0], device=self.rt.device, dtype=torch.int32)
                dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=self.pg)
                ok = bool(int(flag.item()))
            if ok:
                return [('run_graph', g)], {0: g}
            import sys
            why = '%s: %s' % (type(err).__name__, str(err)[:200]) if err is not None else 'the capture failed on another rank'
            sys.stderr.write('TrainStep: capturing the data-parallel step into one graph failed (%s); every rank falls back to the '
                             'host-issued collectives between graph segments\n' % why)
            self.one_graph = False
            self.one_graph_fallback = why[:200]
        # Only the FIRST segment (schedules, forward, the backward up to the first bucket: ~1.1 of the step's 1.6 ms) is replayed as
        # a graph; the segments behind a collective are launched kernel by kernel.  A graph launch costs ~30 us on the GPU before
        # its first kernel starts (rocprofv3 timeline of the one-rank RCCL step, tools/dp_timeline.py: 30 + 29 + 39 us at the three
        # later segment starts), while the host, a millisecond ahead of the GPU at that point, has the eager launches queued long
        # before they can run.  T3D_DP_GRAPH_ALL=1: every segment as a graph (the earlier behaviour).
        first_only = self.dist and os.environ.get('T3D_DP_GRAPH_ALL', '0') != '1'
        seen = False
        for k, (kind, x) in enumerate(prog):
            if kind == 'run':
                if seen and first_only:
                    continue
                seen = True
                g = torch.cuda.CUDAGraph()
                # thread_local: calls of other threads (the RCCL watchdog) during capture must not invalidate it
                with torch.cuda.graph(g, stream=s, capture_error_mode='thread_local'):
                    x.run()
                graphs[k] = g
        return prog, graphs

    def run(self, generate_masks=True):
        """One step.  The first run of a variant is eager (it also loads the code objects); the second captures (capturing
        executes nothing) and replays; later runs replay."""
        self.n_runs += 1
        self._issue(generate_masks)
        if self._sets is not None and self.rider_check_every > 0 and self.n_runs % self.rider_check_every == 0:
            self.check_riders()

    def _issue(self, generate_masks):
        key = bool(generate_masks)
        ent = self.cache.get(key)
        if ent is None:
            ent = self.cache[key] = {'prog': self._program(key), 'graphs': None, 'runs': 0}
        ent['runs'] += 1
        if not self.want_graph or ent['runs'] == 1:
            return self._run_program(ent['prog'])
        if ent['graphs'] is None:
            prog, graphs = self._capture(ent['prog'])
            ent['cprog'], ent['graphs'] = prog, graphs
            if self._sets is not None:
                self._sets.timeouts()      # (loads the two torch kernels of the check now, not in the middle of a timed loop: ~100 ms once)
        if ent['cprog'] and ent['cprog'][0][0] == 'run_graph':
            ent['graphs'][0].replay()
            return
        self._run_program(ent['cprog'], ent['graphs'])

    def dp_report(self):
        """How the data-parallel step is issued, and -- from the events `time_waits` put around the bucket waits -- the time per step
        the launch stream stood still waiting for an all-reduce."""
        mode = 'flat' if len(self._buckets()) == 1 else 'bucketed'
        if self.one_graph:
            mode += ', the step and its collectives captured in ONE graph'
        else:
            mode += ', host-issued collectives between graph segments'
        exposed = None
        if self._wait_events:
            torch.cuda.synchronize()
            n_wait = max(1, sum(1 for kind, _ in self.cache[True]['prog'] if kind == 'wait'))
            steps = max(1, len(self._wait_events) // n_wait)
            exposed = sum(a.elapsed_time(b) for a, b in self._wait_events) * 1e3 / steps
        if getattr(self, 'one_graph_fallback', None):
            mode += ' (one-graph capture failed: %s)' % self.one_graph_fallback
        return {'mode': mode, 'world': self.world, 'gradient_buckets': len(self._buckets()),
                'bucket_floats': [sum(n for _, n in b) for b in self._buckets()],
                'exposed_allreduce_us_per_step': exposed}

    def n_launches(self):
        return sum(len(p) for p in ([self.pre] if self.pre is not None else []) + [self.fwd] + ([self.bwd, self.opt] if self.train else []))

    def n_graph_segments(self, generate_masks=True):
        ent = self.cache.get(bool(generate_masks))
        if ent is None or ent['graphs'] is None:
            return 0
        return len(ent['graphs'])


WORKLOADS = ('A', 'boxpc', 'F')


def workload_flags(workload):
    """The recipe flags of the three training stages (README.md:58-99 of the reference): a = SEMI_MODEL A, b = Box-PC Fit net,
    c = SEMI_MODEL F."""
    from .config import make_parser
    if workload == 'A':
        c = make_parser().parse_special_args(['--SEMI_MODEL', 'A', '--WEAK_WEIGHT_REPROJECTION', '0', '--WEAK_WEIGHT_SURFACE', '0'])
        # the program train_semisup.py runs by default: the recipe's graph, zero-weight weak losses not evaluated (its
        # --weak_loss_summaries adds them for the reference's Weak_Loss/... summaries: two small launches, +1.9 % of the step, loss and
        # gradients bit-unchanged; T3D_WEAK_SUMMARIES=1 times and checks THAT program here)
        import os
        c.WEAK_LOSS_SUMMARIES = os.environ.get('T3D_WEAK_SUMMARIES', '0') == '1'
        return c
    if workload == 'boxpc':
        return make_parser().parse_special_args(['--BOX_PC_MASK_REPRESENTATION', 'A', '--BOXPC_WEIGHT_DELTA', '4'])
    return make_parser().parse_special_args(
        ['--SEMI_MODEL', 'F', '--BOX_PC_MASK_REPRESENTATION', 'A', '--WEAK_WEIGHT_INTRACLASSVAR', '2', '--WEAK_WEIGHT_REPROJECTION', '0',
         '--SEMI_MULTIPLIER_FOR_WEAK_LOSS', '0.05', '--SEMI_BOXPC_FIT_ONLY_ON_2D_CLS', '1', '--SEMI_WEIGHT_BOXPC_FIT_LOSS', '1'])


STAGE_C_TRAIN_CLASSES = [i in (1, 2, 6, 7, 8) for i in range(10)]     # SUNRGBD_SEMI_TEST_CLS of recipe c


def build_training_step(rt, workload, B, N, C, world=1, rank=0, process_group=None, force_dist=False, flat_allreduce=False,
                        use_hip_graph=None, inline_dropout=True, dropout_seed=1234, seed=0, state_dict=None, c=None, dtype='f32',
                        one_graph=None):
    """The step bench.py times and the trajectory tests check: graph + model of `workload` ('A' = BASELINE configs[1],
    'boxpc' = configs[2], 'F' = configs[3]), the device-side schedules (train_semisup.py:127-145), forward, backward, TF-form Adam
    over the recipe's var_list, wrapped in a TrainStep.  Returns (engine graph, model, step, loss buffer)."""
    from .nets import BoxPCModel, Graph, SemiModelA, SemiModelF, make_schedule
    assert workload in WORKLOADS
    g = Graph(B, N, C, rt=rt, seed=seed, dtype=dtype)           # identical initial weights on every rank
    g.inline_dropout, g.dropout_seed = inline_dropout, dropout_seed + rank
    c = c if c is not None else workload_flags(workload)
    prefixes = None
    if workload == 'A':
        model = SemiModelA(g, c)
        loss = model.loss_op.loss
    elif workload == 'boxpc':
        model = BoxPCModel(g, c, False)
        loss = model.loss_op.loss
    else:
        model = SemiModelF(g, c, use_one_hot=True, train_classes=STAGE_C_TRAIN_CLASSES)
        loss = model.loss
        prefixes = list(SemiModelF.VAR_LIST)
    if state_dict is not None:
        g.vars.load_state_dict(state_dict)
    dist_on = process_group is not None and (world > 1 or force_dist)
    g.dp_buckets = dist_on and not flat_allreduce
    g.emit_schedule(g.pre, make_schedule(B * world))
    g.emit_dropout_masks(g.pre, seed=dropout_seed + rank)
    if workload == 'F':
        model.emit_forward(g.fwd, True, True, train=True)
    else:
        model.emit_forward(g.fwd, True, True)
    model.emit_backward(g.bwd)
    g.emit_adam(g.opt, prefixes=prefixes, grad_scale=1.0 / world)
    g.finalize()
    step = TrainStep(g, g.pre, g.fwd, g.bwd, g.opt, process_group=process_group, use_hip_graph=use_hip_graph,
                     force_dist=force_dist, one_graph=one_graph)
    return g, model, step, loss


# ------------------------------------------------------------------------------------------------------------------------------------
# software-pipelined training step (single replica, SEMI_MODEL A)
# ------------------------------------------------------------------------------------------------------------------------------------
class _Program:
    """A launch list that is run eagerly once, captured into a hipGraph on its second run, replayed afterwards."""

    def __init__(self, rt, calls, want_graph):
        self.plan = Plan(rt)
        self.plan.calls, self.plan.lanes = list(calls), [0] * len(calls)
        self.want_graph, self.runs, self.graph = want_graph, 0, None

    def run(self, stream_holder):
        self.runs += 1
        if not self.want_graph or self.runs == 1:
            return self.plan.run()
        if self.graph is None:
            torch.cuda.synchronize()
            if stream_holder[0] is None:
                stream_holder[0] = torch.cuda.Stream()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=stream_holder[0], capture_error_mode='thread_local'):
                self.plan.run()
            self.graph = g
        self.graph.replay()


class PipelinedStep:
    """SEMI_MODEL A with the forward of step k+1 running beside the rest of step k.

    The segmentation net's forward of step k+1 needs only the seg net's weights after step k -- i.e. the seg BACKWARD of step k and
    its Adam update -- not the T-Net / box net chain of step k (forward, loss, backward, their Adam), which is the longer of the two
    chains of a step (718 vs 454 us alone at B=32 N=1024; schedule.py).  So the steady-state unit of work is not
    [forward | backward] but two chains of equal standing,

        SF_k = seg backward of step k -> slab reduction + Adam of the seg net -> schedules + seg forward of step k+1
        T_k  = T-Net / box forward, loss, backward of step k -> slab reduction + Adam of T-Net / box net

    aligned launch by launch by schedule.overlap_chains like the two chains of a single step: the ten finalizers, the global-feature
    FC and the head of the seg forward (80 us with nothing to hide behind in a one-step schedule) ride in T's GEMM launches, and T's
    small launches find ten more GEMM launches to ride in.  Every variable still sees its updates in the sequential order (seg:
    S_k, Adam, F_k+1; T-Net / box: T_k, Adam, T_k+1), so weights, Adam moments, losses and moving statistics are bit-identical to the
    one-step-at-a-time program (tests/test_pipeline_gpu.py, tests/test_pipeline_cpu.py).

    Two CONTEXTS (engine graphs with their own activations, inputs and schedule counters, one shared variable store) alternate:
    step k runs in context k % 2, its forward was started by the previous call.  Protocol: the inputs of step k+1 must be loaded
    (`inputs(k + 1).load(batch)`) before `run()` of step k; `run(last=True)` finishes step k without starting a forward (state equal
    to the sequential program after k+1 steps, moving statistics included)."""

    def __init__(self, rt, ctxs, use_hip_graph=None):
        from . import schedule
        self.rt, self.ctxs = rt, ctxs
        self.on_gpu = rt.device.type == 'cuda'
        self.want_graph = (bool(use_hip_graph) if use_hip_graph is not None else self.on_gpu) and self.on_gpu
        self.dist, self.world, self.time_waits = False, 1, False
        self._sets = schedule.RiderSets(rt)
        self.rider_check_every = int(os.environ.get('T3D_RIDER_CHECK_EVERY', '64'))
        self._stream = [None]
        self.n_runs = 0
        self.fwd_pending = False
        ch = [self._chains(c) for c in ctxs]
        mk = lambda calls: _Program(rt, calls, self.want_graph)
        self.head, self.steady, self.tail, self.reports = [], [], [], []
        for i in (0, 1):
            me, nxt = ch[i], ch[1 - i]
            self.head.append(mk(me['pre'] + me['F']))
            t_chain = me['Tf'] + me['Tb'] + me['adam_T']
            calls, rep = schedule.overlap_chains(rt, me['S'] + me['reduce_S'] + me['adam_S'] + nxt['pre'] + nxt['F'], t_chain, self._sets)
            self.steady.append(mk(calls))
            self.reports.append(rep)
            calls, _ = schedule.overlap_chains(rt, me['S'] + me['reduce_S'] + me['adam_S'], t_chain, self._sets)
            self.tail.append(mk(calls))
        self.schedule_report = self.reports[0]

    @staticmethod
    def _chains(ctx):
        g = ctx['g']
        real = lambda cs: [c for c in cs if not c[0].startswith('__')]

        def split(calls, tag):
            k = next(i for i, c in enumerate(calls) if c[0] == Plan.MARK and c[2] == tag)
            return calls[:k], calls[k + 1:]
        F, Tf = split(g.fwd.calls, 'T_begin')
        Tb, rest = split(g.bwd.calls, 'S_begin')
        S, after = split(rest, 'S_end')
        return dict(pre=real(g.pre.calls), F=real(F), Tf=real(Tf), Tb=real(Tb), S=real(S), reduce_S=real(after),
                    adam_T=real(ctx['opt_T'].calls), adam_S=real(ctx['opt_S'].calls))

    def inputs(self, k=None):
        """nets.Inputs of the context step k runs in (default: the step the next run() call completes)."""
        return self.ctxs[(self.n_runs if k is None else k) % 2]['model'].inputs

    def loss(self, k):
        return self.ctxs[k % 2]['loss']

    def run(self, last=False):
        """Completes step k = n_runs (its forward runs first if no earlier call started it) and, unless `last`, starts the forward
        of step k+1 beside it."""
        i = self.n_runs % 2
        if not self.fwd_pending:
            self.head[i].run(self._stream)
        (self.tail if last else self.steady)[i].run(self._stream)
        self.fwd_pending = not last
        self.n_runs += 1
        if self.rider_check_every > 0 and self.n_runs % self.rider_check_every == 0:
            self.check_riders()

    # ---- what bench.py asks of a step object -------------------------------------------------------------------------------
    def profile_plans(self):
        return [self.steady[0].plan]

    def n_launches(self):
        return len(self.steady[0].plan)

    def n_graph_segments(self, generate_masks=True):
        return 1 if self.steady[0].graph is not None else 0

    def rider_timeouts(self):
        return self._sets.timeouts()

    def check_riders(self):
        self._sets.check()


def build_pipelined_step(rt, B, N, C, use_hip_graph=None, inline_dropout=True, dropout_seed=1234, seed=0, state_dict=None, c=None):
    """Two contexts of SEMI_MODEL A over one variable store + the PipelinedStep that alternates them.  Returns (step, contexts):
    context = dict(g, model, loss, ...); `contexts[0]['g'].vars` is the shared store."""
    from .nets import Graph, SemiModelA, make_schedule
    c = c if c is not None else workload_flags('A')
    ctxs, vs = [], None
    for i in (0, 1):
        g = Graph(B, N, C, rt=rt, seed=seed, vars=vs)
        vs = g.vars
        vs.x3_frag_enabled = False      # (the seg forward of step k+1 starts before the T-Net / box optimiser of step k: no single refresh point)
        g.inline_dropout, g.dropout_seed = inline_dropout, dropout_seed
        g.pool_dz = False               # (two contexts in flight: every layer keeps its own gradient tensor)
        g.split_opt = True
        model = SemiModelA(g, c)
        if i == 0 and state_dict is not None:
            g.vars.load_state_dict(state_dict)
        sched = make_schedule(B)
        sched.step_offset = 1                       # this context runs every other step: step = its counter + 1 ...
        g.hyper[0] = float(i - 1)                   # ... starting at step i
        g.emit_schedule(g.pre, sched)
        g.emit_dropout_masks(g.pre, seed=dropout_seed)
        model.emit_forward(g.fwd, True, True)
        model.emit_backward(g.bwd)
        assert any(cl[0] == Plan.MARK and cl[2] == 'S_begin' for cl in g.bwd.calls), 'the pipelined step needs the two-chain backward'
        opt_T, opt_S = Plan(rt), Plan(rt)
        g.emit_adam(opt_T, prefixes=[model.tnet.scope + '/', model.box.scope + '/'])
        g.emit_adam(opt_S, prefixes=[model.seg.scope + '/'])
        g.trained_prefixes = None
        g.finalize()
        ctxs.append(dict(g=g, model=model, loss=model.loss_op.loss, opt_T=opt_T, opt_S=opt_S))
    n_tr = sum(n for _, n in vs.trainable_ranges(None))
    n_cov = sum(n for pre in ([ctxs[0]['model'].tnet.scope + '/', ctxs[0]['model'].box.scope + '/'], [ctxs[0]['model'].seg.scope + '/'])
                for _, n in vs.trainable_ranges(pre))
    assert n_tr == n_cov, 'the two optimiser launches must cover every trainable variable'
    return PipelinedStep(rt, ctxs, use_hip_graph=use_hip_graph), ctxs
