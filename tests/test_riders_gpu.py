"""GPU: the scheduled step (transferable3d_amd/schedule.py -- small launches of the T-Net / box chain riding in the segmentation net's
backward GEMM launches and vice versa, csrc/rider_dev.h) against the unscheduled one, bit for bit; a rider set with in-launch
barriers between dependent ops, replayed many times under load, against the stand-alone launches."""
import ctypes as C

import numpy as np
import pytest
import torch

from transferable3d_amd import abi
from transferable3d_amd.abi import fptr

pytestmark = pytest.mark.gpu


def _steps(hip_lib, overlap, B, N, n_steps, workload='A'):
    from transferable3d_amd import nets
    from transferable3d_amd.engine import Runtime
    from transferable3d_amd.step import build_training_step
    from transferable3d_amd.synthetic import make_batch
    keep = nets.OVERLAP
    nets.OVERLAP = overlap
    try:
        g, model, step, loss = build_training_step(Runtime(lib=hip_lib), workload, B, N, 4, seed=4)
        losses = []
        for k in range(n_steps):
            model.inputs.load(make_batch(B, N, 4, seed=21 + k))
            step.run()
            losses.append(float(loss))
        torch.cuda.synchronize()
        state = {k: v.clone() for k, v in (('params', g.vars.params[:g.vars.used]), ('state', g.vars.state[:g.vars.state_used]),
                                          ('m', g.vars.adam_m[:g.vars.used]), ('v', g.vars.adam_v[:g.vars.used]))}
        return losses, state, step
    finally:
        nets.OVERLAP = keep


@pytest.mark.parametrize('workload,B,N', [('A', 32, 1024), ('A', 8, 256), ('F', 32, 1024), ('F', 8, 256), ('boxpc', 32, 1024)])
def test_scheduled_step_is_bit_identical_to_the_unscheduled_step(hip_lib, gemm_arithmetic, workload, B, N):
    """Five steps (one eager, the capture, three hipGraph replays) with a new batch each: losses, weights, moving statistics and Adam
    moments of the scheduled program equal the plain one's bit for bit; riders were really hosted; no barrier ever timed out.
    'A': seg backward || T-Net / box chain; 'F' (stage c): the class-agnostic heads + W_ IoU summary || the refinement branch
    (nets.SemiModelF); 'boxpc' has one chain only -- T3D_OVERLAP must change nothing."""
    l0, s0, st0 = _steps(hip_lib, False, B, N, 5, workload)
    l1, s1, st1 = _steps(hip_lib, True, B, N, 5, workload)
    rep = st1.schedule_report
    assert st0.schedule_report is None
    if workload == 'boxpc':
        assert rep is None
    else:
        assert rep is not None
        assert rep['hosted'] >= (5 if workload == 'A' else 1) and rep['rider_ops'] >= rep['hosted'], rep
        assert st1.rider_timeouts() == 0
        st1.check_riders()
        print('schedule %s: %d launches hosted %d small ops, %d pairs, model %.0f -> %.0f us' %
              (workload, rep['hosted'], rep['rider_ops'], rep['pairs'], rep['serial_us'], rep['scheduled_us']))
    assert l0 == l1, (l0, l1)
    for k in s0:
        assert torch.equal(s0[k], s1[k]), k


def test_training_loop_raises_on_a_rider_barrier_timeout(hip_lib):
    """TrainStep.run reads the sets' time-out words every `rider_check_every` steps and raises (the device writes the word when a
    bounded barrier spin gives up: csrc/rider_dev.h); poisoned by hand here."""
    from transferable3d_amd import schedule
    from transferable3d_amd.engine import Runtime
    from transferable3d_amd.step import build_training_step
    from transferable3d_amd.synthetic import make_batch
    g, model, step, loss = build_training_step(Runtime(lib=hip_lib), 'A', 8, 256, 4, seed=4)
    step.rider_check_every = 3
    for k in range(3):
        model.inputs.load(make_batch(8, 256, 4, seed=k))
        step.run()
    assert step._sets is not None and step.rider_timeouts() == 0
    step._sets.sets[0][2][-2] = 1
    step.run(); step.run()
    with pytest.raises(schedule.RiderBarrierTimeout):
        step.run()


def _fc_chain(dev, B, dims, seed):
    """A dependent chain of FC layers (batch-norm + ReLU) in -> h1 -> h2 ...: argument structs of the stand-alone launches."""
    r = np.random.RandomState(seed)
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a, dtype=np.float32)).to(dev)
    x = t(r.randn(B, dims[0]))
    keep, ops, outs = [x], [], []
    cur = x
    for K, N in zip(dims[:-1], dims[1:]):
        w, bias = t(r.randn(K, N) / np.sqrt(K)), t(0.1 * r.randn(N))
        gamma, beta = t(1 + 0.1 * r.randn(N)), t(0.1 * r.randn(N))
        mm, mv, mean, invstd = t(np.zeros(N)), t(np.ones(N)), t(np.zeros(N)), t(np.zeros(N))
        y, out, decay = t(np.zeros((B, N))), t(np.zeros((B, N))), t([0.9])
        a = abi.FcFwdArgs()
        a.in_, a.ld_in, a.K, a.w, a.bias = fptr(cur), K, K, fptr(w), fptr(bias)
        a.gamma, a.beta, a.moving_mean, a.moving_var, a.decay = fptr(gamma), fptr(beta), fptr(mm), fptr(mv), fptr(decay)
        a.eps, a.is_training, a.unbiased_ema, a.act = 1e-3, 1, 1, abi.ACT_RELU
        a.y, a.mean, a.invstd, a.out, a.ld_out, a.B, a.N = fptr(y), fptr(mean), fptr(invstd), fptr(out), N, B, N
        keep += [w, bias, gamma, beta, mm, mv, mean, invstd, y, out, decay]
        ops.append(a)
        outs.append((out, mm, mv))
        cur = out
    return ops, outs, keep


@pytest.mark.parametrize('host', ['four_waves', 'eight_waves'])
def test_rider_set_with_barriers_equals_the_separate_launches_under_load(hip_lib, host):
    """FC chain 256 -> 512 -> 512 -> 256 -> 64 (every op reads what the previous one wrote, through another workgroup's stores): as ONE
    rider set -- alone (t3d_run_riders) and inside a forward GEMM launch that fills the chip -- 40 times each, against the four
    stand-alone launches.  Outputs and moving statistics bit for bit; the consumer workgroups' L1 holds the previous repetition's
    lines when the next one starts (the stale-read case of cdna_hip_programming.md Guideline 16).
    host 'eight_waves': the 128 x 256-tile x3 forward kernel (512-thread workgroups; k_pointmlp_fwd_w8_r runs the riders on the first
    four waves of theirs, the other four leave before the first barrier)."""
    from transferable3d_amd import schedule
    from transferable3d_amd.engine import Runtime
    dev = torch.device('cuda')
    rt = Runtime(lib=hip_lib)
    B, dims = 32, (256, 512, 512, 256, 64)
    s = rt.stream()

    def fresh(seed):
        return _fc_chain(dev, B, dims, seed)

    # reference: stand-alone launches, one repetition per seed
    def reference(seed, reps):
        ops, outs, keep = fresh(seed)
        for _ in range(reps):
            for a in ops:
                abi.check(hip_lib.t3d_fc_fwd(C.byref(a), s), 'fc_fwd')
        torch.cuda.synchronize()
        return [tuple(t.clone() for t in o) for o in outs]

    # a host GEMM: 32768 x 128 x 128 forward (512 workgroups: exactly the chip's resident slots)
    M, K, N = (32768, 128, 128) if host == 'four_waves' else (32768, 128, 512)
    r = np.random.RandomState(5)
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a, dtype=np.float32)).to(dev)
    xg, wg = t(r.randn(M, K)), t(r.randn(K, N) / 11.0)
    yg, ps, pq = t(np.zeros((M, N))), t(np.zeros((M // 128, N))), t(np.zeros((M // 128, N)))
    fa = abi.PointMlpFwdArgs()
    fa.a = abi.ActSrc(fptr(xg), K, 0, None, None, 0, None, 0, abi.F32)
    fa.w, fa.y, fa.psum, fa.psumsq, fa.M, fa.K, fa.N, fa.rows_per_frustum, fa.dtype = fptr(wg), fptr(yg), fptr(ps), fptr(pq), M, K, N, 1024, abi.F32
    if host == 'eight_waves':
        fa.arith = abi.ARITH_BF16X3
        assert hip_lib.t3d_gemm_arithmetic(fa.arith, abi.F32, K, N, 0) == abi.ARITH_BF16X3
    abi.check(hip_lib.t3d_pointmlp_fwd(C.byref(fa), s), 'fwd')
    torch.cuda.synchronize()
    y_ref = yg.clone()

    for mode in ('alone', 'hosted'):
        reps = 40
        ref = reference(7, reps)
        ops, outs, keep = fresh(7)
        sets = schedule.RiderSets(rt)
        rs = sets.make([('t3d_fc_fwd', a) for a in ops])
        assert rs.n_ops == 4 and 1 <= rs.n_wg <= 32
        yg.zero_()
        for _ in range(reps):
            if mode == 'alone':
                abi.check(hip_lib.t3d_run_riders(C.byref(rs), s), 'run_riders')
            else:
                abi.check(hip_lib.t3d_pointmlp_fwd_r(C.byref(fa), C.byref(rs), s), 'fwd_r')
        torch.cuda.synchronize()
        assert sets.timeouts() == 0
        for (o, mm, mv), (ro, rmm, rmv) in zip(outs, ref):
            assert torch.equal(o, ro) and torch.equal(mm, rmm) and torch.equal(mv, rmv), mode
        if mode == 'hosted':
            assert torch.equal(yg, y_ref)


@pytest.mark.parametrize('dtype,B,N', [('f32', 8, 256), ('bf16', 16, 512)])
def test_two_stream_overlap_is_bit_identical_to_the_serial_step(hip_lib, monkeypatch, dtype, B, N):
    """T3D_OVERLAP_STREAMS=1 (the default of large bf16 batches, where nothing can ride: step.TrainStep._two_stream_overlap): the seg
    net's backward on a second stream beside the T-Net / box chain, one fork and one join, parallel branches of the captured graph.
    Same kernels and arguments, disjoint outputs: losses, weights, moving statistics, Adam moments bit for bit the serial step's."""
    from transferable3d_amd import nets
    from transferable3d_amd.engine import Runtime
    from transferable3d_amd.step import build_training_step
    from transferable3d_amd.synthetic import make_batch

    def run(overlap, streams):
        monkeypatch.setenv('T3D_OVERLAP_STREAMS', streams)
        keep = nets.OVERLAP
        nets.OVERLAP = overlap
        try:
            g, model, step, loss = build_training_step(Runtime(lib=hip_lib), 'A', B, N, 4, seed=4, dtype=dtype)
            losses = []
            for k in range(5):
                model.inputs.load(make_batch(B, N, 4, seed=21 + k))
                step.run()
                losses.append(float(loss))
            torch.cuda.synchronize()
            return losses, [t.clone() for t in (g.vars.params[:g.vars.used], g.vars.state[:g.vars.state_used], g.vars.adam_m[:g.vars.used],
                                                g.vars.adam_v[:g.vars.used])], step
        finally:
            nets.OVERLAP = keep
    l0, s0, st0 = run(False, '0')
    l1, s1, st1 = run(True, '1')
    assert st1.schedule_report is not None and st1.schedule_report.get('mode') == 'two streams' and st1.n_graph_segments() == 1
    assert l0 == l1, (l0, l1)
    for a, b in zip(s0, s1):
        assert torch.equal(a, b)
