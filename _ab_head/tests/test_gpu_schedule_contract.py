"""-m gpu: the fused schedule as an explicit contract (round 6).

  * the form of a persistent scan launch is an ARGUMENT of the launch (mgr_scan_launch_opts.form), the launch reports its own
    launch number (seq_out), and a residency wait that has to be enqueued before its launch gets the number through a page-locked
    word (mgr_stream_wait_resident_word); a wait that runs into its bound is counted (mgr_resident_wait_stats)
  * a validation pass (predict_stream) and a SECOND Engine on the same Device between pipelined training steps change nothing:
    the same losses and weights as the plain schedule, and no wait at its bound
  * the drop paths of the two-calls-ahead schedule (a predict between announced steps; the caller comes back with another batch)
    give the plain schedule's losses and weights
  * a guest of the shape of a collective starts promptly beside the FUSED layout (208 eight-wave + 56 four-wave workgroups resident)

Reference call sites: multimodal_fusion/multimodal.py:264-269 (fit_generator with a validation generator: validation passes
between training steps), SURVEY 8e (the RCCL all-reduce kernel is the guest).
"""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _wait_stats(dev):
    out = (C.c_uint * 4)()
    dev.call("mgr_resident_wait_stats", out)
    return int(out[0]), int(out[1])


def _persist(dev):
    nl, ns = C.c_int(), C.c_int()
    dev.call("mgr_persist_stats", C.byref(nl), C.byref(ns))
    return nl.value, ns.value


def _scan_setup(device, B, T, H, seed=0):
    """A random recurrence on the device: (binding module, Z [B, T, 4H] in the packed gate layout, packed U)."""
    from mgr_amd import _capi
    rng = np.random.default_rng(seed)
    Zp = device.array((rng.standard_normal((B, T, 4 * H)) * 0.5).astype(np.float32))
    U = device.array((rng.standard_normal((H, 4 * H)) * 0.1 / np.sqrt(H)).astype(np.float32))
    Up = device.empty((H, 4 * H))
    device.call("mgr_lstm_pack", U, Up, H, H, 0)
    device.sync()
    U.free()
    return _capi, Zp, Up


def test_launch_reports_its_number_and_the_word_hands_it_to_an_earlier_wait(device):
    """seq_out: the number a persistent launch got; MGR_SEQ_NONE for a call that enters no launch into the ledger.  A wait enqueued
    BEFORE its launch (another stream) is released by the number arriving through the page-locked word; a word that is never filled
    costs the wait its bound and is counted; MGR_SEQ_NONE releases at once."""
    B, T, H = 32, 64, 300
    _capi, Zp, Up = _scan_setup(device, B, T, H)
    Y = device.zeros((B, T, H))
    jobs = [dict(Z=Zp, Up=Up, Y=Y, ldy=H, B=B, T=T, H=H, reverse=0)]
    arr = _capi.make_scan_jobs(jobs)
    ws = device.bytes(device.lib.mgr_lstm_scan_multi_ws_bytes(1, arr))
    words = device.pinned((4,), np.uint32)
    words[...] = 0
    device.sync()
    n0, _ = _persist(device)
    w0, b0 = _wait_stats(device)

    # the wait first (stream 3), then the launch (stream 0) with seq_out = the word the wait polls
    device.stream(3)
    device.call("mgr_stream_wait_resident_word", int(words.ctypes.data), 50000)
    device.record(20)
    device.stream(0)
    opts = _capi.make_launch_opts(_capi.SCAN_FORM_PLAIN, words.ctypes.data)
    _capi.check(device.lib.mgr_lstm_scan_fwd_multi_ex(device.ctx, 1, arr, ws.ptr, ws.nbytes, C.byref(opts)))
    assert int(words[0]) == n0 + 1 == _persist(device)[0]
    device.sync()
    w1, b1 = _wait_stats(device)
    assert (w1 - w0, b1 - b0) == (1, 0)          # released by the launch's residency, not by its 50 ms bound

    # the known-number form for a launch that is already enqueued / finished
    device.call("mgr_stream_wait_resident", C.c_uint(int(words[0])), 50000)
    device.sync()
    assert _wait_stats(device) == (w1 + 1, b1)

    # a word nobody fills: the wait ends by its bound and says so
    words[1] = 0
    device.call("mgr_stream_wait_resident_word", int(words.ctypes.data + 4), 300)
    device.sync()
    assert _wait_stats(device) == (w1 + 2, b1 + 1)
    # MGR_SEQ_NONE: no launch to wait for
    words[2] = _capi.SEQ_NONE
    device.call("mgr_stream_wait_resident_word", int(words.ctypes.data + 8), 50000)
    device.sync()
    assert _wait_stats(device) == (w1 + 3, b1 + 1)

    # a call that enqueues no ledger launch (H = 20 has no multi-CU instantiation: fallback kernels) reports MGR_SEQ_NONE
    Hs = 20
    _, Zs, Us = _scan_setup(device, 4, 8, Hs, seed=1)
    Ys = device.zeros((4, 8, Hs))
    arr_s = _capi.make_scan_jobs([dict(Z=Zs, Up=Us, Y=Ys, ldy=Hs, B=4, T=8, H=Hs, reverse=0)])
    ws_s = device.bytes(device.lib.mgr_lstm_scan_multi_ws_bytes(1, arr_s))
    seq = C.c_uint(7)
    opts = _capi.make_launch_opts(0, C.addressof(seq))
    n1, _ = _persist(device)
    _capi.check(device.lib.mgr_lstm_scan_fwd_multi_ex(device.ctx, 1, arr_s, ws_s.ptr, ws_s.nbytes, C.byref(opts)))
    assert (seq.value == _capi.SEQ_NONE) == (_persist(device)[0] == n1)
    device.sync()
    for a in (Zp, Up, Y, ws, Zs, Us, Ys, ws_s):
        a.free()


def test_struct_size_guards_the_options(device):
    """mgr_scan_launch_opts is read through ITS OWN struct_size: a caller built against a shorter struct gets the defaults for the
    members it does not have, an unknown form is refused, and the library reports the struct sizes it was built with."""
    from mgr_amd import _capi
    sizes = (C.c_uint * 4)()
    assert device.lib.mgr_abi_struct_sizes(sizes) == 0
    assert tuple(sizes) == (C.sizeof(_capi.ScanJob), C.sizeof(_capi.ScanBwdJob), C.sizeof(_capi.ScanLaunchOpts), _capi.ABI_REVISION)
    B, T, H = 16, 16, 128
    _, Zp, Up = _scan_setup(device, B, T, H, seed=2)
    Y = device.zeros((B, T, H))
    arr = _capi.make_scan_jobs([dict(Z=Zp, Up=Up, Y=Y, ldy=H, B=B, T=T, H=H, reverse=0)])
    ws = device.bytes(device.lib.mgr_lstm_scan_multi_ws_bytes(1, arr))
    opts = _capi.make_launch_opts(99, 0)
    assert device.lib.mgr_lstm_scan_fwd_multi_ex(device.ctx, 1, arr, ws.ptr, ws.nbytes, C.byref(opts)) != 0
    assert b"unknown scan form" in device.lib.mgr_last_error()
    opts.struct_size = 4          # (a header that knows struct_size only: form 99 is not read, the call takes the defaults)
    _capi.check(device.lib.mgr_lstm_scan_fwd_multi_ex(device.ctx, 1, arr, ws.ptr, ws.nbytes, C.byref(opts)))
    device.sync()
    for a in (Zp, Up, Y, ws):
        a.free()


def _fusion_run(device, B, T, steps, sched, between=None, seed=5):
    """`steps` device-RNG training steps of the reference-size fusion network announced two calls ahead (what bench.py and
    fit_generator do); between(eng, i) runs after step i's loss was read."""
    from mgr_amd.configs import baseline_config
    from mgr_amd.engine import Engine, Schedule
    from mgr_amd.synthetic import synthetic_arrays, synthetic_weights
    spec, _, _, Lmax = baseline_config("F")
    eng = Engine(spec, B, T, Lmax, device=device, seed=seed, schedule=Schedule(**sched))
    eng.set_weights(synthetic_weights(spec, 3))
    xs, labels, il, ll = synthetic_arrays(spec, B, T, Lmax, 11)
    eng._upload_inputs(xs, None, True)
    eng._upload_labels(labels, il, ll)
    pipe = sched.get("pipeline", True)
    out = []
    for i in range(steps):
        eng.enqueue_train_step(None, None, None, None, rand=None, apply_update=True, upload=False,
                               prefetch_next=pipe and i < steps - 1, prefetch_after_next=pipe and i < steps - 2)
        out.append(eng.read_loss())
        if between is not None:
            between(eng, i, xs)
    device.sync()
    w = eng.get_weights()
    eng.close()
    return out, w


def test_validation_pass_and_second_engine_between_pipelined_steps(device):
    """Config F's own shape class (B = 64: the encoder launches take the FUSED form) at T = 160.  Between the pipelined training steps:
    a predict_stream validation pass of the same engine (multimodal.py:264-269: fit_generator validates between steps) and a training
    step of a SECOND engine on the same Device (its persistent launches take launch numbers in between - round 5's predicted numbers
    then pointed at the wrong launches and every wait ran to its 2 ms bound).  Same losses and weights as the plain schedule without
    any of it, the validation results identical every time, and no residency wait at its bound."""
    from mgr_amd.configs import baseline_config
    from mgr_amd.engine import Engine
    from mgr_amd.synthetic import synthetic_arrays, synthetic_weights
    B, T, steps = 64, 160, 7
    spec, _, _, Lmax = baseline_config("F")
    other = Engine(spec, 16, 96, Lmax, device=device, seed=99)
    other.set_weights(synthetic_weights(spec, 4))
    oxs, olab, oil, oll = synthetic_arrays(spec, 16, 96, Lmax, 12)
    other_losses = []
    seen = []

    def between(eng, i, xs):
        if i in (2, 4):
            # (a) a validation pass of two batches through the pipelined inference path of the SAME engine
            seen.append([p.copy() for p in eng.predict_stream([xs, xs], output="posteriors")])
        if i in (1, 2, 5):
            # (b) another engine trains on the same context (its own status block, its own persistent launches)
            other_losses.append(other.train_step(oxs, olab, oil, oll))

    device.sync()
    w0, b0 = _wait_stats(device)
    got, wg = _fusion_run(device, B, T, steps, {}, between)
    w1, b1 = _wait_stats(device)
    plain, wp = _fusion_run(device, B, T, steps, dict(pipeline=False))
    assert got == plain
    for k in wg:
        assert np.array_equal(wg[k], wp[k]), k
    assert w1 > w0, "the default schedule enqueued no residency wait at all: the fused form never engaged"
    # (round 5's predicted launch numbers made EVERY wait run into its bound here.  One expired wait is tolerated: a wait also expires
    #  when the HOST is late by more than the bound with the launch it is for - 5 of 5,999 in the 2,000-step soak of profiles/)
    assert b1 - b0 <= 1, "%d of %d residency waits ran into their bound" % (b1 - b0, w1 - w0)
    # the validation passes saw the weights of their moment: two batches of one pass identical, the passes differ (training moved on)
    assert np.array_equal(seen[0][0], seen[0][1]) and np.array_equal(seen[1][0], seen[1][1])
    assert not np.array_equal(seen[0][0], seen[1][0])
    assert len(other_losses) == 3 and all(np.isfinite(other_losses))
    other.close()


def test_drop_paths_of_the_two_ahead_schedule_equal_the_plain_schedule(device):
    """with two batches announced ahead, (a) a predict between two steps discards the prefetched encoder pass,
    (b) the caller comes back with ANOTHER batch than it announced.  Both used to leave the early generator's claim on a FEAT buffer
    in place while the step chose its own: the next batch's deepest scan then overwrote the buffer this step's deferred dW GEMMs were
    reading - silently wrong fusion gradients.  Host batches (upload = True: the identity of the announced arrays is what the engine
    checks), device RNG; the reference is the same batch sequence on the plain schedule."""
    from mgr_amd.configs import baseline_config
    from mgr_amd.engine import Engine, Schedule
    from mgr_amd.synthetic import synthetic_arrays, synthetic_weights
    spec, _, _, Lmax = baseline_config("F")
    B, T = 16, 256
    data = [synthetic_arrays(spec, B, T, Lmax, 100 + k) for k in range(4)]
    order = [0, 1, 2, 3, 0, 2, 1, 3]            # the batches the steps really train on

    def run(pipelined):
        eng = Engine(spec, B, T, Lmax, device=device, seed=21, schedule=Schedule(pipeline=pipelined))
        eng.set_weights(synthetic_weights(spec, 3))
        losses = []
        for i, k in enumerate(order):
            xs, lab, il, ll = data[k]
            nxt = data[order[i + 1]][0] if i + 1 < len(order) else None
            nxt2 = data[order[i + 2]][0] if i + 2 < len(order) else None
            if pipelined and i == 2:
                nxt = data[3][0]          # announces batch 3 ... and step 3 comes with batch 3 (honoured); the batch AFTER it:
                nxt2 = data[1][0]         # announces batch 1, but step 4 really brings batch 0: (b) a different batch than announced
            losses.append(eng.train_step(xs, lab, il, ll, next_inputs=nxt if pipelined else None,
                                         after_next_inputs=nxt2 if pipelined else None))
            if i == 5:
                eng.predict(data[0][0])    # (a) discards what was prefetched for step 6, with an early generator for step 7 pending
        eng.dev.sync()
        w = eng.get_weights()
        g = eng.get_grads()
        eng.close()
        return losses, w, g

    lp, wp, gp = run(False)
    lq, wq, gq = run(True)
    assert lq == lp
    for k in wp:
        assert np.array_equal(wq[k], wp[k]), k
    for k in gp:
        assert np.array_equal(gq[k], gp[k]), k


def test_tune_keys_of_the_caller_survive_a_fused_step(device):
    """the engine used to set tune keys 4, 12 and 16 around its launches and reset them to 0 - a user's or bench.py's
    own `--tune` setting was gone after the first fused step.  The forms are launch arguments now and key 12 is restored."""
    vals = {4: 2, 12: 2, 16: 1}
    for k, v in vals.items():
        device.call("mgr_tune", k, v)
    try:
        _fusion_run(device, 64, 96, 4, {})
        for k, v in vals.items():
            got = C.c_int()
            device.call("mgr_tune_get", k, C.byref(got))
            assert got.value == v, (k, got.value)
    finally:
        for k in vals:
            device.call("mgr_tune", k, 0)


def test_a_collective_shaped_guest_starts_beside_the_fused_layout(device):
    """the RCCL all-reduce is a guest kernel of a few workgroups (8 channels -> 8 workgroups of 256 threads, tens of KiB of
    LDS) that must start while persistent scans hold the chip.  Round 5's test measured it beside 408 four-wave workgroups; the
    shipped layout is 208 eight-wave workgroups (a CU each) + 56 four-wave ones on the 48 CUs they leave.  Here: the fused encoder
    launch of config F (B = 64) on one stream, the fusion layer's scan behind its residency on another, the guest on a third - it
    starts within a fraction of the scans' run time and ends long before them."""
    from mgr_amd import _capi
    B, T = 64, 1200
    bufs, jobs = [], []
    for H in (500, 300):
        for rev in (0, 1):
            _, Zp, Up = _scan_setup(device, B, T, H, seed=H + rev)
            Y = device.zeros((B, T, H))
            bufs += [Zp, Up, Y]
            jobs.append(dict(Z=Zp, Up=Up, Y=Y, ldy=H, B=B, T=T, H=H, reverse=rev))
    fj = []
    for rev in (0, 1):
        _, Zp, Up = _scan_setup(device, B, T, 100, seed=7 + rev)
        Y = device.zeros((B, T, 100))
        bufs += [Zp, Up, Y]
        fj.append(dict(Z=Zp, Up=Up, Y=Y, ldy=100, B=B, T=T, H=100, reverse=rev))
    arr, farr = _capi.make_scan_jobs(jobs), _capi.make_scan_jobs(fj)
    ws = device.bytes(device.lib.mgr_lstm_scan_multi_ws_bytes(len(jobs), arr))
    fws = device.bytes(device.lib.mgr_lstm_scan_multi_ws_bytes(len(fj), farr))
    nblocks = 8
    out = device.zeros((2 + 2 * nblocks,), np.int64)
    word = device.pinned((1,), np.uint32)
    word[0] = 0
    device.sync()
    w0, b0 = _wait_stats(device)
    device.stream(5)
    device.record(21)
    opts = _capi.make_launch_opts(_capi.SCAN_FORM_FUSED, word.ctypes.data)
    _capi.check(device.lib.mgr_lstm_scan_fwd_multi_ex(device.ctx, len(jobs), arr, ws.ptr, ws.nbytes, C.byref(opts)))
    device.record(22)
    device.stream(0)
    device.call("mgr_stream_wait_resident", C.c_uint(int(word[0])), 20000)
    fopts = _capi.make_launch_opts(_capi.SCAN_FORM_PLAIN, 0)
    _capi.check(device.lib.mgr_lstm_scan_fwd_multi_ex(device.ctx, len(fj), farr, fws.ptr, fws.nbytes, C.byref(fopts)))
    device.stream(3)
    device.call("mgr_stream_wait_resident", C.c_uint(int(word[0]) + 1), 20000)    # (the fusion-shaped launch got the next number)
    device.call("mgr_probe_guest", nblocks, 256, 48 * 1024, 100, out)
    device.sync()
    scan_ms = device.elapsed_ms(21, 22)
    t = out.download().astype(np.float64) / 100.0     # us
    start = t[2::2].min() - t[1]                         # first guest block after the marker ended
    end = t[3::2].max() - t[1]
    assert _wait_stats(device)[1] == b0
    assert scan_ms > 1.5                                 # the scans did hold the chip while the guest ran
    assert start < 300.0, "the guest's first workgroup started %.0f us after its marker" % start
    assert end < 0.5 * scan_ms * 1e3, (end, scan_ms)
    device.stream(0)
    for a in bufs + [ws, fws, out]:
        a.free()


def test_stream_priority_is_a_property_of_the_context_stream(device):
    """mgr_stream_set_priority recreates a stream of the context with a dispatch priority (Schedule.chain_stream_priority; measured:
    no gain for config F, off by default).  Work enqueued on the recreated stream runs and orders as before; levels outside
    -1 .. 1 and stream indices outside the context are refused."""
    from mgr_amd import _capi
    a = device.array(np.arange(1024, dtype=np.float32))
    b = device.zeros((1024,))
    for level in (1, -1, 0):
        device.call("mgr_stream_set_priority", 3, level)
        device.stream(3)
        device.call("mgr_add2d", a, 1024, a, 1024, b, 1024, 1, 1024)
        device.record(23)
        device.stream(0)
        device.wait_event(0, 23)
        assert np.array_equal(b.download(), 2 * np.arange(1024, dtype=np.float32))
    assert device.lib.mgr_stream_set_priority(device.ctx, 3, 2) != 0
    assert device.lib.mgr_stream_set_priority(device.ctx, 99, 0) != 0
    device.stream(0)
    a.free()
    b.free()
