"""-m gpu: residency by construction of the persistent multi-CU scans (lstm.hip: mgr_persist_admit, lstm_cluster.h:
mgr_cluster_enter, mgr_stream_wait_next_resident) and the non-finite guard of the K-split scan step."""
import ctypes
import time

import numpy as np
import pytest

from tests.helpers import rel_err

pytestmark = pytest.mark.gpu


def _scan_jobs(dev, rng, B, T, H, ndir=2, scale=0.1):
    """ndir random recurrences (Z, U) on the device + their job structs; returns (jobs, outputs Y, keep-alive list)."""
    jobs, ys, keep = [], [], []
    for d in range(ndir):
        Z = dev.array((rng.standard_normal((B, T, 4 * H)) * 0.5).astype(np.float32))
        U = dev.array((rng.standard_normal((H, 4 * H)) * scale / np.sqrt(H)).astype(np.float32))
        Up = dev.empty((H, 4 * H))
        dev.call("mgr_lstm_pack", U, Up, H, H, 0)
        Y = dev.zeros((B, T, H))
        jobs.append(dict(Z=Z, Up=Up, Y=Y, ldy=H, R=0, ldr=0, gates=0, cs=0, B=B, T=T, H=H, reverse=d & 1))
        ys.append(Y)
        keep += [Z, U, Up]
    return jobs, ys, keep


def _launch(dev, jobs, ws=None):
    from mgr_amd import _capi
    arr = _capi.make_scan_jobs(jobs)
    if ws is None:
        ws = dev.bytes(dev.lib.mgr_lstm_scan_multi_ws_bytes(len(jobs), arr))
    _capi.check(dev.lib.mgr_lstm_scan_fwd_multi(dev.ctx, len(jobs), arr, ws.ptr, ws.nbytes))
    return ws


def _persist_stats(dev):
    nl, ns = ctypes.c_int(), ctypes.c_int()
    dev.call("mgr_persist_stats", ctypes.byref(nl), ctypes.byref(ns))
    return nl.value, ns.value


def test_persistent_launches_that_do_not_fit_together_are_serialised(device):
    """Three H = 500, B = 64 bidirectional scans (256 workgroups each, two per CU = 512 slots) on three streams: two fit the chip
    together, the third must be ordered behind them by the admission ledger instead of dead-locking with them; every launch
    computes what it computes alone."""
    dev = device
    rng = np.random.default_rng(5)
    B, T, H = 64, 300, 500
    sets = [_scan_jobs(dev, rng, B, T, H) for _ in range(3)]
    # reference: one after the other on one stream
    dev.stream(0)
    ref = []
    for jobs, ys, _ in sets:
        _launch(dev, jobs)
        dev.sync()
        ref.append([y.download() for y in ys])
        for y in ys:
            y.zero()
    dev.sync()
    n0, s0 = _persist_stats(dev)
    wss = []
    for i, (jobs, ys, _) in enumerate(sets):
        dev.stream(1 + i)
        wss.append(_launch(dev, jobs))
    dev.stream(0)
    dev.sync()
    n1, s1 = _persist_stats(dev)
    assert n1 - n0 == 3
    assert s1 - s0 >= 1, "768 spinning workgroups were let onto 512 slots"
    st = ctypes.c_uint(7)
    dev.call("mgr_scan_status", ctypes.byref(st))
    assert st.value == 0
    for (jobs, ys, _), r in zip(sets, ref):
        for y, yr in zip(ys, r):
            assert np.array_equal(y.download(), yr)


def test_wait_next_resident_releases_when_the_scan_is_resident_and_never_hangs(device):
    dev = device
    rng = np.random.default_rng(6)
    jobs, ys, keep = _scan_jobs(dev, rng, 64, 200, 300)
    ws = _launch(dev, jobs)      # warm-up (module load, attributes)
    dev.sync()
    # (a) the gate is enqueued BEFORE the scan it waits for (the order engine.py uses); bound 200 ms, must pass in a few
    dev.stream(1)
    t0 = time.perf_counter()
    dev.call("mgr_stream_wait_next_resident", 100000)
    dev.stream(2)
    _launch(dev, jobs, ws)
    dev.stream(0)
    dev.sync()
    assert time.perf_counter() - t0 < 0.05
    # (b) no persistent launch follows: the gate gives up after its bound - a placement aid never blocks a stream for good
    dev.stream(1)
    t0 = time.perf_counter()
    dev.call("mgr_stream_wait_next_resident", 3000)
    dev.stream(0)
    dev.sync()
    dt = time.perf_counter() - t0
    assert 0.002 < dt < 0.05, dt


@pytest.mark.parametrize("variant", [0])
def test_non_finite_hidden_state_propagates_as_nan_instead_of_hanging(device, variant):
    """A NaN recurrent weight in the candidate gate makes c and h NaN at the first step that multiplies it (an Inf weight only
    saturates a hard-sigmoid / tanh gate - finite, like in the reference).  The K-split step keeps non-finite words out of the
    exchange (the epoch parity rides in the mantissa of a finite word): the cell publishes 0, latches NaN into Y of that (sample,
    unit) from that step on and raises MGR_SCAN_NONFINITE - the launch finishes in its normal time, mgr_scan_status does not fail."""
    from mgr_amd import _capi
    dev = device
    rng = np.random.default_rng(7)
    B, T, H = 20, 50, 300
    Z = dev.array((rng.standard_normal((B, T, 4 * H)) * 0.5).astype(np.float32))
    Uh = (rng.standard_normal((H, 4 * H)) * 0.1 / np.sqrt(H)).astype(np.float32)
    Uh[17, 2 * H + 33] = np.nan          # Keras layout: columns [2H, 3H) are the candidate gate c
    Uh[40, 0 * H + 7] = np.inf           # ... and an Inf into an input gate only saturates it
    Up = dev.empty((H, 4 * H))
    dev.call("mgr_lstm_pack", dev.array(Uh), Up, H, H, 0)
    Y = dev.zeros((B, T, H))
    ws = dev.bytes(dev.lib.mgr_lstm_scan_ws_bytes(B, T, H))
    dev.call("mgr_scan_status_clear")
    dev.call("mgr_tune", 0, 3)
    dev.call("mgr_tune", 7, variant)
    try:
        t0 = time.perf_counter()
        dev.call("mgr_lstm_scan_fwd", Z, Up, Y, H, 0, 0, 0, 0, B, T, H, 0, ws, ws.nbytes)
        dev.sync()
        assert time.perf_counter() - t0 < 0.5          # a give-up takes ~1 s
        st = ctypes.c_uint(0)
        dev.call("mgr_scan_status", ctypes.byref(st))  # does not raise
        assert st.value == _capi.SCAN_NONFINITE
        y = Y.download()
        assert not np.isnan(y[:, 0]).any()             # step 0 has no recurrent term
        assert np.isnan(y[:, 1:, 33]).all()            # unit 33 is NaN from the first step that sees h_0
        assert np.isfinite(y[:, :, 7]).all() or np.isnan(y[:, 2:]).any()
    finally:
        dev.call("mgr_tune", 0, 0)
        dev.call("mgr_tune", 7, 0)
        dev.call("mgr_scan_status_clear")
    st = ctypes.c_uint(7)
    dev.call("mgr_scan_status", ctypes.byref(st))
    assert st.value == 0


def test_engine_reports_nan_loss_for_a_diverged_encoder(device):
    from mgr_amd.configs import fusion_spec
    from mgr_amd.engine import Engine
    from mgr_amd.synthetic import synthetic_arrays, synthetic_weights
    spec = fusion_spec()
    B, T, Lmax = 16, 40, 6
    eng = Engine(spec, B, T, Lmax, device=device, seed=1)
    w = synthetic_weights(spec, 3)
    bad = {k: v.copy() for k, v in w.items()}
    bad["the_input_audio/l0/fwd/U"][3, 2 * 500 + 5] = np.nan      # candidate gate of unit 5
    xs, labels, il, ll = synthetic_arrays(spec, B, T, Lmax, 9, lmin=2, lmax=5)
    eng.set_weights(bad)
    loss = eng.train_step(xs, labels, il, ll, apply_update=False)       # no exception, no 1 s stall
    assert np.isnan(loss) and eng.nonfinite_seen
    eng.clear_scan_status()
    eng.set_weights(w)
    loss = eng.train_step(xs, labels, il, ll, apply_update=False)
    assert np.isfinite(loss) and not eng.nonfinite_seen
    eng.close()


def test_scan_speed_cannot_be_halved_by_launch_order(device):
    """Provocation of the placement effect of round 1 (a persistent cluster scan launched while chip-filling GEMM waves of
    another stream were resident got a lopsided CU set and ran at HALF speed for its whole life: 22.7 instead of 11.4 ms).
    The four encoder scans of config F (408 workgroups, two per CU) are timed alone, then launched (b) 300 us after a short
    burst of projection GEMMs became resident on another stream, (c) with that burst released behind
    mgr_stream_wait_next_resident (the order engine.py uses).  The burst ends early, so a well-placed scan is back at full
    speed for most of its life: its time may exceed the solo time by the burst it shared the chip with, never by its own length."""
    from mgr_amd import _capi
    dev = device
    rng = np.random.default_rng(8)
    B, T = 64, 1000
    jobs, keep = [], []
    for H in (500, 300):
        j, _, k = _scan_jobs(dev, rng, B, T, H)
        jobs += j
        keep += k
    arr = _capi.make_scan_jobs(jobs)
    ws = dev.bytes(dev.lib.mgr_lstm_scan_multi_ws_bytes(len(jobs), arr))
    Fg, Hg = 600, 300
    Xg = dev.array(rng.standard_normal((B, T, Fg)).astype(np.float32))
    Wg, bg, Zg = dev.zeros((Fg, 4 * Hg)), dev.zeros((4 * Hg,)), dev.empty((B, T, 4 * Hg))

    def scan():
        _capi.check(dev.lib.mgr_lstm_scan_fwd_multi(dev.ctx, len(jobs), arr, ws.ptr, ws.nbytes))

    def burst():
        dev.call("mgr_lstm_input_proj", Xg, Fg, 0, Wg, bg, Zg, B, T, Fg, Hg)   # (one GEMM: ~0.9 ms against ~2.3 ms of scan)

    def run(order):
        best = 1e9
        for _ in range(3):
            dev.sync()
            if order == "alone":
                dev.stream(1); dev.record(0); scan(); dev.record(1)
            elif order == "gemms_first":
                dev.stream(2); burst()
                dev.stream(1); dev.call("mgr_stream_delay", 300); dev.record(0); scan(); dev.record(1)
            else:   # gated: the burst waits on the device until the scan launched next is resident
                dev.stream(2); dev.call("mgr_stream_wait_next_resident", 5000); burst()
                dev.stream(1); dev.record(0); scan(); dev.record(1)
            dev.stream(0)
            dev.sync()
            best = min(best, dev.elapsed_ms(0, 1))
        return best

    scan(); burst(); dev.sync()      # warm-up
    dev.stream(2); dev.record(2); burst(); dev.record(3); dev.stream(0); dev.sync()
    t_burst = dev.elapsed_ms(2, 3)
    alone, first, gated = run("alone"), run("gemms_first"), run("gated")
    print("scan alone %.2f ms | GEMM burst %.2f ms | GEMMs resident first: scan %.2f ms | gated: scan %.2f ms" % (alone, t_burst, first, gated))
    assert t_burst < 0.6 * alone                      # the burst is short against the scan
    assert first < alone + 1.5 * t_burst + 0.15 * alone, (alone, t_burst, first)
    assert gated < alone + 1.5 * t_burst + 0.15 * alone, (alone, t_burst, gated)
    st = ctypes.c_uint(7)
    dev.call("mgr_scan_status", ctypes.byref(st))
    assert st.value == 0


def test_a_collective_shaped_guest_starts_beside_the_resident_encoder_scans(device):
    """The UNFUSED layout (Schedule.fused_encoder_scans = False, and every launch the fused form does not take): 408 four-wave
    workgroups resident, two per CU on 152 CUs.  The shipped layout of config F - 208 eight-wave workgroups that hold a CU each + the
    fusion layer's recurrences on the 48 CUs they leave - has its own guest test: tests/test_gpu_schedule_contract.py::
    test_a_collective_shaped_guest_starts_beside_the_fused_layout.
    DESIGN 6 / 8: for N > 1 the RCCL all-reduce kernel is one more guest on stream 0 beside the deepest encoder scan of the
    next step.  HostComm replaces that kernel with two copies, so
    the claim that it "needs no ledger entry" had no measurement.  Here a guest of its shape - 8 workgroups x 256 threads, 64 KiB
    of LDS each, busy for ~100 us - is launched on another stream the moment the four encoder scans are resident.  It must START
    within 200 us (it finds room on the CUs that hold a single scan workgroup), not when the scan ends, and the scan must still be
    running when the guest has finished - i.e. the two really shared the chip."""
    from mgr_amd import _capi
    dev = device
    rng = np.random.default_rng(11)
    B, T = 64, 600
    jobs, keep = [], []
    for H in (500, 300):
        j, _, k = _scan_jobs(dev, rng, B, T, H)
        jobs += j
        keep += k
    arr = _capi.make_scan_jobs(jobs)
    ws = dev.bytes(dev.lib.mgr_lstm_scan_multi_ws_bytes(len(jobs), arr))
    NB = 8
    out = dev.zeros((2 + 2 * NB,), np.int64)

    def scan():
        _capi.check(dev.lib.mgr_lstm_scan_fwd_multi(dev.ctx, len(jobs), arr, ws.ptr, ws.nbytes))

    scan()
    dev.call("mgr_probe_guest", NB, 256, 64 * 1024, 100, out)
    dev.sync()                                                   # warm-up: module load, function attributes
    lat, beside = [], []
    for _ in range(3):
        dev.stream(2)
        dev.call("mgr_stream_wait_next_resident", 20000)         # the guest's stream waits until the scan launched next is resident
        dev.call("mgr_probe_guest", NB, 256, 64 * 1024, 100, out)
        dev.record(12)
        dev.stream(1)
        dev.record(10)
        scan()
        dev.record(11)
        dev.stream(0)
        dev.sync()
        t = out.download()
        marker, starts, ends = t[1], t[2::2], t[3::2]
        lat.append((starts.max() - marker) / 100.0)              # us between "the stream got here" and the LAST guest block starting
        scan_ms, guest_done_ms = dev.elapsed_ms(10, 11), dev.elapsed_ms(10, 12)
        beside.append((scan_ms, guest_done_ms, (ends.max() - starts.min()) / 100.0))
    print("guest start latency beside 408 resident scan workgroups: %s us; (scan ms, guest done at ms, guest span us): %s" % (lat, beside))
    assert min(lat) < 200.0, lat
    scan_ms, guest_done_ms, span_us = min(beside, key=lambda b: b[1])
    assert guest_done_ms < 0.6 * scan_ms, beside                 # the guest came and went while the scan was running
    assert span_us < 1000.0, beside                              # ... and was not starved once it ran (~100 us of work)
    st = ctypes.c_uint(7)
    dev.call("mgr_scan_status", ctypes.byref(st))
    assert st.value == 0
