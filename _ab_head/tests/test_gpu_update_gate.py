"""-m gpu: the update gate - a step whose scans gave up (garbage gradients) or met a non-finite hidden state must not reach the
weights, although the host only learns of it with the loss, after the optimizer kernels of the same step were queued
(engine.Engine.apply_gradients; mgr_update_gate_eval / mgr_update_gate_set in include/mgr.h)."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_gated_adam_and_maxnorm_leave_everything_untouched(device):
    """C ABI: with a non-zero flag set, mgr_adam_step / mgr_maxnorm_cols are no-ops on the device and the skip is counted."""
    from mgr_amd import _capi
    dev = device
    rng = np.random.default_rng(0)
    n = 4096
    p0, g0 = rng.standard_normal(n).astype(np.float32), rng.standard_normal(n).astype(np.float32)
    p, g, m, v = dev.array(p0), dev.array(g0), dev.zeros((n,)), dev.zeros((n,))
    W0 = (rng.standard_normal((64, 32)) * 3).astype(np.float32)
    W = dev.array(W0)
    flag = dev.zeros((4,))
    block = dev.zeros((16,), np.uint32)
    dev.call("mgr_scan_status_bind", block)
    try:
        # a clean status evaluates to an open gate
        dev.call("mgr_update_gate_eval", _capi.SCAN_GAVE_UP | _capi.SCAN_NONFINITE, flag)
        assert flag.download()[0] == 0.0
        dev.call("mgr_scan_status_inject", _capi.SCAN_GAVE_UP)
        dev.call("mgr_update_gate_eval", _capi.SCAN_NONFINITE, flag)     # masked out
        assert flag.download()[0] == 0.0
        dev.call("mgr_update_gate_eval", _capi.SCAN_GAVE_UP | _capi.SCAN_NONFINITE, flag)
        assert flag.download()[0] == 1.0
        dev.call("mgr_update_gate_set", flag)
        dev.call("mgr_adam_step", p, g, m, v, n, 1e-3, 0.9, 0.999, 1e-7, 0.5, 1.0)
        dev.call("mgr_maxnorm_cols", W, 64, 32, 1.0, 1e-7)
        assert np.array_equal(p.download(), p0) and not m.download().any() and not v.download().any()
        assert np.array_equal(W.download(), W0)
        st = (C.c_uint * 4)()
        dev.call("mgr_scan_status_ex", st)
        assert st[0] == _capi.SCAN_GAVE_UP and st[2] == 1
        # flag cleared (as the all-reduce of healthy replicas would leave it): the same calls now update
        flag.zero()
        dev.call("mgr_adam_step", p, g, m, v, n, 1e-3, 0.9, 0.999, 1e-7, 0.5, 1.0)
        dev.call("mgr_maxnorm_cols", W, 64, 32, 1.0, 1e-7)
        assert not np.array_equal(p.download(), p0) and m.download().any()
        assert np.linalg.norm(W.download(), axis=0).max() <= 1.0 + 1e-5
        dev.call("mgr_update_gate_set", 0)
        dev.call("mgr_scan_status_clear")
        dev.call("mgr_scan_status_ex", st)
        assert st[0] == 0 and st[2] == 0
    finally:
        dev.call("mgr_update_gate_set", 0)
        dev.call("mgr_scan_status_bind", 0)


def _fusion_engine(device, B=16, T=40, Lmax=6, seed=1):
    from mgr_amd.configs import fusion_spec
    from mgr_amd.engine import Engine
    from mgr_amd.synthetic import synthetic_arrays, synthetic_weights
    spec = fusion_spec()
    eng = Engine(spec, B, T, Lmax, device=device, seed=seed)
    w = synthetic_weights(spec, 3)
    eng.set_weights(w)
    batch = synthetic_arrays(spec, B, T, Lmax, 9, lmin=2, lmax=5)
    return eng, w, batch


@pytest.mark.parametrize("pipelined", [False, True])
def test_a_give_up_at_step_3_of_5_leaves_the_weights_of_step_2(device, pipelined):
    """A scan give-up is injected into the engine's status block on stream 0 right before step 3 is enqueued (what a scan of that
    step would do).  The optimizer kernels of step 3 are queued long before the host reads the loss - the gate must have kept
    them away from the weights: weights == weights after step 2, the error names step 3, and after clear_scan_status() training
    continues from exactly there."""
    from mgr_amd import _capi
    eng, w, (xs, labels, il, ll) = _fusion_engine(device)
    kw = dict(next_inputs=xs) if pipelined else {}
    for _ in range(3):                                    # steps 0, 1, 2
        assert np.isfinite(eng.train_step(xs, labels, il, ll, **kw))
    eng.dev.sync()
    w2 = eng.get_weights()
    m2, v2, it2 = eng.m.download(), eng.v.download(), eng.iterations
    eng._bind()
    eng.dev.stream(0)
    eng.dev.call("mgr_scan_status_inject", _capi.SCAN_GAVE_UP)
    with pytest.raises(_capi.MgrError) as ei:
        eng.train_step(xs, labels, il, ll, **kw)          # step 3
    assert "step 3" in str(ei.value) and "skipped" in str(ei.value)
    eng.dev.sync()
    assert eng.scan_health() == (_capi.SCAN_GAVE_UP, 1)
    w3 = eng.get_weights()
    for k in w2:
        assert np.array_equal(w3[k], w2[k]), k
    assert np.array_equal(eng.m.download(), m2) and np.array_equal(eng.v.download(), v2)
    assert eng.updates_skipped == 1
    eng.clear_scan_status()
    assert np.isfinite(eng.train_step(xs, labels, il, ll, **kw))     # step 4 trains again
    eng.dev.sync()
    assert not np.array_equal(eng.get_weights()["dense/W"], w2["dense/W"]) and eng.scan_health() == (0, 0)
    eng.close()


def test_a_diverged_step_does_not_poison_the_weights_and_a_restored_checkpoint_trains(device):
    """NaN recurrent weight in a frozen encoder: the loss is NaN, the trainable weights stay what they were (the reference would
    have written NaN into them), and set_weights() of a good checkpoint starts with a clean status."""
    eng, w, (xs, labels, il, ll) = _fusion_engine(device)
    bad = {k: v.copy() for k, v in w.items()}
    bad["the_input_audio/l0/fwd/U"][3, 2 * 500 + 5] = np.nan
    eng.set_weights(bad)
    before = eng.get_weights()
    loss = eng.train_step(xs, labels, il, ll)
    assert np.isnan(loss) and eng.nonfinite_seen
    eng.dev.sync()
    after = eng.get_weights()
    for k in before:
        assert np.array_equal(after[k], before[k], equal_nan=True), k
    assert eng.scan_health() == (8, 1)     # MGR_SCAN_NONFINITE, one update skipped
    eng.set_weights(w)                                   # restoring a checkpoint clears the status
    assert not eng.nonfinite_seen
    loss = eng.train_step(xs, labels, il, ll)
    eng.dev.sync()
    assert np.isfinite(loss) and not eng.nonfinite_seen and eng.scan_health() == (0, 0)
    eng.close()


def test_scan_status_is_per_engine_on_a_shared_device(device):
    """Two engines on ONE Device: the diverged one reports NaN, the healthy one is not affected (each engine binds its own
    status block; round 2 had one context-wide word)."""
    eng_a, w, (xs, labels, il, ll) = _fusion_engine(device, seed=1)
    eng_b, _, _ = _fusion_engine(device, seed=2)
    bad = {k: v.copy() for k, v in w.items()}
    bad["the_input_skeletal/l0/bwd/U"][1, 2 * 300 + 2] = np.nan
    eng_a.set_weights(bad)
    la = eng_a.train_step(xs, labels, il, ll, apply_update=False)
    lb = eng_b.train_step(xs, labels, il, ll)
    la2 = eng_a.train_step(xs, labels, il, ll, apply_update=False)
    assert np.isnan(la) and np.isnan(la2) and eng_a.nonfinite_seen
    assert np.isfinite(lb) and not eng_b.nonfinite_seen and eng_b.updates_skipped == 0
    P = eng_b.predict(xs)
    assert np.isfinite(P).all()
    eng_a.close()
    eng_b.close()
