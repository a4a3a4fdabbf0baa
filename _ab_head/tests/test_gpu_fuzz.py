"""-m gpu: randomised multi-job scan launches against the oracle (forward cluster kernels: K-split step, LDS-image step,
and their mixes; BPTT cluster kernels incl. the split-role one).  MGR_FUZZ_CASES raises the number of cases (default 8)."""
import os

import numpy as np
import pytest

from oracle import keras_ref as kr
from tests.helpers import rel_err

pytestmark = pytest.mark.gpu


def _lstm_weights(rng, F, H, scale_u):
    W = rng.uniform(-0.3, 0.3, (F, 4 * H))
    U = rng.standard_normal((H, 4 * H)) * scale_u
    b = rng.uniform(-0.2, 0.2, (4 * H,))
    return W, U, b


@pytest.mark.parametrize("case", range(int(os.environ.get("MGR_FUZZ_CASES", "8"))))
def test_random_multi_job_scans(device, case):
    from mgr_amd import _capi
    dev = device
    rng = np.random.default_rng(9000 + case)
    f32 = np.float32
    njobs = int(rng.integers(1, 5))
    T = int(rng.integers(3, 120))
    B = int(rng.integers(1, 70))
    jobs, bjobs, refs, outs, keep = [], [], [], [], []
    for _ in range(njobs):
        H = int(rng.choice([100, 300, 500, 128, 32]))
        F = int(rng.integers(3, 9))
        reverse = int(rng.integers(0, 2))
        W, U, b = _lstm_weights(rng, F, H, 0.6 / np.sqrt(H))
        x = rng.standard_normal((B, T, F))
        y_ref, cache = kr.lstm_forward(x, W, U, b, None, bool(reverse))
        dY = rng.standard_normal((B, T, H)) * 0.1
        Wp, Up, bp = dev.empty((F, 4 * H)), dev.empty((H, 4 * H)), dev.empty((4 * H,))
        dev.call("mgr_lstm_pack", dev.array(W.astype(f32)), Wp, F, H, 0)
        dev.call("mgr_lstm_pack", dev.array(U.astype(f32)), Up, H, H, 0)
        dev.call("mgr_lstm_pack", dev.array(b.astype(f32)), bp, 1, H, 0)
        Z = dev.empty((B, T, 4 * H))
        dev.call("mgr_lstm_input_proj", dev.array(x.astype(f32)), F, 0, Wp, bp, Z, B, T, F, H)
        Y, G, Cs = dev.empty((B, T, H)), dev.empty((B, T, H, 4)), dev.empty((B, T, H))
        jobs.append(dict(Z=Z, Up=Up, Y=Y, ldy=H, R=0, ldr=0, gates=G, cs=Cs, B=B, T=T, H=H, reverse=reverse))
        dZ = dev.zeros((B, T, 4 * H))
        bjobs.append(dict(dY=dev.array(dY.astype(f32)), gates=G, cs=Cs, Up=Up, dZ=dZ, lddy=H, B=B, T=T, H=H, reverse=reverse))
        refs.append((y_ref, cache, dY, W, U))
        outs.append((Y, G, Cs, dZ, Z))
        keep += [Wp, Up, bp]
    dev.call("mgr_tune", 1, 1)   # synchronous give-up check
    try:
        arr = _capi.make_scan_jobs(jobs)
        ws = dev.bytes(dev.lib.mgr_lstm_scan_multi_ws_bytes(njobs, arr))
        _capi.check(dev.lib.mgr_lstm_scan_fwd_multi(dev.ctx, njobs, arr, ws.ptr, ws.nbytes))
        for (Y, G, Cs, _, _), (y_ref, cache, _, _, _) in zip(outs, refs):
            assert rel_err(Y.download(), y_ref) < 5e-5
            assert rel_err(Cs.download(), cache["c"]) < 5e-5
        barr = _capi.make_scan_bwd_jobs(bjobs)
        bws = dev.bytes(dev.lib.mgr_lstm_scan_bwd_multi_ws_bytes(njobs, barr))
        _capi.check(dev.lib.mgr_lstm_scan_bwd_multi(dev.ctx, njobs, barr, bws.ptr, bws.nbytes))
        # BPTT check through its consequence: dU = sum_t h_{t-1}^T dz_t against the oracle's dU
        for (Y, G, Cs, dZ, _), (y_ref, cache, dY, W, U), j in zip(outs, refs, bjobs):
            _, _, dU_ref, _ = kr.lstm_backward(dY, cache, need_dx=False)
            H = j["H"]
            dz = dZ.download().reshape(B, T, H, 4).transpose(0, 1, 3, 2).reshape(B, T, 4 * H)   # packed -> Keras gate-major
            h = Y.download()
            hp = np.zeros_like(h)
            if j["reverse"]:
                hp[:, :-1] = h[:, 1:]
            else:
                hp[:, 1:] = h[:, :-1]
            dU = np.einsum("bth,btg->hg", hp.astype(np.float64), dz.astype(np.float64))
            assert rel_err(dU, dU_ref) < 5e-4
    finally:
        dev.call("mgr_tune", 1, 0)


@pytest.mark.parametrize("case", range(int(os.environ.get("MGR_FUZZ_NETS", "6"))))
def test_random_networks(device, case):
    """Random members of the reference's network family (1-2 streams of 1-2 BiLSTM layers, optional fusion BiLSTM, frozen or
    trainable encoders) at random sizes: loss, softmax and every trainable gradient against the fp64 oracle."""
    import mgr_amd  # noqa: F401
    from mgr_amd.engine import Engine
    from mgr_amd.spec import NetworkSpec
    from mgr_amd.synthetic import synthetic_arrays, synthetic_weights
    from oracle import network_ref as nr
    rng = np.random.default_rng(7000 + case)
    hs = [8, 16, 32, 64, 100, 128, 300]
    nstreams = int(rng.integers(1, 3))
    use_fusion = bool(rng.integers(0, 2)) or nstreams == 2
    streams = []
    for si in range(nstreams):
        nl = int(rng.integers(1, 3))
        H = int(rng.choice(hs))
        streams.append({"name": "in%d" % si, "F": int(rng.integers(3, 12)), "noise": float(rng.choice([0.0, 0.5])),
                        "residual": nl == 2, "trainable": bool(rng.integers(0, 2)) or not use_fusion,
                        "layers": [{"H": H, "dropout": float(rng.choice([0.0, 0.4])), "name": "l%d_%d" % (si, k)} for k in range(nl)]})
    fusion = {"H": int(rng.choice([8, 16, 32, 100])), "dropout": 0.5, "name": "fus"} if use_fusion else None
    C = int(rng.integers(5, 23))
    spec = NetworkSpec(streams, fusion, {"dropout": float(rng.choice([0.0, 0.5])), "C": C})
    B, T = int(rng.integers(1, 40)), int(rng.integers(12, 70))
    Lmax = 8
    eng = Engine(spec, B, T, Lmax, device=device, seed=case)
    w = synthetic_weights(spec, 500 + case)
    eng.set_weights(w)
    xs, labels, il, ll = synthetic_arrays(spec, B, T, Lmax, 600 + case, lmin=1, lmax=4)
    sd = spec.to_dict()
    rand = nr.draw_rand(sd, B, T, np.random.default_rng(700 + case))
    w64 = {k: v.astype(np.float64) for k, v in w.items()}
    ref_loss, ref_lb, ref_g, ref_P = nr.loss_and_grads(sd, w64, xs, labels, il, ll, rand)
    eng.enqueue_train_step(xs, labels, il, ll, rand=rand, apply_update=False)
    loss = eng.read_loss()
    assert abs(loss - ref_loss) <= 1e-4 * abs(ref_loss), (loss, ref_loss, sd)
    assert rel_err(eng.P.download(), ref_P) < 2e-4
    g = eng.get_grads()
    assert set(g) == set(ref_g)
    for k in ref_g:
        assert rel_err(g[k], ref_g[k]) < 2e-3, (k, rel_err(g[k], ref_g[k]), sd)
    eng.close()


@pytest.mark.parametrize("case", range(int(os.environ.get("MGR_FUZZ_SEQS", "4"))))
def test_random_operation_sequences_pipelined_equals_plain(device, case):
    """Random interleavings of training steps (with and without an announced next batch), validation losses and
    predictions on a pipelined engine give exactly the numbers of an engine that never overlaps anything: exercises the
    two-stream schedule, the copy stream and the alternating input / label / FEAT buffers."""
    import mgr_amd  # noqa: F401
    from mgr_amd.configs import fusion_spec
    from mgr_amd.engine import Engine
    from mgr_amd.synthetic import synthetic_arrays, synthetic_weights
    rng = np.random.default_rng(3000 + case)
    big = case % 4 == 3     # every fourth sequence at the reference's layer sizes (multi-CU cluster kernels, deferral)
    spec = fusion_spec() if big else fusion_spec(h_audio=32, h_skeletal=16, h_fusion=8)
    B, T, Lmax = int(rng.integers(2, 20)), int(rng.integers(20, 60)), 6
    w = synthetic_weights(spec, 40 + case)
    batches = [synthetic_arrays(spec, B, T, Lmax, 900 + 10 * case + i, lmin=1, lmax=4) for i in range(6)]
    ops = []
    for _ in range(14):
        r = rng.random()
        ops.append(("train", int(rng.integers(0, 6)), bool(rng.integers(0, 2))) if r < 0.6 else
                   (("val" if r < 0.8 else "predict"), int(rng.integers(0, 6)), False))

    def run(pipelined):
        eng = Engine(spec, B, T, Lmax, device=device, seed=77)
        eng.set_weights(w)
        out = []
        for i, (op, bi, announce) in enumerate(ops):
            xs, labels, il, ll = batches[bi]
            if op == "train":
                nxt = None
                if pipelined and announce and i + 1 < len(ops):
                    # usually the batch that really comes next; sometimes a wrong announcement (or one followed by a
                    # validation / prediction call), which the engine has to notice and discard
                    nxt = batches[ops[i + 1][1]][0] if (i + case) % 5 else batches[(ops[i + 1][1] + 1) % 6][0]
                out.append(eng.train_step(xs, labels, il, ll, next_inputs=nxt))
            elif op == "val":
                out.append(float(np.sum(eng.loss_on_batch(xs, labels, il, ll, train_phase=False))))
            else:
                out.append(float(eng.predict(xs).sum()))
        weights = eng.get_weights()
        eng.close()
        return out, weights

    a, wa = run(True)
    b, wb = run(False)
    assert a == b, (ops, a, b)
    for k in wa:
        assert np.array_equal(wa[k], wb[k]), k


@pytest.mark.parametrize("case", range(int(os.environ.get("MGR_FUZZ_GEMMS", "6"))))
def test_random_gemm_shapes(device, case):
    """The three MFMA GEMM entry points at random (ragged) shapes, strides, alignments and masks against numpy fp64:
    packed column j*4+g holds gate g of unit j; the Keras per-gate input-dropout mask multiplies X per (gate, sample, feature)."""
    dev = device
    rng = np.random.default_rng(5000 + case)
    f32 = np.float32
    B, T = int(rng.integers(1, 9)), int(rng.integers(1, 300))
    F, H = int(rng.integers(1, 200)) if case % 3 else int(rng.integers(128, 700)), int(rng.integers(1, 80)) if case % 2 else int(rng.integers(60, 140))
    N = 4 * H
    ldx = F + int(rng.choice([0, 0, 1, 3, 4, 8]))           # ldx not a multiple of 4 -> the unaligned (scalar) loaders
    use_mask = bool(rng.integers(0, 2))
    Xh = np.zeros((B, T, ldx), f32)
    Xh[:, :, :F] = rng.standard_normal((B, T, F))
    Wp = (rng.standard_normal((F, N)) * 0.2).astype(f32)
    bp = rng.standard_normal(N).astype(f32)
    mask = ((rng.random((4, B, F)) > 0.4) * 1.6).astype(f32) if use_mask else None
    X64 = Xh[:, :, :F].astype(np.float64)
    gate = np.arange(N) % 4

    def masked_x(g):   # (B,T,F) seen by gate g
        return X64 * mask[g][:, None, :].astype(np.float64) if use_mask else X64

    dX_ = dev.array(Xh)
    dW_, db_ = dev.array(Wp), dev.array(bp)
    dM = dev.array(mask) if use_mask else 0
    # nn
    Z = dev.empty((B, T, N))
    dev.call("mgr_lstm_input_proj", dX_, ldx, dM, dW_, db_, Z, B, T, F, H)
    Zref = np.empty((B, T, N))
    for g in range(4):
        Zref[:, :, gate == g] = masked_x(g) @ Wp[:, gate == g].astype(np.float64) + bp[gate == g]
    assert rel_err(Z.download(), Zref) < 2e-5
    # both directions of a Bidirectional layer in one call: bit-identical to two calls, whichever kernel it picks
    Wp2 = (rng.standard_normal((F, N)) * 0.2).astype(f32)
    bp2 = rng.standard_normal(N).astype(f32)
    mask2 = ((rng.random((4, B, F)) > 0.4) * 1.6).astype(f32) if use_mask else None
    dW2_, db2_ = dev.array(Wp2), dev.array(bp2)
    dM2 = dev.array(mask2) if use_mask else 0
    Zb = dev.empty((B, T, N))
    dev.call("mgr_lstm_input_proj", dX_, ldx, dM2, dW2_, db2_, Zb, B, T, F, H)
    Za2, Zb2 = dev.empty((B, T, N)), dev.empty((B, T, N))
    dev.call("mgr_lstm_input_proj_pair", dX_, ldx, dM, dW_, db_, Za2, dM2, dW2_, db2_, Zb2, B, T, F, H)
    assert np.array_equal(Za2.download(), Z.download()) and np.array_equal(Zb2.download(), Zb.download())
    # dropout-aware projection (K loops over the kept features; from F = 128 on, the dense kernel below that)
    if use_mask:
        Zs = dev.empty((B, T, N))
        wsd = dev.bytes(dev.lib.mgr_lstm_input_proj_dropout_ws_bytes(B, F, H))
        dev.call("mgr_lstm_input_proj_dropout", dX_, ldx, dM, 0.4, dW_, db_, Zs, B, T, F, H, wsd, wsd.nbytes)
        assert rel_err(Zs.download(), Zref) < 2e-5
    # tn (dW, dU, db) with a time-shifted h
    reverse = int(rng.integers(0, 2))
    ldh = H + int(rng.choice([0, 4, 5]))
    Hh = np.zeros((B, T, ldh), f32)
    Hh[:, :, :H] = rng.standard_normal((B, T, H))
    dZ = (rng.standard_normal((B, T, N)) * 0.3).astype(f32)
    gW, gU, gb = dev.empty((F, N)), dev.empty((H, N)), dev.empty((N,))
    ws = dev.bytes(dev.lib.mgr_lstm_param_grads_ws_bytes(B, T, F, H))
    dev.call("mgr_lstm_param_grads", dX_, ldx, dM, dev.array(Hh), ldh, dev.array(dZ), gW, gU, gb, B, T, F, H, reverse, ws, ws.nbytes)
    dZ64 = dZ.astype(np.float64)
    gW_ref = np.empty((F, N))
    for g in range(4):
        gW_ref[:, gate == g] = np.einsum("btf,btn->fn", masked_x(g), dZ64[:, :, gate == g])
    hprev = np.zeros((B, T, H))
    h64 = Hh[:, :, :H].astype(np.float64)
    if reverse:
        hprev[:, :-1] = h64[:, 1:]
    else:
        hprev[:, 1:] = h64[:, :-1]
    assert rel_err(gW.download(), gW_ref) < 5e-5
    assert rel_err(gU.download(), np.einsum("bth,btn->hn", hprev, dZ64)) < 5e-5
    assert rel_err(gb.download(), dZ64.sum((0, 1))) < 5e-5
    # nt (dX), plain and accumulating
    lddx = F + int(rng.choice([0, 2, 4]))
    base = rng.standard_normal((B, T, lddx)).astype(f32)
    dXo = dev.array(base)
    dev.call("mgr_lstm_input_grad", dev.array(dZ), dW_, dM, dXo, lddx, 1, B, T, F, H)
    ref = np.zeros((B, T, F))
    for g in range(4):
        part = dZ64[:, :, gate == g] @ Wp[:, gate == g].astype(np.float64).T
        ref += part * mask[g][:, None, :] if use_mask else part
    got = dXo.download()
    assert rel_err(got[:, :, :F] - base[:, :, :F], ref) < 5e-5
    assert np.array_equal(got[:, :, F:], base[:, :, F:])      # padding columns untouched


@pytest.mark.parametrize("case", range(int(os.environ.get("MGR_FUZZ_CTC", "6"))))
def test_random_ctc(device, case):
    """CTC loss and logit gradient at random shapes: label rows with repeats, with the blank class itself as a target
    (the reference's empty-label substitution, data_generator.py:228-238), per-sample input lengths, peaky or flat
    posteriors; feasible alignments only (TF returns inf otherwise)."""
    from tests.test_gpu_kernels import _run_ctc
    rng = np.random.default_rng(1000 + case)
    B, T = int(rng.integers(1, 12)), int(rng.integers(5, 260))
    Cn = int(rng.integers(3, 50))
    Lmax = int(rng.integers(1, 40))
    z = rng.standard_normal((B, T, Cn)) * float(rng.choice([0.5, 2.0, 6.0]))
    P = np.exp(z - z.max(-1, keepdims=True))
    P = (P / P.sum(-1, keepdims=True)).astype(np.float32)
    labels = -np.ones((B, Lmax))
    ll = np.zeros(B, np.int64)
    il = np.zeros(B, np.int64)
    for b in range(B):
        Tin = int(rng.integers(max(1, (T - 2) // 3), T - 1))          # frames after the skip
        Lcap = max(1, min(Lmax, (Tin + 1) // 2))                      # room for the blanks between repeats
        L = int(rng.integers(1, Lcap + 1))
        seq = rng.integers(0, Cn, size=L) if rng.random() < 0.3 else rng.integers(0, Cn - 1, size=L)   # sometimes incl. the blank id
        if rng.random() < 0.4 and L > 1:
            seq[1] = seq[0]                                            # adjacent repeat
        labels[b, :L] = seq
        ll[b], il[b] = L, Tin
    ref_loss, ref_dz = kr.ctc_loss_grad(P.astype(np.float64), labels, il, ll)
    ok = np.isfinite(ref_loss)
    loss, dz = _run_ctc(device, P, labels, il, ll)
    assert np.allclose(loss[ok], ref_loss[ok], rtol=1e-4), (loss, ref_loss)
    assert rel_err(dz[ok], ref_dz[ok]) < 1e-3


@pytest.mark.parametrize("case", range(int(os.environ.get("MGR_FUZZ_DECODE", "20"))))
def test_random_decode(device, case):
    """Thresholded best-path decode (with the reference's list.remove quirk) and CTC prefix beam search at random shapes,
    thresholds, beam widths and posteriors from flat to peaky: label sequences identical to the oracle's."""
    from mgr_amd import decoding
    rng = np.random.default_rng(2000 + case)
    N, T = int(rng.integers(1, 7)), int(rng.integers(4, 160))
    Cn = int(rng.integers(2, 45))
    z = rng.standard_normal((N, T, Cn)) * float(rng.choice([0.3, 1.5, 4.0]))
    for n in range(N):   # runs of a dominant class, like a trained network's output
        t = 0
        while t < T:
            run = int(rng.integers(1, 12))
            z[n, t:t + run, int(rng.integers(0, Cn))] += rng.uniform(0, 6)
            t += run
    P = np.exp(z - z.max(-1, keepdims=True))
    P = (P / P.sum(-1, keepdims=True)).astype(np.float32)
    thr = float(rng.choice([0.3, 0.5, 0.75, 0.97]))
    assert decoding.greedy_decode(P, thr, dev=device) == kr.greedy_decode_quirk(P, thr)
    W = int(rng.choice([1, 2, 5, 10, 16]))
    if W * (Cn + 1) <= 64 * 34:
        il = rng.integers(1, T - 1, size=N)
        mr = bool(rng.integers(0, 2))
        ref, rs = kr.ctc_beam_search(P, il, beam_width=W, merge_repeated=mr)
        got, gs = decoding.beam_search_decode(P, il, beam_width=W, merge_repeated=mr, dev=device)
        assert got == ref, (N, T, Cn, W, mr)
        assert np.allclose(gs, rs, rtol=1e-12)
