"""-m gpu: pipelined inference / validation (Engine.predict_stream, Model.predict_generator / evaluate_generator) - the batches of
a run are independent, two are kept in flight; results must be bit for bit what the one-batch-at-a-time calls give
(reference: predict_generator over the whole set, sequence_decoding.py:118-127; the validation loop, multimodal.py:264-269)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _batches(spec, B, T, Lmax, n, seed0=300):
    from mgr_amd.synthetic import synthetic_arrays
    out = []
    for i in range(n):
        xs, labels, il, ll = synthetic_arrays(spec, B, T, Lmax, seed0 + i, lmin=2, lmax=5)
        out.append((xs, labels, il, ll))
    return out


@pytest.mark.parametrize("kind", ["fusion_inference", "fusion_training_engine", "unimodal"])
def test_pipelined_predict_equals_sequential_predict_bit_for_bit(device, kind):
    from mgr_amd.configs import audio_spec, fusion_spec
    from mgr_amd.engine import Engine
    from mgr_amd.synthetic import synthetic_weights
    spec = audio_spec(h=128) if kind == "unimodal" else fusion_spec()
    B, T, Lmax = 16, 72, 6
    eng = Engine(spec, B, T, Lmax, device=device, seed=5, inference_only=(kind != "fusion_training_engine"))
    eng.set_weights(synthetic_weights(spec, 11))
    data = _batches(spec, B, T, Lmax, 5)
    seq = [eng.predict(b[0]) for b in data]
    pipe = list(eng.predict_stream((b[0] for b in data), output="posteriors"))
    assert len(pipe) == 5
    for a, b in zip(seq, pipe):
        assert np.array_equal(a, b)
    assert not np.array_equal(seq[0], seq[1])
    # ... and again (buffers recycled), then the decode outputs computed on the device
    pipe2 = list(eng.predict_stream((b[0] for b in data), output="posteriors"))
    assert all(np.array_equal(a, b) for a, b in zip(seq, pipe2))
    from mgr_amd import decoding
    am = list(eng.predict_stream((b[0] for b in data), output="argmax"))
    for P, (best, prob) in zip(seq, am):
        b0, p0 = decoding.frame_argmax(P, skip=2, dev=device)
        assert np.array_equal(best, b0) and np.array_equal(prob, p0)
    bm = list(eng.predict_stream((b[0] for b in data), output="beam", beam_width=10))
    for P, (paths, logp) in zip(seq, bm):
        p0, s0 = decoding.beam_search_decode(P, None, beam_width=10, skip=2, dev=device)
        assert paths == p0 and np.array_equal(logp, s0)
    eng.close()


def test_pipelined_validation_losses_equal_loss_on_batch(device):
    """output="loss" with the learning phase the reference leaves on during validation (dropout / noise from the device RNG):
    the pipelined run and the one-batch-at-a-time run draw the same random numbers and give the same per-sample losses."""
    from mgr_amd.configs import fusion_spec
    from mgr_amd.engine import Engine
    from mgr_amd.synthetic import synthetic_weights
    spec = fusion_spec()
    B, T, Lmax = 16, 72, 6
    data = _batches(spec, B, T, Lmax, 4)
    res = []
    for mode in ("sequential", "pipelined"):
        eng = Engine(spec, B, T, Lmax, device=device, seed=9)
        eng.set_weights(synthetic_weights(spec, 11))
        if mode == "sequential":
            res.append([eng.loss_on_batch(*b, train_phase=True) for b in data])
        else:
            res.append(list(eng.predict_stream(iter(data), output="loss", train_phase=True)))
        eng.close()
    for a, b in zip(*res):
        assert np.array_equal(a, b) and np.all(np.isfinite(a))
    assert not np.array_equal(res[0][0], res[0][1])


def test_model_predict_generator_pads_a_short_last_batch_and_decodes_on_the_device(device, tmp_path):
    """Model.predict_generator over 5 batches, the last one short: equal to predict_on_batch batch by batch; decode='argmax' feeds
    decode_argmax to the same MLF bytes as decode_batch on the posteriors."""
    from mgr_amd.multimodal_fusion import sequence_decoding as sd
    from mgr_amd.configs import fusion_spec
    from mgr_amd.keras_like import Model
    from mgr_amd.synthetic import synthetic_weights
    from mgr_amd import keras_like as K
    K.set_learning_phase(0)                   # what the decode scripts set (sequence_decoding.py:81); it is process-wide state
    spec = fusion_spec()
    B, T = 8, 64
    data = [b[0] for b in _batches(spec, B, T, 4, 5, seed0=400)]
    data[-1] = {k: v[:3] for k, v in data[-1].items()}                 # a short last batch
    m = Model(spec, device=device)
    m.set_weights_dict(synthetic_weights(spec, 11))
    ref = np.concatenate([m.predict_on_batch({k: np.concatenate([v, np.zeros((B - v.shape[0],) + v.shape[1:], v.dtype)]) for k, v in b.items()})[:next(iter(b.values())).shape[0]]
                          for b in data], axis=0)
    P = m.predict_generator(iter(data), steps=5)
    assert P.shape == (4 * B + 3, T, spec.num_classes) and np.array_equal(P, ref)
    best, prob = m.predict_generator(iter(data), steps=5, decode="argmax")
    f_list = list(range(1, P.shape[0] + 1))
    a = sd.decode_batch(P, f_list, out_file=str(tmp_path / "a.mlf"))
    b = sd.decode_argmax(best, prob, f_list, out_file=str(tmp_path / "b.mlf"))
    assert a == b and open(tmp_path / "a.mlf", "rb").read() == open(tmp_path / "b.mlf", "rb").read()
    paths, logp = m.predict_generator(iter(data), steps=5, decode="beam", beam_width=10)
    assert len(paths) == P.shape[0] and logp.shape == (P.shape[0],)


def test_a_nan_input_marks_its_own_sample_of_its_own_batch_and_nothing_else(device):
    """One corrupt sequence in a decode run (a NaN in sample 3 of batch 1): the reference yields NaN for that sample only.  Here a
    non-finite hidden state is fed back as 0 by the multi-CU exchange, so OTHER units of the sample may look finite - the engine
    hands out NaN scores / no labels for exactly the samples the scans of THAT pass marked (status block words [8, 16), one block
    per inference pass), not for every sample of every later batch (round 4's sticky flag), and the engine's own
    training status is not touched by inference passes."""
    from mgr_amd.configs import fusion_spec
    from mgr_amd.engine import Engine
    from mgr_amd.synthetic import synthetic_weights
    spec = fusion_spec()
    B, T, Lmax = 16, 72, 6
    eng = Engine(spec, B, T, Lmax, device=device, seed=5, inference_only=True)
    eng.set_weights(synthetic_weights(spec, 11))
    data = [b[0] for b in _batches(spec, B, T, Lmax, 4)]
    clean = [eng.predict(x) for x in data]
    bad = {k: v.copy() for k, v in data[1].items()}
    bad["the_input_audio"][3, 10, 7] = np.nan
    feed = [data[0], bad, data[2], data[3]]
    other = np.arange(B) != 3
    for out in ("posteriors", "argmax", "beam"):
        res = list(eng.predict_stream(iter(feed), output=out, beam_width=10))
        ref = list(eng.predict_stream(iter(data), output=out, beam_width=10))
        for i in (0, 2, 3):                                   # the batches around it: bit for bit the clean run
            a, b = res[i], ref[i]
            if out == "posteriors":
                assert np.array_equal(a, b) and np.array_equal(a, clean[i])
            elif out == "argmax":
                assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
            else:
                assert a[0] == b[0] and np.array_equal(a[1], b[1])
        a, b = res[1], ref[1]
        if out == "posteriors":
            assert np.all(np.isnan(a[3])) and np.array_equal(a[other], b[other])
        elif out == "argmax":
            assert np.all(a[0][3] == -1) and np.all(np.isnan(a[1][3]))
            assert np.array_equal(a[0][other], b[0][other]) and np.array_equal(a[1][other], b[1][other])
        else:
            assert a[0][3] == [] and np.isnan(a[1][3])
            assert [p for k, p in enumerate(a[0]) if k != 3] == [p for k, p in enumerate(b[0]) if k != 3]
            assert np.array_equal(a[1][other], b[1][other])
    # one batch at a time: the same marking, and the next call is clean again
    P = eng.predict(bad)
    assert np.all(np.isnan(P[3])) and np.array_equal(P[other], clean[1][other])
    assert np.array_equal(eng.predict(data[2]), clean[2])
    assert eng.scan_health() == (0, 0) and not eng.nonfinite_seen
    eng.close()


def test_inference_on_the_f32_mfma_kernels_equals_the_split_f16_path(device):
    """tune keys 14 / 15 = 1 (the f32 A/B switch bench.py's second leg uses) in an INFERENCE pass: the unmasked wide projections then
    run the f32 kernel over all features (round 4 raised there: no mask and no f16).  Both paths against each other."""
    from mgr_amd.configs import fusion_spec
    from mgr_amd.engine import Engine
    from mgr_amd.synthetic import synthetic_weights
    spec = fusion_spec()
    B, T, Lmax = 16, 72, 6
    eng = Engine(spec, B, T, Lmax, device=device, seed=5, inference_only=True)
    eng.set_weights(synthetic_weights(spec, 11))
    data = [b[0] for b in _batches(spec, B, T, Lmax, 3)]
    split = [eng.predict(x) for x in data]
    device.call("mgr_tune", 14, 1)
    device.call("mgr_tune", 15, 1)
    try:
        f32 = [eng.predict(x) for x in data]
        pipe = list(eng.predict_stream(iter(data), output="posteriors"))
    finally:
        device.call("mgr_tune", 14, 0)
        device.call("mgr_tune", 15, 0)
    for a, b, c in zip(split, f32, pipe):
        assert np.array_equal(b, c)
        assert np.abs(a - b).max() < 2e-5           # posteriors in [0, 1]
    eng.close()
