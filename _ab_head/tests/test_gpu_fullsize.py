"""-m gpu: the BASELINE.json configurations at their FULL sizes (the bench line's shape, the decode set's size) and the
real-data CSV backend on the GPU path.

 F  configs[2]  fusion, B = 64, T = 1900: a step with injected randomness against the fp64 oracle on two of its samples, then
                pipelined device-RNG steps: finite, no scan gave up, a repeated run bit-identical
 D  configs[4]  decode of N = 276 sequences x 1900 frames: thresholded best path and beam = 10 against the CPU oracle on ALL 276
 f2 SURVEY 8(f2) a util/mix_data.py-layout directory -> CsvStore -> DataGenerator -> fit_generator == the same arrays fed by hand
"""
import multiprocessing as mp
import os
import random

import numpy as np
import pytest

from oracle import keras_ref as kr
from oracle import network_ref as nr

pytestmark = pytest.mark.gpu


def test_config_F_full_size_step_and_pipelined_determinism(device):
    import ctypes
    from mgr_amd.configs import baseline_config
    from mgr_amd.engine import Engine
    from mgr_amd.synthetic import synthetic_arrays, synthetic_weights
    spec, B, T, Lmax = baseline_config("F")
    assert (B, T) == (64, 1900)
    w = synthetic_weights(spec, 20131900 + 3)
    xs, labels, il, ll = synthetic_arrays(spec, B, T, Lmax, 20131900 + 3)
    sd = spec.to_dict()
    rand = nr.draw_rand(sd, B, T, np.random.default_rng(77), np.float32)
    device.call("mgr_scan_status_clear")

    def run(seed):
        eng = Engine(spec, B, T, Lmax, device=device, seed=seed)
        eng.set_weights(w)
        # (1) one step with injected masks / noise: per-sample losses of samples 0 and 1 vs the fp64 oracle on those two samples
        eng.enqueue_train_step(xs, labels, il, ll, rand=rand, apply_update=False)
        lb = eng.loss_b.download()
        # (2) three pipelined device-RNG steps from the same weights
        eng._upload_inputs(xs, None, True)
        eng._upload_labels(labels, il, ll)
        losses = []
        for i in range(3):
            eng.enqueue_train_step(None, None, None, None, upload=False, prefetch_next=i < 2)
            losses.append(eng.read_loss())
        device.sync()
        wf = eng.get_weights()
        eng.close()
        return lb, losses, wf

    lb, losses, wf = run(5)
    sub = slice(0, 2)
    rand2 = {k: (v[:, sub] if (k.endswith("/mask") and k != "head/mask") else v[sub]) for k, v in rand.items()}
    w64 = {k: v.astype(np.float64) for k, v in w.items()}
    _, ref_lb, _, _ = nr.loss_and_grads(sd, w64, {k: v[sub] for k, v in xs.items()}, labels[sub], il[sub], ll[sub], rand2)
    assert np.allclose(lb[sub], ref_lb, rtol=1e-4), (lb[sub], ref_lb)           # north-star tolerance: 1e-4 on the CTC loss
    assert np.all(np.isfinite(lb)) and np.all(np.isfinite(losses)) and len(set(losses)) == 3
    st = ctypes.c_uint(7)
    device.call("mgr_scan_status", ctypes.byref(st))
    assert st.value == 0
    lb2, losses2, wf2 = run(5)
    assert np.array_equal(lb, lb2) and losses == losses2                          # run-to-run bit-identical
    for k in wf:
        assert np.array_equal(wf[k], wf2[k]), k
    moved = [k for k in wf if not np.array_equal(wf[k], w[k])]
    assert moved and all(k.startswith(("fusion/", "dense/")) for k in moved)      # only the trainable part moved

    # (3) round 5: the default schedule of bench.py / fit_generator - the encoder stream handed its work TWO calls ahead, FUSED encoder
    # scans (8-wave workgroups, one per CU: only this shape has launches that take the form), the fusion layer's recurrences started
    # behind their residency, the depth-1 projections in front of the deepest scan - against the plain schedule: the same losses and
    # weights to the last bit over six steps
    from mgr_amd.engine import Schedule

    def run6(**sched):
        eng = Engine(spec, B, T, Lmax, device=device, seed=5, schedule=Schedule(**sched))
        eng.set_weights(w)
        eng._upload_inputs(xs, None, True)
        eng._upload_labels(labels, il, ll)
        out = []
        two = sched.get("encoders_two_ahead", True)
        for i in range(6):
            eng.enqueue_train_step(None, None, None, None, upload=False, prefetch_next=i < 5, prefetch_after_next=two and i < 4)
            out.append(eng.read_loss())
        device.sync()
        wf_ = eng.get_weights()
        nl, ns = ctypes.c_int(), ctypes.c_int()
        device.call("mgr_persist_stats", ctypes.byref(nl), ctypes.byref(ns))
        eng.close()
        return out, wf_

    def waits():
        out = (ctypes.c_uint * 4)()
        device.call("mgr_resident_wait_stats", out)
        return int(out[0]), int(out[1])

    device.sync()
    n0, b0 = waits()
    la, wa = run6()
    n1, b1 = waits()
    # (round 6: every residency wait of the fused schedule found its launch - the number is handed over by the launch itself)
    # (one expired wait tolerated: a host that is late by more than the bound with a launch expires the wait for it - noise of the box,
    #  not of the schedule; a broken hand-over expires every one)
    assert n1 - n0 >= 8 and b1 - b0 <= 1, (n1 - n0, b1 - b0)
    lp, wp = run6(fused_encoder_scans=False, depth1_proj_ahead=False, encoders_two_ahead=False)
    assert la == lp and all(np.array_equal(wa[k], wp[k]) for k in wa)
    device.call("mgr_scan_status", ctypes.byref(st))
    assert st.value == 0


def _oracle_chunk(args):
    """Pool worker: the oracle's loss_and_grads on a slice of the batch, in fp64 and in fp32 (the error model of
    tests/test_gpu_baseline_configs.py).  The gradients come back scaled to the FULL batch's mean (x chunk / B)."""
    sd, w, xs, labels, il, ll, rand, B = args
    n = labels.shape[0]
    w64 = {k: v.astype(np.float64) for k, v in w.items()}
    _, lb64, g64, _ = nr.loss_and_grads(sd, w64, xs, labels, il, ll, rand)
    f32 = lambda d: {k: (None if v is None else np.asarray(v, np.float32)) for k, v in d.items()}
    _, _, g32, _ = nr.loss_and_grads(sd, f32(w), f32(xs), labels, il, ll, f32(rand))
    return lb64, {k: v * (n / B) for k, v in g64.items()}, {k: v.astype(np.float64) * (n / B) for k, v in g32.items()}


@pytest.mark.slow
def test_config_F_bench_shape_every_sample_and_the_gradients_against_the_oracle(device):
    """BASELINE configs[2] at the bench line's own shape, B = 64, T = 1900, injected randomness: ALL 64 per-sample CTC losses
    (1e-4 relative, north_star's bound) and every trainable gradient against the fp64 oracle: within 1e-4 of the tensor's largest
    entry (the kernels measure 1e-5; the distance of the SAME oracle run in float32 is printed beside it).  B = 64 means 4 batch groups x 2 directions x multi-CU clusters in every scan - what the
    B = 2 full-T case cannot show.  The oracle runs in a process pool forked from the clean fork server (8 slices of 8 samples)."""
    from mgr_amd.configs import baseline_config
    from mgr_amd.engine import Engine
    from mgr_amd.synthetic import synthetic_arrays, synthetic_weights
    from multiprocessing import forkserver
    from tests.helpers import rel_err
    spec, B, T, Lmax = baseline_config("F")
    assert (B, T) == (64, 1900)
    w = synthetic_weights(spec, 20131900 + 3)
    xs, labels, il, ll = synthetic_arrays(spec, B, T, Lmax, 20131900 + 3)
    sd = spec.to_dict()
    rand = nr.draw_rand(sd, B, T, np.random.default_rng(78), np.float32)
    eng = Engine(spec, B, T, Lmax, device=device, seed=5)
    eng.set_weights(w)
    eng.enqueue_train_step(xs, labels, il, ll, rand=rand, apply_update=False)
    lb = eng.loss_b.download()
    g = eng.get_grads()
    eng._check_scans()
    eng.close()
    jobs = []
    for i in range(0, B, 8):
        sub = slice(i, i + 8)
        r2 = {k: (v[:, sub] if (k.endswith("/mask") and k != "head/mask") else v[sub]) for k, v in rand.items()}
        jobs.append((sd, w, {k: v[sub] for k, v in xs.items()}, labels[sub], il[sub], ll[sub], r2, B))
    if getattr(forkserver._forkserver, "_forkserver_pid", None) is not None:
        with mp.get_context("forkserver").Pool(8) as pool:
            res = pool.map(_oracle_chunk, jobs)
    else:
        res = [_oracle_chunk(j) for j in jobs]
    ref_lb = np.concatenate([r[0] for r in res])
    assert np.allclose(lb, ref_lb, rtol=1e-4), np.abs(lb / ref_lb - 1).max()        # every one of the 64 samples
    assert set(g) == set(res[0][1])
    for k in g:
        ref = sum(r[1][k] for r in res)
        r32 = sum(r[2][k] for r in res)
        eg, eg32 = rel_err(g[k], ref), rel_err(r32, ref)
        print("   grad %-20s gpu %.2e, numpy-f32 %.2e" % (k, eg, eg32))
        # (round 5: the bound is what the kernels achieve with a margin of ten - measured 8e-6 ... 2e-5 - not the 5e-3 of the
        # numpy-float32 error model: a regression of the split-f16 path by one order of magnitude fails here)
        assert eg < 1e-4, (k, eg, eg32)


def _oracle_decode_chunk(args):
    P, il, beam = args
    b, s = kr.ctc_beam_search(P, il, beam_width=beam)
    return b, s, kr.greedy_decode_quirk(P, 0.5)


@pytest.mark.slow
def test_config_D_decode_full_set_matches_oracle_on_every_sequence(device):
    """BASELINE configs[4]: N = 276, T = 1900, C = 22, beam 10.  Label sequences bit-exact and beam scores to 1e-12 relative
    against the CPU oracle on all 276 sequences (the oracle runs in a process pool forked from the clean fork server)."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    from decode_bench import peaky_posteriors
    from mgr_amd import decoding
    N, T, Cn, beam = 276, 1900, 22, 10
    P, _ = peaky_posteriors(N, T, Cn, 20131900 + 5)
    il = np.full(N, T - 2)
    g_beam, g_scores = decoding.beam_search_decode(P, il, beam_width=beam, dev=device)
    g_greedy = decoding.greedy_decode(P, 0.5, dev=device)
    chunks = [(P[i:i + 6], il[i:i + 6], beam) for i in range(0, N, 6)]
    from multiprocessing import forkserver
    if getattr(forkserver._forkserver, "_forkserver_pid", None) is not None:
        with mp.get_context("forkserver").Pool(min(46, os.cpu_count() or 1)) as pool:
            res = pool.map(_oracle_decode_chunk, chunks)
    else:
        res = [_oracle_decode_chunk(c) for c in chunks]
    r_beam = [s for r in res for s in r[0]]
    r_scores = np.array([s for r in res for s in r[1]])
    r_greedy = [s for r in res for s in r[2]]
    assert len(r_beam) == N
    assert g_greedy == r_greedy
    assert g_beam == r_beam
    assert np.max(np.abs((g_scores - r_scores) / r_scores)) < 1e-12
    assert sum(len(s) for s in r_beam) > 10 * N          # a real decode workload, not empty hypotheses


def test_csv_store_feeds_fit_generator_like_hand_built_arrays(device, tmp_path):
    """the reference's on-disk layout (util/mix_data.py:71-82,152-176: per-file 100 fps audio CSVs, one skeletal
    CSV with a file_number column, a label CSV with Id / Sequence) through CsvStore -> DataGenerator -> fit_generator, against
    a second model trained on the same batches assembled by hand from the raw numbers."""
    import pandas as pd
    from mgr_amd import keras_like as K
    from mgr_amd.configs import fusion_spec
    from mgr_amd.datagen import SKELETAL_COLUMNS
    from mgr_amd.keras_like import Adam, Model
    from mgr_amd.multimodal_fusion.data_generator import DataGenerator
    K.set_learning_phase(1)
    rng = np.random.default_rng(3)
    root = tmp_path / "data"
    (root / "train_audio").mkdir(parents=True)
    mb, maxlen, steps = 2, 48, 2
    ids = [4, 9, 11, 17, 23]
    audio, skel, labs = {}, {}, {}
    rows = []
    for fid in ids:
        n100 = int(rng.integers(150, 300))                         # 100 fps MFCC rows -> every 5th is kept
        audio[fid] = rng.standard_normal((n100, 39)) * 3.0
        df = pd.DataFrame(audio[fid], columns=[str(i) for i in range(39)])
        df["file_number"] = fid
        df.to_csv(root / "train_audio" / ("audio_%d.csv" % fid), index=False)
        n20 = int(rng.integers(30, 60))
        skel[fid] = rng.standard_normal((n20, 20)) * 5.0 + 2.0
        for r in skel[fid]:
            rows.append(dict(zip(SKELETAL_COLUMNS, r), file_number=fid))
        labs[fid] = [int(v) for v in rng.integers(1, 21, size=int(rng.integers(2, 5)))]
    pd.DataFrame(rows).to_csv(root / "Training_set_skeletal.csv", index=False)
    pd.DataFrame({"Id": ids, "Sequence": [" ".join(map(str, labs[f])) for f in ids]}).to_csv(root / "training_oov.csv", index=False)

    def model():
        m = Model(fusion_spec(h_audio=32, h_skeletal=16, h_fusion=8), device=device, seed=21)
        m.compile(loss={'ctc': lambda a, b: b}, optimizer=Adam(lr=1e-3, clipvalue=0.5, decay=1e-5))
        return m

    gen = DataGenerator(minibatch_size=mb, numfeats_skeletal=20, numfeats_speech=39, maxlen=maxlen, nb_classes=22,
                        dataset='train', val_split=0.2, data_root=str(root))
    assert type(gen.store).__name__ == "CsvStore"
    order = list(gen.get_file_list(True))
    # the reference's split: sorted ids, random.seed(10) shuffle, first 80 %, whole minibatches only
    exp = sorted(ids)
    random.seed(10)
    random.shuffle(exp)
    exp = exp[:int(len(ids) * 0.8)]
    exp = exp[:len(exp) - len(exp) % mb]
    assert order == exp and len(order) == mb * steps
    a = model()
    hist = a.fit_generator(generator=gen.next_train(), steps_per_epoch=steps, epochs=1, verbose=0)
    # the same batches by hand: audio every 5th row, skeletal z-scored over the WHOLE table, zero post-padding, labels padded -1
    allsk = np.concatenate([skel[f] for f in ids])          # file order of the CSV
    mu, sd = allsk.mean(0), allsk.std(0)
    b = model()
    losses = []
    for s in range(steps):
        xa, xs_ = np.zeros((mb, maxlen, 39)), np.zeros((mb, maxlen, 20))
        lab, ll = -np.ones((mb, 35)), np.zeros((mb, 1))
        for i, fid in enumerate(order[s * mb:(s + 1) * mb]):
            fa = audio[fid][::5][:maxlen]
            fs = ((skel[fid] - mu) / sd)[:maxlen]
            xa[i, :len(fa)], xs_[i, :len(fs)] = fa, fs
            lab[i, :len(labs[fid])], ll[i, 0] = labs[fid], len(labs[fid])
        x = {"the_input_audio": xa, "the_input_skeletal": xs_, "the_labels": lab,
             "input_length": np.full((mb, 1), maxlen - 2.0), "label_length": ll}
        losses.append(b.train_on_batch(x, None))
    assert np.all(np.isfinite(losses))
    # (the store z-scores in one vectorised pass, the hand-built arrays per file: fp64 values agree to ~1e-15, so a few fp32
    # inputs may differ in their last bit - losses to 1e-5 relative, weights to a fraction of one Adam step)
    assert abs(hist.history["loss"][0] - float(np.mean(losses))) < 1e-5 * abs(float(np.mean(losses)))
    wa, wb = a.get_weights_dict(), b.get_weights_dict()
    for k in wa:
        assert np.allclose(wa[k], wb[k], rtol=0, atol=1e-4), k
    assert not np.array_equal(wa["dense/W"], model().get_weights_dict()["dense/W"])      # ... and training moved them
