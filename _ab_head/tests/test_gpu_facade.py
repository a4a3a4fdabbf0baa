"""-m gpu: the reference's script-level flow through the Keras-like façade: build_model -> fit_generator with the
DataGenerator + ModelCheckpoint callbacks -> save/load -> predict_generator -> decode_batch."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_fusion_script_flow(device, tmp_path, monkeypatch):
    import mgr_amd  # noqa: F401
    from mgr_amd import decoding, keras_like as K
    from mgr_amd.configs import fusion_spec
    from mgr_amd.keras_like import Adam, Model, ModelCheckpoint
    from mgr_amd.multimodal_fusion.data_generator import DataGenerator
    from mgr_amd.multimodal_fusion.sequence_decoding import decode_batch
    monkeypatch.chdir(tmp_path)
    decoding._DEV[0] = device
    mb, maxlen = 4, 48
    gen = DataGenerator(minibatch_size=mb, numfeats_skeletal=20, numfeats_speech=39, maxlen=maxlen, dataset='train',
                        val_split=0.2, nb_classes=22, synthetic_files=21)
    K.set_learning_phase(1)
    model = Model(fusion_spec(h_audio=32, h_skeletal=16, h_fusion=8), device=device)
    model.compile(loss={'ctc': lambda a, b: b}, optimizer=Adam(lr=1e-3, clipvalue=0.5, decay=1e-5))
    w0 = model.get_weights_dict()
    ck = ModelCheckpoint("multimodal_ctc_lstm_weights_best.h5", monitor='val_loss', verbose=0, save_best_only=True,
                         save_weights_only=True, mode='auto')
    hist = model.fit_generator(generator=gen.next_train(), steps_per_epoch=gen.get_size(True) // mb, epochs=2,
                               validation_data=gen.next_val(), validation_steps=gen.get_size(False) // mb,
                               callbacks=[ck, gen], verbose=0)
    assert len(hist.history["loss"]) == 2 and np.all(np.isfinite(hist.history["loss"])) and "val_loss" in hist.history
    assert os.path.exists("multimodal_ctc_lstm_weights_best.h5") and os.path.exists("multimodal_ctc_blstm_model.json")
    w1 = model.get_weights_dict()
    assert not np.array_equal(w0["fusion/fwd/W"], w1["fusion/fwd/W"])          # trainable layer moved
    assert np.array_equal(w0["the_input_audio/l0/fwd/U"], w1["the_input_audio/l0/fwd/U"])  # frozen encoder did not
    # max-norm(3) holds on the fusion input kernel
    assert np.all(np.sqrt((w1["fusion/fwd/W"] ** 2).sum(0)) <= 3.0 + 1e-4)
    # decode script flow (sequence_decoding.py:81-127)
    K.set_learning_phase(0)
    loaded = K.model_from_json(open("multimodal_ctc_blstm_model.json").read(), device=device)
    loaded.load_weights("multimodal_ctc_blstm_weights.h5")
    pred_model = Model(inputs=loaded.input, outputs=loaded.get_layer('softmax').output)
    val = DataGenerator(minibatch_size=mb, numfeats_skeletal=20, numfeats_speech=39, maxlen=maxlen, dataset='val',
                        nb_classes=22, synthetic_files=8)
    preds = pred_model.predict_generator(generator=val.next_val(), steps=val.get_size(False) // mb, verbose=0)
    assert preds.shape == (8, maxlen, 22) and np.allclose(preds.sum(-1), 1, atol=1e-5)
    # the deterministic (phase 0) prediction equals the trained model's
    ref = model.predict_generator(generator=DataGenerator(mb, 20, 39, maxlen, 22, 'val', synthetic_files=8).next_val(),
                                  steps=2)
    assert np.allclose(preds, ref, atol=1e-6)
    res = decode_batch(preds, val.get_file_list(False))
    assert len(res) == 8 and os.path.exists("final_ctc_recout.mlf")
    K.set_learning_phase(1)


def test_unimodal_builders(device, tmp_path, monkeypatch):
    import mgr_amd  # noqa: F401
    from mgr_amd.audio_network import speech_lstm_ctc_words as audio
    from mgr_amd.skeletal_network import skeletal_lstm_ctc as skel
    monkeypatch.chdir(tmp_path)
    m = audio.build_model(40, 39, 44, 150, 'no', units=16, device=device)
    g = audio.DataGenerator(minibatch_size=2, numfeats=39, maxlen=40, dataset='train', val_split=0.2, nb_classes=44,
                            synthetic_files=10)
    g.store.lmax = 4
    x, y = next(g.next_train())
    l1 = m.train_on_batch(x, y)
    l2 = m.train_on_batch(x, y)
    assert np.isfinite(l1) and np.isfinite(l2)
    s = skel.build_model(40, 20, 22, 28, 'no', units=12, device=device)
    sg = skel.DataGenerator(minibatch_size=2, numfeats=20, maxlen=40, val_split=0.2, nb_classes=22, synthetic_files=10)
    x, y = next(sg.next_train())
    assert np.isfinite(s.train_on_batch(x, y)) and np.isfinite(s.test_on_batch(x, y))


def test_early_fusion_builder(device, tmp_path, monkeypatch):
    """early_fusion/early_multimodal.py flow: two-input dict -> concatenated 59-d stream -> train / predict / decode."""
    import mgr_amd  # noqa: F401
    from mgr_amd import decoding, keras_like as K
    from mgr_amd.early_fusion import early_multimodal as early
    from mgr_amd.early_fusion.sequence_decoding import decode_batch
    monkeypatch.chdir(tmp_path)
    decoding._DEV[0] = device
    maxlen = 40
    m = early.build_net(maxlen=maxlen, device=device)
    assert os.path.exists("early_multimodal.json")
    assert [l.name for l in m.layers[:2]] == ["the_input_audio", "the_input_skeletal"]
    assert sum(int(np.prod(sh)) for _, sh, _, _ in m.spec.weight_table()) == 2 * (59 + 500 + 1) * 2000 + 2 * (1000 + 500 + 1) * 2000 + 1000 * 22 + 22
    g = early.DataGenerator(minibatch_size=2, numfeats_skeletal=20, numfeats_speech=39, maxlen=maxlen, val_split=0.2,
                            nb_classes=22, synthetic_files=10)
    g.store.lmax = 4
    x, y = next(g.next_train())
    l1 = m.train_on_batch(x, y)
    l2 = m.train_on_batch(x, y)
    assert np.isfinite(l1) and np.isfinite(l2)
    m.save_weights("early_multimodal.h5")
    K.set_learning_phase(0)
    m2 = early.load_model(device=device)
    p1, p2 = m.predict_on_batch(x), m2.predict_on_batch(x)
    # (training and inference engines may run different scan kernels: same arithmetic, different summation order)
    assert p1.shape == (2, maxlen, 22) and np.allclose(p1, p2, rtol=0, atol=1e-6)
    res = decode_batch(p1, [1, 228])
    assert len(res) == 2 and open("final_ctc_recout.mlf").read().count(".rec") == 1   # 228 is on the ignore list
    K.set_learning_phase(1)


def test_fit_generator_prefetch_equals_plain_loop(device):
    """fit_generator hands the engine the NEXT batch so that its frozen-encoder pass overlaps the current step; losses and
    weights must be identical to feeding the same batches one at a time."""
    import mgr_amd  # noqa: F401
    from mgr_amd import keras_like as K
    from mgr_amd.configs import fusion_spec
    from mgr_amd.keras_like import Adam, Model
    from mgr_amd.multimodal_fusion.data_generator import DataGenerator
    K.set_learning_phase(1)
    mb, maxlen, steps = 4, 40, 6

    def make():
        gen = DataGenerator(minibatch_size=mb, numfeats_skeletal=20, numfeats_speech=39, maxlen=maxlen, dataset='train',
                            val_split=0.0, nb_classes=22, synthetic_files=mb * steps)
        gen.store.lmax = 5
        m = Model(fusion_spec(h_audio=32, h_skeletal=16, h_fusion=8), device=device, seed=11)
        m.compile(loss={'ctc': lambda a, b: b}, optimizer=Adam(lr=1e-3, clipvalue=0.5, decay=1e-5))
        return gen, m

    gen_a, a = make()
    hist = a.fit_generator(generator=gen_a.next_train(), steps_per_epoch=steps, epochs=1, verbose=0)
    gen_b, b = make()
    g = gen_b.next_train()
    losses = []
    for _ in range(steps):
        x, y = next(g)
        losses.append(b.train_on_batch(x, y))
    assert abs(hist.history["loss"][0] - float(np.mean(losses))) < 1e-6
    wa, wb = a.get_weights_dict(), b.get_weights_dict()
    for k in wa:
        assert np.array_equal(wa[k], wb[k]), k
