"""Fusion network builder + training entry point (reference multimodal_fusion/multimodal.py:33-273).

``build_model`` keeps the reference signature.  The graph the reference assembles with Keras layers (two frozen
2-layer BiLSTM encoders taken from the pre-trained unimodal models, residual adds, concat, BiLSTM(100) with input
dropout .5 and maxnorm(3), Dropout(.5), Dense, softmax, CTC Lambda, Adam(1e-4, clipvalue .5, decay 1e-5)) is
described by a NetworkSpec and executed by the MI355X engine.
"""
import argparse
import os
import time

from .. import keras_like as K
from ..configs import fusion_spec
from ..keras_like import Adam, EarlyStopping, Model, ModelCheckpoint
from .data_generator import DataGenerator
from .losses import ctc_lambda_func  # noqa: F401  (re-exported like the reference module does)


def layer_trainable(l, freeze, verbose=False, bidir_fix=True):
    """Freeze / unfreeze a Bidirectional layer the way the reference's workaround does:
    the wrapper's own flag is set to ``freeze`` while the two wrapped LSTMs get ``not freeze``."""
    l.trainable = freeze
    if bidir_fix and getattr(l, "kind", None) == "Bidirectional":
        l.backward_layer.trainable = not freeze
        l.forward_layer.trainable = not freeze
    if verbose:
        print("{} {}".format("Froze" if freeze else "Unfroze", l.name))


def _load_encoder(model_file, weights_file):
    """Pre-trained unimodal model (JSON + weights) if present on disk, else None."""
    if os.path.isfile(model_file) and os.path.isfile(weights_file):
        with open(model_file) as f:
            m = K.model_from_json(f.read())
        m.load_weights(weights_file)
        return m
    return None


def build_model(maxlen, numfeats_speech, numfeats_skeletal, nb_classes, lab_seq_len, speech_model=None,
                skeletal_model=None, device=0, seed=1234):
    """Returns the compiled fusion model.  The unimodal encoders are taken from ``speech_model`` /
    ``skeletal_model`` when given, else from the reference's fixed checkpoint paths when those files exist,
    else they keep their seeded initialisation (no trained weights ship with the repository)."""
    K.set_learning_phase(1)
    if skeletal_model is None:
        skeletal_model = _load_encoder('../skeletal_network/sk_ctc_lstm_model.json',
                                       '../skeletal_network/sk_ctc_lstm_weights_best.h5')
    if speech_model is None:
        speech_model = _load_encoder('../audio_network/sp_ctc_lstm_model.json',
                                     '../audio_network/sp_ctc_lstm_weights_best.h5')
    h_a = speech_model.spec.streams[0]["layers"][0]["H"] if speech_model is not None else 500
    h_s = skeletal_model.spec.streams[0]["layers"][0]["H"] if skeletal_model is not None else 300
    spec = fusion_spec(numfeats_speech, numfeats_skeletal, nb_classes, h_a, h_s, 100)
    model = Model(spec, device=device, seed=seed)
    # transplant the two BiLSTM layers of each unimodal network (speech_model.layers[2], [3] in the reference)
    for uni, stream in ((speech_model, "the_input_audio"), (skeletal_model, "the_input_skeletal")):
        if uni is None:
            continue
        src = uni.get_weights_dict()
        uname = uni.spec.streams[0]["name"]
        moved = {}
        for k, v in src.items():
            if k.startswith(uname + "/"):
                moved[stream + k[len(uname):]] = v
        model.set_weights_dict(moved)
    # freeze every Bidirectional layer of the two encoders (reference :135-148)
    for l in model.layers:
        if l.kind == "Bidirectional" and l.weight_prefix != "fusion":
            layer_trainable(l, freeze=True, verbose=True)
    model.summary()
    adam = Adam(lr=0.0001, clipvalue=0.5, decay=1e-5)
    # the loss is computed by the CTC layer itself; Keras' dummy lambda loss just forwards it
    model.compile(loss={'ctc': lambda y_true, y_pred: y_pred}, optimizer=adam)
    return model


def main(argv=None):
    ap = argparse.ArgumentParser(description="train the multimodal fusion network (synthetic data unless ../data exists)")
    ap.add_argument("--minibatch-size", type=int, default=2)
    ap.add_argument("--maxlen", type=int, default=1900)
    ap.add_argument("--epochs", type=int, default=500)
    ap.add_argument("--synthetic-files", type=int, default=None)
    a = ap.parse_args(argv)
    minibatch_size, val_split, maxlen, nb_classes = a.minibatch_size, 0.2, a.maxlen, 22
    numfeats_speech, numfeats_skeletal = 39, 20
    data_gen = DataGenerator(minibatch_size=minibatch_size, numfeats_skeletal=numfeats_skeletal,
                             numfeats_speech=numfeats_speech, maxlen=maxlen, dataset='train', val_split=val_split,
                             nb_classes=nb_classes, synthetic_files=a.synthetic_files)
    lab_seq_len = data_gen.absolute_max_sequence_len
    model = build_model(maxlen, numfeats_speech, numfeats_skeletal, nb_classes, lab_seq_len)
    earlystopping = EarlyStopping(monitor='val_loss', patience=20, verbose=1)  # constructed but unused, as upstream
    checkpoint = ModelCheckpoint("multimodal_ctc_lstm_weights_best.h5", monitor='val_loss', verbose=1,
                                 save_best_only=True, save_weights_only=True, mode='auto')
    print('Start training.')
    start_time = time.time()
    model.fit_generator(generator=data_gen.next_train(),
                        steps_per_epoch=(data_gen.get_size(train=True) // minibatch_size), epochs=a.epochs,
                        validation_data=data_gen.next_val(),
                        validation_steps=(data_gen.get_size(train=False) // minibatch_size),
                        callbacks=[checkpoint, data_gen])
    print("--- Training time: %s seconds ---" % (time.time() - start_time))
    return model


if __name__ == '__main__':
    main()
