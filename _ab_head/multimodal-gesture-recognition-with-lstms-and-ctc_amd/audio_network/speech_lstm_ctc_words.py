"""Audio (MFCC) network builder + training entry point (reference audio_network/speech_lstm_ctc_words.py:32-193):
GaussianNoise(.5) -> BiLSTM(500, drop .4) -> BiLSTM(500, drop .5) -> add -> Dropout(.5) -> Dense(44) -> softmax -> CTC."""
import argparse
import os
import time

from .. import keras_like as K
from ..configs import audio_spec
from ..keras_like import Adam, EarlyStopping, Model, ModelCheckpoint
from .data_generator import DataGenerator
from .losses import ctc_lambda_func  # noqa: F401


def build_model(maxlen, numfeats, nb_classes, lab_seq_len, resume_training, units=500, device=0, seed=1234):
    """Same signature as the reference; ``resume_training == 'yes'`` reloads sp_ctc_lstm_model.json + best weights."""
    K.set_learning_phase(1)
    model = Model(audio_spec(numfeats, nb_classes, units, 2), device=device, seed=seed)
    model.summary()
    adam = Adam(lr=0.0001, clipvalue=0.5)
    if resume_training == 'yes':
        with open('sp_ctc_lstm_model.json') as f:
            model = K.model_from_json(f.read(), device=device)
        model.load_weights("sp_ctc_lstm_weights_best.h5")
        adam = Adam(lr=0.0001, clipvalue=0.5)
        print("Loaded model from disk")
    model.compile(loss={'ctc': lambda y_true, y_pred: y_pred}, optimizer=adam)
    return model


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--minibatch-size", type=int, default=2)
    ap.add_argument("--maxlen", type=int, default=1900)
    ap.add_argument("--epochs", type=int, default=500)
    ap.add_argument("--resume", default="no")
    ap.add_argument("--synthetic-files", type=int, default=None)
    a = ap.parse_args(argv)
    minibatch_size, val_split, maxlen, nb_classes, numfeats = a.minibatch_size, 0.2, a.maxlen, 44, 39
    data_gen = DataGenerator(minibatch_size=minibatch_size, numfeats=numfeats, maxlen=maxlen, dataset='train',
                             val_split=val_split, nb_classes=nb_classes, synthetic_files=a.synthetic_files)
    lab_seq_len = data_gen.absolute_max_sequence_len
    model = build_model(maxlen, numfeats, nb_classes, lab_seq_len, a.resume)
    earlystopping = EarlyStopping(monitor='val_loss', patience=20, verbose=1)  # unused upstream too
    checkpoint = ModelCheckpoint("sp_ctc_lstm_weights_best.h5", monitor='val_loss', verbose=1, save_best_only=True,
                                 save_weights_only=True, mode='auto')
    print('Start training.')
    start_time = time.time()
    model.fit_generator(generator=data_gen.next_train(), steps_per_epoch=(data_gen.get_size(train=True) // minibatch_size),
                        epochs=a.epochs, validation_data=data_gen.next_val(),
                        validation_steps=(data_gen.get_size(train=False) // minibatch_size), callbacks=[checkpoint, data_gen])
    print("--- Training time: %s seconds ---" % (time.time() - start_time))
    return model


if __name__ == '__main__':
    main()
