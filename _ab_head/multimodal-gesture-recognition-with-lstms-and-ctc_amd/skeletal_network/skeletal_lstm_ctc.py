"""Skeletal network: DataGenerator, ctc_lambda_func, model and training entry point in ONE module, like the
reference script (skeletal_network/skeletal_lstm_ctc.py:34-424): GaussianNoise(.5) -> BiLSTM(300, drop .6) x2 + residual
-> Dropout(.6) -> Dense(22) -> softmax -> CTC; Adam(1e-4, clipvalue .5, decay 1e-5)."""
import argparse
import os
import time

from .. import keras_like as K
from ..configs import skeletal_spec
from ..datagen import BaseDataGenerator, CsvStore, SyntheticStore
from ..keras_like import Adam, EarlyStopping, Model, ModelCheckpoint
from ..multimodal_fusion.losses import ctc_lambda_func  # noqa: F401  (the reference defines its own copy, :257-268)


class DataGenerator(BaseDataGenerator):
    streams = (('the_input', 'skeletal', 'numfeats'),)
    model_json_name = "sk_ctc_lstm_model.json"
    model_weights_name = "sk_ctc_lstm_weights.h5"

    def __init__(self, minibatch_size, numfeats, maxlen, val_split, nb_classes, absolute_max_sequence_len=28,
                 in_file='Training_set_skeletal.csv', train_lab_file='../training.csv', synthetic_files=None,
                 seed=20131902, rank=0, world=1):
        self.numfeats = numfeats
        if synthetic_files is None and os.path.isfile(in_file) and os.path.isfile(train_lab_file):
            store = CsvStore(None, in_file, train_lab_file)
        else:
            n = synthetic_files if synthetic_files is not None else 393
            store = SyntheticStore(n, {'skeletal': (numfeats, 1.0)}, maxlen, nb_classes, seed=seed,
                                   lmax=min(20, absolute_max_sequence_len))
        self._setup(minibatch_size, maxlen, nb_classes, 'train', val_split, absolute_max_sequence_len, store, rank=rank, world=world)


def build_model(maxlen, numfeats, nb_classes, lab_seq_len=28, load_previous='no', units=300, layers=2, device=0,
                seed=1234):
    """The reference builds this graph at module level (:298-394); here it is a function with the same pieces."""
    K.set_learning_phase(1)
    model = Model(skeletal_spec(numfeats, nb_classes, units, layers), device=device, seed=seed)
    adam = Adam(lr=0.0001, clipvalue=0.5, decay=1e-5)
    if load_previous == 'yes':
        with open('sk_ctc_lstm_model.json') as f:
            model = K.model_from_json(f.read(), device=device)
        model.load_weights("sk_ctc_lstm_weights_best.h5")
        print("Loaded model from disk")
    model.summary()
    model.compile(loss={'ctc': lambda y_true, y_pred: y_pred}, optimizer=adam)
    return model


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--minibatch-size", type=int, default=2)
    ap.add_argument("--maxlen", type=int, default=1900)
    ap.add_argument("--epochs", type=int, default=500)
    ap.add_argument("--load-previous", default="no")
    ap.add_argument("--synthetic-files", type=int, default=None)
    a = ap.parse_args(argv)
    minibatch_size, val_split, maxlen, nb_classes, numfeats = a.minibatch_size, 0.2, a.maxlen, 22, 20
    data_gen = DataGenerator(minibatch_size=minibatch_size, numfeats=numfeats, maxlen=maxlen, val_split=val_split,
                             nb_classes=nb_classes, synthetic_files=a.synthetic_files)
    model = build_model(maxlen, numfeats, nb_classes, data_gen.absolute_max_sequence_len, a.load_previous)
    earlystopping = EarlyStopping(monitor='val_loss', patience=20, verbose=1)
    checkpoint = ModelCheckpoint("sk_ctc_lstm_weights_best.h5", monitor='val_loss', verbose=1, save_best_only=True,
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
