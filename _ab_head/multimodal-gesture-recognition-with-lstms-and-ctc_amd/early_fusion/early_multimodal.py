"""Early-fusion network builder + training entry point (reference early_fusion/early_multimodal.py:44-503).

The reference concatenates the two noisy inputs along the feature axis (39 + 20 = 59) and trains
2 x Bidirectional(LSTM(500), dropout .4) + residual add -> Dropout(.4) -> Dense -> softmax -> CTC with
Adam(1e-4, clipvalue .5, decay 1e-5) (:321-418).  Zero-mean iid noise on a concatenation equals the concatenation
of the two noises, so the network is a single-stream NetworkSpec whose stream reads both inputs.
"""
import argparse
import time

from .. import keras_like as K
from ..configs import early_fusion_spec
from ..datagen import BaseDataGenerator, SyntheticStore
from ..keras_like import Adam, EarlyStopping, Model, ModelCheckpoint
from ..multimodal_fusion.losses import ctc_lambda_func  # noqa: F401  (the module defines its own copy upstream, :310-319)

minibatch_size = 2
val_split = 0.2
maxlen = 1900
nb_classes = 22
nb_epoch = 500
numfeats_speech = 39
numfeats_skeletal = 20
stamp = 'early_multimodal'


class DataGenerator(BaseDataGenerator):
    """early_multimodal.py:44-304: same batch dict as the late-fusion generator (two inputs, labels, lengths)."""
    streams = (('the_input_audio', 'audio', 'numfeats_speech'), ('the_input_skeletal', 'skeletal', 'numfeats_skeletal'))
    model_json_name = stamp + ".json"
    model_weights_name = stamp + ".h5"

    def __init__(self, minibatch_size, numfeats_skeletal, numfeats_speech, maxlen, val_split, nb_classes,
                 absolute_max_sequence_len=28, store=None, synthetic_files=470, seed=20131900, rank=0, world=1):
        self.numfeats_speech = numfeats_speech
        self.numfeats_skeletal = numfeats_skeletal
        if store is None:
            store = SyntheticStore(synthetic_files, {'audio': (numfeats_speech, 3.0), 'skeletal': (numfeats_skeletal, 1.0)},
                                   maxlen, nb_classes, seed=seed, lmax=min(20, absolute_max_sequence_len))
        self._setup(minibatch_size, maxlen, nb_classes, 'train', val_split, absolute_max_sequence_len, store, rank=rank, world=world)


def build_net(maxlen=maxlen, numfeats_speech=numfeats_speech, numfeats_skeletal=numfeats_skeletal,
              nb_classes=nb_classes, device=0, seed=1234):
    """Compiled early-fusion model (reference :321-418; the upstream function reads these sizes from module globals)."""
    K.set_learning_phase(1)
    model = Model(early_fusion_spec(numfeats_speech, numfeats_skeletal, nb_classes), device=device, seed=seed)
    model.summary()
    adam = Adam(lr=0.0001, clipvalue=0.5, decay=1e-5)
    model.compile(loss={'ctc': lambda y_true, y_pred: y_pred}, optimizer=adam)
    with open(stamp + ".json", "w") as json_file:
        json_file.write(model.to_json())
    return model


def load_model(device=0):
    """Resume from ``early_multimodal.json`` / ``.h5`` (reference :421-440; note: no lr decay on resume)."""
    with open(stamp + '.json') as f:
        model = K.model_from_json(f.read(), device=device)
    model.load_weights(stamp + '.h5')
    adam = Adam(lr=0.0001, clipvalue=0.5)
    print("Loaded model from disk")
    model.compile(loss={'ctc': lambda y_true, y_pred: y_pred}, optimizer=adam)
    return model


def main(argv=None):
    ap = argparse.ArgumentParser(description="train the early-fusion network (synthetic data)")
    ap.add_argument("--minibatch-size", type=int, default=minibatch_size)
    ap.add_argument("--maxlen", type=int, default=maxlen)
    ap.add_argument("--epochs", type=int, default=nb_epoch)
    ap.add_argument("--synthetic-files", type=int, default=470)
    ap.add_argument("--load-previous", default="no")
    a = ap.parse_args(argv)
    data_gen = DataGenerator(minibatch_size=a.minibatch_size, numfeats_skeletal=numfeats_skeletal,
                             numfeats_speech=numfeats_speech, maxlen=a.maxlen, val_split=val_split,
                             nb_classes=nb_classes, synthetic_files=a.synthetic_files)
    model = load_model() if a.load_previous == 'yes' else build_net(a.maxlen)
    earlystopping = EarlyStopping(monitor='val_loss', patience=20, verbose=1)
    checkpoint = ModelCheckpoint(stamp + ".h5", monitor='val_loss', verbose=1, save_best_only=True,
                                 save_weights_only=True, mode='auto')
    print('Start training.')
    start_time = time.time()
    model.fit_generator(generator=data_gen.next_train(),
                        steps_per_epoch=(data_gen.get_size(train=True) // a.minibatch_size), epochs=a.epochs,
                        validation_data=data_gen.next_val(),
                        validation_steps=(data_gen.get_size(train=False) // a.minibatch_size),
                        callbacks=[earlystopping, checkpoint, data_gen])
    print("--- Training time: %s seconds ---" % (time.time() - start_time))
    return model


if __name__ == '__main__':
    main()
