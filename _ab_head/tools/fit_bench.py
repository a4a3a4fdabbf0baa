#!/usr/bin/env python3
"""End-to-end training-loop rate through the Keras-like facade (host batches included): fusion model at the reference
sizes, DataGenerator over a synthetic store, fit_generator for a few dozen steps.  Complements bench.py, whose inputs are
resident in HBM."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mgr_amd  # noqa: E402,F401
from mgr_amd import keras_like as K  # noqa: E402
from mgr_amd.keras_like import Adam, Callback, Model  # noqa: E402
from mgr_amd.configs import fusion_spec  # noqa: E402
from mgr_amd.multimodal_fusion.data_generator import DataGenerator  # noqa: E402

mb, maxlen, steps = 64, 1900, int(os.environ.get("FIT_STEPS", 24))
gen = DataGenerator(minibatch_size=mb, numfeats_skeletal=20, numfeats_speech=39, maxlen=maxlen, dataset='train', val_split=0.0,
                    nb_classes=22, synthetic_files=mb * 4)
K.set_learning_phase(1)
model = Model(fusion_spec())
model.compile(loss={'ctc': lambda a, b: b}, optimizer=Adam(lr=1e-4, clipvalue=0.5, decay=1e-5))


class Timer(Callback):
    def on_train_begin(self, logs=None):
        self.t = []

    def on_epoch_begin(self, epoch, logs=None):
        self.t0 = time.perf_counter()

    def on_epoch_end(self, epoch, logs=None):
        self.t.append(time.perf_counter() - self.t0)


# time the generator alone (host work per batch; the store's file cache is warm after one pass over the files)
g = gen.next_train()
for _ in range(4):
    next(g)
t0 = time.perf_counter()
for _ in range(4):
    next(g)
t_gen = (time.perf_counter() - t0) / 4

# host-side phase timers
from mgr_amd.engine import Engine  # noqa: E402
ACC = {}


def timed(cls, name):
    fn = getattr(cls, name)

    def wrap(*a, **k):
        t = time.perf_counter()
        try:
            return fn(*a, **k)
        finally:
            ACC[name] = ACC.get(name, 0.0) + time.perf_counter() - t
    setattr(cls, name, wrap)


for nme in ("_upload_inputs", "_upload_labels", "read_loss", "enqueue_train_step"):
    timed(Engine, nme)
tm = Timer()
model.fit_generator(generator=gen.next_train(), steps_per_epoch=steps, epochs=3, callbacks=[tm], verbose=0)
best = min(tm.t[1:])
n_all = 3 * steps
print("host ms/step: " + ", ".join("%s %.1f" % (k, v / n_all * 1e3) for k, v in ACC.items())
      + "  (enqueue_train_step includes the two uploads)")
print("generator alone: %.1f ms/batch;  fit_generator: %.1f ms/step = %.0f frames/s (best of epochs 2-3, %d steps each)"
      % (t_gen * 1e3, best / steps * 1e3, mb * maxlen * steps / best, steps))
