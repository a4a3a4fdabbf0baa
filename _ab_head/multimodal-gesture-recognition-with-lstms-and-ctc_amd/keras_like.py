"""The slice of the Keras 2.1.4 API the reference's scripts call, backed by the MI355X engine.

Covered call sites: ``model.compile / fit_generator / predict_generator / summary / to_json /
save_weights / load_weights / get_layer / layers / input`` (multimodal_fusion/multimodal.py:206-269,
multimodal_fusion/sequence_decoding.py:99-123, multimodal_fusion/data_generator.py:317-321),
``ModelCheckpoint`` / ``EarlyStopping`` (multimodal.py:248-258), ``Adam(lr, clipvalue, decay)`` (:206-208),
``model_from_json`` (:74-85).  Graph construction is replaced by a ``NetworkSpec``; the arithmetic runs in
libmgr.so (HIP) - there is no CPU execution path.

Weights files: h5py is not available in the build image, so ``save_weights`` writes a numpy ``.npz``
archive under the exact file name given (Keras layouts, Keras weight-list order); ``load_weights`` reads it.
"""
import json
import sys
import time

import numpy as np

from .spec import NetworkSpec

_LEARNING_PHASE = [1]


def set_learning_phase(value):
    """K.set_learning_phase (multimodal.py:66 forces 1 for training AND validation; decoders set 0)."""
    _LEARNING_PHASE[0] = int(value)


def learning_phase():
    return _LEARNING_PHASE[0]


# ------------------------------------------------------------------------------------------ optimizer
class Adam:
    """keras.optimizers.Adam as configured by the reference (lr=1e-4, clipvalue=.5[, decay=1e-5])."""

    def __init__(self, lr=0.001, beta_1=0.9, beta_2=0.999, epsilon=None, decay=0.0, clipvalue=None, **kwargs):
        self.lr = lr
        self.beta_1 = beta_1
        self.beta_2 = beta_2
        self.epsilon = 1e-7 if epsilon is None else epsilon  # K.epsilon()
        self.decay = decay
        self.clipvalue = clipvalue
        if "clipnorm" in kwargs and kwargs["clipnorm"]:
            raise NotImplementedError("clipnorm is not used on the reference's training path")

    def config(self):
        return {"lr": self.lr, "beta_1": self.beta_1, "beta_2": self.beta_2, "epsilon": self.epsilon,
                "decay": self.decay, "clipvalue": self.clipvalue or 0.0}


class RMSprop:
    """Accepted by compile() for the decode scripts (sequence_decoding.py:112-115), which never train: the device optimizer
    is Adam (the only one on the reference's training path), so training after compile(optimizer=RMSprop(...)) is refused
    (Model._require_trainable) instead of silently running Adam with default settings."""

    def __init__(self, lr=0.001, **kwargs):
        self.lr = lr

    def config(self):
        return {"lr": self.lr}


# ------------------------------------------------------------------------------------------ callbacks
class Callback:
    def __init__(self):
        self.model = None

    def set_model(self, model):
        self.model = model

    def on_train_begin(self, logs=None):
        pass

    def on_epoch_begin(self, epoch, logs=None):
        pass

    def on_epoch_end(self, epoch, logs=None):
        pass

    def on_train_end(self, logs=None):
        pass


class ModelCheckpoint(Callback):
    def __init__(self, filepath, monitor="val_loss", verbose=0, save_best_only=False, save_weights_only=False,
                 mode="auto", period=1):
        super().__init__()
        self.filepath = filepath
        self.monitor = monitor
        self.verbose = verbose
        self.save_best_only = save_best_only
        self.save_weights_only = save_weights_only
        self.best = np.inf
        self.sign = -1.0 if (mode == "max" or (mode == "auto" and ("acc" in monitor))) else 1.0

    def on_epoch_end(self, epoch, logs=None):
        logs = logs or {}
        path = self.filepath.format(epoch=epoch + 1, **logs)
        cur = logs.get(self.monitor)
        if self.save_best_only:
            if cur is None:
                return
            if self.sign * cur < self.best:
                if self.verbose:
                    print("Epoch %05d: %s improved from %0.5f to %0.5f, saving model to %s"
                          % (epoch + 1, self.monitor, self.sign * self.best, cur, path))
                self.best = self.sign * cur
                self.model.save_weights(path)
            elif self.verbose:
                print("Epoch %05d: %s did not improve" % (epoch + 1, self.monitor))
        else:
            self.model.save_weights(path)


class EarlyStopping(Callback):
    def __init__(self, monitor="val_loss", min_delta=0, patience=0, verbose=0, mode="auto"):
        super().__init__()
        self.monitor, self.min_delta, self.patience, self.verbose = monitor, min_delta, patience, verbose
        self.best = np.inf
        self.wait = 0

    def on_epoch_end(self, epoch, logs=None):
        cur = (logs or {}).get(self.monitor)
        if cur is None:
            return
        if cur < self.best - self.min_delta:
            self.best, self.wait = cur, 0
        else:
            self.wait += 1
            if self.wait >= self.patience:
                self.model.stop_training = True
                if self.verbose:
                    print("Epoch %05d: early stopping" % (epoch + 1))


class History(Callback):
    def __init__(self):
        super().__init__()
        self.history = {}
        self.epoch = []

    def on_epoch_end(self, epoch, logs=None):
        self.epoch.append(epoch)
        for k, v in (logs or {}).items():
            self.history.setdefault(k, []).append(v)


# ------------------------------------------------------------------------------------------ layers (descriptive)
class _SubLayer:
    def __init__(self, name, trainable):
        self.name = name
        self.trainable = trainable


class Layer:
    """Descriptor of one graph node, enough for ``model.layers[i]``, ``get_layer(name)``, ``layer_trainable``."""

    def __init__(self, name, kind, model=None, weight_prefix=None, trainable=True, config=None):
        self.name = name
        self.kind = kind
        self._model = model
        self.weight_prefix = weight_prefix
        self.trainable = trainable
        self.config = config or {}
        if kind == "Bidirectional":
            self.forward_layer = _SubLayer("forward_" + name, trainable)
            self.backward_layer = _SubLayer("backward_" + name, trainable)

    @property
    def trainable_weights(self):
        if self.weight_prefix is None or self._model is None:
            return []
        return [n for n, _, tr, _ in self._model.spec.weight_table() if n.startswith(self.weight_prefix + "/") and tr]

    @property
    def output(self):
        return _Tensor(self.name, self._model)

    def get_weights(self):
        w = self._model.get_weights_dict()
        return [w[n] for n, _, _, _ in self._model.spec.weight_table() if n.startswith(self.weight_prefix + "/")] \
            if self.weight_prefix else []


class _Tensor:
    def __init__(self, name, model):
        self.name = name
        self.model = model


# ------------------------------------------------------------------------------------------ model
class Model:
    """Keras-like façade over one NetworkSpec + device Engine.

    ``Model(inputs=loaded.input, outputs=loaded.get_layer('softmax').output)`` (sequence_decoding.py:118)
    yields a prediction view sharing the same engine and weights.
    """

    def __init__(self, spec=None, inputs=None, outputs=None, input=None, output=None, device=0, seed=1234):
        outputs = outputs if outputs is not None else output
        if spec is None and isinstance(outputs, _Tensor):
            src = outputs.model
            self.__dict__.update(src.__dict__)
            self._predict_view = True
            self._shared = src
            return
        if not isinstance(spec, NetworkSpec):
            raise TypeError("Model needs a NetworkSpec (graph construction is replaced by the spec)")
        self.spec = spec
        self.device = device
        self.seed = seed
        self._engine = None
        self._weights = None      # host copy (Keras layouts) until an engine exists
        self._predict_view = False
        self._shared = None
        self.optimizer = None
        self.stop_training = False
        self.comm = None
        self.world = 1
        self.name = spec.name
        self._init_default_weights()
        self.layers = self._describe_layers()
        self.input = [_Tensor(s["name"], self) for s in spec.streams]

    # -- structure ---------------------------------------------------------------------------------
    def _describe_layers(self):
        sp = self.spec
        L = []
        for s in sp.streams:
            for n in (s.get("inputs") or [s["name"]]):
                L.append(Layer(n, "InputLayer", self))
        for s in sp.streams:
            L.append(Layer("gaussian_noise_" + s["name"], "GaussianNoise", self, config={"stddev": s["noise"]}))
            if s.get("inputs"):
                L.append(Layer(s["name"], "Concatenate", self))
        for s in sp.streams:
            for k, lay in enumerate(s["layers"]):
                L.append(Layer(lay.get("name", "%s_blstm_%d" % (s["name"], k + 1)), "Bidirectional", self,
                               "%s/l%d" % (s["name"], k), s["trainable"], dict(lay)))
            if s.get("residual") and len(s["layers"]) == 2:
                L.append(Layer(s["name"] + "_residual", "Add", self))
        if len(sp.streams) > 1:
            L.append(Layer("merge_1", "Merge", self))
        if sp.fusion:
            L.append(Layer(sp.fusion.get("name", "blstm_2"), "Bidirectional", self, "fusion", True, dict(sp.fusion)))
        L.append(Layer(sp.head.get("dropout_name", "dropout_layer"), "Dropout", self, config={"rate": sp.head["dropout"]}))
        L.append(Layer("dense_1", "Dense", self, "dense", True))
        L.append(Layer("softmax", "Activation", self))
        L.append(Layer("the_labels", "InputLayer", self))
        L.append(Layer("input_length", "InputLayer", self))
        L.append(Layer("label_length", "InputLayer", self))
        L.append(Layer("ctc", "Lambda", self))
        return L

    def get_layer(self, name=None, index=None):
        if index is not None:
            return self.layers[index]
        for l in self.layers:
            if l.name == name:
                return l
        raise ValueError("No such layer: %s" % name)

    def summary(self, file=None):
        out = file or sys.stdout
        print("_" * 80, file=out)
        print("%-34s %-22s %12s" % ("Layer (type)", "Weights", "Param #"), file=out)
        print("=" * 80, file=out)
        tab = self.spec.weight_table()
        for l in self.layers:
            n = 0
            if l.weight_prefix:
                n = sum(int(np.prod(sh)) for nm, sh, _, _ in tab if nm.startswith(l.weight_prefix + "/"))
            print("%-34s %-22s %12d" % ("%s (%s)" % (l.name, l.kind), l.weight_prefix or "-", n), file=out)
        tot = self.spec.count_params()
        tr = self.spec.count_params(trainable_only=True)
        print("=" * 80, file=out)
        print("Total params: {:,}\nTrainable params: {:,}\nNon-trainable params: {:,}".format(tot, tr, tot - tr), file=out)

    def to_json(self):
        return self.spec.to_json()

    def count_params(self):
        return self.spec.count_params()

    # -- weights -----------------------------------------------------------------------------------
    def _init_default_weights(self):
        """RandomUniform(-.05,.05, seed=47) kernels (multimodal.py:88-90), orthogonal recurrent, unit forget bias."""
        rng = np.random.RandomState(47)
        w = {}
        for name, shape, _, kind in self.spec.weight_table():
            if kind in ("kernel", "dense"):
                w[name] = rng.uniform(-0.05, 0.05, size=shape).astype(np.float32)
            elif kind == "recurrent":
                H = shape[0]
                blocks = []
                for _ in range(4):
                    q, r = np.linalg.qr(rng.standard_normal((H, H)))
                    blocks.append(q * np.sign(np.diag(r)))
                w[name] = np.concatenate(blocks, axis=1).astype(np.float32)
            elif kind == "bias":
                H = shape[0] // 4
                b = np.zeros(shape, np.float32)
                b[H:2 * H] = 1.0
                w[name] = b
            else:
                w[name] = np.zeros(shape, np.float32)
        self._weights = w

    def get_weights_dict(self):
        if self._engine is not None:
            return self._engine.get_weights()
        return {k: v.copy() for k, v in self._weights.items()}

    def set_weights_dict(self, weights):
        if self._engine is not None:
            self._engine.set_weights(weights)
        else:
            for k, v in weights.items():
                if k in self._weights:
                    if tuple(np.shape(v)) != self._weights[k].shape:
                        raise ValueError("weight %s: expected %s got %s" % (k, self._weights[k].shape, np.shape(v)))
                    self._weights[k] = np.asarray(v, np.float32).copy()

    def clear_scan_status(self):
        """Forget a recorded non-finite state / scan give-up of the engine (Engine.clear_scan_status); set_weights_dict /
        load_weights of a good checkpoint do it implicitly."""
        if self._engine is not None:
            self._engine.clear_scan_status()

    def get_weights(self):
        w = self.get_weights_dict()
        return [w[n] for n, _, _, _ in self.spec.weight_table()]

    def set_weights(self, lst):
        names = [n for n, _, _, _ in self.spec.weight_table()]
        self.set_weights_dict(dict(zip(names, lst)))

    @property
    def is_chief(self):
        """Data parallel: rank 0 writes checkpoints / model files; every replica holds the same weights and - the validation loss
        being all-reduced (evaluate_generator) - takes the same save_best_only / early-stopping decisions."""
        return int(getattr(self.comm, "rank", 0) or 0) == 0

    def save_weights(self, filepath, overwrite=True):
        """``*.h5`` / ``*.hdf5``: an HDF5 file in the Keras 2.1.4 ``save_weights`` layout (keras_io / h5lite - no h5py);
        any other name: a numpy ``.npz`` archive under exactly that name.  Data parallel: only rank 0 writes."""
        if not self.is_chief:
            return
        w = self.get_weights_dict()
        if str(filepath).lower().endswith((".h5", ".hdf5")):
            from .keras_io import save_keras_weights
            save_keras_weights(filepath, self.spec, w)
            return
        with open(filepath, "wb") as f:
            np.savez(f, **{k.replace("/", "__"): v for k, v in w.items()})

    def load_weights(self, filepath, by_name=False):
        """Reads a Keras HDF5 weights file (written by Keras itself or by save_weights above) or an ``.npz`` archive."""
        from . import h5lite
        if h5lite.is_hdf5(filepath):
            from .keras_io import load_keras_weights
            w = load_keras_weights(filepath, self.spec)
        else:
            with np.load(filepath) as z:
                w = {k.replace("__", "/"): z[k] for k in z.files}
        if by_name:
            w = {k: v for k, v in w.items() if k in dict((n, 1) for n, _, _, _ in self.spec.weight_table())}
        self.set_weights_dict(w)

    # -- engine ------------------------------------------------------------------------------------
    def compile(self, loss=None, optimizer=None, **kwargs):
        """The reference's loss is the identity on the CTC Lambda output (multimodal.py:212); only the optimizer matters."""
        self.optimizer = optimizer
        if isinstance(optimizer, Adam):
            self.spec.optimizer.update(optimizer.config())
        if self._engine is not None:
            self._engine.spec = self.spec

    def distribute(self, comm, world):
        """Attach a data-parallel communicator (mgr_amd.parallel.RcclComm / HostComm) before the first batch.  The engine is
        built on the communicator's Device (the all-reduce must be ordered on the engine's own stream) and draws its
        dropout / noise from a per-rank seed; every rank feeds ITS shard of the global batch (parallel.shard_batch)."""
        if self._engine is not None:
            raise RuntimeError("distribute() must be called before the first batch")
        self.comm, self.world = comm, int(world)
        if getattr(comm, "dev", None) is not None:
            self.device = comm.dev

    def _ensure_engine(self, B, T, Lmax, inference_only=False):
        from .engine import Engine
        e = self._engine
        if e is not None and (e.B, e.T) == (B, T) and e.Lmax >= Lmax and (inference_only or not e.inference_only):
            return e
        w = self.get_weights_dict()
        if e is not None:
            opt_state = None if e.inference_only else (e.m.download(), e.v.download(), e.iterations)
            e.close()
        else:
            opt_state = None
        rank = int(getattr(self.comm, "rank", 0) or 0)
        self._engine = Engine(self.spec, B, T, max(Lmax, 1), device=self.device, seed=self.seed + 7919 * rank, comm=self.comm,
                              world=self.world, inference_only=inference_only)
        self._engine.set_weights(w)
        if opt_state is not None and not inference_only:
            self._engine.m.upload(opt_state[0])
            self._engine.v.upload(opt_state[1])
            self._engine.iterations = opt_state[2]
        if self._shared is not None:
            self._shared._engine = self._engine
        return self._engine

    def _split_inputs(self, x):
        names = [s["name"] for s in self.spec.streams]
        if isinstance(x, dict):
            # a stream may read several graph inputs concatenated on the feature axis (early fusion)
            return {s["name"]: (np.concatenate([np.asarray(x[k]) for k in s["inputs"]], axis=2) if s.get("inputs")
                                else x[s["name"]]) for s in self.spec.streams}
        if isinstance(x, (list, tuple)):
            return dict(zip(names, x))
        return {names[0]: x}

    def _require_trainable(self):
        if self.optimizer is not None and not isinstance(self.optimizer, Adam):
            raise NotImplementedError("this model was compiled with %s: the device optimizer is Adam (the reference's training "
                                      "path, multimodal.py:206-208) - compile(optimizer=Adam(...)) before training"
                                      % type(self.optimizer).__name__)

    def train_on_batch(self, x, y=None, rand=None, next_x=None, _lagged=False, after_next_x=None):
        """next_x: the batch of the FOLLOWING call (fit_generator passes it): with frozen encoders the engine overlaps
        that batch's encoder pass with this step's trainable part; after_next_x: the batch of the call after that - the
        encoder stream is handed the first part of its pass a call early (Engine.enqueue_train_step, prefetch_after_next).
        Data parallel: returns the mean over the GLOBAL batch (Engine.read_global_loss), the same number on every rank."""
        self._require_trainable()
        ins = self._cached_split(x)
        first = next(iter(ins.values()))
        B, T = first.shape[0], first.shape[1]
        labels = np.asarray(x["the_labels"])
        e = self._ensure_engine(B, T, labels.shape[1])
        nxt = self._cached_split(next_x) if next_x is not None else None
        nxt2 = self._cached_split(after_next_x) if (after_next_x is not None and nxt is not None) else None
        if not _lagged:
            return e.train_step(ins, labels, x["input_length"], x["label_length"], rand=rand, next_inputs=nxt, after_next_inputs=nxt2)
        # fit_generator, world > 1: the global loss arrives with the gradient all-reduce at the END of the step; waiting for it
        # here would leave the device idle while the host assembles the next batch - pace on the local loss, collect the global
        # one a step later (Engine.read_global_loss)
        e.enqueue_train_step(ins, labels, x["input_length"], x["label_length"], rand, True, prefetch_next=nxt is not None,
                             next_inputs=nxt, prefetch_after_next=nxt2 is not None, after_next_inputs=nxt2)
        e.read_loss(local=True)
        return e._step_id - 1

    def _cached_split(self, x):
        """_split_inputs with identity preserved across calls (the engine matches a prefetched batch by identity).
        The cache entry holds a reference to `x` itself and is matched with `is`: an `id()` key alone can be reused by
        CPython for a NEW batch as soon as the old dict is freed, which would silently train on stale inputs."""
        c = getattr(self, "_split_cache", None) or []
        for cx, cins in c:
            if cx is x:
                return cins
        ins = self._split_inputs(x)
        self._split_cache = (c + [(x, ins)])[-3:]     # (the batch of this call and the two announced behind it)
        return ins

    def test_on_batch(self, x, y=None, rand=None):
        ins = self._split_inputs(x)
        first = next(iter(ins.values()))
        B, T = first.shape[0], first.shape[1]
        labels = np.asarray(x["the_labels"])
        e = self._ensure_engine(B, T, labels.shape[1])
        return float(np.mean(e.loss_on_batch(ins, labels, x["input_length"], x["label_length"], rand=rand,
                                             train_phase=bool(learning_phase()))))

    def predict_on_batch(self, x):
        ins = self._split_inputs(x)
        first = next(iter(ins.values()))
        B, T = first.shape[0], first.shape[1]
        e = self._ensure_engine(B, T, self._engine.Lmax if self._engine else 1, inference_only=self._engine is None)
        if learning_phase():
            return e.forward_train_phase(ins)
        return e.predict(ins)

    def fit_generator(self, generator, steps_per_epoch, epochs=1, verbose=1, callbacks=None, validation_data=None,
                      validation_steps=None, initial_epoch=0, **kwargs):
        self._require_trainable()
        callbacks = list(callbacks or [])
        hist = History()
        callbacks.append(hist)
        for cb in callbacks:
            if hasattr(cb, "set_model"):
                cb.set_model(self)
            else:
                cb.model = self
        steps_per_epoch = int(steps_per_epoch)
        self.stop_training = False
        for cb in callbacks:
            if hasattr(cb, "on_train_begin"):
                cb.on_train_begin({})
        for epoch in range(initial_epoch, int(epochs)):
            t0 = time.time()
            for cb in callbacks:
                if hasattr(cb, "on_epoch_begin"):
                    cb.on_epoch_begin(epoch, {})
            losses = []
            lagged = self.world > 1 and self.comm is not None
            owed = None       # (data parallel) id of the step whose global loss has not been collected yet
            skipped0 = self._engine.updates_skipped if self._engine is not None else 0
            pending = next(generator) if steps_per_epoch > 0 else None
            pending2 = next(generator) if steps_per_epoch > 1 else None
            for step in range(steps_per_epoch):
                x, y = pending
                # fetch the next TWO batches early (never across an epoch boundary: on_epoch_end reshuffles the file lists)
                pending, pending2 = pending2, (next(generator) if step + 2 < steps_per_epoch else None)
                r = self.train_on_batch(x, y, next_x=pending[0] if pending is not None else None, _lagged=lagged,
                                        after_next_x=pending2[0] if pending2 is not None else None)
                if lagged:
                    if owed is not None:
                        losses.append(self._engine.read_global_loss(owed))
                    owed = r
                else:
                    losses.append(r)
            if owed is not None:
                losses.append(self._engine.read_global_loss(owed))
            logs = {"loss": float(np.mean(losses)) if losses else float("nan")}
            if self._engine is not None and self._engine.updates_skipped > skipped0:
                import warnings
                warnings.warn("%d optimizer update(s) of this epoch were skipped on the device (a scan of the step reported a non-finite "
                              "hidden state or gave up): the weights are those of the last good step - Model.clear_scan_status() "
                              "after restoring a good state" % (self._engine.updates_skipped - skipped0))
            if validation_data is not None and validation_steps:
                logs["val_loss"] = self.evaluate_generator(validation_data, validation_steps)
            if verbose:
                print("Epoch %d/%d - %.1fs - %s" % (epoch + 1, epochs, time.time() - t0,
                                                    " - ".join("%s: %.4f" % kv for kv in logs.items())))
            for cb in callbacks:
                if hasattr(cb, "on_epoch_end"):
                    cb.on_epoch_end(epoch, logs)
            if self.stop_training:
                break
        for cb in callbacks:
            if hasattr(cb, "on_train_end"):
                cb.on_train_end({})
        return hist

    @staticmethod
    def _pad_batch(arrs, B):
        """A short last batch is padded with zero rows to the engine's batch size (a sample's results do not depend on the
        other rows of its batch); returns (padded arrays, true row count)."""
        n = next(iter(arrs.values())).shape[0] if isinstance(arrs, dict) else arrs.shape[0]
        if n == B:
            return arrs, n
        pad = lambda a: np.concatenate([np.asarray(a), np.zeros((B - n,) + np.asarray(a).shape[1:], np.asarray(a).dtype)], axis=0)
        return ({k: pad(v) for k, v in arrs.items()} if isinstance(arrs, dict) else pad(arrs)), n

    def predict_generator(self, generator, steps, verbose=0, decode=None, beam_width=10, **kwargs):
        """keras Model.predict_generator (sequence_decoding.py:118-127): the batches of the run are pipelined through
        Engine.predict_stream - upload and encoder pass of batch n + 1 beside the fusion layer / head of batch n and the
        download of batch n - 1 - and give bit for bit what predict_on_batch gives one batch at a time.
        decode=None returns the softmax outputs (N, T, C) like Keras; decode="argmax" returns (best, prob), each (N, T - skip):
        the per-frame best label and its probability computed on the device (what decode_batch needs; the (N, T, C) posteriors
        never cross PCIe); decode="beam" returns (paths, log-probabilities) of mgr_ctc_beam_search(beam_width)."""
        steps = int(steps)
        if steps <= 0:
            return np.zeros((0,))
        first = next(generator)
        x0 = first[0] if isinstance(first, tuple) else first
        ins0 = self._split_inputs(x0)
        f0 = next(iter(ins0.values()))
        B, T = f0.shape[0], f0.shape[1]
        e = self._ensure_engine(B, T, self._engine.Lmax if self._engine else 1, inference_only=self._engine is None)
        counts = []

        def feed():
            for i in range(steps):
                batch = first if i == 0 else next(generator)
                x = batch[0] if isinstance(batch, tuple) else batch
                ins, n = self._pad_batch(self._split_inputs(x), B)
                counts.append(n)
                yield ins

        output = {None: "posteriors", "argmax": "argmax", "beam": "beam"}[decode]
        outs = []
        for i, r in enumerate(e.predict_stream(feed(), output=output, train_phase=bool(learning_phase()), beam_width=beam_width)):
            n = counts[i]
            if output == "posteriors":
                outs.append(r[:n])
            elif output == "argmax":
                outs.append((r[0][:n], r[1][:n]))
            else:
                outs.append((r[0][:n], r[1][:n]))
            if verbose:
                print("%d/%d" % (i + 1, steps))
        if output == "posteriors":
            return np.concatenate(outs, axis=0)
        if output == "argmax":
            return np.concatenate([o[0] for o in outs], axis=0), np.concatenate([o[1] for o in outs], axis=0)
        return [p for o in outs for p in o[0]], np.concatenate([o[1] for o in outs], axis=0)

    def evaluate_generator(self, generator, steps, **kwargs):
        """Mean CTC loss over `steps` batches (the validation loop of fit_generator, multimodal.py:264-269), pipelined like
        predict_generator; the learning phase is whatever is set (the reference leaves it at 1 during validation)."""
        steps = int(steps)
        if steps <= 0:
            return float("nan")
        first = next(generator)
        x0 = first[0]
        ins0 = self._split_inputs(x0)
        f0 = next(iter(ins0.values()))
        e = self._ensure_engine(f0.shape[0], f0.shape[1], np.asarray(x0["the_labels"]).shape[1])

        def feed():
            for i in range(steps):
                x = (first if i == 0 else next(generator))[0]
                yield self._split_inputs(x), np.asarray(x["the_labels"]), x["input_length"], x["label_length"]

        means = [float(np.mean(l)) for l in e.predict_stream(feed(), output="loss", train_phase=bool(learning_phase()))]
        v = float(np.mean(means))
        if self.comm is not None and self.world > 1:
            # data parallel: every rank evaluated ITS shard of each batch; the validation loss every replica logs - and on which
            # ModelCheckpoint(save_best_only) / EarlyStopping decide - is the mean over the global batches (one scalar all-reduce)
            v = self.comm.allreduce_sum_scalar(v) / self.world
        return v


def model_from_json(text, device=0):
    """keras.models.model_from_json: accepts the JSON this package writes (Model.to_json) and the functional-API JSON
    Keras 2.1.4 writes for the reference's networks (keras_io.spec_from_keras_json)."""
    d = json.loads(text)
    if d.get("class_name") == "Model":
        from .keras_io import spec_from_keras_json
        spec, _, _ = spec_from_keras_json(d)
        return Model(spec, device=device)
    return Model(NetworkSpec.from_json(text), device=device)
