"""Batch assembly shared by the three DataGenerator classes (reference multimodal_fusion/data_generator.py:157-323,
audio_network/data_generator.py:153-283, skeletal_network/skeletal_lstm_ctc.py:34-254).

Two sample stores provide per-file feature matrices and label rows:
  * SyntheticStore - seeded ChaLearn-shaped sequences (no dataset exists in the build / GPU images),
  * CsvStore       - the reference's on-disk layout (per-file ``audio_<id>.csv``, one skeletal CSV with a
                     ``file_number`` column, a label CSV with ``Id`` / ``Sequence``), read once and cached instead of
                     one ``pd.read_csv`` per file per step.
The batch contract (dict keys, float64 arrays, np.ones initialisation, post-padding with zeros, labels padded with
-1, the blank substitution for empty label rows, input_length = maxlen - 2) is the reference's.
"""
import os
import random
import re

import numpy as np

from .keras_like import Callback

SKELETAL_COLUMNS = ['lh_v', 'rh_v', 'le_v', 're_v', 'lh_dist_rp', 'rh_dist_rp', 'lh_hip_d', 'rh_hip_d', 'le_hip_d',
                    're_hip_d', 'lh_shc_d', 'rh_shc_d', 'le_shc_d', 're_shc_d', 'lh_hip_ang', 'rh_hip_ang',
                    'lh_shc_ang', 'rh_shc_ang', 'lh_el_ang', 'rh_el_ang']


def pad_post(seq, maxlen):
    """keras pad_sequences(padding='post', truncating='post', dtype='float32') for one (n,F) matrix."""
    seq = np.asarray(seq, dtype=np.float32)
    out = np.zeros((maxlen,) + seq.shape[1:], np.float32)
    n = min(maxlen, seq.shape[0])
    out[:n] = seq[:n]
    return out


class SyntheticStore:
    """Deterministic per-file synthetic features; a file's data depends only on (seed, file id)."""

    def __init__(self, n_files, feats, maxlen, nb_classes, seed=20131900, lmin=8, lmax=20, empty_every=0):
        self.ids = list(range(1, n_files + 1))
        self.feats = dict(feats)          # modality -> (F, scale)
        self.maxlen = maxlen
        self.nb_classes = nb_classes
        self.seed = seed
        self.lmin, self.lmax = lmin, lmax
        self.empty_every = empty_every    # every k-th file has an empty label row (exercises the blank substitution)
        self._cache, self._cache_bytes, self.cache_limit_bytes = {}, 0, 2 << 30

    def file_ids(self):
        return list(self.ids)

    def _rng(self, file_id, salt):
        return np.random.default_rng([self.seed, int(file_id), salt])

    def length(self, file_id):
        r = self._rng(file_id, 0)
        return int(r.integers(int(np.ceil(0.6 * self.maxlen)), self.maxlen + 1))

    def features(self, file_id, modality):
        key = (int(file_id), modality)
        hit = self._cache.get(key)
        if hit is None:
            F, scale = self.feats[modality]
            r = self._rng(file_id, 1 + sorted(self.feats).index(modality))
            hit = (r.standard_normal((self.length(file_id), F)) * scale).astype(np.float32)
            if self._cache_bytes + hit.nbytes <= self.cache_limit_bytes:   # a real store keeps its files in memory too
                self._cache[key] = hit
                self._cache_bytes += hit.nbytes
        return hit

    def labels(self, file_id):
        if self.empty_every and int(file_id) % self.empty_every == 0:
            return np.zeros((0,), np.float32)
        r = self._rng(file_id, 99)
        lmax = max(1, min(self.lmax, (self.maxlen - 2) // 2))
        L = int(r.integers(min(self.lmin, lmax), lmax + 1))
        seq = r.integers(1, self.nb_classes - 1, size=L)
        return np.where(r.random(L) < 0.05, 0, seq).astype(np.float32)


class CsvStore:
    """The reference's data layout (util/mix_data.py:71-82,152-176), loaded once."""

    def __init__(self, audio_dir=None, skeletal_csv=None, label_csv=None, audio_stride=5, skeletal_columns=None):
        import pandas as pd
        self.audio = {}
        self.skel = {}
        self.labs = {}
        if audio_dir is not None:
            for name in sorted(os.listdir(audio_dir)):
                m = re.findall(r'audio_(\d+).csv', name)
                if not m:
                    continue
                df = pd.read_csv(os.path.join(audio_dir, name))
                df = df.drop(columns=[c for c in ('file_number', '39', '40') if c in df.columns])
                self.audio[int(m[0])] = df.iloc[::audio_stride, :].to_numpy(dtype=float)   # 100 fps MFCC -> 20 fps
        if skeletal_csv is not None:
            df = pd.read_csv(skeletal_csv)
            cols = skeletal_columns or SKELETAL_COLUMNS
            data = df[cols].to_numpy(dtype=float)
            data = (data - data.mean(axis=0)) / data.std(axis=0)       # sklearn.preprocessing.scale over the whole file
            fn = df['file_number'].to_numpy()
            for f in np.unique(fn):
                self.skel[int(f)] = data[fn == f]
        if label_csv is not None:
            df = pd.read_csv(label_csv)
            for fid, seq in zip(df['Id'], df['Sequence']):
                self.labs[int(fid)] = np.array([int(v) for v in str(seq).split()], np.float32)

    def file_ids(self):
        return sorted(self.audio) if self.audio else sorted(self.skel)

    def features(self, file_id, modality):
        return (self.audio if modality == 'audio' else self.skel).get(int(file_id), np.zeros((0, 1)))

    def labels(self, file_id):
        return self.labs.get(int(file_id), np.zeros((0,), np.float32))


class BaseDataGenerator(Callback):
    """Index bookkeeping, split, generators and the epoch-end checkpoint of the reference's DataGenerator."""

    #: (batch key, modality, attribute holding the feature count) per input stream
    streams = ()
    model_json_name = "model.json"
    model_weights_name = "weights.h5"

    def _setup(self, minibatch_size, maxlen, nb_classes, dataset, val_split, absolute_max_sequence_len, store, rank=0, world=1):
        Callback.__init__(self)
        self.minibatch_size = minibatch_size
        # data parallel (not in the reference): `minibatch_size` stays the GLOBAL batch - file lists, split, wrap-around
        # and shuffles are those of the single-process generator on every rank - and get_batch assembles only this rank's
        # contiguous slice of it: rank-aware batch == parallel.shard_batch(full batch, rank, world)
        self.rank, self.world = int(rank), int(world)
        if not 0 <= self.rank < self.world:
            raise ValueError("rank %d outside world %d" % (self.rank, self.world))
        if minibatch_size % self.world:
            raise ValueError("global minibatch %d not divisible by world size %d" % (minibatch_size, self.world))
        self.maxlen = maxlen
        self.val_split = val_split
        self.absolute_max_sequence_len = absolute_max_sequence_len
        self.train_index = 0
        self.val_index = 0
        self.nb_classes = nb_classes
        self.blank_label = np.array([self.nb_classes - 1])
        self.dataset = dataset
        self.store = store
        self.load_dataset()

    def load_dataset(self):
        file_list = sorted(self.store.file_ids())
        self.train_list, self.train_size = [], 0
        if self.dataset == 'train':
            random.seed(10)
            random.shuffle(file_list)
            split_point = int(len(file_list) * (1 - self.val_split))
            self.train_list, self.val_list = file_list[:split_point], file_list[split_point:]
            self.train_size, self.val_size = len(self.train_list), len(self.val_list)
            # keep only whole minibatches
            r = self.train_size % self.minibatch_size
            if r:
                del self.train_list[-r:]
                self.train_size -= r
            r = self.val_size % self.minibatch_size
            if r:
                del self.val_list[-r:]
                self.val_size -= r
        else:
            self.val_list = file_list
            self.val_size = len(self.val_list)

    def get_size(self, train):
        return self.train_size if train else self.val_size

    def get_file_list(self, train):
        return self.train_list if train else self.val_list

    def expand_labels(self, lab_seq):
        return lab_seq

    def get_batch(self, train):
        file_list, index = (self.train_list, self.train_index) if train else (self.val_list, self.val_index)
        batch = file_list[index:index + self.minibatch_size]
        if self.world > 1:
            # (a short last batch of an un-truncated list is sliced the way shard_batch would slice it)
            if len(batch) % self.world:
                raise ValueError("global batch %d not divisible by world size %d" % (len(batch), self.world))
            per = len(batch) // self.world
            batch = batch[self.rank * per:(self.rank + 1) * per]
        size = len(batch)
        # (np.ones of the reference, :178-186; rows are fully overwritten below unless the file has no label row, so the
        # buffers come from a small ring instead of 57 MB of fresh pages per call - a batch stays valid until
        # BATCH_RING further batches have been drawn)
        X = {key: self._batch_buffer(key, (size, self.maxlen, getattr(self, attr))) for key, _, attr in self.streams}
        labels = np.ones([size, self.absolute_max_sequence_len])
        input_length = np.zeros([size, 1])
        label_length = np.zeros([size, 1])
        for i, fid in enumerate(batch):
            lab_seq = self.store.labels(fid) if self.dataset != 'final' else np.array([0], np.float32)
            lab_seq = self.expand_labels(np.asarray(lab_seq, np.float32))
            if lab_seq.shape[0] == 0:
                # no label row: the inputs stay all-ones and the target is the single blank label
                for key, _, _ in self.streams:
                    X[key][i, :, :] = 1.0
                row = -np.ones(self.absolute_max_sequence_len)
                row[0] = self.blank_label[0]
                labels[i, :] = row
                label_length[i] = 1
            else:
                for key, modality, attr in self.streams:
                    feats = np.asarray(self.store.features(fid, modality))
                    if feats.ndim == 2 and feats.shape[1] == getattr(self, attr):
                        n = min(self.maxlen, feats.shape[0])     # pad_sequences(padding='post', truncating='post')
                        X[key][i, :n, :] = feats[:n]
                        X[key][i, n:, :] = 0.0
                    else:
                        X[key][i, :, :] = 1.0
                label_length[i] = min(lab_seq.shape[0], self.absolute_max_sequence_len)
                row = -np.ones(self.absolute_max_sequence_len)
                n = int(label_length[i, 0])
                row[:n] = lab_seq[-n:] if lab_seq.shape[0] > n else lab_seq   # pad_sequences truncates 'pre' by default
                labels[i, :] = row
            input_length[i] = self.maxlen - 2
        inputs = dict(X)
        inputs['the_labels'] = labels
        inputs['input_length'] = input_length
        inputs['label_length'] = label_length
        outputs = {'ctc': np.zeros([size])}
        return (inputs, outputs)

    BATCH_RING = 4

    def _batch_buffer(self, key, shape):
        ring = self.__dict__.setdefault("_batch_ring", {})
        slot = ring.setdefault(key, {"i": 0, "bufs": []})
        bufs = slot["bufs"]
        if len(bufs) < self.BATCH_RING or bufs[slot["i"] % self.BATCH_RING].shape != tuple(shape):
            buf = np.empty(shape, np.float64)
            if len(bufs) < self.BATCH_RING:
                bufs.append(buf)
            else:
                bufs[slot["i"] % self.BATCH_RING] = buf
        buf = bufs[slot["i"] % self.BATCH_RING] if len(bufs) == self.BATCH_RING else bufs[-1]
        slot["i"] += 1
        return buf

    def next_train(self):
        while 1:
            ret = self.get_batch(train=True)
            self.train_index += self.minibatch_size
            if self.train_index >= self.train_size:
                self.train_index = 0
            yield ret

    def next_val(self):
        while 1:
            ret = self.get_batch(train=False)
            self.val_index += self.minibatch_size
            if self.val_index >= self.val_size:
                self.val_index = 0
            yield ret

    def on_epoch_end(self, epoch, logs=None):
        self.train_index = 0
        self.val_index = 0
        random.shuffle(self.train_list)
        random.shuffle(self.val_list)
        if self.model is not None and getattr(self.model, "is_chief", True):    # (data parallel: rank 0 writes)
            with open(self.model_json_name, "w") as f:
                f.write(self.model.to_json())
            self.model.save_weights(self.model_weights_name)
            print("Saved model to disk")
