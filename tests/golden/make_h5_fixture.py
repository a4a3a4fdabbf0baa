"""Writes tests/golden/keras_weights_tiny.h5 (+ .npz with the same arrays) with the REAL HDF5 library through h5py,
in the group / attribute layout of Keras 2.1.4 ``model.save_weights`` (keras/engine/topology.py
``save_weights_to_hdf5_group``: root attrs layer_names / backend / keras_version, one group per layer with a
``weight_names`` attr and one dataset per weight named after the TF variable).  It pins the pure-Python reader in
``h5lite.py`` against bytes produced by libhdf5, and is also the layout ``Model.save_weights('*.h5')`` reproduces.

Run where h5py exists (in the build image: /opt/conda/bin/python3.9 tests/golden/make_h5_fixture.py).
Layout knowledge is recalled from Keras 2.1.4 (not installed anywhere here).
"""
import os
import sys

import h5py
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def tiny_layers(rng):
    """(layer name, [(weight name, array)]) in model.layers order for a 2-layer unimodal net F=5, H=3, C=4."""
    def lstm(fin, h, wrapper, inner):
        out = []
        for d in ("forward", "backward"):
            base = "%s/%s_%s/" % (wrapper, d, inner)
            out.append((base + "kernel:0", rng.standard_normal((fin, 4 * h)).astype(np.float32)))
            out.append((base + "recurrent_kernel:0", rng.standard_normal((h, 4 * h)).astype(np.float32)))
            out.append((base + "bias:0", rng.standard_normal((4 * h,)).astype(np.float32)))
        return out
    return [
        ("the_input", []),
        ("gaussian_noise_1", []),
        ("bidirectional_1", lstm(5, 3, "bidirectional_1", "blstm_1")),
        ("bidirectional_2", lstm(6, 3, "bidirectional_2", "blstm_2")),
        ("add_1", []),
        ("dropout_layer_1", []),
        ("dense_1", [("dense_1/kernel:0", rng.standard_normal((6, 4)).astype(np.float32)),
                     ("dense_1/bias:0", rng.standard_normal((4,)).astype(np.float32))]),
        ("softmax", []),
        ("the_labels", []),
        ("input_length", []),
        ("label_length", []),
        ("ctc", []),
    ]


def write(path, layers, vlen=False, **kw):
    """vlen=False stores the name lists the way h5py 2.7.1 (requirements.txt:1) did - numpy 'S' arrays, i.e. fixed-length
    null-padded strings; vlen=True is what h5py >= 3 does with a list of bytes (variable-length strings, global heap)."""
    conv = (lambda names: names) if vlen else (lambda names: np.array(names, dtype='S') if names else names)
    one = (lambda s: s) if vlen else np.bytes_
    with h5py.File(path, "w", **kw) as f:
        f.attrs['layer_names'] = conv([n.encode('utf8') for n, _ in layers])
        f.attrs['backend'] = one('tensorflow'.encode('utf8'))
        f.attrs['keras_version'] = one('2.1.4'.encode('utf8'))
        for name, ws in layers:
            g = f.create_group(name)
            g.attrs['weight_names'] = conv([wn.encode('utf8') for wn, _ in ws])
            for wn, val in ws:
                d = g.create_dataset(wn, val.shape, dtype=val.dtype)
                d[:] = val


if __name__ == "__main__":
    out = sys.argv[1] if len(sys.argv) > 1 else HERE
    layers = tiny_layers(np.random.default_rng(214))
    write(os.path.join(out, "keras_weights_tiny.h5"), layers)
    write(os.path.join(out, "keras_weights_tiny_vlen.h5"), layers, vlen=True)
    write(os.path.join(out, "keras_weights_tiny_latest.h5"), layers, libver="latest")
    np.savez(os.path.join(out, "keras_weights_tiny.npz"),
             **{wn: val for _, ws in layers for wn, val in ws})
    print("h5py", h5py.__version__, "hdf5", h5py.version.hdf5_version)
