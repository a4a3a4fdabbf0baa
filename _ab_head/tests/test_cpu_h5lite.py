"""CPU: Keras checkpoint interop.
 * h5lite reader against files written by the REAL HDF5 library (tests/golden/keras_weights_tiny*.h5, generated with
   h5py / libhdf5 1.10.6 by tests/golden/make_h5_fixture.py; expected arrays in keras_weights_tiny.npz)
 * h5lite writer: self round trip, and - where an interpreter with h5py exists (build image: /opt/conda) - re-read by h5py
 * Keras functional-model JSON -> NetworkSpec, Model.save_weights/load_weights through HDF5
"""
import json
import os
import subprocess

import numpy as np
import pytest

import mgr_amd  # noqa: F401
from mgr_amd import configs, h5lite, keras_io
from mgr_amd.keras_like import Model, model_from_json

GOLD = os.path.join(os.path.dirname(__file__), "golden")
H5PY_PYTHON = "/opt/conda/bin/python3.9"


@pytest.mark.parametrize("fname", ["keras_weights_tiny.h5", "keras_weights_tiny_vlen.h5"])
def test_reader_against_libhdf5_files(fname):
    exp = np.load(os.path.join(GOLD, "keras_weights_tiny.npz"))
    layers = h5lite.read_keras_weights(os.path.join(GOLD, fname))
    assert [n for n, _ in layers] == ["the_input", "gaussian_noise_1", "bidirectional_1", "bidirectional_2", "add_1",
                                      "dropout_layer_1", "dense_1", "softmax", "the_labels", "input_length",
                                      "label_length", "ctc"]
    seen = 0
    for _, ws in layers:
        for wn, v in ws:
            assert v.dtype == np.float32 and np.array_equal(v, exp[wn]), wn
            seen += 1
    assert seen == len(exp.files) == 14
    root = h5lite.read_file(os.path.join(GOLD, fname))
    assert root.attrs["backend"] == b"tensorflow" and root.attrs["keras_version"] == b"2.1.4"
    assert np.size(root["add_1"].attrs["weight_names"]) == 0
    assert root["dense_1/dense_1/kernel:0"].shape == (6, 4)


def test_reader_rejects_what_it_does_not_implement(tmp_path):
    # libver='latest' stores the 12-entry root group in a fractal heap: must be refused loudly, never mis-read
    with pytest.raises(h5lite.H5Error, match="dense"):
        h5lite.read_file(os.path.join(GOLD, "keras_weights_tiny_latest.h5"))
    p = tmp_path / "x.h5"
    p.write_bytes(b"not hdf5 at all")
    assert not h5lite.is_hdf5(str(p))
    with pytest.raises(h5lite.H5Error):
        h5lite.read_file(str(p))


def _many_layers(rng, n):
    return [("layer_%03d" % i, [("layer_%03d/w_%d:0" % (i, j), rng.standard_normal((3, 2 + j)).astype(np.float32))
                                for j in range(i % 3)]) for i in range(n)]


def test_writer_self_round_trip(tmp_path):
    rng = np.random.default_rng(1)
    layers = _many_layers(rng, 150)           # > 64 entries: several symbol-table nodes under one B-tree node
    p = str(tmp_path / "many.h5")
    h5lite.write_keras_weights(p, layers)
    back = h5lite.read_keras_weights(p)
    assert [n for n, _ in back] == [n for n, _ in layers]
    for (_, a), (_, b) in zip(layers, back):
        assert [n for n, _ in a] == [n for n, _ in b]
        for (_, x), (_, y) in zip(a, b):
            assert np.array_equal(x, y)
    w = h5lite.Writer()
    w.create_dataset("g/ints", np.arange(6, dtype=np.int64).reshape(2, 3))
    w.create_dataset("g/scalar", np.float64(2.5))
    w.set_attr("g", "note", "hello")
    w.set_attr("g", "vals", np.array([1.5, 2.5], np.float32))
    w.save(str(tmp_path / "misc.h5"))
    r = h5lite.read_file(str(tmp_path / "misc.h5"))
    assert np.array_equal(r["g/ints"].value, np.arange(6).reshape(2, 3)) and float(r["g/scalar"].value) == 2.5
    assert r["g"].attrs["note"] == b"hello" and np.array_equal(r["g"].attrs["vals"], [1.5, 2.5])


@pytest.mark.skipif(not os.path.exists(H5PY_PYTHON), reason="no interpreter with h5py on this machine")
def test_writer_output_is_read_by_libhdf5(tmp_path):
    rng = np.random.default_rng(2)
    layers = _many_layers(rng, 70)
    p = str(tmp_path / "w.h5")
    h5lite.write_keras_weights(p, layers)
    np.savez(str(tmp_path / "exp.npz"), **{wn: v for _, ws in layers for wn, v in ws})
    code = r'''
import sys, h5py, numpy as np
f = h5py.File(sys.argv[1], "r"); exp = np.load(sys.argv[2])
names = [n.decode() for n in f.attrs["layer_names"]]
assert names == ["layer_%03d" % i for i in range(70)], names[:3]
assert f.attrs["backend"] == b"tensorflow" and f.attrs["keras_version"] == b"2.1.4"
n = 0
for ln in names:
    g = f[ln]
    for wn in g.attrs["weight_names"]:
        wn = wn.decode()
        assert np.array_equal(g[wn][()], exp[wn]) and g[wn].dtype == np.float32
        n += 1
assert n == len(exp.files)
print("OK", n)
'''
    r = subprocess.run([H5PY_PYTHON, "-c", code, p, str(tmp_path / "exp.npz")], capture_output=True, text=True,
                       env={"PATH": os.environ.get("PATH", "")})
    assert r.returncode == 0 and r.stdout.startswith("OK"), r.stderr[-2000:]


# ------------------------------------------------------------------------------------------------ Keras JSON
def _k_input(name, shape, dtype="float32"):
    return {"name": name, "class_name": "InputLayer", "inbound_nodes": [],
            "config": {"batch_input_shape": [None] + list(shape), "dtype": dtype, "sparse": False, "name": name}}


def _k_layer(cls, name, cfg, inbound):
    return {"name": name, "class_name": cls, "config": dict(cfg, name=name),
            "inbound_nodes": [[[n, 0, 0, {}] for n in inbound]]}


def _k_bilstm(wrapper, inner, units, dropout, inbound, trainable=True):
    lstm = {"class_name": "LSTM", "config": {
        "name": inner, "trainable": trainable, "return_sequences": True, "return_state": False, "go_backwards": False,
        "stateful": False, "unroll": False, "implementation": 1, "units": units, "activation": "tanh",
        "recurrent_activation": "hard_sigmoid", "use_bias": True, "unit_forget_bias": True,
        "kernel_constraint": {"class_name": "MaxNorm", "config": {"max_value": 3, "axis": 0}},
        "dropout": dropout, "recurrent_dropout": 0.0}}
    return _k_layer("Bidirectional", wrapper, {"trainable": trainable, "layer": lstm, "merge_mode": "concat"}, inbound)


def _k_tail(layers, top, C, lab_len, drop, drop_name):
    layers += [_k_layer("Dropout", drop_name, {"trainable": True, "rate": drop}, [top]),
               _k_layer("Dense", "dense_1", {"trainable": True, "units": C, "activation": "linear", "use_bias": True},
                        [drop_name]),
               _k_layer("Activation", "softmax", {"trainable": True, "activation": "softmax"}, ["dense_1"]),
               _k_input("the_labels", [lab_len]), _k_input("input_length", [1], "int64"),
               _k_input("label_length", [1], "int64"),
               _k_layer("Lambda", "ctc", {"trainable": True, "function": ["<marshalled code>", None, None],
                                          "function_type": "lambda", "output_shape": [1], "arguments": {}},
                        ["softmax", "the_labels", "input_length", "label_length"])]
    return layers


def _wrap(layers, inputs):
    return json.dumps({"class_name": "Model", "keras_version": "2.1.4", "backend": "tensorflow",
                       "config": {"name": "model_2", "layers": layers,
                                  "input_layers": [[n, 0, 0] for n in inputs], "output_layers": [["ctc", 0, 0]]}})


def keras_json_unimodal(T=1900, F=39, H=500, C=44, lab=150, drops=(0.4, 0.5, 0.5)):
    L = [_k_input("the_input", [T, F]),
         _k_layer("GaussianNoise", "gaussian_noise_1", {"trainable": True, "stddev": 0.5}, ["the_input"]),
         _k_bilstm("bidirectional_1", "blstm_1", H, drops[0], ["gaussian_noise_1"]),
         _k_bilstm("bidirectional_2", "blstm_2", H, drops[1], ["bidirectional_1"]),
         _k_layer("Add", "add_1", {"trainable": True}, ["bidirectional_1", "bidirectional_2"])]
    return _wrap(_k_tail(L, "add_1", C, lab, drops[2], "dropout_layer_1"), ["the_input", "the_labels", "input_length", "label_length"])


def keras_json_fusion(T=1900):
    L = [_k_input("the_input_audio", [T, 39]), _k_input("the_input_skeletal", [T, 20]),
         _k_layer("GaussianNoise", "gaussian_noise_a", {"trainable": True, "stddev": 0.5}, ["the_input_audio"]),
         _k_layer("GaussianNoise", "gaussian_noise_s", {"trainable": True, "stddev": 0.0}, ["the_input_skeletal"]),
         _k_bilstm("speech_blstm_1", "blstm_1", 500, 0.4, ["gaussian_noise_a"], trainable=False),
         _k_bilstm("skeletal_blstm_1", "blstm_1", 300, 0.6, ["gaussian_noise_s"], trainable=False),
         _k_bilstm("speech_blstm_2", "blstm_2", 500, 0.5, ["speech_blstm_1"], trainable=False),
         _k_bilstm("skeletal_blstm_2", "blstm_2", 300, 0.6, ["skeletal_blstm_1"], trainable=False),
         _k_layer("Add", "speech_residual", {"trainable": True}, ["speech_blstm_1", "speech_blstm_2"]),
         _k_layer("Add", "skeletal_residual", {"trainable": True}, ["skeletal_blstm_1", "skeletal_blstm_2"]),
         _k_layer("Merge", "merge_1", {"mode": "concat", "concat_axis": -1}, ["speech_residual", "skeletal_residual"]),
         _k_bilstm("bidirectional_3", "blstm_2", 100, 0.5, ["merge_1"])]
    return _wrap(_k_tail(L, "bidirectional_3", 22, 35, 0.5, "dropout_layer_3"),
                 ["the_input_audio", "the_input_skeletal", "the_labels", "input_length", "label_length"])


def keras_json_early(T=1900):
    L = [_k_input("the_input_audio", [T, 39]), _k_input("the_input_skeletal", [T, 20]),
         _k_layer("GaussianNoise", "gaussian_noise_a", {"trainable": True, "stddev": 0.5}, ["the_input_audio"]),
         _k_layer("GaussianNoise", "gaussian_noise_s", {"trainable": True, "stddev": 0.5}, ["the_input_skeletal"]),
         _k_layer("Concatenate", "concatenate_1", {"trainable": True, "axis": 2}, ["gaussian_noise_a", "gaussian_noise_s"]),
         _k_bilstm("bidirectional_1", "blstm_1", 500, 0.4, ["concatenate_1"]),
         _k_bilstm("bidirectional_2", "blstm_2", 500, 0.4, ["bidirectional_1"]),
         _k_layer("Add", "add_1", {"trainable": True}, ["bidirectional_1", "bidirectional_2"])]
    return _wrap(_k_tail(L, "add_1", 22, 28, 0.4, "dropout_layer_1"),
                 ["the_input_audio", "the_input_skeletal", "the_labels", "input_length", "label_length"])


def _same_network(a, b):
    da, db = a.to_dict(), b.to_dict()
    for d in (da, db):
        d.pop("name")
        d.pop("optimizer")              # the optimizer is not part of a Keras model JSON
        for lay in [l for s in d["streams"] for l in s["layers"]] + ([d["fusion"]] if d.get("fusion") else []):
            lay.pop("maxnorm", None)    # (a JSON carries the constraint per layer, the builders use the optimizer-wide default)
    return da == db


def test_keras_json_to_spec():
    spec, T, lab = keras_io.spec_from_keras_json(keras_json_unimodal())
    assert (T, lab) == (1900, 150) and _same_network(spec, configs.audio_spec(39, 44, 500, 2))
    assert spec.count_params() == 8208044
    spec, T, lab = keras_io.spec_from_keras_json(keras_json_fusion())
    assert (T, lab) == (1900, 35) and _same_network(spec, configs.fusion_spec())
    assert spec.count_params(trainable_only=True) == 1365222
    spec, _, _ = keras_io.spec_from_keras_json(keras_json_early())
    assert _same_network(spec, configs.early_fusion_spec())
    with pytest.raises(ValueError, match="not on the reference"):
        keras_io.spec_from_keras_json(_wrap([_k_input("x", [10, 3]), _k_layer("Conv1D", "c", {}, ["x"])], ["x"]))
    # kernel_constraint is read per layer (the reference: MaxNorm(3, axis 0) on every LSTM kernel), not assumed
    spec, _, _ = keras_io.spec_from_keras_json(keras_json_fusion())
    assert spec.kernel_maxnorm("fusion") == 3.0 and spec.kernel_maxnorm("the_input_audio/l1") == 3.0
    d = json.loads(keras_json_unimodal(T=30, F=5, H=3, C=4, lab=6))
    lstm_cfgs = [l["config"]["layer"]["config"] for l in d["config"]["layers"] if l["class_name"] == "Bidirectional"]
    lstm_cfgs[0]["kernel_constraint"] = None
    lstm_cfgs[1]["kernel_constraint"]["config"]["max_value"] = 1.5
    spec, _, _ = keras_io.spec_from_keras_json(json.dumps(d))
    assert spec.kernel_maxnorm("the_input/l0") == 0.0 and spec.kernel_maxnorm("the_input/l1") == 1.5
    assert configs.audio_spec().kernel_maxnorm("the_input/l0") == 3.0          # builders: the optimizer-wide default
    lstm_cfgs[1]["kernel_constraint"] = {"class_name": "UnitNorm", "config": {"axis": 0}}
    with pytest.raises(ValueError, match="only MaxNorm"):
        keras_io.spec_from_keras_json(json.dumps(d))
    # a recurrent / bias constraint is refused whether or not the layer also has a kernel constraint (it used to be accepted -
    # and silently ignored - on layers WITHOUT one)
    for kc in (None, {"class_name": "MaxNorm", "config": {"max_value": 3, "axis": 0}}):
        for other in ("recurrent_constraint", "bias_constraint"):
            lstm_cfgs[1]["kernel_constraint"] = kc
            lstm_cfgs[1][other] = {"class_name": "MaxNorm", "config": {"max_value": 3, "axis": 0}}
            with pytest.raises(ValueError, match=other + " of layer"):
                keras_io.spec_from_keras_json(json.dumps(d))
            lstm_cfgs[1][other] = None


def test_model_hdf5_checkpoint_round_trip(tmp_path):
    spec = configs.fusion_spec(h_audio=8, h_skeletal=4, h_fusion=4)
    m = Model(spec, seed=3)
    p = str(tmp_path / "multimodal_ctc_lstm_weights_best.h5")
    m.save_weights(p)
    assert h5lite.is_hdf5(p)
    layers = h5lite.read_keras_weights(p)
    # depth-major Keras layer order, TF variable names
    assert [n for n, _ in layers] == ["bidirectional_1", "bidirectional_2", "bidirectional_3", "bidirectional_4",
                                      "bidirectional_5", "dense_1"]
    assert layers[0][1][0][0] == "bidirectional_1/forward_speech_blstm_1/kernel:0"
    assert layers[1][1][3][0] == "bidirectional_2/backward_skeletal_blstm_1/kernel:0"
    m2 = Model(spec, seed=4)
    m2.load_weights(p)
    for a, b in zip(m.get_weights(), m2.get_weights()):
        assert np.array_equal(a, b)
    # a weights file whose shapes fit no layer is refused
    other = Model(configs.fusion_spec(h_audio=8, h_skeletal=4, h_fusion=6))
    with pytest.raises(ValueError, match="matches no layer"):
        other.load_weights(p)
    # the libhdf5-written tiny unimodal checkpoint loads into the network its Keras JSON describes
    km = model_from_json(keras_json_unimodal(T=30, F=5, H=3, C=4, lab=6))
    km.load_weights(os.path.join(GOLD, "keras_weights_tiny.h5"))
    exp = np.load(os.path.join(GOLD, "keras_weights_tiny.npz"))
    w = km.get_weights_dict()
    assert np.array_equal(w["the_input/l1/bwd/U"], exp["bidirectional_2/backward_blstm_2/recurrent_kernel:0"])
    assert np.array_equal(w["dense/W"], exp["dense_1/kernel:0"])
