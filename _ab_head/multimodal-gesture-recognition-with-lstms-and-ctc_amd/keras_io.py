"""Keras 2.1.4 checkpoint interop: ``*_model.json`` -> NetworkSpec, ``*_weights*.h5`` <-> weight dict.

The reference writes both files from its DataGenerator callback (multimodal_fusion/data_generator.py:317-321,
audio_network/data_generator.py, skeletal_network/skeletal_lstm_ctc.py) and reads them back in
multimodal_fusion/multimodal.py:68-85 (pre-trained encoders), audio_network/speech_lstm_ctc_words.py:118-127
(resume) and every ``sequence_decoding.py``.  Keras / h5py are not installed anywhere this package runs, so the JSON
schema (functional-API ``Model`` config: ``layers[].{name,class_name,config,inbound_nodes}``) and the HDF5 layout
(root attr ``layer_names``; per-layer group attr ``weight_names``; datasets named after the TF variables, e.g.
``bidirectional_1/forward_blstm_1/kernel:0``) are restated from Keras 2.1.4 as recalled: the HDF5 container itself
is pinned against libhdf5-written files (h5lite.py), the naming conventions are not (no Keras-written file exists in
the reference).  Loading therefore never relies on names: weighted layers are matched by array shapes, in file order.
"""
import json

import numpy as np

from . import h5lite
from .spec import NetworkSpec


# ------------------------------------------------------------------------------------------------ JSON -> spec
class _T:
    """Symbolic tensor of the Keras graph."""

    def __init__(self, op, args=(), **kw):
        self.op, self.args, self.kw = op, list(args), kw


def _walk_keras_graph(cfg):
    layers = cfg["layers"]
    out = {}

    def inbound(l):
        nodes = l.get("inbound_nodes") or []
        if not nodes:
            return []
        return [out[e[0]] for e in nodes[0]]

    pending = list(layers)
    guard = 0
    while pending and guard < 10 * len(layers) + 10:
        guard += 1
        l = pending.pop(0)
        try:
            ins = inbound(l)
        except KeyError:
            pending.append(l)          # an input of this layer has not been visited yet
            continue
        c, k = l["class_name"], l["config"]
        name = l.get("name", k.get("name"))
        if c == "InputLayer":
            shape = k.get("batch_input_shape") or [None]
            t = _T("input", name=name, shape=shape)
        elif c == "GaussianNoise":
            t = _T("noise", ins, std=float(k.get("stddev", 0.0)))
        elif c == "Bidirectional":
            inner = k["layer"]["config"]
            if k["layer"]["class_name"] != "LSTM" or k.get("merge_mode", "concat") != "concat":
                raise ValueError("only Bidirectional(LSTM, merge_mode='concat') is on the reference path (%s)" % name)
            if inner.get("activation", "tanh") != "tanh" or inner.get("recurrent_activation", "hard_sigmoid") != "hard_sigmoid":
                raise ValueError("LSTM activations other than tanh / hard_sigmoid are not implemented (%s)" % name)
            if float(inner.get("recurrent_dropout", 0.0)) != 0.0:
                raise ValueError("recurrent_dropout is not used by the reference and not implemented (%s)" % name)
            t = _T("bilstm", ins, H=int(inner["units"]), dropout=float(inner.get("dropout", 0.0)),
                   maxnorm=_kernel_maxnorm(inner, name),
                   # the reference renames the wrappers of the transplanted encoders (multimodal.py:123-130); an
                   # auto-named wrapper ("bidirectional_N") is identified by its inner LSTM's name instead
                   name=inner.get("name", name) if name.startswith("bidirectional_") else name, wrapper=name,
                   trainable=bool(k.get("trainable", True)) and bool(inner.get("trainable", True)))
        elif c == "Add":
            t = _T("add", ins)
        elif c in ("Concatenate", "Merge"):
            if c == "Merge" and k.get("mode", "concat") != "concat":
                raise ValueError("Merge mode %r is not implemented" % k.get("mode"))
            t = _T("concat", ins)
        elif c == "Dropout":
            t = _T("dropout", ins, rate=float(k.get("rate", k.get("p", 0.0))), name=name)
        elif c == "Dense":
            t = _T("dense", ins, C=int(k["units"]), name=name)
        elif c == "Activation":
            t = _T("act", ins, fn=k.get("activation"), name=name)
        elif c == "Lambda":
            t = _T("lambda", ins, name=name)
        elif c in ("Model", "Sequential"):
            raise ValueError("nested models are not supported in a Keras JSON (layer %s)" % name)
        else:
            raise ValueError("layer class %s is not on the reference's BiLSTM+CTC path" % c)
        out[name] = t
    if pending:
        raise ValueError("unresolvable inbound nodes for layers %s" % [l.get("name") for l in pending])
    return out


def _kernel_maxnorm(inner, name):
    """kernel_constraint of an LSTM config -> max-norm value (0.0 = unconstrained).  The reference constrains every LSTM input
    kernel with maxnorm(3) over axis 0 (multimodal.py:162, speech_lstm_ctc_words.py:60); anything else is refused rather than
    silently replaced."""
    for other in ("recurrent_constraint", "bias_constraint"):      # (checked FIRST: a layer without a kernel constraint may carry one)
        if inner.get(other) is not None:
            raise ValueError("%s of layer %s is not implemented" % (other, name))
    kc = inner.get("kernel_constraint")
    if kc is None:
        return 0.0
    cls = kc.get("class_name") if isinstance(kc, dict) else kc
    cfg = kc.get("config", {}) if isinstance(kc, dict) else {}
    if cls not in ("MaxNorm", "max_norm", "maxnorm"):
        raise ValueError("kernel_constraint %r of layer %s is not implemented (only MaxNorm)" % (cls, name))
    if int(cfg.get("axis", 0)) != 0:
        raise ValueError("MaxNorm over axis %r (layer %s) is not implemented (the reference uses axis 0)" % (cfg.get("axis"), name))
    return float(cfg.get("max_value", 2))


def _parse_stream(t):
    """t: output tensor of one encoder stream -> stream dict."""
    residual = False
    if t.op == "add":
        a, b = t.args
        # add([lstm_1, lstm_2]) with lstm_2 = f(lstm_1)
        deep, shallow = (b, a) if (b.op == "bilstm" and b.args[0] is a) else (a, b)
        if not (deep.op == "bilstm" and deep.args[0] is shallow and shallow.op == "bilstm"):
            raise ValueError("unsupported residual pattern")
        residual = True
        t = deep
    chain = []
    while t.op == "bilstm":
        chain.append(t)
        t = t.args[0]
    chain.reverse()
    if not chain:
        raise ValueError("a stream must contain at least one Bidirectional(LSTM)")
    noise = 0.0

    def strip(x):
        nonlocal noise
        if x.op == "noise":
            noise = max(noise, x.kw["std"])
            return strip(x.args[0])
        return x
    base = strip(t)
    if base.op == "concat":            # early fusion: concat of (noisy) inputs
        members = [strip(m) for m in base.args]
        if any(m.op != "input" for m in members):
            raise ValueError("unsupported concat below the first LSTM")
        s = {"name": "early_concat", "inputs": [m.kw["name"] for m in members],
             "F": int(sum(m.kw["shape"][-1] for m in members))}
    elif base.op == "input":
        s = {"name": base.kw["name"], "F": int(base.kw["shape"][-1])}
    else:
        raise ValueError("unsupported stream input %s" % base.op)
    s.update({"noise": noise, "residual": residual, "trainable": all(c.kw["trainable"] for c in chain),
              "layers": [{"H": c.kw["H"], "dropout": c.kw["dropout"], "name": c.kw["name"], "maxnorm": c.kw["maxnorm"]}
                         for c in chain]})
    return s


def spec_from_keras_json(text):
    """Keras ``model.to_json()`` of a network of the reference's family -> NetworkSpec (also returns maxlen, Lmax)."""
    d = json.loads(text) if isinstance(text, str) else text
    if d.get("class_name") not in ("Model",):
        raise ValueError("not a Keras functional Model JSON")
    g = _walk_keras_graph(d["config"])
    soft = [t for t in g.values() if t.op == "act" and t.kw["fn"] == "softmax"]
    if len(soft) != 1:
        raise ValueError("expected exactly one softmax Activation")
    t = soft[0].args[0]
    if t.op != "dense":
        raise ValueError("softmax must follow a Dense layer")
    C = t.kw["C"]
    t = t.args[0]
    head = {"C": C, "dropout": 0.0}
    if t.op == "dropout":
        head.update({"dropout": t.kw["rate"], "dropout_name": t.kw["name"]})
        t = t.args[0]
    fusion = None
    is_late = t.op == "bilstm" and t.args[0].op == "concat" and any(m.op in ("add", "bilstm") for m in t.args[0].args)
    if is_late:
        fusion = {"H": t.kw["H"], "dropout": t.kw["dropout"], "name": t.kw["name"], "maxnorm": t.kw["maxnorm"]}
        streams = [_parse_stream(m) for m in t.args[0].args]
    else:
        streams = [_parse_stream(t)]
    name = d["config"].get("name", "model")
    spec = NetworkSpec(streams, fusion, head, name=name)
    maxlen = Lmax = None
    for x in g.values():
        if x.op == "input":
            sh = x.kw["shape"]
            if len(sh) == 3:
                maxlen = sh[1]
            elif x.kw["name"] == "the_labels":
                Lmax = sh[1]
    return spec, maxlen, Lmax


# ------------------------------------------------------------------------------------------------ weights
def keras_layer_order(spec):
    """[(keras layer name, [(variable name, our weight key)])] for every WEIGHTED layer, in Keras ``model.layers``
    order (depth-major: first BiLSTM of every stream, then the second, ..., fusion BiLSTM, Dense)."""
    out = []
    n = 0
    depth = max(len(s["layers"]) for s in spec.streams)
    for k in range(depth):
        for s in spec.streams:
            if k >= len(s["layers"]):
                continue
            n += 1
            out.append(_bilstm_entry("bidirectional_%d" % n, s["layers"][k].get("name", "lstm_%d" % n),
                                     "%s/l%d" % (s["name"], k)))
    if spec.fusion:
        n += 1
        out.append(_bilstm_entry("bidirectional_%d" % n, spec.fusion.get("name", "lstm_%d" % n), "fusion"))
    out.append(("dense_1", [("dense_1/kernel:0", "dense/W"), ("dense_1/bias:0", "dense/b")]))
    return out


def _bilstm_entry(wrapper, inner, prefix):
    ws = []
    for d, kd in (("forward", "fwd"), ("backward", "bwd")):
        for var, key in (("kernel:0", "W"), ("recurrent_kernel:0", "U"), ("bias:0", "b")):
            ws.append(("%s/%s_%s/%s" % (wrapper, d, inner, var), "%s/%s/%s" % (prefix, kd, key)))
    return wrapper, ws


def save_keras_weights(path, spec, weights):
    """weights: our dict (Keras layouts) -> HDF5 file in the Keras 2.1.4 ``save_weights`` layout."""
    layers = [(ln, [(var, np.asarray(weights[key], np.float32)) for var, key in ws]) for ln, ws in keras_layer_order(spec)]
    h5lite.write_keras_weights(path, layers)


def load_keras_weights(path, spec):
    """HDF5 Keras weights file -> our weight dict.  Weighted layers of the file are assigned, in file order, to the
    first not-yet-filled layer of the spec whose array shapes agree (6 arrays = Bidirectional LSTM, 2 = Dense)."""
    table = {n: tuple(sh) for n, sh, _, _ in spec.weight_table()}
    slots = [ws for _, ws in keras_layer_order(spec)]
    free = [True] * len(slots)
    out = {}
    for lname, ws in h5lite.read_keras_weights(path):
        if not ws:
            continue
        shapes = [tuple(np.shape(v)) for _, v in ws]
        for i, slot in enumerate(slots):
            if free[i] and len(slot) == len(ws) and [table[key] for _, key in slot] == shapes:
                free[i] = False
                for (_, key), (_, v) in zip(slot, ws):
                    out[key] = np.asarray(v, np.float32)
                break
        else:
            raise ValueError("%s: layer %s with weight shapes %s matches no layer of the network" % (path, lname, shapes))
    return out
