"""Network description shared by the model builders and the device engine.

A network of the reference's family is: per-modality encoder streams (GaussianNoise -> stacked
Bidirectional(LSTM) [-> residual add]) -> concat -> optional fusion Bidirectional(LSTM) -> Dropout ->
Dense -> softmax -> CTC (reference multimodal_fusion/multimodal.py:88-213,
audio_network/speech_lstm_ctc_words.py:42-132, skeletal_network/skeletal_lstm_ctc.py:282-394).
"""
import copy
import json


class NetworkSpec:
    def __init__(self, streams, fusion, head, ctc=None, optimizer=None, name="model"):
        self.streams = copy.deepcopy(streams)
        self.fusion = copy.deepcopy(fusion)
        self.head = copy.deepcopy(head)
        self.ctc = dict({"skip": 2, "eps": 1e-8}, **(ctc or {}))
        self.optimizer = dict({"lr": 1e-4, "decay": 0.0, "clipvalue": 0.5, "beta_1": 0.9, "beta_2": 0.999,
                               "epsilon": 1e-7, "maxnorm": 3.0}, **(optimizer or {}))
        self.name = name
        for s in self.streams:
            s.setdefault("noise", 0.0)
            s.setdefault("residual", len(s["layers"]) == 2)
            s.setdefault("trainable", True)
            for lay in s["layers"]:
                lay.setdefault("dropout", 0.0)
        if self.fusion:
            self.fusion.setdefault("dropout", 0.0)
        self.head.setdefault("dropout", 0.0)

    # ------------------------------------------------------------------------------------------
    def to_dict(self):
        return {"name": self.name, "streams": copy.deepcopy(self.streams), "fusion": copy.deepcopy(self.fusion),
                "head": copy.deepcopy(self.head), "ctc": dict(self.ctc), "optimizer": dict(self.optimizer)}

    @classmethod
    def from_dict(cls, d):
        return cls(d["streams"], d.get("fusion"), d["head"], d.get("ctc"), d.get("optimizer"), d.get("name", "model"))

    def to_json(self):
        return json.dumps({"class_name": "MgrNetwork", "config": self.to_dict()}, indent=1)

    @classmethod
    def from_json(cls, text):
        d = json.loads(text)
        return cls.from_dict(d["config"] if "config" in d else d)

    # ------------------------------------------------------------------------------------------
    def stream_width(self, s):
        return 2 * s["layers"][-1]["H"]

    @property
    def concat_width(self):
        return sum(self.stream_width(s) for s in self.streams)

    @property
    def head_width(self):
        return 2 * self.fusion["H"] if self.fusion else self.concat_width

    @property
    def num_classes(self):
        return self.head["C"]

    def lstm_layers(self):
        """Yield (prefix, F_in, H, dropout, trainable) for every Bidirectional layer in forward order."""
        for s in self.streams:
            fin = s["F"]
            for k, lay in enumerate(s["layers"]):
                yield ("%s/l%d" % (s["name"], k), fin, lay["H"], lay["dropout"], s["trainable"])
                fin = 2 * lay["H"]
        if self.fusion:
            yield ("fusion", self.concat_width, self.fusion["H"], self.fusion["dropout"], True)

    def kernel_maxnorm(self, prefix):
        """Max-norm (over axis 0, per column) of the input kernel of the Bidirectional layer `prefix` ("<stream>/l<k>" or
        "fusion"); 0.0 = unconstrained.  A layer's own "maxnorm" entry (set when the model came from a Keras JSON) wins over
        the optimizer-wide default, which is what the reference's builders use for every LSTM (maxnorm(3))."""
        lay = self.fusion if prefix == "fusion" else None
        if lay is None:
            sname, k = prefix.rsplit("/l", 1)
            lay = next(s for s in self.streams if s["name"] == sname)["layers"][int(k)]
        v = lay.get("maxnorm")
        return float(self.optimizer.get("maxnorm") or 0.0) if v is None else float(v)

    def weight_table(self):
        """Ordered (name, keras_shape, trainable, kind) - Keras weight-list order: fwd W,U,b then bwd W,U,b."""
        out = []
        for prefix, fin, H, _, tr in self.lstm_layers():
            for d in ("fwd", "bwd"):
                out.append(("%s/%s/W" % (prefix, d), (fin, 4 * H), tr, "kernel"))
                out.append(("%s/%s/U" % (prefix, d), (H, 4 * H), tr, "recurrent"))
                out.append(("%s/%s/b" % (prefix, d), (4 * H,), tr, "bias"))
        out.append(("dense/W", (self.head_width, self.head["C"]), True, "dense"))
        out.append(("dense/b", (self.head["C"],), True, "dense_bias"))
        return out

    def count_params(self, trainable_only=False):
        n = 0
        for _, shape, tr, _ in self.weight_table():
            if trainable_only and not tr:
                continue
            k = 1
            for s in shape:
                k *= s
            n += k
        return n

    def flops_per_frame(self, executed=False):
        """Algorithmic FLOP per padded frame of one training step (2 x MAC), SURVEY 8(d) accounting:
        forward of every layer, backward (dZ.U^T, dW, dU [, dX]) of trainable ones, Dense fwd+bwd.
        executed=True: what the device really multiplies in a training step - the dropout-aware projection and dW kernels
        (used from p >= 0.3 on, 16 <= F <= 2048) skip the products with dropped features, i.e. run the (1 - p) share of
        their K loops; the skipped terms are exact zeros, the result is the dense one."""
        mac = 0
        for prefix, fin, H, p, tr in self.lstm_layers():
            keep = (1.0 - p) if (executed and p >= 0.3 and 16 <= fin <= 2048) else 1.0
            proj = fin * 4 * H
            fwd = 2 * (proj * keep + H * 4 * H)
            mac += fwd
            if tr:
                bwd = 2 * (proj * keep + 2 * H * 4 * H)  # dW + (dh_rec, dU)
                first = prefix.endswith("/l0")
                if not first and prefix != "fusion":
                    bwd += 2 * fin * 4 * H  # dX to a trainable layer below
                elif prefix == "fusion" and any(s["trainable"] for s in self.streams):
                    bwd += 2 * fin * 4 * H
                mac += bwd
        mac += self.head_width * self.head["C"] * 3
        return 2 * mac if executed else int(2 * mac)
