"""CPU ORACLE (test infrastructure, NOT product code) - whole-network train / predict step.

Restates, on top of ``oracle/keras_ref.py``, the graphs the reference builds:
  * fusion:   multimodal_fusion/multimodal.py:88-213  (two frozen 2-layer BiLSTM encoders with
              residual add, concat, trainable BiLSTM, Dropout, Dense, softmax, CTC Lambda)
  * unimodal: audio_network/speech_lstm_ctc_words.py:42-132 and
              skeletal_network/skeletal_lstm_ctc.py:282-394 (noise, BiLSTM x2 + residual, ...)
PARITY UNPINNED - see keras_ref.py header.  Only tests/, smoke() and bench.py's
cpu_baseline leg may import this.

A network is described by a plain dict ``spec`` (the same shape the product's
``NetworkSpec.to_dict()`` emits):
  {"streams": [{"name","F","noise","layers":[{"H","dropout"}...],"residual":bool,"trainable":bool}],
   "fusion": {"H","dropout","maxnorm"} | None,
   "head": {"dropout","C"}, "ctc": {"skip","eps"},
   "optimizer": {"lr","decay","clipvalue","beta_1","beta_2","epsilon","maxnorm"}}
weights: dict name -> array with Keras layouts; names
  "<stream>/l<k>/<fwd|bwd>/<W|U|b>", "fusion/<fwd|bwd>/<W|U|b>", "dense/W", "dense/b".
rand: dict of host-injected randomness (all optional -> deterministic / eval behaviour):
  "<stream>/noise" (B,T,F), "<stream>/l<k>/<fwd|bwd>/mask" (4,B,F_in),
  "fusion/<fwd|bwd>/mask" (4,B,F_in), "head/mask" (B,T,D).
"""
from __future__ import annotations

import numpy as np

from . import keras_ref as kr


def weight_names(spec):
    """Canonical ordered list of (name, shape, trainable, kind)."""
    out = []
    width = 0
    for s in spec["streams"]:
        fin = s["F"]
        for k, lay in enumerate(s["layers"]):
            H = lay["H"]
            for d in ("fwd", "bwd"):
                p = "%s/l%d/%s" % (s["name"], k, d)
                out.append((p + "/W", (fin, 4 * H), s["trainable"], "kernel"))
                out.append((p + "/U", (H, 4 * H), s["trainable"], "recurrent"))
                out.append((p + "/b", (4 * H,), s["trainable"], "bias"))
            fin = 2 * H
        width += fin
    if spec.get("fusion"):
        H = spec["fusion"]["H"]
        for d in ("fwd", "bwd"):
            p = "fusion/%s" % d
            out.append((p + "/W", (width, 4 * H), True, "kernel"))
            out.append((p + "/U", (H, 4 * H), True, "recurrent"))
            out.append((p + "/b", (4 * H,), True, "bias"))
        width = 2 * H
    C = spec["head"]["C"]
    out.append(("dense/W", (width, C), True, "dense"))
    out.append(("dense/b", (C,), True, "dense_bias"))
    return out


def init_weights(spec, rng, dtype=np.float64):
    """SURVEY 8(d) recipe: kernels U(-.05,.05), recurrent orthogonal, bias 0 with forget block 1."""
    w = {}
    for name, shape, _, kind in weight_names(spec):
        if kind in ("kernel", "dense"):
            w[name] = rng.uniform(-0.05, 0.05, size=shape).astype(dtype)
        elif kind == "recurrent":
            H = shape[0]
            blocks = []
            for _ in range(4):
                q, r = np.linalg.qr(rng.standard_normal((H, H)))
                blocks.append(q * np.sign(np.diag(r)))
            w[name] = np.concatenate(blocks, axis=1).astype(dtype)
        elif kind == "bias":
            H = shape[0] // 4
            b = np.zeros(shape, dtype)
            b[H:2 * H] = 1.0
            w[name] = b
        else:
            w[name] = np.zeros(shape, dtype)
    return w


def draw_rand(spec, B, T, rng, dtype=np.float64, train=True):
    """Host-side draw of every random tensor a training-phase forward consumes (App. A.2/A.4)."""
    r = {}
    if not train:
        return r

    def mask(shape, p):
        if p <= 0:
            return None
        return ((rng.random(shape) >= p).astype(dtype) / dtype(1.0 - p)).astype(dtype)

    width = 0
    for s in spec["streams"]:
        if s.get("noise", 0.0) > 0:
            r[s["name"] + "/noise"] = (rng.standard_normal((B, T, s["F"])) * s["noise"]).astype(dtype)
        fin = s["F"]
        for k, lay in enumerate(s["layers"]):
            for d in ("fwd", "bwd"):
                m = mask((4, B, fin), lay.get("dropout", 0.0))
                if m is not None:
                    r["%s/l%d/%s/mask" % (s["name"], k, d)] = m
            fin = 2 * lay["H"]
        width += fin
    if spec.get("fusion"):
        for d in ("fwd", "bwd"):
            m = mask((4, B, width), spec["fusion"].get("dropout", 0.0))
            if m is not None:
                r["fusion/%s/mask" % d] = m
        width = 2 * spec["fusion"]["H"]
    m = mask((B, T, width), spec["head"].get("dropout", 0.0))
    if m is not None:
        r["head/mask"] = m
    return r


def _wb(w, prefix):
    return (w[prefix + "/W"], w[prefix + "/U"], w[prefix + "/b"])


def forward(spec, w, inputs, rand=None):
    """inputs: dict stream-name -> (B,T,F).  Returns P (B,T,C) and caches."""
    rand = rand or {}
    caches = {}
    outs = []
    for s in spec["streams"]:
        x = inputs[s["name"]]
        nz = rand.get(s["name"] + "/noise")
        if nz is not None:
            x = x + nz  # GaussianNoise, multimodal.py:103-106
        ys = []
        cur = x
        for k in range(len(s["layers"])):
            p = "%s/l%d" % (s["name"], k)
            y, c = kr.bilstm_forward(cur, _wb(w, p + "/fwd"), _wb(w, p + "/bwd"),
                                     rand.get(p + "/fwd/mask"), rand.get(p + "/bwd/mask"))
            caches[p] = c
            ys.append(y)
            cur = y
        if s.get("residual") and len(ys) == 2:
            out = ys[0] + ys[1]  # layers.add, multimodal.py:111,117
        else:
            out = ys[-1]
        outs.append(out)
    feat = np.concatenate(outs, axis=2) if len(outs) > 1 else outs[0]  # Merge(concat), :155
    if spec.get("fusion"):
        feat, c = kr.bilstm_forward(feat, _wb(w, "fusion/fwd"), _wb(w, "fusion/bwd"),
                                    rand.get("fusion/fwd/mask"), rand.get("fusion/bwd/mask"))
        caches["fusion"] = c
    P, c = kr.dense_softmax_forward(feat, rand.get("head/mask"), w["dense/W"], w["dense/b"])
    caches["head"] = c
    return P, caches


def loss_and_grads(spec, w, inputs, labels, input_length, label_length, rand=None):
    """Mean CTC loss over the batch (Keras averages the dummy loss, multimodal.py:212) and
    gradients of that mean w.r.t. every trainable weight. Returns (loss_mean, loss_b, grads, P)."""
    P, caches = forward(spec, w, inputs, rand)
    B = P.shape[0]
    ctc = spec.get("ctc", {})
    loss_b, dz = kr.ctc_loss_grad(P, labels, input_length, label_length,
                                  skip=ctc.get("skip", 2), eps=ctc.get("eps", 1e-8))
    dz = dz / B
    grads = {}
    da, grads["dense/W"], grads["dense/b"] = kr.dense_backward(dz, caches["head"])
    any_trainable_stream = any(s["trainable"] for s in spec["streams"])
    if spec.get("fusion"):
        da, gf, gb = kr.bilstm_backward(da, caches["fusion"], need_dx=any_trainable_stream)
        for d, g in (("fwd", gf), ("bwd", gb)):
            grads["fusion/%s/W" % d], grads["fusion/%s/U" % d], grads["fusion/%s/b" % d] = g
    if any_trainable_stream:
        off = 0
        for s in spec["streams"]:
            wout = 2 * s["layers"][-1]["H"]
            dout = da[:, :, off:off + wout]
            off += wout
            if not s["trainable"]:
                continue
            nl = len(s["layers"])
            if nl == 2:
                p1 = "%s/l1" % s["name"]
                dx2, gf, gb = kr.bilstm_backward(dout, caches[p1], need_dx=True)
                for d, g in (("fwd", gf), ("bwd", gb)):
                    grads[p1 + "/%s/W" % d], grads[p1 + "/%s/U" % d], grads[p1 + "/%s/b" % d] = g
                dy1 = dx2 + dout if s.get("residual") else dx2
            else:
                dy1 = dout
            p0 = "%s/l0" % s["name"]
            _, gf, gb = kr.bilstm_backward(dy1, caches[p0], need_dx=False)
            for d, g in (("fwd", gf), ("bwd", gb)):
                grads[p0 + "/%s/W" % d], grads[p0 + "/%s/U" % d], grads[p0 + "/%s/b" % d] = g
    return float(loss_b.mean()), loss_b, grads, P


class Trainer:
    """Adam(clipvalue) + maxnorm state for the trainable weights (App. A.6)."""

    def __init__(self, spec, w):
        self.spec = spec
        self.w = w
        self.names = [n for n, _, tr, _ in weight_names(spec) if tr]
        self.kinds = {n: k for n, _, _, k in weight_names(spec)}
        self.m = {n: np.zeros_like(w[n]) for n in self.names}
        self.v = {n: np.zeros_like(w[n]) for n in self.names}
        self.iterations = 0
        o = spec.get("optimizer", {})
        self.lr = o.get("lr", 1e-4)
        self.decay = o.get("decay", 0.0)
        self.clipvalue = o.get("clipvalue", 0.5)
        self.b1 = o.get("beta_1", 0.9)
        self.b2 = o.get("beta_2", 0.999)
        self.eps = o.get("epsilon", 1e-7)
        self.maxnorm = o.get("maxnorm", 3.0)

    def apply(self, grads, gscale=1.0):
        lr_t = kr.adam_lr_t(self.lr, self.decay, self.iterations, self.b1, self.b2)
        for n in self.names:
            kr.adam_step(self.w[n], grads[n], self.m[n], self.v[n], lr_t, self.b1, self.b2,
                         self.eps, self.clipvalue, gscale)
            if self.kinds[n] == "kernel" and self.maxnorm and self.maxnorm > 0:
                kr.maxnorm_cols(self.w[n], self.maxnorm)
        self.iterations += 1

    def train_on_batch(self, inputs, labels, input_length, label_length, rand=None):
        loss, loss_b, grads, _ = loss_and_grads(self.spec, self.w, inputs, labels,
                                                input_length, label_length, rand)
        self.apply(grads)
        return loss


# ----------------------------------------------------------------------------------------
# synthetic ChaLearn-shaped batches (SURVEY 8(d) recipe; shapes of DataGenerator.get_batch,
# multimodal_fusion/data_generator.py:157-278)
# ----------------------------------------------------------------------------------------
def synthetic_batch(spec, B, T, Lmax, rng, dtype=np.float64, lmin=8, lmax=20):
    C = spec["head"]["C"]
    lmax = min(lmax, Lmax, max(1, (T - 2) // 2))
    lmin = min(lmin, lmax)
    inputs = {}
    n = rng.integers(int(np.ceil(0.6 * T)), T + 1, size=B)
    for s in spec["streams"]:
        scale = 3.0 if s["name"].startswith("audio") or s["name"] == "speech" else 1.0
        x = (rng.standard_normal((B, T, s["F"])) * scale).astype(dtype)
        for b in range(B):
            x[b, n[b]:, :] = 0.0
        inputs[s["name"]] = x
    labels = -np.ones((B, Lmax), dtype)
    label_length = np.zeros((B, 1), np.int64)
    for b in range(B):
        L = int(rng.integers(lmin, lmax + 1))
        seq = rng.integers(1, C - 1, size=L)
        oov = rng.random(L) < 0.05
        seq = np.where(oov, 0, seq)
        labels[b, :L] = seq
        label_length[b, 0] = L
    input_length = np.full((B, 1), T - 2, np.int64)
    return inputs, labels, input_length, label_length
