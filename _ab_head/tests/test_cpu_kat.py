"""CPU suite: the oracle against THIRD-PARTY known-answer vectors (tests/golden/thirdparty_kat.json: TensorFlow's
ctc_loss_op_test / ctc_decoder_ops_test and Keras' backend_test, the functions behind K.ctc_batch_cost at
/root/reference/multimodal_fusion/losses.py:13) and against exhaustive enumeration - evidence that does not rest on code
written for this repository."""
import itertools
import json
import math
import os

import numpy as np
import pytest

from oracle import keras_ref as kr
from tests.helpers import GOLDEN

KAT = json.load(open(os.path.join(GOLDEN, "thirdparty_kat.json")))


def exhaustive_labelling_probs(y, blank):
    """P(labelling) for every labelling, by enumerating all C^T frame paths of the (T, C) probability matrix y."""
    T, C = y.shape
    tot = {}
    for path in itertools.product(range(C), repeat=T):
        p = 1.0
        for t, c in enumerate(path):
            p *= y[t, c]
        lab = tuple(k for k, _ in itertools.groupby(path) if k != blank)
        tot[lab] = tot.get(lab, 0.0) + p
    return tot


def test_vectors_verify_themselves():
    """A recalled vector is only evidence if it is internally consistent: rows are probability distributions, the published
    gradient is (prob - occupancy) - so each row of it sums to 0 and it differs from prob only in the classes of the target."""
    k = KAT["ctc_loss_testBasic"]
    P, G = np.array(k["probs"]), np.array(k["grad_wrt_logits"])
    assert np.allclose(P.sum(-1), 1.0, atol=3e-6)
    assert np.allclose(G.sum(-1), 0.0, atol=3e-6)
    for b, lab in enumerate(k["labels"]):
        used = set(c for c in lab if c >= 0) | {k["blank"]}
        for c in range(k["num_classes"]):
            if c not in used:
                assert np.array_equal(G[b, :, c], P[b, :, c])
    # entry 0 has as many labels as frames: ONE alignment, its probability is a plain product
    p0 = np.prod([P[0, t, c] for t, c in enumerate(k["labels"][0])])
    assert abs(-math.log(p0) - k["loss"][0]) < 1e-5
    assert np.allclose(np.array(KAT["ctc_decode_beam"]["probs"]).sum(-1), 1.0, atol=3e-6)


def test_oracle_ctc_reproduces_tensorflow_testBasic():
    """tf.nn.ctc_loss semantics (logits = log p, so eps = 0): losses and d loss / d logits to the published precision."""
    k = KAT["ctc_loss_testBasic"]
    P = np.array(k["probs"], np.float64)
    loss, g = kr.ctc_loss_grad(P, np.array(k["labels"]), k["input_length"], k["label_length"], skip=0, blank=k["blank"], eps=0.0)
    assert np.allclose(loss, k["loss"], rtol=0, atol=k["loss_atol"])
    assert np.abs(g - np.array(k["grad_wrt_logits"])).max() < k["grad_atol"]
    # exhaustive enumeration agrees with both
    for b in range(2):
        lab = tuple(k["labels"][b][:k["label_length"][b]])
        y = P[b] / P[b].sum(-1, keepdims=True)
        assert abs(-math.log(exhaustive_labelling_probs(y, k["blank"])[lab]) - k["loss"][b]) < 1e-5


def test_oracle_ctc_reproduces_keras_test_ctc():
    """K.ctc_batch_cost path as the reference calls it (log(y + 1e-8) -> TF softmax): Keras' own test tolerance, atol 1e-5."""
    k = KAT["ctc_loss_testBasic"]
    P = np.array(k["probs"], np.float32)
    loss, _ = kr.ctc_loss_grad(P, np.array(k["labels"], np.float32), np.array(k["input_length"])[:, None],
                               np.array(k["label_length"])[:, None], skip=0, eps=1e-8, need_grad=False)
    assert np.allclose(loss, k["loss"], rtol=0, atol=k["loss_atol"])


def _greedy(P, lengths, blank):
    out, nlp = [], []
    for b, n in enumerate(lengths):
        best = np.argmax(P[b, :n], -1)
        out.append([int(c) for c, _ in itertools.groupby(best) if c != blank])
        nlp.append(-float(np.sum(np.log(P[b, :n].max(-1)))))
    return out, nlp


def test_keras_ctc_decode_greedy_vector():
    k = KAT["ctc_decode_greedy"]
    P = np.array(k["probs"])
    dec, nlp = _greedy(P, k["input_length"], k["blank"])
    assert dec == k["decoded"]
    assert np.allclose(nlp, [-math.log(1.0 * 0.6 * 0.6 * 0.9), -5 * math.log(0.9)])
    # best PATH is not best LABELLING: for entry 0 the labelling [0, 2, 1] collects three alignments (0.16 + 0.24 + 0.24) x 0.9
    # = 0.576 against 0.324 for the greedy answer [0, 1]; a wide beam search must find it, exhaustive enumeration agrees
    full, sc = kr.ctc_beam_search(P, k["input_length"], beam_width=16, skip=0, blank=k["blank"], eps=0.0, merge_repeated=False)
    assert full == [[0, 2, 1], [1, 1, 0]]
    ex = exhaustive_labelling_probs(P[0, :4], k["blank"])
    assert max(ex, key=ex.get) == (0, 2, 1) and abs(ex[(0, 2, 1)] - 0.576) < 1e-12 and abs(sc[0] - math.log(0.576)) < 1e-9


def test_oracle_beam_search_reproduces_tensorflow_beam_vector():
    k = KAT["ctc_decode_beam"]
    P = np.array(k["probs"], np.float64)
    seqs, scores = kr.ctc_beam_search(P, k["input_length"], beam_width=k["beam_width"], skip=0, blank=k["blank"], eps=0.0,
                                      merge_repeated=True, top_paths=k["top_paths"])
    assert seqs[0] == k["decoded_top_paths"]
    assert scores[0][0] > scores[0][1]
    # ... and that is the PRUNED answer: exhaustively, [0, 1, 0] is the most probable labelling, which a wide beam finds
    ex = exhaustive_labelling_probs(P[0, :5] / P[0, :5].sum(-1, keepdims=True), k["blank"])
    best = max(ex, key=ex.get)
    assert best == (0, 1, 0) and abs(ex[best] - 0.110429) < 1e-6 and abs(ex[(1, 0)] - 0.100626) < 1e-6
    wide, sc = kr.ctc_beam_search(P, k["input_length"], beam_width=64, skip=0, blank=k["blank"], eps=0.0, merge_repeated=False)
    assert wide[0] == [0, 1, 0] and abs(sc[0] - math.log(ex[best])) < 1e-9


@pytest.mark.parametrize("seed", range(12))
def test_oracle_beam_search_equals_exhaustive_enumeration(seed):
    """Independent of any library: with a beam wide enough never to prune, prefix beam search must return the most probable
    LABELLING and exactly its probability (sum over all its alignments), for random tiny (T, C) - including peaky rows, ties in
    the arg-max path and repeated labels."""
    rng = np.random.default_rng(seed)
    T, C = int(rng.integers(2, 6)), int(rng.integers(2, 5))
    P = rng.random((1, T, C)) ** (1 + 3 * rng.random())
    P /= P.sum(-1, keepdims=True)
    ex = exhaustive_labelling_probs(P[0], C - 1)
    ranked = sorted(ex.items(), key=lambda kv: -kv[1])
    seqs, scores = kr.ctc_beam_search(P, [T], beam_width=400, skip=0, eps=0.0, merge_repeated=False, top_paths=3)
    for r in range(min(3, len(ranked), len(seqs[0]))):
        assert tuple(seqs[0][r]) == ranked[r][0], (r, seqs[0], ranked[:3])
        assert abs(scores[0][r] - math.log(ranked[r][1])) < 1e-9
    assert abs(sum(ex.values()) - 1.0) < 1e-12


# ---- round 3: max-norm, pad_sequences, hard_sigmoid (Keras constraints_test / sequence_test / activations_test) ----------------
def _maxnorm_arrays():
    k = KAT["max_norm_explicit"]
    x = np.array(k["x_columns"], np.float64).T                       # the Keras test transposes: listed rows are columns
    tgt = np.array([[eval(v.replace("sqrt", "math.sqrt")) if isinstance(v, str) else v for v in col]
                    for col in k["target_columns"]], np.float64).T
    return k, x, tgt


def test_max_norm_vector_verifies_itself_and_pins_the_oracle():
    """keras constraints.max_norm(2.0), axis 0: each column keeps its direction and gets norm min(norm, 2) - and the oracle's
    maxnorm_cols (what oracle/network_ref.py applies after Adam, multimodal.py:164) reproduces the published target."""
    k, x, tgt = _maxnorm_arrays()
    assert x.shape == (3, 4)
    n_in, n_out = np.linalg.norm(x, axis=0), np.linalg.norm(tgt, axis=0)
    assert np.allclose(n_out, np.minimum(n_in, k["max_value"]), rtol=1e-12)
    nz = n_in > 0
    assert np.allclose(tgt[:, nz] / n_out[nz], x[:, nz] / n_in[nz], rtol=1e-12)
    for dt in (np.float64, np.float32):
        w = x.astype(dt)
        kr.maxnorm_cols(w, maxv=k["max_value"])          # (in place, like the constraint applied to a variable)
        np.testing.assert_allclose(w, tgt, rtol=k["rtol"], atol=1e-12)


def test_pad_sequences_vectors_pin_the_data_generator_padding():
    """The published vectors first verify themselves (the formula Keras documents), then pin the two padding rules of the
    reference's generator as this build implements them: features post/post (datagen.pad_post) and label rows padded behind
    with -1, truncated in FRONT (BaseDataGenerator, Keras' default truncating='pre')."""
    from mgr_amd.datagen import pad_post
    k = KAT["pad_sequences"]

    def keras_pad(seqs, maxlen, padding="pre", truncating="pre", value=0.0):   # the documented algorithm, for the self-check
        out = []
        for s in seqs:
            s = np.asarray(s, np.float64)
            s = s[-maxlen:] if truncating == "pre" else s[:maxlen]
            pad = np.full((maxlen - len(s),) + s.shape[1:], value)
            out.append(np.concatenate([pad, s]) if padding == "pre" else np.concatenate([s, pad]))
        return np.array(out)

    a, av = k["a"], k["a_vector"]
    assert np.array_equal(keras_pad(a, 3), k["maxlen3_padding_pre"])
    assert np.array_equal(keras_pad(a, 3, padding="post"), k["maxlen3_padding_post"])
    assert np.array_equal(keras_pad(a, 2, truncating="pre"), k["maxlen2_truncating_pre"])
    assert np.array_equal(keras_pad(a, 2, truncating="post"), k["maxlen2_truncating_post"])
    assert np.array_equal(keras_pad(a, 3, value=1), k["maxlen3_value1"])
    assert np.array_equal(keras_pad(av, 3, padding="post"), k["vector_maxlen3_padding_post"])
    assert np.array_equal(keras_pad(av, 2, truncating="post"), k["vector_maxlen2_truncating_post"])
    # features: padding='post', truncating='post'
    for s, want in zip(av, k["vector_maxlen3_padding_post"]):
        got = pad_post(s, 3)
        assert got.dtype == np.float32 and np.array_equal(got, want)
    for s, want in zip(av[1:], k["vector_maxlen2_truncating_post"][1:]):     # rows that are not padded: only the truncation shows
        assert np.array_equal(pad_post(s, 2), want)
    assert np.array_equal(pad_post(av[0], 2), [[1, 1], [0, 0]])              # post/post of the short row (both rules combined)
    for s, want in zip(a, k["maxlen3_padding_post"]):
        assert np.array_equal(pad_post(np.array(s, np.float32)[:, None], 3)[:, 0], want)


def test_pad_sequences_vectors_pin_the_generator_batches():
    """The same rules through the product's DataGenerator.get_batch: a store whose files hold the published sequences."""
    from mgr_amd.datagen import BaseDataGenerator
    k = KAT["pad_sequences"]

    class Store:
        def file_ids(self):
            return [1, 2, 3]

        def features(self, fid, modality):
            return np.array(k["a_vector"][fid - 1], np.float64)

        def labels(self, fid):
            return np.array(k["a"][fid - 1], np.float32)

    class Gen(BaseDataGenerator):
        streams = (("x", "m", "feat_dim"),)
        feat_dim = 2

    g = Gen()
    g._setup(minibatch_size=3, maxlen=2, nb_classes=6, dataset="val", val_split=0.0, absolute_max_sequence_len=2, store=Store())
    assert g.get_file_list(False) == [1, 2, 3]
    inputs, _ = g.get_batch(train=False)
    assert np.array_equal(inputs["x"][1:], np.array(k["vector_maxlen2_truncating_post"], np.float64)[1:])
    assert np.array_equal(inputs["x"][0], [[1, 1], [0, 0]])                  # padding='post' of the short row
    assert np.array_equal(inputs["the_labels"], [[1, -1], [1, 2], [2, 3]])   # value -1 behind, Keras' default truncating='pre'
    assert np.array_equal(inputs["label_length"][:, 0], [1, 2, 2]) and np.array_equal(inputs["input_length"][:, 0], [0, 0, 0])


def test_hard_sigmoid_vector_pins_the_oracle():
    k = KAT["hard_sigmoid"]
    for xs, want in ((k["standard_values"], k["expected_standard"]), (k["edge_values"], k["expected_edge"])):
        x = np.array(xs, np.float64)
        assert np.allclose(np.clip(0.2 * x + 0.5, 0.0, 1.0), want, rtol=1e-12)          # the vector is the published formula
        np.testing.assert_allclose(kr.hard_sigmoid(x), want, rtol=k["rtol"])
        np.testing.assert_allclose(kr.hard_sigmoid(x.astype(np.float32)), want, rtol=k["rtol"])


def test_oracle_ctc_accepts_an_empty_label_sequence():
    """oracle/keras_ref.py::ctc_loss_grad with label_length 0 (TF's ctc_loss accepts it): closed form - sum_t log y_t(blank), and
    the gradient matches central differences."""
    rng = np.random.default_rng(0)
    B, T, Cn, Lmax = 2, 12, 6, 4
    z = rng.standard_normal((B, T, Cn))

    def probs(zz):
        e = np.exp(zz - zz.max(-1, keepdims=True))
        return e / e.sum(-1, keepdims=True)
    labels = -np.ones((B, Lmax))
    labels[1, :2] = [1, 3]
    ll, il = np.array([0, 2]), np.full(B, T - 2)
    loss, dz = kr.ctc_loss_grad(probs(z), labels, il, ll)
    u = probs(z)[0, 2:2 + il[0]] + 1e-8
    assert abs(loss[0] + np.log(u[:, Cn - 1] / u.sum(-1)).sum()) < 1e-10
    g = np.zeros((T, Cn))
    for t in range(T):
        for c in range(Cn):
            zp, zm = z.copy(), z.copy()
            zp[0, t, c] += 1e-6
            zm[0, t, c] -= 1e-6
            g[t, c] = (kr.ctc_loss_grad(probs(zp), labels, il, ll, need_grad=False)[0].sum()
                       - kr.ctc_loss_grad(probs(zm), labels, il, ll, need_grad=False)[0].sum()) / 2e-6
    assert np.abs(g - dz[0]).max() < 1e-7
