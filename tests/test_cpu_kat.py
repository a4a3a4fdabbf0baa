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
