"""CPU suite (no GPU): the oracle against the committed golden vectors and against first principles."""
import json
import os

import numpy as np
import pytest

from oracle import keras_ref as kr
from oracle import network_ref as nr
from tests.helpers import GOLDEN, load_case


def test_ctc_oracle_reproduces_golden():
    z = np.load(GOLDEN + "/ctc_small.npz")
    loss, dz = kr.ctc_loss_grad(z["P"], z["labels"], z["input_length"], z["label_length"])
    assert np.allclose(loss, z["loss"], rtol=1e-12)
    assert np.allclose(dz, z["dlogits"], rtol=1e-10, atol=1e-14)


def test_ctc_oracle_brute_force_tiny():
    """-log sum over all alignments, enumerated, equals the DP (independent of any library)."""
    import itertools
    rng = np.random.default_rng(0)
    T, C = 5, 3
    P = rng.random((1, T + 2, C))
    P /= P.sum(-1, keepdims=True)
    lab = [0, 1]
    eps = 1e-8
    y = (P[0, 2:] + eps) / (P[0, 2:] + eps).sum(-1, keepdims=True)
    tot = 0.0
    for path in itertools.product(range(C), repeat=T):
        col = [k for k, _ in itertools.groupby(path)]
        col = [k for k in col if k != C - 1]
        if col == lab:
            tot += np.prod([y[t, c] for t, c in enumerate(path)])
    loss, _ = kr.ctc_loss_grad(P, np.array([[0., 1., -1.]]), [T], [2])
    assert abs(loss[0] + np.log(tot)) < 1e-10


@pytest.mark.parametrize("reverse", [False, True])
def test_lstm_oracle_gradients_finite_difference(reverse):
    rng = np.random.default_rng(1)
    B, T, F, H = 2, 5, 3, 4
    x = rng.standard_normal((B, T, F))
    W = rng.uniform(-.5, .5, (F, 4 * H))
    U = rng.uniform(-.5, .5, (H, 4 * H))
    b = rng.uniform(-.2, .2, 4 * H)
    mask = ((rng.random((4, B, F)) > 0.4) / 0.6)
    dy = rng.standard_normal((B, T, H))

    def f(W_, U_, b_, x_):
        y, _ = kr.lstm_forward(x_, W_, U_, b_, mask, reverse)
        return float((y * dy).sum())

    _, cache = kr.lstm_forward(x, W, U, b, mask, reverse)
    dx, dW, dU, db = kr.lstm_backward(dy, cache)
    for arr, g, name in ((W, dW, "W"), (U, dU, "U"), (b, db, "b"), (x, dx, "x")):
        for idx in [tuple(rng.integers(0, s) for s in arr.shape) for _ in range(5)]:
            ap, am = arr.copy(), arr.copy()
            ap[idx] += 1e-6
            am[idx] -= 1e-6
            args = {"W": (W, U, b, x), "U": (W, U, b, x), "b": (W, U, b, x), "x": (W, U, b, x)}[name]
            def sub(a):
                l = list(args)
                l["WUbx".index(name)] = a
                return f(*l)
            fd = (sub(ap) - sub(am)) / 2e-6
            assert abs(fd - g[idx]) < 1e-6 * max(1, abs(fd)), (name, idx)


@pytest.mark.parametrize("case", ["fusion_tiny", "unimodal_tiny"])
def test_network_oracle_reproduces_golden(case):
    z, meta, grab = load_case(case)
    loss, loss_b, grads, P = nr.loss_and_grads(meta["spec"], grab("w__"), grab("x__"), z["labels"], z["input_length"],
                                               z["label_length"], grab("r__"))
    assert abs(loss - float(z["loss"])) < 1e-12 * abs(loss)
    assert np.allclose(P, z["P"], rtol=1e-12)
    for k, g in grab("g__").items():
        assert np.allclose(grads[k], g, rtol=1e-10, atol=1e-14), k
    tr = nr.Trainer(meta["spec"], {k: v.copy() for k, v in grab("w__").items()})
    traj = [tr.train_on_batch(grab("x__"), z["labels"], z["input_length"], z["label_length"], grab("rs%d__" % s))
            for s in range(meta["steps"])]
    assert np.allclose(traj, z["traj"], rtol=1e-12)
    for k, v in grab("wfinal__").items():
        assert np.allclose(tr.w[k], v, rtol=1e-12, atol=1e-15), k


def test_decode_oracle_reproduces_golden_and_quirk():
    z = np.load(GOLDEN + "/decode_small.npz")
    unpad = lambda a: [[int(v) for v in r if v >= 0] for r in a]
    assert kr.greedy_decode_quirk(z["P"], 0.5) == unpad(z["greedy_thr05"])
    assert kr.greedy_decode_quirk(z["P"], 0.75) == unpad(z["greedy_thr075"])
    seqs, sc = kr.ctc_beam_search(z["P"], np.full(z["P"].shape[0], z["P"].shape[1] - 2), beam_width=10)
    assert seqs == unpad(z["beam10"]) and np.allclose(sc, z["beam10_score"], rtol=1e-13)
    # the quirk: a low-confidence frame removes the FIRST occurrence of its label, not itself
    P = np.zeros((1, 7, 3), np.float32)
    best = [0, 1, 0, 0, 1]
    conf = [0.9, 0.9, 0.9, 0.4, 0.9]  # frame 3 (label 0) is weak -> the first label-0 frame disappears
    for t, (c, p) in enumerate(zip(best, conf)):
        P[0, 2 + t, :] = (1 - p) / 2
        P[0, 2 + t, c] = p
    assert kr.greedy_decode_quirk(P, 0.5) == [[1, 0, 1]]


def test_beam_width_1_equals_bestpath_on_peaky_input():
    rng = np.random.default_rng(2)
    P = rng.random((2, 30, 6)) ** 20
    P = (P / P.sum(-1, keepdims=True)).astype(np.float32)
    seqs, _ = kr.ctc_beam_search(P, [28, 28], beam_width=1, merge_repeated=False)
    for b in range(2):
        best = P[b, 2:].argmax(-1)
        col = [int(k) for i, k in enumerate(best) if i == 0 or k != best[i - 1]]
        assert seqs[b] == [k for k in col if k != 5]


def test_adam_maxnorm_oracle():
    rng = np.random.default_rng(4)
    p = rng.standard_normal(10)
    g = rng.standard_normal(10) * 3
    m, v = np.zeros(10), np.zeros(10)
    p0 = p.copy()
    kr.adam_step(p, g, m, v, kr.adam_lr_t(1e-4, 1e-5, 0), clipvalue=0.5)
    gc = np.clip(g, -.5, .5)
    # first Adam step moves every weight by ~lr against the sign of its (clipped) gradient
    assert np.allclose(p0 - p, 1e-4 * np.sign(gc), rtol=1e-3)
    W = rng.standard_normal((6, 4)) * 5
    kr.maxnorm_cols(W, 3.0)
    assert np.all(np.sqrt((W * W).sum(0)) <= 3.0 + 1e-6)
