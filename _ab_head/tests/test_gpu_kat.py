"""-m gpu: the HIP kernels (through the C ABI) against THIRD-PARTY known-answer vectors - tests/golden/thirdparty_kat.json,
TensorFlow ctc_loss_op_test / ctc_decoder_ops_test and Keras backend_test, i.e. the functions behind K.ctc_batch_cost at
/root/reference/multimodal_fusion/losses.py:13 - and against exhaustive enumeration of all frame paths."""
import ctypes as C
import itertools
import math

import numpy as np
import pytest

from tests.test_cpu_kat import KAT, exhaustive_labelling_probs

pytestmark = pytest.mark.gpu


def _ctc(dev, P, labels, il, ll, skip, eps):
    P = np.ascontiguousarray(P, np.float32)
    B, T, Cn = P.shape
    lab = np.ascontiguousarray(labels, np.int32)
    Lmax = lab.shape[1]
    dP, dl = dev.array(P), dev.array(lab)
    dil, dll = dev.array(np.asarray(il, np.int32)), dev.array(np.asarray(ll, np.int32))
    loss, dz = dev.empty((B,)), dev.empty((B, T, Cn))
    ws = dev.bytes(dev.lib.mgr_ctc_ws_bytes(B, T, Cn, Lmax))
    dev.call("mgr_ctc_loss_grad", dP, dl, dil, dll, B, T, Cn, Lmax, skip, Cn - 1, C.c_float(eps), C.c_float(1.0), loss, dz,
             ws, ws.nbytes)
    return loss.download(), dz.download()


def test_ctc_kernel_reproduces_tensorflow_and_keras_vectors(device):
    """mgr_ctc_loss_grad on the 5 x 6 matrices of TF's testBasic / Keras' test_ctc: losses 3.34211 / 5.42262 (Keras' own
    tolerance 1e-5) and the published gradient w.r.t. the logits (6 published digits)."""
    k = KAT["ctc_loss_testBasic"]
    loss, dz = _ctc(device, k["probs"], k["labels"], k["input_length"], k["label_length"], 0, 1e-8)
    assert np.allclose(loss, k["loss"], rtol=0, atol=k["loss_atol"]), loss
    assert np.abs(dz - np.array(k["grad_wrt_logits"])).max() < k["grad_atol"]
    # the reference's own call shape: two leading frames dropped (losses.py:11), same matrices behind them
    P = np.concatenate([np.full((2, 2, 6), 1 / 6.0), np.array(k["probs"])], axis=1)
    loss2, dz2 = _ctc(device, P, k["labels"], k["input_length"], k["label_length"], 2, 1e-8)
    assert np.allclose(loss2, k["loss"], rtol=0, atol=k["loss_atol"])
    assert np.all(dz2[:, :2] == 0) and np.abs(dz2[:, 2:] - np.array(k["grad_wrt_logits"])).max() < k["grad_atol"]


def test_frame_argmax_reproduces_keras_greedy_vector(device):
    k = KAT["ctc_decode_greedy"]
    P = np.ascontiguousarray(k["probs"], np.float32)
    B, T, Cn = P.shape
    best, prob = device.empty((B, T), np.int32), device.empty((B, T))
    device.call("mgr_frame_argmax", device.array(P), B, T, Cn, 0, best, prob)
    b, p = best.download(), prob.download()
    dec, nlp = [], []
    for i, n in enumerate(k["input_length"]):
        dec.append([int(c) for c, _ in itertools.groupby(b[i, :n]) if c != k["blank"]])
        nlp.append(-float(np.sum(np.log(p[i, :n]))))
    assert dec == k["decoded"]
    assert np.allclose(nlp, [-math.log(1.0 * 0.6 * 0.6 * 0.9), -5 * math.log(0.9)], rtol=1e-6)


def test_beam_kernel_reproduces_tensorflow_beam_vector(device):
    """beam_width 2 returns [1, 0] (the pruned answer TF / Keras publish), a wide beam the truly most probable labelling
    [0, 1, 0] with exactly its exhaustive probability."""
    from mgr_amd.decoding import beam_search_decode
    k = KAT["ctc_decode_beam"]
    P = np.array(k["probs"], np.float32)
    seqs, _ = beam_search_decode(P, k["input_length"], beam_width=k["beam_width"], skip=0, merge_repeated=True, dev=device)
    assert seqs[0] == k["decoded_top_paths"][0]
    seqs, sc = beam_search_decode(P, k["input_length"], beam_width=32, skip=0, merge_repeated=False, dev=device)
    y = (P[0, :5].astype(np.float64) + 1e-8)
    ex = exhaustive_labelling_probs(y / y.sum(-1, keepdims=True), k["blank"])
    assert seqs[0] == [0, 1, 0] and abs(sc[0] - math.log(ex[(0, 1, 0)])) < 1e-9


@pytest.mark.parametrize("T,Cn", [(3, 3), (4, 3), (2, 5), (2, 6), (5, 2)])
def test_beam_kernel_equals_exhaustive_enumeration(device, T, Cn):
    """With at most 32 distinct label prefixes (= the kernel's maximum beam) nothing is ever pruned: the kernel must return
    the most probable labelling and exactly its probability, for 40 random matrices per shape."""
    from mgr_amd.decoding import beam_search_decode
    assert sum((Cn - 1) ** i for i in range(T + 1)) <= 32
    rng = np.random.default_rng(T * 10 + Cn)
    N = 40
    P = rng.random((N, T, Cn)) ** (1 + 3 * rng.random((N, 1, 1)))
    P = (P / P.sum(-1, keepdims=True)).astype(np.float32)
    seqs, sc = beam_search_decode(P, np.full(N, T), beam_width=32, skip=0, merge_repeated=False, dev=device)
    for i in range(N):
        y = P[i].astype(np.float64) + 1e-8
        ex = exhaustive_labelling_probs(y / y.sum(-1, keepdims=True), Cn - 1)
        ranked = sorted(ex.items(), key=lambda kv: -kv[1])
        if len(ranked) > 1 and ranked[0][1] - ranked[1][1] < 1e-9 * ranked[0][1]:
            continue   # an exact tie between labellings: either is a correct answer
        assert tuple(seqs[i]) == ranked[0][0], (i, seqs[i], ranked[:2])
        assert abs(sc[i] - math.log(ranked[0][1])) < 1e-9


# ---- round 3: max-norm, hard_sigmoid, pad_sequences (Keras constraints_test / activations_test / sequence_test) -----------------
def test_maxnorm_kernel_reproduces_keras_explicit_example(device):
    """mgr_maxnorm_cols (kernel_constraint=maxnorm(3), multimodal.py:164) on Keras' 'more explicit example': max 2.0 over the
    columns of [[0,0,0],[1,0,0],[3,0,0],[3,3,3]].T -> [.., [2,0,0], [2/sqrt 3]*3].T, Keras' tolerance rtol 1e-5."""
    from tests.test_cpu_kat import _maxnorm_arrays
    k, x, tgt = _maxnorm_arrays()
    W = device.array(np.ascontiguousarray(x, np.float32))
    device.call("mgr_maxnorm_cols", W, x.shape[0], x.shape[1], C.c_float(k["max_value"]), C.c_float(1e-7))
    np.testing.assert_allclose(W.download(), tgt, rtol=k["rtol"], atol=1e-12)
    # ... and in the layout the optimizer applies it to: a packed (gate-interleaved) [F, 4H] kernel is a column permutation
    Wk = np.tile(x, (1, 3)).astype(np.float32)                        # F = 3, 4H = 12
    Wp, back = device.empty((3, 12)), device.empty((3, 12))
    device.call("mgr_lstm_pack", device.array(Wk), Wp, 3, 3, 0)
    device.call("mgr_maxnorm_cols", Wp, 3, 12, C.c_float(k["max_value"]), C.c_float(1e-7))
    device.call("mgr_lstm_pack", Wp, back, 3, 3, 1)
    np.testing.assert_allclose(back.download(), np.tile(tgt, (1, 3)), rtol=k["rtol"], atol=1e-12)


@pytest.mark.parametrize("H", [4, 100, 300])
def test_scan_gates_reproduce_keras_hard_sigmoid_vector(device, H):
    """recurrent_activation='hard_sigmoid' through mgr_lstm_scan_fwd: one time step from h_0 = c_0 = 0, so the saved gates are
    the activations of the pre-activations Z themselves - i, f, o = hard_sigmoid(z) must give Keras' published values (standard
    values and the two break points), whichever kernel family serves the shape (H = 4 / 100: single-CU; 300: multi-CU cluster)."""
    k = KAT["hard_sigmoid"]
    xs = np.array(k["standard_values"] + k["edge_values"], np.float32)
    want = np.array(k["expected_standard"] + k["expected_edge"], np.float32)
    B, T = len(xs), 1
    Zk = np.zeros((B, T, 4, H), np.float32)                  # Keras gate order i, f, c, o
    for g in (0, 1, 3):
        Zk[:, 0, g, :] = xs[:, None]
    Zk[:, 0, 2, :] = 0.5
    Zp = np.ascontiguousarray(Zk.transpose(0, 1, 3, 2)).reshape(B, T, 4 * H)       # packed: column u * 4 + g
    Up = device.zeros((H, 4 * H))
    Y, G, Cs = device.empty((B, T, H)), device.empty((B, T, H, 4)), device.empty((B, T, H))
    ws = device.bytes(device.lib.mgr_lstm_scan_ws_bytes(B, T, H))
    device.call("mgr_lstm_scan_fwd", device.array(Zp), Up, Y, H, 0, 0, G, Cs, B, T, H, 0, ws, ws.nbytes)
    g = G.download()
    for gi in (0, 1, 3):
        np.testing.assert_allclose(g[:, 0, :, gi], np.repeat(want[:, None], H, 1), rtol=k["rtol"], atol=1e-7)
    assert np.allclose(g[:, 0, :, 2], np.tanh(0.5), atol=2e-7)
    # c_1 = i * tanh(.5), h_1 = o * tanh(c_1): the published activation inside the cell arithmetic
    c1 = want[:, None] * np.tanh(0.5)
    assert np.allclose(Cs.download()[:, 0], c1, atol=3e-7) and np.allclose(Y.download()[:, 0], want[:, None] * np.tanh(c1), atol=3e-7)


def test_pad_sequences_vectors_reach_the_device_unchanged(device):
    """The Keras pad_sequences vectors through the product's generator AND its upload path: what the encoder kernels read
    (Engine._upload_inputs -> device buffers) is the post/post-padded float32 batch."""
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
    g._setup(minibatch_size=3, maxlen=3, nb_classes=6, dataset="val", val_split=0.0, absolute_max_sequence_len=2, store=Store())
    inputs, _ = g.get_batch(train=False)
    want = np.array(k["vector_maxlen3_padding_post"], np.float32)
    d = device.array(np.ascontiguousarray(inputs["x"], np.float32))
    assert np.array_equal(d.download(), want)
