"""CPU ORACLE (test infrastructure, NOT product code).

numpy restatement of the arithmetic Keras 2.1.4 / TensorFlow 1.12.1 perform for the
reference's BiLSTM + CTC training path.  Only ``tests/``, ``__graft_entry__.smoke()`` and
``bench.py``'s ``cpu_baseline`` leg may import this module; the product package
(``mgr_amd``) never does and fails loudly when its HIP library is missing.

PARITY UNPINNED: the reference (/root/reference) holds no tests, golden vectors, data or
weights, is Python-2 only, and its arithmetic lives in un-vendored third-party wheels
(Keras==2.1.4, tensorflow==1.12.1, requirements.txt:4,8) that cannot be imported here.
This oracle therefore follows the reference *call sites* plus the published semantics of
those libraries and is cross-checked, in the build container only,
against finite differences and torch-CPU (``tests/golden/make_golden.py``).
Since round 2 the CTC part is additionally held to THIRD-PARTY known-answer vectors
(tests/golden/thirdparty_kat.json: TensorFlow's ctc_loss_op_test.testBasic, Keras'
backend_test.test_ctc / test_ctc_decode_greedy / test_ctc_decode_beam - recalled, self-verifying,
reproduced to the published precision: tests/test_cpu_kat.py) and to exhaustive path
enumeration.  The LSTM cell, the optimizer and the network glue remain pinned by torch
autograd / finite differences only: no vector of the reference itself exists, so the
label "parity unpinned" stays.

Every function cites the reference file:line whose behaviour it restates.
All functions take a ``dtype`` (np.float64 for the fp64 oracle, np.float32 for the
"Keras CPU" timing leg) and are vectorised over the batch with a Python loop over time,
which is the op structure Keras' ``K.rnn``/``tf.while_loop`` executes.
"""
from __future__ import annotations

import itertools
import math

import numpy as np

NEG_INF = -np.inf


# ----------------------------------------------------------------------------------------
# activations  (Keras LSTM args fixed by multimodal_fusion/multimodal.py:159-168:
# activation='tanh', recurrent_activation='hard_sigmoid')
# ----------------------------------------------------------------------------------------
def hard_sigmoid(z):
    """TF-backend hard_sigmoid: clip(0.2*z + 0.5, 0, 1)."""
    return np.clip(0.2 * z + 0.5, 0.0, 1.0)


def hard_sigmoid_grad_from_out(a):
    """d hs/dz expressed on the activation value: 0.2 strictly inside (0,1), else 0.

    (clip_by_value passes the gradient at the exact bounds z = +-2.5; that measure-zero
    case is treated as saturated here and in the HIP kernels - documented in DESIGN.md.)
    """
    return np.where((a > 0.0) & (a < 1.0), 0.2, 0.0)


# ----------------------------------------------------------------------------------------
# LSTM  (Keras LSTMCell implementation=1; call sites multimodal_fusion/multimodal.py:109-118,
# 159-168; audio_network/speech_lstm_ctc_words.py:56-77; skeletal_network/skeletal_lstm_ctc.py:309-335)
# weights: W (F,4H), U (H,4H), b (4H); gate column blocks i,f,c,o
# ----------------------------------------------------------------------------------------
def lstm_forward(x, W, U, b, mask4=None, reverse=False):
    """One direction of a Keras LSTM with return_sequences=True.

    x: (B,T,F).  mask4: (4,B,F) input-dropout masks (entries 0 or 1/(1-p)), constant over
    time, one per gate (Keras ``dropout=p``) or None.
    reverse=True is ``go_backwards`` + the ``K.reverse`` Bidirectional applies, i.e. the
    output at index t is the state after consuming frames T-1..t.
    Returns y (B,T,H) and a cache for lstm_backward.
    """
    dt = x.dtype
    B, T, F = x.shape
    H = U.shape[0]
    h = np.zeros((B, H), dt)
    c = np.zeros((B, H), dt)
    y = np.empty((B, T, H), dt)
    gi = np.empty((B, T, H), dt)
    gf = np.empty((B, T, H), dt)
    gg = np.empty((B, T, H), dt)
    go = np.empty((B, T, H), dt)
    cs = np.empty((B, T, H), dt)
    Wg = [W[:, k * H:(k + 1) * H] for k in range(4)]
    Ug = [U[:, k * H:(k + 1) * H] for k in range(4)]
    bg = [b[k * H:(k + 1) * H] for k in range(4)]
    order = range(T - 1, -1, -1) if reverse else range(T)
    for t in order:
        xt = x[:, t, :]
        if mask4 is None:
            xs = [xt, xt, xt, xt]
        else:
            xs = [xt * mask4[k] for k in range(4)]
        zi = xs[0] @ Wg[0] + bg[0] + h @ Ug[0]
        zf = xs[1] @ Wg[1] + bg[1] + h @ Ug[1]
        zc = xs[2] @ Wg[2] + bg[2] + h @ Ug[2]
        zo = xs[3] @ Wg[3] + bg[3] + h @ Ug[3]
        i = hard_sigmoid(zi)
        f = hard_sigmoid(zf)
        g = np.tanh(zc)
        o = hard_sigmoid(zo)
        c = f * c + i * g
        h = o * np.tanh(c)
        y[:, t, :] = h
        gi[:, t, :] = i
        gf[:, t, :] = f
        gg[:, t, :] = g
        go[:, t, :] = o
        cs[:, t, :] = c
    cache = dict(x=x, W=W, U=U, mask4=mask4, reverse=reverse, y=y, i=gi, f=gf, g=gg, o=go, c=cs)
    return y, cache


def lstm_backward(dy, cache, need_dx=True):
    """BPTT for lstm_forward. dy: (B,T,H). Returns dx (or None), dW, dU, db."""
    x, W, U, mask4, reverse = cache["x"], cache["W"], cache["U"], cache["mask4"], cache["reverse"]
    y, gi, gf, gg, go, cs = cache["y"], cache["i"], cache["f"], cache["g"], cache["o"], cache["c"]
    dt = x.dtype
    B, T, F = x.shape
    H = U.shape[0]
    dW = np.zeros_like(W)
    dU = np.zeros_like(U)
    db = np.zeros(4 * H, dt)
    dx = np.zeros_like(x) if need_dx else None
    dh_rec = np.zeros((B, H), dt)
    dc_carry = np.zeros((B, H), dt)
    # time order of the forward recursion; walk it backwards
    order = list(range(T - 1, -1, -1)) if reverse else list(range(T))
    for n in range(T - 1, -1, -1):
        t = order[n]
        tp = order[n - 1] if n > 0 else None  # previous step in recursion order
        i, f, g, o, c = gi[:, t], gf[:, t], gg[:, t], go[:, t], cs[:, t]
        c_prev = cs[:, tp] if tp is not None else np.zeros((B, H), dt)
        h_prev = y[:, tp] if tp is not None else np.zeros((B, H), dt)
        tc = np.tanh(c)
        dh = dy[:, t] + dh_rec
        do = dh * tc
        dc = dh * o * (1.0 - tc * tc) + dc_carry
        di = dc * g
        df = dc * c_prev
        dg = dc * i
        dc_carry = dc * f
        dzi = di * hard_sigmoid_grad_from_out(i)
        dzf = df * hard_sigmoid_grad_from_out(f)
        dzc = dg * (1.0 - g * g)
        dzo = do * hard_sigmoid_grad_from_out(o)
        dz = np.concatenate([dzi, dzf, dzc, dzo], axis=1)  # (B,4H)
        dh_rec = dz @ U.T
        dU += h_prev.T @ dz
        db += dz.sum(axis=0)
        xt = x[:, t, :]
        for k, dzk in enumerate((dzi, dzf, dzc, dzo)):
            xk = xt if mask4 is None else xt * mask4[k]
            dW[:, k * H:(k + 1) * H] += xk.T @ dzk
            if need_dx:
                dxk = dzk @ W[:, k * H:(k + 1) * H].T
                dx[:, t, :] += dxk if mask4 is None else dxk * mask4[k]
    return dx, dW, dU, db


def bilstm_forward(x, wf, wb, maskf=None, maskb=None):
    """Bidirectional(LSTM, merge_mode='concat').

    wf, wb: (W,U,b) of the forward / backward sub-layer (Keras weight-list order).
    Returns y (B,T,2H), cache.
    """
    yf, cf = lstm_forward(x, *wf, mask4=maskf, reverse=False)
    yb, cb = lstm_forward(x, *wb, mask4=maskb, reverse=True)
    return np.concatenate([yf, yb], axis=2), (cf, cb)


def bilstm_backward(dy, cache, need_dx=True):
    cf, cb = cache
    H = cf["U"].shape[0]
    dxf, dWf, dUf, dbf = lstm_backward(dy[:, :, :H], cf, need_dx)
    dxb, dWb, dUb, dbb = lstm_backward(dy[:, :, H:], cb, need_dx)
    dx = (dxf + dxb) if need_dx else None
    return dx, (dWf, dUf, dbf), (dWb, dUb, dbb)


# ----------------------------------------------------------------------------------------
# Dropout / Dense / softmax   (multimodal_fusion/multimodal.py:171-179)
# ----------------------------------------------------------------------------------------
def dense_softmax_forward(a, dmask, Wd, bd):
    """P = softmax((a * dmask) @ Wd + bd) over the last axis. dmask (B,T,D) or None."""
    ad = a if dmask is None else a * dmask
    z = ad @ Wd + bd
    z = z - z.max(axis=-1, keepdims=True)
    e = np.exp(z)
    P = e / e.sum(axis=-1, keepdims=True)
    return P, dict(ad=ad, dmask=dmask, Wd=Wd)


def dense_backward(dlogits, cache):
    ad, dmask, Wd = cache["ad"], cache["dmask"], cache["Wd"]
    D = ad.shape[-1]
    C = Wd.shape[1]
    dWd = ad.reshape(-1, D).T @ dlogits.reshape(-1, C)
    dbd = dlogits.reshape(-1, C).sum(axis=0)
    da = dlogits @ Wd.T
    if dmask is not None:
        da = da * dmask
    return da, dWd, dbd


# ----------------------------------------------------------------------------------------
# CTC  (multimodal_fusion/losses.py:4-15 -> K.ctc_batch_cost -> tf.nn.ctc_loss; App. A.5)
# ----------------------------------------------------------------------------------------
def _lse2(a, b):
    m = np.maximum(a, b)
    with np.errstate(invalid="ignore", divide="ignore"):
        r = m + np.log(np.exp(a - m) + np.exp(b - m))
    return np.where(np.isneginf(m), NEG_INF, r)


def _lse3(a, b, c):
    m = np.maximum(np.maximum(a, b), c)
    with np.errstate(invalid="ignore", divide="ignore"):
        r = m + np.log(np.exp(a - m) + np.exp(b - m) + np.exp(c - m))
    return np.where(np.isneginf(m), NEG_INF, r)


def ctc_loss_grad(P, labels, input_length, label_length, skip=2, blank=None, eps=1e-8,
                  need_grad=True):
    """ctc_lambda_func (losses.py:11-13): y_pred[:, skip:, :] then K.ctc_batch_cost.

    P: (B,T,C) softmax output of the network.  labels: (B,Lmax) padded with -1 (any
    numeric dtype; cast to int like ctc_label_dense_to_sparse does).  input_length /
    label_length: (B,) or (B,1).  blank defaults to C-1 (TF convention).
    Returns loss (B,) and, if need_grad, the gradient of sum_b loss_b w.r.t. the
    pre-softmax Dense logits z (P = softmax(z)), shape (B,T,C), zero on the `skip`
    dropped frames and on frames >= skip+input_length.
    """
    dt = P.dtype
    B, T, C = P.shape
    if blank is None:
        blank = C - 1
    labels = np.asarray(labels).astype(np.int64)
    input_length = np.asarray(input_length).reshape(B).astype(np.int64)
    label_length = np.asarray(label_length).reshape(B).astype(np.int64)
    loss = np.zeros(B, dt)
    dz = np.zeros((B, T, C), dt) if need_grad else None
    for bidx in range(B):
        Tp = int(input_length[bidx])
        L = int(label_length[bidx])
        lab = labels[bidx, :L]
        Pp = P[bidx, skip:skip + Tp, :]
        # Keras: log(y_pred + eps); TF ctc_loss: softmax over it
        u = Pp + dt.type(eps)
        ynorm = u / u.sum(axis=1, keepdims=True)
        logy = np.log(ynorm)
        S = 2 * L + 1
        lp = np.full(S, blank, np.int64)
        lp[1::2] = lab
        can_skip = np.zeros(S, bool)
        if S > 2:
            can_skip[2:] = (lp[2:] != blank) & (lp[2:] != lp[:-2])
        em = logy[:, lp]  # (Tp,S)
        alpha = np.full((Tp, S), NEG_INF, dt)
        alpha[0, 0] = em[0, 0]
        if S > 1:
            alpha[0, 1] = em[0, 1]
        for t in range(1, Tp):
            a = alpha[t - 1]
            a1 = np.concatenate(([NEG_INF], a[:-1]))
            a2 = np.concatenate(([NEG_INF, NEG_INF], a[:-2]))[:S]     # (an empty label sequence has the single state "blank")
            a2 = np.where(can_skip, a2, NEG_INF)
            alpha[t] = em[t] + _lse3(a, a1, a2)
        if S > 1:
            logp = _lse2(alpha[Tp - 1, S - 1], alpha[Tp - 1, S - 2])
        else:
            logp = alpha[Tp - 1, S - 1]
        loss[bidx] = -logp
        if not need_grad:
            continue
        beta = np.full((Tp, S), NEG_INF, dt)
        beta[Tp - 1, S - 1] = 0.0
        if S > 1:
            beta[Tp - 1, S - 2] = 0.0
        # skip INTO state u+2 is allowed iff can_skip[u+2]
        for t in range(Tp - 2, -1, -1):
            nb = beta[t + 1] + em[t + 1]
            n1 = np.concatenate((nb[1:], [NEG_INF]))
            n2 = np.concatenate((nb[2:], [NEG_INF, NEG_INF]))[:S]
            cs2 = np.concatenate((can_skip[2:], [False, False]))[:S]
            n2 = np.where(cs2, n2, NEG_INF)
            beta[t] = _lse3(nb, n1, n2)
        ab = alpha + beta  # (Tp,S) log prob of all paths through (t,u)
        occ = np.zeros((Tp, C), dt)
        with np.errstate(invalid="ignore"):
            w = np.exp(ab - logp)
        w = np.where(np.isfinite(w), w, 0.0)
        np.add.at(occ.T, lp, w.T)
        du = ynorm - occ  # d loss / d u  where y = softmax(u), u = log(P+eps)
        # chain: u = log(P+eps) -> g_P = du/(P+eps); P = softmax(z) -> g_z = P*(g_P - sum P g_P)
        gP = du / u
        gz = Pp * (gP - (Pp * gP).sum(axis=1, keepdims=True))
        dz[bidx, skip:skip + Tp, :] = gz
    return loss, dz


# ----------------------------------------------------------------------------------------
# optimizer  (Adam(lr, clipvalue=.5, decay) + maxnorm(3); multimodal_fusion/multimodal.py:159-168,206-213; App. A.6)
# ----------------------------------------------------------------------------------------
def adam_lr_t(lr, decay, k, b1=0.9, b2=0.999):
    """Keras Adam effective step size for iteration index k (0-based count of prior updates)."""
    lr_k = lr * (1.0 / (1.0 + decay * k))
    t = k + 1
    return lr_k * math.sqrt(1.0 - b2 ** t) / (1.0 - b1 ** t)


def adam_step(p, g, m, v, lr_t, b1=0.9, b2=0.999, eps=1e-7, clipvalue=0.5, gscale=1.0):
    """In-place Keras-Adam update with elementwise clip (clipvalue<=0 disables)."""
    g = g * p.dtype.type(gscale)
    if clipvalue and clipvalue > 0:
        g = np.clip(g, -clipvalue, clipvalue)
    m[...] = b1 * m + (1.0 - b1) * g
    v[...] = b2 * v + (1.0 - b2) * g * g
    p[...] = p - lr_t * m / (np.sqrt(v) + eps)


def maxnorm_cols(W, maxv=3.0, eps=1e-7):
    """keras.constraints.maxnorm(3) with axis=0: per column of an LSTM input kernel."""
    n = np.sqrt((W * W).sum(axis=0, keepdims=True))
    W[...] = W * (np.clip(n, 0, maxv) / (eps + n))


# ----------------------------------------------------------------------------------------
# decode  (multimodal_fusion/sequence_decoding.py:21-69, audio_network/sequence_decoding.py:19-69; App. C)
# ----------------------------------------------------------------------------------------
def greedy_decode_quirk(out, thr=0.5, skip=2):
    """Literal Python-2 behaviour of decode_batch's filter loop (sequence_decoding.py:41-50).

    Python 2's zip() materialises a list snapshot, and list.remove deletes the FIRST
    element equal to the value - NOT the element being visited.  Returns, per sample,
    the label-id list after the filter and the groupby collapse (blanks kept).
    """
    res = []
    for j in range(out.shape[0]):
        out_prob = list(np.max(out[j, skip:], 1))
        out_best = list(np.argmax(out[j, skip:], 1))
        for p, s in list(zip(out_prob, out_best)):  # snapshot, like Py2 zip
            if p < thr:
                out_prob.remove(p)
                out_best.remove(s)
        res.append([int(k) for k, _ in itertools.groupby(out_best)])
    return res


def _lse64(a, b):
    """fp64 log-sum-exp with -inf as log-zero; same formula as the HIP beam kernel."""
    if a == NEG_INF:
        return b
    if b == NEG_INF:
        return a
    m = a if a > b else b
    return m + math.log1p(math.exp(-abs(a - b)))


def ctc_beam_search(P, input_length, beam_width=10, skip=2, blank=None, eps=1e-8, merge_repeated=True, top_paths=1):
    """CTC prefix beam search without LM (spec: K.ctc_decode(greedy=False, beam_width=10,
    top_paths=1) -> tf.nn.ctc_beam_search_decoder; BASELINE.json config 5).
    NOT present in the reference - this oracle IS the specification the HIP path matches.

    Arithmetic is fp64 (fp32 scores near -5000 have ulp 5e-4, which would make the beam
    cut depend on libm rounding).  Per frame: logy = log(P+eps) - log(sum(P+eps)).
    Beams are a ranked list; candidates are enumerated as  idx = r*(C+1) + slot  with
    slot 0 = "stay on prefix r" (blank, or repeat of its last label) and slot 1+c =
    "extend prefix r by label c".  An extension that reproduces the prefix of another
    live beam r2 is merged into r2's stay candidate (stay term first).  Candidates are
    ranked by lse(p_blank, p_nonblank) descending, ties by smaller idx; -inf dropped.
    Returns (label lists of the best path per sample, their log-probabilities); with top_paths > 1 every sample's entry
    is the ranked list of its top_paths best beams instead (tf.nn.ctc_beam_search_decoder's top_paths).

    Pinned to third-party vectors (tests/golden/thirdparty_kat.json): the decoded sequences of TensorFlow's
    ctc_decoder_ops_test.testCTCDecoderBeamSearch / Keras' backend_test.test_ctc_decode_beam (beam_width 2: the narrow beam
    returns [1, 0] ahead of the truly most probable labelling [0, 1, 0]), and - independent of any library - to the
    exhaustive enumeration of all C^T paths on tiny cases (tests/test_cpu_kat.py).
    """
    B, T, C = P.shape
    if blank is None:
        blank = C - 1
    input_length = np.asarray(input_length).reshape(B).astype(np.int64)
    outs, scores = [], []
    for bidx in range(B):
        Tp = int(input_length[bidx])
        beams = [((), 0.0, NEG_INF)]  # ranked: (prefix, log p_blank, log p_nonblank)
        for t in range(Tp):
            u = P[bidx, skip + t].astype(np.float64) + eps
            with np.errstate(divide="ignore"):      # (eps = 0 with exact zeros in P: log 0 = -inf is the intended log-zero)
                logy = np.log(u) - math.log(float(u.sum()))
            index = {pref: r for r, (pref, _, _) in enumerate(beams)}
            cand = {}  # idx -> [prefix, pb, pnb]
            for r, (pref, pb, pnb) in enumerate(beams):
                tot = _lse64(pb, pnb)
                stay_nb = pnb + logy[pref[-1]] if pref else NEG_INF
                cand[r * (C + 1)] = [pref, tot + logy[blank], stay_nb]
            for r, (pref, pb, pnb) in enumerate(beams):
                tot = _lse64(pb, pnb)
                for c in range(C):
                    if c == blank:
                        continue
                    val = (pb if (pref and c == pref[-1]) else tot) + logy[c]
                    npref = pref + (c,)
                    r2 = index.get(npref)
                    if r2 is not None:
                        e = cand[r2 * (C + 1)]
                        e[2] = _lse64(e[2], val)
                    else:
                        cand[r * (C + 1) + 1 + c] = [npref, NEG_INF, val]
            ranked = sorted(((-_lse64(e[1], e[2]), idx) for idx, e in cand.items()
                             if _lse64(e[1], e[2]) != NEG_INF))
            beams = [tuple(cand[idx]) for _, idx in ranked[:beam_width]]
        seqs, scs = [], []
        for pref, pb, pnb in beams[:max(1, top_paths)]:
            seq = list(pref)
            if merge_repeated:
                seq = [k for k, _ in itertools.groupby(seq)]
            seqs.append(seq)
            scs.append(_lse64(pb, pnb))
        outs.append(seqs[0] if top_paths == 1 else seqs)
        scores.append(scs[0] if top_paths == 1 else scs)
    return outs, scores


def edit_distance(a, b):
    """Levenshtein distance between two label lists (for label-error-rate)."""
    la, lb = len(a), len(b)
    d = list(range(lb + 1))
    for i in range(1, la + 1):
        prev, d[0] = d[0], i
        for j in range(1, lb + 1):
            cur = d[j]
            d[j] = min(d[j] + 1, d[j - 1] + 1, prev + (a[i - 1] != b[j - 1]))
            prev = cur
    return d[lb]
