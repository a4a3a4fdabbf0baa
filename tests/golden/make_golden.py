"""Generate the committed golden fixtures under tests/golden/ (run in the BUILD container).

The reference cannot run anywhere (Python 2 + Keras 2.1.4 + TF 1.12.1, none present), so
the fixtures are produced by the fp64 oracle (oracle/) and - where an independent
implementation exists in this container - cross-checked here against torch-CPU autograd
(``torch.nn.functional.ctc_loss`` and a hand-written Keras-semantics LSTM) before being
written.  torch is used ONLY by this generator; neither the oracle nor the product imports it.

    python tests/golden/make_golden.py

Fixtures are data only: seeded inputs, weights, injected randomness and expected outputs.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle import keras_ref as kr  # noqa: E402
from oracle import network_ref as nr  # noqa: E402


def torch_network_loss_grads(spec, w, inputs, labels, input_length, label_length, rand):
    """Independent torch-autograd implementation of the same graph (fp64)."""
    import torch
    import torch.nn.functional as F

    tw = {k: torch.tensor(v, dtype=torch.float64, requires_grad=True) for k, v in w.items()}
    tr = {k: torch.tensor(v, dtype=torch.float64) for k, v in (rand or {}).items()}

    def hs(z):
        return torch.clamp(0.2 * z + 0.5, 0.0, 1.0)

    def lstm(x, W, U, b, mask4, reverse):
        B, T, _ = x.shape
        H = U.shape[0]
        h = torch.zeros(B, H, dtype=torch.float64)
        c = torch.zeros(B, H, dtype=torch.float64)
        ys = [None] * T
        order = range(T - 1, -1, -1) if reverse else range(T)
        for t in order:
            xt = x[:, t]
            zs = []
            for k in range(4):
                xk = xt if mask4 is None else xt * mask4[k]
                zs.append(xk @ W[:, k * H:(k + 1) * H] + b[k * H:(k + 1) * H] + h @ U[:, k * H:(k + 1) * H])
            i, f, g, o = hs(zs[0]), hs(zs[1]), torch.tanh(zs[2]), hs(zs[3])
            c = f * c + i * g
            h = o * torch.tanh(c)
            ys[t] = h
        return torch.stack(ys, dim=1)

    def bilstm(x, p):
        yf = lstm(x, tw[p + "/fwd/W"], tw[p + "/fwd/U"], tw[p + "/fwd/b"], tr.get(p + "/fwd/mask"), False)
        yb = lstm(x, tw[p + "/bwd/W"], tw[p + "/bwd/U"], tw[p + "/bwd/b"], tr.get(p + "/bwd/mask"), True)
        return torch.cat([yf, yb], dim=2)

    outs = []
    for s in spec["streams"]:
        x = torch.tensor(inputs[s["name"]], dtype=torch.float64)
        if s["name"] + "/noise" in tr:
            x = x + tr[s["name"] + "/noise"]
        ys = []
        cur = x
        for k in range(len(s["layers"])):
            cur = bilstm(cur, "%s/l%d" % (s["name"], k))
            ys.append(cur)
        outs.append(ys[0] + ys[1] if (s.get("residual") and len(ys) == 2) else ys[-1])
    feat = torch.cat(outs, dim=2)
    if spec.get("fusion"):
        feat = bilstm(feat, "fusion")
    if "head/mask" in tr:
        feat = feat * tr["head/mask"]
    z = feat @ tw["dense/W"] + tw["dense/b"]
    P = torch.softmax(z, dim=-1)
    skip = spec["ctc"]["skip"]
    eps = spec["ctc"]["eps"]
    logp = F.log_softmax(torch.log(P[:, skip:, :] + eps), dim=-1).permute(1, 0, 2)
    C = P.shape[-1]
    B = P.shape[0]
    il = torch.tensor(np.asarray(input_length).reshape(B), dtype=torch.long)
    ll = torch.tensor(np.asarray(label_length).reshape(B), dtype=torch.long)
    tg = torch.cat([torch.tensor(np.asarray(labels[b][:int(ll[b])]), dtype=torch.long) for b in range(B)])
    loss_b = F.ctc_loss(logp, tg, il, ll, blank=C - 1, reduction="none", zero_infinity=False)
    loss = loss_b.mean()
    loss.backward()
    grads = {k: (v.grad.numpy() if v.grad is not None else None) for k, v in tw.items()}
    return float(loss), loss_b.detach().numpy(), grads, P.detach().numpy()


def tiny_fusion_spec():
    return {
        "streams": [
            {"name": "audio", "F": 5, "noise": 0.5,
             "layers": [{"H": 8, "dropout": 0.4}, {"H": 8, "dropout": 0.5}], "residual": True, "trainable": False},
            {"name": "skeletal", "F": 3, "noise": 0.0,
             "layers": [{"H": 4, "dropout": 0.6}, {"H": 4, "dropout": 0.6}], "residual": True, "trainable": False},
        ],
        "fusion": {"H": 4, "dropout": 0.5, "maxnorm": 3.0},
        "head": {"dropout": 0.5, "C": 6},
        "ctc": {"skip": 2, "eps": 1e-8},
        "optimizer": {"lr": 1e-4, "decay": 1e-5, "clipvalue": 0.5, "beta_1": 0.9, "beta_2": 0.999,
                      "epsilon": 1e-7, "maxnorm": 3.0},
    }


def tiny_unimodal_spec():
    return {
        "streams": [
            {"name": "the_input", "F": 5, "noise": 0.5,
             "layers": [{"H": 8, "dropout": 0.4}, {"H": 8, "dropout": 0.5}], "residual": True, "trainable": True},
        ],
        "fusion": None,
        "head": {"dropout": 0.5, "C": 7},
        "ctc": {"skip": 2, "eps": 1e-8},
        "optimizer": {"lr": 1e-4, "decay": 0.0, "clipvalue": 0.5, "beta_1": 0.9, "beta_2": 0.999,
                      "epsilon": 1e-7, "maxnorm": 3.0},
    }


def pack(d, prefix):
    return {prefix + k.replace("/", "__"): v for k, v in d.items() if v is not None}


def make_network_case(name, spec, B, T, Lmax, seed, steps):
    rng = np.random.default_rng(seed)
    w = nr.init_weights(spec, rng)
    # make weights larger than the init recipe so gates leave the linear region
    for k in w:
        if k.endswith("/W") or k.endswith("/U"):
            w[k] = w[k] * 4.0
    inputs, labels, il, ll = nr.synthetic_batch(spec, B, T, Lmax, rng, lmin=1, lmax=3)
    rand = nr.draw_rand(spec, B, T, rng)
    # (1) torch cross-check on ordinary label rows
    loss, loss_b, grads, P = nr.loss_and_grads(spec, w, inputs, labels, il, ll, rand)
    tloss, tloss_b, tgrads, tP = torch_network_loss_grads(spec, w, inputs, labels, il, ll, rand)
    assert abs(loss - tloss) < 1e-9 * max(1, abs(tloss)), (loss, tloss)
    assert np.allclose(P, tP, rtol=1e-10, atol=1e-12)
    assert np.allclose(loss_b, tloss_b, rtol=1e-9)
    for k, g in grads.items():
        assert np.allclose(g, tgrads[k], rtol=1e-7, atol=1e-10), (k, np.abs(g - tgrads[k]).max())
    # (2) the fixture itself carries one sample with the "empty label -> [blank]" substitution of
    # data_generator.py:228-238.  torch's CTC backward is inconsistent with its own loss when a
    # target equals the blank index, so that row is validated by central finite differences.
    labels[-1, :] = -1
    labels[-1, 0] = spec["head"]["C"] - 1
    ll[-1, 0] = 1
    loss, loss_b, grads, P = nr.loss_and_grads(spec, w, inputs, labels, il, ll, rand)
    for k in ("dense/b", "fusion/fwd/b" if spec.get("fusion") else "the_input/l0/fwd/b"):
        for idx in range(min(6, w[k].size)):
            wp = {n: v.copy() for n, v in w.items()}
            wm = {n: v.copy() for n, v in w.items()}
            wp[k].flat[idx] += 1e-6
            wm[k].flat[idx] -= 1e-6
            fd = (nr.loss_and_grads(spec, wp, inputs, labels, il, ll, rand)[0]
                  - nr.loss_and_grads(spec, wm, inputs, labels, il, ll, rand)[0]) / 2e-6
            assert abs(fd - grads[k].flat[idx]) < 1e-6 * max(1.0, abs(fd)), (k, idx, fd, grads[k].flat[idx])
    # multi-step trajectory with the oracle trainer (fresh randomness per step)
    w0 = {k: v.copy() for k, v in w.items()}
    tr = nr.Trainer(spec, {k: v.copy() for k, v in w.items()})
    traj = []
    rands = []
    for s in range(steps):
        r = nr.draw_rand(spec, B, T, rng)
        rands.append(r)
        traj.append(tr.train_on_batch(inputs, labels, il, ll, r))
    out = {}
    out.update(pack(w0, "w__"))
    out.update(pack(inputs, "x__"))
    out.update(pack(rand, "r__"))
    for s, r in enumerate(rands):
        out.update(pack(r, "rs%d__" % s))
    out.update(pack(grads, "g__"))
    out.update(pack(tr.w, "wfinal__"))
    out["labels"] = labels
    out["input_length"] = il
    out["label_length"] = ll
    out["loss"] = np.float64(loss)
    out["loss_b"] = loss_b
    out["P"] = P
    out["traj"] = np.array(traj)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    import json
    with open(os.path.join(HERE, name + ".json"), "w") as f:
        json.dump({"spec": spec, "B": B, "T": T, "Lmax": Lmax, "seed": seed, "steps": steps}, f, indent=1)
    print(name, "loss", loss, "torch", tloss, "traj", traj)


def make_ctc_case():
    """Stand-alone CTC vectors incl. repeated labels, blank-as-label, L=1, long label rows."""
    import torch
    import torch.nn.functional as F
    rng = np.random.default_rng(7)
    B, T, C, Lmax = 6, 40, 8, 12
    z = rng.standard_normal((B, T, C)) * 2.0
    P = np.exp(z - z.max(-1, keepdims=True))
    P /= P.sum(-1, keepdims=True)
    labels = -np.ones((B, Lmax))
    rows = [[1, 1, 2], [3], [0, 5, 5, 5, 1], [7], [2, 4, 6, 0, 1, 3, 5, 2, 4, 6, 0, 1], [6, 6]]
    ll = np.zeros((B, 1), np.int64)
    for b, r in enumerate(rows):
        labels[b, :len(r)] = r
        ll[b, 0] = len(r)
    il = np.array([[38], [38], [30], [38], [38], [5]], np.int64)
    loss, dz = kr.ctc_loss_grad(P, labels, il, ll)
    tz = torch.tensor(z, dtype=torch.float64, requires_grad=True)
    tP = torch.softmax(tz, -1)
    tot = 0
    tl = []
    for b in range(B):
        Tp = int(il[b, 0])
        lp = F.log_softmax(torch.log(tP[b:b + 1, 2:2 + Tp] + 1e-8), -1).permute(1, 0, 2)
        l = F.ctc_loss(lp, torch.tensor([rows[b]]), torch.tensor([Tp]), torch.tensor([len(rows[b])]),
                       blank=C - 1, reduction="sum")
        tl.append(float(l.detach()))
        tot = tot + l
    tot.backward()
    assert np.allclose(loss, tl, rtol=1e-10), (loss, tl)
    tg = tz.grad.numpy()
    for b in range(B):
        if C - 1 in rows[b]:
            # torch's backward is inconsistent with its own loss for target == blank: use finite differences
            for idx in [(5, 0), (9, C - 1), (20, 3), (37, C - 1)]:
                zp, zm = z.copy(), z.copy()
                zp[(b,) + idx] += 1e-6
                zm[(b,) + idx] -= 1e-6

                def lo(zz):
                    Pz = np.exp(zz - zz.max(-1, keepdims=True))
                    Pz /= Pz.sum(-1, keepdims=True)
                    return kr.ctc_loss_grad(Pz, labels, il, ll, need_grad=False)[0][b]
                fd = (lo(zp) - lo(zm)) / 2e-6
                assert abs(fd - dz[(b,) + idx]) < 1e-6, (b, idx, fd, dz[(b,) + idx])
        else:
            assert np.allclose(dz[b], tg[b], rtol=1e-7, atol=1e-11), (b, np.abs(dz[b] - tg[b]).max())
    np.savez_compressed(os.path.join(HERE, "ctc_small.npz"), P=P, labels=labels, input_length=il,
                        label_length=ll, loss=loss, dlogits=dz)
    print("ctc_small loss", loss)


def make_decode_case():
    rng = np.random.default_rng(11)
    N, T, C = 5, 60, 22
    z = rng.standard_normal((N, T, C)) * 3.0
    # make it peaky and run-structured so that the filter and collapse both matter
    for n in range(N):
        t = 0
        while t < T:
            run = int(rng.integers(1, 7))
            c = int(rng.integers(0, C))
            z[n, t:t + run, c] += rng.uniform(0.0, 6.0)
            t += run
    P = np.exp(z - z.max(-1, keepdims=True))
    P = (P / P.sum(-1, keepdims=True)).astype(np.float32)
    f_list = np.array([17, 228, 301, 375, 402])  # two of them are on the reference's ignore list
    dec05 = kr.greedy_decode_quirk(P, 0.5)
    dec075 = kr.greedy_decode_quirk(P, 0.75)
    beam, score = kr.ctc_beam_search(P, np.full(N, T - 2), beam_width=10)
    mx = max(len(d) for d in dec05 + dec075 + beam)

    def padded(lst):
        a = -np.ones((N, mx), np.int64)
        for i, d in enumerate(lst):
            a[i, :len(d)] = d
        return a

    np.savez_compressed(os.path.join(HERE, "decode_small.npz"), P=P, f_list=f_list, greedy_thr05=padded(dec05),
                        greedy_thr075=padded(dec075), beam10=padded(beam), beam10_score=np.array(score))
    print("decode_small", [len(d) for d in dec05], [len(d) for d in beam])


if __name__ == "__main__":
    make_ctc_case()
    make_decode_case()
    make_network_case("fusion_tiny", tiny_fusion_spec(), B=3, T=12, Lmax=4, seed=20131900, steps=4)
    make_network_case("unimodal_tiny", tiny_unimodal_spec(), B=3, T=12, Lmax=4, seed=20131901, steps=4)
