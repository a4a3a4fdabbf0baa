"""Synthetic ChaLearn-shaped data and weights (SURVEY.md 8(d) recipe): there is no dataset or checkpoint to load.

Shapes and dtypes follow DataGenerator.get_batch (reference multimodal_fusion/data_generator.py:157-278):
float64 arrays, post-padded with zeros, labels padded with -1, input_length = maxlen - 2.
"""
import numpy as np


def synthetic_arrays(spec, n, T, Lmax, seed, lmin=8, lmax=20):
    """n sequences: dict stream-name -> (n,T,F) float64, labels (n,Lmax) float64 (-1 padded), label_length (n,1)."""
    rng = np.random.default_rng(seed)
    C = spec.num_classes
    lmax = max(1, min(lmax, Lmax, (T - 2) // 2))
    lmin = min(lmin, lmax)
    length = rng.integers(int(np.ceil(0.6 * T)), T + 1, size=n)
    xs = {}
    for s in spec.streams:
        scale = 3.0 if "audio" in s["name"] or s["F"] == 39 else 1.0   # MFCCs are not normalised on this path
        x = rng.standard_normal((n, T, s["F"])) * scale
        for i in range(n):
            x[i, length[i]:, :] = 0.0
        xs[s["name"]] = x
    labels = -np.ones((n, Lmax))
    label_length = np.zeros((n, 1), np.int64)
    for i in range(n):
        L = int(rng.integers(lmin, lmax + 1))
        seq = rng.integers(1, C - 1, size=L)
        seq = np.where(rng.random(L) < 0.05, 0, seq)
        labels[i, :L] = seq
        label_length[i, 0] = L
    input_length = np.full((n, 1), T - 2, np.int64)
    return xs, labels, input_length, label_length


def synthetic_weights(spec, seed):
    """kernels U(-.05,.05), recurrent orthogonal per gate, bias 0 with forget block 1, Dense U(-.05,.05)."""
    rng = np.random.default_rng(seed)
    w = {}
    for name, shape, _, kind in spec.weight_table():
        if kind in ("kernel", "dense"):
            w[name] = rng.uniform(-0.05, 0.05, size=shape).astype(np.float32)
        elif kind == "recurrent":
            H = shape[0]
            blocks = []
            for _ in range(4):
                q, r = np.linalg.qr(rng.standard_normal((H, H)))
                blocks.append(q * np.sign(np.diag(r)))
            w[name] = np.concatenate(blocks, axis=1).astype(np.float32)
        elif kind == "bias":
            H = shape[0] // 4
            b = np.zeros(shape, np.float32)
            b[H:2 * H] = 1.0
            w[name] = b
        else:
            w[name] = np.zeros(shape, np.float32)
    return w
