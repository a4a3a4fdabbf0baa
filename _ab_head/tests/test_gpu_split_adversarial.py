"""-m gpu: adversarial inputs for the split-f16 arithmetic - the default path forms every f32 product of the recurrences,
the wide projections and the dropout-aware dW as three f16 matrix products of (hi, lo)-split operands.  Each case runs the SAME
C-ABI call three ways - split-f16 (default), f32 MFMA (no bound stated / tune key 14 = 1) and numpy fp64 - and requires, element by
element,

        |split - fp64|  <=  2 x (largest |f32-MFMA - fp64| of the tensor)  +  2^-22 x sum_i |a_i b_i|

i.e. the split path may be no worse than the f32 path plus the representation error its design promises (22 bits per product).
Cases (Next 1c): heavy-tailed trained-like weights (max / median 1e4, a few at 50), cancellation-heavy dot products
(sum ~ 0 of large terms), activations exactly at +-x_absmax, and recurrent states whose hi half has an odd last mantissa bit
under both epoch parities (the exchange steals that bit, lstm_cluster.hip).

The recurrence's own constant is 2^-20, not 2^-22, and the test says why: the h operand carries one flag bit per published word,
which costs it up to one bit (|h s - hi - lo| <= 2^-21 |h s| instead of 2^-22; U adds 2^-22, the dropped lo lo term 2^-22).  This
file is what measured it: round 4's form (bit 0 of BOTH halves of the even unit forced to the parity) reached 8.4e-7 = 2^-20.2
relative on a single product here and 2^-18.8 in the worst case; since round 5 the value moves to the NEAREST f16 with the flag
bit, and the two flag bits of a unit pair sit on different units.
"""
import numpy as np
import pytest

from oracle import keras_ref as kr

pytestmark = pytest.mark.gpu
f32 = np.float32
EPS22 = 2.0 ** -22


def _heavy_tailed(rng, shape, median=5e-3, top=50.0, n_top=6):
    """|w| log-normal around `median` with a tail; a few entries at +-top: max / median = 1e4."""
    w = median * np.exp(rng.standard_normal(shape) * 1.5) * rng.choice([-1.0, 1.0], size=shape)
    w = np.clip(w, -top / 4, top / 4)
    flat = w.reshape(-1)
    idx = rng.choice(flat.size, size=n_top, replace=False)
    flat[idx] = top * rng.choice([-1.0, 1.0], size=n_top)
    return w.astype(f32)


def _check(name, split, mfma, ref, absdot):
    e_s, e_m = np.abs(split.astype(np.float64) - ref), np.abs(mfma.astype(np.float64) - ref)
    assert np.all(np.isfinite(split)), name
    bound = 2.0 * e_m.max() + EPS22 * absdot
    worst = (e_s / np.maximum(bound, 1e-300)).max()
    print("   %-28s split %.3e  f32-mfma %.3e  worst / bound %.3f" % (name, e_s.max(), e_m.max(), worst))
    assert np.all(e_s <= bound), (name, float(e_s.max()), float(e_m.max()), float(worst))


def _proj_inputs(kind, rng, B, T, F, N, bound):
    X = (rng.uniform(-1, 1, (B, T, F)) * bound).astype(f32)
    W = (rng.standard_normal((F, N)) * 0.1).astype(f32)
    if kind == "heavy_tailed_weights":
        W = _heavy_tailed(rng, (F, N))
    elif kind == "cancellation":
        # features come in pairs (f, f + F/2) with the same activation and opposite weights up to the last bits: every dot product
        # is a sum ~ 0 of terms ~ 1
        h = F // 2
        X[:, :, h:2 * h] = X[:, :, :h]
        W = (rng.uniform(0.5, 2.0, (F, N)) * rng.choice([-1.0, 1.0], (F, N))).astype(f32)
        W[h:2 * h] = -W[:h] * (1.0 + rng.integers(-2, 3, (h, N)) * 2.0 ** -23).astype(f32)
    elif kind == "at_the_bound":
        X = (rng.choice([-1.0, 1.0], (B, T, F)) * bound).astype(f32)        # every activation exactly +-x_absmax
    return X, W


@pytest.mark.parametrize("kind", ["heavy_tailed_weights", "cancellation", "at_the_bound"])
@pytest.mark.parametrize("B,T,F,H,p", [(2, 200, 1000, 132, 0.5), (2, 130, 600, 75, 0.6)])
def test_projection_split_vs_f32_vs_fp64(device, kind, B, T, F, H, p):
    """mgr_lstm_input_proj_dropout_t with a bound on |X| (k_gemm_nn_sparse16; tune key 10 = 2: k_gemm_nn_dense16; no mask: the
    inference projection) against the same call without a bound (the f32 MFMA kernel) and fp64."""
    dev = device
    rng = np.random.default_rng(sum(map(ord, kind)) + F)
    N, bound = 4 * H, 2.0
    X, W = _proj_inputs(kind, rng, B, T, F, N, bound)
    if kind == "cancellation":
        M = np.full((4, B, F), 1.0 / (1.0 - p), f32)
        h = F // 2
        keep = rng.random((4, B, h)) >= p
        M[:, :, :h] *= keep
        M[:, :, h:2 * h] *= keep          # a pair is kept or dropped together, so that the cancellation survives the mask
    else:
        M = ((rng.random((4, B, F)) >= p) / (1.0 - p)).astype(f32)
    bias = rng.standard_normal(N).astype(f32)
    ldt = (T + 127) // 128 * 128
    dX, dW, db, dM = dev.array(X), dev.array(W), dev.array(bias), dev.array(M)
    XT = dev.zeros((B, F, ldt))
    dev.call("mgr_transpose_bt", dX, F, XT, ldt, B, T, F)
    ws = dev.bytes(dev.lib.mgr_lstm_input_proj_dropout_ws_bytes(B, F, H))
    gate = np.arange(N) % 4
    X64, W64 = X.astype(np.float64), W.astype(np.float64)
    ref, absdot = np.empty((B, T, N)), np.empty((B, T, N))
    for g in range(4):
        ref[:, :, gate == g] = (X64 * M[g][:, None, :]) @ W64[:, gate == g] + bias[gate == g]
        absdot[:, :, gate == g] = (np.abs(X64) * M[g][:, None, :]) @ np.abs(W64[:, gate == g])

    def run(mask, drop, x_absmax, kernel=0):
        out = dev.empty((B, T, N))
        out.upload(np.full((B, T, N), np.nan, f32))
        dev.call("mgr_tune", 10, kernel)
        try:
            dev.call("mgr_lstm_input_proj_dropout_t", XT, ldt, mask, drop, dW, db, out, B, T, F, H, ws, ws.nbytes, x_absmax)
        finally:
            dev.call("mgr_tune", 10, 0)
        return out.download()

    mfma = run(dM, p, 0.0)
    _check(kind + " / kept-feature loops", run(dM, p, bound), mfma, ref, absdot)
    _check(kind + " / dense K loop", run(dM, p, bound, kernel=2), mfma, ref, absdot)
    # inference: no mask.  The f32 leg is the plain projection kernel.
    ref0 = X64 @ W64 + bias
    plain = dev.empty((B, T, N))
    dev.call("mgr_lstm_input_proj", dX, F, 0, dW, db, plain, B, T, F, H)
    _check(kind + " / no mask", run(0, 0.0, bound), plain.download(), ref0, np.abs(X64) @ np.abs(W64))


@pytest.mark.parametrize("kind", ["heavy_tailed_gradients", "cancellation", "at_the_bound"])
def test_weight_gradient_split_vs_f32_vs_fp64(device, kind):
    """mgr_lstm_param_grads_dropout_t (k_gemm_tn_sparse16: K = time, dZ scaled per (sample, gate column)) the same three ways."""
    dev = device
    B, T, F, H, p, bound = 2, 288, 600, 100, 0.5, 2.0
    rng = np.random.default_rng(len(kind))
    N = 4 * H
    X = (rng.uniform(-1, 1, (B, T, F)) * bound).astype(f32)
    dZ = (rng.standard_normal((B, T, N)) * 0.3).astype(f32)
    if kind == "heavy_tailed_gradients":
        dZ = _heavy_tailed(rng, (B, T, N), median=1e-6, top=1e-2, n_top=40)      # a trained net's gate gradients: tiny, with spikes
    elif kind == "cancellation":
        h = T // 2                                                               # time steps come in cancelling pairs
        X[:, h:2 * h] = X[:, :h]
        dZ[:, h:2 * h] = -dZ[:, :h] * (1.0 + rng.integers(-2, 3, (B, h, N)) * 2.0 ** -23).astype(f32)
    elif kind == "at_the_bound":
        X = (rng.choice([-1.0, 1.0], (B, T, F)) * bound).astype(f32)
    Hs = rng.uniform(-1, 1, (B, T, H)).astype(f32)
    M = ((rng.random((4, B, F)) >= p) / (1.0 - p)).astype(f32)
    ldt = (T + 127) // 128 * 128
    dX, dH, ddZ, dM = dev.array(X), dev.array(Hs), dev.array(dZ), dev.array(M)
    XT = dev.zeros((B, F, ldt))
    dev.call("mgr_transpose_bt", dX, F, XT, ldt, B, T, F)
    ws = dev.bytes(dev.lib.mgr_lstm_param_grads_dropout_t_ws_bytes(B, T, F, H, ldt))
    gate = np.arange(N) % 4
    X64, Z64 = X.astype(np.float64), dZ.astype(np.float64)
    ref, absdot = np.empty((F, N)), np.empty((F, N))
    for g in range(4):
        ref[:, gate == g] = np.einsum("btf,btn->fn", X64 * M[g][:, None, :], Z64[:, :, gate == g])
        absdot[:, gate == g] = np.einsum("btf,btn->fn", np.abs(X64) * M[g][:, None, :], np.abs(Z64[:, :, gate == g]))

    def run(x_absmax):
        gW, gU, gb = dev.empty((F, N)), dev.empty((H, N)), dev.empty((N,))
        gW.upload(np.full((F, N), np.nan, f32))
        dev.call("mgr_memset", ws, 0xFF, ws.nbytes)
        dev.call("mgr_lstm_param_grads_dropout_t", XT, ldt, dM, p, dH, H, ddZ, gW, gU, gb, B, T, F, H, 0, ws, ws.nbytes, x_absmax)
        return gW.download()

    _check(kind, run(bound), run(0.0), ref, absdot)


def _act(z):
    """activated gates (i, f, g, o) of packed pre-activations z[..., unit, gate] (Keras: hard_sigmoid, hard_sigmoid, tanh, hard_sigmoid)"""
    hs = lambda v: np.clip(0.2 * v + 0.5, 0.0, 1.0)
    return np.stack([hs(z[..., 0]), hs(z[..., 1]), np.tanh(z[..., 2]), hs(z[..., 3])], axis=-1)


@pytest.mark.parametrize("H,kind", [(100, "heavy_tailed"), (300, "heavy_tailed"), (500, "cancellation"), (300, "single_unit")])
def test_recurrence_one_step_operand_fidelity(device, H, kind):
    """The recurrent product h_{t-1} U of the multi-CU scan, ONE STEP AT A TIME: with the kernel's own outputs Y_{t-1} as the
    operand, fp64 recomputes the gates of step t and compares them with the gates the kernel saved - for the split-f16 step (where
    h travels as an f16 (hi, lo) pair whose hi loses its last mantissa bit to the epoch parity on every second unit) and for the f32
    MFMA step.  Time steps 1 .. T-1 cover both epoch parities ((t >> 1) & 1); about half of all h values have an odd last hi bit
    (counted and asserted), i.e. the stolen bit is exercised under both parities.  Isolating single steps keeps the chaotic
    amplification of a long recurrence out of the bound."""
    dev = device
    rng = np.random.default_rng(H + len(kind))
    B, T = 20, 12
    N = 4 * H
    U = rng.uniform(-0.4, 0.4, (H, N)).astype(f32) * f32(0.1 if H >= 300 else 1.0)
    if kind == "heavy_tailed":
        U = _heavy_tailed(rng, (H, N), median=4e-4, top=4.0, n_top=8)
    elif kind == "cancellation":
        h2 = H // 2                        # units in pairs with opposite recurrent rows: sums ~ 0 of terms ~ 1 once the states agree
        U = (rng.uniform(0.5, 1.5, (H, N)) * rng.choice([-1.0, 1.0], (H, N))).astype(f32)
        U[h2:2 * h2] = -U[:h2]
        U[:, :] *= f32(0.05)
    elif kind == "single_unit":
        U[:] = 0                           # every gate of step t hangs on ONE even-lane unit's h (the lane that carries the parity)
        U[2] = rng.uniform(-3, 3, N).astype(f32)
    Zk = rng.standard_normal((B, T, H, 4)).astype(f32)     # packed: [unit][gate]
    if kind == "cancellation":
        Zk[:, :, H // 2:2 * (H // 2)] = Zk[:, :, :H // 2]  # identical pre-activations -> identical states in a pair
    Ud = dev.array(U)
    Up = dev.empty((H, N))
    dev.call("mgr_lstm_pack", Ud, Up, H, H, 0)
    Z = dev.array(Zk.reshape(B, T, N))
    ws = dev.bytes(dev.lib.mgr_lstm_scan_ws_bytes(B, T, H))
    # Keras layout of U: column g * H + u
    U64 = U.astype(np.float64).reshape(H, 4, H).transpose(0, 2, 1)        # [k][unit][gate]
    out = {}
    dev.call("mgr_tune", 0, 3)     # clusters with an exchange at every H (the K-split step)
    dev.call("mgr_tune", 1, 1)
    try:
        for f32_mfma in (0, 1):
            dev.call("mgr_tune", 14, f32_mfma)
            Y, G, Cs = dev.zeros((B, T, H)), dev.zeros((B, T, H, 4)), dev.zeros((B, T, H))
            dev.call("mgr_lstm_scan_fwd", Z, Up, Y, H, 0, 0, G, Cs, B, T, H, 0, ws, ws.nbytes)
            out[f32_mfma] = (Y.download().astype(np.float64), G.download().astype(np.float64))
    finally:
        dev.call("mgr_tune", 14, 0)
        dev.call("mgr_tune", 0, 0)
        dev.call("mgr_tune", 1, 0)
    errs = {}
    for f32_mfma, (Y, G) in out.items():
        z = Zk[:, 1:].astype(np.float64) + np.einsum("btk,kug->btug", Y[:, :-1], U64)
        errs[f32_mfma] = np.abs(G[:, 1:] - _act(z))
    absdot = np.einsum("btk,kug->btug", np.abs(out[0][0][:, :-1]), np.abs(U64))
    # how many of the h values the split step exchanged had an odd last bit in their f16 hi half, per epoch parity of the step
    hi = (out[0][0][:, :-1] * 32768.0).astype(np.float16).view(np.uint16) & 1
    for par in (0, 1):
        steps = [t for t in range(T - 1) if ((t >> 1) & 1) == par]
        frac = float(hi[:, steps][:, :, 0::2].mean())
        assert 0.2 < frac < 0.8, (par, frac)           # both kinds of last bit meet both parities on the parity-carrying units
    e_s, e_m = errs[0], errs[1]
    bound = 2.0 * e_m.max() + 2.0 ** -20 * absdot + 2.0 ** -23  # (+ half an ulp of a gate value in [0.5, 1]: G is stored as f32)
    worst = (e_s / bound).max()
    print("   H=%d %-14s split %.3e  f32-mfma %.3e  worst / bound %.3f" % (H, kind, e_s.max(), e_m.max(), worst))
    assert np.all(e_s <= bound), (float(e_s.max()), float(e_m.max()), float(worst))
