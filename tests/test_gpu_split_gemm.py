"""-m gpu: the projection / weight-gradient products on PRE-SPLIT operands (gemm_split.hip, round 5) against numpy fp64 and against
the kernels they replace, through the C ABI."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
f32 = np.float32


def _split_rows(X, ldt):
    """numpy statement of the split row format (mgr.h): XS[b][f] = hi(t) f16 x ldt | lo(t) f16 x ldt of x 2^13, as float32 words."""
    B, T, F = X.shape
    xs = np.zeros((B, F, 2, ldt), np.float16)
    s = (X.transpose(0, 2, 1) * f32(8192.0)).astype(f32)
    hi = s.astype(np.float16)
    lo = (s - hi.astype(f32)).astype(np.float16)
    xs[:, :, 0, :T] = hi
    xs[:, :, 1, :T] = lo
    return xs.reshape(B, F, 2 * ldt).view(f32)


@pytest.mark.parametrize("B,T,F,H,p", [(2, 200, 1000, 132, 0.5), (3, 130, 64, 100, 0.5), (2, 257, 1600, 100, 0.5), (1, 128, 48, 300, 0.6),
                                        (2, 90, 16, 20, 0.9), (2, 140, 600, 300, 0.6), (2, 77, 131, 500, 0.4), (2, 64, 160, 40, 1.0),
                                        (2, 100, 96, 64, 0.0)])
def test_projection_from_split_rows(device, B, T, F, H, p):
    dev = device
    rng = np.random.default_rng(B * 1000 + T + F + H)
    N = 4 * H
    X = rng.uniform(-2, 2, (B, T, F)).astype(f32)
    W = (rng.standard_normal((F, N)) * 0.1).astype(f32)
    bias = rng.standard_normal(N).astype(f32)
    c = f32(1.0 / (1.0 - p)) if p < 1.0 else f32(1.0)
    M = ((rng.random((4, B, F)) >= p) * c).astype(f32)
    if p == 0.0:
        M[:] = 1.0
    ldt = (T + 127) // 128 * 128
    dX, dW, db, dM = dev.array(X), dev.array(W), dev.array(bias), dev.array(M)
    XS = dev.empty((B, F, ldt))
    XS.upload(np.full((B, F, ldt), np.nan, f32))            # the producer must write the padding too
    dev.call("mgr_transpose_bt_split", dX, F, XS, ldt, B, T, F)
    assert np.array_equal(XS.download().view(np.uint32), _split_rows(X, ldt).view(np.uint32))
    ws = dev.bytes(dev.lib.mgr_lstm_input_proj_dropout_ts_ws_bytes(B, F, H))
    dev.call("mgr_memset", ws, 0xFF, ws.nbytes)              # the workspace arrives dirty
    Z = dev.empty((B, T, N))
    gate = np.arange(N) % 4
    ref = np.empty((B, T, N))
    for g in range(4):
        ref[:, :, gate == g] = (X.astype(np.float64) * M[g][:, None, :]) @ W[:, gate == g].astype(np.float64) + bias[gate == g]
    tol = 2e-5 * max(1.0, np.abs(ref).max())
    outs = []
    for tile in (1, 2, 0):        # tune key 12: 128 x 64 tiles (4 waves), 128 x 128 (8 waves), the library's choice
        dev.call("mgr_tune", 12, tile)
        Z.upload(np.full((B, T, N), np.nan, f32))
        dev.call("mgr_lstm_input_proj_dropout_ts", XS, ldt, dM, p if p < 0.99 else 0.5, dW, db, Z, B, T, F, H, ws, ws.nbytes)
        got = Z.download()
        assert np.all(np.isfinite(got)) and np.abs(got - ref).max() <= tol, (tile, np.abs(got - ref).max())
        outs.append(got)
    assert np.array_equal(outs[0], outs[1])      # the same sums in the same order, whatever the tile
    # no mask at all (inference): every feature, factor 1
    dev.call("mgr_lstm_input_proj_dropout_ts", XS, ldt, 0, 0.0, dW, db, Z, B, T, F, H, ws, ws.nbytes)
    ref0 = X.astype(np.float64) @ W.astype(np.float64) + bias
    assert np.abs(Z.download() - ref0).max() <= 2e-5 * max(1.0, np.abs(ref0).max())
    # two different mask factors in one call: not what the kernel was written for - NaN, never a plausible number
    if 0.0 < p < 1.0 and M.max() > 0:
        M2 = M.copy()
        g0, b0, f0 = np.argwhere(M2 > 0)[0]
        M2[g0, b0, f0] *= f32(1.5)
        dev.call("mgr_lstm_input_proj_dropout_ts", XS, ldt, dev.array(M2), p, dW, db, Z, B, T, F, H, ws, ws.nbytes)
        assert np.all(np.isnan(Z.download()))
