"""Debug: mgr_lstm_param_grads_dropout_ts repeated on the same inputs must give the same bits every time."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mgr_amd
import numpy as np
from mgr_amd import _capi
dev = _capi.Device(0)
f32 = np.float32
for (B, T, F, H) in ((16, 96, 1600, 100), (64, 1900, 1600, 100)):
    rng = np.random.default_rng(1)
    N = 4 * H
    X = rng.uniform(-2, 2, (B, T, F)).astype(f32)
    Hs = rng.standard_normal((B, T, H)).astype(f32)
    dZ = (rng.standard_normal((B, T, N)) * 0.3).astype(f32)
    M = ((rng.random((4, B, F)) >= 0.5) * f32(2.0)).astype(f32)
    ldt = (T + 127) // 128 * 128
    dX, dH, ddZ, dM = dev.array(X), dev.array(Hs), dev.array(dZ), dev.array(M)
    XS = dev.zeros((B, F, ldt))
    dev.call("mgr_transpose_bt_split", dX, F, XS, ldt, B, T, F)
    ws = dev.bytes(dev.lib.mgr_lstm_param_grads_dropout_ts_ws_bytes(B, T, F, H, ldt))
    zmx = dev.array(np.abs(dZ).max(axis=1).astype(f32).view(np.uint32))
    for tile in (1, 0):
        dev.call("mgr_tune", 12, tile)
        ref = None
        bad = 0
        for rep in range(40):
            gW, gU, gb = dev.empty((F, N)), dev.empty((H, N)), dev.empty((N,))
            dev.call("mgr_memset", ws, 0xFF if rep % 2 else 0x00, ws.nbytes)
            dev.call("mgr_lstm_param_grads_dropout_ts", XS, ldt, dM, 0.5, dH, H, ddZ, gW, gU, gb, B, T, F, H, 0, ws, ws.nbytes, zmx if rep % 3 else 0, 0, 0, 0)
            g = gW.download()
            if ref is None:
                ref = g
            elif not np.array_equal(g, ref, equal_nan=True) or np.isnan(g).any():
                bad += 1
                idx = np.argwhere((g != ref) | np.isnan(g))
                print("   rep", rep, "n diff", len(idx), "nan", int(np.isnan(g).sum()), "rows", sorted(set(idx[:, 0]))[:6], "cols", sorted(set(idx[:, 1]))[:6])
            for a in (gW, gU, gb):
                a.free()
        print("B=%d T=%d tile=%d: %d of 39 repeats differ" % (B, T, tile, bad))
    dev.call("mgr_tune", 12, 0)
