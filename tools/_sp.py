import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import kernel_bench as kb
from mgr_amd import _capi
dev = _capi.Device(0)
def run(B, T, F, H, p):
    rng = np.random.default_rng(0)
    X = dev.array(rng.standard_normal((B, T, F)).astype(np.float32))
    Wp = dev.array(rng.standard_normal((F, 4 * H)).astype(np.float32) * 0.05)
    bp = dev.zeros((4 * H,))
    m = dev.array(((rng.random((4, B, F)) >= p) / (1 - p)).astype(np.float32))
    Z = dev.empty((B, T, 4 * H))
    ws = dev.bytes(dev.lib.mgr_lstm_input_proj_dropout_ws_bytes(B, F, H))
    d = kb.timeit(dev, lambda: dev.call("mgr_lstm_input_proj", X, F, m, Wp, bp, Z, B, T, F, H))
    s = kb.timeit(dev, lambda: dev.call("mgr_lstm_input_proj_dropout", X, F, m, p, Wp, bp, Z, B, T, F, H, ws, ws.nbytes))
    fl = 2.0 * B * T * F * 4 * H
    print("F=%4d H=%3d p=%.1f dense %.3f ms (%.1f TF)  sparse %.3f ms (%.1f TF algorithmic)" % (F, H, p, d, fl / d / 1e9, s, fl / s / 1e9))
    for a in (X, Wp, bp, m, Z, ws): a.free()
for mode in (0,):
    dev.call("mgr_tune", 11, mode)
    print("mode", mode)
    run(64, 1900, 1000, 500, 0.5)
    run(64, 1900, 600, 300, 0.6)
    run(64, 1900, 1600, 100, 0.5)
