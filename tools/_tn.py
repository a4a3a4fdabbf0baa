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
    Hs = dev.array(rng.standard_normal((B, T, H)).astype(np.float32))
    dZ = dev.array(rng.standard_normal((B, T, 4 * H)).astype(np.float32))
    m = dev.array(((rng.random((4, B, F)) >= p) / (1 - p)).astype(np.float32))
    gW, gU, gb = dev.empty((F, 4 * H)), dev.empty((H, 4 * H)), dev.empty((4 * H,))
    ws = dev.bytes(dev.lib.mgr_lstm_param_grads_dropout_ws_bytes(B, T, F, H))
    d = kb.timeit(dev, lambda: dev.call("mgr_lstm_param_grads", X, F, m, Hs, H, dZ, gW, gU, gb, B, T, F, H, 0, ws, ws.nbytes))
    s = kb.timeit(dev, lambda: dev.call("mgr_lstm_param_grads_dropout", X, F, m, p, Hs, H, dZ, gW, gU, gb, B, T, F, H, 0, ws, ws.nbytes))
    print("F=%4d H=%3d p=%.1f dense %.3f ms  sparse %.3f ms" % (F, H, p, d, s))
    for a in (X, Hs, dZ, m, gW, gU, gb, ws): a.free()
run(64, 1900, 1600, 100, 0.5)
run(64, 1900, 1600, 100, 0.5)
run(16, 1900, 1000, 500, 0.5)
