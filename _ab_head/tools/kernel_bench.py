#!/usr/bin/env python3
"""Micro-benchmarks of individual libmgr kernels at the F-config shapes (device time via HIP events)."""
import argparse
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mgr_amd  # noqa: E402,F401
from mgr_amd import _capi  # noqa: E402


def timeit(dev, fn, reps=5):
    fn()
    dev.sync()
    dev.record(0)
    for _ in range(reps):
        fn()
    dev.record(1)
    return dev.elapsed_ms(0, 1) / reps


def gemm(dev, B, T, F, H, mask=True):
    rng = np.random.default_rng(0)
    X = dev.array(rng.standard_normal((B, T, F)).astype(np.float32))
    Wp = dev.array(rng.standard_normal((F, 4 * H)).astype(np.float32) * 0.05)
    bp = dev.zeros((4 * H,))
    m = dev.array(((rng.random((4, B, F)) > 0.5) * 2.0).astype(np.float32)) if mask else 0
    Z = dev.empty((B, T, 4 * H))
    ms = timeit(dev, lambda: dev.call("mgr_lstm_input_proj", X, F, m, Wp, bp, Z, B, T, F, H))
    fl = 2.0 * B * T * F * 4 * H
    print("gemm_nn  B=%d T=%d F=%4d H=%3d mask=%d : %7.3f ms  %6.1f TF" % (B, T, F, H, mask, ms, fl / ms / 1e9))
    if mask and F >= 16:
        wsd = dev.bytes(dev.lib.mgr_lstm_input_proj_dropout_ws_bytes(B, F, H))
        ms = timeit(dev, lambda: dev.call("mgr_lstm_input_proj_dropout", X, F, m, 0.5, Wp, bp, Z, B, T, F, H, wsd, wsd.nbytes))
        print("  dropout-aware (gathered columns)        : %7.3f ms  %6.1f TF executed" % (ms, 0.5 * fl / ms / 1e9))
        ldt = (T + 127) // 128 * 128
        XT = dev.zeros((B, F, ldt))
        mt = timeit(dev, lambda: dev.call("mgr_transpose_bt", X, F, XT, ldt, B, T, F))
        ms = timeit(dev, lambda: dev.call("mgr_lstm_input_proj_dropout_t", XT, ldt, m, 0.5, Wp, bp, Z, B, T, F, H, wsd, wsd.nbytes, 0.0))
        print("  dropout-aware (transposed copy)         : %7.3f ms  %6.1f TF executed  (+ transpose %.3f ms, %.0f GB/s)"
              % (ms, 0.5 * fl / ms / 1e9, mt, 2.0 * B * T * F * 4 / mt / 1e6))
        ms = timeit(dev, lambda: dev.call("mgr_lstm_input_proj_dropout_t", XT, ldt, m, 0.5, Wp, bp, Z, B, T, F, H, wsd, wsd.nbytes, 8.0))
        print("  split f16, per-gate K loops (kept only) : %7.3f ms  %6.1f TF executed (f32-equivalent)" % (ms, 0.5 * fl / ms / 1e9))
        dev.call("mgr_tune", 10, 2)
        ms = timeit(dev, lambda: dev.call("mgr_lstm_input_proj_dropout_t", XT, ldt, m, 0.5, Wp, bp, Z, B, T, F, H, wsd, wsd.nbytes, 8.0))
        dev.call("mgr_tune", 10, 0)
        print("  split f16, dense K, mask as a factor    : %7.3f ms  %6.1f TF (f32-equivalent, dense)" % (ms, fl / ms / 1e9))
        ms = timeit(dev, lambda: dev.call("mgr_lstm_input_proj_dropout_t", XT, ldt, 0, 0.0, Wp, bp, Z, B, T, F, H, wsd, wsd.nbytes, 8.0))
        print("  split f16, dense K, no mask (inference) : %7.3f ms  %6.1f TF (f32-equivalent)" % (ms, fl / ms / 1e9))
        # pre-split rows + loader / matrix pipeline (gemm_split.hip); the bench mask has the factor 2 = 1 / (1 - 0.5)
        XS = dev.zeros((B, F, ldt))
        Xb = dev.array((np.clip(X.download(), -7.9, 7.9)).astype(np.float32))
        st = timeit(dev, lambda: dev.call("mgr_transpose_bt_split", Xb, F, XS, ldt, B, T, F))
        wss = dev.bytes(dev.lib.mgr_lstm_input_proj_dropout_ts_ws_bytes(B, F, H))
        for tile in (1, 2):
            dev.call("mgr_tune", 12, tile)
            ms = timeit(dev, lambda: dev.call("mgr_lstm_input_proj_dropout_ts", XS, ldt, m, 0.5, Wp, bp, Z, B, T, F, H, wss, wss.nbytes))
            print("  PRE-SPLIT rows, DMA ring, 128 x %3d tile : %7.3f ms  %6.1f TF executed (f32-equivalent)  (+ split transpose %.3f ms)"
                  % (64 * tile, ms, 0.5 * fl / ms / 1e9, st))
            ms = timeit(dev, lambda: dev.call("mgr_lstm_input_proj_dropout_ts", XS, ldt, 0, 0.0, Wp, bp, Z, B, T, F, H, wss, wss.nbytes))
            print("  ... no mask                             : %7.3f ms  %6.1f TF (f32-equivalent)" % (ms, fl / ms / 1e9))
        dev.call("mgr_tune", 12, 0)
        ms = timeit(dev, lambda: dev.call("mgr_lstm_input_proj_dropout_ts", XS, ldt, 0, 0.0, Wp, bp, Z, B, T, F, H, wss, wss.nbytes))
        print("  PRE-SPLIT rows, DMA ring, no mask       : %7.3f ms  %6.1f TF (f32-equivalent)" % (ms, fl / ms / 1e9))
        XS.free(); Xb.free(); wss.free()
        XT.free(); wsd.free()
    dZ = Z
    gW, gU, gb = dev.empty((F, 4 * H)), dev.empty((H, 4 * H)), dev.empty((4 * H,))
    Y = dev.array(rng.standard_normal((B, T, H)).astype(np.float32))
    ws = dev.bytes(dev.lib.mgr_lstm_param_grads_ws_bytes(B, T, F, H))
    ms = timeit(dev, lambda: dev.call("mgr_lstm_param_grads", X, F, m, Y, H, dZ, gW, gU, gb, B, T, F, H, 0, ws, ws.nbytes))
    fl = 2.0 * B * T * (F + H) * 4 * H
    print("gemm_tn  (dW,dU,db)                       : %7.3f ms  %6.1f TF" % (ms, fl / ms / 1e9))
    if mask and F >= 128:
        ldt = (T + 127) // 128 * 128
        XT = dev.zeros((B, F, ldt))
        dev.call("mgr_transpose_bt", X, F, XT, ldt, B, T, F)
        wsd = dev.bytes(dev.lib.mgr_lstm_param_grads_dropout_t_ws_bytes(B, T, F, H, ldt))
        gW2 = dev.empty((F, 4 * H))
        ms = timeit(dev, lambda: dev.call("mgr_lstm_param_grads_dropout", X, F, m, 0.5, Y, H, dZ, gW, gU, gb, B, T, F, H, 0, wsd, wsd.nbytes))
        ms2 = timeit(dev, lambda: dev.call("mgr_lstm_param_grads_dropout_t", XT, ldt, m, 0.5, Y, H, dZ, gW2, gU, gb, B, T, F, H, 0, wsd, wsd.nbytes, 0.0))
        same = np.array_equal(gW.download(), gW2.download())
        ms3 = timeit(dev, lambda: dev.call("mgr_lstm_param_grads_dropout_t", XT, ldt, m, 0.5, Y, H, dZ, gW2, gU, gb, B, T, F, H, 0, wsd, wsd.nbytes, 8.0))
        print("  dropout-aware dW, split f16 operands: %7.3f ms (%5.1f TF f32-equivalent)  max |diff| / max |dW| = %.2e"
              % (ms3, 2.0 * B * T * (0.5 * F + H) * 4 * H / ms3 / 1e9, np.abs(gW.download() - gW2.download()).max() / np.abs(gW.download()).max()))
        fx = 2.0 * B * T * (0.5 * F + H) * 4 * H
        print("  dropout-aware dW: gathered %7.3f ms (%5.1f TF executed) | transposed operands %7.3f ms (%5.1f TF)  bit-identical=%s"
              % (ms, fx / ms / 1e9, ms2, fx / ms2 / 1e9, same))
        XT.free(); wsd.free(); gW2.free()
    for a in (X, Wp, bp, Z, gW, gU, gb, Y, ws):
        a.free()


def scan(dev, B, T, H):
    rng = np.random.default_rng(0)
    Z = dev.array((rng.standard_normal((B, T, 4 * H)) * 0.5).astype(np.float32))
    Up = dev.array((rng.standard_normal((H, 4 * H)) * 0.05).astype(np.float32))
    Y, G, Cs = dev.empty((B, T, H)), dev.empty((B, T, H, 4)), dev.empty((B, T, H))
    ws = dev.bytes(dev.lib.mgr_lstm_scan_ws_bytes(B, T, H))
    ms = timeit(dev, lambda: dev.call("mgr_lstm_scan_fwd", Z, Up, Y, H, 0, 0, G, Cs, B, T, H, 0, ws, ws.nbytes), reps=3)
    print("scan_fwd B=%d T=%d H=%3d : %7.3f ms  %6.2f us/step" % (B, T, H, ms, ms * 1e3 / T))
    if H <= 128:
        dY, dZ = Y, dev.empty((B, T, 4 * H))
        ms = timeit(dev, lambda: dev.call("mgr_lstm_scan_bwd", dY, H, G, Cs, Up, dZ, B, T, H, 0, ws, ws.nbytes), reps=3)
        print("scan_bwd B=%d T=%d H=%3d : %7.3f ms  %6.2f us/step" % (B, T, H, ms, ms * 1e3 / T))


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--what", default="gemm,scan")
    ap.add_argument("--tune", action="append", default=[], help="mgr_tune KEY=VALUE (repeatable)")
    a = ap.parse_args()
    dev = _capi.Device(0)
    for kv in a.tune:
        k, v = kv.split("=")
        dev.call("mgr_tune", int(k), int(v))
    print(dev.name, dev.cu_count, "CUs")
    if a.what == "gemm1":
        gemm(dev, 64, 1900, 1000, 500)
    elif "gemm" in a.what:
        gemm(dev, 64, 1900, 1000, 500)
        gemm(dev, 64, 1900, 1000, 500, mask=False)
        gemm(dev, 64, 1900, 600, 300)
        gemm(dev, 64, 1900, 1600, 100)
        gemm(dev, 64, 1900, 39, 500)
    if a.what == "scan100":
        scan(dev, 64, 1900, 100)
    elif "scan" in a.what:
        scan(dev, 64, 1900, 100)
        scan(dev, 64, 1900, 128)
        scan(dev, 64, 1900, 300)
        scan(dev, 64, 1900, 500)
