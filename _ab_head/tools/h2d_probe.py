#!/usr/bin/env python3
"""Does a small host->device copy on an idle stream wait for work queued on OTHER streams?  (pageable + sync vs pinned async)"""
import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mgr_amd  # noqa
from mgr_amd import _capi
dev = _capi.Device(0); lib = dev.lib
B, T, H = 64, 1900, 500
rng = np.random.default_rng(0)
Z = dev.array((rng.standard_normal((B, T, 4 * H)) * 0.5).astype(np.float32))
Up = dev.array((rng.standard_normal((H, 4 * H)) * 0.05).astype(np.float32))
Y = dev.empty((B, T, H))
ws = dev.bytes(lib.mgr_lstm_scan_ws_bytes(B, T, H))
small = dev.empty((2240,), np.int32)
host = np.arange(2240, dtype=np.int32)
pin = C.c_void_p(); _capi.check(lib.mgr_host_alloc(dev.ctx, 8960, C.byref(pin)))
C.memmove(pin, host.ctypes.data, 8960)
def busy():
    dev.stream(1)
    for _ in range(3):
        dev.call("mgr_lstm_scan_fwd", Z, Up, Y, H, 0, 0, 0, 0, B, T, H, 0, ws, ws.nbytes)   # ~8 ms each
for mode in ("pageable+sync", "pinned async", "pageable+sync (device idle)"):
    dev.sync()
    if "idle" not in mode:
        busy()
    dev.stream(7)
    t0 = time.perf_counter()
    if mode.startswith("pageable"):
        small.upload(host)
    else:
        _capi.check(lib.mgr_h2d_async(dev.ctx, small.ptr, pin, 8960))
    t1 = time.perf_counter()
    dev.sync()
    print("%-28s host blocked %.2f ms (device busy ~%.0f ms)" % (mode, (t1 - t0) * 1e3, (time.perf_counter() - t0) * 1e3))
