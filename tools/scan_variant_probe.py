import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
import mgr_amd
from mgr_amd import _capi
dev = _capi.Device(0); lib = dev.lib
B, T = 64, 1900
rng = np.random.default_rng(0)
for hs in ((500,), (500, 300)):
    jobs, keep = [], []
    for H in hs:
        for rev in (0, 1):
            Z = dev.array((rng.standard_normal((B, T, 4 * H)) * 0.5).astype(np.float32))
            Up = dev.array((rng.standard_normal((H, 4 * H)) * 0.05).astype(np.float32))
            Y = dev.empty((B, T, 2 * H)); keep += [Z, Up, Y]
            jobs.append({"Z": Z, "Up": Up, "Y": Y.view(rev * H, (B, T, H)), "ldy": 2 * H, "B": B, "T": T, "H": H, "reverse": rev})
    arr = _capi.make_scan_jobs(jobs)
    ws = dev.bytes(lib.mgr_lstm_scan_multi_ws_bytes(len(jobs), arr))
    ref = None
    for xl in (0, 2, 1):   # K-split (permuted unit order) | K-split (identity order) | LDS-image step
        dev.call("mgr_tune", 7, xl); dev.call("mgr_tune", 1, 1)
        _capi.check(lib.mgr_lstm_scan_fwd_multi(dev.ctx, len(jobs), arr, ws.ptr, ws.nbytes)); dev.sync()
        dev.record(0)
        for _ in range(3):
            _capi.check(lib.mgr_lstm_scan_fwd_multi(dev.ctx, len(jobs), arr, ws.ptr, ws.nbytes))
        dev.record(1); dev.sync()
        ms = dev.elapsed_ms(0, 1) / 3
        y = keep[2].download()
        if ref is None: ref = y
        print("H=%-10s tune7=%d : %7.3f ms  %5.2f us/step  status=%d  same=%s" % (hs, xl, ms, ms * 1e3 / T, int(ws.download().view(np.uint32)[0]), float(np.abs(y - ref).max())))
    dev.call("mgr_tune", 7, 0)
    for a in keep + [ws]: a.free()
