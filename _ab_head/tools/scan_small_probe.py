"""XCD-local exchange vs write-through exchange of the K-split scan for SMALL clusters (H = 100 / 128: G = 7 / 8 workgroups)."""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
import mgr_amd
from mgr_amd import _capi
dev = _capi.Device(0); lib = dev.lib
rng = np.random.default_rng(0)
for (H, B, T) in ((128, 32, 1000), (100, 64, 1900), (128, 64, 1900), (300, 32, 1000), (500, 32, 1900)):
    jobs, keep = [], []
    for rev in (0, 1):
        Z = dev.array((rng.standard_normal((B, T, 4 * H)) * 0.5).astype(np.float32))
        Up = dev.array((rng.standard_normal((H, 4 * H)) * 0.05).astype(np.float32))
        Y = dev.empty((B, T, 2 * H)); keep += [Z, Up, Y]
        jobs.append({"Z": Z, "Up": Up, "Y": Y.view(rev * H, (B, T, H)), "ldy": 2 * H, "B": B, "T": T, "H": H, "reverse": rev})
    arr = _capi.make_scan_jobs(jobs)
    ws = dev.bytes(lib.mgr_lstm_scan_multi_ws_bytes(len(jobs), arr))
    for t3 in (0, 1):
        dev.call("mgr_tune", 3, t3); dev.call("mgr_tune", 1, 1); dev.call("mgr_tune", 2, 1 if t3 == 0 else 0)
        _capi.check(lib.mgr_lstm_scan_fwd_multi(dev.ctx, len(jobs), arr, ws.ptr, ws.nbytes)); dev.sync()
        dev.call("mgr_tune", 2, 0)
        dev.record(0)
        for _ in range(3):
            _capi.check(lib.mgr_lstm_scan_fwd_multi(dev.ctx, len(jobs), arr, ws.ptr, ws.nbytes))
        dev.record(1); dev.sync()
        ms = dev.elapsed_ms(0, 1) / 3
        print("H=%d B=%d T=%d xcd_local=%d : %7.3f ms  %5.2f us/step" % (H, B, T, 1 - t3, ms, ms * 1e3 / T), flush=True)
    dev.call("mgr_tune", 3, 0); dev.call("mgr_tune", 1, 0)
    for a in keep + [ws]: a.free()
