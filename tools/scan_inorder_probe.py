"""In-order block consumption (mgr_tune 7 = 3 / 4, only with tools/probes/lstm_cluster_inorder.hip.txt built in place of
csrc/lstm_cluster.hip + the dispatch lines quoted in it) vs all-at-once polling (7 = 0) of the K-split scan: alone and as the 4-scan
encoder launch of config F (two workgroups per CU)."""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
import mgr_amd
from mgr_amd import _capi
dev = _capi.Device(0); lib = dev.lib
rng = np.random.default_rng(0)


def mk(Hs, B, T):
    jobs, keep = [], []
    for H in Hs:
        for rev in (0, 1):
            Z = dev.array((rng.standard_normal((B, T, 4 * H)) * 0.5).astype(np.float32))
            Up = dev.array((rng.standard_normal((H, 4 * H)) * 0.05).astype(np.float32))
            Y = dev.zeros((B, T, 2 * H)); keep += [Z, Up, Y]
            jobs.append({"Z": Z, "Up": Up, "Y": Y.view(rev * H, (B, T, H)), "ldy": 2 * H, "B": B, "T": T, "H": H, "reverse": rev})
    return jobs, keep


for name, Hs, B, T in (("H=500 alone", (500,), 64, 1900), ("H=300 alone", (300,), 64, 1900), ("H=100 alone", (100,), 64, 1900),
                       ("F encoders (500+300, 4 scans)", (500, 300), 64, 1900)):
    jobs, keep = mk(Hs, B, T)
    arr = _capi.make_scan_jobs(jobs)
    ws = dev.bytes(lib.mgr_lstm_scan_multi_ws_bytes(len(jobs), arr))
    ref = None
    for t7, rf in ((0, 0), (0, 0)):
        dev.call("mgr_tune", 7, t7); dev.call("mgr_tune", 10, rf); dev.call("mgr_tune", 1, 1)
        _capi.check(lib.mgr_lstm_scan_fwd_multi(dev.ctx, len(jobs), arr, ws.ptr, ws.nbytes)); dev.sync()
        ys = [k.download() for k in keep[2::3]]
        if ref is None:
            ref = ys
        same = all(np.array_equal(a, b) for a, b in zip(ys, ref))
        dev.record(0)
        for _ in range(3):
            _capi.check(lib.mgr_lstm_scan_fwd_multi(dev.ctx, len(jobs), arr, ws.ptr, ws.nbytes))
        dev.record(1); dev.sync()
        ms = dev.elapsed_ms(0, 1) / 3
        print("%-32s tune7=%d refetch=%2d : %7.3f ms  %5.2f us/step  bit-identical=%s" % (name, t7, rf, ms, ms * 1e3 / T, same), flush=True)
    dev.call("mgr_tune", 7, 0); dev.call("mgr_tune", 10, 0); dev.call("mgr_tune", 1, 0)
    for a in keep + [ws]: a.free()
