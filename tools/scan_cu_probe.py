"""The CU-owning single-CU scans (lstm_cu.hip, mgr_tune(12, 1)) against the cluster kernels for a narrow layer: outputs
(max abs difference of Y / gates / c, dZ) and time per launch, alone on the chip.  python tools/scan_cu_probe.py [H B T]"""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
import mgr_amd
from mgr_amd import _capi
dev = _capi.Device(0); lib = dev.lib
H, B, T = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (100, 64, 1900)
rng = np.random.default_rng(0)
res = {}
jobs, bjobs, keep = [], [], []
Y = dev.empty((B, T, 2 * H))
dY = dev.array((rng.standard_normal((B, T, 2 * H)) * 0.1).astype(np.float32))
for rev in (0, 1):
    Z = dev.array((rng.standard_normal((B, T, 4 * H)) * 0.5).astype(np.float32))
    Up = dev.array((rng.standard_normal((H, 4 * H)) * 0.05).astype(np.float32))
    G = dev.empty((B, T, H, 4)); Cs = dev.empty((B, T, H)); dZ = dev.empty((B, T, 4 * H))
    keep += [Z, Up, G, Cs, dZ]
    jobs.append({"Z": Z, "Up": Up, "Y": Y.view(rev * H, (1,)), "ldy": 2 * H, "gates": G, "cs": Cs, "B": B, "T": T, "H": H, "reverse": rev})
    bjobs.append({"dY": dY.view(rev * H, (1,)), "gates": G, "cs": Cs, "Up": Up, "dZ": dZ, "lddy": 2 * H, "B": B, "T": T, "H": H, "reverse": rev})
arr = _capi.make_scan_jobs(jobs)
barr = _capi.make_scan_bwd_jobs(bjobs)
ws = dev.bytes(lib.mgr_lstm_scan_multi_ws_bytes(len(jobs), arr))
wsb = dev.bytes(lib.mgr_lstm_scan_bwd_multi_ws_bytes(len(bjobs), barr))
dev.call("mgr_tune", 1, 1)
for mode in (0, 1):
    dev.call("mgr_tune", 12, mode)
    for what, call in (("fwd", lambda: _capi.check(lib.mgr_lstm_scan_fwd_multi(dev.ctx, len(jobs), arr, ws.ptr, ws.nbytes))),
                       ("bwd", lambda: _capi.check(lib.mgr_lstm_scan_bwd_multi(dev.ctx, len(bjobs), barr, wsb.ptr, wsb.nbytes)))):
        call(); dev.sync()
        dev.record(0)
        for _ in range(3):
            call()
        dev.record(1); dev.sync()
        ms = dev.elapsed_ms(0, 1) / 3
        print("H=%d B=%d T=%d single_cu=%d %s: %7.3f ms  %5.2f us/step" % (H, B, T, mode, what, ms, ms * 1e3 / T), flush=True)
    out = {"Y": Y.download()}
    for d in range(2):
        out["G%d" % d] = keep[5 * d + 2].download(); out["C%d" % d] = keep[5 * d + 3].download(); out["dZ%d" % d] = keep[5 * d + 4].download()
    res[mode] = out
if os.environ.get("CU_STAMP"):
    dev.call("mgr_tune", 12, 1); dev.call("mgr_tune", 1, 0)
    _capi.check(lib.mgr_lstm_scan_fwd_multi(dev.ctx, len(jobs), arr, ws.ptr, ws.nbytes)); dev.sync()
    import ctypes
    buf = (ctypes.c_uint32 * 1024)()
    _capi.check(lib.mgr_d2h(dev.ctx, buf, ws.ptr, 4096))
    st = np.array(buf[64:64 + 128], dtype=np.int64).reshape(8, 16)
    base = st[0, 0]
    for r in range(8):
        print("role %d:" % r, " ".join("%6d" % ((v - base) & 0xFFFFFFFF if v else -1) for v in st[r, :10]))
for k in res[0]:
    a, b = res[0][k], res[1][k]
    print("%-4s max|cluster - single_cu| = %.3e  (max |value| %.3e, finite %s)" % (k, np.abs(a - b).max(), np.abs(a).max(), np.isfinite(b).all()))
