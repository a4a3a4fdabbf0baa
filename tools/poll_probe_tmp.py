import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
import mgr_amd
from mgr_amd import _capi
dev = _capi.Device(0); lib = dev.lib
T = 1900
rng = np.random.default_rng(0)
def run(tag, shapes, B):
    jobs, keep = [], []
    for H in shapes:
        for rev in (0, 1):
            Z = dev.array((rng.standard_normal((B, T, 4 * H)) * 0.5).astype(np.float32))
            Up = dev.array((rng.standard_normal((H, 4 * H)) * 0.05).astype(np.float32))
            Y = dev.empty((B, T, 2 * H)); keep += [Z, Up, Y]
            jobs.append({"Z": Z, "Up": Up, "Y": Y.view(rev * H, (B, T, H)), "ldy": 2 * H, "B": B, "T": T, "H": H, "reverse": rev})
    arr = _capi.make_scan_jobs(jobs)
    ws = dev.bytes(lib.mgr_lstm_scan_multi_ws_bytes(len(jobs), arr))
    for ps in (0, 100):
        dev.call("mgr_tune", 13, 0); dev.call("mgr_tune", 12, 1 if ps == 100 else 0); dev.call("mgr_tune", 1, 1)
        _capi.check(lib.mgr_lstm_scan_fwd_multi(dev.ctx, len(jobs), arr, ws.ptr, ws.nbytes)); dev.sync()
        dev.record(0)
        for _ in range(3):
            _capi.check(lib.mgr_lstm_scan_fwd_multi(dev.ctx, len(jobs), arr, ws.ptr, ws.nbytes))
        dev.record(1); dev.sync()
        ms = dev.elapsed_ms(0, 1) / 3
        hdr = ws.download().view(np.uint32)[768:768 + 250]
        fol = (hdr >> 31) == 1; rounds = (hdr & 0xFFFFF) / (T - 1.0); miss = (hdr >> 20) & 0x7FF
        for nm, m in (("leaders/uncoupled", ~fol), ("followers", fol)):
            if m.any(): print("   %s: %d wgs, rounds per step mean %.2f max %.2f, gate polls per step max %d" % (nm, m.sum(), rounds[m].mean(), rounds[m].max(), miss[m].max()))
        print("%-28s poll_sleep=%d : %7.3f ms  %5.2f us/step" % (tag, ps, ms, ms * 1e3 / T), flush=True)
    dev.call("mgr_tune", 13, 0); dev.call("mgr_tune", 12, 0); dev.call("mgr_tune", 1, 0)
    for a in keep + [ws]: a.free()
run("audio+skeletal B=64 (408)", (500, 300), 64)
run("audio B=128 (512 WGs)", (500,), 128)
