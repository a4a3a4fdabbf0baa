#!/usr/bin/env python3
"""Diagnostic (never timed): -DMGR_STAMP build, ONE multi-scan launch of the 4 encoder layer-directions of config F;
prints the per-step cycle split of the first audio cluster's waves while the skeletal clusters share the CUs."""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["MGR_CXXFLAGS"] = "-DMGR_STAMP"
pkg = os.path.join(ROOT, "multimodal-gesture-recognition-with-lstms-and-ctc_amd")
subprocess.check_call([sys.executable, os.path.join(pkg, "_build.py"), "--force"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
import mgr_amd  # noqa: E402,F401
from mgr_amd import _capi  # noqa: E402

dev = _capi.Device(0)
lib = dev.lib
B, T = 64, 1900
rng = np.random.default_rng(0)
for hs in ((500, 300), (500,), (300,)):
    jobs, keep = [], []
    for H in hs:
        for rev in (0, 1):
            Z = dev.array((rng.standard_normal((B, T, 4 * H)) * 0.5).astype(np.float32))
            Up = dev.array((rng.standard_normal((H, 4 * H)) * 0.05).astype(np.float32))
            Y = dev.empty((B, T, 2 * H))
            keep += [Z, Up, Y]
            jobs.append({"Z": Z, "Up": Up, "Y": Y.view(rev * H, (B, T, H)), "ldy": 2 * H, "B": B, "T": T, "H": H, "reverse": rev})
    arr = _capi.make_scan_jobs(jobs)
    ws = dev.bytes(lib.mgr_lstm_scan_multi_ws_bytes(len(jobs), arr))
    dev.call("mgr_tune", 2, 1)
    _capi.check(lib.mgr_lstm_scan_fwd_multi(dev.ctx, len(jobs), arr, ws.ptr, ws.nbytes))
    dev.call("mgr_tune", 2, 0)
    dev.sync()
    dev.record(0)
    _capi.check(lib.mgr_lstm_scan_fwd_multi(dev.ctx, len(jobs), arr, ws.ptr, ws.nbytes))
    dev.record(1)
    ms = dev.elapsed_ms(0, 1)
    raw = ws.download().view(np.uint64)
    dbg = raw[8:8 + 16 * 8].reshape(16, 8)
    print("jobs H=%s: %.3f ms (%.2f us/step, stamped build) status=%d" % (hs, ms, ms * 1e3 / T, raw[0] & 0xFFFFFFFF))
    print("   wave      mfma  compute  gather  barrier  passes/step   (cycles per step; K-split kernel: mfma, cell+stores, poll wait, reduce+barrier, rounds)")
    for w in range(16):
        m = dbg[w]
        if m[1] == 0:
            continue
        print("   bg%d.w%d %7.0f %7.0f %7.0f %7.0f   %.2f   total %7.0f pre %6.0f" % (w // 8, w % 8, m[0] / T, m[1] / T, m[2] / T, m[3] / T, m[4] / T, m[5] / T, m[6] / T))
    for a in keep + [ws]:
        a.free()
os.environ.pop("MGR_CXXFLAGS")
subprocess.check_call([sys.executable, os.path.join(pkg, "_build.py"), "--force"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
