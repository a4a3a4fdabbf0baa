#!/usr/bin/env python3
"""Timing ablation of the K-split cluster step (-DMGR_ABLATE build; results are WRONG in modes 101-103, only the time
is meaningful): mode = 200 + bit flags: 1 never wait for a peer's epoch, 2 no MFMAs, 4 no gather loads, 8 no publish store,
16 no Y store, 32 no Z prefetch."""
import os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["MGR_CXXFLAGS"] = "-DMGR_ABLATE"
pkg = os.path.join(ROOT, "multimodal-gesture-recognition-with-lstms-and-ctc_amd")
subprocess.check_call([sys.executable, os.path.join(pkg, "_build.py"), "--force"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
import mgr_amd  # noqa
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
    for mode in (0, 200 + 1, 200 + 3, 200 + 3 + 32, 200 + 3 + 16, 200 + 3 + 8, 200 + 3 + 8 + 16 + 32, 200 + 7, 200 + 7 + 8 + 16 + 32, 200 + 4 + 8 + 16 + 32):
        dev.call("mgr_tune", 6, mode)
        _capi.check(lib.mgr_lstm_scan_fwd_multi(dev.ctx, len(jobs), arr, ws.ptr, ws.nbytes)); dev.sync()
        dev.record(0)
        for _ in range(3):
            _capi.check(lib.mgr_lstm_scan_fwd_multi(dev.ctx, len(jobs), arr, ws.ptr, ws.nbytes))
        dev.record(1); dev.sync()
        ms = dev.elapsed_ms(0, 1) / 3
        print("H=%-10s mode %3d : %7.3f ms  %5.2f us/step" % (hs, mode, ms, ms * 1e3 / T))
    dev.call("mgr_tune", 6, 0)
    for a in keep + [ws]:
        a.free()
os.environ.pop("MGR_CXXFLAGS")
subprocess.check_call([sys.executable, os.path.join(pkg, "_build.py"), "--force"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
