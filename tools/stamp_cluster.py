#!/usr/bin/env python3
"""Diagnostic (never timed): rebuild with -DMGR_STAMP and print where a cluster-scan step spends its cycles."""
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
B, T = 64, 1900
for H, path in ((500, 0), (500, 4), (300, 0)):
    rng = np.random.default_rng(0)
    Z = dev.array((rng.standard_normal((B, T, 4 * H)) * 0.5).astype(np.float32))
    Up = dev.array((rng.standard_normal((H, 4 * H)) * 0.05).astype(np.float32))
    Y = dev.empty((B, T, H))
    ws = dev.bytes(dev.lib.mgr_lstm_scan_ws_bytes(B, T, H))
    dev.call("mgr_tune", 0, path)
    dev.call("mgr_lstm_scan_fwd", Z, Up, Y, H, 0, 0, 0, 0, B, T, H, 0, ws, ws.nbytes)
    dev.sync()
    dev.record(0)
    dev.call("mgr_lstm_scan_fwd", Z, Up, Y, H, 0, 0, 0, 0, B, T, H, 0, ws, ws.nbytes)
    dev.record(1)
    ms = dev.elapsed_ms(0, 1)
    raw = ws.download().view(np.uint64)
    dbg = raw[8:8 + 16 * 8].reshape(16, 8)
    print("H=%d path=%d: %.3f ms (%.2f us/step, stamped build)  status=%d" % (H, path, ms, ms * 1e3 / T, raw[0] & 0xFFFFFFFF))
    print("   wave  mfma  compute gather barrier  passes/step   (cycles per step, s_memtime ticks @100MHz?)")
    for w in range(16):
        m = dbg[w]
        if m[1] == 0:
            continue
        print("   wg%d.w%d %6.0f %6.0f %6.0f %6.0f   %.2f" % (w // 8, w % 8, m[0] / T, m[1] / T, m[2] / T, m[3] / T, m[4] / T))
dev.call("mgr_tune", 0, 0)
# restore the normal build
os.environ.pop("MGR_CXXFLAGS")
subprocess.check_call([sys.executable, os.path.join(pkg, "_build.py"), "--force"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
