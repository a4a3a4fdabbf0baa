#!/usr/bin/env python3
"""Diagnostic (never timed): stamped build of the cluster scan inside the real F-config step (4 jobs in one launch)."""
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
from mgr_amd.configs import baseline_config  # noqa: E402
from mgr_amd.engine import Engine  # noqa: E402
from mgr_amd.synthetic import synthetic_arrays, synthetic_weights  # noqa: E402

path = int(sys.argv[1]) if len(sys.argv) > 1 else 0
spec, B, T, Lmax = baseline_config("F")
dev = _capi.Device(0)
dev.call("mgr_tune", 2, 1)
dev.call("mgr_tune", 0, path)
eng = Engine(spec, B, T, Lmax, device=dev)
eng.set_weights(synthetic_weights(spec, 3))
xs, labels, il, ll = synthetic_arrays(spec, B, T, Lmax, 5)
eng._upload_inputs(xs, None, True)
eng._upload_labels(labels, il, ll)
# run only the encoder part twice: hijack by calling forward then reading the multi workspace after depth-1 scan
orig = eng._scan_multi
calls = []


def spy(jobs):
    orig(jobs)
    dev.sync()
    raw = eng._ws_multi.download().view(np.uint64)
    calls.append(raw[8:8 + 16 * 8].reshape(16, 8).copy())


eng._scan_multi = spy
eng._forward(True, None)
dev.sync()
for ci, dbg in enumerate(calls):
    print("scan call %d (cycles per step: mfma-issue, compute, gather, barrier, passes)" % ci)
    for w in range(16):
        m = dbg[w]
        if m[1] == 0:
            continue
        print("   wg%d.w%d %6.0f %6.0f %6.0f %6.0f   %.2f" % (w // 8, w % 8, m[0] / T, m[1] / T, m[2] / T, m[3] / T, m[4] / T))
os.environ.pop("MGR_CXXFLAGS")
subprocess.check_call([sys.executable, os.path.join(pkg, "_build.py"), "--force"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
