"""Forward cluster scans of config F's encoder depth (audio H=500, skeletal H=300, both directions, B = 64, T = 1900): audio alone,
skeletal alone, all four in ONE launch - under mgr_tune settings given on the command line as key=value[,key=value...] groups:
    python tools/scan_variant_probe.py "" 3=1 7=1      # default (K-split step, XCD-local) | write-through exchange | LDS-image step
Prints ms per launch, us per time step, the launch's give-up word and the largest difference to the first setting."""
import os, sys
sys.path.insert(0, os.getcwd())
import mgr_amd   # (before numpy: _hostenv.py)
import numpy as np
from mgr_amd import _capi
dev = _capi.Device(0); lib = dev.lib
B, T = 64, 1900
rng = np.random.default_rng(0)
settings = sys.argv[1:] or [""]


def parse(s):
    return [tuple(int(x) for x in kv.split("=")) for kv in s.split(",") if kv]


HS = [tuple(int(h) for h in a.split("+")) for a in os.environ.get("SCAN_PROBE_H", "500,300,500+300").split(",")]
for hs in HS:
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
    for s in settings:
        kv = parse(s)
        for k, v in kv:
            dev.call("mgr_tune", k, v)
        dev.call("mgr_tune", 1, 1)
        try:
            _capi.check(lib.mgr_lstm_scan_fwd_multi(dev.ctx, len(jobs), arr, ws.ptr, ws.nbytes)); dev.sync()
            dev.record(0)
            for _ in range(4):
                _capi.check(lib.mgr_lstm_scan_fwd_multi(dev.ctx, len(jobs), arr, ws.ptr, ws.nbytes))
            dev.record(1); dev.sync()
            ms = dev.elapsed_ms(0, 1) / 4
            y = np.concatenate([keep[2 + 3 * i].download().ravel() for i in range(0, len(jobs), 2)])
            if ref is None: ref = y
            print("H=%-10s tune %-14s : %7.3f ms  %5.2f us/step  status=%d  maxdiff=%.2e" % (hs, s or "-", ms, ms * 1e3 / T, int(ws.download().view(np.uint32)[0]), float(np.abs(y - ref).max())), flush=True)
        except _capi.MgrError as e:
            print("H=%-10s tune %-14s : FAILED %s" % (hs, s, e), flush=True)
        for k, v in kv:
            dev.call("mgr_tune", k, 0)
        dev.call("mgr_tune", 1, 0)
    for a in keep + [ws]: a.free()
