#!/usr/bin/env python3
"""Experiment: what does a big persistent cluster scan cost / lose when input-projection GEMMs run beside it on another
stream?  (Decides whether chunking the depth-2 projection under the depth-1 scan can pay.)

  scan  = ONE multi-scan launch of the 4 encoder layer-directions of config F (audio H=500 fwd/rev, skeletal H=300
          fwd/rev, B=64, T=1900)
  gemm  = the audio depth-2 projection shape (B*T x 1000 -> 2000), launched n times back to back
Reports: scan alone, gemms alone, both concurrently (scan on stream 1, gemms on stream 2).
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mgr_amd  # noqa: E402,F401
from mgr_amd import _capi  # noqa: E402


def main():
    dev = _capi.Device(0)
    lib = dev.lib
    B, T = 64, int(os.environ.get("PROBE_T", 1900))
    rng = np.random.default_rng(0)
    jobs, keep = [], []
    for H in [int(h) for h in os.environ.get("PROBE_HS", "500,300").split(",")]:
        for rev in (0, 1):
            Z = dev.array((rng.standard_normal((B, T, 4 * H)) * 0.5).astype(np.float32))
            Up = dev.array((rng.standard_normal((H, 4 * H)) * 0.05).astype(np.float32))
            Y = dev.empty((B, T, 2 * H))
            keep += [Z, Up, Y]
            jobs.append({"Z": Z, "Up": Up, "Y": Y.view(rev * H, (B, T, H)), "ldy": 2 * H, "B": B, "T": T,
                         "H": H, "reverse": rev})
    arr = _capi.make_scan_jobs(jobs)
    need = lib.mgr_lstm_scan_multi_ws_bytes(len(jobs), arr)
    ws = dev.bytes(need)

    def scan():
        _capi.check(lib.mgr_lstm_scan_fwd_multi(dev.ctx, len(jobs), arr, ws.ptr, ws.nbytes))

    F, H = 1000, 500
    frac = float(os.environ.get("PROBE_TFRAC", 0.25))      # GEMM over a T-chunk (what a chunked schedule would launch)
    Tc = max(16, int(T * frac))
    X = dev.array(rng.standard_normal((B, Tc, F)).astype(np.float32))
    Wp = dev.array(rng.standard_normal((F, 4 * H)).astype(np.float32) * 0.05)
    bp = dev.zeros((4 * H,))
    m = dev.array(((rng.random((4, B, F)) > 0.5) * 2.0).astype(np.float32))
    Zg = dev.empty((B, Tc, 4 * H))
    ngemm = int(os.environ.get("PROBE_NGEMM", 8))

    def gemms():
        for _ in range(ngemm):
            dev.call("mgr_lstm_input_proj", X, F, m, Wp, bp, Zg, B, Tc, F, H)

    def timed(fn_a, fn_b=None):
        # events 0/1 on stream 1, 2/3 on stream 2
        dev.sync()
        if fn_a:
            dev.stream(1)
            dev.record(0)
            fn_a()
            dev.record(1)
        if fn_b:
            dev.stream(2)
            dev.record(2)
            fn_b()
            dev.record(3)
        dev.stream(0)
        dev.sync()
        return (dev.elapsed_ms(0, 1) if fn_a else None, dev.elapsed_ms(2, 3) if fn_b else None)

    def timed_gemm_first(delay_us):
        # GEMMs are already resident when the scan arrives (the placement case the schedule avoids with mgr_stream_delay)
        dev.sync()
        dev.stream(2)
        dev.record(2)
        gemms()
        dev.record(3)
        dev.stream(1)
        dev.call("mgr_stream_delay", delay_us)
        dev.record(0)
        scan()
        dev.record(1)
        dev.stream(0)
        dev.sync()
        return dev.elapsed_ms(0, 1), dev.elapsed_ms(2, 3)

    timed(scan, gemms)
    a, _ = timed(scan)
    _, g = timed(None, gemms)
    a2, g2 = timed(scan, gemms)
    fl = 2.0 * B * Tc * F * 4 * H * ngemm
    print("scan alone        : %7.2f ms" % a)
    print("gemms alone (x%d)  : %7.2f ms  %6.1f TF" % (ngemm, g, fl / g / 1e9))
    print("concurrent        : scan %7.2f ms (x%.2f)   gemms %7.2f ms (x%.2f)" % (a2, a2 / a, g2, g2 / g))
    print("serial sum %.2f ms, concurrent wall ~%.2f ms" % (a + g, max(a2, g2)))
    for d in (500, 2000):
        a3, g3 = timed_gemm_first(d)
        print("GEMMs first, scan %4d us later: scan %7.2f ms (x%.2f)   gemms %7.2f ms (x%.2f)" % (d, a3, a3 / a, g3, g3 / g))


if __name__ == "__main__":
    main()
