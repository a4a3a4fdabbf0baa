#!/usr/bin/env python3
"""Experiment: do the depth-1 input projections (F = 39 / 20: Z-store bound, 3.1 GB written) run cheaply BESIDE a resident encoder
scan launch, i.e. could the next step's depth-1 projections leave the encoder stream's chain and hide under the depth-2 scans?
  scan = the 4 encoder recurrences of one depth (audio H=500, skeletal H=300, both directions), B=64, T=1900, stream 1
  p1   = the four depth-1 projections (dropout-aware path, keep 0.6 / 0.4), stream 2, started 1 ms after the scan launch
Reports each alone and both together."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mgr_amd  # noqa: E402,F401
import numpy as np  # noqa: E402
from mgr_amd import _capi  # noqa: E402


def main():
    dev = _capi.Device(0)
    lib = dev.lib
    B, T = 64, 1900
    rng = np.random.default_rng(0)
    jobs, keep = [], []
    for H in (500, 300):
        for rev in (0, 1):
            Z = dev.array((rng.standard_normal((B, T, 4 * H)) * 0.5).astype(np.float32))
            Up = dev.array((rng.standard_normal((H, 4 * H)) * 0.05).astype(np.float32))
            Y = dev.empty((B, T, 2 * H))
            keep += [Z, Up, Y]
            jobs.append({"Z": Z, "Up": Up, "Y": Y.view(rev * H, (B, T, H)), "ldy": 2 * H, "B": B, "T": T, "H": H, "reverse": rev})
    arr = _capi.make_scan_jobs(jobs)
    ws = dev.bytes(lib.mgr_lstm_scan_multi_ws_bytes(len(jobs), arr))

    def scan():
        _capi.check(lib.mgr_lstm_scan_fwd_multi(dev.ctx, len(jobs), arr, ws.ptr, ws.nbytes))

    projs = []
    for F, H, drop in ((39, 500, 0.4), (39, 500, 0.4), (20, 300, 0.6), (20, 300, 0.6)):
        X = dev.array(rng.standard_normal((B, T, F)).astype(np.float32))
        Wp = dev.array(rng.standard_normal((F, 4 * H)).astype(np.float32) * 0.05)
        bp = dev.zeros((4 * H,))
        m = dev.array(((rng.random((4, B, F)) > drop) / (1 - drop)).astype(np.float32))
        Zg = dev.empty((B, T, 4 * H))
        wsp = dev.bytes(lib.mgr_lstm_input_proj_dropout_ws_bytes(B, F, H))
        projs.append((X, F, m, drop, Wp, bp, Zg, H, wsp))

    def p1():
        for X, F, m, drop, Wp, bp, Zg, H, wsp in projs:
            dev.call("mgr_lstm_input_proj_dropout", X, F, m, float(drop), Wp, bp, Zg, B, T, F, H, wsp, wsp.nbytes)

    def timed(fa, fb, delay_us=1000):
        dev.sync()
        if fa:
            dev.stream(1)
            dev.record(0)
            fa()
            dev.record(1)
        if fb:
            dev.stream(2)
            if fa:
                dev.call("mgr_stream_delay", delay_us)
            dev.record(2)
            fb()
            dev.record(3)
        dev.stream(0)
        dev.sync()
        return (dev.elapsed_ms(0, 1) if fa else None, dev.elapsed_ms(2, 3) if fb else None)

    timed(scan, p1)
    a, _ = timed(scan, None)
    _, g = timed(None, p1)
    a2, g2 = timed(scan, p1)
    print("scan alone %.2f ms | depth-1 projections alone %.2f ms | together: scan %.2f ms (x%.2f), projections %.2f ms (x%.2f)"
          % (a, g, a2, a2 / a, g2, g2 / g))


if __name__ == "__main__":
    main()
