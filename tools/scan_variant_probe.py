"""Forward cluster scan variants at the config-F shapes (B = 64, T = 1900), audio alone and audio + skeletal in one launch:
  tune7 = 0  K-split step, permuted unit order      tune7 = 2  same, identity unit order      tune7 = 1  LDS-image step
(round 2 also measured a PAIRED form - two batch groups per 8-wave workgroup, matrix / cell wave roles, LDS-DMA landing zones;
 source kept as tools/probes/lstm_cluster_pair.hip.txt, numbers in profiles/r02_pair_scan_probe.txt, discussion in DESIGN.md 5)
Prints ms per launch, us per time step, the launch's give-up word and the largest difference to the first variant."""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
import mgr_amd
from mgr_amd import _capi
dev = _capi.Device(0); lib = dev.lib
B, T = 64, 1900
rng = np.random.default_rng(0)
variants = [(0, 0), (0, 3), (2, 0), (1, 0)]   # (tune7, x): x = 3 turns the XCD-local exchange of the K-split step OFF
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
    ref = None
    for t7, t10 in variants:
        dev.call("mgr_tune", 7, t7); dev.call("mgr_tune", 3, 1 if t10 == 3 else 0); dev.call("mgr_tune", 1, 1)
        try:
            _capi.check(lib.mgr_lstm_scan_fwd_multi(dev.ctx, len(jobs), arr, ws.ptr, ws.nbytes)); dev.sync()
            dev.record(0)
            for _ in range(3):
                _capi.check(lib.mgr_lstm_scan_fwd_multi(dev.ctx, len(jobs), arr, ws.ptr, ws.nbytes))
            dev.record(1); dev.sync()
        except _capi.MgrError as e:
            print("H=%-10s tune7=%d tune3=%d : FAILED %s" % (hs, t7, t10, e)); continue
        ms = dev.elapsed_ms(0, 1) / 3
        y = keep[2].download()
        if ref is None: ref = y
        print("H=%-10s tune7=%d tune3=%d : %7.3f ms  %5.2f us/step  status=%d  maxdiff=%.2e" % (hs, t7, t10, ms, ms * 1e3 / T, int(ws.download().view(np.uint32)[0]), float(np.abs(y - ref).max())), flush=True)
    dev.call("mgr_tune", 7, 0); dev.call("mgr_tune", 3, 0); dev.call("mgr_tune", 1, 0)
    for a in keep + [ws]: a.free()
