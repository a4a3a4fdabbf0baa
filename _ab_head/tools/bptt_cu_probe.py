"""The narrow-layer BPTT alone on the chip, per launch form (mgr_scan_launch_opts.form): ms per launch, us per step, max relative
difference of dZ to the trimmed multi-CU form.  python tools/bptt_cu_probe.py [H] [B] [T]"""
import ctypes, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import mgr_amd
from mgr_amd import _capi
H = int(sys.argv[1]) if len(sys.argv) > 1 else 100
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
T = int(sys.argv[3]) if len(sys.argv) > 3 else 1900
if len(sys.argv) > 4:
    _capi.LIB_PATH = os.path.abspath(sys.argv[4])   # (a variant library: tools/build_variants.sh)
forms = [("single-CU", 6)] if len(sys.argv) > 4 else [("trimmed", 1), ("yielding", 2), ("direct", 3), ("fused", 4), ("single-CU", 6)]
dev = _capi.Device(0)
rng = np.random.default_rng(0)
jobs, keep = [], []
for d in range(2):
    gates = dev.array(rng.uniform(0.05, 0.95, (B, T, H, 4)).astype(np.float32))
    cs = dev.array(rng.standard_normal((B, T, H)).astype(np.float32) * 0.5)
    dY = dev.array((rng.standard_normal((B, T, H)) * 1e-3).astype(np.float32))
    U = dev.array((rng.standard_normal((H, 4 * H)) * 0.3 / np.sqrt(H)).astype(np.float32))
    Up = dev.empty((H, 4 * H)); dev.call("mgr_lstm_pack", U, Up, H, H, 0)
    dZ = dev.zeros((B, T, 4 * H)); zm = dev.zeros((B, 4 * H), np.uint32)
    jobs.append(dict(dY=dY, gates=gates, cs=cs, Up=Up, dZ=dZ, lddy=H, B=B, T=T, H=H, reverse=d, dzmax=zm))
arr = _capi.make_scan_bwd_jobs(jobs)
ws = dev.bytes(dev.lib.mgr_lstm_scan_bwd_multi_ws_bytes(2, arr))
ref = None
for name, form in forms:
    opts = _capi.make_launch_opts(form, 0)
    best = 1e9
    for rep in range(3):
        dev.stream(0); dev.record(0)
        _capi.check(dev.lib.mgr_lstm_scan_bwd_multi_ex(dev.ctx, 2, arr, ws.ptr, ws.nbytes, ctypes.byref(opts)))
        dev.record(1); dev.sync()
        best = min(best, dev.elapsed_ms(0, 1))
    dz = jobs[0]["dZ"].download()
    if ref is None:
        ref = dz
    err = float(np.abs(dz - ref).max() / np.abs(ref).max())
    print("%-10s %.3f ms  %.2f us/step   max rel diff to trimmed %.2e" % (name, best, best * 1e3 / T, err), flush=True)
