"""BPTT cluster scan of the fusion layer's shape (H = 100, both directions in one launch, B = 64, T = 1900) through library builds:
python tools/bptt_probe.py <lib.so> [<lib.so> ...]   - time per launch, dZ / dzmax compared with the FIRST library's (bit-identical or the
largest difference), and the cycle stamps of a -DMGR_STAMP build.  BPTT_PROBE_H=100,300 selects the layer sizes."""
import ctypes as C, os, sys
import numpy as np
vp, i32, sz = C.c_void_p, C.c_int, C.c_size_t


class BwdJob(C.Structure):
    _fields_ = [("dY", vp), ("gates", vp), ("cs", vp), ("Up", vp), ("dZ", vp), ("lddy", i32), ("B", i32), ("T", i32), ("H", i32),
                ("reverse", i32), ("dzmax", vp), ("dbsum", vp)]


B, T = 64, int(os.environ.get("BPTT_PROBE_T", "1900"))
HS = [int(h) for h in os.environ.get("BPTT_PROBE_H", "100").split(",")]
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
ref = {}
for path in sys.argv[1:]:
    tune16 = 0
    if path.endswith(":16"):      # "<lib.so>:16" = the same library with tune key 16 = 1 (the form used beside other persistent launches)
        path, tune16 = path[:-3], 1
    elif path.endswith(":16=2"):  # ... = 2: the direct gather (one barrier per step; the form used beside fused encoder scans)
        path, tune16 = path[:-5], 2
    lib = C.CDLL(path)
    lib.mgr_lstm_scan_bwd_multi_ws_bytes.restype = sz
    lib.mgr_lstm_scan_bwd_multi_ws_bytes.argtypes = [i32, vp]
    lib.mgr_last_error.restype = C.c_char_p
    lib.mgr_ctx_create.argtypes = [i32, C.POINTER(vp)]
    lib.mgr_alloc.argtypes = [vp, sz, C.POINTER(vp)]
    lib.mgr_h2d.argtypes = [vp, vp, vp, sz]
    lib.mgr_d2h.argtypes = [vp, vp, vp, sz]
    lib.mgr_lstm_scan_bwd_multi.argtypes = [vp, i32, vp, vp, sz]
    lib.mgr_sync.argtypes = [vp]
    lib.mgr_event_record.argtypes = [vp, i32]
    lib.mgr_event_elapsed_ms.argtypes = [vp, i32, i32, C.POINTER(C.c_float)]
    ctx = vp()
    assert lib.mgr_ctx_create(0, C.byref(ctx)) == 0
    if tune16:
        lib.mgr_tune.argtypes = [vp, i32, i32]
        assert lib.mgr_tune(ctx, 16, tune16) == 0

    def alloc(n):
        p = vp()
        assert lib.mgr_alloc(ctx, n, C.byref(p)) == 0
        return p

    def up(a):
        p = alloc(a.nbytes)
        lib.mgr_h2d(ctx, p, a.ctypes.data, a.nbytes)
        return p

    for H in HS:
        rng = np.random.default_rng(H)
        jobs, outs = [], []
        for rev in (0, 1):
            dy = (rng.standard_normal((B, T, H)) * 0.01).astype(np.float32)
            g = rng.random((B, T, H, 4)).astype(np.float32)
            g[..., 2] = g[..., 2] * 2 - 1
            cs = (rng.standard_normal((B, T, H)) * 0.5).astype(np.float32)
            u = (rng.standard_normal((H, 4 * H)) * 0.05).astype(np.float32)
            dZ, mx = alloc(B * T * 4 * H * 4), alloc(B * 4 * H * 4)
            jobs.append((up(dy), up(g), up(cs), up(u), dZ, mx, rev))
            outs.append((dZ, mx))
        arr = (BwdJob * len(jobs))()
        for a, (dy, g, cs, u, dZ, mx, rev) in zip(arr, jobs):
            a.dY, a.gates, a.cs, a.Up, a.dZ, a.dzmax = dy.value, g.value, cs.value, u.value, dZ.value, mx.value
            a.lddy, a.B, a.T, a.H, a.reverse = H, B, T, H, rev
        n = lib.mgr_lstm_scan_bwd_multi_ws_bytes(len(jobs), C.cast(arr, vp))
        ws = alloc(n)
        assert lib.mgr_lstm_scan_bwd_multi(ctx, len(jobs), C.cast(arr, vp), ws, n) == 0, lib.mgr_last_error()
        lib.mgr_sync(ctx)
        lib.mgr_event_record(ctx, 0)
        for _ in range(4):
            lib.mgr_lstm_scan_bwd_multi(ctx, len(jobs), C.cast(arr, vp), ws, n)
        lib.mgr_event_record(ctx, 1)
        ms = C.c_float()
        lib.mgr_event_elapsed_ms(ctx, 0, 1, C.byref(ms))
        got = []
        for dZ, mx in outs:
            a, m = np.empty((B, T, 4 * H), np.float32), np.empty((B, 4 * H), np.float32)
            lib.mgr_d2h(ctx, a.ctypes.data, dZ, a.nbytes)
            lib.mgr_d2h(ctx, m.ctypes.data, mx, m.nbytes)
            got += [a, m]
        if H not in ref:
            ref[H] = got
            cmp = "reference"
        else:
            same = all(np.array_equal(a, b) for a, b in zip(got, ref[H]))
            cmp = "bit-identical" if same else "max |diff| / max %.2e" % max(np.abs(a - b).max() / np.abs(b).max() for a, b in zip(got, ref[H]))
            if not same and os.environ.get("BPTT_PROBE_WHERE"):
                for nm, a, b in zip(("dZ fwd", "dzmax fwd", "dZ rev", "dzmax rev"), got, ref[H]):
                    d = np.argwhere(a != b)
                    if len(d):
                        i = tuple(d[0])
                        print("   %s: %d of %d differ; first at %s: %r vs %r; gates differing: %s" % (nm, len(d), a.size, i, a[i], b[i], sorted(set(int(x[-1]) % 4 for x in d[:2000]))))
        print("%-24s H=%-4d %7.3f ms  %5.2f us/step   %s  (finite: %s)" % (os.path.basename(path) + (":16=%d" % tune16 if tune16 else ""), H, ms.value / 4, ms.value / 4 * 1e3 / T, cmp,
                                                                             all(np.isfinite(a).all() for a in got)), flush=True)
        if hasattr(lib, "mgr_debug_bstamps"):
            out = (C.c_ulonglong * 64)()
            lib.mgr_debug_bstamps(out)
            if out[8]:
                nn = float(out[8])
                names = ("gather+verify", "reduce+barrier", "cell bwd+dZ", "scale+image", "barrier", "mfma+publish")
                print("   cycles/step: " + " | ".join("%s %5.0f" % (nm, out[i] / nn) for i, nm in enumerate(names))
                      + "  (sum %5.0f)" % (sum(out[i] for i in range(6)) / nn))
