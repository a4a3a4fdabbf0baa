"""Same-box A/B of the forward cluster scans between library builds with DIFFERENT C ABIs (e.g. an older round's libmgr.so): raw ctypes,
only the entry points every round has.  python tools/scan_ab_raw.py <lib.so> [<lib.so> ...]; SCAN_PROBE_H as tools/scan_variant_probe.py."""
import ctypes as C, os, sys
import numpy as np
vp, i32, sz = C.c_void_p, C.c_int, C.c_size_t


class ScanJob(C.Structure):
    _fields_ = [("Z", vp), ("Up", vp), ("Y", vp), ("R", vp), ("gates", vp), ("cs", vp), ("ldy", i32), ("ldr", i32), ("B", i32), ("T", i32),
                ("H", i32), ("reverse", i32), ("YT", vp), ("ytb", C.c_longlong), ("ldt", i32), ("yt_split", i32)]


B, T = 64, 1900
HS = [tuple(int(h) for h in a.split("+")) for a in os.environ.get("SCAN_PROBE_H", "500+300,500").split(",")]
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
ref_out = {}
for path in sys.argv[1:]:
    tunes = []
    while ":" in path and "=" in path.rsplit(":", 1)[1]:      # "<lib.so>:KEY=VALUE[:KEY=VALUE]" = mgr_tune settings for this run
        path, kv = path.rsplit(":", 1)
        tunes.append(tuple(int(x) for x in kv.split("=")))
    lib = C.CDLL(path)
    lib.mgr_lstm_scan_multi_ws_bytes.restype = sz
    lib.mgr_lstm_scan_multi_ws_bytes.argtypes = [i32, vp]
    lib.mgr_last_error.restype = C.c_char_p
    lib.mgr_ctx_create.argtypes = [i32, C.POINTER(vp)]
    lib.mgr_alloc.argtypes = [vp, sz, C.POINTER(vp)]
    lib.mgr_h2d.argtypes = [vp, vp, vp, sz]
    lib.mgr_d2h.argtypes = [vp, vp, vp, sz]
    lib.mgr_lstm_scan_fwd_multi.argtypes = [vp, i32, vp, vp, sz]
    lib.mgr_sync.argtypes = [vp]
    lib.mgr_event_record.argtypes = [vp, i32]
    lib.mgr_event_elapsed_ms.argtypes = [vp, i32, i32, C.POINTER(C.c_float)]
    lib.mgr_ctx_destroy.argtypes = [vp]
    ctx = vp()
    assert lib.mgr_ctx_create(0, C.byref(ctx)) == 0
    if tunes:
        lib.mgr_tune.argtypes = [vp, i32, i32]
        for k, v in tunes:
            assert lib.mgr_tune(ctx, k, v) == 0

    def alloc(n):
        p = vp()
        assert lib.mgr_alloc(ctx, n, C.byref(p)) == 0
        return p

    rng = np.random.default_rng(0)
    for hs in HS:
        jobs = []
        for H in hs:
            for rev in (0, 1):
                z = (rng.standard_normal((B, T, 4 * H)) * 0.5).astype(np.float32)
                u = (rng.standard_normal((H, 4 * H)) * 0.05).astype(np.float32)
                Z, U, Y = alloc(z.nbytes), alloc(u.nbytes), alloc(B * T * H * 4)
                lib.mgr_h2d(ctx, Z, z.ctypes.data, z.nbytes); lib.mgr_h2d(ctx, U, u.ctypes.data, u.nbytes)
                jobs.append((Z, U, Y, H, rev))
        arr = (ScanJob * len(jobs))()
        for a, (Z, U, Y, H, rev) in zip(arr, jobs):
            a.Z, a.Up, a.Y, a.ldy, a.B, a.T, a.H, a.reverse = Z.value, U.value, Y.value, H, B, T, H, rev
        n = lib.mgr_lstm_scan_multi_ws_bytes(len(jobs), C.cast(arr, vp))
        ws = alloc(n)
        assert lib.mgr_lstm_scan_fwd_multi(ctx, len(jobs), C.cast(arr, vp), ws, n) == 0, lib.mgr_last_error()
        lib.mgr_sync(ctx)
        lib.mgr_event_record(ctx, 0)
        for _ in range(4):
            lib.mgr_lstm_scan_fwd_multi(ctx, len(jobs), C.cast(arr, vp), ws, n)
        lib.mgr_event_record(ctx, 1)
        ms = C.c_float()
        lib.mgr_event_elapsed_ms(ctx, 0, 1, C.byref(ms))
        print("%-28s H=%-10s %7.3f ms  %5.2f us/step" % (os.path.basename(path) + "".join(":%d=%d" % kv for kv in tunes), hs, ms.value / 4, ms.value / 4 * 1e3 / T), flush=True)
        outs = []
        for (Z, U, Y, H, rev) in jobs:
            y = np.empty((B, T, H), np.float32)
            lib.mgr_d2h(ctx, y.ctypes.data, Y, y.nbytes)
            outs.append(y)
        if hs not in ref_out:
            ref_out[hs] = outs
        else:
            same = all(np.array_equal(a, b) for a, b in zip(outs, ref_out[hs]))
            print("   outputs vs the first run: %s" % ("bit-identical" if same else "DIFFERENT (max |diff| %.3e)" % max(np.abs(a - b).max() for a, b in zip(outs, ref_out[hs]))), flush=True)
        if hasattr(lib, "mgr_debug_stamps"):   # a -DMGR_STAMP build: cycles per phase of the k16 step, averaged over all waves
            out = (C.c_ulonglong * 64)()
            lib.mgr_debug_stamps(out)
            for cls, name in ((0, "H>400"), (16, "H<=400")):
                if out[cls + 8]:
                    n = float(out[cls + 8])
                    print("   %-7s cycles/step: gather %5.0f | mfma+partials %5.0f | wait+barrier %5.0f | reduce+cell+flags %5.0f | publish %5.0f | outputs %5.0f"
                          "  (sum %5.0f; %.3f re-fetch rounds/step)" % ((name,) + tuple(out[cls + i] / n for i in range(6))
                                                                         + (sum(out[cls + i] for i in range(6)) / n, out[cls + 9] / n)))
                    print("   %-7s one wave (wg 0, wave 0):  %s" % (name, " ".join("%5.0f" % (out[32 + cls // 2 + i] / float(T)) for i in range(6))))
    lib.mgr_ctx_destroy(ctx)
