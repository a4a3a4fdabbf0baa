"""CTC loss + gradient kernel alone at config F's shape (B = 64, T = 1900, C = 22, Lmax = 35, labels of 8 - 20) through library builds:
python tools/ctc_probe.py <lib.so> [<lib.so> ...]   - ms per launch; loss / dLogits compared with the FIRST library's.
CTC_PROBE_B / CTC_PROBE_T override the shape; CTC_PROBE_LOSS_ONLY=1 passes dLogits = NULL (no gradient phase)."""
import ctypes as C, os, sys
import numpy as np
vp, i32, sz, f32 = C.c_void_p, C.c_int, C.c_size_t, C.c_float

B, T = int(os.environ.get("CTC_PROBE_B", "64")), int(os.environ.get("CTC_PROBE_T", "1900"))
Cn, Lmax, skip = 22, 35, 2
ref = None
for path in sys.argv[1:]:
    lib = C.CDLL(path)
    lib.mgr_ctc_ws_bytes.restype = sz
    lib.mgr_ctc_ws_bytes.argtypes = [i32, i32, i32, i32]
    lib.mgr_last_error.restype = C.c_char_p
    lib.mgr_ctx_create.argtypes = [i32, C.POINTER(vp)]
    lib.mgr_alloc.argtypes = [vp, sz, C.POINTER(vp)]
    lib.mgr_h2d.argtypes = [vp, vp, vp, sz]
    lib.mgr_d2h.argtypes = [vp, vp, vp, sz]
    lib.mgr_ctc_loss_grad.argtypes = [vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, f32, f32, vp, vp, vp, sz]
    lib.mgr_sync.argtypes = [vp]
    lib.mgr_event_record.argtypes = [vp, i32]
    lib.mgr_event_elapsed_ms.argtypes = [vp, i32, i32, C.POINTER(f32)]
    ctx = vp()
    assert lib.mgr_ctx_create(0, C.byref(ctx)) == 0

    def alloc(n):
        p = vp()
        assert lib.mgr_alloc(ctx, n, C.byref(p)) == 0
        return p

    def up(a):
        p = alloc(a.nbytes)
        lib.mgr_h2d(ctx, p, a.ctypes.data, a.nbytes)
        return p

    rng = np.random.default_rng(7)
    z = rng.standard_normal((B, T, Cn)) * 2.0
    P = np.exp(z - z.max(-1, keepdims=True))
    P = (P / P.sum(-1, keepdims=True)).astype(np.float32)
    labels = np.zeros((B, Lmax), np.int32)
    ll = rng.integers(8, 21, B).astype(np.int32)
    for b in range(B):
        labels[b, :ll[b]] = rng.integers(0, Cn - 1, ll[b])
    il = np.full(B, T - skip, np.int32)
    dP, dl, dil, dll = up(P), up(labels), up(il), up(ll)
    loss, dL = alloc(B * 4), alloc(B * T * Cn * 4)
    n = lib.mgr_ctc_ws_bytes(B, T, Cn, Lmax)
    ws = alloc(n)
    args = (ctx, dP, dl, dil, dll, B, T, Cn, Lmax, skip, Cn - 1, 1e-7, 1.0 / B, loss, None if os.environ.get("CTC_PROBE_LOSS_ONLY") else dL, ws, n)
    assert lib.mgr_ctc_loss_grad(*args) == 0, lib.mgr_last_error()
    lib.mgr_sync(ctx)
    lib.mgr_event_record(ctx, 0)
    for _ in range(10):
        lib.mgr_ctc_loss_grad(*args)
    lib.mgr_event_record(ctx, 1)
    ms = f32()
    lib.mgr_event_elapsed_ms(ctx, 0, 1, C.byref(ms))
    lo, g = np.empty(B, np.float32), np.empty((B, T, Cn), np.float32)
    lib.mgr_d2h(ctx, lo.ctypes.data, loss, lo.nbytes)
    lib.mgr_d2h(ctx, g.ctypes.data, dL, g.nbytes)
    if ref is None:
        ref, cmp = (lo, g), "reference"
    else:
        cmp = "loss rel %.2e, grad max |diff| / max %.2e" % (np.abs(lo - ref[0]).max() / np.abs(ref[0]).max(), np.abs(g - ref[1]).max() / np.abs(ref[1]).max())
    print("%-28s %7.3f ms per launch  %6.1f ns per step   %s (finite %s)" % (os.path.basename(path), ms.value / 10, ms.value / 10 * 1e6 / T, cmp,
                                                                          bool(np.isfinite(lo).all() and np.isfinite(g).all())), flush=True)
