"""-m gpu: the projection / weight-gradient products on PRE-SPLIT operands (gemm_split.hip, round 5) against numpy fp64 and against
the kernels they replace, through the C ABI."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
f32 = np.float32


def _split_rows(X, ldt):
    """numpy statement of the split row format (mgr.h): XS[b][f] = hi(t) f16 x ldt | lo(t) f16 x ldt of x 2^13, as float32 words."""
    B, T, F = X.shape
    xs = np.zeros((B, F, 2, ldt), np.float16)
    s = (X.transpose(0, 2, 1) * f32(8192.0)).astype(f32)
    hi = s.astype(np.float16)
    lo = (s - hi.astype(f32)).astype(np.float16)
    xs[:, :, 0, :T] = hi
    xs[:, :, 1, :T] = lo
    return xs.reshape(B, F, 2 * ldt).view(f32)


@pytest.mark.parametrize("B,T,F,H,p", [(2, 200, 1000, 132, 0.5), (3, 130, 64, 100, 0.5), (2, 257, 1600, 100, 0.5), (1, 128, 48, 300, 0.6),
                                        (2, 90, 16, 20, 0.9), (2, 140, 600, 300, 0.6), (2, 77, 131, 500, 0.4), (2, 64, 160, 40, 1.0),
                                        (2, 100, 96, 64, 0.0)])
def test_projection_from_split_rows(device, B, T, F, H, p):
    dev = device
    rng = np.random.default_rng(B * 1000 + T + F + H)
    N = 4 * H
    X = rng.uniform(-2, 2, (B, T, F)).astype(f32)
    W = (rng.standard_normal((F, N)) * 0.1).astype(f32)
    bias = rng.standard_normal(N).astype(f32)
    c = f32(1.0 / (1.0 - p)) if p < 1.0 else f32(1.0)
    M = ((rng.random((4, B, F)) >= p) * c).astype(f32)
    if p == 0.0:
        M[:] = 1.0
    ldt = (T + 127) // 128 * 128
    dX, dW, db, dM = dev.array(X), dev.array(W), dev.array(bias), dev.array(M)
    XS = dev.empty((B, F, ldt))
    XS.upload(np.full((B, F, ldt), np.nan, f32))            # the producer must write the padding too
    dev.call("mgr_transpose_bt_split", dX, F, XS, ldt, B, T, F)
    assert np.array_equal(XS.download().view(np.uint32), _split_rows(X, ldt).view(np.uint32))
    ws = dev.bytes(dev.lib.mgr_lstm_input_proj_dropout_ts_ws_bytes(B, F, H))
    dev.call("mgr_memset", ws, 0xFF, ws.nbytes)              # the workspace arrives dirty
    Z = dev.empty((B, T, N))
    gate = np.arange(N) % 4
    ref = np.empty((B, T, N))
    for g in range(4):
        ref[:, :, gate == g] = (X.astype(np.float64) * M[g][:, None, :]) @ W[:, gate == g].astype(np.float64) + bias[gate == g]
    tol = 2e-5 * max(1.0, np.abs(ref).max())
    outs = []
    for tile in (1, 2, 0):        # tune key 12: 128 x 64 tiles (4 waves), 128 x 128 (8 waves), the library's choice
        dev.call("mgr_tune", 12, tile)
        Z.upload(np.full((B, T, N), np.nan, f32))
        dev.call("mgr_lstm_input_proj_dropout_ts", XS, ldt, dM, p if p < 0.99 else 0.5, dW, db, Z, B, T, F, H, ws, ws.nbytes)
        got = Z.download()
        assert np.all(np.isfinite(got)) and np.abs(got - ref).max() <= tol, (tile, np.abs(got - ref).max())
        outs.append(got)
    assert np.array_equal(outs[0], outs[1])      # the same sums in the same order, whatever the tile
    # no mask at all (inference): every feature, factor 1
    dev.call("mgr_lstm_input_proj_dropout_ts", XS, ldt, 0, 0.0, dW, db, Z, B, T, F, H, ws, ws.nbytes)
    ref0 = X.astype(np.float64) @ W.astype(np.float64) + bias
    assert np.abs(Z.download() - ref0).max() <= 2e-5 * max(1.0, np.abs(ref0).max())
    # two different mask factors in one call: not what the kernel was written for - NaN, never a plausible number
    if 0.0 < p < 1.0 and M.max() > 0:
        M2 = M.copy()
        g0, b0, f0 = np.argwhere(M2 > 0)[0]
        M2[g0, b0, f0] *= f32(1.5)
        dev.call("mgr_lstm_input_proj_dropout_ts", XS, ldt, dev.array(M2), p, dW, db, Z, B, T, F, H, ws, ws.nbytes)
        assert np.all(np.isnan(Z.download()))


@pytest.mark.parametrize("B,T,F,H,p,reverse", [(3, 130, 128, 100, 0.5, 0), (2, 300, 1600, 100, 0.5, 1), (2, 77, 1000, 130, 0.5, 0),
                                               (2, 140, 600, 300, 0.6, 1), (2, 100, 131, 20, 0.9, 0), (2, 64, 160, 40, 1.0, 0),
                                               (9, 50, 200, 64, 0.5, 0)])
def test_weight_gradient_from_split_rows(device, B, T, F, H, p, reverse):
    """mgr_lstm_param_grads_dropout_ts against numpy fp64 (dW) and against mgr_lstm_param_grads (dU, db: the shared path, bit for
    bit); the gate gradients spread over 24 orders of magnitude per (sample, gate column) - the per-row scaling must not care."""
    dev = device
    rng = np.random.default_rng(B * 977 + T + F + H)
    N = 4 * H
    X = rng.uniform(-2, 2, (B, T, F)).astype(f32)
    Hs = rng.standard_normal((B, T, H)).astype(f32)
    c = f32(1.0 / (1.0 - p)) if p < 1.0 else f32(1.0)
    M = ((rng.random((4, B, F)) >= p) * c).astype(f32)
    ldt = (T + 127) // 128 * 128
    dX, dH, dM = dev.array(X), dev.array(Hs), dev.array(M)
    XS = dev.zeros((B, F, ldt))
    dev.call("mgr_transpose_bt_split", dX, F, XS, ldt, B, T, F)
    ws = dev.bytes(dev.lib.mgr_lstm_param_grads_dropout_ts_ws_bytes(B, T, F, H, ldt))
    wsr = dev.bytes(dev.lib.mgr_lstm_param_grads_ws_bytes(B, T, F, H))
    gate = np.arange(N) % 4
    base = (rng.standard_normal((B, T, N)) * 0.3).astype(f32)
    for spread in (False, True):
        dZ = base * (10.0 ** rng.uniform(-12, 12, size=(B, 1, N))).astype(f32) if spread else base
        ref = np.empty((F, N))
        for g in range(4):
            ref[:, gate == g] = np.einsum("btf,btn->fn", X.astype(np.float64) * M[g][:, None, :], dZ[:, :, gate == g].astype(np.float64))
        ddZ = dev.array(dZ)
        gW, gU, gb = dev.empty((F, N)), dev.empty((H, N)), dev.empty((N,))
        gW.upload(np.full((F, N), np.nan, f32))
        dev.call("mgr_memset", ws, 0xFF, ws.nbytes)          # the workspace arrives dirty
        dev.call("mgr_lstm_param_grads_dropout_ts", XS, ldt, dM, p if p < 0.99 else 0.5, dH, H, ddZ, gW, gU, gb, B, T, F, H, reverse, ws, ws.nbytes, 0, 0, 0, 0)
        # ... and with the row maxima handed in (what the BPTT leaves: mgr_scan_bwd_job.dzmax): the same bits
        zmx = dev.array(np.abs(dZ).max(axis=1).astype(f32).view(np.uint32))
        gW3 = dev.empty((F, N))
        dev.call("mgr_lstm_param_grads_dropout_ts", XS, ldt, dM, p if p < 0.99 else 0.5, dH, H, ddZ, gW3, gU, gb, B, T, F, H, reverse, ws, ws.nbytes, zmx, 0, 0, 0)
        assert np.array_equal(gW3.download(), gW.download())
        # ... and with the sums over time handed in (mgr_scan_bwd_job.dbsum): db is their sum over the samples, dW / dU the same bits
        zsm = dev.array(dZ.astype(np.float64).sum(axis=1).astype(f32))
        gU3, gb3 = dev.empty((H, N)), dev.empty((N,))
        dev.call("mgr_lstm_param_grads_dropout_ts", XS, ldt, dM, p if p < 0.99 else 0.5, dH, H, ddZ, gW3, gU3, gb3, B, T, F, H, reverse, ws, ws.nbytes, zmx, zsm, 0, 0)
        assert np.array_equal(gW3.download(), gW.download()) and np.array_equal(gU3.download(), gU.download())
        dbref = dZ.astype(np.float64).sum(axis=(0, 1))
        assert np.abs(gb3.download() - dbref).max() <= 2e-6 * np.abs(dZ).sum(axis=(0, 1)).max()
        # ... and with the kept lists of the PROJECTION of the same mask (proj_ws: the workspace mgr_lstm_input_proj_dropout_ts left
        # behind) instead of lists of its own: the same bits - with a dirty own workspace, so that nothing stale can be what it reads
        if p < 0.99:
            Wp_, bp_, Z_ = dev.array(rng.standard_normal((F, N)).astype(f32)), dev.zeros((N,)), dev.empty((B, T, N))
            pws = dev.bytes(dev.lib.mgr_lstm_input_proj_dropout_ts_ws_bytes(B, F, H))
            dev.call("mgr_memset", pws, 0xFF, pws.nbytes)
            dev.call("mgr_lstm_input_proj_dropout_ts", XS, ldt, dM, p, Wp_, bp_, Z_, B, T, F, H, pws, pws.nbytes)
            dev.call("mgr_memset", ws, 0xFF, ws.nbytes)
            gW4 = dev.empty((F, N))
            dev.call("mgr_lstm_param_grads_dropout_ts", XS, ldt, dM, p, dH, H, ddZ, gW4, gU3, gb3, B, T, F, H, reverse, ws, ws.nbytes, zmx, zsm, pws, 0)
            assert np.array_equal(gW4.download(), gW.download())
        # ... and dU formed like dW (HsT: the split rows of h_prev along time, mgr_transpose_bt_split_shift): against fp64, with the bound of
        # dW (per column: the spread of the gate gradients is per column); dW itself the same bits
        if H >= 16:
            HsT = dev.zeros((B, H, ldt))
            dev.call("mgr_transpose_bt_split_shift", dH, H, HsT, ldt, B, T, H, 1 if reverse else -1)
            hp = np.zeros_like(Hs, dtype=np.float64)
            if reverse:
                hp[:, :-1] = Hs[:, 1:]
            else:
                hp[:, 1:] = Hs[:, :-1]
            refU = np.einsum("btk,btn->kn", hp, dZ.astype(np.float64))
            gW5, gU5 = dev.empty((F, N)), dev.empty((H, N))
            gU5.upload(np.full((H, N), np.nan, f32))
            dev.call("mgr_memset", ws, 0xFF, ws.nbytes)
            dev.call("mgr_lstm_param_grads_dropout_ts", XS, ldt, dM, p if p < 0.99 else 0.5, dH, H, ddZ, gW5, gU5, gb3, B, T, F, H, reverse, ws, ws.nbytes, zmx, zsm, 0, HsT)
            assert np.array_equal(gW5.download(), gW.download())
            gotU = gU5.download()
            cs = np.maximum(np.abs(refU).max(axis=0, keepdims=True), 1e-30)
            assert np.all(np.isfinite(gotU)) and (np.abs(gotU - refU) / cs).max() <= 3e-5, (spread, (np.abs(gotU - refU) / cs).max())
            refU32 = gU.download()      # (the f32 split-K product: how far the two forms are apart, in the same measure)
            assert (np.abs(gotU - refU32) / cs).max() <= 6e-5
        got = gW.download()
        colscale = np.maximum(np.abs(ref).max(axis=0, keepdims=True), 1e-30)      # per column: the spread is per column
        assert np.all(np.isfinite(got)) and (np.abs(got - ref) / colscale).max() <= 3e-5, (spread, (np.abs(got - ref) / colscale).max())
        gW2, gU2, gb2 = dev.empty((F, N)), dev.empty((H, N)), dev.empty((N,))
        dev.call("mgr_lstm_param_grads", dX, F, dM, dH, H, ddZ, gW2, gU2, gb2, B, T, F, H, reverse, wsr, wsr.nbytes)
        assert np.array_equal(gU.download(), gU2.download()) and np.array_equal(gb.download(), gb2.download())


@pytest.mark.parametrize("hs,B,T", [((500, 300), 40, 21), ((300,), 64, 19), ((100,), 33, 24), ((500,), 16, 9)])
def test_pair_form_of_the_scan_is_bit_identical_and_scans_write_split_rows(device, hs, B, T):
    """lstm_cluster.hip, cluster_run_k16p (two 16-sample groups per workgroup, one workgroup per CU; tune key 4 = 2 forces it, 1
    forbids it): Y, gates, c and the transposed copies bit for bit those of the one-group form - B = 40 / 33 leave the last cluster
    with a single group, B = 16 has nothing to pair.  And mgr_scan_job.yt_split: the transposed copy in the split row format is, bit
    for bit, the split of the f32 copy (hi = rn_f16(y 2^13), lo = rn_f16(y 2^13 - hi)), zeros behind T."""
    from mgr_amd import _capi
    dev = device
    rng = np.random.default_rng(sum(hs) + B + T)
    ldt = (T + 127) // 128 * 128
    W = 2 * sum(hs)
    R = dev.array(rng.uniform(-1, 1, (B, T, W)).astype(f32))
    keep, base_jobs = [], []
    col = 0
    for H in hs:
        for d in range(2):
            Z = dev.array((rng.standard_normal((B, T, 4 * H)) * 0.5).astype(f32))
            U = dev.array((rng.standard_normal((H, 4 * H)) * 0.1 / np.sqrt(H)).astype(f32))
            Up = dev.empty((H, 4 * H))
            dev.call("mgr_lstm_pack", U, Up, H, H, 0)
            keep += [Z, U, Up]
            base_jobs.append(dict(Z=Z, Up=Up, H=H, reverse=d, col=col))
            col += H

    def run(pair, split):
        Y = dev.zeros((B, T, W))
        YT = dev.array(np.full((B, W, ldt), 7.0, f32))
        jobs = []
        outs = []
        for j in base_jobs:
            H, c0 = j["H"], j["col"]
            G, Cs = dev.zeros((B, T, H, 4)), dev.zeros((B, T, H))
            outs += [G, Cs]
            jobs.append(dict(Z=j["Z"], Up=j["Up"], Y=Y.view(c0, (1,)), ldy=W, R=R.view(c0, (1,)), ldr=W, gates=G, cs=Cs, B=B, T=T, H=H,
                             reverse=j["reverse"], YT=YT.ptr + c0 * ldt * 4, ytb=W * ldt, ldt=ldt, yt_split=int(split)))
        dev.call("mgr_tune", 0, 3)      # clusters with an exchange at every H (the K-split step)
        dev.call("mgr_tune", 1, 1)
        dev.call("mgr_tune", 4, 2 if pair else 1)
        try:
            arr = _capi.make_scan_jobs(jobs)
            ws = dev.bytes(dev.lib.mgr_lstm_scan_multi_ws_bytes(len(jobs), arr))
            _capi.check(dev.lib.mgr_lstm_scan_fwd_multi(dev.ctx, len(jobs), arr, ws.ptr, ws.nbytes))
        finally:
            dev.call("mgr_tune", 0, 0)
            dev.call("mgr_tune", 1, 0)
            dev.call("mgr_tune", 4, 0)
        return Y.download(), YT.download(), [o.download() for o in outs]

    y0, yt0, o0 = run(False, False)
    y1, yt1, o1 = run(True, False)
    assert np.array_equal(y0, y1) and np.array_equal(yt0, yt1) and all(np.array_equal(a, b) for a, b in zip(o0, o1))
    assert np.array_equal(yt0[:, :, :T], y0.transpose(0, 2, 1)) and not yt0[:, :, T:].any()
    for pair in (False, True):
        y2, yts, o2 = run(pair, True)
        assert np.array_equal(y2, y0) and all(np.array_equal(a, b) for a, b in zip(o0, o2))
        assert np.array_equal(yts.view(np.uint32), _split_rows(y0, ldt).view(np.uint32))


@pytest.mark.parametrize("B,T", [(64, 11), (55, 7)])
def test_fused_form_of_the_encoder_scans_is_bit_identical(device, B, T):
    """lstm_cluster.hip, k_scan_cluster_k16f (tune key 4 = 3): 8-wave workgroups that run TWO unit groups of their cluster, one workgroup
    per CU - taken only by launches that do not fit one workgroup per CU as they are, i.e. the encoder depths of config F (audio H = 500 +
    skeletal H = 300, both directions, B = 64: 408 workgroups -> 208).  Y, gates, c and the split transposed copies bit for bit those of
    the default form; skeletal clusters have an odd number of unit groups (19): the last workgroup's second half only keeps the barriers."""
    from mgr_amd import _capi
    dev = device
    hs = (500, 300)
    rng = np.random.default_rng(B + T)
    ldt = (T + 127) // 128 * 128
    W = 2 * sum(hs)
    R = dev.array(rng.uniform(-1, 1, (B, T, W)).astype(f32))
    keep, base_jobs = [], []
    col = 0
    for H in hs:
        for d in range(2):
            Z = dev.array((rng.standard_normal((B, T, 4 * H)) * 0.5).astype(f32))
            U = dev.array((rng.standard_normal((H, 4 * H)) * 0.1 / np.sqrt(H)).astype(f32))
            Up = dev.empty((H, 4 * H))
            dev.call("mgr_lstm_pack", U, Up, H, H, 0)
            keep += [Z, U, Up]
            base_jobs.append(dict(Z=Z, Up=Up, H=H, reverse=d, col=col))
            col += H

    def run(fused, own_gather=0):
        Y = dev.zeros((B, T, W))
        YT = dev.array(np.full((B, W, ldt), 7.0, f32))
        jobs, outs = [], []
        for j in base_jobs:
            H, c0 = j["H"], j["col"]
            G, Cs = dev.zeros((B, T, H, 4)), dev.zeros((B, T, H))
            outs += [G, Cs]
            jobs.append(dict(Z=j["Z"], Up=j["Up"], Y=Y.view(c0, (1,)), ldy=W, R=R.view(c0, (1,)), ldr=W, gates=G, cs=Cs, B=B, T=T, H=H,
                             reverse=j["reverse"], YT=YT.ptr + c0 * ldt * 4, ytb=W * ldt, ldt=ldt, yt_split=1))
        dev.call("mgr_tune", 1, 1)
        dev.call("mgr_tune", 4, 3 if fused else 0)
        dev.call("mgr_tune", 17, own_gather)
        try:
            arr = _capi.make_scan_jobs(jobs)
            ws = dev.bytes(dev.lib.mgr_lstm_scan_multi_ws_bytes(len(jobs), arr))
            _capi.check(dev.lib.mgr_lstm_scan_fwd_multi(dev.ctx, len(jobs), arr, ws.ptr, ws.nbytes))
        finally:
            dev.call("mgr_tune", 1, 0)
            dev.call("mgr_tune", 4, 0)
            dev.call("mgr_tune", 17, 0)
        return Y.download(), YT.download(), [o.download() for o in outs]

    y0, yt0, o0 = run(False)
    assert np.isfinite(y0).all() and np.abs(y0).max() > 0.1
    # round 6: the two halves of a fused workgroup SHARE one gather of the h image through LDS (k_scan_cluster_k16fs, the default);
    # tune key 17 = 1: each half fetches the whole image itself (round 5's k_scan_cluster_k16f) - all three the same bits
    for own_gather in (0, 1):
        y1, yt1, o1 = run(True, own_gather)
        assert np.array_equal(y0, y1) and np.array_equal(yt0.view(np.uint32), yt1.view(np.uint32)), own_gather
        assert all(np.array_equal(a, b) for a, b in zip(o0, o1)), own_gather


@pytest.mark.parametrize("H,B,T", [(100, 64, 40), (128, 20, 9), (64, 33, 17), (100, 16, 1)])
def test_fused_form_of_a_narrow_layer_by_launch_option(device, H, B, T):
    """Round 6: MGR_SCAN_FORM_FUSED_ANY as an argument of the launch (mgr_scan_launch_opts) gives a NARROW layer the fused form too - the
    fusion layer of config F: H = 100, 7 unit groups -> 4 eight-wave workgroups per cluster (the last one's second half only keeps the
    barriers), 32 workgroups that hold a CU each instead of 56 four-wave ones.  Y, gates and c bit for bit those of the plain form; the
    launch reports its number."""
    import ctypes
    from mgr_amd import _capi
    dev = device
    rng = np.random.default_rng(H + B + T)
    keep, jobs_of = [], []
    Zs, Ups = [], []
    for d in range(2):
        Z = dev.array((rng.standard_normal((B, T, 4 * H)) * 0.5).astype(f32))
        U = dev.array((rng.standard_normal((H, 4 * H)) / np.sqrt(H)).astype(f32))
        Up = dev.empty((H, 4 * H))
        dev.call("mgr_lstm_pack", U, Up, H, H, 0)
        Zs.append(Z)
        Ups.append(Up)
        keep += [Z, U, Up]

    def run(form):
        Y = dev.zeros((B, T, 2 * H))
        outs, jobs = [], []
        for d in range(2):
            G, Cs = dev.zeros((B, T, H, 4)), dev.zeros((B, T, H))
            outs += [G, Cs]
            jobs.append(dict(Z=Zs[d], Up=Ups[d], Y=Y.view(d * H, (1,)), ldy=2 * H, gates=G, cs=Cs, B=B, T=T, H=H, reverse=d))
        arr = _capi.make_scan_jobs(jobs)
        ws = dev.bytes(dev.lib.mgr_lstm_scan_multi_ws_bytes(len(jobs), arr))
        seq = ctypes.c_uint(0)
        opts = _capi.make_launch_opts(form, ctypes.addressof(seq))
        dev.call("mgr_tune", 1, 1)
        try:
            _capi.check(dev.lib.mgr_lstm_scan_fwd_multi_ex(dev.ctx, len(jobs), arr, ws.ptr, ws.nbytes, ctypes.byref(opts)))
        finally:
            dev.call("mgr_tune", 1, 0)
        n = ctypes.c_int()
        dev.call("mgr_persist_stats", ctypes.byref(n), None)
        # (the launch number the call reports is the context's newest - or none: a layer the planner keeps on ONE CU per batch group
        #  has no exchange and enters no launch into the residency ledger)
        assert seq.value in (n.value, _capi.SEQ_NONE)
        res = (Y.download(), [o.download() for o in outs])
        for a in [Y, ws] + outs:
            a.free()
        return res

    y0, o0 = run(_capi.SCAN_FORM_PLAIN)
    y1, o1 = run(_capi.SCAN_FORM_FUSED_ANY)
    y2, o2 = run(_capi.SCAN_FORM_FUSED)          # (fits one workgroup per CU as it is: FUSED leaves it plain)
    assert np.isfinite(y0).all() and np.abs(y0).max() > 0.05
    for y, o in ((y1, o1), (y2, o2)):
        assert np.array_equal(y0, y)
        assert all(np.array_equal(a, b) for a, b in zip(o0, o))
    for a in keep:
        a.free()


def test_frozen_weight_planes_are_kept_and_dropped(device):
    """mgr_weight_planes_cache (round 6): for weights the caller declares frozen, mgr_lstm_input_proj_dropout_ts keeps the (hi, lo) planes
    and the largest |W| it left in its workspace - the same bits as a call that rebuilds them; weights rewritten BEHIND the library's
    back show that the planes really were reused (the result is the old weights'), and declaring them again drops the planes."""
    dev = device
    B, T, F, H, p = 2, 150, 600, 300, 0.5
    rng = np.random.default_rng(4)
    N = 4 * H
    X = rng.uniform(-2, 2, (B, T, F)).astype(f32)
    W1 = (rng.standard_normal((F, N)) * 0.1).astype(f32)
    W2 = (rng.standard_normal((F, N)) * 0.3).astype(f32)
    bias = rng.standard_normal(N).astype(f32)
    M = ((rng.random((4, B, F)) >= p) * f32(2.0)).astype(f32)
    ldt = (T + 127) // 128 * 128
    dX, dW, db, dM = dev.array(X), dev.array(W1), dev.array(bias), dev.array(M)
    XS = dev.zeros((B, F, ldt))
    dev.call("mgr_transpose_bt_split", dX, F, XS, ldt, B, T, F)
    ws = dev.bytes(dev.lib.mgr_lstm_input_proj_dropout_ts_ws_bytes(B, F, H))
    Z = dev.empty((B, T, N))

    def proj(mask=dM):
        Z.upload(np.full((B, T, N), np.nan, f32))
        dev.call("mgr_lstm_input_proj_dropout_ts", XS, ldt, mask, p, dW, db, Z, B, T, F, H, ws, ws.nbytes)
        return Z.download()

    z1 = proj()                                       # not frozen: planes rebuilt by every call
    dev.call("mgr_weight_planes_cache", dW, 1)
    assert np.array_equal(proj(), z1)                 # builds and keeps the planes
    assert np.array_equal(proj(), z1)                 # ... reuses them: the same bits
    M2 = ((rng.random((4, B, F)) >= p) * f32(2.0)).astype(f32)
    zm = proj(dev.array(M2))                          # another mask with kept planes (the mask factor word is reset per call)
    dev.call("mgr_weight_planes_cache", dW, 0)
    assert np.array_equal(proj(dev.array(M2)), zm)
    dev.call("mgr_weight_planes_cache", dW, 1)
    assert np.array_equal(proj(), z1)
    # the workspace is shared with the f32-row entry points (inference passes of the same layer): a call of one of them overwrites
    # the kept planes - the library must notice (it did not at first: a predict between two training steps poisoned the next step)
    XT = dev.zeros((B, F, ldt))
    dev.call("mgr_transpose_bt", dX, F, XT, ldt, B, T, F)
    Zt = dev.empty((B, T, N))
    wst = dev.lib.mgr_lstm_input_proj_dropout_ws_bytes(B, F, H)
    assert wst <= ws.nbytes
    dev.call("mgr_lstm_input_proj_dropout_t", XT, ldt, dM, p, dW, db, Zt, B, T, F, H, ws, ws.nbytes, 2.0)
    assert np.array_equal(proj(), z1)
    dW.upload(W2)                                     # rewritten behind the promise: the kept planes are still W1's
    assert np.array_equal(proj(), z1)
    dev.call("mgr_weight_planes_cache", dW, 1)        # declared again = "rewritten": dropped, rebuilt from W2
    z2 = proj()
    assert not np.array_equal(z2, z1)
    dev.call("mgr_weight_planes_cache", dW, 0)
    assert np.array_equal(proj(), z2)
    for a in (dX, dW, db, dM, XS, ws, Z):
        a.free()
