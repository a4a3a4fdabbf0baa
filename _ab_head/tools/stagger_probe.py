#!/usr/bin/env python3
"""Probe: the depth-2 part of config F's cycle in two orders (raw library calls, no engine).
  A (shipped):   ES: 4 projection GEMMs (audio f/r, skeletal f/r)  ->  ONE launch of the 4 depth-2 scans
                 S0: fusion BPTT beside the GEMMs, then the fusion dW / dU / db beside the scans
  B (staggered): ES: skeletal GEMMs -> audio GEMMs -> audio scans (2 jobs);   S3: skeletal scans as soon as their GEMMs are done
                 (beside the audio GEMMs);  S0 as in A
Reports the time from the first kernel to the last one."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mgr_amd  # noqa: E402,F401
from mgr_amd import _capi  # noqa: E402

dev = _capi.Device(0); lib = dev.lib
rng = np.random.default_rng(0)
B, T = 64, 1900
ldt = (T + 127) // 128 * 128
f32 = np.float32


def layer(F, H):
    d = {"F": F, "H": H}
    d["XT"] = dev.array(rng.standard_normal((B, F, ldt)).astype(f32) * 0.3)
    d["dir"] = []
    for rev in (0, 1):
        W = dev.array((rng.standard_normal((F, 4 * H)) * 0.03).astype(f32)); b = dev.zeros((4 * H,))
        m = dev.array(((rng.random((4, B, F)) > 0.5) * 2.0).astype(f32))
        Z = dev.empty((B, T, 4 * H))
        U = dev.array((rng.standard_normal((H, 4 * H)) * 0.05).astype(f32)); Up = dev.empty((H, 4 * H))
        dev.call("mgr_lstm_pack", U, Up, H, H, 0)
        ws = dev.bytes(lib.mgr_lstm_input_proj_dropout_ws_bytes(B, F, H))
        d["dir"].append(dict(W=W, b=b, m=m, Z=Z, Up=Up, ws=ws, rev=rev))
    d["Y"] = dev.empty((B, T, 2 * H))
    return d


audio, skel = layer(1000, 500), layer(600, 300)


def proj(L):
    for D in L["dir"]:
        dev.call("mgr_lstm_input_proj_dropout_t", L["XT"], ldt, D["m"], 0.5, D["W"], D["b"], D["Z"], B, T, L["F"], L["H"], D["ws"], D["ws"].nbytes, 2.0)


def scan_jobs(layers):
    jobs = []
    for L in layers:
        for D in L["dir"]:
            jobs.append(dict(Z=D["Z"], Up=D["Up"], Y=L["Y"].view(D["rev"] * L["H"], (1,)), ldy=2 * L["H"], B=B, T=T, H=L["H"], reverse=D["rev"]))
    arr = _capi.make_scan_jobs(jobs)
    ws = dev.bytes(lib.mgr_lstm_scan_multi_ws_bytes(len(jobs), arr))
    return lambda: _capi.check(lib.mgr_lstm_scan_fwd_multi(dev.ctx, len(jobs), arr, ws.ptr, ws.nbytes))


scan_all, scan_audio, scan_skel = scan_jobs([audio, skel]), scan_jobs([audio]), scan_jobs([skel])

# fusion layer: BPTT (H = 100) and dW / dU / db
Fw, Hf = 1600, 100
featT = dev.array(rng.standard_normal((B, Fw, ldt)).astype(f32) * 0.3)
YF = dev.array(rng.standard_normal((B, T, 2 * Hf)).astype(f32) * 0.3)
fus = []
for rev in (0, 1):
    g = dev.array(rng.random((B, T, Hf, 4)).astype(f32)); cs = dev.array(rng.standard_normal((B, T, Hf)).astype(f32) * 0.3)
    U = dev.array((rng.standard_normal((Hf, 4 * Hf)) * 0.05).astype(f32)); Up = dev.empty((Hf, 4 * Hf)); dev.call("mgr_lstm_pack", U, Up, Hf, Hf, 0)
    dY = dev.array(rng.standard_normal((B, T, 2 * Hf)).astype(f32) * 0.1)
    dZ = dev.empty((B, T, 4 * Hf))
    m = dev.array(((rng.random((4, B, Fw)) > 0.5) * 2.0).astype(f32))
    gW, gU, gb = dev.empty((Fw, 4 * Hf)), dev.empty((Hf, 4 * Hf)), dev.empty((4 * Hf,))
    ws = dev.bytes(lib.mgr_lstm_param_grads_dropout_t_ws_bytes(B, T, Fw, Hf, ldt))
    fus.append(dict(g=g, cs=cs, Up=Up, dY=dY, dZ=dZ, m=m, gW=gW, gU=gU, gb=gb, ws=ws, rev=rev))
bj = _capi.make_scan_bwd_jobs([dict(dY=D["dY"].view(D["rev"] * Hf, (1,)), gates=D["g"], cs=D["cs"], Up=D["Up"], dZ=D["dZ"], lddy=2 * Hf, B=B, T=T, H=Hf, reverse=D["rev"]) for D in fus])
bws = dev.bytes(lib.mgr_lstm_scan_bwd_multi_ws_bytes(2, bj))


def bptt():
    _capi.check(lib.mgr_lstm_scan_bwd_multi(dev.ctx, 2, bj, bws.ptr, bws.nbytes))


def dw():
    for D in fus:
        dev.call("mgr_lstm_param_grads_dropout_t", featT, ldt, D["m"], 0.5, YF.view(D["rev"] * Hf, (1,)), 2 * Hf, D["dZ"], D["gW"], D["gU"], D["gb"],
                 B, T, Fw, Hf, D["rev"], D["ws"], D["ws"].nbytes, 2.0)


ES, S0, S3 = 1, 2, 3
EV_PROJ, EV_SK = 8, 9


def order_a():
    dev.stream(ES); dev.record(0)
    dev.stream(S0); bptt()
    dev.stream(ES); proj(audio); proj(skel); dev.record(EV_PROJ)
    scan_all(); dev.record(1)
    dev.stream(S0); dev.wait_event(S0, EV_PROJ); dw(); dev.record(2)
    return [(0, 1), (0, 2)]


def order_b():
    dev.stream(ES); dev.record(0)
    dev.stream(S0); bptt()
    dev.stream(ES); proj(skel); dev.record(EV_SK)
    dev.stream(S3); dev.wait_event(S3, EV_SK); scan_skel(); dev.record(3)
    dev.stream(ES); proj(audio); dev.record(EV_PROJ)
    scan_audio(); dev.record(1)
    dev.stream(S0); dev.wait_event(S0, EV_PROJ); dw(); dev.record(2)
    return [(0, 1), (0, 2), (0, 3)]


for name, fn in (("A shipped order", order_a), ("B staggered", order_b), ("A shipped order", order_a), ("B staggered", order_b)):
    res = []
    for rep in range(4):
        dev.sync()
        pairs = fn()
        dev.stream(0); dev.sync()
        res.append([dev.elapsed_ms(a, b) for a, b in pairs])
    best = min(res[1:], key=lambda r: max(r))
    print("%-18s end of: scans(ES) %6.2f  dW %6.2f %s  -> segment %.2f ms" % (name, best[0], best[1], ("skeletal scans %6.2f" % best[2]) if len(best) > 2 else "", max(best)), flush=True)
st = __import__("ctypes").c_uint(0)
dev.call("mgr_scan_status", __import__("ctypes").byref(st)); print("scan status", st.value)
