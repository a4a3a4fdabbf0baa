"""Start-up transient of the pipelined training schedule: host-side ms per step of the first steps of a run, the persistent-launch
ledger after each (launches, serialised) and the residency waits.  python tools/startup_probe.py [--inline] [--pg2]"""
import sys, time, ctypes
import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import mgr_amd
from mgr_amd import _capi
from mgr_amd.configs import baseline_config
from mgr_amd.engine import Engine, Schedule
from mgr_amd.synthetic import synthetic_arrays, synthetic_weights
spec, B, T, Lmax = baseline_config("F")
dev = _capi.Device(0)
eng = Engine(spec, B, T, Lmax, device=dev, seed=1000, schedule=Schedule(first_pass_on_encoder_stream="--inline" not in sys.argv, param_grads_two_streams="--pg2" in sys.argv))
eng.set_weights(synthetic_weights(spec, 20131903))
xs, labels, il, ll = synthetic_arrays(spec, B, T, Lmax, 20131903)
eng._upload_inputs(xs, None, True); eng._upload_labels(labels, il, ll); dev.sync()
def stats():
    a, b = ctypes.c_int(), ctypes.c_int()
    dev.call("mgr_persist_stats", ctypes.byref(a), ctypes.byref(b)); return a.value, b.value
for region in range(2):
    n = 8
    dev.sync(); t0 = time.perf_counter(); marks = [t0]; ss = [stats()]; losses = []
    for i in range(n):
        eng.enqueue_train_step(None, None, None, None, rand=None, apply_update=True, upload=False, prefetch_next=i + 1 < n, prefetch_after_next=i + 2 < n)
        losses.append(eng.read_loss(local=True)); marks.append(time.perf_counter()); ss.append(stats())
    dev.sync(); tend = time.perf_counter()
    print("region", region, "steps ms:", [round((b - a) * 1e3, 2) for a, b in zip(marks, marks[1:])], "tail", round((tend - marks[-1]) * 1e3, 2), "total/step", round((tend - t0) / n * 1e3, 3))
    print("   losses", losses)
    print("   persist (launches, serialised) after each step:", ss, "waits", eng.resident_wait_stats())
    out = (ctypes.c_uint * 4)(); dev.call("mgr_resident_wait_stats", out)
    base = eng._seq_words.ctypes.data
    print("   last expired wait: seq", out[2], "word index", ((out[3] - (base & 0xFFFFFFFF)) & 0xFFFFFFFF) // 4, "| gate log (step, kind, word indices):", [(a, b, tuple(None if w is None else (w - base) // 4 for w in ws)) for a, b, ws in list(eng._gate_log)[-8:]], "| words now", list(eng._seq_words))
