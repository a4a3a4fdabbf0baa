#!/usr/bin/env python3
"""Where a small configuration's step goes on the HOST: wall time of enqueue_train_step / read_loss per step and the GPU time of the
same steps (events), for bench.py's configurations.   python tools/host_time_probe.py [S] [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mgr_amd  # noqa: F401
from mgr_amd import _capi
from mgr_amd.configs import baseline_config
from mgr_amd.engine import Engine
from mgr_amd.synthetic import synthetic_arrays, synthetic_weights

cfg = sys.argv[1] if len(sys.argv) > 1 else "S"
K = int(sys.argv[2]) if len(sys.argv) > 2 else 20
spec, B, T, Lmax = baseline_config(cfg)
dev = _capi.Device(0)
eng = Engine(spec, B, T, Lmax, device=dev, seed=1)
eng.set_weights(synthetic_weights(spec, 3))
xs, labels, il, ll = synthetic_arrays(spec, B, T, Lmax, 5)
eng._upload_inputs(xs, None, True)
eng._upload_labels(labels, il, ll)
for _ in range(3):
    eng.enqueue_train_step(None, None, None, None, upload=False)
    eng.read_loss()
dev.sync()
calls = [0]
orig = dev.call
def counted(*a, **k):
    calls[0] += 1
    return orig(*a, **k)
dev.call = counted
te = tr = 0.0
slow = {}
def timed(name, fn):
    def w(*a, **k):
        t = time.perf_counter()
        r = fn(*a, **k)
        d = time.perf_counter() - t
        if d > slow.get(name, (0,))[0]:
            slow[name] = (d, a[0] if a and isinstance(a[0], str) else "")
        return r
    return w
dev.call = timed("call", dev.call)
type(eng.loss_mean).download = timed("download", type(eng.loss_mean).download)
eng.scan_health = timed("scan_health", eng.scan_health)
steps = []
t00 = time.perf_counter()
for _ in range(K):
    t0 = time.perf_counter()
    eng.enqueue_train_step(None, None, None, None, upload=False)
    t1 = time.perf_counter()
    eng.read_loss()
    t2 = time.perf_counter()
    te += t1 - t0
    tr += t2 - t1
    steps.append((t2 - t0, t1 - t0, t2 - t1))
dev.sync()
tt = time.perf_counter() - t00
w = max(steps)
print("slowest step: %.2f ms (enqueue %.2f, read_loss %.2f); slowest single calls: %s"
      % (w[0] * 1e3, w[1] * 1e3, w[2] * 1e3, {k: (round(v[0] * 1e3, 2), v[1]) for k, v in slow.items()}))
print("config %s  B=%d T=%d: %.2f ms/step wall | host: enqueue %.2f ms, read_loss (blocks until the loss is there) %.2f ms | %d C-ABI calls per step"
      % (cfg, B, T, tt / K * 1e3, te / K * 1e3, tr / K * 1e3, calls[0] // K))
