"""Two identical short trainings must produce bit-identical losses (deterministic kernels); also a random-network sweep at
feature counts that take the dropout-aware projection."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import mgr_amd
from mgr_amd import _capi
from mgr_amd.configs import baseline_config
from mgr_amd.engine import Engine
from mgr_amd.synthetic import synthetic_arrays, synthetic_weights
dev = _capi.Device(0)
spec, B, T, Lmax = baseline_config("F")
B, T = 16, 400
def run():
    eng = Engine(spec, B, T, Lmax, device=dev, seed=7)
    eng.set_weights(synthetic_weights(spec, 3))
    xs, labels, il, ll = synthetic_arrays(spec, B, T, Lmax, 11)
    eng._upload_inputs(xs, None, True)
    eng._upload_labels(labels, il, ll)
    out = []
    for i in range(12):
        eng.enqueue_train_step(None, None, None, None, rand=None, apply_update=True, upload=False, prefetch_next=(i < 11))
        out.append(eng.read_loss())
    eng.close()
    return np.array(out)
a, b = run(), run()
print("losses", a[:4], "identical:", np.array_equal(a, b))
