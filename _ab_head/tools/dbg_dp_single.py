"""Debug: the dp lock-step scenario in ONE process (B = 16), repeated: NaN / run-to-run differences in the gradients?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mgr_amd
import numpy as np
from mgr_amd import _capi
from mgr_amd.engine import Engine, Schedule
from mgr_amd.synthetic import synthetic_weights
from tests import dp_worker
dev = _capi.Device(0)
spec = dp_worker.dp_spec(False)
B, T, Lmax, steps = 16, 96, 8, 5
ref = None
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 12):
    eng = Engine(spec, B, T, Lmax, device=dev, seed=100)
    eng.set_weights(synthetic_weights(spec, 3))
    batches = [dp_worker.__dict__["shard_batch"](b, 0, 2) if False else {k: v[:B] for k, v in b.items()} for b in dp_worker.dp_batches(spec, 32, T, Lmax, 2)]
    losses = dp_worker.run_steps(eng, spec, batches, steps)
    dev.sync()
    g = eng.get_grads()
    nan = {k: int(np.isnan(v).sum()) for k, v in g.items() if np.isnan(v).any()}
    same = ref is None or all(np.array_equal(g[k], ref[k], equal_nan=True) for k in g)
    if ref is None:
        ref = g
    print("rep", rep, "losses", ["%.4f" % l for l in losses], "nan", nan, "same as first", same, flush=True)
    eng.close()
