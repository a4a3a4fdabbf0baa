"""Debug aid for tests/test_gpu_schedule_contract.py::test_drop_paths_...: which drop path leaves the plain schedule's losses, and where."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import mgr_amd
from mgr_amd import _capi
from mgr_amd.configs import baseline_config
from mgr_amd.engine import Engine, Schedule
from mgr_amd.synthetic import synthetic_arrays, synthetic_weights
dev = _capi.Device(0)
spec, _, _, Lmax = baseline_config("F")
B, T = 16, 256
data = [synthetic_arrays(spec, B, T, Lmax, 100 + k) for k in range(4)]
order = [0, 1, 2, 3, 0, 2, 1, 3]


def run(pipelined, a, b, **sched):
    eng = Engine(spec, B, T, Lmax, device=dev, seed=21, schedule=Schedule(pipeline=pipelined, **sched))
    eng.set_weights(synthetic_weights(spec, 3))
    losses = []
    for i, k in enumerate(order):
        xs, lab, il, ll = data[k]
        nxt = data[order[i + 1]][0] if i + 1 < len(order) else None
        nxt2 = data[order[i + 2]][0] if i + 2 < len(order) else None
        if pipelined and b and i == 2:
            nxt, nxt2 = data[3][0], data[1][0]
        losses.append(eng.train_step(xs, lab, il, ll, next_inputs=nxt if pipelined else None, after_next_inputs=nxt2 if pipelined else None))
        if a and i == 5:
            eng.predict(data[0][0])
    dev.sync()
    eng.close()
    return losses


plain = run(False, False, False)
for a, b in ((False, False), (True, False), (False, True), (True, True)):
    for sched in ({}, dict(first_pass_on_encoder_stream=False), dict(fused_encoder_scans=False)):
        got = run(True, a, b, **sched)
        print("predict=%s other_batch=%s" % (a, b), sched, "OK" if got == plain else "DIFF at %s" % [i for i, (x, y) in enumerate(zip(got, plain)) if x != y], flush=True)
