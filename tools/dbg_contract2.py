"""Debug aid: does state left by the validation scenario change the first-step loss of a later engine?"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np
import mgr_amd
from mgr_amd import _capi
from mgr_amd.configs import baseline_config
from mgr_amd.engine import Engine, Schedule
from mgr_amd.synthetic import synthetic_arrays, synthetic_weights
import test_gpu_schedule_contract as tc
dev = _capi.Device(0)
spec, _, _, Lmax = baseline_config("F")
B, T = 16, 256
data = [synthetic_arrays(spec, B, T, Lmax, 100 + k) for k in range(4)]


def first(pipelined, **sched):
    eng = Engine(spec, B, T, Lmax, device=dev, seed=21, schedule=Schedule(pipeline=pipelined, **sched))
    eng.set_weights(synthetic_weights(spec, 3))
    xs, lab, il, ll = data[0]
    l = eng.train_step(xs, lab, il, ll, next_inputs=data[1][0] if pipelined else None, after_next_inputs=data[2][0] if pipelined else None)
    dev.sync()
    eng.close()
    return l


print("fresh process:", first(False), first(False), first(True), first(True), first(True, first_pass_on_encoder_stream=False), flush=True)
what = sys.argv[1] if len(sys.argv) > 1 else "validation"
if what == "validation":
    tc.test_validation_pass_and_second_engine_between_pipelined_steps(dev)
elif what == "fusion_run":
    tc._fusion_run(dev, 64, 160, 7, {})
elif what == "fusion_run_plain":
    tc._fusion_run(dev, 64, 160, 7, dict(pipeline=False))
print("after %s:" % what, first(False), first(False), first(True), first(True), first(True, first_pass_on_encoder_stream=False), first(True, fused_encoder_scans=False), flush=True)
