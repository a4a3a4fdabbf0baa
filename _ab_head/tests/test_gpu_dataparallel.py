"""-m gpu: a REAL world-size-2 data-parallel training step (BASELINE configs[3]), in two variants: comm=host on the one GPU
of the test box, comm=rccl on two GPUs (skipped where there is one: it enables itself on the first multi-GPU box).

Two rank processes share GPU 0; each owns half of the global batch and its own Engine with the pipelined two-stream schedule
on; gradients are summed by parallel.HostComm (RCCL refuses two ranks on one device).  Everything else is the path the 8-GPU
bench runs: shard_batch, Engine.apply_gradients' world > 1 branch (all-reduce -> /world -> clip -> Adam -> max-norm), the deferred
dW -> all-reduce -> Adam ordering of the pipelined schedule.  The rank processes are forked from a fork server that
tests/conftest.py starts before anything touches the GPU (a process that has initialised the GPU must not exec)."""
import multiprocessing as mp
import os
import socket

import numpy as np
import pytest

from tests import dp_worker
from tests.helpers import rel_err

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _need(comm_kind, world):
    """RCCL refuses two ranks on one device: the rccl variants need `world` GPUs and enable themselves on a box that has them."""
    if comm_kind == "rccl":
        from mgr_amd._capi import device_count
        if device_count() < world:
            pytest.skip("comm=rccl needs %d GPUs (this box has %d): RCCL refuses two ranks on one device" % (world, device_count()))


def _run_ranks(tmp_path, world, exact, B, T, Lmax, steps, comm_kind="host", target=None, args=None):
    from multiprocessing import forkserver
    if getattr(forkserver._forkserver, "_forkserver_pid", None) is None:
        pytest.skip("the fork server must be started before the GPU is initialised: run through `pytest -m gpu` (tests/conftest.py)")
    ctx = mp.get_context("forkserver")
    port = _free_port()
    outs = [str(tmp_path / ("rank%d.npz" % r)) for r in range(world)]
    if target is None:
        procs = [ctx.Process(target=dp_worker.dp_rank, args=(r, world, port, outs[r], exact, B, T, Lmax, steps, comm_kind)) for r in range(world)]
    else:
        procs = [ctx.Process(target=target, args=(r, world, port, outs[r]) + tuple(args or ())) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(600)
    for r, p in enumerate(procs):
        if p.is_alive():
            p.kill()
            pytest.fail("rank %d did not finish" % r)
        err = outs[r] + ".err"
        assert p.exitcode == 0, open(err).read() if os.path.exists(err) else "rank %d exit code %s" % (r, p.exitcode)
    return [np.load(o) for o in outs]



@pytest.mark.parametrize("comm_kind", ["host", "rccl"])
def test_two_ranks_equal_one_process_on_the_full_batch(device, tmp_path, comm_kind):
    """comm=host: two ranks share GPU 0 (HostComm); comm=rccl: one GPU per rank, ncclAllReduce over xGMI - the same equalities.
    No random draws (dropout / noise 0): 4 pipelined steps of 2 ranks x B/2 must reproduce 1 process x B - per-step losses,
    the all-reduced gradient, the updated weights - up to fp32 summation order; the two replicas must agree bit for bit."""
    from mgr_amd.engine import Engine
    from mgr_amd.synthetic import synthetic_weights
    B, T, Lmax, steps, world = 32, 96, 8, 4, 2
    _need(comm_kind, world)
    ranks = _run_ranks(tmp_path, world, True, B, T, Lmax, steps, comm_kind)
    # the same steps in one process on the full batch (this process, same GPU)
    spec = dp_worker.dp_spec(True)
    eng = Engine(spec, B, T, Lmax, device=device, seed=100)
    eng.set_weights(synthetic_weights(spec, 3))
    ref_losses = dp_worker.run_steps(eng, spec, dp_worker.dp_batches(spec, B, T, Lmax, 2), steps)
    device.sync()
    w_ref, g_ref = eng.get_weights(), eng.get_grads()
    eng.close()
    r0, r1 = ranks
    assert int(r0["status"]) == 0 and int(r1["status"]) == 0            # no persistent scan gave up, nothing went non-finite
    # losses: every rank reports the GLOBAL mean (it rides on the gradient all-reduce) = the mean of the ranks' local
    # means = the one-process loss on the full batch
    assert np.array_equal(r0["losses"], r1["losses"])
    glob = (r0["local"] + r1["local"]) / world
    assert np.allclose(r0["losses"], glob, rtol=1e-6), (r0["losses"], glob)
    assert not np.array_equal(r0["local"], r1["local"])
    assert np.allclose(glob, ref_losses, rtol=2e-6), (glob, ref_losses)
    assert np.all(np.isfinite(glob)) and len(set(np.round(glob, 3))) > 1
    for k, v in w_ref.items():
        kk = k.replace("/", "__")
        assert np.array_equal(r0["w__" + kk], r1["w__" + kk]), k          # replicas stay bit-identical
        # Adam's step is lr * m / (sqrt(v) + eps): where a gradient is ~0 a summation-order difference moves it by a fraction of lr
        assert np.allclose(r0["w__" + kk], v, rtol=0, atol=5e-5), (k, np.abs(r0["w__" + kk] - v).max())
    for k, v in g_ref.items():
        kk = k.replace("/", "__")
        assert np.array_equal(r0["g__" + kk], r1["g__" + kk]), k
        # the gradient buffer holds the all-reduced SUM of the ranks' local means = world x the full-batch mean gradient
        assert rel_err(r0["g__" + kk] / world, v) < 2e-5, (k, rel_err(r0["g__" + kk] / world, v))
    # weights moved by about lr per step where gradients are non-trivial
    assert np.abs(w_ref["dense/W"] - synthetic_weights(spec, 3)["dense/W"]).max() > 1e-4


@pytest.mark.parametrize("comm_kind", ["host", "rccl"])
def test_two_ranks_with_device_rng_stay_in_lockstep(device, tmp_path, comm_kind):
    """The real configuration (dropout .4-.6, noise .5, per-rank RNG seeds): replicas must still hold identical weights after
    5 pipelined steps, losses finite and different between the ranks (different shards, different masks)."""
    _need(comm_kind, 2)
    ranks = _run_ranks(tmp_path, 2, False, 32, 96, 8, 5, comm_kind)
    r0, r1 = ranks
    assert int(r0["status"]) == 0 and int(r1["status"]) == 0
    assert np.all(np.isfinite(r0["losses"])) and np.all(np.isfinite(r1["losses"]))
    assert np.array_equal(r0["losses"], r1["losses"])            # the reported loss is the global one on every rank ...
    assert not np.array_equal(r0["local"], r1["local"])          # ... the local means differ (different shards, different masks)
    for k in r0.files:
        if k.startswith(("w__", "g__")):
            assert np.array_equal(r0[k], r1[k]), k


def test_fit_generator_on_two_ranks_logs_global_losses_and_takes_one_checkpoint_decision(device, tmp_path):
    """The reference's training loop (compile / ModelCheckpoint(val_loss, save_best_only) / fit_generator with validation,
    multimodal.py:206-269) on two ranks with the rank-aware DataGenerator: both ranks log the same loss and val_loss per epoch
    = what one process logs on the global batches (2e-6), both take the same save_best_only decision in each of 3 epochs, and
    rank 0 alone writes the files."""
    world = 2
    ranks = _run_ranks(tmp_path, world, True, 0, 0, 0, 0, target=dp_worker.dp_fit_rank, args=(str(tmp_path),))
    one = tmp_path / "one"
    one.mkdir()
    loss, val, dec, w = dp_worker.fit_model(None, 1, 0, str(one))
    r0, r1 = ranks
    assert len(loss) == 3 and np.all(np.isfinite(loss)) and np.all(np.isfinite(val))
    for key in ("loss", "val", "dec"):
        assert np.array_equal(r0[key], r1[key]), key                 # same numbers on every rank: same decisions
    assert np.allclose(r0["loss"], loss, rtol=2e-6), (r0["loss"], loss)
    assert np.allclose(r0["val"], val, rtol=2e-6), (r0["val"], val)
    assert np.allclose(r0["dec"], dec, rtol=2e-6)
    for k, v in w.items():
        kk = "w__" + k.replace("/", "__")
        assert np.array_equal(r0[kk], r1[kk]), k
        assert np.allclose(r0[kk], v, rtol=0, atol=2e-3), (k, np.abs(r0[kk] - v).max())     # (6 Adam steps at lr 1e-3)
    files = sorted(os.listdir(tmp_path))
    assert "best_rank0.h5" in files and "model_rank0.json" in files and "weights_rank0.h5" in files
    assert not any("rank1." in f and not f.startswith("rank1.npz") for f in files), files


def _bench_in_clean_process(tmp_path, argv, timeout):
    import json
    from multiprocessing import forkserver
    if getattr(forkserver._forkserver, "_forkserver_pid", None) is None:
        pytest.skip("the fork server must be started before the GPU is initialised: run through `pytest -m gpu` (tests/conftest.py)")
    out = str(tmp_path / "bench.json")
    p = mp.get_context("forkserver").Process(target=dp_worker.run_bench, args=(out, argv, timeout))
    p.start()
    p.join(timeout + 30)
    assert not p.is_alive() and os.path.exists(out)
    return json.load(open(out))


def test_bench_two_ranks_host_comm_prints_one_line(device, tmp_path):
    """`bench.py --gpus 2 --comm host` end to end on the 1-GPU box: launcher, two rank processes, TCP rendezvous, the pipelined
    data-parallel step, max-over-ranks timing, ONE JSON line from rank 0."""
    import json
    rec = _bench_in_clean_process(tmp_path, ["--gpus", "2", "--comm", "host", "--steps", "3", "--warmup", "1", "--batch", "16",
                                             "--maxlen", "96", "--no-cpu", "--no-parity"], 240)
    assert rec["rc"] == 0, rec
    lines = [l for l in rec["stdout"].splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["comm"] == "host" and d["config"]["global_batch"] == 32 and np.isfinite(d["loss"])


def test_bench_watchdog_turns_a_stalled_rank_into_a_prompt_failure(device, tmp_path):
    """One rank stops making progress at step 2 (test hook); its peer is then stuck in the gradient all-reduce.  Without the
    watchdog this is a hang until the driver's limit; with it the stalled rank exits with code 3 after --watchdog seconds, the
    launcher terminates the peer, and the whole command is back within a minute - no rank process survives."""
    rec = _bench_in_clean_process(tmp_path, ["--gpus", "2", "--comm", "host", "--steps", "6", "--warmup", "1", "--batch", "16",
                                             "--maxlen", "96", "--no-cpu", "--no-parity", "--watchdog", "6", "--stall-at-step", "2"], 150)
    assert rec["rc"] == 3, rec
    assert "made no progress" in rec["stderr"]
    assert rec["seconds"] < 90, rec["seconds"]
