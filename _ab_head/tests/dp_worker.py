"""Rank process of the 2-rank data-parallel GPU tests (tests/test_gpu_dataparallel.py).  Runs in a process forked from a fork
server that never touched the GPU; initialises the GPU itself.  Either one GPU is shared by all ranks (Device(0)) and gradients
are summed by parallel.HostComm, or every rank has its own GPU and RCCL sums them - everything else is the product path: Engine.enqueue_train_step's pipelined schedule, shard_batch,
Engine.apply_gradients' world > 1 branch (all-reduce -> /world -> clip -> Adam -> max-norm)."""
import os
import sys
import traceback

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def dp_spec(exact):
    """The reference-size fusion network; `exact` removes every random draw (dropout 0, noise 0) so that a sharded run and a
    one-process run see identical arithmetic and can be compared number by number."""
    import mgr_amd  # noqa: F401
    from mgr_amd.configs import fusion_spec
    from mgr_amd.spec import NetworkSpec
    d = fusion_spec().to_dict()
    if exact:
        for s in d["streams"]:
            s["noise"] = 0.0
            for lay in s["layers"]:
                lay["dropout"] = 0.0
        d["fusion"]["dropout"] = 0.0
        d["head"]["dropout"] = 0.0
    return NetworkSpec.from_dict(d)


def dp_batches(spec, B, T, Lmax, nbatch):
    from mgr_amd.synthetic import synthetic_arrays
    out = []
    for i in range(nbatch):
        xs, labels, il, ll = synthetic_arrays(spec, B, T, Lmax, 500 + i, lmin=2, lmax=6)
        out.append(dict(xs, the_labels=labels, input_length=il, label_length=ll))
    return out


def run_steps(eng, spec, batches, steps, pipelined=True, local=None):
    """`steps` training steps over the batches in turn through the host-batch path fit_generator uses (next batch announced so
    that its encoder pass is prefetched).  Returns the per-step losses train_step reports (world > 1: the mean over the GLOBAL
    batch, Engine.read_global_loss); `local` (a list) receives this rank's own means."""
    names = [s["name"] for s in spec.streams]
    splits = [{n: b[n] for n in names} for b in batches]     # stable objects: the engine matches the announced batch by identity
    losses = []
    for i in range(steps):
        b, ins = batches[i % len(batches)], splits[i % len(batches)]
        nxt = splits[(i + 1) % len(batches)] if (pipelined and i + 1 < steps) else None
        losses.append(eng.train_step(ins, b["the_labels"], b["input_length"], b["label_length"], next_inputs=nxt))
        if local is not None:
            local.append(float(eng.loss_host[0]))      # (the page-locked word the step's own mean was copied to)
    return losses


def dp_rank(rank, world, port, out_path, exact, B, T, Lmax, steps, comm_kind="host"):
    """comm_kind "host": every rank on GPU 0, gradients summed by parallel.HostComm; "rccl": rank r on GPU r, gradients summed by
    ncclAllReduce over xGMI (parallel.RcclComm; the unique id travels over a plain TCP exchange on 127.0.0.1:port)."""
    try:
        sys.path.insert(0, ROOT)
        import mgr_amd  # noqa: F401
        from mgr_amd import _capi
        from mgr_amd.engine import Engine
        from mgr_amd.parallel import HostComm, RcclComm, shard_batch, tcp_bootstrap
        from mgr_amd.synthetic import synthetic_weights
        if comm_kind == "rccl":
            dev = _capi.Device(rank)
            comm = RcclComm(dev, rank, world, tcp_bootstrap(rank, world, addr="127.0.0.1", port=port, timeout=120.0))
        else:
            dev = _capi.Device(0)
            comm = HostComm(dev, rank, world, addr="127.0.0.1", port=port, timeout=120.0)
        spec = dp_spec(exact)
        eng = Engine(spec, B // world, T, Lmax, device=dev, seed=100 + rank, comm=comm, world=world)
        assert eng.can_pipeline
        eng.set_weights(synthetic_weights(spec, 3))
        mine = [shard_batch(b, rank, world) for b in dp_batches(spec, B, T, Lmax, 2)]
        local = []
        losses = run_steps(eng, spec, mine, steps, local=local)
        dev.sync()
        import ctypes
        st = ctypes.c_uint(7)
        dev.call("mgr_scan_status", ctypes.byref(st))
        nl, ns = ctypes.c_int(), ctypes.c_int()
        dev.call("mgr_persist_stats", ctypes.byref(nl), ctypes.byref(ns))
        w, g = eng.get_weights(), eng.get_grads()
        comm.barrier()
        np.savez(out_path, losses=np.array(losses, np.float64), local=np.array(local, np.float64), status=st.value, persist=np.array([nl.value, ns.value]),
                 **{"w__" + k.replace("/", "__"): v for k, v in w.items()},
                 **{"g__" + k.replace("/", "__"): v for k, v in g.items()})
        comm.close()
        eng.close()
    except BaseException:
        with open(out_path + ".err", "w") as f:
            f.write(traceback.format_exc())
        raise


def fit_model(comm, world, rank, workdir, epochs=3):
    """The reference's training script in small (multimodal.py:206-269): compile(Adam), ModelCheckpoint(val_loss, save_best_only),
    fit_generator over a (rank-aware) DataGenerator with validation.  Returns (loss history, val_loss history, checkpoint
    decisions per epoch, weights after training)."""
    import mgr_amd  # noqa: F401
    from mgr_amd.keras_like import Adam, Callback, Model, ModelCheckpoint
    from mgr_amd.multimodal_fusion.data_generator import DataGenerator
    spec = dp_spec(True)
    m = Model(spec, device=comm.dev if comm is not None else 0)
    if comm is not None:
        m.distribute(comm, world)
    m.compile(loss={'ctc': lambda y_true, y_pred: y_pred}, optimizer=Adam(lr=1e-3, clipvalue=0.5, decay=1e-5))
    from mgr_amd.synthetic import synthetic_weights
    m.set_weights_dict(synthetic_weights(spec, 3))
    gen = DataGenerator(8, 20, 39, 64, 22, 'train', synthetic_files=45, rank=rank, world=world, absolute_max_sequence_len=8)
    gen.model_json_name = os.path.join(workdir, "model_rank%d.json" % rank)
    gen.model_weights_name = os.path.join(workdir, "weights_rank%d.h5" % rank)
    ck = ModelCheckpoint(os.path.join(workdir, "best_rank%d.h5" % rank), monitor='val_loss', save_best_only=True, save_weights_only=True)
    decisions = []

    class Watch(Callback):
        def on_epoch_end(self, epoch, logs=None):
            decisions.append(float(ck.best))

    h = m.fit_generator(gen.next_train(), steps_per_epoch=gen.get_size(True) // 8, epochs=epochs, verbose=0,
                        validation_data=gen.next_val(), validation_steps=gen.get_size(False) // 8, callbacks=[ck, Watch(), gen])
    return h.history["loss"], h.history["val_loss"], decisions, m.get_weights_dict()


def dp_fit_rank(rank, world, port, out_path, workdir):
    """Rank process of the Model-level data-parallel test: fit_generator with the rank-aware DataGenerator on GPU 0 (HostComm)."""
    try:
        sys.path.insert(0, ROOT)
        import mgr_amd  # noqa: F401
        from mgr_amd import _capi
        from mgr_amd.parallel import HostComm
        dev = _capi.Device(0)
        comm = HostComm(dev, rank, world, addr="127.0.0.1", port=port, timeout=120.0)
        loss, val, dec, w = fit_model(comm, world, rank, workdir)
        comm.barrier()
        np.savez(out_path, loss=np.array(loss, np.float64), val=np.array(val, np.float64), dec=np.array(dec, np.float64),
                 **{"w__" + k.replace("/", "__"): v for k, v in w.items()})
        comm.close()
    except BaseException:
        with open(out_path + ".err", "w") as f:
            f.write(traceback.format_exc())
        raise


def run_bench(out_path, argv, timeout):
    """Runs `python bench.py <argv>` as a child of THIS process (forked from the fork server: it never touched the GPU, so it may
    exec) and leaves {rc, seconds, stdout, stderr} in out_path as JSON."""
    import json
    import subprocess
    import time
    t0 = time.time()
    try:
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(argv), capture_output=True, text=True, timeout=timeout)
        rec = dict(rc=r.returncode, stdout=r.stdout[-4000:], stderr=r.stderr[-4000:])
    except subprocess.TimeoutExpired as e:
        rec = dict(rc=None, stdout=str(e.stdout)[-2000:], stderr=str(e.stderr)[-2000:])
    rec["seconds"] = time.time() - t0
    with open(out_path, "w") as f:
        json.dump(rec, f)
