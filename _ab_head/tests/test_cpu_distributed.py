"""world_size-2 gloo test (CPU) of the data-parallel update rule: two ranks, each with half of the global batch,
all-reduce(sum) of the local mean-gradients, /world, THEN clip + Adam + max-norm  ==  one process on the full batch.
The compute backend here is the oracle (tests may use it); what is under test is mgr_amd.parallel's sharding and
reduction order, the same code path bench.py drives with RcclComm on GPUs."""
import os
import socket

import numpy as np
import pytest

from oracle import network_ref as nr
from tests.helpers import load_case


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _is_lstm_mask(key):
    """LSTM input-dropout masks are [4,B,F]; noise and the head mask are [B,T,*]."""
    return key.endswith("/mask") and key != "head/mask"


def _extend_rand(rand):
    """Duplicate sample 0 so that the 3-sample fixture becomes a 4-sample global batch."""
    return {k: (np.concatenate([v, v[:, :1]], 1) if _is_lstm_mask(k) else np.concatenate([v, v[:1]], 0))
            for k, v in rand.items()}


def _worker(rank, world, port, out_dir):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch
    import torch.distributed as dist
    import mgr_amd  # noqa: F401
    from mgr_amd.parallel import data_parallel_update, shard_batch
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    z, meta, grab = load_case("fusion_tiny")
    spec = meta["spec"]
    B = meta["B"] + 1  # 4 samples: duplicate one so the batch splits evenly
    inputs = {k: np.concatenate([v, v[:1]], 0) for k, v in grab("x__").items()}
    labels = np.concatenate([z["labels"], z["labels"][:1]], 0)
    il = np.concatenate([z["input_length"], z["input_length"][:1]], 0)
    ll = np.concatenate([z["label_length"], z["label_length"][:1]], 0)
    rand = _extend_rand(grab("r__"))
    batch = dict(inputs, the_labels=labels, input_length=il, label_length=ll)
    mine = shard_batch(batch, rank, world)
    per = B // world
    my_rand = {k: (v[:, rank * per:(rank + 1) * per] if _is_lstm_mask(k) else v[rank * per:(rank + 1) * per])
               for k, v in rand.items()}
    w = {k: v.copy() for k, v in grab("w__").items()}
    loss, _, grads, _ = nr.loss_and_grads(spec, w, {k: mine[k] for k in inputs}, mine["the_labels"], mine["input_length"],
                                          mine["label_length"], my_rand)

    def allreduce_sum(a):
        t = torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64))
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return t.numpy()

    names = sorted(grads)
    flat = np.concatenate([grads[n].ravel() for n in names])
    flat = data_parallel_update(flat, allreduce_sum, world)
    off = 0
    g2 = {}
    for n in names:
        g2[n] = flat[off:off + grads[n].size].reshape(grads[n].shape)
        off += grads[n].size
    tr = nr.Trainer(spec, w)
    tr.apply(g2)
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), loss=loss, **{k.replace("/", "__"): v for k, v in tr.w.items()})
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_equals_single_process(tmp_path):
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0 = np.load(tmp_path / "rank0.npz")
    r1 = np.load(tmp_path / "rank1.npz")
    # single process, global batch of 4
    z, meta, grab = load_case("fusion_tiny")
    inputs = {k: np.concatenate([v, v[:1]], 0) for k, v in grab("x__").items()}
    labels = np.concatenate([z["labels"], z["labels"][:1]], 0)
    il = np.concatenate([z["input_length"], z["input_length"][:1]], 0)
    ll = np.concatenate([z["label_length"], z["label_length"][:1]], 0)
    rand = _extend_rand(grab("r__"))
    tr = nr.Trainer(meta["spec"], {k: v.copy() for k, v in grab("w__").items()})
    loss = tr.train_on_batch(inputs, labels, il, ll, rand)
    assert abs((float(r0["loss"]) + float(r1["loss"])) / 2 - loss) < 1e-10 * abs(loss)
    for k, v in tr.w.items():
        kk = k.replace("/", "__")
        assert np.array_equal(r0[kk], r1[kk]), k           # replicas stay bit-identical
        assert np.allclose(r0[kk], v, rtol=1e-10, atol=1e-14), k


def _tcp_rank(rank, world, port, q):
    from mgr_amd.parallel import tcp_bootstrap
    boot = tcp_bootstrap(rank, world, addr="127.0.0.1", port=port, timeout=30.0)
    uid = bytes(range(128)) if rank == 0 else None
    q.put((rank, boot(uid)))


def test_tcp_bootstrap_distributes_the_unique_id():
    """The RCCL unique id travels from rank 0 to every other rank over a plain TCP exchange (no torch in the ranks)."""
    import multiprocessing as mp
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    world = 3
    procs = [ctx.Process(target=_tcp_rank, args=(r, world, port, q)) for r in (2, 1, 0)]   # clients may start first
    for p in procs:
        p.start()
    got = dict(q.get(timeout=60) for _ in range(world))
    for p in procs:
        p.join(30)
    assert all(got[r] == bytes(range(128)) for r in range(world))


def _hostcomm_rank(rank, world, port, q):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import mgr_amd  # noqa: F401
    from mgr_amd.parallel import HostComm
    comm = HostComm(None, rank, world, addr="127.0.0.1", port=port, timeout=30.0)
    rng = np.random.default_rng(rank)
    a = (rng.standard_normal(1001) * 10.0 ** rng.integers(-3, 4, 1001)).astype(np.float32)
    s1 = comm.allreduce_sum_host(a)
    s2 = comm.allreduce_sum_host(a.reshape(7, 143))           # a second collective on the same connections, another shape
    mx = comm.allreduce_max_scalar(float(rank) + 0.5)
    comm.barrier()
    seen = comm.ranks_seen()
    comm.close()
    q.put((rank, a, s1, s2, mx, seen))


def test_hostcomm_sums_in_rank_order_and_every_rank_gets_the_same_bits():
    """parallel.HostComm (the communicator of the 2-rank GPU test and of `bench.py --comm host`): star reduction through rank
    0 in RANK ORDER ((r0 + r1) + r2 in fp32), the result broadcast - so replicas stay bit-identical."""
    import multiprocessing as mp
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    world = 3
    procs = [ctx.Process(target=_hostcomm_rank, args=(r, world, port, q)) for r in (1, 2, 0)]   # clients may start first
    for p in procs:
        p.start()
    got = {}
    for _ in range(world):
        r, a, s1, s2, mx, seen = q.get(timeout=60)
        got[r] = (a, s1, s2, mx)
        assert seen == (world, r)      # what bench.py prints as comm.nranks_seen: the ranks that really met at rank 0
    for p in procs:
        p.join(30)
    expect = (got[0][0] + got[1][0]) + got[2][0]                # fp32, rank order
    for r in range(world):
        assert np.array_equal(got[r][1], expect)
        assert np.array_equal(got[r][2], expect.reshape(7, 143))
        assert got[r][3] == 2.5


def test_hostcomm_fails_loudly_when_a_peer_dies_or_never_shows_up():
    """A rank that goes away must surface as an error on the others (a closed connection / a timeout), never as a hang:
    rank 1 connects and closes before contributing; a second communicator waits for a rank that never connects."""
    import socket
    import threading
    from mgr_amd.parallel import HostComm
    port = _free_port()
    err = {}

    def rank0():
        try:
            c = HostComm(None, 0, 2, addr="127.0.0.1", port=port, timeout=10.0)
            try:
                c.allreduce_sum_host(np.ones(5, np.float32))
            finally:
                c.close()
        except Exception as e:      # noqa: BLE001 - the test inspects it
            err[0] = e

    t = threading.Thread(target=rank0)
    t.start()
    c1 = HostComm(None, 1, 2, addr="127.0.0.1", port=port, timeout=10.0)
    c1.close()                                   # dies before its first all-reduce
    t.join(20)
    assert not t.is_alive()
    assert isinstance(err.get(0), RuntimeError) and "closed" in str(err[0])
    # nobody connects: the accept times out instead of blocking for good
    with pytest.raises((socket.timeout, TimeoutError, OSError)):
        HostComm(None, 0, 2, addr="127.0.0.1", port=_free_port(), timeout=0.5)
    # rank 0 is not there: the client gives up with a message that names the address
    with pytest.raises(RuntimeError, match="not reachable"):
        HostComm(None, 1, 2, addr="127.0.0.1", port=_free_port(), timeout=0.5)
