"""Data parallelism (BASELINE.json config 4; not present in the reference).

One process per GPU.  Every replica holds all weights; the global minibatch is split into contiguous
per-rank slices of the DataGenerator dict; after the local backward pass ONE all-reduce(sum) runs over the
flat trainable-gradient buffer, then every replica applies  g/world -> clip -> Adam -> max-norm
identically (clipping after the reduce is what makes 8x64 equivalent to 1x512).

Communicators:
  RcclComm  - mgr_comm_* of libmgr.so (RCCL over xGMI); the 128-byte unique id is distributed by the
              caller's bootstrap callable (torch.distributed's store in bench.py, or anything else).
  HostComm  - a host-side stand-in with the same interface used by the world_size-2 gloo CPU tests.
"""
import ctypes as C

import numpy as np

from . import _capi


def shard_batch(batch, rank, world):
    """Contiguous slice `rank` of `world` of every array in a DataGenerator batch dict."""
    out = {}
    for k, v in batch.items():
        v = np.asarray(v)
        n = v.shape[0]
        if n % world:
            raise ValueError("global batch %d not divisible by world size %d" % (n, world))
        per = n // world
        out[k] = v[rank * per:(rank + 1) * per]
    return out


class RcclComm:
    def __init__(self, dev, rank, world, bootstrap):
        """bootstrap(bytes_or_None) -> bytes : rank 0 passes the id, every rank gets it back."""
        self.dev, self.rank, self.world = dev, rank, world
        lib = dev.lib
        uid = None
        if rank == 0:
            buf = C.create_string_buffer(128)
            _capi.check(lib.mgr_comm_unique_id(buf))
            uid = buf.raw
        uid = bootstrap(uid)
        comm = C.c_void_p()
        _capi.check(lib.mgr_comm_init_rank(dev.ctx, world, rank, uid, C.byref(comm)))
        self.comm = comm
        self._scratch = dev.zeros((4,))

    def allreduce_sum(self, darr, n):
        _capi.check(self.dev.lib.mgr_allreduce_sum(self.comm, darr.ptr, n))

    def allreduce_max_scalar(self, value):
        self._scratch.upload(np.array([value, 0, 0, 0], np.float32))
        _capi.check(self.dev.lib.mgr_allreduce_max(self.comm, self._scratch.ptr, 1))
        return float(self._scratch.download()[0])

    def barrier(self):
        self.allreduce_max_scalar(0.0)

    def close(self):
        if self.comm:
            self.dev.lib.mgr_comm_destroy(self.comm)
            self.comm = None


def data_parallel_update(local_grads, allreduce_sum, world):
    """Host-visible statement of the update rule's reduction order (used by the CPU gloo tests):
    returns the gradient every replica feeds to clip+Adam:  (sum over ranks of local mean-grads) / world."""
    total = allreduce_sum(np.asarray(local_grads))
    return total / float(world)


def tcp_bootstrap(rank, world, addr=None, port=None, timeout=600.0):
    """bootstrap(uid) callable for RcclComm that needs nothing but the launcher's MASTER_ADDR / MASTER_PORT: rank 0 serves the
    128-byte unique id on MASTER_PORT + 101 (the launcher's own store owns MASTER_PORT), every other rank fetches it.
    No torch import: a PyTorch wheel brings its own HIP / HSA / RCCL copies into the process."""
    import os
    import socket
    import time
    addr = addr or os.environ.get("MASTER_ADDR", "127.0.0.1")
    port = int(port or int(os.environ.get("MASTER_PORT", "29500")) + 101)

    def bootstrap(uid):
        deadline = time.time() + timeout
        if rank == 0:
            srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
            srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
            srv.bind((addr if addr not in ("localhost",) else "127.0.0.1", port))
            srv.listen(world)
            srv.settimeout(timeout)
            for _ in range(world - 1):
                c, _ = srv.accept()
                c.sendall(uid)
                c.close()
            srv.close()
            return uid
        while True:
            try:
                c = socket.create_connection((addr, port), timeout=5.0)
                break
            except OSError:
                if time.time() > deadline:
                    raise RuntimeError("rank %d: no unique id from rank 0 at %s:%d" % (rank, addr, port))
                time.sleep(0.1)
        buf = b""
        while len(buf) < 128:
            chunk = c.recv(128 - len(buf))
            if not chunk:
                raise RuntimeError("rank %d: connection closed while receiving the unique id" % rank)
            buf += chunk
        c.close()
        return buf
    return bootstrap
