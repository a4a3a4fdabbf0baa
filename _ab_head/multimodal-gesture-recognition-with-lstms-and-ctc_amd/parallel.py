"""Data parallelism (BASELINE.json config 4; not present in the reference).

One process per GPU.  Every replica holds all weights; the global minibatch is split into contiguous
per-rank slices of the DataGenerator dict; after the local backward pass ONE all-reduce(sum) runs over the
flat trainable-gradient buffer, then every replica applies  g/world -> clip -> Adam -> max-norm
identically (clipping after the reduce is what makes 8x64 equivalent to 1x512).

Communicators:
  RcclComm  - mgr_comm_* of libmgr.so (RCCL over xGMI); the 128-byte unique id is distributed by the
              caller's bootstrap callable (torch.distributed's store in bench.py, or anything else).
  HostComm  - the same interface without RCCL: device -> host copy, rank-ordered sum over plain TCP sockets (star through
              rank 0), host -> device copy, all on the engine's stream.  Lets a world > 1 step run where RCCL cannot (several
              ranks on ONE GPU: `bench.py --gpus 2 --comm host`, tests/test_gpu_dataparallel.py) and is the reference the RCCL
              path is compared with.  Not a performance path.
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
    #: NCCL_MAX_NCHANNELS this process created its communicator under (None: RCCL's default) - bench.py records it
    max_channels_in_effect = None

    def __init__(self, dev, rank, world, bootstrap, max_channels=8):
        """bootstrap(bytes_or_None) -> bytes : rank 0 passes the id, every rank gets it back.
        max_channels: upper bound of RCCL channels (= workgroups of its all-reduce kernel) unless the environment already
        says NCCL_MAX_NCHANNELS: the gradient vector is a few MB (latency-bound), and the kernel has to find room on CUs that
        persistent scans of this context occupy - 8 workgroups of a collective's shape start within 200 us beside 408 resident
        scan workgroups (tests/test_gpu_residency.py::test_a_collective_shaped_guest_...), RCCL's default of dozens may have to
        wait for a scan to end.  None / 0 leaves RCCL's default.  The variable is process-wide: what was in effect when the
        communicator was created is kept in `max_channels_in_effect` and printed in bench.py's JSON line."""
        import os
        self.dev, self.rank, self.world = dev, rank, world
        if max_channels and "NCCL_MAX_NCHANNELS" not in os.environ:
            os.environ["NCCL_MAX_NCHANNELS"] = str(int(max_channels))     # read by RCCL when the communicator is created
        v = os.environ.get("NCCL_MAX_NCHANNELS")
        RcclComm.max_channels_in_effect = self.max_channels_in_effect = int(v) if v and v.isdigit() else None
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

    def ranks_seen(self):
        """(ranks, own rank) as RCCL itself reports them for this communicator (ncclCommCount / ncclCommUserRank) - not what this
        process was told by its launcher.  bench.py prints it: a multi-GPU line then proves that N ranks met inside RCCL."""
        n, r = C.c_int(-1), C.c_int(-1)
        _capi.check(self.dev.lib.mgr_comm_count(self.comm, C.byref(n), C.byref(r)))
        return n.value, r.value

    def allreduce_ms(self):
        """(gradient all-reduces, their summed device time in ms) since the last Device.prof_reset(): profiling family
        `allreduce` (HIP events around ncclAllReduce on the stream it is enqueued on; needs Device.prof_enable)."""
        return self.dev.prof_get(_capi.K_ALLREDUCE)

    def allreduce_max_scalar(self, value):
        self._scratch.upload(np.array([value, 0, 0, 0], np.float32))
        _capi.check(self.dev.lib.mgr_allreduce_max(self.comm, self._scratch.ptr, 1))
        return float(self._scratch.download()[0])

    def allreduce_sum_scalar(self, value):
        """Sum of one float over the ranks (validation loss, Model.evaluate_generator); the same bits on every rank."""
        self._scratch.upload(np.array([value, 0, 0, 0], np.float32))
        _capi.check(self.dev.lib.mgr_allreduce_sum(self.comm, self._scratch.ptr, 1))
        return float(self._scratch.download()[0])

    def barrier(self):
        self.allreduce_max_scalar(0.0)

    def close(self):
        if self.comm:
            self.dev.lib.mgr_comm_destroy(self.comm)
            self.comm = None


class HostComm:
    """Drop-in for RcclComm that reduces on the host.  `dev` is the engine's Device (the copies run on ITS current stream, so
    they are ordered after the gradient kernels and before the optimizer exactly like the RCCL kernel would be); `dev=None`
    gives a host-only communicator (numpy in, numpy out) for CPU tests.  Reduction order is rank order on rank 0 and the
    result is broadcast, so every replica receives the same bits."""

    PORT_OFFSET = 102
    host_blocking = True      # allreduce_sum blocks the host: the engine must not park a device-side wait in front of it

    def __init__(self, dev, rank, world, addr=None, port=None, timeout=600.0):
        import os
        import socket
        import time
        self.dev, self.rank, self.world = dev, int(rank), int(world)
        self.addr = addr or os.environ.get("MASTER_ADDR", "127.0.0.1")
        self.port = int(port or int(os.environ.get("MASTER_PORT", "29500")) + self.PORT_OFFSET)
        self.peers = {}        # rank 0: rank -> socket
        self.sock = None       # other ranks: socket to rank 0
        self._seen = 1         # ranks that met at rank 0 (ranks_seen)
        self._ar_n, self._ar_s = 0, 0.0     # gradient all-reduces and their host wall time (allreduce_ms)
        if self.world == 1:
            return
        if self.rank == 0:
            srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
            srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
            srv.bind(("127.0.0.1" if self.addr == "localhost" else self.addr, self.port))
            srv.listen(self.world)
            srv.settimeout(timeout)
            for _ in range(self.world - 1):
                c, _ = srv.accept()
                c.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                c.settimeout(timeout)
                r = int.from_bytes(self._recv_exact(c, 4), "little")
                self.peers[r] = c
            srv.close()
            if sorted(self.peers) != list(range(1, self.world)):
                raise RuntimeError("HostComm: expected ranks 1..%d, got %s" % (self.world - 1, sorted(self.peers)))
            self._seen = len(self.peers) + 1
            for c in self.peers.values():      # every rank learns how many ranks really met at rank 0 (ranks_seen)
                c.sendall(self._seen.to_bytes(4, "little"))
        else:
            deadline = time.time() + timeout
            while True:
                try:
                    c = socket.create_connection((self.addr, self.port), timeout=5.0)
                    break
                except OSError:
                    if time.time() > deadline:
                        raise RuntimeError("HostComm rank %d: rank 0 not reachable at %s:%d" % (self.rank, self.addr, self.port))
                    time.sleep(0.05)
            c.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
            c.settimeout(timeout)
            c.sendall(self.rank.to_bytes(4, "little"))
            self._seen = int.from_bytes(self._recv_exact(c, 4), "little")
            self.sock = c

    @staticmethod
    def _recv_exact(c, n):
        buf = bytearray(n)
        view = memoryview(buf)
        got = 0
        while got < n:
            k = c.recv_into(view[got:], n - got)
            if k == 0:
                raise RuntimeError("HostComm: peer closed the connection")
            got += k
        return bytes(buf)

    def _reduce_host(self, a, op):
        """a: contiguous float32 numpy array; returns the reduction over all ranks (same bits on every rank)."""
        if self.world == 1:
            return a
        if self.rank == 0:
            acc = a.copy()
            for r in range(1, self.world):          # rank order: deterministic
                other = np.frombuffer(self._recv_exact(self.peers[r], a.nbytes), dtype=np.float32)
                acc = op(acc, other)
            raw = acc.tobytes()
            for r in range(1, self.world):
                self.peers[r].sendall(raw)
            return acc
        self.sock.sendall(a.tobytes())
        return np.frombuffer(self._recv_exact(self.sock, a.nbytes), dtype=np.float32).copy()

    def allreduce_sum_host(self, a):
        a = np.ascontiguousarray(a, dtype=np.float32)
        return self._reduce_host(a.ravel(), np.add).reshape(a.shape)

    def allreduce_sum(self, darr, n):
        import time
        view = darr.view(0, (int(n),))
        local = view.download()               # (synchronises the engine's current stream: the gradient kernels are done)
        t0 = time.perf_counter()
        view.upload(self._reduce_host(local, np.add))
        self._ar_n += 1
        self._ar_s += time.perf_counter() - t0

    def ranks_seen(self):
        """(ranks that connected at rank 0, own rank): the counterpart of RcclComm.ranks_seen for the host communicator."""
        return self._seen, self.rank

    def allreduce_ms(self, reset=False):
        """(gradient all-reduces, their summed host wall time in ms: TCP exchange + host sum + upload) since the last reset."""
        out = (self._ar_n, self._ar_s * 1e3)
        if reset:
            self._ar_n, self._ar_s = 0, 0.0
        return out

    def allreduce_max_scalar(self, value):
        return float(self._reduce_host(np.array([value], np.float32), np.maximum)[0])

    def allreduce_sum_scalar(self, value):
        return float(self._reduce_host(np.array([value], np.float32), np.add)[0])

    def barrier(self):
        self.allreduce_max_scalar(0.0)

    def close(self):
        for c in list(self.peers.values()) + ([self.sock] if self.sock else []):
            try:
                c.close()
            except OSError:
                pass
        self.peers, self.sock = {}, None


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
