"""ctypes binding of include/mgr.h (libmgr.so).  No CPU fallback: a missing library raises."""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libmgr.so")

c_float_p = C.POINTER(C.c_float)
c_int32_p = C.POINTER(C.c_int32)
vp = C.c_void_p
sz = C.c_size_t
u64 = C.c_uint64
i32 = C.c_int

# name -> (restype, argtypes).  Must list every symbol include/mgr.h declares (tests check this).
SIGNATURES = {
    "mgr_version": (i32, []),
    "mgr_last_error": (C.c_char_p, []),
    "mgr_device_count": (i32, [C.POINTER(i32)]),
    "mgr_ctx_create": (i32, [i32, C.POINTER(vp)]),
    "mgr_ctx_destroy": (i32, [vp]),
    "mgr_device_info": (i32, [vp, C.POINTER(i32), C.POINTER(sz), C.c_char_p, i32]),
    "mgr_alloc": (i32, [vp, sz, C.POINTER(vp)]),
    "mgr_free": (i32, [vp, vp]),
    "mgr_memset": (i32, [vp, vp, i32, sz]),
    "mgr_h2d": (i32, [vp, vp, vp, sz]),
    "mgr_d2h": (i32, [vp, vp, vp, sz]),
    "mgr_d2d": (i32, [vp, vp, vp, sz]),
    "mgr_d2h_async": (i32, [vp, vp, vp, sz]),
    "mgr_event_sync": (i32, [vp, i32]),
    "mgr_sync": (i32, [vp]),
    "mgr_stream_set": (i32, [vp, i32]),
    "mgr_stream_wait": (i32, [vp, i32, i32]),
    "mgr_stream_set_priority": (i32, [vp, i32, i32]),
    "mgr_event_record": (i32, [vp, i32]),
    "mgr_stream_wait_event": (i32, [vp, i32, i32]),
    "mgr_event_elapsed_ms": (i32, [vp, i32, i32, C.POINTER(C.c_float)]),
    "mgr_prof_enable": (i32, [vp, i32]),
    "mgr_prof_get": (i32, [vp, i32, C.POINTER(i32), C.POINTER(C.c_float)]),
    "mgr_prof_reset": (i32, [vp]),
    "mgr_add_gaussian_noise": (i32, [vp, vp, vp, sz, C.c_float, u64]),
    "mgr_dropout_mask": (i32, [vp, vp, sz, C.c_float, u64]),
    "mgr_lstm_pack": (i32, [vp, vp, vp, i32, i32, i32]),
    "mgr_transpose": (i32, [vp, vp, vp, i32, i32]),
    "mgr_lstm_input_proj": (i32, [vp, vp, i32, vp, vp, vp, vp, i32, i32, i32, i32]),
    "mgr_lstm_input_proj_dropout_ws_bytes": (sz, [i32, i32, i32]),
    "mgr_lstm_input_proj_dropout": (i32, [vp, vp, i32, vp, C.c_float, vp, vp, vp, i32, i32, i32, i32, vp, sz]),
    "mgr_lstm_input_proj_dropout_wants_transposed": (i32, [vp, C.c_float, i32]),
    "mgr_lstm_input_proj_dropout_t": (i32, [vp, vp, i32, vp, C.c_float, vp, vp, vp, i32, i32, i32, i32, vp, sz, C.c_float]),
    "mgr_transpose_bt": (i32, [vp, vp, i32, vp, i32, i32, i32, i32]),
    "mgr_lstm_input_proj_dropout_ts_ws_bytes": (sz, [i32, i32, i32]),
    "mgr_lstm_input_proj_dropout_ts": (i32, [vp, vp, i32, vp, C.c_float, vp, vp, vp, i32, i32, i32, i32, vp, sz]),
    "mgr_transpose_bt_split": (i32, [vp, vp, i32, vp, i32, i32, i32, i32]),
    "mgr_transpose_bt_split_shift": (i32, [vp, vp, i32, vp, i32, i32, i32, i32, i32]),
    "mgr_weight_planes_cache": (i32, [vp, vp, i32]),
    "mgr_lstm_input_proj_pair": (i32, [vp, vp, i32, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32]),
    "mgr_lstm_scan_ws_bytes": (sz, [i32, i32, i32]),
    "mgr_lstm_scan_fwd": (i32, [vp, vp, vp, vp, i32, vp, i32, vp, vp, i32, i32, i32, i32, vp, sz]),
    "mgr_lstm_scan_multi_ws_bytes": (sz, [i32, vp]),
    "mgr_lstm_scan_fwd_multi": (i32, [vp, i32, vp, vp, sz]),
    "mgr_lstm_scan_fwd_multi_ex": (i32, [vp, i32, vp, vp, sz, vp]),
    "mgr_abi_struct_sizes": (i32, [vp]),
    "mgr_tune": (i32, [vp, i32, i32]),
    "mgr_tune_get": (i32, [vp, i32, C.POINTER(i32)]),
    "mgr_probe_xcc": (i32, [vp, i32, i32, i32, vp]),
    "mgr_stream_delay": (i32, [vp, i32]),
    "mgr_probe_guest": (i32, [vp, i32, i32, i32, i32, vp]),
    "mgr_host_alloc": (i32, [vp, sz, C.POINTER(vp)]),
    "mgr_host_free": (i32, [vp, vp]),
    "mgr_h2d_async": (i32, [vp, vp, vp, sz]),
    "mgr_scan_status": (i32, [vp, vp]),
    "mgr_scan_status_clear": (i32, [vp]),
    "mgr_scan_status_ex": (i32, [vp, vp]),
    "mgr_scan_status_bind": (i32, [vp, vp]),
    "mgr_scan_status_inject": (i32, [vp, C.c_uint]),
    "mgr_update_gate_eval": (i32, [vp, C.c_uint, vp]),
    "mgr_update_gate_set": (i32, [vp, vp]),
    "mgr_stream_wait_next_resident": (i32, [vp, i32]),
    "mgr_stream_wait_resident": (i32, [vp, C.c_uint, i32]),
    "mgr_stream_wait_resident_word": (i32, [vp, vp, i32]),
    "mgr_resident_wait_stats": (i32, [vp, vp]),
    "mgr_persist_stats": (i32, [vp, C.POINTER(i32), C.POINTER(i32)]),
    "mgr_skeletal_features": (i32, [vp, vp, sz, vp]),
    "mgr_lstm_scan_bwd_multi_ws_bytes": (sz, [i32, vp]),
    "mgr_lstm_scan_bwd_multi": (i32, [vp, i32, vp, vp, sz]),
    "mgr_lstm_scan_bwd_multi_ex": (i32, [vp, i32, vp, vp, sz, vp]),
    "mgr_lstm_scan_bwd": (i32, [vp, vp, i32, vp, vp, vp, vp, i32, i32, i32, i32, vp, sz]),
    "mgr_lstm_param_grads_ws_bytes": (sz, [i32, i32, i32, i32]),
    "mgr_lstm_param_grads": (i32, [vp, vp, i32, vp, vp, i32, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp, sz]),
    "mgr_lstm_param_grads_dropout_ws_bytes": (sz, [i32, i32, i32, i32]),
    "mgr_lstm_param_grads_dropout": (i32, [vp, vp, i32, vp, C.c_float, vp, i32, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp, sz]),
    "mgr_lstm_param_grads_dropout_wants_transposed": (i32, [vp, C.c_float, i32]),
    "mgr_lstm_param_grads_dropout_t_ws_bytes": (sz, [i32, i32, i32, i32, i32]),
    "mgr_lstm_param_grads_dropout_t": (i32, [vp, vp, i32, vp, C.c_float, vp, i32, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp, sz, C.c_float]),
    "mgr_lstm_param_grads_dropout_ts_ws_bytes": (sz, [i32, i32, i32, i32, i32]),
    "mgr_lstm_param_grads_dropout_ts": (i32, [vp, vp, i32, vp, C.c_float, vp, i32, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp, sz, vp, vp, vp, vp]),
    "mgr_lstm_input_grad": (i32, [vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32]),
    "mgr_dense_softmax_fwd": (i32, [vp, vp, i32, vp, C.c_float, u64, vp, vp, vp, i32, i32, i32, i32]),
    "mgr_dense_bwd_ws_bytes": (sz, [i32, i32, i32, i32]),
    "mgr_dense_bwd": (i32, [vp, vp, i32, vp, C.c_float, u64, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp, sz]),
    "mgr_head_ws_bytes": (sz, [i32, i32, i32, i32, i32]),
    "mgr_head_fwd_bwd": (i32, [vp, vp, i32, vp, C.c_float, u64, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, C.c_float, C.c_float,
                               vp, vp, vp, vp, vp, vp, vp, i32, vp, sz]),
    "mgr_ctc_ws_bytes": (sz, [i32, i32, i32, i32]),
    "mgr_ctc_loss_grad": (i32, [vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, C.c_float, C.c_float, vp, vp, vp, sz]),
    "mgr_adam_step": (i32, [vp, vp, vp, vp, vp, sz, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float]),
    "mgr_maxnorm_cols": (i32, [vp, vp, i32, i32, C.c_float, C.c_float]),
    "mgr_add2d": (i32, [vp, vp, i32, vp, i32, vp, i32, sz, i32]),
    "mgr_mean": (i32, [vp, vp, i32, vp]),
    "mgr_comm_unique_id": (i32, [C.c_char_p]),
    "mgr_comm_init_rank": (i32, [vp, i32, i32, C.c_char_p, C.POINTER(vp)]),
    "mgr_allreduce_sum": (i32, [vp, vp, sz]),
    "mgr_allreduce_max": (i32, [vp, vp, sz]),
    "mgr_comm_destroy": (i32, [vp]),
    "mgr_comm_count": (i32, [vp, C.POINTER(i32), C.POINTER(i32)]),
    "mgr_frame_argmax": (i32, [vp, vp, i32, i32, i32, i32, vp, vp]),
    "mgr_ctc_beam_ws_bytes": (sz, [i32, i32, i32, i32]),
    "mgr_ctc_beam_search": (i32, [vp, vp, vp, i32, i32, i32, i32, i32, i32, C.c_float, i32, vp, vp, vp, vp, sz]),
}

SCAN_GAVE_UP, SCAN_NONFINITE = 1, 8   # enum in include/mgr.h (mgr_scan_status)

(K_GEMM_NN, K_GEMM_TN, K_GEMM_NT, K_SCAN_FWD, K_SCAN_BWD, K_DENSE_FWD, K_DENSE_BWD, K_CTC, K_ADAM, K_MISC, K_ALLREDUCE, K_SCAN_FWD_NARROW) = range(12)
KERNEL_FAMILIES = ["gemm_nn", "gemm_tn", "gemm_nt", "scan_fwd", "scan_bwd", "dense_fwd", "dense_bwd", "ctc", "adam", "misc", "allreduce",
                   "scan_fwd_narrow"]



class ScanJob(C.Structure):
    """struct mgr_scan_job"""
    _fields_ = [("Z", vp), ("Up", vp), ("Y", vp), ("R", vp), ("gates", vp), ("cs", vp),
                ("ldy", i32), ("ldr", i32), ("B", i32), ("T", i32), ("H", i32), ("reverse", i32),
                ("YT", vp), ("ytb", C.c_longlong), ("ldt", i32), ("yt_split", i32)]


class ScanBwdJob(C.Structure):
    """struct mgr_scan_bwd_job"""
    _fields_ = [("dY", vp), ("gates", vp), ("cs", vp), ("Up", vp), ("dZ", vp),
                ("lddy", i32), ("B", i32), ("T", i32), ("H", i32), ("reverse", i32), ("dzmax", vp), ("dbsum", vp)]


class ScanLaunchOpts(C.Structure):
    """struct mgr_scan_launch_opts"""
    _fields_ = [("struct_size", C.c_uint), ("form", i32), ("seq_out", vp)]


# enums of include/mgr.h (mgr_scan_launch_opts.form)
SCAN_FORM_AUTO, SCAN_FORM_PLAIN, SCAN_FORM_PAIR, SCAN_FORM_FUSED, SCAN_FORM_FUSED_ANY = range(5)
BPTT_FORM_AUTO, BPTT_FORM_TRIMMED, BPTT_FORM_YIELDING, BPTT_FORM_DIRECT, BPTT_FORM_FUSED, BPTT_FORM_FUSED_DIRECT, BPTT_FORM_SINGLE_CU = range(7)
SEQ_NONE = 0xFFFFFFFF
ABI_REVISION = 7


def make_launch_opts(form=0, seq_out=0):
    """seq_out: address of a host word (e.g. element of a Device.pinned() array: arr.ctypes.data + 4 * i) or 0."""
    o = ScanLaunchOpts()
    o.struct_size = C.sizeof(ScanLaunchOpts)
    o.form = int(form)
    o.seq_out = int(seq_out) or None
    return o


def make_scan_bwd_jobs(jobs):
    arr = (ScanBwdJob * len(jobs))()
    for a, j in zip(arr, jobs):
        for k in ("dY", "gates", "cs", "Up", "dZ", "dzmax", "dbsum"):
            v = j.get(k, 0)
            setattr(a, k, v.ptr if isinstance(v, DeviceArray) else (v or 0))
        for k in ("lddy", "B", "T", "H", "reverse"):
            setattr(a, k, int(j[k]))
    return arr


def make_scan_jobs(jobs):
    """jobs: list of dicts with the mgr_scan_job fields (DeviceArray or int pointers)."""
    arr = (ScanJob * len(jobs))()
    for a, j in zip(arr, jobs):
        for k in ("Z", "Up", "Y", "R", "gates", "cs", "YT"):
            v = j.get(k, 0)
            setattr(a, k, v.ptr if isinstance(v, DeviceArray) else (v or 0))
        for k in ("ldy", "ldr", "B", "T", "H", "reverse", "ytb", "ldt", "yt_split"):
            setattr(a, k, int(j.get(k, 0)))
    return arr


_lib = None


class MgrError(RuntimeError):
    pass


def load_library(build_if_missing=True):
    """dlopen libmgr.so (building it with hipcc first if absent).  Raises if neither is possible."""
    global _lib
    if _lib is not None:
        return _lib
    # The HIP runtime multiplexes streams onto 4 hardware queues by default; a context here has 8 streams, and two logically
    # independent streams that share a hardware queue serialise (measured: the host->device batch copy queued behind a whole
    # training step and took the encoder stream with it, 51 instead of 42 ms/step).  Must be set before the runtime starts.
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
    if not os.path.exists(LIB_PATH) and build_if_missing:
        from . import _build
        _build.build(verbose=False)
    if not os.path.exists(LIB_PATH):
        raise MgrError("libmgr.so is missing (%s) and could not be built; there is no CPU fallback" % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    # ABI guard: this binding fills the job structs field by field - the library must have been built from the same header
    sizes = (C.c_uint * 4)()
    if lib.mgr_abi_struct_sizes(sizes) != 0 or tuple(sizes) != (C.sizeof(ScanJob), C.sizeof(ScanBwdJob), C.sizeof(ScanLaunchOpts), ABI_REVISION):
        raise MgrError("libmgr.so was built from another include/mgr.h than this binding (struct sizes / revision %s, expected %s): rebuild it"
                       % (tuple(sizes), (C.sizeof(ScanJob), C.sizeof(ScanBwdJob), C.sizeof(ScanLaunchOpts), ABI_REVISION)))
    _lib = lib
    return lib


def check(rc, lib=None):
    if rc != 0:
        lib = lib or load_library()
        raise MgrError("libmgr error %d: %s" % (rc, lib.mgr_last_error().decode(errors="replace")))


class DeviceArray:
    """A typed view of device memory owned by a Device."""

    __slots__ = ("dev", "ptr", "shape", "dtype", "nbytes", "_own")

    def __init__(self, dev, ptr, shape, dtype, own=True):
        self.dev = dev
        self.ptr = ptr
        self.shape = tuple(int(s) for s in shape)
        self.dtype = np.dtype(dtype)
        self.nbytes = int(np.prod(self.shape, dtype=np.int64)) * self.dtype.itemsize
        self._own = own

    @property
    def size(self):
        return int(np.prod(self.shape, dtype=np.int64))

    def view(self, offset_elems, shape):
        """Non-owning sub-view starting offset_elems elements in."""
        return DeviceArray(self.dev, self.ptr + offset_elems * self.dtype.itemsize, shape, self.dtype, own=False)

    def upload(self, host):
        host = np.ascontiguousarray(host, dtype=self.dtype)
        if host.size != self.size:
            raise ValueError("upload size mismatch: host %s vs device %s" % (host.shape, self.shape))
        check(self.dev.lib.mgr_h2d(self.dev.ctx, self.ptr, host.ctypes.data, self.nbytes))
        return self

    def download(self):
        out = np.empty(self.shape, self.dtype)
        check(self.dev.lib.mgr_d2h(self.dev.ctx, out.ctypes.data, self.ptr, self.nbytes))
        return out

    def zero(self):
        check(self.dev.lib.mgr_memset(self.dev.ctx, self.ptr, 0, self.nbytes))
        return self

    def free(self):
        if self._own and self.ptr:
            self.dev.lib.mgr_free(self.dev.ctx, self.ptr)
            self.ptr = 0


class Device:
    """One mgr_ctx (one GPU).  Fails loudly when no GPU / no library is available."""

    def __init__(self, index=0):
        self.lib = load_library()
        ctx = vp()
        check(self.lib.mgr_ctx_create(int(index), C.byref(ctx)), self.lib)
        self.ctx = ctx
        self.index = index
        cu = i32()
        hbm = sz()
        name = C.create_string_buffer(64)
        check(self.lib.mgr_device_info(self.ctx, C.byref(cu), C.byref(hbm), name, 64))
        self.cu_count = cu.value
        self.hbm_bytes = hbm.value
        self.name = name.value.decode()
        self._arrays = []
        self._pinned = []

    # -- memory ------------------------------------------------------------------------------------
    def empty(self, shape, dtype=np.float32):
        if isinstance(shape, int):
            shape = (shape,)
        n = int(np.prod(shape, dtype=np.int64)) * np.dtype(dtype).itemsize
        p = vp()
        check(self.lib.mgr_alloc(self.ctx, max(n, 16), C.byref(p)))
        a = DeviceArray(self, p.value, shape, dtype)
        self._arrays.append(a)
        return a

    def pinned(self, shape, dtype=np.float32):
        """Page-locked host array (numpy view) for mgr_h2d_async; freed with the Device."""
        n = int(np.prod(shape, dtype=np.int64)) * np.dtype(dtype).itemsize
        p = vp()
        check(self.lib.mgr_host_alloc(self.ctx, max(n, 16), C.byref(p)))
        self._pinned.append(p.value)
        buf = (C.c_char * n).from_address(p.value)
        return np.frombuffer(buf, dtype=dtype).reshape(shape)

    def h2d_async(self, darr, host_pinned):
        """Enqueue a copy from a pinned() array on the current stream; does not block the host."""
        check(self.lib.mgr_h2d_async(self.ctx, darr.ptr, host_pinned.ctypes.data, host_pinned.nbytes))

    def d2h_async(self, host_pinned, darr):
        """Enqueue a copy into a pinned() array on the current stream; valid after an event recorded behind it (event_sync)."""
        check(self.lib.mgr_d2h_async(self.ctx, host_pinned.ctypes.data, darr.ptr, min(host_pinned.nbytes, darr.nbytes)))

    def event_sync(self, ev):
        check(self.lib.mgr_event_sync(self.ctx, ev))

    def zeros(self, shape, dtype=np.float32):
        return self.empty(shape, dtype).zero()

    def array(self, host, dtype=None):
        host = np.asarray(host)
        dtype = dtype or host.dtype
        return self.empty(host.shape, dtype).upload(host)

    def bytes(self, nbytes):
        return self.empty((int(nbytes + 3) // 4,), np.float32)

    # -- streams / timing ------------------------------------------------------------------------
    def sync(self):
        check(self.lib.mgr_sync(self.ctx))

    def stream(self, idx):
        check(self.lib.mgr_stream_set(self.ctx, idx))

    def wait(self, waiter, waited):
        check(self.lib.mgr_stream_wait(self.ctx, waiter, waited))

    def record(self, ev):
        check(self.lib.mgr_event_record(self.ctx, ev))

    def wait_event(self, waiter, ev):
        check(self.lib.mgr_stream_wait_event(self.ctx, waiter, ev))

    def elapsed_ms(self, ev0, ev1):
        ms = C.c_float()
        check(self.lib.mgr_event_elapsed_ms(self.ctx, ev0, ev1, C.byref(ms)))
        return ms.value

    def prof_enable(self, mask):
        check(self.lib.mgr_prof_enable(self.ctx, mask))

    def prof_reset(self):
        check(self.lib.mgr_prof_reset(self.ctx))

    def prof_get(self, family):
        n = i32()
        ms = C.c_float()
        check(self.lib.mgr_prof_get(self.ctx, family, C.byref(n), C.byref(ms)))
        return n.value, ms.value

    def call(self, name, *args):
        """Invoke a C-ABI function with the ctx prepended; DeviceArrays are passed by pointer."""
        conv = [a.ptr if isinstance(a, DeviceArray) else a for a in args]
        check(getattr(self.lib, name)(self.ctx, *conv))

    def close(self):
        if self.ctx:
            for a in self._arrays:
                a.free()
            self._arrays = []
            for p in self._pinned:
                self.lib.mgr_host_free(self.ctx, p)
            self._pinned = []
            self.lib.mgr_ctx_destroy(self.ctx)
            self.ctx = None


def device_count():
    lib = load_library()
    n = i32()
    rc = lib.mgr_device_count(C.byref(n))
    return n.value if rc == 0 else 0
