"""Device engine: runs one network of the reference's family on one MI355X through the libmgr C ABI.

It owns the device-resident weights (packed layouts), the activation workspaces sized for a fixed
(B, T, Lmax), and sequences the HIP kernels for predict / train steps over the ctx's streams (the two
directions of every Bidirectional layer and the modality encoders run concurrently).  This is what
Keras' ``train_on_batch`` / ``predict_on_batch`` do for the reference (multimodal_fusion/multimodal.py:264,
multimodal_fusion/sequence_decoding.py:121); there is no CPU fallback.
"""
import collections
import ctypes as C
import math

import numpy as np

from . import _capi
from ._capi import Device, DeviceArray


def _pad4(n):
    return (n + 3) // 4 * 4


class _Arena:
    """Allocations that live and die with one Engine (the Device itself may be shared)."""

    def __init__(self, dev):
        self.dev = dev
        self.arrays = []

    def _keep(self, a):
        self.arrays.append(a)
        return a

    def empty(self, shape, dtype=np.float32):
        return self._keep(self.dev.empty(shape, dtype))

    def zeros(self, shape, dtype=np.float32):
        return self._keep(self.dev.zeros(shape, dtype))

    def bytes(self, nbytes):
        return self._keep(self.dev.bytes(nbytes))

    def pinned(self, shape, dtype=np.float32):
        return self.dev.pinned(shape, dtype)   # (tiny; released with the Device)

    def free_all(self):
        dead = set(id(a) for a in self.arrays)
        for a in self.arrays:
            a.free()
        self.dev._arrays = [a for a in self.dev._arrays if id(a) not in dead]
        self.arrays = []


class _LstmDir:
    """Device state of one direction of one Bidirectional(LSTM) layer."""

    def __init__(self, prefix, d, fin, H, trainable):
        self.prefix = prefix
        self.d = d  # "fwd" | "bwd"
        self.reverse = 1 if d == "bwd" else 0
        self.fin = fin
        self.H = H
        self.trainable = trainable
        self.Wp = self.Up = self.bp = None
        self.gWp = self.gUp = self.gbp = None
        self.mask = None       # device [4,B,fin] when input dropout is active
        self.Z = None
        self.gates = self.cs = self.dZ = self.dzmax = self.dbsum = None
        self.ws_scan = self.ws_pg = self.ws_sp = None
        self.lists_mask = 0     # the mask whose kept lists the projection of this step left in ws_sp (0: none)
        self.HsT = None         # [B][H][ldt] split rows of h_prev (the recurrent weight gradient as a product along time; first use)
        self.hst_ready = False  # ... already written for the step in flight (Engine._prep_hst)


class Schedule:
    """How one training step is laid out over the context's streams.  The default is the schedule bench.py
    and fit_generator run; the switches exist so that tests can show every layout computes the same step.

    pipeline            with frozen encoders, the encoder pass of step n+1 runs on a second stream beside step n's
                        fusion layer / CTC / BPTT / optimizer
    defer_param_grads   step n's dW / dU / db GEMMs and the optimizer are held back until the deepest projection GEMMs of
                        step n+1 are done and then run beside its deepest encoder scan (GEMM beside GEMM gains nothing)
    encoders_run_ahead  the encoder stream does not wait for the previous step as a whole: only the scan that overwrites
                        the FEAT buffer that step's dW GEMMs read waits for it
    transposed_inputs   keep a transposed copy of the inputs of the wide dropout layers for the dropout-aware projection GEMMs
    resident_wait_us    upper bound of the device-side wait that lets the deepest encoder scan become resident before the
                        deferred GEMMs are released (mgr_stream_wait_next_resident); 0 = no wait
    encoders_two_ahead  (round 5; needs bptt_beside_deepest_scan) a caller that announces TWO batches ahead (enqueue_train_step's
                        prefetch_after_next) gets the first part of the batch-after-next's encoder pass enqueued at the end of a
                        step instead of at the start of the next call, which has to wait for the step's loss
    deepest_scan_after_fusion_proj  (round 5) the next batch's deepest encoder scan is launched behind this step's fusion projections
                        instead of at the same instant
    depth1_proj_ahead   (round 5; needs encoders_two_ahead) the depth-1 projections of the batch after next are enqueued on the encoder
                        stream IN FRONT of the next batch's deepest scan - into gate pre-activation buffers of their own - and run while
                        that stream would wait for this step's fusion projections (a store-bound f32 GEMM beside a staging-bound one)
    bptt_yields_beside_scans  (round 5) a narrow layer's BPTT that the schedule puts beside the next batch's encoder scans takes the form
                        that yields to them (mgr.h, tune key 16) instead of the one trimmed along its dependent chain (faster alone,
                        costs the step 0.1 - 0.2 ms there); bit-identical either way
    bptt_direct_when_alone  (round 5; measured, not the default) beside fused encoder scans the fusion layer's BPTT takes the direct-gather
                        form (mgr.h, tune key 16 = 2: one barrier per step, the fastest form alone) - 16.64 - 16.72 against 16.60 - 16.61 ms
    fused_wide_tiles    (round 5) with fused encoder scans no GEMM shares a CU with a scan workgroup: the fusion layer's pre-split
                        products take the library's own tile choice instead of the 4-wave forms
    fused_encoder_scans (round 5; needs encoders_two_ahead) the encoder scans take the FUSED form (mgr.h, MGR_SCAN_FORM_FUSED: 8-wave
                        workgroups that hold a CU each, 208 instead of 408) and leave 48 CUs to the fusion layer's recurrences, which
                        are started once the encoder scan they run beside is resident.  Round 6: the form is an argument of each
                        launch and the launch number each wait is for is HANDED OVER by the launch itself (a page-locked word the
                        wait kernel polls, mgr_stream_wait_resident_word) - no context-wide tune key, no predicted numbers; a wait
                        that runs into its bound is counted (Engine.resident_wait_stats)
    fusion_scan_fused   (round 6; with fused_encoder_scans) the fusion layer's own forward scan takes the fused form as well
                        (MGR_SCAN_FORM_FUSED_ANY: H = 100 -> 4 eight-wave workgroups per cluster, 32 workgroups that hold a CU each,
                        instead of 56 four-wave workgroups on the 48 CUs the encoder scans leave, eight of which carry two).  The scan
                        itself is no faster (3.6 against 3.4 ms in the step); the 16 CUs it leaves free are what the head's Dense
                        kernels get (dense_bwd 0.85 -> 0.21 ms): 16.2 - 16.3 -> 15.9 - 16.0 ms per step (profiles/r06_schedule_probes.txt)
    bptt_fused          (round 6; with fused_encoder_scans) the fusion layer's BPTT takes the fused form too (MGR_BPTT_FORM_FUSED /
                        _FUSED_DIRECT with bptt_direct_when_alone: 32 eight-wave workgroups, a CU each, instead of 56 four-wave ones)
    chain_stream_priority  (round 6) dispatch priority of stream 0 against the encoder stream: 1 = stream 0 (the step's dependent chain)
                        high, -1 = the encoder stream low, 0 = equal (rounds 1 - 5).  Round 5 measured it zero-sum while both streams were
                        equally long; with the shared gather the encoder stream has slack
    param_grads_two_streams  (round 6) the dW / dU / db chains of the two directions of a Bidirectional layer run on two streams (the
                        second one: PG_STREAM) instead of one behind the other: the short kernels of one chain (lists, row maxima,
                        transposes, reductions) run under the long ones of the other
    first_pass_on_encoder_stream  (round 6; with fused_encoder_scans) a pipelined step that finds no prefetched encoder pass - the first step
                        of a run, the step after a validation pass - runs its own pass on the ENCODER stream in the fused forms and is a
                        steady-state call from there on (the next batch's pass follows at once, fused, its scans paired with this step's
                        recurrences through their launch numbers); round 5: in line on stream 0 in the plain forms, the next pass behind
                        a stream-wide wait, the fused forms only from the third step of a run on
    bptt_single_cu      (round 6; measured, not the default) narrow trainable layers (H in {32, 64, 100}) run their BPTT on ONE CU per
                        (direction, 16-sample group) without an inter-CU exchange (MGR_BPTT_FORM_SINGLE_CU, lstm_cu_bwd.hip) - in every
                        schedule (its results equal the multi-CU forms' to rounding, not bit for bit)
    du_split            (round 6) the recurrent weight gradient dU of a layer whose dW runs on pre-split rows is formed the same way - the
                        split rows of h_prev along time (one transposing pass over the layer's outputs) against the dZ^T rows the dW
                        product reads anyway - instead of by the f32 split-K product and its slab reduction
    split_rows          (round 5) the transposed copies are written in the split row format (f16 hi / lo pairs) and the wide products
                        run as loader / matrix pipelines on pre-split operands (gemm_split.hip); False: f32 rows, converted by
                        every product that reads them (round 4's kernels)
    bptt_beside_deepest_scan  (round 4; needs the three switches above) the trainable layer's BPTT is held back together with its
                        GEMMs: recurrence beside recurrence, GEMM beside GEMM.  The encoder stream then runs free - the next batch's
                        encoder pass up to its deepest projections is enqueued BEFORE this step's fusion work, and its depth-1
                        projections no longer wait for this step's fusion projections
    """

    def __init__(self, pipeline=True, defer_param_grads=True, encoders_run_ahead=True, resident_wait_us=2000,
                 transposed_inputs=True, bptt_beside_deepest_scan=True, split_rows=True, encoders_two_ahead=True,
                 deepest_scan_after_fusion_proj=True, depth1_proj_ahead=True, bptt_yields_beside_scans=True,
                 fused_encoder_scans=True, fused_wide_tiles=True, bptt_direct_when_alone=False, fusion_scan_fused=True,
                 bptt_fused=False, chain_stream_priority=0, param_grads_two_streams=False, first_pass_on_encoder_stream=True,
                 bptt_single_cu=False, du_split=True):
        self.bptt_single_cu = bool(bptt_single_cu)
        self.du_split = bool(du_split)
        self.first_pass_on_encoder_stream = bool(first_pass_on_encoder_stream)
        self.param_grads_two_streams = bool(param_grads_two_streams)
        self.chain_stream_priority = int(chain_stream_priority)
        self.bptt_fused = bool(bptt_fused)
        self.fusion_scan_fused = bool(fusion_scan_fused)
        self.bptt_direct_when_alone = bool(bptt_direct_when_alone)
        self.fused_wide_tiles = bool(fused_wide_tiles)
        self.fused_encoder_scans = bool(fused_encoder_scans)
        self.bptt_yields_beside_scans = bool(bptt_yields_beside_scans)
        self.depth1_proj_ahead = bool(depth1_proj_ahead)
        self.deepest_scan_after_fusion_proj = bool(deepest_scan_after_fusion_proj)
        self.encoders_two_ahead = bool(encoders_two_ahead)
        self.split_rows = bool(split_rows)
        self.bptt_beside_deepest_scan = bool(bptt_beside_deepest_scan)
        self.transposed_inputs = bool(transposed_inputs)
        self.pipeline = bool(pipeline)
        self.defer_param_grads = bool(defer_param_grads)
        self.encoders_run_ahead = bool(encoders_run_ahead)
        self.resident_wait_us = int(resident_wait_us)


class Engine:
    def __init__(self, spec, B, T, Lmax, device=0, seed=1234, comm=None, world=1, inference_only=False, schedule=None):
        self.spec = spec
        self.B, self.T, self.Lmax = int(B), int(T), int(Lmax)
        cdev = getattr(comm, "dev", None)
        if cdev is not None:
            # the all-reduce is enqueued on the communicator's context: it must be THIS engine's context, or it would not be
            # ordered against the gradient kernels and the optimizer
            if isinstance(device, Device) and device is not cdev:
                raise ValueError("the communicator was created on another Device than the engine's")
            device = cdev
        self._own_dev = not isinstance(device, Device)
        self.dev = device if isinstance(device, Device) else Device(device)
        self.schedule = schedule or Schedule()
        self.mem = _Arena(self.dev)           # device buffers owned by this engine
        self.lib = self.dev.lib
        self.seed = int(seed)
        self.comm = comm
        self.world = int(world)
        self.inference_only = inference_only
        if self.schedule.chain_stream_priority:
            # (a property of the CONTEXT's streams: engines that share a Device share it; the call waits for the stream to be idle)
            if self.schedule.chain_stream_priority > 0:
                self.dev.call("mgr_stream_set_priority", 0, 1)
            else:
                self.dev.call("mgr_stream_set_priority", self.ES, -1)
        self._adam_calls = 0  # optimizer steps enqueued; `iterations` (Keras) = those the update gate did not skip
        self.rng_step = 0     # advances per forward pass that draws randomness
        self._build()

    # ------------------------------------------------------------------------------------------ build
    def _build(self):
        sp, dev, B, T = self.spec, self.mem, self.B, self.T
        table = sp.weight_table()
        # flat trainable buffer layout (every segment padded to 4 floats)
        self.seg = {}
        off = 0
        for name, shape, tr, kind in table:
            n = int(np.prod(shape))
            if tr:
                self.seg[name] = (off, n, shape, kind)
                off += _pad4(n)
        self.n_train = off
        self.params = dev.zeros((max(off, 4),))
        # this engine's own scan-status block (several engines may share one Device): [0] status bits, [2] skipped updates
        self.status = dev.zeros((16,), np.uint32)
        # inference passes (predict / loss_on_batch / predict_stream) report into a block of their OWN per pass (two: batches of a
        # pipelined run alternate), so that one batch with a NaN input marks ITS samples and nothing else (words [8, 16): mgr.h)
        self._pass_status = [dev.zeros((16,), np.uint32) for _ in range(2)]
        self.loss_host = dev.pinned((4,), np.float32)
        self.status_host = dev.pinned((4,), np.uint32)
        # launch numbers handed from a persistent scan launch to the residency wait that was enqueued before it (page-locked: the
        # wait kernel polls the word; mgr_stream_wait_resident_word).  A ring: two words per step, reused eight steps later.
        self._seq_words = dev.pinned((16,), np.uint32)
        self._seq_words[...] = 0
        self._seq_next = 0
        # (diagnostic, bounded) per fused step: (step id, kind, the words its two residency waits poll) - tools/startup_probe.py
        self._gate_log = collections.deque(maxlen=16)
        # data parallel: [gate flag, sum over ranks of the local mean losses, -, -] of step s in slot s & 1, copied from behind the
        # all-reduced gradient buffer (apply_gradients); read by read_global_loss
        self.gloss_host = [dev.pinned((4,), np.float32) for _ in range(2)]
        self._gloss_step = [-1, -1]
        if not self.inference_only:
            # the 4 floats behind the gradients ride through the gradient all-reduce (apply_gradients): [0] the update-gate flag,
            # [1] this rank's mean loss of the step ("piggy-backed as one extra float")
            self.grads = dev.zeros((max(off, 4) + 4,))
            self.gate_flag = self.grads.view(max(off, 4), (4,))
            self.loss_slot = self.grads.view(max(off, 4) + 1, (1,))
            self.m = dev.zeros((max(off, 4),))
            self.v = dev.zeros((max(off, 4),))
        self.frozen = {}
        for name, shape, tr, kind in table:
            if not tr:
                self.frozen[name] = dev.zeros((_pad4(int(np.prod(shape))),))
        self.kinds = {name: kind for name, _, _, kind in table}
        self.shapes = {name: shape for name, shape, _, _ in table}

        # LSTM layer-direction objects
        self.dirs = {}
        train = not self.inference_only
        for prefix, fin, H, p, tr in sp.lstm_layers():
            for d in ("fwd", "bwd"):
                L = _LstmDir(prefix, d, fin, H, tr)
                base = "%s/%s" % (prefix, d)
                L.Wp, L.Up, L.bp = (self._wview(base + "/W"), self._wview(base + "/U"), self._wview(base + "/b"))
                if tr and train:
                    L.gWp, L.gUp, L.gbp = (self._gview(base + "/W"), self._gview(base + "/U"), self._gview(base + "/b"))
                    L.gates = dev.empty((B, T, H, 4))
                    L.cs = dev.empty((B, T, H))
                    L.dZ = dev.empty((B, T, 4 * H))
                    L.dzmax = dev.zeros((B, 4 * H), np.uint32)   # row maxima of dZ^T, left by the BPTT (mgr_scan_bwd_job.dzmax)
                    L.dbsum = dev.zeros((B, 4 * H))              # and its sums of dZ over time: db without a pass over dZ
                    L.ws_scan = dev.bytes(self.lib.mgr_lstm_scan_ws_bytes(B, T, H))
                    need = (self.lib.mgr_lstm_param_grads_dropout_ws_bytes(B, T, fin, H) if p > 0
                            else self.lib.mgr_lstm_param_grads_ws_bytes(B, T, fin, H))
                    if p > 0 and self.schedule.transposed_inputs:
                        # (+ the transposed dZ of the dropout-aware dW from the transposed activation copy: sized here, not by a
                        # hipMalloc in the middle of the first training step)
                        need = max(need, self.lib.mgr_lstm_param_grads_dropout_t_ws_bytes(B, T, fin, H, (T + 127) // 128 * 128))
                        if self._ts_shape(fin):
                            need = max(need, self.lib.mgr_lstm_param_grads_dropout_ts_ws_bytes(B, T, fin, H, (T + 127) // 128 * 128))
                    L.ws_pg = dev.bytes(need)
                if p > 0:
                    L.mask = dev.empty((4, B, fin))
                    need = self.lib.mgr_lstm_input_proj_dropout_ws_bytes(B, fin, H)                 # kept-feature lists, weight copies
                    if self._ts_shape(fin):
                        need = max(need, self.lib.mgr_lstm_input_proj_dropout_ts_ws_bytes(B, fin, H))
                    L.ws_sp = dev.bytes(need)
                L.p = p
                self.dirs[base] = L

        # activations
        W = sp.concat_width
        self.X = {}
        self.Y1 = {}
        self.Y2 = {}
        self.dY1 = {}
        self.Zbuf = {}
        self.Z1buf = {}
        self._xcur = {}
        # host batches arrive through a dedicated copy stream into two alternating input / label buffer sets, so an upload
        # never has to wait for (or stall the host behind) whatever the compute streams still have queued
        self._xin_ring = [{}, {}]
        self._xin_slot = 0
        self._xin_pin = None
        self._xin_user = [-1, -1]      # id of the last step that reads input set 0 / 1 (see _upload_inputs)
        self._xin_copied = [False, False]   # EV_XIN_COPIED / EV_LAB_COPIED of set 0 / 1 have been recorded
        self._lab_copied = [False, False]
        self._lab_user = [-1, -1]
        self._step_id = 0              # training steps enqueued so far
        self._synced_step = -1         # the host has seen the loss of this step (everything before its CTC is complete)
        for s in sp.streams:
            for ring in self._xin_ring:
                ring[s["name"]] = dev.empty((B, T, s["F"]))
            if s["noise"] > 0:
                self.X[s["name"]] = dev.empty((B, T, s["F"]))
            self.Xin = self._xin_ring[0]
            Hs = [lay["H"] for lay in s["layers"]]
            zf = dev.empty((B, T, 4 * max(Hs)))
            zb = dev.empty((B, T, 4 * max(Hs)))
            self.Zbuf[s["name"]] = (zf, zb)
            if train and self.can_pipeline and self.schedule.depth1_proj_ahead and len(Hs) == 2:
                # (Schedule.depth1_proj_ahead: the depth-1 projections of a batch are written while the previous batch's deepest scan
                #  still reads the shared buffers)
                self.Z1buf[s["name"]] = (dev.empty((B, T, 4 * Hs[0])), dev.empty((B, T, 4 * Hs[0])))
            if len(Hs) == 2:
                self.Y1[s["name"]] = dev.empty((B, T, 2 * Hs[0]))
                if s["trainable"] and train:
                    if s["residual"]:
                        self.Y2[s["name"]] = dev.empty((B, T, 2 * Hs[1]))
                    self.dY1[s["name"]] = dev.empty((B, T, 2 * Hs[0]))
        self.FEAT = dev.empty((B, T, W))
        self._feat_ring = [self.FEAT]
        if (train and self.can_pipeline) or self.inference_only:
            # second FEAT buffer: cross-step pipelining of training (frozen encoders) / batch pipelining of inference
            self._feat_ring.append(dev.empty((B, T, W)))
        # Transposed copies [B, features, T padded to 128] of the inputs of the WIDE dropout layers (depth-2 encoder layers, the
        # fusion layer): the dropout-aware projection gathers kept FEATURES, which are contiguous rows there (gemm.hip,
        # k_gemm_nn_sparse<.., true>).  Row-major stays what everything else reads (dW GEMMs, residual adds, dense layer).
        self.ldt = (T + 127) // 128 * 128
        self.Y1T = {}
        self._featT = {}
        self._featT_ready = {}   # FEAT buffer -> its transposed copy was written by the scans of the current pass
        self._xt_split = {}      # transposed copy (device pointer) -> its rows are in the split row format (as last written)
        if self.schedule.transposed_inputs:   # (training: dropout-aware GEMMs; inference: the dense split-f16 projection reads it too)
            want = lambda p, F: bool(self.lib.mgr_lstm_input_proj_dropout_wants_transposed(self.dev.ctx, C.c_float(float(p)), int(F)))
            for s in sp.streams:
                if len(s["layers"]) == 2 and want(s["layers"][1]["dropout"], 2 * s["layers"][0]["H"]):
                    self.Y1T[s["name"]] = dev.zeros((B, 2 * s["layers"][0]["H"], self.ldt))
            if sp.fusion and want(sp.fusion["dropout"], W):
                for fb in self._feat_ring:
                    self._featT[fb.ptr] = dev.zeros((B, W, self.ldt))
        self._feat_idx = 0
        self._prefetched = None
        self._prefetched_for = None
        self._masks = {}
        any_tr_stream = any(s["trainable"] for s in sp.streams)
        if sp.fusion:
            Hf = sp.fusion["H"]
            self.ZF = (dev.empty((B, T, 4 * Hf)), dev.empty((B, T, 4 * Hf)))
            self.YF = dev.empty((B, T, 2 * Hf))
            if train:
                self.dYF = dev.empty((B, T, 2 * Hf))
        if train and any_tr_stream:
            self.dFEAT = dev.empty((B, T, W))
        D, Cn = sp.head_width, sp.num_classes
        self.P = dev.empty((B, T, Cn))
        self.head_mask = None
        if train:
            self.dLogits = dev.empty((B, T, Cn))
            self.loss_b = dev.empty((B,))
            self.loss_mean = dev.empty((4,))
            self._lab_ring = [(dev.empty((B, self.Lmax), np.int32), dev.empty((B,), np.int32), dev.empty((B,), np.int32))
                              for _ in range(2)]
            self._lab_slot = 0
            self.labels_d, self.ilen_d, self.llen_d = self._lab_ring[0]
            # page-locked staging for the (tiny) label arrays: their copies are enqueued without blocking the host
            self._lab_pin = [(dev.pinned((B, self.Lmax), np.int32), dev.pinned((B,), np.int32), dev.pinned((B,), np.int32))
                             for _ in range(2)]
            self.ws_ctc = dev.bytes(self.lib.mgr_ctc_ws_bytes(B, T, Cn, self.Lmax))
            self.ws_dense = dev.bytes(self.lib.mgr_dense_bwd_ws_bytes(B, T, D, Cn))
            self.ws_head = dev.bytes(self.lib.mgr_head_ws_bytes(B, T, D, Cn, self.Lmax))
        self.dev.sync()

    def _wview(self, name):
        if name in self.seg:
            off, n, shape, _ = self.seg[name]
            return self.params.view(off, (n,))
        return self.frozen[name]

    def _gview(self, name):
        off, n, shape, _ = self.seg[name]
        return self.grads.view(off, (n,))

    # ------------------------------------------------------------------------------------------ weights
    def _bind(self, block=None):
        """Scans enqueued from here on report into THIS engine's status block (the Device may be shared) - or into `block`."""
        self.dev.call("mgr_scan_status_bind", self.status if block is None else block)

    def _begin_pass(self, slot=0):
        """An inference pass gets a zeroed status block of its own (current stream): what its scans report - a give-up, WHICH
        samples met a non-finite state - belongs to this pass only, whatever earlier passes or the training loop recorded."""
        blk = self._pass_status[slot]
        blk.zero()
        self._bind(blk)
        return blk

    def _end_pass(self, blk_words, out, what="inference pass"):
        """blk_words: the 16 words of the pass's status block on the host (read behind the pass).  Raises on a give-up; returns
        `out` with NaN in the rows of exactly the samples whose hidden state went NaN / Inf in THIS pass (the multi-CU exchange
        feeds 0 back for such a state, so other units of the sample may look finite where the reference's whole sample is NaN)."""
        self._bind()
        bits = int(blk_words[0])
        if bits & ~_capi.SCAN_NONFINITE:
            raise _capi.MgrError("a persistent scan of this %s gave up on a bounded spin (status %d): its outputs are invalid" % (what, bits))
        if not bits & _capi.SCAN_NONFINITE:
            return out
        field = np.asarray(blk_words[8:16], np.uint32)
        bad = np.array([bool((int(field[(b & 255) >> 5]) >> (b & 31)) & 1) for b in range(self.B)])
        return self._nan_rows(out, bad)

    def _nan_rows(self, out, bad):
        """NaN (floats) / -1 (integer label outputs) / empty paths in the rows of the samples marked in `bad`."""
        if isinstance(out, tuple):
            return tuple(self._nan_rows(o, bad) for o in out)
        if isinstance(out, list) and len(out) == len(bad):            # decoded label sequences: no labels for an invalid sample
            return [[] if bad[b] else o for b, o in enumerate(out)]
        if isinstance(out, np.ndarray) and out.shape[:1] == bad.shape:
            out = out.copy()
            out[bad] = np.nan if out.dtype.kind == "f" else -1
        return out

    def set_weights(self, weights):
        """weights: dict name -> numpy array in Keras layout (see NetworkSpec.weight_table).  Fresh weights start with a clean
        scan status: a NaN / give-up recorded under the old ones (diverged run) must not haunt a restored checkpoint."""
        dev = self.dev
        dev.stream(0)
        self.clear_scan_status()
        for name, shape, tr, kind in self.spec.weight_table():
            if name not in weights:
                continue
            w = np.ascontiguousarray(np.asarray(weights[name], dtype=np.float32))
            if tuple(w.shape) != tuple(shape):
                raise ValueError("weight %s: expected shape %s, got %s" % (name, shape, w.shape))
            dst = self._wview(name)
            if kind in ("kernel", "recurrent", "bias"):
                H = shape[-1] // 4
                rows = 1 if kind == "bias" else shape[0]
                tmp = dev.array(w.reshape(-1))
                dev.call("mgr_lstm_pack", tmp, dst, rows, H, 0)
                dev.sync()
                tmp.free()
            else:
                dst.view(0, (w.size,)).upload(w.reshape(-1))
        dev.sync()
        self._freeze_planes()

    def _freeze_planes(self):
        """The input weights of FROZEN layers (the encoders of the fusion network) are rewritten by set_weights only: the library may
        keep their split (hi, lo) planes in the projection workspaces from call to call (mgr_weight_planes_cache; every call drops
        what was kept, so this runs after every rewrite)."""
        for L in self.dirs.values():
            if L.ws_sp is not None and (L.prefix + "/" + L.d + "/W") in self.frozen:
                self.dev.call("mgr_weight_planes_cache", L.Wp, 1)

    def get_weights(self):
        dev = self.dev
        dev.stream(0)
        out = {}
        for name, shape, tr, kind in self.spec.weight_table():
            n = int(np.prod(shape))
            src = self._wview(name)
            if kind in ("kernel", "recurrent", "bias"):
                H = shape[-1] // 4
                rows = 1 if kind == "bias" else shape[0]
                tmp = dev.empty((n,))
                dev.call("mgr_lstm_pack", src, tmp, rows, H, 1)
                out[name] = tmp.download().reshape(shape)
                tmp.free()
            else:
                out[name] = src.view(0, (n,)).download().reshape(shape)
        return out

    def get_grads(self):
        """Trainable gradients of the last train step's backward pass, Keras layouts (tests / debugging)."""
        dev = self.dev
        dev.stream(0)
        out = {}
        for name, (off, n, shape, kind) in self.seg.items():
            src = self.grads.view(off, (n,))
            if kind in ("kernel", "recurrent", "bias"):
                H = shape[-1] // 4
                rows = 1 if kind == "bias" else shape[0]
                tmp = dev.empty((n,))
                dev.call("mgr_lstm_pack", src, tmp, rows, H, 1)
                out[name] = tmp.download().reshape(shape)
                tmp.free()
            else:
                out[name] = src.download().reshape(shape)
        return out

    def reset_optimizer(self):
        self.m.zero()
        self.v.zero()
        self.clear_scan_status()
        self.iterations = 0

    # ------------------------------------------------------------------------------------------ helpers
    def _seed(self, slot):
        return (self.seed * 1000003 + self.rng_step * 131 + slot) & 0xFFFFFFFFFFFFFFFF

    COPY_STREAM = 7
    EV_IN = (41, 42)     # last reader of input buffer set 0 / 1
    EV_LAB = (43, 44)    # last reader of label buffer set 0 / 1
    EV_XIN_COPIED = (59, 60)   # the host-to-device copies out of input staging set 0 / 1 are done (the HOST waits for it before it
    EV_LAB_COPIED = (61, 62)   # refills that page-locked set; ... label staging set 0 / 1)

    def _upload_inputs(self, inputs, rand, train, stream=0):
        """Host batch -> the input buffer set that is NOT the one read last, on the copy stream; `stream` (where the
        encoder pass that reads it will be enqueued) waits for the copies."""
        dev = self.dev
        slot = self._xin_slot ^ 1
        dev.stream(self.COPY_STREAM)
        # the set was last read by step _xin_user[slot]; once the host has read a loss at or after that step the readers
        # are known to be complete (stream order) and no device-side wait is needed; otherwise wait for the readers' event
        if self._xin_user[slot] > self._synced_step:
            dev.wait_event(self.COPY_STREAM, self.EV_IN[slot])
        if self._xin_pin is None:   # page-locked staging, allocated at the first host batch (resident-input runs never need it)
            self._xin_pin = [{s["name"]: dev.pinned((self.B, self.T, s["F"]), np.float32) for s in self.spec.streams}
                             for _ in range(2)]
        # The page-locked staging set is refilled by the HOST: the copies enqueued out of it last time must be done.  (Round 6: a call
        # of the two-calls-ahead schedule that also runs its own encoder pass uploads THREE batches - its own, the next, the one after
        # next - and the third refill met the first copy still in flight: a few input values of the step changed, a loss moved by 1e-6,
        # one run in three.  Found by tests/test_gpu_schedule_contract.py.)
        if self._xin_copied[slot]:
            dev.event_sync(self.EV_XIN_COPIED[slot])
        for s in self.spec.streams:
            x = np.asarray(inputs[s["name"]])
            if x.shape != (self.B, self.T, s["F"]):
                raise ValueError("input %s: expected %s got %s" % (s["name"], (self.B, self.T, s["F"]), x.shape))
            stage = self._xin_pin[slot][s["name"]]
            np.copyto(stage, x, casting="unsafe")          # float64 batch -> float32 staging in one pass
            nz = rand.get(s["name"] + "/noise") if rand else None
            if train and nz is not None:
                stage += np.asarray(nz, dtype=np.float32)
            dev.h2d_async(self._xin_ring[slot][s["name"]], stage)   # the host does not wait for the copy
        dev.record(self.EV_XIN_COPIED[slot])
        self._xin_copied[slot] = True
        dev.wait(stream, self.COPY_STREAM)
        self._xin_slot = slot
        self._xin_user[slot] = 1 << 60     # set by the step that consumes it (enqueue_train_step)
        self.Xin = self._xin_ring[slot]
        dev.stream(stream)

    # Bound on the transposed activation copies the engine keeps (Y1T, FEAT^T), in the convention of mgr.h's x_absmax: NEGATIVE =
    # guaranteed by the producer, not checked.  Every such copy holds outputs of LSTM layers of this library - h = o * tanh(c) with
    # o in [0, 1], so |h| <= 1 - or the residual sum of two of them (encoder stacks, multimodal.py:111,117): <= 2 by construction,
    # whether the scans wrote the copy themselves or mgr_transpose_bt made it from such a buffer.  A diverged state is NaN, not large.
    XT_BOUND = -2.0
    _fmt_train = True      # the learning phase of the pass being enqueued (decides the row format of new transposed copies)

    @staticmethod
    def _ts_shape(fin):
        """Layer widths the pre-split products handle (mgr_lstm_input_proj_dropout_ts / mgr_lstm_param_grads_dropout_ts)."""
        return 128 <= fin <= 2048

    def _split_rows_wanted(self):
        """The format new transposed activation copies are written in: split rows (f16 hi / lo pairs, mgr.h) for the pre-split
        products on the f16 matrix pipe - unless tune key 15 keeps the GEMMs on their f32 MFMA kernels (bench.py's second leg)."""
        if not self.schedule.split_rows or self.inference_only or not self._fmt_train:
            # (learning phase 0 has no dropout mask: ONE dense K loop stages the A tile once for the four gates - k_gemm_nn_dense16 on
            #  f32 rows; the pre-split kernel would stage it once per gate: 20.3 against 19.4 ms per pipelined batch.  By PHASE, not by
            #  engine: predict on a training engine and on an inference engine give the same bits)
            return False
        v = C.c_int()
        self.dev.call("mgr_tune_get", 15, C.byref(v))
        return v.value == 0

    def _make_xt(self, X, ldx, XT, B, T, fin):
        """Transposed copy of a row-major activation buffer the scans did not write themselves, in the format wanted now."""
        split = self._split_rows_wanted() and self._ts_shape(fin)
        self.dev.call("mgr_transpose_bt_split" if split else "mgr_transpose_bt", X, ldx, XT, self.ldt, B, T, fin)
        self._xt_split[XT.ptr] = split

    def _project_pair(self, X, ldx, pair, Ls, B, T, fin, H, XT=None, xt_ready=False):
        """Input projections of the two directions of one Bidirectional layer (pair = [mask, Wp, bp, Z] x 2).  With input
        dropout active the library runs each direction's K loops over the kept features only (from p >= 0.3 on) - from the
        transposed copy XT of the input where the engine keeps one (pre-split rows: the loader / matrix pipeline of gemm_split.hip;
        f32 rows: the kernels of gemm.hip); otherwise both directions go through one call that fuses them into one GEMM where that
        saves tiles."""
        masked = bool(pair[0]) and Ls[0].ws_sp is not None
        unmasked_wide = not pair[0] and XT is not None and 128 <= fin <= 2048 and Ls[0].ws_sp is not None
        if XT is not None and (masked or unmasked_wide) and not xt_ready:   # (xt_ready: the scans that produced X wrote XT themselves, mgr_scan_job.YT)
            self._make_xt(X, ldx, XT, B, T, fin)
        split = XT is not None and self._xt_split.get(XT.ptr, False)
        if masked or unmasked_wide:
            # (no dropout - inference - on a wide layer whose input the engine keeps transposed: the same kernels without a mask)
            for d in range(2):
                m, Wp, bp, Z = pair[4 * d:4 * d + 4]
                ws = Ls[d].ws_sp
                p = float(Ls[d].p) if masked else 0.0
                if split:
                    self.dev.call("mgr_lstm_input_proj_dropout_ts", XT, self.ldt, m, p, Wp, bp, Z, B, T, fin, H, ws, ws.nbytes)
                    Ls[d].lists_mask = m if masked else 0     # (the kept lists of this mask now sit in ws: the dW product of the step reuses them)
                elif XT is not None:
                    self.dev.call("mgr_lstm_input_proj_dropout_t", XT, self.ldt, m, p, Wp, bp, Z, B, T, fin, H, ws, ws.nbytes, self.XT_BOUND)
                else:
                    self.dev.call("mgr_lstm_input_proj_dropout", X, ldx, m, p, Wp, bp, Z, B, T, fin, H, ws, ws.nbytes)
        else:
            self.dev.call("mgr_lstm_input_proj_pair", X, ldx, *pair, B, T, fin, H)

    def _prep_mask(self, L, train, rand, slot):
        """Returns the device pointer (or 0) of the [4,B,fin] input-dropout mask for this pass."""
        if not train or L.p <= 0:
            return 0
        key = "%s/%s/mask" % (L.prefix, L.d)
        if rand is not None:
            if key in rand and rand[key] is not None:
                L.mask.upload(np.asarray(rand[key], dtype=np.float32))
                return L.mask.ptr
            return 0  # explicit randomness given but no mask for this layer: no dropout
        self.dev.call("mgr_dropout_mask", L.mask, L.mask.size, float(L.p), C.c_uint64(self._seed(slot)))
        return L.mask.ptr

    # ------------------------------------------------------------------------------------------ forward
    def _forward(self, train, rand):
        """Encoders + fusion + head, all on stream 0 (predict / parity / non-pipelined training)."""
        if self._prefetched is not None:   # a pipelined encoder pass is in flight: let it finish, then discard it
            self.dev.wait(0, self.ES)
            self._prefetched = None
        self._enqueue_encoders(train, rand, self.FEAT, 0, self.rng_step)
        self._enqueue_fusion_head(train, rand, self.FEAT, self.rng_step)
        if train:
            self.rng_step += 1

    def _enqueue_encoders(self, train, rand, feat_buf, es, rng_step, hold_scans_for=None):
        """Noise + every encoder depth (input-projection GEMMs, then all recurrences of the depth in one multi-scan
        call), written into feat_buf.  Everything is enqueued on stream `es`."""
        for tag, k in self._encoder_phases(train, rand, feat_buf, es, rng_step):
            if tag != "projected":
                continue
            if hold_scans_for is not None and k >= 1:
                # pipelined with another stream: a persistent cluster launch that starts while chip-filling GEMMs of the
                # other stream are draining gets a lopsided workgroup placement for its whole life (measured 23 ms
                # instead of 11 ms), so the deep scans wait for that stream's queued work first
                self.dev.stream(es)
                self.dev.wait(es, hold_scans_for)

    def _encoder_phases(self, train, rand, feat_buf, es, rng_step, first_stream=None, z_first=None, seq_words=None):
        """Generator form of the encoder pass: yields ("projected", k) after the projection GEMMs of depth k are enqueued
        (before its scan) and ("scanned", k) after its multi-scan launch, so that a caller can interleave work of another
        stream at those points.  Re-selects stream `es` after every resume; self.rng_step is only switched to `rng_step`
        while the generator body runs.
        first_stream / z_first (pipelined inference): the noise kernels and the depth-1 projections are enqueued on stream
        `first_stream` instead, into the gate pre-activation buffers z_first[name] = (fwd, bwd) instead of the stream's shared
        Zbuf - the caller orders `es` behind them before it resumes the generator.
        seq_words: dict {depth k: address of a host word} - the scan launch of depth k hands its launch number to that word (looked
        up when the launch is enqueued, so a caller may fill the dict while the generator is suspended).  The FORM of each scan
        launch is self._enc_scan_form at the moment it is enqueued (MGR_SCAN_FORM_*: the schedule's choice for that launch)."""
        sp, dev, B, T = self.spec, self.dev, self.B, self.T
        W = sp.concat_width
        save = train and not self.inference_only
        self._fmt_train = bool(train)
        saved_step, self.rng_step = self.rng_step, rng_step
        slot = 0
        cols = {}
        col = 0
        for s in sp.streams:
            cols[s["name"]] = col
            col += sp.stream_width(s)
        dev.stream(es if first_stream is None else first_stream)
        # GaussianNoise (K1) per stream
        for si, s in enumerate(sp.streams):
            name = s["name"]
            X = self.Xin[name]
            if train and rand is None and s["noise"] > 0:
                # on device; the resident input stays pristine for the next step
                dev.call("mgr_add_gaussian_noise", X, self.X[name], X.size, float(s["noise"]),
                         C.c_uint64(self._seed(900 + si)))
                X = self.X[name]
            self._xcur[name] = X
        depth = max(len(s["layers"]) for s in sp.streams)
        feat_by_scans = True     # every stream's last layer writes FEAT (and its transposed copy) from its scan
        split_now = self._split_rows_wanted()   # the format of the transposed copies this pass's scans write
        for k in range(depth):
            jobs = []
            for si, s in enumerate(sp.streams):
                nl = len(s["layers"])
                if k >= nl:
                    continue
                name = s["name"]
                H = s["layers"][k]["H"]
                last = k == nl - 1
                col = cols[name]
                if k == 0:
                    cur, ldcur, fin = self._xcur[name], s["F"], s["F"]
                else:
                    Hp = s["layers"][k - 1]["H"]
                    cur, ldcur, fin = self.Y1[name], 2 * Hp, 2 * Hp
                pair, Ls_pair = [], []
                for di, dname in enumerate(("fwd", "bwd")):
                    L = self.dirs["%s/l%d/%s" % (name, k, dname)]
                    Ls_pair.append(L)
                    slot += 1   # (every projection GEMM fills the chip: one stream keeps their timings honest)
                    mptr = self._prep_mask(L, train, rand, slot)
                    self._masks[(L.prefix, L.d)] = mptr
                    zb = z_first[name] if (k == 0 and z_first is not None) else self.Zbuf[name]
                    pair += [mptr, L.Wp, L.bp, zb[di]]
                self._project_pair(cur, ldcur, pair, Ls_pair, B, T, fin, H, XT=self.Y1T.get(name) if k == 1 else None,
                                   xt_ready=True)
                for di, dname in enumerate(("fwd", "bwd")):
                    L = self.dirs["%s/l%d/%s" % (name, k, dname)]
                    Z = (z_first[name] if (k == 0 and z_first is not None) else self.Zbuf[name])[di]
                    R, ldr = 0, 0
                    YT, ytb = 0, 0      # transposed copy written by the scan itself (what the next dropout layer's GEMMs read)
                    yt_fmt = 0
                    if not last:
                        Y, ldy = self.Y1[name].view(di * H, (1,)), 2 * H
                        if name in self.Y1T:
                            YT, ytb = self.Y1T[name].ptr + di * H * self.ldt * 4, 2 * H * self.ldt
                            yt_fmt = int(split_now and self._ts_shape(2 * H))
                            self._xt_split[self.Y1T[name].ptr] = bool(yt_fmt)
                    elif nl == 2 and s["residual"] and name in self.Y2 and save:
                        Y, ldy = self.Y2[name].view(di * H, (1,)), 2 * H
                        feat_by_scans = False
                    else:
                        Y, ldy = feat_buf.view(col + di * H, (1,)), W
                        if nl == 2 and s["residual"]:
                            R, ldr = self.Y1[name].view(di * H, (1,)), 2 * H
                        if feat_buf.ptr in self._featT:
                            YT, ytb = self._featT[feat_buf.ptr].ptr + (col + di * H) * self.ldt * 4, W * self.ldt
                            yt_fmt = int(split_now and self._ts_shape(W))
                            self._xt_split[self._featT[feat_buf.ptr].ptr] = bool(yt_fmt)
                    keep = save and L.trainable
                    jobs.append(dict(Z=Z, Up=L.Up, Y=Y, ldy=ldy, R=R, ldr=ldr, gates=L.gates if keep else 0,
                                     cs=L.cs if keep else 0, B=B, T=T, H=H, reverse=L.reverse, YT=YT, ytb=ytb,
                                     ldt=self.ldt if YT else 0, yt_split=yt_fmt))
            if k == 0:
                dev.record(self.EV_IN[self._xin_slot])   # the inputs have been read (noise kernel / depth-1 projections)
            self.rng_step = saved_step
            yield ("projected", k)
            saved_step, self.rng_step = self.rng_step, rng_step
            self._fmt_train = bool(train)
            dev.stream(es)
            # all recurrences of this depth in ONE call (one persistent multi-CU launch when H is large)
            self._scan_multi(jobs, "_ws_multi", form=self._enc_scan_form, seq_word=(seq_words or {}).get(k, 0))
            self.rng_step = saved_step
            yield ("scanned", k)
            saved_step, self.rng_step = self.rng_step, rng_step
            self._fmt_train = bool(train)
            dev.stream(es)
        self._featT_ready[feat_buf.ptr] = feat_by_scans
        for si, s in enumerate(sp.streams):
            name = s["name"]
            if len(s["layers"]) == 2 and s["residual"] and name in self.Y2 and save:
                H = s["layers"][-1]["H"]
                dev.call("mgr_add2d", self.Y1[name], 2 * H, self.Y2[name], 2 * H, feat_buf.view(cols[name], (1,)), W,
                         B * T, 2 * H)
        self.rng_step = saved_step

    def _enqueue_fusion_head(self, train, rand, feat_buf, rng_step, dense=True):
        """Fusion BiLSTM (projection GEMMs + recurrences) and Dropout/Dense/softmax on stream 0 (dense=False: the caller runs
        the Dense layer itself - the training step's mgr_head_fwd_bwd)."""
        sp, dev, B, T = self.spec, self.dev, self.B, self.T
        W = sp.concat_width
        save = train and not self.inference_only
        self._fmt_train = bool(train)
        saved_step, self.rng_step = self.rng_step, rng_step
        dev.stream(0)
        feat, ldf = feat_buf, W
        self._featin = feat_buf
        if sp.fusion:
            Hf = sp.fusion["H"]
            jobs = []
            pair, Ls_pair = [], []
            for di, dname in enumerate(("fwd", "bwd")):
                L = self.dirs["fusion/%s" % dname]
                Ls_pair.append(L)
                mptr = self._prep_mask(L, train, rand, 500 + di)
                self._masks[(L.prefix, L.d)] = mptr
                pair += [mptr, L.Wp, L.bp, self.ZF[di]]
            # (pipelined training: these GEMMs run while the next batch's deepest encoder scan holds the chip - the forms of the
            # pre-split products that fit on a CU beside a scan workgroup, not the 8-wave ones that wait for the scan to end)
            with self._narrow_tiles(self._beside_scans and not self._wide_ok):
                self._project_pair(feat_buf, W, pair, Ls_pair, B, T, W, Hf, XT=self._featT.get(feat_buf.ptr),
                                   xt_ready=self._featT_ready.get(feat_buf.ptr, False))
            dev.record(self.EV_FPROJ)   # (the next step's depth-1 scan is launched after these GEMMs, _enqueue_next_encoders)
            if self._gate_words[0] is not None:
                # (Schedule.fused_encoder_scans: the fusion scan's 56 workgroups must land on the CUs the next batch's deepest encoder
                #  scan - 208 whole CUs, released by the same event - leaves free, not on 56 CUs of their own.  That scan is enqueued
                #  LATER in this call: it hands its launch number to the word this wait polls)
                dev.call("mgr_stream_wait_resident_word", int(self._gate_words[0]), int(self.schedule.resident_wait_us))
            for di, dname in enumerate(("fwd", "bwd")):
                L = self.dirs["fusion/%s" % dname]
                jobs.append(dict(Z=self.ZF[di], Up=L.Up, Y=self.YF.view(di * Hf, (1,)), ldy=2 * Hf, R=0, ldr=0,
                                 gates=L.gates if save else 0, cs=L.cs if save else 0, B=B, T=T, H=Hf,
                                 reverse=L.reverse))
            self._scan_multi(jobs, "_ws_multi_f", form=self._fusion_scan_form)
            feat, ldf = self.YF, 2 * Hf
        # head
        D, Cn = sp.head_width, sp.num_classes
        p_head = float(sp.head["dropout"]) if train else 0.0
        hm = 0
        self._head_seed = 0
        if train and rand is not None:
            p_head = 0.0
            if rand.get("head/mask") is not None:
                if self.head_mask is None:
                    self.head_mask = self.mem.empty((B, T, D))
                self.head_mask.upload(np.asarray(rand["head/mask"], dtype=np.float32))
                hm = self.head_mask.ptr
        elif train and p_head > 0:
            self._head_seed = self._seed(999)
        self._head_args = (hm, p_head, self._head_seed)
        if dense:
            dev.call("mgr_dense_softmax_fwd", feat, ldf, hm, p_head, C.c_uint64(self._head_seed),
                     self._wview("dense/W"), self._wview("dense/b"), self.P, B, T, D, Cn)
        self._feat = (feat, ldf)
        self.rng_step = saved_step

    _wide_ok = False          # probe: with fused encoder scans no GEMM shares a CU with a scan workgroup - the library's own tile choice
    _gate_words = (None, None)  # fused encoder scans: the host words through which the launches the fusion scan / the BPTT of the step
                                # being enqueued wait for hand over their launch numbers
    _enc_scan_form = _capi.SCAN_FORM_AUTO      # form of the encoder scan launches enqueued NOW (AUTO: the context's tune key 4)
    _fusion_scan_form = _capi.SCAN_FORM_AUTO   # form of the fusion layer's scan launch
    _early_words = None      # seq_words dict of the generator in _early_gen
    _since_fresh = 0         # pipelined calls since the last one that started the next batch's pass itself (0 in such a call)
    _early_for = None        # the inputs the generator started last was announced for
    _early_gen = None        # _next_encoders_free generator of the batch after next, its first part already enqueued
    _beside_scans = False    # the fusion layer's GEMMs of the step being enqueued run beside encoder scans of the next batch

    def _tuned(self, on, keys, keep_set=False):
        """Context manager: the library calls enqueued inside run with the tune keys `keys` ({key: value}, mgr.h); a caller's own
        settings - bench.py --tune ... - come back afterwards (keep_set: and are not overridden where they are non-zero)."""
        eng = self

        class _Ctx:
            def __enter__(self_):
                self_.old = {}
                if on:
                    for k, val in keys.items():
                        v = C.c_int()
                        eng.dev.call("mgr_tune_get", k, C.byref(v))
                        if keep_set and v.value != 0:
                            continue
                        self_.old[k] = v.value
                        eng.dev.call("mgr_tune", k, val)

            def __exit__(self_, *exc):
                for k, val in self_.old.items():
                    eng.dev.call("mgr_tune", k, val)
                return False
        return _Ctx()

    def _narrow_tiles(self, on):
        """Context manager: the pre-split products enqueued inside take their 4-wave forms (tune key 12 = 1, mgr.h)."""
        return self._tuned(on, {12: 1})

    def _new_seq_word(self):
        """Address of a zeroed page-locked word: a launch number on its way from the launch (mgr_scan_launch_opts.seq_out) to the
        wait that was enqueued before it (mgr_stream_wait_resident_word)."""
        i = self._seq_next
        self._seq_next = (i + 1) % self._seq_words.size
        self._seq_words[i] = 0
        return self._seq_words.ctypes.data + 4 * i

    def _scan_multi(self, jobs, wsname, form=_capi.SCAN_FORM_AUTO, seq_word=0):
        """One multi-scan call on the current stream; `wsname` keeps the encoder and fusion workspaces apart (they may
        be in flight at the same time when steps are pipelined).  form: mgr.h MGR_SCAN_FORM_* (AUTO: the context's tune key 4);
        seq_word: address of the host word that receives the launch number (0: not wanted)."""
        arr = _capi.make_scan_jobs(jobs)
        need = self.lib.mgr_lstm_scan_multi_ws_bytes(len(jobs), arr)
        ws = getattr(self, wsname, None)
        if ws is None or ws.nbytes < need:
            ws = self.mem.bytes(need)
            setattr(self, wsname, ws)
        opts = _capi.make_launch_opts(form, seq_word)
        _capi.check(self.lib.mgr_lstm_scan_fwd_multi_ex(self.dev.ctx, len(jobs), arr, ws.ptr, ws.nbytes, C.byref(opts)))

    def resident_wait_stats(self):
        """(residency waits of this engine's context that have ended, those that ran into their bound) - mgr_resident_wait_stats.
        A wait at its bound costs Schedule.resident_wait_us of its stream and means the schedule's hand-over failed."""
        out = (C.c_uint * 4)()
        self.dev.call("mgr_resident_wait_stats", out)
        return int(out[0]), int(out[1])

    # ------------------------------------------------------------------------------------------ public
    def scan_health(self, snapshot=None):
        """(status bits, optimizer updates skipped by the update gate) of THIS engine since the last clear_scan_status(),
        read on the current stream (or taken from `snapshot`, the words a step copied to the host with its loss); never raises."""
        st = (C.c_uint * 4)()
        if snapshot is not None:
            st[0], st[2] = int(snapshot[0]), int(snapshot[2])
        else:
            self._bind()
            self.dev.call("mgr_scan_status_ex", st)
        self.updates_skipped = int(st[2])
        if st[0] & _capi.SCAN_NONFINITE:
            self.nonfinite_seen = True
        return int(st[0]), int(st[2])

    def _check_scans(self, step=None, snapshot=None):
        """Raises if a persistent scan of this engine gave up on a bounded spin (its outputs are garbage; the update gate keeps
        that step's gradients away from the weights, apply_gradients); a non-finite hidden state is not an error of the
        engine - the outputs / loss carry the NaN like the reference's would - and is remembered in `nonfinite_seen` until
        clear_scan_status().  `updates_skipped` counts the optimizer steps the gate has dropped since then."""
        bits, skipped = self.scan_health(snapshot)
        if bits & ~_capi.SCAN_NONFINITE:
            where = "" if step is None else " (noticed with the loss of training step %d)" % step
            raise _capi.MgrError("a persistent scan gave up on a bounded spin (status %d)%s: the outputs of that pass are invalid and "
                                 "its optimizer update is skipped on the device (the weights stay as they were) - "
                                 "clear_scan_status() to continue" % (bits, where))

    nonfinite_seen = False
    updates_skipped = 0

    @property
    def iterations(self):
        """Optimizer updates APPLIED so far (Keras `iterations`: learning-rate decay, Adam bias correction): the steps enqueued
        minus those the update gate skipped on the device, as far as the host knows (updates_skipped is refreshed with every
        loss read-back and by scan_health())."""
        return max(0, self._adam_calls - int(self.updates_skipped))

    @iterations.setter
    def iterations(self, value):
        self._adam_calls = int(value) + int(self.updates_skipped)

    def clear_scan_status(self):
        """Forget recorded scan status bits (after recovering from a diverged run / a give-up)."""
        self.scan_health()             # (the device's skipped-update count restarts at 0: fold it into the host's counter first)
        applied = self.iterations
        self.dev.call("mgr_scan_status_clear")
        self.nonfinite_seen = False
        self.updates_skipped = 0
        self._adam_calls = applied

    def predict(self, inputs):
        """Softmax output (B,T,C) with learning phase 0 (sequence_decoding.py:81)."""
        self._upload_inputs(inputs, None, False)
        blk = self._begin_pass()
        self._forward(False, None)
        P = self.P.download()
        return self._end_pass(blk.download(), P)

    def forward_train_phase(self, inputs, rand=None):
        """Softmax output with learning phase 1 (dropout / noise active) - no gradient."""
        self._upload_inputs(inputs, rand, True)
        blk = self._begin_pass()
        self._forward(True, rand)
        P = self.P.download()
        return self._end_pass(blk.download(), P)

    # ------------------------------------------------------------------------------------------ pipelined inference / validation
    EV_ENC = (46, 47)     # the encoder pass into FEAT buffer 0 / 1 is complete
    EV_FUSED = (48, 49)   # the fusion / head pass that read FEAT buffer 0 / 1 (and whatever decodes its output) is complete
    EV_OUT = (50, 51)     # the result of the batch in output slot 0 / 1 has reached its pinned host buffer
    EV_D1P = (55, 56)     # the depth-1 projections into depth-1 Z set 0 / 1 are done (stream 0)
    EV_D1S = (57, 58)     # the depth-1 scans that read depth-1 Z set 0 / 1 are done (stream ES)

    def predict_stream(self, batches, output="posteriors", train_phase=False, beam_width=10, merge_repeated=True):
        """Batches of an inference / validation run are independent of each other: this generator keeps two of them in flight.
        Batch n + 1 is uploaded (copy stream) and runs its encoder pass (stream ES, into the other FEAT buffer) beside batch n's
        fusion layer / head / decode kernels (stream 0) and beside the download of batch n - 1's result (its own stream, into
        pinned host memory); the host only ever waits for the result it is about to hand out.  What the reference does with
        predict_generator over the whole set (sequence_decoding.py:118-127) and with the validation loop of every epoch
        (multimodal.py:264-269), one blocking batch at a time.

        batches: iterable of input dicts {stream name: (B, T, F)} - for output="loss" of tuples (inputs, labels, input_length,
        label_length).  Yields, in order, per batch:
          "posteriors"  P (B, T, C) float32                  (learning phase 0 unless train_phase)
          "argmax"      (best (B, T - skip) int32, prob (B, T - skip) float32): mgr_frame_argmax on the device, the (B, T, C)
                        posteriors never travel to the host (decoding.confidence_filter_collapse does the rest)
          "beam"        (paths: list of B label lists, log-probabilities (B,) float64): mgr_ctc_beam_search on the device
          "loss"        per-sample CTC losses (B,) float32  (a training engine; learning phase as train_phase)
        Results are bit-identical to the one-batch-at-a-time calls (predict / loss_on_batch): same kernels, same order per batch."""
        sp, dev, B, T = self.spec, self.dev, self.B, self.T
        Cn, skip = sp.num_classes, int(sp.ctc["skip"])
        if output not in ("posteriors", "argmax", "beam", "loss"):
            raise ValueError("unknown output %r" % (output,))
        if output == "loss" and self.inference_only:
            raise ValueError("output='loss' needs a training engine (labels, CTC workspace)")
        self._bind()
        if self._prefetched is not None:     # a pipelined training encoder pass is in flight: let it finish, then discard it
            dev.wait(0, self.ES)
            self._prefetched = None
        dev.sync()                           # (nothing of an earlier call may still use the buffers the pipeline cycles through)
        ring = self._feat_ring
        ES, OUT = (self.ES if len(ring) > 1 else 0), self.LOSS_STREAM
        st = self.__dict__.setdefault("_stream_bufs", {})

        def bufs(key, make):
            if key not in st:
                st[key] = make()
            return st[key]

        # two P buffers: the decode kernels / the download of batch n read one on their own stream while the head of batch n + 1
        # writes the other
        pring = bufs("pring", lambda: [self.P, self.mem.empty((B, T, Cn))])
        if output == "posteriors":
            pins = bufs("P", lambda: [dev.pinned((B, T, Cn), np.float32) for _ in range(2)])
        elif output == "argmax":
            dbest = bufs("dbest", lambda: self.mem.empty((B, T - skip), np.int32))
            dprob = bufs("dprob", lambda: self.mem.empty((B, T - skip), np.float32))
            pins = bufs("am", lambda: [(dev.pinned((B, T - skip), np.int32), dev.pinned((B, T - skip), np.float32)) for _ in range(2)])
        elif output == "beam":
            dil = bufs("dil", lambda: self.mem.empty((B,), np.int32))
            dil.upload(np.full(B, T - skip, np.int32))
            dout = bufs("dout", lambda: self.mem.empty((B, T - skip), np.int32))
            dlen = bufs("dlen", lambda: self.mem.empty((B,), np.int32))
            dlogp = bufs("dlogp", lambda: self.mem.empty((B,), np.float64))
            wsb = bufs("wsb%d" % beam_width, lambda: self.mem.bytes(self.lib.mgr_ctc_beam_ws_bytes(B, T, Cn, int(beam_width))))
            pins = bufs("beam", lambda: [(dev.pinned((B, T - skip), np.int32), dev.pinned((B,), np.int32), dev.pinned((B,), np.float64))
                                         for _ in range(2)])
        else:
            pins = bufs("loss", lambda: [dev.pinned((B,), np.float32) for _ in range(2)])
            # one device buffer per output slot, like pring: batch i's copy (its own stream, behind EV_FUSED only) must not find
            # batch i + 1's losses in the buffer - stream 0 waits for EV_OUT[o] of batch i - 2 before it reuses slot o
            lring = bufs("lring", lambda: [self.loss_b, self.mem.empty((B,))])

        spin = bufs("status16", lambda: [dev.pinned((16,), np.uint32) for _ in range(2)])

        def collect(i):
            o = i & 1
            dev.event_sync(self.EV_OUT[o])
            if output == "posteriors":
                r = pins[o].copy()
            elif output == "argmax":
                r = (pins[o][0].copy(), pins[o][1].copy())
            elif output == "beam":
                po, pl, ps = pins[o]
                r = ([[int(v) for v in po[b, :pl[b]]] for b in range(B)], ps.copy())
            else:
                r = pins[o].copy()
            # the status block of THIS batch's pass travels with its result (no extra synchronisation): the samples whose hidden
            # state went NaN / Inf in it get NaN scores and no labels - not plausible numbers, and not every later batch
            return self._end_pass(spin[o].copy(), r, what="batch %d of the pipelined run" % i)

        # Depth-1 projections of the NEXT batch on stream 0 (round 4).  The cycle of the pipeline is the encoder stream's chain
        # (depth-1 projections 4.4 | depth-1 scans 8.2 | depth-2 projections 11.0 | depth-2 scans 8.3 ms at config F) while stream 0
        # idles two thirds of it; with two sets of depth-1 gate pre-activation buffers the projections of batch i + 1 are enqueued
        # on stream 0 in front of batch i's fusion pass - they run beside batch i's encoder scans - and the encoder stream's chain
        # loses them.  Same kernels, same inputs per batch: results stay bit-identical to the one-batch-at-a-time calls.
        two_stage = len(ring) > 1 and max(len(s_["layers"]) for s_ in sp.streams) >= 2
        z1 = None
        if two_stage:
            z1 = bufs("z1", lambda: [{s_["name"]: (self.mem.empty((B, T, 4 * s_["layers"][0]["H"])), self.mem.empty((B, T, 4 * s_["layers"][0]["H"])))
                                      for s_ in sp.streams} for _ in range(2)])

        started = [0]
        # launch numbers of the depth-1 / depth-2 scan launches of the batch in Z set 0 / 1, as the launches themselves report them
        # (mgr_scan_launch_opts.seq_out: round 5 predicted them from the context's counter, which another engine on the same Device or
        # a launch that does not enter the ledger made wrong - and every wait then ran into its bound)
        scan_seq = np.zeros((2, 2), np.uint32)

        def wait_resident(seq, us):
            if int(seq) not in (0, _capi.SEQ_NONE):     # (no persistent launch: nothing to wait for)
                dev.call("mgr_stream_wait_resident", C.c_uint(int(seq)), int(us))

        def start_encoders(i, item):
            """Upload batch i and enqueue its noise / depth-1 projections on stream 0 (two_stage), or nothing yet; returns the
            generator that enqueues the rest of its encoder pass on ES."""
            inputs = item[0] if output == "loss" else item
            f = i % len(ring)
            first = 0 if two_stage else ES
            self._upload_inputs(inputs, None, train_phase, stream=first)
            self._xin_user[self._xin_slot] = 1 << 60       # (its readers are known by event, not by a loss read-back)
            scan_seq[i & 1, :] = 0
            gen = self._encoder_phases(train_phase, None, ring[f], ES, self.rng_step + (i - started[0]),
                                       first_stream=0 if two_stage else None, z_first=z1[i & 1] if two_stage else None,
                                       seq_words={k: scan_seq.ctypes.data + 4 * (2 * (i & 1) + k) for k in range(2)})
            if two_stage:
                dev.stream(0)
                dev.wait_event(0, self.EV_D1S[i & 1])       # the depth-1 scans that read this Z set two batches ago
                if i >= 1 and self.schedule.resident_wait_us > 0:
                    # ... and not beside the previous batch's depth-2 projection GEMMs on ES (GEMM beside GEMM: the sum of both), but
                    # beside its depth-2 SCANS, whose launch follows its depth-1 scan launch (bounded wait: placement only)
                    wait_resident(scan_seq[(i - 1) & 1, 1], 5 * self.schedule.resident_wait_us)
                tag = next(gen)                             # noise + depth-1 projections -> stream 0
                assert tag == ("projected", 0)
                dev.stream(0)
                dev.record(self.EV_D1P[i & 1])
            return gen

        def finish_encoders(i, gen):
            f = i % len(ring)
            dev.stream(ES)
            dev.wait_event(ES, self.EV_FUSED[f])            # the fusion pass that read this FEAT buffer two batches ago
            dev.wait_event(ES, self.EV_OUT[i & 1])          # ... and the copy of that batch's status block: this batch's pass
            self._begin_pass(i & 1)                         # zeroes it (stream ES) and reports into it from here on
            if two_stage:
                dev.wait_event(ES, self.EV_D1P[i & 1])
            for tag in gen:
                if two_stage and tag == ("scanned", 0):
                    dev.stream(ES)
                    dev.record(self.EV_D1S[i & 1])
            dev.stream(ES)
            dev.record(self.EV_ENC[f])

        n = 0
        try:
            it = iter(batches)
            nxt_item = next(it, None)
            if nxt_item is not None:
                gen = start_encoders(0, nxt_item)
                if two_stage:
                    finish_encoders(0, gen)
            i = -1
            while nxt_item is not None:
                i += 1
                item = nxt_item
                o, f = i & 1, i % len(ring)
                if not two_stage:
                    # ---- batch i's encoder pass (stream ES), beside what stream 0 still does for batch i - 1
                    finish_encoders(i, gen)
                nxt_item = next(it, None)
                if two_stage:
                    # ---- batch i + 1: upload, depth-1 projections on stream 0 (in front of batch i's fusion pass), and the REST of its
                    # encoder pass on ES right away - so that the fusion pass below can be ordered behind the residency of that pass's
                    # first scan launch
                    if nxt_item is not None:
                        finish_encoders(i + 1, start_encoders(i + 1, nxt_item))
                elif nxt_item is not None:
                    gen = start_encoders(i + 1, nxt_item)
                # ---- fusion layer, head, decode / loss kernels of batch i (stream 0)
                dev.stream(0)
                dev.wait_event(0, self.EV_ENC[f])
                if two_stage and nxt_item is not None and self.schedule.resident_wait_us > 0:
                    # Batch i's encoder pass ends and batch i + 1's depth-1 scans start at the same instant on ES: fusion projection
                    # GEMMs released at that instant race the scan's workgroups for the CUs and the scan runs at half speed for its
                    # whole life (13.1 instead of 8.3 ms: profiles/r04_predict_timeline.txt) - they wait until it is resident
                    wait_resident(scan_seq[(i + 1) & 1, 0], self.schedule.resident_wait_us)
                dev.wait_event(0, self.EV_OUT[o])               # batch i - 2's decode / download read the P buffer this pass overwrites
                if output == "loss":
                    self._upload_labels(item[1], item[2], item[3])
                self.P = pring[o]
                self._bind(self._pass_status[o])        # (batch i + 1's encoder pass, enqueued above, bound the other block)
                self._enqueue_fusion_head(train_phase, None, ring[f], self.rng_step + (i - started[0]))
                dev.stream(0)
                if output == "loss":
                    dev.call("mgr_ctc_loss_grad", self.P, self.labels_d, self.ilen_d, self.llen_d, B, T, Cn, self.Lmax, skip, Cn - 1,
                             float(sp.ctc["eps"]), 1.0, lring[o], 0, self.ws_ctc, self.ws_ctc.nbytes)
                    dev.record(self.EV_LAB[self._lab_slot])
                    self._lab_user[self._lab_slot] = 1 << 60
                dev.record(self.EV_FUSED[f])
                # ---- decode kernels and the way back to pinned host memory: their own stream, beside batch i + 1's fusion layer
                dev.stream(OUT)
                dev.wait_event(OUT, self.EV_FUSED[f])
                if output == "posteriors":
                    dev.d2h_async(pins[o], pring[o])
                elif output == "argmax":
                    dev.call("mgr_frame_argmax", pring[o], B, T, Cn, skip, dbest, dprob)
                    dev.d2h_async(pins[o][0], dbest)
                    dev.d2h_async(pins[o][1], dprob)
                elif output == "beam":
                    dev.call("mgr_ctc_beam_search", pring[o], dil, B, T, Cn, skip, Cn - 1, int(beam_width), C.c_float(float(sp.ctc["eps"])),
                             1 if merge_repeated else 0, dout, dlen, dlogp, wsb, wsb.nbytes)
                    dev.d2h_async(pins[o][0], dout)
                    dev.d2h_async(pins[o][1], dlen)
                    dev.d2h_async(pins[o][2], dlogp)
                else:
                    dev.d2h_async(pins[o], lring[o])
                dev.d2h_async(spin[o], self._pass_status[o])
                dev.record(self.EV_OUT[o])
                dev.stream(0)
                n = i + 1
                if i >= 1:
                    yield collect(i - 1)
            if n >= 1:
                yield collect(n - 1)
        finally:
            self.P = pring[0]
            if train_phase:
                self.rng_step += n          # (one draw of randomness per batch, as the one-batch-at-a-time calls)
            dev.stream(0)
            dev.sync()
            self._bind()

    def _upload_labels(self, labels, input_length, label_length):
        lab = np.asarray(labels)
        lab = np.where(np.isfinite(lab), lab, -1).astype(np.int32).reshape(self.B, -1)
        if lab.shape[1] != self.Lmax:
            if lab.shape[1] > self.Lmax:
                raise ValueError("label rows longer (%d) than Lmax=%d" % (lab.shape[1], self.Lmax))
            lab = np.concatenate([lab, -np.ones((self.B, self.Lmax - lab.shape[1]), np.int32)], axis=1)
        dev = self.dev
        slot = self._lab_slot ^ 1
        dev.stream(self.COPY_STREAM)
        if self._lab_user[slot] > self._synced_step:       # see _upload_inputs
            dev.wait_event(self.COPY_STREAM, self.EV_LAB[slot])
        labels_d, ilen_d, llen_d = self._lab_ring[slot]
        plab, pil, pll = self._lab_pin[slot]    # (the staging set is reused together with the device set: same ordering)
        if self._lab_copied[slot]:              # (... and refilled by the host only once its last copies are done: _upload_inputs)
            dev.event_sync(self.EV_LAB_COPIED[slot])
        plab[...] = lab
        pil[...] = np.asarray(input_length).reshape(self.B)
        pll[...] = np.asarray(label_length).reshape(self.B)
        dev.h2d_async(labels_d, plab)
        dev.h2d_async(ilen_d, pil)
        dev.h2d_async(llen_d, pll)
        dev.record(self.EV_LAB_COPIED[slot])
        self._lab_copied[slot] = True
        dev.wait(0, self.COPY_STREAM)          # the CTC kernel runs on stream 0
        self._lab_slot = slot
        self._lab_user[slot] = 1 << 60
        self.labels_d, self.ilen_d, self.llen_d = labels_d, ilen_d, llen_d
        dev.stream(0)

    def loss_on_batch(self, inputs, labels, input_length, label_length, rand=None, train_phase=True):
        """Per-sample CTC loss (validation inside fit_generator: learning phase stays 1, multimodal.py:66)."""
        self._upload_inputs(inputs, rand, train_phase)
        self._upload_labels(labels, input_length, label_length)
        blk = self._begin_pass()
        self._forward(train_phase, rand)
        sp = self.spec
        self.dev.call("mgr_ctc_loss_grad", self.P, self.labels_d, self.ilen_d, self.llen_d, self.B, self.T,
                      sp.num_classes, self.Lmax, int(sp.ctc["skip"]), sp.num_classes - 1, float(sp.ctc["eps"]),
                      1.0, self.loss_b, 0, self.ws_ctc, self.ws_ctc.nbytes)
        lb = self.loss_b.download()
        return self._end_pass(blk.download(), lb)

    def train_step(self, inputs, labels, input_length, label_length, rand=None, apply_update=True, next_inputs=None,
                   after_next_inputs=None):
        """One optimizer step (Keras train_on_batch).  Returns the mean CTC loss of the local batch.
        next_inputs (optional): the inputs of the FOLLOWING call; when the encoders are frozen their pass for that batch
        is overlapped with this step's fusion / BPTT / Adam work (bit-identical results); after_next_inputs: those of the call
        after that (the encoder stream gets the first part of its pass a call early)."""
        self.enqueue_train_step(inputs, labels, input_length, label_length, rand, apply_update,
                                prefetch_next=next_inputs is not None, next_inputs=next_inputs,
                                prefetch_after_next=next_inputs is not None and after_next_inputs is not None,
                                after_next_inputs=after_next_inputs)
        return self.read_loss()

    LOSS_STREAM = 6   # (pipelined inference: decode / result copies)
    PG_STREAM = 4     # (Schedule.param_grads_two_streams: the second direction's parameter-gradient chain)
    EV_LOSS = 52      # the loss and the scan status of the step enqueued last have reached their page-locked host words

    def read_loss(self, local=False):
        """Mean CTC loss of the step enqueued last: waits for the event behind the loss kernels and their read-back - not for
        the gradient GEMMs / optimizer queued behind them - so the host can enqueue the next step early.
        Data parallel (world > 1): the mean over the GLOBAL batch, the same number on every rank - it arrives with the gradient
        all-reduce, i.e. at the END of the step (read_global_loss); local=True returns this rank's own mean without that wait
        (a training loop that wants to keep the host ahead of the device reads the global loss one step late)."""
        dev = self.dev
        dev.event_sync(self.EV_LOSS)
        v = float(self.loss_host[0])
        step = self._step_id - 1
        self._synced_step = step
        self._check_scans(step=step, snapshot=self.status_host)   # raises if a persistent scan gave up: results would be garbage
        if not local and self.world > 1 and self._gloss_step[step & 1] == step:
            return self.read_global_loss(step)
        if self.nonfinite_seen:
            # a hidden state went NaN / Inf: in the reference every later op propagates the NaN into the loss.  Here a NaN
            # feature that input dropout happens to drop is SKIPPED by the dropout-aware kernels (no 0 x NaN), so the number
            # the device computed may look finite - report what the reference would report
            return float("nan")
        return v

    EV_GLOSS = (53, 54)   # the all-reduced (gate flag, loss sum) of step s has reached gloss_host[s & 1]

    def read_global_loss(self, step=None):
        """Mean CTC loss over the global batch of training step `step` (default: the one enqueued last), identical on every
        rank: (sum over ranks of the local means) / world, carried by the gradient all-reduce of that step.  NaN on EVERY rank
        when any rank's scans reported a non-finite state or gave up in that step (the all-reduced update-gate flag)."""
        step = self._step_id - 1 if step is None else int(step)
        slot = step & 1
        if self._gloss_step[slot] != step:
            raise ValueError("no all-reduced loss for step %d (not a data-parallel step, its update was not applied, or two "
                             "later steps have been enqueued since)" % step)
        self.dev.event_sync(self.EV_GLOSS[slot])
        flag, total = float(self.gloss_host[slot][0]), float(self.gloss_host[slot][1])
        return float("nan") if flag != 0.0 else total / self.world

    @property
    def can_pipeline(self):
        """Frozen encoders do not depend on the weights the step updates, so the encoder pass of step n+1 may run
        while step n's fusion layer / CTC / BPTT / Adam are in flight (the fusion config of the reference)."""
        sp = self.spec
        return bool(sp.fusion) and not any(s["trainable"] for s in sp.streams) and not self.inference_only

    ES = 5          # encoder stream when steps are pipelined
    EV_PREV = 40    # end of the previous step's fusion phase on stream 0 (its dW GEMMs read the other FEAT buffer)
    EV_FPROJ = 45   # this step's fusion projection GEMMs are done

    def enqueue_train_step(self, inputs, labels, input_length, label_length, rand=None, apply_update=True,
                           upload=True, prefetch_next=False, next_inputs=None, prefetch_after_next=False, after_next_inputs=None):
        """Enqueue one training step.  With prefetch_next (device-RNG training of a network whose encoders are frozen)
        the NEXT step's encoder pass is enqueued on a second stream into the other FEAT buffer right after this step's
        fusion work, and this step consumes the encoder pass enqueued by the previous call (Schedule).
        prefetch_after_next (round 5; the caller promises that the call after the next one comes, with prefetch_next, for
        after_next_inputs): the encoder stream also gets the first part of THAT batch's pass - up to its deepest projections - at
        the end of this call instead of at the start of the next one.  The next call can only be made once the host has this step's
        loss (the CTC kernel, half-way through the step): the encoder stream used to sit idle from the end of its deepest scan
        until then (0.9 - 3.9 ms per step in the traces of round 5)."""
        sp, dev, B, T = self.spec, self.dev, self.B, self.T
        sch, ES = self.schedule, self.ES
        self._bind()
        pipelined = prefetch_next and rand is None and self.can_pipeline and sch.pipeline
        depth = max(len(s_["layers"]) for s_ in sp.streams)
        # ---- 1. this step's encoder pass: the one the previous call prefetched, or in line on stream 0
        have = self._prefetched
        if have is not None:
            stale = rand is not None or not self.can_pipeline
            if upload and self._prefetched_for is not None and inputs is not self._prefetched_for:
                stale = True   # the caller did not come back with the batch it announced
            if stale:
                dev.wait(0, ES)
                have = None
        self._prefetched = None
        defer = pipelined and sch.defer_param_grads and sp.fusion is not None and depth >= 2
        any_tr_stream_ = any(s_["trainable"] for s_ in sp.streams)
        host_blocking = bool(getattr(self.comm, "host_blocking", False))
        # What the encoder stream was handed a call early (the first part of the pass of the batch after this one) is settled BEFORE
        # this step's FEAT buffer is chosen: when the caller did not come back as announced - another batch, a predict /
        # loss_on_batch in between that discarded the prefetched pass - the early generator is dropped and its claim on the other FEAT
        # buffer with it; choosing `cur` first put the NEXT batch's deepest scan on top of the buffer this step's deferred dW GEMMs read.
        # (A step without a prefetched pass always drops it: its own pass uses the encoder buffers the early part has written.)
        early, self._early_gen = self._early_gen, None
        early_words, self._early_words = self._early_words, None
        keep_early = (have is not None and pipelined and defer and sch.encoders_run_ahead and sch.bptt_beside_deepest_scan and sp.fusion
                      and not any_tr_stream_ and (not upload or next_inputs is self._early_for))
        if early is not None and not keep_early:
            dev.wait(0, ES)
            self._feat_idx ^= 1        # (the generator toggled it when it was started; nothing of its pass has touched a FEAT buffer yet)
            early.close()
            early = None
        # The fused forms of the encoder scans belong to every pass the encoder stream runs for a pipelined step (round 6: also the
        # passes of the first steps of a run, which round 5 ran in the plain form until a step had been announced two calls ahead)
        fused_ok = bool(pipelined and defer and sch.fused_encoder_scans and sch.encoders_run_ahead and sch.bptt_beside_deepest_scan
                        and sp.fusion and not any_tr_stream_ and not host_blocking)
        own_on_es = False
        if have is None:
            if fused_ok and sch.first_pass_on_encoder_stream:
                # Round 6: a pipelined step WITHOUT a prefetched pass (the first step of a run, the step after a validation pass) runs
                # its own encoder pass on the ENCODER stream, in the fused forms - nothing of this step can run beside it anyway, and
                # from here on the call is a steady-state call: the next batch's pass follows on that stream at once, beside this
                # step's trainable part, instead of behind a whole in-line pass in the plain forms and a stream-wide wait.  The step
                # itself is what it was: same kernels' results, same order of randomness.
                dev.wait(ES, 0)        # (the encoder buffers / this FEAT buffer may still be read by what stream 0 has queued)
                if upload:
                    self._upload_inputs(inputs, None, True, stream=ES)
                    self._xin_user[self._xin_slot] = self._step_id     # (this step reads them; frozen encoders: in this pass only)
                cur = self._feat_ring[self._feat_idx]
                self._enc_scan_form = _capi.SCAN_FORM_FUSED
                try:
                    self._enqueue_encoders(True, None, cur, ES, self.rng_step)
                finally:
                    self._enc_scan_form = _capi.SCAN_FORM_AUTO
                dev.wait(0, ES)
                own_on_es = True
            else:
                if upload:
                    self._upload_inputs(inputs, rand, True)
                cur = self._feat_ring[self._feat_idx]
                self._enqueue_encoders(True, rand, cur, 0, self.rng_step)
        else:
            cur = have
            # this step's encoder pass (enqueued by the previous call) must be complete: the event behind its deepest scan - not
            # everything the encoder stream has been handed since (round 5 waited for the whole stream: in steady state the first part of
            # the NEXT batch's pass, queued behind it, is done by then anyway; in the first steps of a run it is not, and the second step
            # started 7 ms late)
            dev.wait_event(0, self.EV_ENC[self._feat_ring.index(cur)])
        # (not in a call that ran its own encoder pass in line on stream 0: that pass uses the encoder buffers)
        ahead = defer and sch.encoders_run_ahead and (have is not None or own_on_es)
        free_ok = pipelined and ahead and sch.bptt_beside_deepest_scan and sp.fusion and not any_tr_stream_
        if upload:
            self._upload_labels(labels, input_length, label_length)
        if pipelined:
            # the other FEAT buffer was last read by the previous step's fusion phase (its dW GEMMs, queued on stream 0)
            if ahead:
                dev.stream(0)
                dev.record(self.EV_PREV)   # only the launch that WRITES that buffer (the deepest scan) has to wait for it
            else:
                dev.wait(ES, 0)
        # Free-running encoder stream (Schedule.bptt_beside_deepest_scan): the next batch's encoder pass up to its deepest projections
        # is handed to the encoder stream BEFORE this step's fusion work is enqueued - it depends on nothing of this step, and the
        # host needs ~1-2 ms to enqueue the fusion layer and the head, during which that stream would sit idle
        free_gen = None
        free_words = None
        fused = bool(fused_ok and free_ok)
        fresh = False
        self._gate_words = (None, None)
        new_words = None
        try:
            if fused:
                self._enc_scan_form = _capi.SCAN_FORM_FUSED     # every encoder scan launch enqueued from here to the end of this call
            if early is not None:
                free_gen, free_words = early, early_words   # (its first part was enqueued at the end of the previous call)
            elif free_ok:
                fresh = True
                free_words = {}
                if fused:
                    free_words[0] = self._new_seq_word()
                free_gen = self._next_encoders_free(next_inputs, depth, self.rng_step + 1, self._step_id + 1, seq_words=free_words)
                next(free_gen)
            # ---- 2., 3. fusion layer, head, CTC, loss read-back point, backward
            # (fused: not with a host-blocking all-reduce - HostComm holds the host inside finish(), the depth-1 scan the BPTT's wait is
            #  for would only be enqueued after it, the wait would always run into its bound; such runs share a GPU and use small per-rank
            #  batches anyway)
            two = bool(pipelined and free_gen is not None and prefetch_after_next and sch.encoders_two_ahead)
            self._since_fresh = 0 if (fresh or free_gen is None) else self._since_fresh + 1
            if fused:
                # Persistent launches from here on, in host order: fusion scan, BPTT, the next batch's deepest scan and - when the batch
                # after next is announced - its depth-1 scan.  The encoder scans take the FUSED form (an argument of their launch), and
                # each hands its launch number to the word the residency wait of the recurrence that runs BESIDE it polls (the waits are
                # enqueued first).  Steady state: the fusion scan beside the next batch's deepest scan, the BPTT beside the depth-1 scan
                # of the batch after next.  A call that starts the next batch's pass itself (fresh: the first steps of a run): the
                # fusion scan beside that pass's depth-1 scan (already enqueued: its word is filled), the BPTT beside its deepest scan.
                wx = self._new_seq_word()
                free_words[depth - 1] = wx
                wy = None
                if two and depth > 1:
                    wy = self._new_seq_word()
                    new_words = {0: wy}
                # (the call right after a fresh one: the encoder stream is still a phase behind - the deepest scan this call enqueues
                #  starts milliseconds after the fusion scan is ready, and the wait for it ran into its bound, once per run: no wait)
                late = (not fresh) and self._since_fresh == 1
                self._gate_words = (free_words.get(0), wx) if fresh else ((None, wy) if late else (wx, wy))
                self._gate_log.append((self._step_id, "fresh" if fresh else "steady", self._gate_words))
                self._wide_ok = bool(sch.fused_wide_tiles)
                if sch.fusion_scan_fused:
                    self._fusion_scan_form = _capi.SCAN_FORM_FUSED_ANY
            finish = self._enqueue_trainable_part(cur, rand, pipelined, defer, sch.bptt_beside_deepest_scan, have is None and not own_on_es,
                                                  apply_update)

            # ---- 4. parameter gradients + optimizer, and (pipelined) the next step's encoder pass
            if not pipelined:
                finish()
            elif free_gen is not None:
                if new_words is None:
                    new_words = {}
                if two and sch.depth1_proj_ahead and self.Z1buf and depth > 1:
                    # the batch AFTER the next one: its depth-1 projections in front of the next batch's deepest scan, the rest of its
                    # first part at the end of this call (rng_step / _step_id were advanced above)
                    def pre():
                        self._early_gen = self._next_encoders_free(after_next_inputs, depth, self.rng_step + 1, self._step_id + 1,
                                                                   split_first=True, seq_words=new_words)
                        self._early_words = new_words
                        next(self._early_gen)
                    free_gen.send((finish, pre))
                    next(self._early_gen)
                else:
                    free_gen.send(finish)
                    if two:
                        # the first part of the pass of the batch AFTER the next one (rng_step / _step_id were advanced above)
                        self._early_gen = self._next_encoders_free(after_next_inputs, depth, self.rng_step + 1, self._step_id + 1,
                                                                   seq_words=new_words)
                        self._early_words = new_words
                        next(self._early_gen)
            else:
                self._enqueue_next_encoders(next_inputs, finish, defer, ahead, depth,
                                            free_running=bool(defer and sch.bptt_beside_deepest_scan and sp.fusion and not any_tr_stream_))
        finally:
            self._beside_scans = False
            self._enc_scan_form = _capi.SCAN_FORM_AUTO
            self._fusion_scan_form = _capi.SCAN_FORM_AUTO
            self._wide_ok = False
            self._gate_words = (None, None)
        dev.stream(0)

    def _enqueue_trainable_part(self, cur, rand, beside_scans, defer, late_ok, own_inputs, apply_update):
        """Steps 2 and 3 of a training step on stream 0 - fusion layer, head, CTC, the loss read-back point, the backward pass up to
        what `defer` / `late_ok` hold back - and the closure `finish(gate=None)` that enqueues the rest (held-back BPTT, dW / dU / db
        GEMMs, optimizer)."""
        sp, dev, B, T = self.spec, self.dev, self.B, self.T
        # ---- 2. fusion layer, head, CTC, loss read-back point
        self._beside_scans = bool(beside_scans)
        self._enqueue_fusion_head(True, rand, cur, self.rng_step, dense=False)
        self.rng_step += 1
        Cn, D = sp.num_classes, sp.head_width
        dev.stream(0)
        feat, ldf = self._feat
        hm, p_head, hseed = self._head_args
        W = sp.concat_width
        any_tr_stream = any(s["trainable"] for s in sp.streams)
        if sp.fusion:
            dA, ldda = self.dYF, 2 * sp.fusion["H"]
        elif any_tr_stream:
            dA, ldda = self.dFEAT, W
        else:
            dA, ldda = 0, D
        # the whole head in one call (mgr.h): Dropout / Dense / softmax, CTC loss + gradient, the mean loss, Dense backward
        # (a step of the fused schedule: the head runs beside the next batch's deepest encoder scan - 208 whole CUs.  The CTC kernels ask
        #  for more LDS than a scan workgroup leaves on its CU - tune keys 20 / 21, mgr.h - so that the 32 recurrence workgroups get one of
        #  the 48 other CUs each, and neither they nor the per-frame kernels share SIMDs with the scan, which is on the step's critical path)
        with self._tuned(self._gate_words[0] is not None or self._gate_words[1] is not None, {20: 96, 21: 64}, keep_set=True):
            dev.call("mgr_head_fwd_bwd", feat, ldf, hm, p_head, C.c_uint64(hseed), self._wview("dense/W"), self._wview("dense/b"),
                     self.labels_d, self.ilen_d, self.llen_d, B, T, D, Cn, self.Lmax, int(sp.ctc["skip"]), Cn - 1, float(sp.ctc["eps"]),
                     1.0 / B, self.P, self.loss_b, self.loss_mean, self.dLogits, self._gview("dense/W"), self._gview("dense/b"),
                     dA, ldda, self.ws_head, self.ws_head.nbytes)
        if self.comm is not None:
            dev.call("mgr_mean", self.loss_b, B, self.loss_slot)   # travels with the gradient all-reduce (apply_gradients)
        dev.record(self.EV_LAB[self._lab_slot])
        # The loss and this engine's scan status go to page-locked host words from THIS stream, an event behind them: the host
        # polls that event (read_loss) while the rest of the backward pass queued behind it runs.  Round 2 read them on a stream
        # of their own that waited for stream 0 - a copy on an otherwise idle hardware queue now and then started 20-45 ms after
        # the event it waited for (one step in ten of a 5 ms step; profiles/r03_loss_readback_stall.txt).
        dev.d2h_async(self.loss_host, self.loss_mean)
        dev.d2h_async(self.status_host, self.status)
        dev.record(self.EV_LOSS)
        this_step = self._step_id
        self._step_id += 1
        self._lab_user[self._lab_slot] = this_step
        if own_inputs:   # this step's own inputs (a prefetched set was tagged when it was uploaded)
            # trainable first layers read them again in their dW GEMMs, which only the NEXT step's loss read-back covers
            self._xin_user[self._xin_slot] = this_step + (1 if any_tr_stream else 0)
        # ---- 3. backward
        deferred = None
        late_bptt = None    # (Schedule.bptt_beside_deepest_scan) the fusion layer's BPTT itself is held back with its GEMMs
        if sp.fusion:
            Hf = sp.fusion["H"]
            bargs = ("fusion", self.dYF, 2 * Hf, self._featin, W, W, self.YF, 2 * Hf, self.dFEAT if any_tr_stream else None, W)
            bkw = dict(defer_param_grads=defer, XinT=self._featT.get(self._featin.ptr))
            if defer and late_ok and not any_tr_stream:
                late_bptt = lambda: self._bilstm_backward(*bargs, **bkw)
            else:
                deferred = self._bilstm_backward(*bargs, **bkw)
        if any_tr_stream:
            col = 0
            for s in sp.streams:
                if s["trainable"]:
                    self._stream_backward(s, col)
                col += sp.stream_width(s)
            dev.stream(0)
            dev.record(self.EV_IN[self._xin_slot])   # trainable first layers read the inputs again in their dW GEMMs

        gate_b = self._gate_words[1]

        def finish(gate=None):
            """This step's (held-back) BPTT, its dW / dU / db GEMMs and the optimizer on stream 0.  gate: enqueues the device-side
            wait for the residency of the next persistent launch of the context - the deepest encoder scan the caller enqueues
            right after this returns.  It goes BEHIND the BPTT (a persistent launch itself: in front of it, "the next persistent
            launch" would be the BPTT queued behind the gate) and in FRONT of the chip-filling GEMMs, which are what
            must not be placed before the scan's workgroups: recurrence beside recurrence starts at once, GEMMs wait."""
            dev.stream(0)
            if gate_b is not None and late_bptt is not None:
                # (as for the fusion scan: the depth-1 scan of the batch after next, enqueued at the end of this call, fills the word)
                dev.call("mgr_stream_wait_resident_word", int(gate_b), int(self.schedule.resident_wait_us))
            d = late_bptt() if late_bptt is not None else deferred
            if gate is not None:
                gate()
            if d is not None:
                d()
            if apply_update:
                self.apply_gradients()
        return finish

    def _next_encoders_free(self, next_inputs, depth, rng_step, consumer_step, split_first=False, seq_words=None):
        """Schedule.bptt_beside_deepest_scan, as a generator.  Part 1 (before this step's fusion work is enqueued): the next
        batch's encoder pass on stream ES up to and including its deepest projection GEMMs.  Part 2 (after the loss read-back point,
        resumed with send(finish) or send((finish, pre))): stream 0 waits for those GEMMs, then - once the deepest scan launched behind
        them is resident - runs `finish` (this step's BPTT, dW / dU / db GEMMs, optimizer) beside that scan; `pre`, if given, is called
        with stream ES selected right in front of the deepest scan's waits.
        split_first (Schedule.depth1_proj_ahead): part 1 comes in two pieces - upload, noise and the depth-1 projections into the
        engine's second set of gate pre-activation buffers, then one more yield, then the rest - so that a caller can put the first
        piece in front of the PREVIOUS batch's deepest scan (its `pre`)."""
        dev, ES = self.dev, self.ES
        self._feat_idx ^= 1
        nxt = self._feat_ring[self._feat_idx]
        self._early_for = next_inputs          # (becomes _prefetched_for when the pass is complete: two announcements may be in flight)
        if next_inputs is not None:
            dev.stream(ES)
            self._upload_inputs(next_inputs, None, True, stream=ES)
            self._xin_user[self._xin_slot] = consumer_step
        split_first = bool(split_first and self.Z1buf)
        phases = self._encoder_phases(True, None, nxt, ES, rng_step, z_first=self.Z1buf if split_first else None, seq_words=seq_words)
        for tag, k in phases:
            if tag == "projected" and k == 0 and split_first and depth > 1:
                dev.stream(0)
                yield "depth-1 projections"
                continue
            if tag == "projected" and k == depth - 1:
                break
        dev.stream(0)
        got = yield
        finish, pre = got if isinstance(got, tuple) else (got, None)
        dev.wait(0, ES)
        dev.stream(0)
        finish(self._resident_gate())
        if pre is not None:
            dev.stream(ES)
            pre()
            dev.stream(ES)
        dev.wait_event(ES, self.EV_PREV)   # (the deepest scan overwrites the FEAT buffer the previous step's dW GEMMs read)
        if self.schedule.deepest_scan_after_fusion_proj:
            # ... and it lets THIS step's fusion projections go first: both become ready at the same instant (the end of the previous
            # step's optimizer), and beside the scan's 408 workgroups the two GEMMs took 4.5 ms instead of 1.1 - on the chain of
            # the stream that sets the step time, while the encoder stream has ~3 ms to spare (profiles/r05_scan_probes.txt)
            dev.wait_event(ES, self.EV_FPROJ)
        for _ in phases:
            pass
        dev.stream(ES)
        dev.record(self.EV_ENC[self._feat_ring.index(nxt)])     # this pass is complete (what the step that consumes it waits for)
        self._prefetched = nxt
        self._prefetched_for = next_inputs
        dev.stream(0)
        yield

    def _resident_gate(self):
        """The device-side wait (current stream) until the persistent launch enqueued NEXT on this context - the deepest encoder scan -
        reports every workgroup resident, bounded by Schedule.resident_wait_us; None where it must not be used (switched off; in
        front of a host-blocking all-reduce: HostComm holds the host inside finish(), the scan the gate waits for would only be
        enqueued after it - the gate would always run into its bound)."""
        if self.schedule.resident_wait_us <= 0 or getattr(self.comm, "host_blocking", False):
            return None
        return lambda: self.dev.call("mgr_stream_wait_next_resident", self.schedule.resident_wait_us)

    def _enqueue_next_encoders(self, next_inputs, finish, defer, ahead, depth, free_running=False):
        """Encoder pass of the NEXT step on stream ES into the other FEAT buffer, concurrent with what enqueue_train_step
        has put on stream 0; `finish` (this step's parameter-gradient GEMMs + optimizer) is placed according to the schedule."""
        dev, ES = self.dev, self.ES
        self._feat_idx ^= 1
        nxt = self._feat_ring[self._feat_idx]
        self._prefetched_for = next_inputs
        if next_inputs is not None:
            dev.stream(ES)
            self._upload_inputs(next_inputs, None, True, stream=ES)
            self._xin_user[self._xin_slot] = self._step_id   # (already advanced: the step that will consume it)
        if not defer:
            finish()
            self._enqueue_encoders(True, None, nxt, ES, self.rng_step, hold_scans_for=0)
        else:
            # The dW/dU GEMMs of THIS step and the optimizer are held back until the next step's deepest projection GEMMs are
            # done and then run beside its deepest encoder scan: GEMM next to GEMM gains nothing, whereas a big scan leaves
            # most of the MFMA issue slots free (tools/overlap_probe.py: scan x1.12, GEMM at 1/3 speed).
            # With `ahead` the encoder stream does not wait for this stream's previous step as a whole: its depth-1
            # projections start as soon as its own previous pass is done (beside the small kernels that end that step), the
            # depth-1 scan is launched after this step's fusion projections (a persistent launch placed among chip-filling
            # GEMMs gets a poor CU set), and only the deepest scan - the one that overwrites the FEAT buffer the previous
            # step's dW GEMMs read - waits for that step.
            for tag, k in self._encoder_phases(True, None, nxt, ES, self.rng_step):
                if tag != "projected":
                    continue
                if ahead and k == 0 and not free_running:
                    dev.stream(ES)
                    dev.wait_event(ES, self.EV_FPROJ)
                if k == depth - 1:
                    dev.wait(0, ES)
                    dev.stream(0)
                    # Both streams become ready at the same instant.  The scan's workgroups must be placed first (one or
                    # two per CU): if GEMM waves get there first the scan's workgroups trickle in behind them and the
                    # whole scan runs at half speed (measured 22.7 vs 11.4 ms).  So the GEMMs wait - on the device - until
                    # the scan launched next on this context reports every workgroup resident.
                    # (not in front of a host-blocking all-reduce: HostComm holds the host inside finish(), the scan the gate waits
                    # for would only be enqueued after it - the gate would always run into its bound)
                    finish(self._resident_gate())
                    if ahead:
                        dev.wait_event(ES, self.EV_PREV)
        dev.stream(ES)
        dev.record(self.EV_ENC[self._feat_ring.index(nxt)])
        self._prefetched = nxt

    def _bilstm_backward(self, prefix, dY, lddy, Xin, ldx, fin, Hbuf, ldh, dX, lddx, defer_param_grads=False, XinT=None):
        """BPTT + parameter grads of one Bidirectional layer (both directions in one persistent launch).
        defer_param_grads: return the dW/dU/db GEMM launches as a closure instead of enqueuing them now.
        XinT: the transposed copy of Xin the forward projection was fed (same step), if the engine keeps one - the
        dropout-aware dW then reads both of its operands along time."""
        dev, B, T = self.dev, self.B, self.T
        jobs = []
        for di, dname in enumerate(("fwd", "bwd")):
            L = self.dirs["%s/%s" % (prefix, dname)]
            H = L.H
            dYv = dY.view(di * H, (1,)) if isinstance(dY, DeviceArray) else dY
            jobs.append(dict(dY=dYv, gates=L.gates, cs=L.cs, Up=L.Up, dZ=L.dZ, lddy=lddy, B=B, T=T, H=H,
                             reverse=L.reverse, dzmax=L.dzmax, dbsum=L.dbsum))
        dev.stream(0)
        arr = _capi.make_scan_bwd_jobs(jobs)   # both directions in ONE call (one persistent launch of CU clusters)
        need = self.lib.mgr_lstm_scan_bwd_multi_ws_bytes(len(jobs), arr)
        if getattr(self, "_ws_bwd_multi", None) is None or self._ws_bwd_multi.nbytes < need:
            self._ws_bwd_multi = self.mem.bytes(need)
        beside_scans = self._beside_scans     # (the deferred GEMMs run beside the next batch's encoder scans as well)
        wide_ok = self._wide_ok
        # The form of the narrow-layer BPTT is an argument of the launch (mgr.h MGR_BPTT_FORM_*; AUTO = the context's tune key 16):
        # beside the next batch's encoder scans the form that yields to them - or, beside FUSED encoder scans (CUs of its own) and if the
        # schedule asks for it, the direct gather with one barrier per step; a launch that has the chip to itself takes the context's.
        form = _capi.BPTT_FORM_AUTO
        if beside_scans and self.schedule.bptt_yields_beside_scans:
            direct = (self._wide_ok or self._gate_words[1] is not None) and self.schedule.bptt_direct_when_alone
            form = _capi.BPTT_FORM_DIRECT if direct else _capi.BPTT_FORM_YIELDING
            if self._gate_words[0] is not None and self.schedule.bptt_fused:     # (a step of the fused schedule)
                form = _capi.BPTT_FORM_FUSED_DIRECT if direct else _capi.BPTT_FORM_FUSED
        if self.schedule.bptt_single_cu and jobs[0]["H"] in (32, 64, 100):
            form = _capi.BPTT_FORM_SINGLE_CU       # (whatever the schedule puts beside it: the choice must not depend on the layout)
        opts = _capi.make_launch_opts(form, 0)
        _capi.check(self.lib.mgr_lstm_scan_bwd_multi_ex(dev.ctx, len(jobs), arr, self._ws_bwd_multi.ptr, self._ws_bwd_multi.nbytes,
                                                        C.byref(opts)))

        two = self.schedule.param_grads_two_streams

        def param_grads():
            if two:
                dev.stream(0)
                dev.wait(self.PG_STREAM, 0)     # (everything queued on stream 0 so far: the BPTT, the previous step's optimizer)
            for di, dname in enumerate(("fwd", "bwd")):
                L = self.dirs["%s/%s" % (prefix, dname)]
                H = L.H
                mptr = self._masks.get((L.prefix, L.d), 0)
                if two:
                    dev.stream(self.PG_STREAM if di == 1 else 0)
                if mptr and XinT is not None and self._xt_split.get(XinT.ptr, False):
                    # (the projection of this step left the kept lists of this very mask in its workspace: not built again)
                    pws = L.ws_sp if (L.ws_sp is not None and L.lists_mask == mptr) else 0
                    L.lists_mask = 0
                    # dU like dW: from the split rows of h_prev along time against the same dZ^T rows (Schedule.du_split)
                    hst = 0
                    if self.schedule.du_split and H >= 16:
                        self._prep_hst(L, Hbuf.view(di * H, (1,)), ldh)
                        hst, L.hst_ready = L.HsT, False
                    with self._narrow_tiles(beside_scans and not wide_ok):
                        dev.call("mgr_lstm_param_grads_dropout_ts", XinT, self.ldt, mptr, float(L.p), Hbuf.view(di * H, (1,)), ldh, L.dZ,
                                 L.gWp, L.gUp, L.gbp, B, T, fin, H, L.reverse, L.ws_pg, L.ws_pg.nbytes, L.dzmax, L.dbsum, pws, hst)
                elif mptr and XinT is not None and self.lib.mgr_lstm_param_grads_dropout_wants_transposed(
                        dev.ctx, C.c_float(float(L.p)), int(fin)):
                    need = self.lib.mgr_lstm_param_grads_dropout_t_ws_bytes(B, T, fin, H, self.ldt)
                    if L.ws_pg.nbytes < need:          # (+ the transposed dZ; first use only)
                        L.ws_pg = self.mem.bytes(need)
                    dev.call("mgr_lstm_param_grads_dropout_t", XinT, self.ldt, mptr, float(L.p), Hbuf.view(di * H, (1,)), ldh, L.dZ,
                             L.gWp, L.gUp, L.gbp, B, T, fin, H, L.reverse, L.ws_pg, L.ws_pg.nbytes, self.XT_BOUND)
                elif mptr:   # input dropout was applied: dW only has rows for the kept features of each (gate, sample)
                    dev.call("mgr_lstm_param_grads_dropout", Xin, ldx, mptr, float(L.p), Hbuf.view(di * H, (1,)), ldh, L.dZ,
                             L.gWp, L.gUp, L.gbp, B, T, fin, H, L.reverse, L.ws_pg, L.ws_pg.nbytes)
                else:
                    dev.call("mgr_lstm_param_grads", Xin, ldx, mptr, Hbuf.view(di * H, (1,)), ldh, L.dZ, L.gWp, L.gUp,
                             L.gbp, B, T, fin, H, L.reverse, L.ws_pg, L.ws_pg.nbytes)
            if two:
                dev.stream(0)
                dev.wait(0, self.PG_STREAM)     # (the optimizer / the input-gradient GEMMs behind this need both chains)
        if not defer_param_grads:
            param_grads()
        if dX is not None:
            for di, dname in enumerate(("fwd", "bwd")):
                L = self.dirs["%s/%s" % (prefix, dname)]
                mptr = self._masks.get((L.prefix, L.d), 0)
                dev.call("mgr_lstm_input_grad", L.dZ, L.Wp, mptr, dX, lddx, 1 if di == 1 else 0, B, T, fin, L.H)
        return param_grads if defer_param_grads else None

    def _prep_hst(self, L, Hdir, ldh):
        """The split transposed copy of h_prev of direction L (rows h_{t-1} forward, h_{t+1} reverse) from its outputs Hdir, on the
        current stream, unless this step's copy exists already."""
        if L.hst_ready:
            return
        if L.HsT is None:
            L.HsT = self.dev.zeros((self.B, L.H, self.ldt))
        self.dev.call("mgr_transpose_bt_split_shift", Hdir, ldh, L.HsT, self.ldt, self.B, self.T, L.H, 1 if L.reverse else -1)
        L.hst_ready = True

    def _stream_backward(self, s, col):
        sp, dev, B, T = self.spec, self.dev, self.B, self.T
        W = sp.concat_width
        name = s["name"]
        nl = len(s["layers"])
        dout = self.dFEAT.view(col, (1,))
        if nl == 2:
            H1, H2 = s["layers"][0]["H"], s["layers"][1]["H"]
            if s["residual"]:
                Hbuf, ldh = self.Y2[name], 2 * H2
            else:
                Hbuf, ldh = self.FEAT.view(col, (1,)), W
            self._bilstm_backward("%s/l1" % name, dout, W, self.Y1[name], 2 * H1, 2 * H1, Hbuf, ldh,
                                  self.dY1[name], 2 * H1, XinT=self.Y1T.get(name))
            if s["residual"]:
                dev.call("mgr_add2d", self.dY1[name], 2 * H1, dout, W, self.dY1[name], 2 * H1, B * T, 2 * H1)
            self._bilstm_backward("%s/l0" % name, self.dY1[name], 2 * H1, self._xcur[name], s["F"], s["F"],
                                  self.Y1[name], 2 * H1, None, 0)
        else:
            H1 = s["layers"][0]["H"]
            self._bilstm_backward("%s/l0" % name, dout, W, self._xcur[name], s["F"], s["F"],
                                  self.FEAT.view(col, (1,)), W, None, 0)

    def apply_gradients(self):
        """all-reduce (if data parallel) -> clip -> Adam -> max-norm; identical on every replica."""
        dev, o = self.dev, self.spec.optimizer
        dev.stream(0)
        self._bind()
        # Update gate: a scan of this step that gave up (garbage gradients) or met a non-finite state must not reach the weights,
        # and the host only learns of it with the loss - after these kernels are queued.  So the decision is taken on the device:
        # the flag is evaluated here (stream order: behind every scan and GEMM of the step), rides behind the gradients through
        # the all-reduce (all replicas skip together) and closes Adam + max-norm; read_loss then raises with the weights intact.
        dev.call("mgr_update_gate_eval", _capi.SCAN_GAVE_UP | _capi.SCAN_NONFINITE, self.gate_flag)
        gscale = 1.0
        if self.comm is not None:   # (a 1-rank communicator is legal: the reduction is then the identity)
            self.comm.allreduce_sum(self.grads, max(self.n_train, 4) + 4)
            gscale = 1.0 / self.world
            step = self._step_id - 1
            dev.d2h_async(self.gloss_host[step & 1], self.gate_flag)      # [flag sum, loss sum]: read_global_loss
            dev.record(self.EV_GLOSS[step & 1])
            self._gloss_step[step & 1] = step
        # Keras' `iterations` counts APPLIED updates: what the gate skipped on the device (counted there; the host learns the
        # number with each loss read-back, i.e. one or two steps late) does not advance the learning-rate decay / bias correction
        k = self.iterations
        lr_k = o["lr"] * (1.0 / (1.0 + o["decay"] * k))
        t = k + 1
        lr_t = lr_k * math.sqrt(1.0 - o["beta_2"] ** t) / (1.0 - o["beta_1"] ** t)
        dev.call("mgr_update_gate_set", self.gate_flag)
        try:
            dev.call("mgr_adam_step", self.params, self.grads, self.m, self.v, self.n_train, lr_t, o["beta_1"],
                     o["beta_2"], o["epsilon"], o["clipvalue"] or 0.0, gscale)
            for name, (off, n, shape, kind) in self.seg.items():
                if kind == "kernel":
                    mv = self.spec.kernel_maxnorm(name.rsplit("/", 2)[0])   # "<prefix>/<fwd|bwd>/W" -> "<prefix>"
                    if mv > 0:
                        dev.call("mgr_maxnorm_cols", self.params.view(off, (n,)), shape[0], shape[1], mv, 1e-7)
        finally:
            dev.call("mgr_update_gate_set", 0)      # (never leave the gate pointer installed in the context)
        self._adam_calls += 1

    def close(self):
        """Free this engine's device buffers; destroy the context only if the engine created it."""
        if self.dev.ctx is None:
            return
        self.dev.sync()
        if self._own_dev:
            self.dev.close()
            return
        self.dev.call("mgr_scan_status_bind", 0)      # (the context goes back to its own status block)
        self.mem.free_all()
