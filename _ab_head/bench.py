#!/usr/bin/env python3
"""bench.py - headline metric of BASELINE.json on MI355X:
train frames/sec (+ CTC-loss delta vs the fp64 oracle), fusion BiLSTM+CTC, B=64 per GPU, T=1900.

    python bench.py --gpus 1 --steps 5 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One JSON line on rank 0.  Inputs are synthetic and resident in HBM before the timed region; a step is one
full training step (frozen encoders fwd, fusion BiLSTM fwd/bwd, CTC, Adam) with device-side noise/dropout
RNG and one loss read-back, exactly K of them between barrier+sync pairs; value = all ranks' frames / max time.
The multi-process rendezvous (RCCL unique id) is a plain TCP exchange on MASTER_ADDR:MASTER_PORT+101; torch is not imported.

    python bench.py --gpus 2 --comm host      # world > 1 WITHOUT RCCL: gradients are summed on the host (parallel.HostComm);
                                              # ranks may then share one GPU (per-rank batch is divided so that every rank's
                                              # persistent scans stay co-resident) - a functional check of the data-parallel
                                              # step on a 1-GPU box, not a scaling measurement
"""
import argparse
import json
import os
import sys
import time


ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# (importing the package sizes the BLAS / OpenMP pools to the cgroup CPU quota BEFORE numpy starts them: _hostenv.py)
import mgr_amd  # noqa: E402,F401
from mgr_amd._hostenv import effective_cores  # noqa: E402
import numpy as np  # noqa: E402

MFMA_F32_PEAK_TFLOPS = 157.3  # /opt/skills/guides/MI355X_MICROARCH.md, dense f32 MFMA
# The scans and the wide GEMMs compute their f32 products on the f16 matrix pipe (dense peak 16 x the f32 MFMA rate, same guide) as
# THREE f16 products per f32 product (split-f16 operands): the ceiling for ALGORITHMIC f32 FLOP on that path
MFMA_F16_PEAK_TFLOPS = 16 * MFMA_F32_PEAK_TFLOPS
MFMA_SPLIT16_PEAK_TFLOPS = MFMA_F16_PEAK_TFLOPS / 3.0


def cpu_baseline(spec_dict, seed, T_cpu, B_cpu):
    """Reference-CPU leg: the oracle's fp32 numpy restatement of the Keras op structure, bounded sample."""
    from oracle import network_ref as nr
    rng = np.random.default_rng(seed)
    w = nr.init_weights(spec_dict, rng, np.float32)
    inputs, labels, il, ll = nr.synthetic_batch(spec_dict, B_cpu, T_cpu, 35, rng, np.float32)
    rand = nr.draw_rand(spec_dict, B_cpu, T_cpu, rng, np.float32)
    tr = nr.Trainer(spec_dict, w)
    t0 = time.time()
    loss = tr.train_on_batch(inputs, labels, il, ll, rand)
    dt = time.time() - t0
    return B_cpu * T_cpu / dt, dt, float(loss)


def rank_environments(n, port, base=None):
    """The environments `python bench.py --gpus N` gives its N rank processes when it is its own launcher: what
    torch.distributed.run would set (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*), N distinct LOCAL_RANKs on one node."""
    envs = []
    for r in range(n):
        env = dict(os.environ if base is None else base, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        for v in ("OPENBLAS_NUM_THREADS", "OMP_NUM_THREADS", "MKL_NUM_THREADS"):   # the ranks share the host's CPU quota
            env[v] = str(max(1, effective_cores() // n))
        envs.append(env)
    return envs


def rank_device(local_rank, ndev, world, comm):
    """Device index of a rank and how many ranks share a GPU: rank r of a node binds GPU r (LOCAL_RANK, one process per GPU);
    more ranks than GPUs is only legal with the host communicator (RCCL refuses two ranks on one device) and wraps around."""
    if ndev < 1:
        raise SystemExit("bench.py: no GPU visible")
    if world > ndev and comm != "host":
        raise SystemExit("bench.py: %d ranks on %d GPU(s) needs --comm host" % (world, ndev))
    share = (world + ndev - 1) // ndev if world > ndev else 1
    return local_rank % ndev, share


def _spawn_ranks(n):
    """`python bench.py --gpus N` without torchrun: start N copies of this command, one rank per GPU, rendezvous on
    127.0.0.1; rank 0's JSON line goes to stdout.  Returns the first non-zero exit code."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = [subprocess.Popen([sys.executable] + sys.argv, env=env) for env in rank_environments(n, port)]
    rc = 0
    live = list(procs)
    while live:
        time.sleep(0.2)
        for p in list(live):
            if p.poll() is not None:
                live.remove(p)
                if p.returncode and not rc:   # one rank failed: the others would only wait for it
                    rc = p.returncode
                    for q in live:
                        q.terminate()
    return rc


class Watchdog:
    """A rank that makes no progress for `limit` seconds ends itself with exit code 3 (os._exit from a timer thread: no exec, no
    cleanup that could block on the GPU).  Under torch.distributed.run the launcher then terminates the other ranks; under
    _spawn_ranks the first non-zero exit does.  The first RCCL run with N > 1 must not turn into a hang on somebody's box: a
    communicator that never finishes its rendezvous, an all-reduce a peer never joins, a scan that waits for a workgroup that
    was never placed - all of them stop the beat."""

    def __init__(self, limit, rank):
        import threading
        self.limit, self.rank, self.last, self.where = float(limit), rank, time.time(), "start"
        if self.limit > 0:
            t = threading.Thread(target=self._run, daemon=True)
            t.start()

    def beat(self, where):
        self.last, self.where = time.time(), where

    def _run(self):
        while True:
            time.sleep(min(5.0, max(0.05, self.limit / 4)))
            idle = time.time() - self.last
            if idle > self.limit:
                sys.stderr.write("bench.py: rank %d made no progress for %.0f s (last: %s) - giving up\n" % (self.rank, idle, self.where))
                sys.stderr.flush()
                os._exit(3)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="F")
    ap.add_argument("--batch", type=int, default=0, help="per-GPU batch override")
    ap.add_argument("--maxlen", type=int, default=0)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--show-plan", action="store_true")
    ap.add_argument("--no-pipeline", action="store_true", help="do not overlap the next step's frozen-encoder pass")
    ap.add_argument("--scan-path", type=int, default=0, help="0 auto, 3/4 clusters of 4/8 tiles per workgroup")
    ap.add_argument("--no-parity", action="store_true")
    ap.add_argument("--no-f32-leg", action="store_true", help="skip the second timed leg on the f32 MFMA kernels (tune 14 = 15 = 1)")
    ap.add_argument("--tune", action="append", default=[], help="KEY=VALUE for mgr_tune (A/B of kernel variants; may be repeated)")
    ap.add_argument("--no-transposed", action="store_true", help="dropout-aware projections gather columns of the row-major input")
    ap.add_argument("--scan-with-fproj", action="store_true", help="the deepest encoder scan starts together with the fusion projections (round 4)")
    ap.add_argument("--no-two-ahead", action="store_true", help="the encoder stream is handed a batch's pass one call ahead only (round 4)")
    ap.add_argument("--no-d1-ahead", action="store_true", help="Schedule.depth1_proj_ahead off: the depth-1 projections behind the previous batch's deepest scan (round 4)")
    ap.add_argument("--no-fused-scans", action="store_true", help="Schedule.fused_encoder_scans off: 408 four-wave encoder scan workgroups, two per CU where needed")
    ap.add_argument("--no-fusion-scan-fused", action="store_true", help="Schedule.fusion_scan_fused off: the fusion layer's forward scan as 56 four-wave workgroups (round 5)")
    ap.add_argument("--chain-priority", type=int, default=0, help="Schedule.chain_stream_priority: 1 = stream 0 high, -1 = the encoder stream low")
    ap.add_argument("--pg-two-streams", action="store_true", help="Schedule.param_grads_two_streams: the two directions' dW / dU / db chains on two streams")
    ap.add_argument("--first-pass-inline", action="store_true", help="Schedule.first_pass_on_encoder_stream off: a step without a prefetched pass runs it in line on stream 0 (round 5)")
    ap.add_argument("--du-f32", action="store_true", help="Schedule.du_split off: dU by the f32 split-K product (rounds 1 - 5)")
    ap.add_argument("--bptt-single-cu", action="store_true", help="Schedule.bptt_single_cu: the fusion layer's BPTT on one CU per (direction, 16-sample group), no inter-CU exchange")
    ap.add_argument("--bptt-fused", action="store_true", help="Schedule.bptt_fused: the fusion layer's BPTT as 32 eight-wave workgroups, a CU each")
    ap.add_argument("--bptt-direct", action="store_true", help="beside fused encoder scans the fusion layer's BPTT takes the direct-gather form (one barrier per step)")
    ap.add_argument("--no-fused-wide", action="store_true", help="with fused encoder scans the fusion layer's GEMMs keep their 4-wave tiles")
    ap.add_argument("--bptt-lean", action="store_true", help="the fusion layer's BPTT in its trimmed form also beside the encoder scans (Schedule.bptt_yields_beside_scans=False)")
    ap.add_argument("--no-split-rows", action="store_true", help="transposed copies as f32 rows, converted by every product (round 4's kernels)")
    ap.add_argument("--cpu-T", type=int, default=0, help="T of the CPU leg's sample; 0 (default) = the configuration's own T: the full step")
    ap.add_argument("--cpu-B", type=int, default=0, help="batch of the CPU leg's sample; 0 (default) = the configuration's own")
    ap.add_argument("--comm", choices=("rccl", "host"), default="rccl",
                    help="gradient all-reduce: RCCL over xGMI (one GPU per rank) or summed on the host (ranks may share a GPU)")
    ap.add_argument("--rccl-channels", type=int, default=8,
                    help="NCCL_MAX_NCHANNELS for the gradient all-reduce unless the environment sets it (0: RCCL's default)")
    ap.add_argument("--watchdog", type=float, default=120.0, help="seconds without progress after which a rank exits with code 3 (0: off)")
    ap.add_argument("--stall-at-step", type=int, default=-1, help=argparse.SUPPRESS)   # test hook: stop making progress there
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        args.gpus = world
    if world == 1 and args.gpus > 1 and "RANK" not in os.environ:
        # started without a launcher: be the launcher (one child per GPU; this process never touches the GPU)
        sys.exit(_spawn_ranks(args.gpus))

    dog = Watchdog(args.watchdog, rank)
    import mgr_amd  # noqa: F401
    from mgr_amd import _capi
    from mgr_amd.configs import baseline_config
    from mgr_amd.engine import Engine
    from mgr_amd.parallel import RcclComm
    from mgr_amd.synthetic import synthetic_arrays, synthetic_weights

    from mgr_amd.parallel import HostComm
    spec, B, T, Lmax = baseline_config(args.config)
    if args.batch:
        B = args.batch
    if args.maxlen:
        T = args.maxlen
    ndev = _capi.device_count()
    dev_index, share = rank_device(local_rank, ndev, world, args.comm)
    if share > 1 and not args.batch:
        # ranks that share a GPU: each rank's persistent scans must be co-resident with the other ranks', so the per-rank batch is divided
        B = max(16, B // share)
    dev = _capi.Device(dev_index)
    dog.beat("device")
    if args.show_plan:
        dev.call("mgr_tune", 2, 1)
    if args.scan_path:
        dev.call("mgr_tune", 0, args.scan_path)
    for kv in args.tune:
        k, v = kv.split("=")
        dev.call("mgr_tune", int(k), int(v))

    comm = None
    if world > 1:
        if args.comm == "host":
            comm = HostComm(dev, rank, world)
        else:
            # rendezvous over the launcher's MASTER_ADDR / MASTER_PORT (+101); importing torch here would pull the wheel's
            # own HIP / HSA / RCCL copies into the process next to the ROCm installation's
            from mgr_amd.parallel import tcp_bootstrap
            comm = RcclComm(dev, rank, world, tcp_bootstrap(rank, world, timeout=max(30.0, args.watchdog)), max_channels=args.rccl_channels)
    dog.beat("communicator")

    from mgr_amd.engine import Schedule
    eng = Engine(spec, B, T, Lmax, device=dev, seed=1000 + rank, comm=comm, world=world,
                 schedule=Schedule(transposed_inputs=not args.no_transposed, split_rows=not args.no_split_rows,
                                   deepest_scan_after_fusion_proj=not args.scan_with_fproj, depth1_proj_ahead=not args.no_d1_ahead,
                                   bptt_yields_beside_scans=not args.bptt_lean, fused_encoder_scans=not args.no_fused_scans, fused_wide_tiles=not args.no_fused_wide, bptt_direct_when_alone=args.bptt_direct, fusion_scan_fused=not args.no_fusion_scan_fused, bptt_fused=args.bptt_fused, chain_stream_priority=args.chain_priority, param_grads_two_streams=args.pg_two_streams, first_pass_on_encoder_stream=not args.first_pass_inline, bptt_single_cu=args.bptt_single_cu, du_split=not args.du_f32))
    eng.set_weights(synthetic_weights(spec, 20131900 + 3))
    xs, labels, il, ll = synthetic_arrays(spec, B, T, Lmax, 20131900 + 3 + 17 * rank)
    eng._upload_inputs(xs, None, True)
    eng._upload_labels(labels, il, ll)
    dev.sync()
    dog.beat("engine")

    def step(prefetch, prefetch2=False):
        # frozen encoders: the encoder pass of the NEXT step runs concurrently with this step's fusion / CTC / BPTT /
        # Adam (Engine.can_pipeline).  Never across the timing boundary: the last warm-up and the last timed step do
        # not prefetch, so exactly K complete steps - K encoder passes, K fusion passes - lie inside the timed region.
        eng.enqueue_train_step(None, None, None, None, rand=None, apply_update=True, upload=False,
                               prefetch_next=prefetch and not args.no_pipeline,
                               prefetch_after_next=prefetch2 and not args.no_pipeline and not args.no_two_ahead)
        # (world > 1: the global loss arrives with the gradient all-reduce at the end of the step; the loop paces itself on the
        # rank's own loss - one read-back per step, like N = 1 - and the line reports the global loss of the last step)
        loss = eng.read_loss(local=True)
        dog.beat("step %d" % (eng._step_id - 1))
        if eng._step_id - 1 == args.stall_at_step:
            time.sleep(1e6)
        return loss

    def timed_region():
        """W untimed warm-up steps, then exactly K steps between barrier + device-sync pairs."""
        for i in range(args.warmup):
            step(i + 1 < args.warmup, i + 2 < args.warmup)
        dev.prof_enable((1 << len(_capi.KERNEL_FAMILIES)) - 1)
        dev.prof_reset()
        if isinstance(comm, HostComm):
            comm.allreduce_ms(reset=True)
        if comm:
            comm.barrier()
        dev.sync()
        t0 = time.perf_counter()
        losses, marks = [], [t0]
        for i in range(args.steps):
            # (prefetch2: the step after the next one is inside the timed region as well - the encoder stream may be handed the
            # first part of its pass a call early; nothing crosses the boundary of the region)
            losses.append(step(i + 1 < args.steps, i + 2 < args.steps))
            marks.append(time.perf_counter())     # (the moment the host has step i's loss: diagnostic only, `value` is frames / dt)
        dev.sync()
        if comm:
            comm.barrier()
        return time.perf_counter() - t0, losses, marks

    dt, losses, marks = timed_region()
    if comm and world > 1:
        losses[-1] = eng.read_global_loss()     # mean over the global batch, identical on every rank
    if comm:
        dt = comm.allreduce_max_scalar(dt)
    dog.beat("timed region done")
    fam = {}
    for i, name in enumerate(_capi.KERNEL_FAMILIES):
        n, ms = dev.prof_get(i)
        fam[name] = {"launches": n, "ms": round(ms, 3)}
    dev.prof_enable(0)
    import ctypes
    n_persist, n_serial = ctypes.c_int(), ctypes.c_int()
    dev.call("mgr_persist_stats", ctypes.byref(n_persist), ctypes.byref(n_serial))
    # residency waits of the whole run (warm-up and every leg so far): how many were enqueued and how many ran into their bound - a
    # wait at its bound costs its stream Schedule.resident_wait_us (2 ms) silently; the fused schedule is healthy when the second is 0
    n_waits, n_waits_bound = eng.resident_wait_stats()

    frames = B * T * world * args.steps
    value = frames / dt
    ms_per_step = dt / args.steps * 1e3

    # ---- what the communicator itself saw: a multi-GPU line must prove "N ranks met" from the line alone.
    # rccl: ncclCommCount / ncclCommUserRank of the communicator + the device time of the gradient all-reduces (HIP events on
    # stream 0 around ncclAllReduce); host: the ranks that connected at rank 0 + the host wall time of the exchange.
    comm_info = None
    if comm is not None:
        seen, me = comm.ranks_seen()
        n_ar, ms_ar = comm.allreduce_ms() if isinstance(comm, HostComm) else (fam["allreduce"]["launches"], fam["allreduce"]["ms"])
        comm_info = {"backend": args.comm, "nranks_seen": seen, "rank_seen": me, "world_size_env": world,
                     "local_rank_env": local_rank, "device_index": dev.index, "allreduces": n_ar,
                     "allreduce_ms_per_step": round(ms_ar / max(1, args.steps), 4),
                     "allreduce_timing": "host wall time (download excluded)" if isinstance(comm, HostComm) else "HIP events around ncclAllReduce on stream 0"}
        if seen != world:
            raise SystemExit("bench.py: the communicator reports %d ranks, the launcher %d" % (seen, world))

    # ---- the same K steps on the f32 MFMA kernels (tune keys 14 = 15 = 1), same process, same warm-up: the strict-f32 time beside
    # the split-f16 headline.  `value` above is never touched by it.
    f32_leg = None
    tuned0 = dict(kv.split("=") for kv in args.tune)
    if not args.no_f32_leg and world == 1 and tuned0.get("14", "0") == "0" and tuned0.get("15", "0") == "0":
        dev.call("mgr_tune", 14, 1)
        dev.call("mgr_tune", 15, 1)
        try:
            dt32, losses32, _ = timed_region()
        finally:
            dev.call("mgr_tune", 14, 0)
            dev.call("mgr_tune", 15, 0)
            dev.prof_enable(0)
        f32_leg = {"ms_per_step": round(dt32 / args.steps * 1e3, 3), "value": round(B * T * args.steps / dt32, 1), "unit": "frames/s",
                   "steps": args.steps, "warmup": args.warmup, "loss": losses32[-1],
                   "what": "the same engine and schedule with --tune 14=1 --tune 15=1: every product on v_mfma_f32_*_f32"}
        dog.beat("f32 leg done")

    out = None
    if rank == 0:
        # ---- roofline of the dominant kernel family (device time from HIP events on the launch streams) ----
        dom = max(fam, key=lambda k: fam[k]["ms"])
        flops_family = {}
        gemm_nn = scan_fwd = scan_fwd_narrow = scan_bwd = gemm_tn = gemm_nt = 0
        # (a multi-scan call is counted as scan_fwd when its widest layer has H > 128 - the encoder depths of config F - and as
        #  scan_fwd_narrow otherwise: include/mgr.h.  Every call of the reference networks holds layers of one depth: widths of one class)
        for prefix, fin, H, _, tr in spec.lstm_layers():
            gemm_nn += 2 * 2 * fin * 4 * H
            if H > 128:
                scan_fwd += 2 * 2 * H * 4 * H
            else:
                scan_fwd_narrow += 2 * 2 * H * 4 * H
            if tr:
                scan_bwd += 2 * 2 * H * 4 * H
                gemm_tn += 2 * 2 * (fin * 4 * H + H * 4 * H)
        flops_family = {"gemm_nn": gemm_nn, "scan_fwd": scan_fwd, "scan_fwd_narrow": scan_fwd_narrow, "scan_bwd": scan_bwd, "gemm_tn": gemm_tn,
                        "gemm_nt": gemm_nt}
        roof = None
        if dom in flops_family and fam[dom]["ms"] > 0:
            fl = flops_family[dom] * B * T * args.steps
            ach = fl / (fam[dom]["ms"] * 1e-3) / 1e12
            # HBM bytes per launch of that family from the committed PMC pass (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in
            # their own runs, tools/profile_round.sh + summarize_profile.py) - only if that pass was taken on THIS tree
            # (hash of the device sources + engine schedule); null otherwise
            traffic = traffic_kernel = None
            tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
            if args.config == "F" and world == 1 and os.path.exists(tpath):
                from mgr_amd._build import source_hash
                with open(tpath) as fh:
                    rec = json.load(fh)
                if rec.get("src_sha") == source_hash():
                    traffic = rec.get("bytes_per_launch", {}).get(dom)       # (the family's dominant KERNEL in the profiled run)
                    traffic = round(traffic) if traffic else None
                    traffic_kernel = rec.get("family_kernel", {}).get(dom)
            tuned = dict(kv.split("=") for kv in args.tune)
            split16 = {"scan_fwd": tuned.get("14", "0") == "0", "scan_fwd_narrow": tuned.get("14", "0") == "0", "gemm_nn": tuned.get("15", "0") == "0",
                       "gemm_tn": tuned.get("15", "0") == "0"}.get(dom, False)
            peak = MFMA_SPLIT16_PEAK_TFLOPS if split16 else MFMA_F32_PEAK_TFLOPS
            roof = {"bound": "mfma", "kernel": dom, "achieved": round(ach, 3), "peak": round(peak, 1), "unit": "TFLOP/s",
                    "frac": round(ach / peak, 5), "traffic": traffic, "traffic_kernel": traffic_kernel,
                    "avg_launch_ms": round(fam[dom]["ms"] / max(1, fam[dom]["launches"]), 4),
                    # achieved = algorithmic f32 FLOP / device time.  peak: the instruction mix that kernel issues - f16 MFMA
                    # (2516.8 TF dense) at three f16 products per f32 product, or the f32 MFMA rate; both fractions are given
                    "mfma_operands": "f16 (hi, lo) pairs of f32 values, f32 accumulate" if split16 else "f32",
                    "frac_of_f32_mfma_peak": round(ach / MFMA_F32_PEAK_TFLOPS, 5),
                    "executed_f16_tflops": round(3 * ach, 3) if split16 else None}
        whole = spec.flops_per_frame() * value / 1e12                    # algorithmic (dense) FLOP of SURVEY 8(d)
        whole_ex = spec.flops_per_frame(executed=True) * value / 1e12     # what the dropout-aware kernels really multiply
        # ---- parity: same weights / batch / injected randomness on a short-T slice vs the fp64 oracle ----
        parity = None
        dog.limit = max(dog.limit, 600.0) if dog.limit > 0 else 0   # (the CPU legs below are minutes of host work, not a hang)
        if not args.no_parity and world == 1:
            parity = loss_parity(spec, dev, T, dict(kv.split("=") for kv in args.tune))
        cpu = None
        if not args.no_cpu and world == 1:   # the CPU leg is reported at N=1 only (torchrun also pins OMP_NUM_THREADS=1)
            cpu_T = args.cpu_T or T
            cpu_B = args.cpu_B or B
            v, sec, _ = cpu_baseline(spec.to_dict(), 7, cpu_T, cpu_B)
            full = cpu_T == T and cpu_B == B
            cpu = {"value": round(v, 2), "unit": "frames/s", "cores": effective_cores(), "kind": "port",
                   "sample": "%s: one train step of the same network, B=%d T=%d, numpy/OpenBLAS fp32 oracle, %.1f s"
                             % ("full" if full else "sub-sample (--cpu-T / --cpu-B)", cpu_B, cpu_T, sec)}
        out = {"metric": "train frames/sec, fusion BiLSTM+CTC" if args.config == "F" else "train frames/sec, config " + args.config,
               "value": round(value, 1), "unit": "frames/s",
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3),
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
               "dtype_note": "f32 tensors, f32 accumulation; the recurrent and wide input/weight-gradient products run on the f16 MFMA pipe "
                             "as three f16 products of (hi, lo)-split f32 operands (22+ significant bits each; --tune 14=1 --tune 15=1: "
                             "f32 MFMA); parity bounds in tests/ are unchanged",
               "config": {"workload": "BASELINE configs[2]: multimodal_fusion fusion BiLSTM+CTC train step "
                                      "(frozen audio 2xBiLSTM(500)+skeletal 2xBiLSTM(300), fusion BiLSTM(100), Dense 22, CTC)"
                          if args.config == "F" else args.config,
                          "per_gpu_batch": B, "global_batch": B * world, "maxlen": T, "parallelism": "dp%d" % world,
                          "comm": (args.comm if world > 1 else None), "ranks_per_gpu": share,
                          "rccl_max_channels": (RcclComm.max_channels_in_effect if world > 1 and args.comm == "rccl" else None)},
               "roofline": roof, "cpu_baseline": cpu,
               # (algorithmic = the dense f32 FLOP of SURVEY 8(d); executed = without the products input dropout zeroes.  Fractions
               # of the f32 MFMA peak - above 1 is possible since most products run on the f16 pipe - and of the split-f16 ceiling)
               "whole_step_tflops_algorithmic": round(whole, 3),
               "whole_step_frac_of_f32_mfma_peak_algorithmic": round(whole / MFMA_F32_PEAK_TFLOPS, 5),
               "whole_step_frac_of_split16_peak_algorithmic": round(whole / MFMA_SPLIT16_PEAK_TFLOPS, 5),
               "whole_step_tflops_executed": round(whole_ex, 3),
               "whole_step_frac_of_f32_mfma_peak_executed": round(whole_ex / MFMA_F32_PEAK_TFLOPS, 5),
               "persistent_launches": {"total": n_persist.value, "serialised_by_admission": n_serial.value,
                                       "residency_waits": n_waits, "waits_at_bound": n_waits_bound},
               "loss": losses[-1], "ctc_loss_parity": parity, "f32_mfma_path": f32_leg, "comm": comm_info, "kernel_ms": fam,
               "host_step_ms": {"median": round(sorted(b - a for a, b in zip(marks, marks[1:]))[len(marks) // 2 - 1 if len(marks) > 1 else 0] * 1e3, 3),
                                "max": round(max(b - a for a, b in zip(marks, marks[1:])) * 1e3, 3),
                                "argmax": int(max(range(len(marks) - 1), key=lambda i: marks[i + 1] - marks[i]))},
               "speedup_vs_cpu": round(value / cpu["value"], 1) if cpu else None}
        print(json.dumps(out))
    if comm:
        comm.barrier()   # nobody tears its communicator down while a peer is still in a collective
        comm.close()
    return out


def loss_parity(spec, dev, T_full, tuned=None):
    """CTC-loss delta vs the fp64 oracle on identical inputs: same network, the config's full sequence length, two
    sequences (the fp64 CPU side then takes a few seconds)."""
    from mgr_amd.engine import Engine
    from mgr_amd.synthetic import synthetic_arrays, synthetic_weights
    from oracle import network_ref as nr
    B, T, Lmax = 2, T_full, 35
    eng = Engine(spec, B, T, Lmax, device=dev, seed=5)
    w = synthetic_weights(spec, 99)
    eng.set_weights(w)
    xs, labels, il, ll = synthetic_arrays(spec, B, T, Lmax, 123, lmin=3, lmax=10)
    sd = spec.to_dict()
    rng = np.random.default_rng(4)
    rand = nr.draw_rand(sd, B, T, rng)
    w64 = {k: v.astype(np.float64) for k, v in w.items()}
    ref, _, gref, _ = nr.loss_and_grads(sd, w64, xs, labels, il, ll, rand)

    def run():
        eng.enqueue_train_step(xs, labels, il, ll, rand=rand, apply_update=False)
        got = float(eng.loss_mean.download()[0])
        g = eng.get_grads()
        # largest error of any trainable gradient tensor, relative to that tensor's largest entry
        gerr = max(float(np.abs(g[k] - gref[k]).max() / max(np.abs(gref[k]).max(), 1e-30)) for k in gref)
        return got, gerr

    got, gerr = run()
    out = {"gpu": got, "oracle_fp64": ref, "rel_delta": abs(got - ref) / abs(ref), "grad_max_rel_err": gerr,
           "shape": "B=%d,T=%d" % (B, T)}
    # the same step with the products on the f32 MFMA kernels (tune keys 14 / 15): what the split-f16 arithmetic of the default
    # path costs in accuracy against the same fp64 oracle - nothing
    tuned = tuned or {}
    dev.call("mgr_tune", 14, 1)
    dev.call("mgr_tune", 15, 1)
    try:
        got32, gerr32 = run()
    finally:
        dev.call("mgr_tune", 14, int(tuned.get("14", 0)))
        dev.call("mgr_tune", 15, int(tuned.get("15", 0)))
    out["f32_mfma_path"] = {"gpu": got32, "rel_delta": abs(got32 - ref) / abs(ref), "grad_max_rel_err": gerr32}
    eng.close()
    return out


if __name__ == "__main__":
    main()
