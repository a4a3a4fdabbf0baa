"""Build libmgr.so (the HIP/gfx950 C-ABI library) in-tree with hipcc.

The product never falls back to a CPU path: if the library is missing and cannot be built,
importing the compute layer raises.
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libmgr.so")
SOURCES = ["ctx.hip", "elementwise.hip", "ctc.hip", "dense.hip", "gemm.hip", "lstm_simple.hip", "lstm_mfma.hip",
           "lstm_cluster.hip", "lstm_cluster_bwd.hip", "lstm.hip", "comm.hip", "beam.hip", "skeletal.hip"]
ARCH = "gfx950"


def _hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found; cannot build libmgr.so")


def source_hash():
    """sha256 over the device sources, the C ABI header and the engine's schedule: profiles/pmc_traffic.json records it, and
    bench.py reports the profiled HBM traffic only for the tree it was measured on."""
    import hashlib
    h = hashlib.sha256()
    files = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".h")))
    files += [os.path.join(HERE, "..", "include", "mgr.h"), os.path.join(HERE, "engine.py")]
    for f in files:
        h.update(os.path.basename(f).encode() + b"\0")
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(HERE, "..", "include", "mgr.h"), os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build(force=False, verbose=True, jobs=4):
    """Compile every translation unit for gfx950 and link libmgr.so next to this file."""
    if not force and not _stale():
        return LIB
    hipcc = _hipcc()
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    flags = ["--offload-arch=" + ARCH, "-O3", "-fPIC", "-std=c++17", "-Wno-unused-value", "-Wno-unused-result"]
    flags += os.environ.get("MGR_CXXFLAGS", "").split()  # e.g. -DMGR_STAMP for the diagnostic build
    srcs = [s for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    procs = []
    objs = []
    for s in srcs:
        src = os.path.join(CSRC, s)
        obj = os.path.join(objdir, s.replace(".hip", ".o"))
        objs.append(obj)
        hdrs = [os.path.join(CSRC, h) for h in os.listdir(CSRC) if h.endswith(".h")]
        hdrs.append(os.path.join(HERE, "..", "include", "mgr.h"))
        newest = max(os.path.getmtime(p) for p in [src] + hdrs)
        asm_ok = s not in ISA_CHECKED or any(f.startswith(s.replace(".hip", "") + "-hip-amdgcn") and f.endswith(".s")
                                             for f in os.listdir(objdir))
        if not force and os.path.exists(obj) and os.path.getmtime(obj) > newest and asm_ok:
            continue
        cmd = [hipcc] + flags + ["-c", src, "-o", obj]
        if s in ISA_CHECKED:
            cmd.append("--save-temps=obj")   # keeps the device assembly next to the object for check_hidden_loads()
        if verbose:
            print("[mgr build]", " ".join(cmd), file=sys.stderr)
        procs.append((s, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
        while len([p for _, p in procs if p.poll() is None]) >= jobs:
            for _, p in procs:
                if p.poll() is None:
                    p.wait()
                    break
    for s, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            if os.path.exists(LIB):
                os.remove(LIB)
            raise RuntimeError("hipcc failed on %s:\n%s" % (s, out.decode(errors="replace")))
    # the ISA check comes BEFORE the link, and a library of an older build does not survive a failed build: nothing may
    # load (or carry to the GPU box) a libmgr.so whose register-polling kernels were not checked
    try:
        check_hidden_loads(objdir)
        cmd = [hipcc, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", LIB] + objs + ["-ldl"]
        if verbose:
            print("[mgr build]", " ".join(cmd), file=sys.stderr)
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n%s" % r.stdout.decode(errors="replace"))
    except Exception:
        if os.path.exists(LIB):
            os.remove(LIB)
        raise
    # --save-temps leaves bitcode / preprocessed sources behind; only the device assembly is of further use
    for f in os.listdir(objdir):
        if f.endswith((".bc", ".hipi", ".hipfb", ".out", ".cui")) or (f.endswith(".s") and "amdgcn" not in f):
            os.remove(os.path.join(objdir, f))
    return LIB


ISA_CHECKED = {"lstm_cluster.hip": ["k_scan_cluster_ks", "k_scan_cluster_ks_id"]}


def _regs(tok):
    """'v[16:19]' / 'v7' -> set of VGPR indices (empty for anything else)."""
    import re
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r"v(\d+)", tok)
    return {int(m.group(1))} if m else set()


def check_hidden_loads(objdir):
    """The K-split scan step issues its gather loads from inline asm and polls the destination REGISTERS (csrc/
    lstm_cluster.hip, cluster_run_ks).  That only works if hipcc keeps each polled variable in the registers the load
    writes: a compiler-inserted copy taken while the load is in flight freezes a stale value (seen once, in an
    experimental BPTT variant: the workgroup then spins until its bounded give-up).  This check reads the device assembly
    kept by --save-temps and fails the build if any move / spill instruction reads those registers."""
    import re
    for src, kernels in ISA_CHECKED.items():
        stem = src.replace(".hip", "")
        cands = [f for f in os.listdir(objdir) if f.startswith(stem + "-hip-amdgcn") and f.endswith(".s")]
        if not cands:
            raise RuntimeError("ISA check: no device assembly of %s in %s (build() keeps it with --save-temps=obj); the "
                               "register-polling scan step must not ship unchecked" % (src, objdir))
        text = open(os.path.join(objdir, cands[0])).read().split("\n")
        for kname in kernels:
            start = next((i for i, l in enumerate(text) if re.match(r"^_Z\w*\d%sE\w*:" % kname, l)), None)
            if start is None:
                raise RuntimeError("ISA check: kernel %s not found in %s" % (kname, cands[0]))
            end = next(i for i in range(start, len(text)) if "s_endpgm" in text[i])
            # a polling window runs from a step's first hidden load to the statement that drains the loads (poll_end() in
            # cluster_run_ks leaves the marker MGR_POLL_END in the assembly); MFMAs inside it read the registers in place
            polled, nloads, nends = set(), 0, 0
            for raw in text[start:end]:
                if "MGR_POLL_END" in raw:
                    polled = set()
                    nends += 1
                    continue
                l = raw.split(";")[0].strip()
                if not l or " " not in l:
                    continue
                mnem, rest = l.split(None, 1)
                ops = [o.strip() for o in rest.split(",")]
                if mnem == "global_load_dwordx4" and " sc1" in l:
                    polled |= _regs(ops[0])
                    nloads += 1
                    continue
                if not polled:
                    continue
                if mnem.startswith(("v_mov", "v_accvgpr", "v_swap", "v_cndmask", "v_perm")) and any(_regs(o) & polled for o in ops[1:]):
                    raise RuntimeError("ISA check (%s): '%s' copies a register that a hidden gather load writes while it may still "
                                       "be in flight; the polling loop would watch a stale copy.  Restructure cluster_run_ks "
                                       "(register pressure?)" % (kname, l))
                if mnem.startswith("scratch_store") and any(_regs(o) & polled for o in ops):
                    raise RuntimeError("ISA check (%s): '%s' spills a polled register" % (kname, l))
            if nloads == 0 or nends == 0:
                raise RuntimeError("ISA check: no hidden gather loads / no MGR_POLL_END marker found in %s" % kname)


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
