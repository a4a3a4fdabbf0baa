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
OBJDIR = os.path.join(HERE, "build")
MGR_H = os.path.join(HERE, "..", "include", "mgr.h")
SOURCES = ["ctx.hip", "elementwise.hip", "ctc.hip", "dense.hip", "gemm.hip", "gemm_split.hip", "lstm_simple.hip", "lstm_mfma.hip",
           "lstm_cluster.hip", "lstm_cluster_bwd.hip", "lstm_cu_bwd.hip", "lstm.hip", "comm.hip", "beam.hip", "skeletal.hip"]
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
    files += [MGR_H, os.path.join(HERE, "engine.py")]
    for f in files:
        h.update(os.path.basename(f).encode() + b"\0")
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [MGR_H, os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build(force=False, verbose=True, jobs=4):
    """Compile every translation unit for gfx950 and link libmgr.so next to this file."""
    if not force and not _stale():
        return LIB
    hipcc = _hipcc()
    objdir = OBJDIR
    os.makedirs(objdir, exist_ok=True)
    flags = ["--offload-arch=" + ARCH, "-O3", "-fPIC", "-std=c++17", "-Wno-unused-value", "-Wno-unused-result"]
    srcs = [s for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    procs = []
    objs = []
    hdrs = [os.path.join(CSRC, h) for h in os.listdir(CSRC) if h.endswith(".h")]
    hdrs.append(MGR_H)
    for s in srcs:
        src = os.path.join(CSRC, s)
        obj = os.path.join(objdir, s.replace(".hip", ".o"))
        objs.append(obj)
        newest = max(os.path.getmtime(p) for p in [src] + hdrs)
        if not force and os.path.exists(obj) and os.path.getmtime(obj) > newest:
            continue
        # --save-temps=obj keeps the device assembly next to the object: nothing depends on it (the register-polling scan step
        # whose assembly had to be checked is gone), it is what tools/kasm.sh shows
        cmd = [hipcc] + flags + ["-c", src, "-o", obj, "--save-temps=obj"]
        if verbose:
            print("[mgr build]", " ".join(cmd), file=sys.stderr)
        procs.append((s, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
        while len([p for _, p in procs if p.poll() is None]) >= jobs:
            for _, p in procs:
                if p.poll() is None:
                    p.wait()
                    break
    failed = None
    for s, p in procs:   # every compiler process is waited for before anything is reported: none may outlive a failed build
        out, _ = p.communicate()
        if p.returncode != 0 and failed is None:
            failed = (s, out.decode(errors="replace"))
    try:
        if failed:
            raise RuntimeError("hipcc failed on %s:\n%s" % failed)
        cmd = [hipcc, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", LIB] + objs + ["-ldl"]
        if verbose:
            print("[mgr build]", " ".join(cmd), file=sys.stderr)
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n%s" % r.stdout.decode(errors="replace"))
    except Exception:
        if os.path.exists(LIB):      # a library of an older build does not survive a failed build
            os.remove(LIB)
        raise
    # --save-temps leaves bitcode / preprocessed sources behind; only the device assembly is of further use
    for f in os.listdir(objdir):
        if f.endswith((".bc", ".hipi", ".hipfb", ".out", ".cui")) or (f.endswith(".s") and "amdgcn" not in f):
            os.remove(os.path.join(objdir, f))
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
