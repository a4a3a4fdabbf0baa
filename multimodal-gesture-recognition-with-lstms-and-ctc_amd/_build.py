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


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(HERE, "..", "include", "mgr.h")]
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
        if not force and os.path.exists(obj) and os.path.getmtime(obj) > newest:
            continue
        cmd = [hipcc] + flags + ["-c", src, "-o", obj]
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
            raise RuntimeError("hipcc failed on %s:\n%s" % (s, out.decode(errors="replace")))
    cmd = [hipcc, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", LIB] + objs + ["-ldl"]
    if verbose:
        print("[mgr build]", " ".join(cmd), file=sys.stderr)
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    if r.returncode != 0:
        raise RuntimeError("link failed:\n%s" % r.stdout.decode(errors="replace"))
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
