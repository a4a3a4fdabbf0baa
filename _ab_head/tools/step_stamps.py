"""Cycle stamps of the scan kernels INSIDE the training step: runs bench.py's main() in this process on a -DMGR_STAMP build of libmgr.so
(tools/build_variants.sh "<name>:-DMGR_STAMP", copied over the package's libmgr.so by the caller) and prints the stamp accumulators."""
import ctypes as C, io, os, sys, contextlib
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
sys.argv = ["bench.py", "--steps", "30", "--no-cpu", "--no-parity", "--no-f32-leg"] + sys.argv[1:]
import bench  # noqa: E402
buf = io.StringIO()
with contextlib.redirect_stdout(buf):
    bench.main()
import json  # noqa: E402
d = json.loads(buf.getvalue().strip().splitlines()[-1])
print("ms_per_step", d["ms_per_step"], {k: round(v["ms"] / max(v["launches"], 1), 3) for k, v in d["kernel_ms"].items() if v["launches"]})
lib = C.CDLL(os.path.join(R, "multimodal-gesture-recognition-with-lstms-and-ctc_amd", "libmgr.so"))
out = (C.c_ulonglong * 64)()
if hasattr(lib, "mgr_debug_stamps"):
    lib.mgr_debug_stamps(out)
    for cls, name in ((0, "fwd H>400"), (16, "fwd H<=400")):
        if out[cls + 8]:
            n = float(out[cls + 8])
            print("  %-10s cycles/step: gather %5.0f | mfma+partials %5.0f | wait+barrier %5.0f | reduce+cell+flags %5.0f | publish %5.0f | outputs %5.0f  (sum %5.0f; %.2f re-fetch rounds)"
                  % ((name,) + tuple(out[cls + i] / n for i in range(6)) + (sum(out[cls + i] for i in range(6)) / n, out[cls + 9] / n)))
if hasattr(lib, "mgr_debug_bstamps"):
    lib.mgr_debug_bstamps(out)
    if out[8]:
        n = float(out[8])
        names = ("gather+verify", "reduce+barrier", "cell bwd+dZ", "scale+image", "barrier", "mfma+publish")
        print("  bptt       cycles/step: " + " | ".join("%s %5.0f" % (nm, out[i] / n) for i, nm in enumerate(names)) + "  (sum %5.0f; %.2f re-fetch rounds)"
              % (sum(out[i] for i in range(6)) / n, out[9] / n))
