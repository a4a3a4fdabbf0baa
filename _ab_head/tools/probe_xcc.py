import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mgr_amd
from mgr_amd import _capi
dev = _capi.Device(0)
for (n, th, lds) in [(256, 512, 86016), (208, 512, 86016), (408, 256, 65536), (56, 256, 20000), (152, 256, 86016)]:
    out = dev.empty((n,), np.int32)
    for rep in range(3):
        dev.call("mgr_probe_xcc", n, th, lds, out)
        x = out.download()
        ok = np.all(x == (x[0] + np.arange(n)) % 8)
        print(n, th, lds, "rep", rep, "first8", x[:8].tolist(), "pure round-robin:", bool(ok), "counts", np.bincount(x, minlength=8).tolist())
