import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_case(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    meta = json.load(open(os.path.join(GOLDEN, name + ".json")))

    def grab(prefix):
        return {k[len(prefix):].replace("__", "/"): z[k] for k in z.files if k.startswith(prefix)}

    return z, meta, grab


def rel_err(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(1e-30, np.abs(b).max()))
