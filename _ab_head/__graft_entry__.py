"""Driver hooks: build() compiles the HIP library for gfx950; smoke() runs one tiny train step on cuda:0."""
import os
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def build():
    """hipcc --offload-arch=gfx950 every translation unit of csrc/ into libmgr.so (in-tree) and import the package.
    The oracle is numpy (nothing to compile); the reference is Python 2 + Keras/TF and unbuildable."""
    import mgr_amd  # noqa: F401
    from mgr_amd import _build, _capi
    _build.build(verbose=True)
    lib = _capi.load_library()
    assert lib.mgr_version() >= 100
    missing = [n for n in _capi.SIGNATURES if not hasattr(lib, n)]
    assert not missing, missing


def smoke():
    """One tiny fusion train step (fwd + CTC + BPTT + Adam) on the GPU, checked against the fp64 oracle."""
    import numpy as np
    import mgr_amd  # noqa: F401
    from mgr_amd.engine import Engine
    from mgr_amd.spec import NetworkSpec
    from oracle import network_ref as nr
    from tests.helpers import load_case
    z, meta, grab = load_case("fusion_tiny")
    spec = NetworkSpec.from_dict(meta["spec"])
    eng = Engine(spec, meta["B"], meta["T"], meta["Lmax"], device=0)
    eng.set_weights(grab("w__"))
    loss = eng.train_step(grab("x__"), z["labels"], z["input_length"], z["label_length"], rand=grab("rs0__"))
    ref = float(z["traj"][0])
    assert abs(loss - ref) <= 1e-4 * abs(ref), (loss, ref)
    tr = nr.Trainer(meta["spec"], {k: v.astype(np.float64) for k, v in grab("w__").items()})
    ref2 = tr.train_on_batch(grab("x__"), z["labels"], z["input_length"], z["label_length"], grab("rs0__"))
    assert abs(loss - ref2) <= 1e-4 * abs(ref2), (loss, ref2)
    w = eng.get_weights()
    for k, v in tr.w.items():
        assert np.allclose(w[k], v, rtol=1e-4, atol=1e-6), k
    print("smoke ok: loss %.6f (oracle %.6f) on %s" % (loss, ref2, eng.dev.name))
    eng.close()


if __name__ == "__main__":
    build()
    if "--smoke" in sys.argv:
        smoke()
