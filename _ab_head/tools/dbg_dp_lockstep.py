"""Debug driver: the two-rank lock-step scenario of tests/test_gpu_dataparallel.py, repeated; prints where the replicas' gradient
buffers differ.  python tools/dbg_dp_lockstep.py [reps]  (spawned ranks: no GPU touched in the parent)"""
import multiprocessing as mp, os, socket, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tests import dp_worker

if __name__ == "__main__":
    ctx = mp.get_context("spawn")
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    for rep in range(reps):
        d = tempfile.mkdtemp()
        s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
        outs = [os.path.join(d, "rank%d.npz" % r) for r in range(2)]
        procs = [ctx.Process(target=dp_worker.dp_rank, args=(r, 2, port, outs[r], False, 32, 96, 8, 5, "host")) for r in range(2)]
        for p in procs: p.start()
        for p in procs: p.join(300)
        r0, r1 = np.load(outs[0]), np.load(outs[1])
        bad = [k for k in r0.files if k.startswith(("w__", "g__")) and not np.array_equal(r0[k], r1[k])]
        print("rep", rep, "differ:", bad)
        for k in bad:
            a, b = r0[k], r1[k]
            idx = np.argwhere(a != b)
            rows, cols = sorted(set(int(v) for v in idx[:, 0])), sorted(set(int(v) for v in idx[:, 1]))
            print("   ", k, a.shape, "n diff", len(idx), "nan r0", int(np.isnan(a).sum()), "nan r1", int(np.isnan(b).sum()),
                  "| rows", len(rows), rows[0], "..", rows[-1], "| cols", len(cols), cols[:8], "..", cols[-4:],
                  "| gates", sorted(set(c % 4 for c in cols)), "units", sorted(set(c // 4 for c in cols))[:4], "..", sorted(set(c // 4 for c in cols))[-2:])
