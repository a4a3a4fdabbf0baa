# ON the GPU box: kernel timeline of Engine.predict_stream (pipelined inference, config F sizes) - kernels > 0.15 ms per hardware queue
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pprof
cat > /tmp/pp.py <<'PY'
import os, sys
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import mgr_amd
import numpy as np
from mgr_amd import _capi
from mgr_amd.configs import fusion_spec
from mgr_amd.engine import Engine
from mgr_amd.synthetic import synthetic_arrays, synthetic_weights
dev = _capi.Device(0)
spec = fusion_spec()
B, T = 64, 1900
eng = Engine(spec, B, T, 1, device=dev, seed=1, inference_only=True)
eng.set_weights(synthetic_weights(spec, 5))
chunks = [synthetic_arrays(spec, B, T, 1, 7 + i)[0] for i in range(2)]
feed = lambda n: (chunks[i % 2] for i in range(n))
list(eng.predict_stream(feed(3), output="argmax"))
import time
t0 = time.perf_counter()
list(eng.predict_stream(feed(8), output="argmax"))
print("ms per batch", (time.perf_counter() - t0) / 8 * 1e3)
PY
timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/pprof -- python3 /tmp/pp.py 2>&1 | tail -1
python3 - <<'PY'
import csv, glob
rows=[]
for f in glob.glob('/tmp/pprof/*/*_kernel_trace.csv'):
    for r in csv.DictReader(open(f)): rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), 'q%s %s' % (r.get('Queue_Id','?'), r['Kernel_Name'].replace('(anonymous namespace)::','').replace('void ','')[:26])))
for f in glob.glob('/tmp/pprof/*/*_memory_copy_trace.csv'):
    for r in csv.DictReader(open(f)): rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), 'COPY %s' % (r.get('Direction','')[12:])))
rows.sort()
t_end=rows[-1][1]
sel=[r for r in rows if r[0] > t_end-110e6 and r[0] < t_end - 30e6 and r[1]-r[0] > 0.15e6]
t0=sel[0][0]
for s,e,n in sel: print("%8.2f %8.2f %6.2f %s" % ((s-t0)/1e6,(e-t0)/1e6,(e-s)/1e6,n))
PY
