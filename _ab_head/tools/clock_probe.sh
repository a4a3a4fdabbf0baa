# Samples GPU clock / power while a GEMM loop runs (diagnostic: is the f32-MFMA GEMM power-limited?)
cd $GRAFT_REPO_ROOT
(for i in $(seq 1 40); do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power|fclk|mclk" | tr '\n' ' '; echo; sleep 0.25; done) > gpurun_out/clock_samples.txt &
SAMPLER=$!
sleep 1.5
timeout 100 python - <<'PY'
import sys, os, numpy as np
sys.path.insert(0, os.getcwd())
import mgr_amd
from mgr_amd import _capi
dev = _capi.Device(0)
B,T,F,H = 64,1900,1000,500
rng = np.random.default_rng(0)
X = dev.array(rng.standard_normal((B,T,F)).astype(np.float32)); Wp = dev.array(rng.standard_normal((F,4*H)).astype(np.float32)*0.05)
bp = dev.zeros((4*H,)); Z = dev.empty((B,T,4*H))
import time
t0=time.time()
n=0
while time.time()-t0 < 6.0:
    for _ in range(50):
        dev.call("mgr_lstm_input_proj", X, F, 0, Wp, bp, Z, B, T, F, H)
    dev.sync(); n+=50
dt=time.time()-t0
print("sustained: %.1f TF over %.1f s" % (n*2.0*B*T*F*4*H/dt/1e12, dt))
PY
wait $SAMPLER
cat gpurun_out/clock_samples.txt | sed -n 1,40p
