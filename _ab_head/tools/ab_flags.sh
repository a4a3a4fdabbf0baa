# bash tools/ab_flags.sh "<bench flags A>" "<bench flags B>" ... : three rounds of 30-step bench.py lines per flag set, ON the GPU box (ms per step, loss, per-launch ms)
cd $GRAFT_REPO_ROOT
for i in 1 2 3; do for F in "$@"; do (timeout 300 python bench.py --steps 30 --no-cpu --no-parity --no-f32-leg $F 2>/dev/null || true) | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('[$F]', d['ms_per_step'], d['loss'], {k:round(v['ms']/max(v['launches'],1),3) for k,v in d['kernel_ms'].items() if v['launches']})"; done; done
