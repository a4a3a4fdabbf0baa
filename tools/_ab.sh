cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -k "cluster or scan" 2>&1 | tail -3
for t in "14 2" "14 0" "14 2" "14 0"; do MGR_TUNE="$t" timeout 300 python bench.py --no-cpu --no-parity 2>&1 | tail -1 | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read()); print('tune=$t', d['value'], d['ms_per_step'], {k:round(v['ms']/max(1,v['launches']),2) for k,v in d['kernel_ms'].items() if v['launches']})
except Exception as e: print('tune=$t failed', e)"; done
