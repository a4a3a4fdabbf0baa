cd $GRAFT_REPO_ROOT
for t in "14 1" "14 0" "14 1" "14 0"; do MGR_TUNE="$t" timeout 300 python bench.py --no-cpu --no-parity 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('tune=$t', d['value'], d['ms_per_step'], {k:round(v['ms']/max(1,v['launches']),2) for k,v in d['kernel_ms'].items() if v['launches']})"; done
