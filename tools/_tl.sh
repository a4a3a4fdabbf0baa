cd $GRAFT_REPO_ROOT
grep -E "nr_throttled|throttled_usec|nr_periods" /sys/fs/cgroup/cpu.stat | tr '\n' ' '; echo
for k in 1 2 3 4 5 6; do python bench.py --config S --steps 10 --no-cpu --no-parity 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d.get('host_step_ms'))"; grep -E "nr_throttled|throttled_usec|nr_periods" /sys/fs/cgroup/cpu.stat | tr '\n' ' '; echo; done
python bench.py 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('F', d['ms_per_step'], d.get('host_step_ms'), d['cpu_baseline'])"; grep -E "nr_throttled|throttled_usec|nr_periods" /sys/fs/cgroup/cpu.stat | tr '\n' ' '; echo
