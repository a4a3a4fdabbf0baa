# Usage (on the GPU box): bash tools/ab_quick.sh <rounds> "<bench args A>" "<bench args B>" ...
# Round-robin A/B of bench.py variants on ONE box (30 steps each, no profiler): prints ms per step per variant and round.  Box to box the
# step moves by ~0.5 %, run to run by ~0.1 ms: compare variants of the same call only.
R=$GRAFT_REPO_ROOT; N=$1; shift
for r in $(seq 1 $N); do
  i=0
  for ARGS in "$@"; do
    v=$(cd $R && timeout 300 python bench.py --steps 30 --no-cpu --no-parity --no-f32-leg $ARGS 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['persistent_launches'].get('waits_at_bound'), {k: round(v['ms']/max(v['launches'],1),2) for k,v in d['kernel_ms'].items() if v['launches']})")
    echo "round $r v$i [$ARGS]: $v"
    i=$((i+1))
  done
done
