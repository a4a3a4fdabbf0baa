run() { (cd $GRAFT_REPO_ROOT && python bench.py --steps 30 --no-cpu --no-parity --no-f32-leg "$@" 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); k=d['kernel_ms']; print(d['ms_per_step'], d['persistent_launches'].get('waits_at_bound'), d.get('loss'), {n: round(k[n]['ms']/max(k[n]['launches'],1),3) for n in ('gemm_tn','misc') if n in k})"); }
for r in 1 2 3; do
  echo "round $r du f32:   $(run --du-f32)"
  echo "round $r du split: $(run)"
done
for C in E S_ref A_ref; do
  echo "$C du f32:   $(run --config $C --steps 10 --du-f32)"
  echo "$C du split: $(run --config $C --steps 10)"
done
