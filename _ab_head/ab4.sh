run() { (cd $GRAFT_REPO_ROOT/$1 && shift && python bench.py --steps 30 --no-cpu --no-parity --no-f32-leg "$@" 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); k=d['kernel_ms']; print(d['ms_per_step'], d['persistent_launches'].get('waits_at_bound'), d.get('loss'), {n: round(k[n]['ms']/max(k[n]['launches'],1),3) for n in ('gemm_nn','gemm_tn','scan_fwd','scan_fwd_narrow','scan_bwd') if n in k})"); }
for r in 1 2 3 4; do
  echo "round $r head:        $(run _ab_head)"
  echo "round $r new:         $(run .)"
done
