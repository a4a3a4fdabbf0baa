# bash tools/ab_cfg.sh "<config> ..." <variant> ... : 30-step bench lines of library variants on other BASELINE configurations, two rounds
CFGS=$1; shift
for C in $CFGS; do export BENCH_ARGS="--config $C"; echo "== $C"; bash $GRAFT_REPO_ROOT/tools/ab_lib_quick.sh "$@" "$@"; done
