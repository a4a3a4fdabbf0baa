# Usage (on the GPU box): bash tools/ab_lib.sh <rounds> <name> <name> ...   [BENCH_ARGS="..."]
# Same-box A/B of library variants built by tools/build_variants.sh: the variants are benched round-robin <rounds> times (box to
# box the step time moves by ~0.5 %, which is the size of the effects being compared). The shipped libmgr.so is put back at the end.
N=$1; shift
R=$GRAFT_REPO_ROOT
PKG=$R/multimodal-gesture-recognition-with-lstms-and-ctc_amd
set -e
cp $PKG/libmgr.so /tmp/libmgr_shipped.so
# whatever ends this script - the last line, an interrupt, a timeout of the caller, a missing variant (set -e) - the shipped library
# is back in place before anything else can measure the wrong one
trap 'cp /tmp/libmgr_shipped.so $PKG/libmgr.so' EXIT
for V in "$@"; do test -f $R/variants/lib_$V.so || { echo "missing variants/lib_$V.so" >&2; exit 2; }; done
for r in $(seq 1 $N); do
  for V in "$@"; do
    cp $R/variants/lib_$V.so $PKG/libmgr.so
    cd $R && (timeout 300 python bench.py --steps 30 --no-cpu --no-parity --no-f32-leg $BENCH_ARGS 2>/dev/null || true) | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$V', d['ms_per_step'], {k:round(v['ms']/max(v['launches'],1),3) for k,v in d['kernel_ms'].items() if v['launches']})"
  done
done
