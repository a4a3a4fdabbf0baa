# bash tools/ab_lib_quick.sh <variant> ... : one 30-step bench line per library variant (tools/build_variants.sh), BENCH_ARGS passed on
R=$GRAFT_REPO_ROOT; PKG=$R/multimodal-gesture-recognition-with-lstms-and-ctc_amd
cp $PKG/libmgr.so /tmp/shipped.so
trap 'cp /tmp/shipped.so $PKG/libmgr.so' EXIT
for V in "$@"; do cp $R/variants/lib_$V.so $PKG/libmgr.so; cd $R && (timeout 300 python bench.py --steps 30 --no-cpu --no-parity --no-f32-leg $BENCH_ARGS 2>/dev/null || true) | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$V', d['ms_per_step'], {k:round(v['ms']/max(v['launches'],1),3) for k,v in d['kernel_ms'].items() if v['launches']})"; done
