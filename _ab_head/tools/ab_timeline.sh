# bash tools/ab_timeline.sh <variant> ... : steady-state timeline (tools/trace_steps.sh + tools/steady_timeline.py) per library variant
R=$GRAFT_REPO_ROOT; PKG=$R/multimodal-gesture-recognition-with-lstms-and-ctc_amd
cp $PKG/libmgr.so /tmp/shipped.so
trap 'cp /tmp/shipped.so $PKG/libmgr.so' EXIT
for V in "$@"; do cp $R/variants/lib_$V.so $PKG/libmgr.so; echo "== $V"; bash $R/tools/trace_steps.sh tl_$V 12; cd $R && python tools/steady_timeline.py gpurun_out/tl_${V}_kernel_trace.csv > gpurun_out/tl_$V.txt 2>&1; rm -f gpurun_out/tl_${V}_kernel_trace.csv; done
