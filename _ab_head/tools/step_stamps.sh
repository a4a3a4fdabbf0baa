R=$GRAFT_REPO_ROOT; PKG=$R/multimodal-gesture-recognition-with-lstms-and-ctc_amd
cp $PKG/libmgr.so /tmp/shipped.so
trap 'cp /tmp/shipped.so $PKG/libmgr.so' EXIT
for V in "$@"; do cp $R/variants/lib_$V.so $PKG/libmgr.so; echo "== $V"; cd $R && timeout 300 python tools/step_stamps.py 2>&1 | tail -5; done
