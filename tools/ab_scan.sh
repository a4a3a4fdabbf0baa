R=$GRAFT_REPO_ROOT; PKG=$R/multimodal-gesture-recognition-with-lstms-and-ctc_amd
cp $PKG/libmgr.so /tmp/shipped.so
trap 'cp /tmp/shipped.so $PKG/libmgr.so' EXIT
for V in "$@"; do cp $R/variants/lib_$V.so $PKG/libmgr.so; echo "== $V"; cd $R && SCAN_PROBE_H=${SCAN_PROBE_H:-500+300,500} timeout 200 python tools/scan_variant_probe.py "4=1" "4=2" 2>&1 | tail -4; done
