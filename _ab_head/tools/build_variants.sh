# Usage: bash tools/build_variants.sh "<name>:<extra hipcc flags>" ...   (CPU container)
# Builds one libmgr.so per variant into variants/lib_<name>.so (git-ignored, travels with gpurun) for same-box A/B runs with
# tools/ab_lib.sh. Objects go to /tmp, the shipped build/ and libmgr.so are left alone.
set -e
R=$(cd $(dirname $0)/.. && pwd)
PKG=$R/multimodal-gesture-recognition-with-lstms-and-ctc_amd
mkdir -p $R/variants
for V in "$@"; do
  NAME=${V%%:*}; FLAGS=${V#*:}
  O=/tmp/mgr_variant_$NAME; mkdir -p $O
  for S in ctx elementwise ctc dense gemm gemm_split lstm_simple lstm_mfma lstm_cluster lstm_cluster_bwd lstm comm beam skeletal; do
    echo "hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-value -Wno-unused-result $FLAGS -c $PKG/csrc/$S.hip -o $O/$S.o"
  done | xargs -P 6 -I{} sh -c "{}"
  hipcc --offload-arch=gfx950 -shared -fPIC -o $R/variants/lib_$NAME.so $O/*.o -ldl
  echo "built variants/lib_$NAME.so ($FLAGS)"
done
