cd $GRAFT_REPO_ROOT
for cfg in "16 2" "32 1" "32 2" "16 1"; do
  set -- $cfg
  MGR_CXXFLAGS="-DMGR_GEMM_BK=$1 -DMGR_GEMM_NBUF=$2" python multimodal-gesture-recognition-with-lstms-and-ctc_amd/_build.py --force > /dev/null 2>&1
  echo "== BK=$1 NBUF=$2"
  timeout 200 python tools/kernel_bench.py --what gemm 2>&1 | grep -E "gemm" | head -8
done
