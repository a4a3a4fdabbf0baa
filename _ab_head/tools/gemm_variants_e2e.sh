# End-to-end effect of the GEMM staging parameters (built on the GPU box; the tree's own libmgr.so is restored by a final default build)
cd $GRAFT_REPO_ROOT
for cfg in "32 2" "32 1" "16 2"; do
  set -- $cfg
  MGR_CXXFLAGS="-DMGR_GEMM_BK=$1 -DMGR_GEMM_NBUF=$2" python multimodal-gesture-recognition-with-lstms-and-ctc_amd/_build.py --force > /dev/null 2>&1
  echo "== BK=$1 NBUF=$2"
  timeout 200 python bench.py --no-cpu --no-parity --no-f32-leg 2>&1 | tail -1 | cut -c50-140
  timeout 200 python tools/kernel_bench.py --what gemm 2>&1 | grep -E "gemm_nn" | head -2
done
