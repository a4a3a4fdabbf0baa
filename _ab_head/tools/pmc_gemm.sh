# PMC pass over the GEMM micro-benchmark (counters in their own run, kernel-trace only)
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_gemm
rm -rf $OUT
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/tools/kernel_bench.py --what gemm > $OUT.log 2>&1
tail -3 $OUT.log
ls $OUT/*/
