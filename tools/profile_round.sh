# Usage (on the GPU box): bash tools/profile_round.sh <tag>
# Produces gpurun_out/<tag>_bench.json, <tag>_kernel_stats.csv, <tag>_kernel_trace.csv, <tag>_pmc_*.csv
TAG=${1:-r01}
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R && python -c "import mgr_amd; from mgr_amd._build import source_hash; print(source_hash())" > gpurun_out/${TAG}_src_sha.txt   # the tree that is measured
cd $R && timeout 600 python bench.py > gpurun_out/${TAG}_bench.log 2>&1
tail -1 gpurun_out/${TAG}_bench.log > gpurun_out/${TAG}_bench.json
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/${TAG}_prof
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_prof -- python3 $R/bench.py --no-cpu --no-parity > $R/gpurun_out/${TAG}_prof.log 2>&1
cp $R/gpurun_out/${TAG}_prof/*/*_kernel_stats.csv $R/gpurun_out/${TAG}_kernel_stats.csv
cp $R/gpurun_out/${TAG}_prof/*/*_kernel_trace.csv $R/gpurun_out/${TAG}_kernel_trace.csv
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT"; do
  name=$(echo $pass | cut -d' ' -f1)
  rm -rf $R/gpurun_out/${TAG}_pmc_$name
  timeout 400 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $R/gpurun_out/${TAG}_pmc_$name -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu --no-parity > $R/gpurun_out/${TAG}_pmc_$name.log 2>&1
  cp $R/gpurun_out/${TAG}_pmc_$name/*/*_counter_collection.csv $R/gpurun_out/${TAG}_pmc_$name.csv
done
ls -la $R/gpurun_out | tail -20
cat $R/gpurun_out/${TAG}_bench.json | cut -c1-600
