# Usage (on the GPU box): bash tools/profile_round.sh <tag>
# The per-round evidence set, all under gpurun_out/ (copy what is to be judged into profiles/ with tools/summarize_profile.py):
#   <tag>_bench.json                     the default bench.py line (config F = BASELINE configs[2])
#   <tag>_kernel_stats.csv, _kernel_trace.csv   rocprofv3 --kernel-trace --stats of the same command
#   <tag>_pmc_*.csv                      separate --pmc passes (FETCH_SIZE | WRITE_SIZE | SQ counters), bench.py --steps 8 --warmup 3 (the fused schedule
#                                        engages once a step was announced two calls ahead: a 2-step run never shows the shipped kernels)
#   <tag>_cfg_<C>_bench.json, <tag>_cfg_<C>_kernel_stats.csv   the other BASELINE configurations (S, S_ref, A_ref, E): untruncated bench
#                                        line + rocprof kernel stats each
#   <tag>_cfg_<C>_b64_bench.json         the all-trainable configurations (E, S_ref, A_ref) once more at B = 64 (a batch that fills the chip)
#   <tag>_decode.json, <tag>_decode_kernel_stats.csv          BASELINE configs[4]: tools/decode_bench.py + its kernel stats (k_beam, k_frame_argmax)
#   <tag>_fit.txt, <tag>_dp2_host.json   fit_generator with host batches; bench.py --gpus 2 --comm host
TAG=${1:-r01}
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R && python -c "import mgr_amd; from mgr_amd._build import source_hash; print(source_hash())" > gpurun_out/${TAG}_src_sha.txt   # the tree that is measured
cd $R && timeout 900 python bench.py > gpurun_out/${TAG}_bench.log 2>&1
tail -1 gpurun_out/${TAG}_bench.log > gpurun_out/${TAG}_bench.json
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/${TAG}_prof
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_prof -- python3 $R/bench.py --no-cpu --no-parity --no-f32-leg > $R/gpurun_out/${TAG}_prof.log 2>&1
cp $R/gpurun_out/${TAG}_prof/*/*_kernel_stats.csv $R/gpurun_out/${TAG}_kernel_stats.csv
cp $R/gpurun_out/${TAG}_prof/*/*_kernel_trace.csv $R/gpurun_out/${TAG}_kernel_trace.csv
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT"; do
  name=$(echo $pass | cut -d' ' -f1)
  rm -rf $R/gpurun_out/${TAG}_pmc_$name
  timeout 400 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $R/gpurun_out/${TAG}_pmc_$name -- python3 $R/bench.py --steps 8 --warmup 3 --no-cpu --no-parity --no-f32-leg > $R/gpurun_out/${TAG}_pmc_$name.log 2>&1
  cp $R/gpurun_out/${TAG}_pmc_$name/*/*_counter_collection.csv $R/gpurun_out/${TAG}_pmc_$name.csv
done
if [ "$2" != "quick" ]; then
  for C in S S_ref A_ref E; do
    cd $R && timeout 300 python bench.py --config $C --steps 10 --no-cpu > gpurun_out/${TAG}_cfg_${C}_bench.log 2>&1
    tail -1 gpurun_out/${TAG}_cfg_${C}_bench.log > gpurun_out/${TAG}_cfg_${C}_bench.json
    cd /tmp; rm -rf $R/gpurun_out/${TAG}_cfgprof
    timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_cfgprof -- python3 $R/bench.py --config $C --steps 5 --warmup 2 --no-cpu --no-parity --no-f32-leg > /dev/null 2>&1
    cp $R/gpurun_out/${TAG}_cfgprof/*/*_kernel_stats.csv $R/gpurun_out/${TAG}_cfg_${C}_kernel_stats.csv
  done
  # the all-trainable configurations at a batch that fills the chip (B = 64: 4 batch groups x 2 directions of 32 / 19 clusters' workgroups)
  for C in E S_ref A_ref; do
    cd $R && timeout 300 python bench.py --config $C --batch 64 --steps 10 --no-cpu --no-parity 2> /dev/null | tail -1 > gpurun_out/${TAG}_cfg_${C}_b64_bench.json
  done
  cd $R && timeout 600 python tools/decode_bench.py > gpurun_out/${TAG}_decode.json 2> gpurun_out/${TAG}_decode.err
  cd /tmp; rm -rf $R/gpurun_out/${TAG}_cfgprof
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_cfgprof -- python3 $R/tools/decode_bench.py --cpu-n 1 --reps 1 > /dev/null 2>&1
  cp $R/gpurun_out/${TAG}_cfgprof/*/*_kernel_stats.csv $R/gpurun_out/${TAG}_decode_kernel_stats.csv
  rm -rf $R/gpurun_out/${TAG}_cfgprof
  cd $R && timeout 600 python tools/fit_bench.py > gpurun_out/${TAG}_fit.txt 2>&1
  cd $R && timeout 600 python bench.py --gpus 2 --comm host --no-cpu --no-parity --no-f32-leg 2> gpurun_out/${TAG}_dp2_host.err | tail -1 > gpurun_out/${TAG}_dp2_host.json
fi
ls $R/gpurun_out | grep "^${TAG}_" | tr '\n' ' '
cut -c1-400 $R/gpurun_out/${TAG}_bench.json
