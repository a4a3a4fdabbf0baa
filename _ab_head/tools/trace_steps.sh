# Usage (on the GPU box): bash tools/trace_steps.sh <tag> <steps> [bench args]   -> gpurun_out/<tag>_kernel_trace.csv
TAG=$1; STEPS=$2; shift; shift
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/${TAG}_prof
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/${TAG}_prof -- python3 $R/bench.py --steps $STEPS --warmup 3 --no-cpu --no-parity --no-f32-leg "$@" > /dev/null 2>&1
cp $R/gpurun_out/${TAG}_prof/*/*_kernel_trace.csv $R/gpurun_out/${TAG}_kernel_trace.csv
rm -rf $R/gpurun_out/${TAG}_prof
