# Usage (on the GPU box): bash tools/ab_bench.sh <tag> "<bench args A>" "<bench args B>" ...
# A/B of bench.py variants inside the pipelined step: each variant's JSON line + rocprofv3 kernel stats / trace under gpurun_out/.
TAG=$1; shift
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
i=0
for ARGS in "$@"; do
  cd $R && timeout 300 python bench.py --steps 30 --no-cpu --no-parity --no-f32-leg $ARGS 2> gpurun_out/${TAG}_v${i}.err | tail -1 > gpurun_out/${TAG}_v${i}.json
  cd /tmp && export TMPDIR=/tmp
  rm -rf $R/gpurun_out/${TAG}_prof
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_prof -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu --no-parity --no-f32-leg $ARGS > /dev/null 2>&1
  cp $R/gpurun_out/${TAG}_prof/*/*_kernel_stats.csv $R/gpurun_out/${TAG}_v${i}_kernel_stats.csv
  cp $R/gpurun_out/${TAG}_prof/*/*_kernel_trace.csv $R/gpurun_out/${TAG}_v${i}_kernel_trace.csv
  rm -rf $R/gpurun_out/${TAG}_prof
  echo "== v$i: $ARGS"; python3 -c "
import json,sys
d=json.load(open('$R/gpurun_out/${TAG}_v${i}.json')); print(d['ms_per_step'], {k:round(v['ms']/max(v['launches'],1),3) for k,v in d['kernel_ms'].items()})"
  head -8 $R/gpurun_out/${TAG}_v${i}_kernel_stats.csv | cut -c1-150
  i=$((i+1))
done
