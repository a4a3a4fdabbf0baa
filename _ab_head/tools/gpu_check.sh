# Usage (on the GPU box): bash tools/gpu_check.sh <tag>  - the whole -m gpu suite, then the default bench line
TAG=${1:-chk}
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/${TAG}_gputest.txt 2>&1
tail -8 gpurun_out/${TAG}_gputest.txt
timeout 600 python bench.py > gpurun_out/${TAG}_bench.log 2>&1
tail -1 gpurun_out/${TAG}_bench.log > gpurun_out/${TAG}_bench.json
python - <<PY
import json
d = json.load(open("gpurun_out/${TAG}_bench.json"))
print({k: d[k] for k in ("value", "ms_per_step", "roofline", "ctc_loss_parity", "persistent_launches")})
print(d["kernel_ms"])
PY
