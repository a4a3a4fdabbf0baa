#!/bin/bash
# Build libmgr.so HERE (the GPU box receives the built .so with the snapshot), then run the given command on an MI355X box.
#   tools/gpu.sh [--timeout S] -- '<command>'
set -e
cd "$(dirname "$0")/.."
python __graft_entry__.py > /tmp/mgr_build.log 2>&1 || { tail -30 /tmp/mgr_build.log; exit 1; }
exec /usr/local/graft/bin/gpurun "$@"
