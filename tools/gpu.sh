#!/bin/bash
# Build libt3d.so for the current sources, then hand the command to gpurun (a stale library is refused on the box: abi.load).
#   tools/gpu.sh <timeout-seconds> '<command run on the GPU box from the repo root>'
set -e
cd "$(dirname "$0")/.."
python transferable3d_amd/build.py > /tmp/t3d_build.log 2>&1 || { tail -30 /tmp/t3d_build.log; exit 1; }
python -c "from transferable3d_amd import build as B; assert B.embedded_hash() == B.lib_source_hash(), 'stale library'"
t=$1; shift
exec /usr/local/graft/bin/gpurun --timeout "$t" -- "$@"
