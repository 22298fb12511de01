#!/bin/bash
# usage: bash scripts/per_layer.sh <tag> [env assignments...]   -> gpurun_out/<tag>/ + profiles/<tag>_per_layer.md
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
tag=$1; shift
mkdir -p gpurun_out/$tag
rm -f gpurun_out/$tag/launch.log
export TFC_LAUNCH_LOG=$PWD/gpurun_out/$tag/launch.log
for kv in "$@"; do export "$kv"; done
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$tag -o p -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline > gpurun_out/$tag/bench.json 2> gpurun_out/$tag/bench.err
python3 scripts/per_layer.py gpurun_out/$tag gpurun_out/$tag/launch.log gpurun_out/$tag/per_layer.md > /dev/null
tail -4 gpurun_out/$tag/per_layer.md
