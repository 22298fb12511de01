#!/bin/bash
# usage (on the GPU box, via gpurun): bash scripts/collect_profiles.sh r02
# Collects every artefact DESIGN.md section 5 cites into gpurun_out/<tag>_prof/ ; copy the summaries to profiles/ afterwards (scripts/collect_profiles.py).
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
tag=$1
o=gpurun_out/${tag}_prof
mkdir -p $o
# 1. kernel trace + stats of the bench command (7 steps incl. warm-up), joined with the launch log for the per-layer table
#    TFC_WGRAD_STREAM=0: every launch alone on the chip, so that a traced duration is kernel time (the product overlaps the weight gradients with the
#    input-gradient chain on a second stream; bench.py's instrumented steps serialise the same way). The un-profiled run (3.) uses the product default.
export TFC_WGRAD_STREAM=0
rm -f $o/launch.log
export TFC_LAUNCH_LOG=$PWD/$o/launch.log
rocprofv3 --kernel-trace --stats --output-format csv -d $o/trace -o p -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline > $o/bench_traced.json 2> $o/bench_traced.err
unset TFC_LAUNCH_LOG
python3 scripts/per_layer.py $o/trace $o/launch.log $o/per_layer.md > /dev/null
echo "trace done" 
# 2. HBM traffic: FETCH_SIZE and WRITE_SIZE in separate passes
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $o/pmc/fetch -o p -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > $o/pmc_fetch.json 2> $o/pmc_fetch.err
echo "fetch done"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $o/pmc/write -o p -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > $o/pmc_write.json 2> $o/pmc_write.err
echo "write done"
python3 scripts/pmc_traffic.py $o/pmc $o/pmc_traffic.json | head -8
# 3. the un-profiled default run
unset TFC_WGRAD_STREAM
python3 bench.py > $o/bench.json 2> $o/bench.err
cut -c1-300 $o/bench.json
# the raw counter CSVs are large: keep the summaries only
rm -rf $o/pmc/fetch $o/pmc/write
