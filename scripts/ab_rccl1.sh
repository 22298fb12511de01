#!/bin/bash
# one-GPU rehearsal of the RCCL path (a ONE-rank "nccl" group, TFC_FORCE_COLLECTIVES=1): what do the collectives cost when they move nothing?
# usage (GPU box): bash scripts/ab_rccl1.sh > gpurun_out/ab_rccl1.log
run() {
  echo -n "$1: "
  env $2 timeout -k 10 200 python3 bench.py --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | tail -1 | python3 -c "import sys,json; b=json.loads(sys.stdin.read()); print(round(b['ms_per_step'],3), 'ms/step, exposed', round(b['exposed_allreduce_ms'],3), b['allreduce_backend'])"
}
run "plain, two streams          " "A=1"
run "rccl1, two streams          " "TFC_FORCE_COLLECTIVES=1"
run "plain, one stream           " "TFC_WGRAD_STREAM=0"
run "rccl1, one stream           " "TFC_FORCE_COLLECTIVES=1 TFC_WGRAD_STREAM=0"
run "rccl1, two streams, no record_stream" "TFC_FORCE_COLLECTIVES=1 TORCH_NCCL_AVOID_RECORD_STREAMS=1"
run "rccl1, two streams, one bucket per net" "TFC_FORCE_COLLECTIVES=1 TFC_BUCKET_MB=512"
run "rccl1, two streams, 64 MiB buckets" "TFC_FORCE_COLLECTIVES=1 TFC_BUCKET_MB=64"
run "plain, two streams (again)  " "A=1"
