#!/bin/bash
# In-situ variant of ablate.sh: runs the whole training step (values are garbage) under rocprofv3 with one ablated library and
# prints the per-launch times of the main implicit-GEMM variant. usage: bash scripts/ablate_step.sh "<ablation ids>"
cd "$(dirname "$0")/../tfc-gan_amd"
for a in $1; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-unused-value -DTFC_ABL=$a -Wl,-rpath,/opt/rocm/lib \
    -o /tmp/libabl$a.so csrc/api.hip csrc/igemm.hip csrc/elementwise.hip csrc/losses.hip csrc/probe.hip || exit 1
done
cd ..
R=$PWD
cd /tmp && export TMPDIR=/tmp
for a in 0 $1; do
  if [ "$a" != "0" ]; then export TFC_SO_OVERRIDE=/tmp/libabl$a.so; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/abl_step$a -o p -- python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline > $R/gpurun_out/abl_step$a.log 2>&1
  echo "== ablation $a"; python $R/scripts/kstats.py $R/gpurun_out/abl_step$a/p_kernel_stats.csv 5 8
done
