#!/bin/bash
# Diagnostic builds of the library with parts of the implicit-GEMM main loop removed (results are WRONG by construction; timing only):
#   8 = no global stores in the epilogue, 9 = no LDS staging writes in the epilogue
#   6 = no main loop (prologue + epilogue only), 7 = full main loop, no epilogue
#   1 = no weight-stream loads in the loop, 2 = no A-fragment LDS reads, 3 = both, 4 = 3 + no halo staging, 5 = 4 + no barrier
# usage (on the GPU box): bash scripts/ablate.sh
cd "$(dirname "$0")/../tfc-gan_amd"
for a in 10; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-unused-value -DTFC_ABL=$a -Wl,-rpath,/opt/rocm/lib \
    -o /tmp/libabl$a.so csrc/api.hip csrc/igemm.hip csrc/elementwise.hip csrc/losses.hip csrc/probe.hip || exit 1
done
cd ..
echo "== baseline"; python scripts/microbench.py cfg 2>&1 | grep "cfg 0"
for a in 10; do echo "== ablation $a"; TFC_SO_OVERRIDE=/tmp/libabl$a.so python scripts/microbench.py cfg 2>&1 | grep "cfg 0"; done
