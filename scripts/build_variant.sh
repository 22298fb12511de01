#!/bin/bash
# Build a variant of the library with extra hipcc flags into diag/lib_<name>.so (git-ignored, but it travels to the GPU box with the snapshot):
#   bash scripts/build_variant.sh nt1_16 -DTFC_BD2_NT1=16
#   gpurun -- 'TFC_SO_OVERRIDE=$PWD/diag/lib_nt1_16.so python scripts/ab_step.py nt1_16'
# Cross-compiles here (no GPU needed), so an A/B costs GPU time only for the timing itself.
name=$1; shift
cd "$(dirname "$0")/.." && mkdir -p diag
srcs=$(python -c "
import sys; sys.path.insert(0,'.')
from tfc_gan_amd import _lib
import os
print(' '.join(os.path.join(_lib.CSRC, f) for f in _lib.SOURCES))")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-unused-value "$@" -Wl,-rpath,/opt/rocm/lib -o diag/lib_$name.so $srcs
