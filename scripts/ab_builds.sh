#!/bin/bash
# In-situ A/B of compile-time knobs: builds variant libraries on the GPU box and times the full training step with each (same process
# layout, same box).  usage: bash scripts/ab_builds.sh "<name>:<hipcc -D flags>" ...
cd "$(dirname "$0")/.."
cat > /tmp/ab_step.py <<'PY'
import sys, time, torch, os
sys.path.insert(0, os.getcwd())
import tfc_gan_amd as T
dev = torch.device('cuda', 0)
T.set_compute_dtype(torch.bfloat16); torch.manual_seed(42)
G = T.GeneratorUNet((3,256,256)).to(dev); D = T.Discriminator1((3,256,256)).to(dev)
G.apply(T.weights_init_normal); D.apply(T.weights_init_normal)
ts = T.TrainStep(G, D, compute_dtype=torch.bfloat16)
A, B = T.synthetic_pairs(32, seed=1234); A, B = A.to(dev), B.to(dev)
for _ in range(8): ts.step(A, B)
best = 1e9
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): ts.step(A, B)
    torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) / 20)
print(f"{sys.argv[1]}: {best*1e3:.3f} ms/step  {32/best:.1f} img/s")
PY
python /tmp/ab_step.py baseline
for spec in "$@"; do
  name="${spec%%:*}"; flags="${spec#*:}"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-unused-value $flags -Wl,-rpath,/opt/rocm/lib \
    -o /tmp/lib_$name.so tfc-gan_amd/csrc/api.hip tfc-gan_amd/csrc/igemm.hip tfc-gan_amd/csrc/elementwise.hip tfc-gan_amd/csrc/losses.hip tfc-gan_amd/csrc/probe.hip || exit 1
  TFC_SO_OVERRIDE=/tmp/lib_$name.so python /tmp/ab_step.py "$name"
done
python /tmp/ab_step.py baseline
