"""Time the full training step (batch 32, bf16) with the library TFC_SO_OVERRIDE names (or the in-tree one): best of 3 x 20 steps.
usage: [TFC_SO_OVERRIDE=diag/lib_x.so] python scripts/ab_step.py <label>"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import tfc_gan_amd as T  # noqa: E402

dev = torch.device("cuda", 0)
T.set_compute_dtype(torch.bfloat16)
torch.manual_seed(42)
G = T.GeneratorUNet((3, 256, 256)).to(dev)
D = T.Discriminator1((3, 256, 256)).to(dev)
G.apply(T.weights_init_normal)
D.apply(T.weights_init_normal)
ts = T.TrainStep(G, D, compute_dtype=torch.bfloat16)
A, B = T.synthetic_pairs(32, seed=1234)
A, B = A.to(dev), B.to(dev)
for _ in range(8):
    ts.step(A, B)
best = 1e9
for rep in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        ts.step(A, B)
    torch.cuda.synchronize()
    best = min(best, (time.perf_counter() - t0) / 20)
print(f"{sys.argv[1] if len(sys.argv) > 1 else 'in-tree'}: {best * 1e3:.3f} ms/step  {32 / best:.1f} img/s", flush=True)
