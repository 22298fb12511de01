"""Time the LPIPS term alone (VGG16 forward x2 + backward w.r.t. the first image) at the bench batch: python scripts/lpips_bench.py [N] [iters]"""
import os
import sys
import time
import warnings

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import tfc_gan_amd as T  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 32
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 5
dev = torch.device("cuda:0")
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    m = T.LPIPS().to(dev)
A, B = T.synthetic_pairs(N, seed=1)
A, B = A.to(dev), B.to(dev)
for _ in range(2):
    m.value_and_grad(A, B)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(iters):
    v, g = m.value_and_grad(A, B)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / iters
gflop = 3 * 2 * sum(9 * ci * co * (256 // s) ** 2 for ci, co, s in
                    [(3, 64, 1), (64, 64, 1), (64, 128, 2), (128, 128, 2), (128, 256, 4), (256, 256, 4), (256, 256, 4), (256, 512, 8), (512, 512, 8),
                     (512, 512, 8), (512, 512, 16), (512, 512, 16), (512, 512, 16)]) / 1e9
print(f"LPIPS value+grad N={N}: {dt * 1e3:.2f} ms  ({N * gflop / dt / 1e3:.0f} TFLOP/s over {gflop:.1f} GFLOP/image), value {v.item():.4f}")
