"""Time the STN21 step (config C5 on one GPU, module-level path): python scripts/stn21_bench.py [batch] [steps]"""
import os
import sys
import time
import warnings

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import tfc_gan_amd as T  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 32
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
dev = torch.device("cuda:0")
torch.manual_seed(1)
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    crit = T.LPIPS().to(dev)
st = T.STN21Step((3, 256, 256), lpips=crit, device=dev)
A, B = T.synthetic_pairs(N, seed=3)
A, B = A.to(dev), B.to(dev)
for _ in range(2):
    out = st.step(A, B)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    out = st.step(A, B)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
print(f"STN21 step, batch {N}: {dt * 1e3:.1f} ms/step  {N / dt:.0f} images/s  " + " ".join(f"{k}={float(v):.4f}" for k, v in out.items() if v.dim() == 0), flush=True)
