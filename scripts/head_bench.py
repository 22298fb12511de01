"""time the PatchGAN head forward (512 -> 1 at 16 x 16, batch 32): python scripts/head_bench.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import tfc_gan_amd as T  # noqa: E402
from tfc_gan_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
dt = ops.DT_BF16
x = ops.View(torch.randn((32, 16, 16, 512), device=dev).to(torch.bfloat16), 512)
w = torch.randn((1, 512, 4, 4), device=dev) * 0.02
y = ops.new_act(32, 16, 16, 8, dt, dev, zero=True)
for _ in range(5):
    ops.patchgan_head_fwd(dt, x, w, ops.View(y.t, 1, 0))
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(50):
    ops.patchgan_head_fwd(dt, x, w, ops.View(y.t, 1, 0))
b.record()
torch.cuda.synchronize()
print(f"patchgan head fwd: {a.elapsed_time(b) / 50 * 1e3:.1f} us", flush=True)
