"""batched spectral-norm power iteration of the four discriminator blocks (3 launches): python scripts/ab_sn.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tfc_gan_amd import ops
DEV = "cuda:0"
shapes = [(64, 6 * 16), (128, 64 * 16), (256, 128 * 16), (512, 256 * 16)]
Ws = [torch.randn(r, k // 16, 4, 4, device=DEV) * 0.05 for r, k in shapes]
us = [torch.nn.functional.normalize(torch.randn(r, device=DEV), dim=0) for r, k in shapes]
vs = [torch.nn.functional.normalize(torch.randn(k, device=DEV), dim=0) for r, k in shapes]
sig = [torch.zeros(2, device=DEV) for _ in shapes]
ws = None
def run():
    global ws
    ws = ops.spectral_norm_step_batched(Ws, us, vs, sig, power_iter=True, u_snaps=None, v_snaps=None, ws=ws)
for _ in range(5):
    run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50):
    run()
e1.record()
torch.cuda.synchronize()
print(f"batched power iteration: {e0.elapsed_time(e1) / 50 * 1e3:.1f} us per call, sigma {[round(float(s[0]), 6) for s in sig]}")
