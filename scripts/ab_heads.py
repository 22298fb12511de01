"""the single-call 3-channel kernels of the step (batch 32): D block-1 input gradient (rows packed), generator head forward / input gradient / weight gradient"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tfc_gan_amd import ops
DEV, dt, N = "cuda:0", ops.DT_BF16, 32
def t(f, n=20):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
dy = ops.View(torch.randn(N, 255, 255, 64, device=DEV).to(torch.bfloat16), 64)
w = torch.randn(64, 6, 4, 4, device=DEV) * 0.1
osc = torch.tensor([0.7], device=DEV)
print(f"conv_dgrad_image (tfc_dgrad_rows4_kernel): {t(lambda: ops.conv_dgrad_image(dt, dy, N, 256, 256, 6, w, osc, 3)):7.1f} us")
x = ops.View(torch.randn(N, 128, 128, 128, device=DEV).to(torch.bfloat16), 128)
wh = torch.randn(3, 128, 4, 4, device=DEV) * 0.05
bh = torch.zeros(3, device=DEV)
out = torch.empty(N, 3, 256, 256, device=DEV)
print(f"upconv_head_fwd:                           {t(lambda: ops.upconv_head_fwd(dt, x, wh, bh, out)):7.1f} us")
dyh = ops.View(torch.randn(N, 256, 256, 8, device=DEV).to(torch.bfloat16), 3)
dx = ops.new_act(N, 128, 128, 128, dt, DEV)
print(f"upconv_head_dgrad:                         {t(lambda: ops.upconv_head_dgrad(dt, dyh, N, 128, 128, wh, dx)):7.1f} us")
