"""transposed-convolution weight gradients of the generator's up path (batch 32): run once plain and once with TFC_WGRADT_NARROW=1 (the 32 x 32 workgroup tile)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tfc_gan_amd import ops
DEV, dt, N = "cuda:0", ops.DT_BF16, 32
ws = None
tot = 0.0
for H, cin, cout in ((4, 512, 512), (8, 1024, 512), (16, 1024, 256), (32, 512, 128), (64, 256, 64)):
    x = ops.View(torch.randn(N, H, H, cin, device=DEV).to(torch.bfloat16), cin)
    dy = ops.View(torch.randn(N, 2 * H, 2 * H, cout, device=DEV).to(torch.bfloat16), cout)
    dw = torch.zeros(cin, cout, 4, 4, device=DEV)
    for _ in range(3):
        ws = ops.conv_wgrad(dt, ops.OP_CONVT, x, dy, cin, cout, dw, False, ws)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        ws = ops.conv_wgrad(dt, ops.OP_CONVT, x, dy, cin, cout, dw, False, ws)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    tot += us
    gf = 2.0 * N * H * H * cin * cout * 16 / 1e6                    # MFLOP: MFLOP / us = TFLOP/s
    print(f"convT wgrad {cin:5d} -> {cout:4d} @ {H:2d}x{H:<2d}: {us:7.1f} us  ({gf / us / 1e3 * 1e3:6.0f} TFLOP/s)   checksum {dw.double().abs().sum().item():.6e}")
print(f"sum {tot:.1f} us ({'narrow 32x32' if os.environ.get('TFC_WGRADT_NARROW') else 'wide 32x64'})")
