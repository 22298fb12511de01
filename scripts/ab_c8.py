"""first-layer convolution (6 -> 64 at 256 x 256, batch 32) with and without the sign words"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import tfc_gan_amd as T
from tfc_gan_amd import ops
DEV, dt, N = "cuda:0", ops.DT_BF16, 32
x = ops.View(torch.randn(N, 256, 256, 8, device=DEV).to(torch.bfloat16), 6)
w = torch.randn(64, 6, 4, 4, device=DEV) * 0.1
pk = ops.pack_weight(dt, ops.OP_CONV, 0, w, 6, 64)
bias = torch.randn(64, device=DEV) * 0.1
raw = ops.new_act(N, 255, 255, 64, dt, DEV)
mask = torch.empty(N, 255, 255, 8, dtype=torch.uint8, device=DEV)
def t(f):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 20 * 1e3
print("generic entry (no mask):", t(lambda: ops.conv_fwd(dt, ops.OP_CONV, x, 6, 64, pk, raw, bias=bias, flags=ops.EP_LEAKY)))
print("first_fwd, no mask     :", t(lambda: ops.conv_first_fwd(dt, x, 6, 64, pk, raw, bias=bias, flags=ops.EP_LEAKY)))
print("first_fwd, sign words  :", t(lambda: ops.conv_first_fwd(dt, x, 6, 64, pk, raw, bias=bias, flags=ops.EP_LEAKY, sign_mask=mask)))
out = ops.new_act(N, 128, 128, 64, dt, DEV)
print("blur-pool               :", t(lambda: ops.act_fwd(dt, raw, out, stats=None, slope=1.0, pool=2)))
osc = torch.tensor([0.7], device=DEV)
print("fused first block (D)   :", t(lambda: ops.first_block_fwd(dt, x, 6, 64, pk, out, bias=bias, oscale=osc, slope=0.2, sign_mask=mask)))
print("fused first block (D, no mask):", t(lambda: ops.first_block_fwd(dt, x, 6, 64, pk, out, bias=bias, oscale=osc, slope=0.2)))
print("fused first block (G)   :", t(lambda: ops.first_block_fwd(dt, x, 6, 64, pk, out, slope=0.2, act_after_rounding=True, sign_mask=mask)))
