"""A/B of the fused first-block backward: VALU transposed blur (TFC_FIRST_BWD_VALU=1) vs the blur as a GEMM on the matrix core, batch 32."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import tfc_gan_amd as T
from tfc_gan_amd import ops
DEV, dt, N = "cuda:0", ops.DT_BF16, 32
def rnd(*s):
    return ops.View(torch.randn(*s, device=DEV).to(torch.bfloat16), s[-1])
for Cin in (6, 3):
    x = ops.View(torch.randn(N, 256, 256, 8, device=DEV).to(torch.bfloat16), Cin)
    y, g = rnd(N, 255, 255, 64), rnd(N, 128, 128, 64)
    dw = torch.zeros(64, Cin, 4, 4, device=DEV)
    bs = torch.zeros(N, 64, device=DEV)
    ws = None
    mask = (torch.rand(N, 255, 255, 8, device=DEV) * 255).to(torch.uint8)
    for valu in ("1", "0", "mask"):
        os.environ["TFC_FIRST_BWD_VALU"] = "1" if valu == "1" else "0"
        kw = dict(sign_mask=mask) if valu == "mask" else {}
        for _ in range(3):
            ws = ops.first_block_bwd_wgrad(dt, x, y, g, Cin, 64, dw, slope=0.2, ws=ws, bias_sums=bs, **kw)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            ws = ops.first_block_bwd_wgrad(dt, x, y, g, Cin, 64, dw, slope=0.2, ws=ws, bias_sums=bs, **kw)
        e1.record()
        torch.cuda.synchronize()
        print(f"Cin={Cin} {dict([('1', 'VALU blur'), ('0', 'MFMA blur'), ('mask', 'MFMA blur + sign mask')])[valu]}: {e0.elapsed_time(e1) / 20 * 1e3:7.1f} us per call (kernel + slab reduce + finish + bias reduce)", flush=True)
