"""A/B of the gather-GEMM epilogue variants at the shape of down2 / D block 2 (64 -> 128 @ 128x128, batch 32)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import tfc_gan_amd as T
from tfc_gan_amd import ops
DEV = "cuda:0"; dt = ops.DT_BF16; N = 32
H, Cin, Cout = 128, 64, 128
x = ops.View(torch.randn(N, H, H, Cin, device=DEV).to(torch.bfloat16), Cin)
w = torch.randn(Cout, Cin, 4, 4, device=DEV) * 0.03
y = ops.new_act(N, H - 1, H - 1, Cout, dt, DEV)
pk = ops.pack_weight(dt, ops.OP_CONV, 0, w, Cin, Cout)
bias = torch.randn(Cout, device=DEV); osc = torch.tensor([0.5], device=DEV)
stats = torch.zeros(N, Cout, 2, device=DEV)
def t(fn, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e6
cases = {"plain": dict(), "bias": dict(bias=bias), "bias+leaky+oscale": dict(bias=bias, oscale=osc, flags=ops.EP_LEAKY), "oscale": dict(oscale=osc),
         "leaky": dict(flags=ops.EP_LEAKY), "stats": dict(stats=stats)}
for name, kw in cases.items():
    print(f"{name:20s} {t(lambda: ops.conv_fwd(dt, ops.OP_CONV, x, Cin, Cout, pk, y, **kw)):8.1f} us")
