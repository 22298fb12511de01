"""BlurPool(stride 2) backward / forward passes at the sizes of the step (batch 32, bf16): python scripts/ab_pool2.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tfc_gan_amd import ops
DEV, dt, N = "cuda:0", ops.DT_BF16, 32
torch.manual_seed(0)
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
tb = tf = 0.0
for H, C in ((127, 128), (63, 256), (31, 512), (15, 512)):
    Po = (H - 1) // 2 + 1
    x = ops.View(torch.randn(N, H, H, C, device=DEV).to(torch.bfloat16), C)
    g = ops.View(torch.randn(N, Po, Po, C, device=DEV).to(torch.bfloat16), C)
    dx = ops.new_act(N, H, H, C, dt, DEV)
    y = ops.new_act(N, Po, Po, C, dt, DEV)
    st = torch.rand(N, C, 2, device=DEV) * H * H
    st[..., 1] += st[..., 0] ** 2 / (H * H)
    rs = torch.zeros(N, C, 2, device=DEV)
    b0 = t(lambda: ops.act_bwd(dt, 0, g, x, N, H, H, C, dx, stats=None, slope=0.2, pool=2))
    b1 = t(lambda: ops.act_bwd(dt, 1, g, x, N, H, H, C, None, stats=st, slope=0.2, pool=2, rstats=rs))
    b2 = t(lambda: ops.act_bwd(dt, 2, g, x, N, H, H, C, dx, stats=st, slope=0.2, pool=2, rstats=rs))
    f0 = t(lambda: ops.act_fwd(dt, x, y, stats=None, slope=1.0, pool=2))
    f1 = t(lambda: ops.act_fwd(dt, x, y, stats=st, slope=0.2, pool=2))
    mb = N * H * H * C * 2 / 1e6
    ops.act_bwd(dt, 0, g, x, N, H, H, C, dx, stats=None, slope=0.2, pool=2)
    c0 = dx.t.double().sum().item()
    rs.zero_(); ops.act_bwd(dt, 1, g, x, N, H, H, C, None, stats=st, slope=0.2, pool=2, rstats=rs); c1 = rs.double().sum().item()
    ops.act_bwd(dt, 2, g, x, N, H, H, C, dx, stats=st, slope=0.2, pool=2, rstats=rs); c2 = dx.t.double().abs().sum().item()
    print(f"       checksums (bit-identity across builds): {c0!r} {c1!r} {c2!r}")
    print(f"{H:4d}^2 x {C:3d} ({mb:6.1f} MB tensor): bwd plain {b0:6.1f}  reduce {b1:6.1f}  apply {b2:6.1f} us | fwd blur {f0:6.1f}  norm+act+blur {f1:6.1f} us")
    tb += b0 + b1 + b2
    tf += f0 + f1
print(f"sum backward {tb:.1f} us, forward {tf:.1f} us")
tu = 0.0
for H, C in ((128, 64), (64, 128), (32, 256), (16, 512), (8, 512)):       # stride-1 blur of the up path: forward (+ InstanceNorm sums) and transpose
    x = ops.View(torch.randn(N, H, H, C, device=DEV).to(torch.bfloat16), C)
    y = ops.new_act(N, H, H, C, dt, DEV)
    so = torch.zeros(N, C, 2, device=DEV)
    f = t(lambda: ops.act_fwd(dt, x, y, stats=None, slope=1.0, pool=1, stats_out=so))
    b = t(lambda: ops.act_bwd(dt, 0, x, None, N, H, H, C, y, stats=None, slope=1.0, pool=1))
    print(f"{H:4d}^2 x {C:3d}: blur1 forward + sums {f:6.1f} us, transpose {b:6.1f} us")
    tu += f + b
print(f"sum blur1 {tu:.1f} us")
