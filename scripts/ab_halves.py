"""A/B of the persistent gather GEMM's two forms at the PATCH-16 layer shapes (batch 32): two independent 256-thread workgroups per CU (cfg 15 / forced tile)
against one 512-thread workgroup whose halves run (stages + 1) / 2 barrier slots apart (cfg | 32). usage: python scripts/ab_halves.py [shift ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import tfc_gan_amd as T
from tfc_gan_amd import ops
DEV, dt, N = "cuda:0", ops.DT_BF16, 32
lib = T._lib.load()
def rnd(*s):
    return ops.View(torch.randn(*s, device=DEV).to(torch.bfloat16), s[-1])
# (op, pass, H (forward input), Cin, Cout)
CASES = [(ops.OP_CONV, 0, 128, 64, 128), (ops.OP_CONV, 1, 128, 64, 128), (ops.OP_CONV, 0, 64, 128, 256), (ops.OP_CONV, 1, 64, 128, 256),
         (ops.OP_CONV, 0, 32, 256, 512), (ops.OP_CONV, 1, 32, 256, 512), (ops.OP_CONVT, 0, 64, 256, 64), (ops.OP_CONVT, 1, 64, 256, 64),
         (ops.OP_CONVT, 0, 32, 512, 128), (ops.OP_CONVT, 1, 32, 512, 128), (ops.OP_CONV, 0, 16, 512, 512), (ops.OP_CONVT, 0, 16, 1024, 256)]
def run(op, pas, H, Cin, Cout, cfg):
    lib.tfc_debug_set_igemm_config(cfg)
    oh = ops.OUT_HW[op](H)
    w = torch.randn((Cin, Cout, 4, 4) if op == ops.OP_CONVT else (Cout, Cin, 4, 4), device=DEV) * 0.03
    pk = ops.pack_weight(dt, op, pas, w, Cin, Cout)
    if pas == 0:
        x, y = rnd(N, H, H, Cin), ops.new_act(N, oh, oh, Cout, dt, DEV)
        f = lambda: ops.conv_fwd(dt, op, x, Cin, Cout, pk, y)
    else:
        dy, dx = rnd(N, oh, oh, Cout), ops.new_act(N, H, H, Cin, dt, DEV)
        f = lambda: ops.conv_dgrad(dt, op, dy, N, H, H, Cin, Cout, pk, dx)
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(20):
        f()
    ev[1].record()
    torch.cuda.synchronize()
    return ev[0].elapsed_time(ev[1]) / 20 * 1e3
for op, pas, H, Cin, Cout in CASES:
    taps = 4 if op == ops.OP_CONVT else 16
    oh = ops.OUT_HW[op](H)
    fl = 2.0 * N * oh * oh * Cin * Cout * taps
    a, b = run(op, pas, H, Cin, Cout, 15), run(op, pas, H, Cin, Cout, 15 | 32)
    print(f"op {op} pass {pas} H={H:3d} {Cin:4d}->{Cout:3d}: two workgroups {a:7.1f} us ({fl / a / 1e6:6.0f} TF)   two halves {b:7.1f} us ({fl / b / 1e6:6.0f} TF)   x{a / b:.3f}", flush=True)
lib.tfc_debug_set_igemm_config(-1)
