"""Diagnostic: phase shares of the persistent gather GEMM (tfc_igemm2_kernel) from s_memtime stamps; needs a -DTFC_STAMP build (TFC_SO_OVERRIDE).
Read the SHARES, never the run time of this build."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import tfc_gan_amd as T
from tfc_gan_amd import ops, _lib
DEV = "cuda:0"
dt = ops.DT_BF16
N = 32
lib = _lib.load()
shapes = [(128, 64, 128), (64, 128, 256), (32, 256, 512), (16, 512, 512), (8, 512, 512)]
for H, Cin, Cout in shapes:
    x = ops.View(torch.randn(N, H, H, Cin, device=DEV).to(torch.bfloat16), Cin)
    w = torch.randn(Cout, Cin, 4, 4, device=DEV) * 0.03
    y = ops.new_act(N, H - 1, H - 1, Cout, dt, DEV)
    pk = ops.pack_weight(dt, ops.OP_CONV, 0, w, Cin, Cout)
    stamps = torch.zeros(1 << 20, dtype=torch.int64, device=DEV)
    for _ in range(3):
        ops.check(lib.tfc_conv_fwd(ops.stream_ptr(), dt, ops.OP_CONV, x.ptr, x.pitch, N, H, H, Cin, Cout, ops._p(pk), y.ptr, y.pitch, None, None,
                                   ops._p(stamps), None, 0, None), "conv")
    torch.cuda.synchronize()
    s = stamps.cpu().numpy().reshape(-1, 16)
    s = s[s[:, 0] != 0].astype(np.int64)
    life = s[:, 0]
    pro = s[:, 1] * 0
    print(f"   shader clock during the kernel: {100e6 * (s[:, 0] / s[:, 6]).mean() / 1e9:.3f} GHz")
    print(f"H={H} {Cin}->{Cout}: {len(s)} waves, tiles/wave {s[:, 7].mean():.2f}, lifetime mean {life.mean():.0f} cycles, ")
    print(f"   prologue(first tile) {pro.mean():8.0f}  {pro.sum() / life.sum():6.1%}")
    for i, n in ((2, "mainloop"), (1, " of which stage sync"), (8, " of which halo issue"), (9, " of which first A read"), (3, "ep:acc->lds"), (4, "ep:barrier"), (5, "ep:stores")):
        print(f"   {n:20s} {s[:, i].mean():8.0f}  {s[:, i].sum() / life.sum():6.1%}   per tile {(s[:, i] / s[:, 7]).mean():8.0f}")
