"""Diagnostic (-DTFC_STAMP build): phase shares of the persistent gather GEMM per epilogue variant at the shape of down2."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import tfc_gan_amd as T
from tfc_gan_amd import ops, _lib
DEV = "cuda:0"; dt = ops.DT_BF16; N = 32
lib = _lib.load()
H, Cin, Cout = 128, 64, 128
x = ops.View(torch.randn(N, H, H, Cin, device=DEV).to(torch.bfloat16), Cin)
w = torch.randn(Cout, Cin, 4, 4, device=DEV) * 0.03
y = ops.new_act(N, H - 1, H - 1, Cout, dt, DEV)
pk = ops.pack_weight(dt, ops.OP_CONV, 0, w, Cin, Cout)
bias = torch.randn(Cout, device=DEV); osc = torch.tensor([0.5], device=DEV); stats = torch.zeros(N, Cout, 2, device=DEV)
for name, b, s_, o, fl in (("plain", None, None, None, 0), ("bias", bias, None, None, 1), ("stats", None, stats, None, 2), ("bias+leaky+osc", bias, None, osc, 1 | 16)):
    stamps = torch.zeros(1 << 20, dtype=torch.int64, device=DEV)
    for _ in range(3):
        ops.check(lib.tfc_conv_fwd(ops.stream_ptr(), dt, ops.OP_CONV, x.ptr, x.pitch, N, H, H, Cin, Cout, ops._p(pk), y.ptr, y.pitch, ops._p(b), ops._p(s_),
                                   ops._p(stamps), ops._p(o), fl), "conv")
    torch.cuda.synchronize()
    s = stamps.cpu().numpy().reshape(-1, 16)
    s = s[s[:, 0] != 0].astype(np.int64)
    print(f"{name:16s} lifetime {s[:, 0].mean():8.0f} | per tile: mainloop {(s[:, 2] / s[:, 7]).mean():7.0f} (sync {(s[:, 1] / s[:, 7]).mean():5.0f} halo {(s[:, 8] / s[:, 7]).mean():5.0f}) "
          f"ep-reg {(s[:, 3] / s[:, 7]).mean():6.0f} ep-bar {(s[:, 4] / s[:, 7]).mean():6.0f} ep-store {(s[:, 5] / s[:, 7]).mean():6.0f}")
