"""Diagnostic (-DTFC_STAMP build): which workgroups of the persistent gather GEMM share a CU?"""
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
stamps = torch.zeros(1 << 20, dtype=torch.int64, device=DEV)
for rep in range(3):
    stamps.zero_()
    ops.check(lib.tfc_conv_fwd(ops.stream_ptr(), dt, ops.OP_CONV, x.ptr, x.pitch, N, H, H, Cin, Cout, ops._p(pk), y.ptr, y.pitch, None, None, ops._p(stamps), None, 0), "conv")
    torch.cuda.synchronize()
    s = stamps.cpu().numpy().reshape(-1, 8)[:512 * 4]
    hw = s[::4, 6]                      # wave 0 of each workgroup
    xcc = hw >> 32; lo = hw & 0xFFFFFFFF
    wave_id = lo & 0xF; simd = (lo >> 4) & 3; cu = (lo >> 8) & 0xF; sh = (lo >> 12) & 1; se = (lo >> 13) & 7
    key = xcc * 10000 + se * 1000 + sh * 100 + cu
    print("rep", rep, "distinct CUs:", len(np.unique(key)))
    for b in (0, 1, 2, 8, 9):
        mates = np.nonzero(key == key[b])[0]
        print(f"  wg {b}: xcc {xcc[b]} se {se[b]} sh {sh[b]} cu {cu[b]} simd {simd[b]} wave slot {wave_id[b]}  -> shares its CU with workgroups {mates.tolist()}")
    d = {}
    for b in range(512):
        d.setdefault(key[b], []).append(b)
    diffs = [v[1] - v[0] for v in d.values() if len(v) == 2]
    print("  pair index differences (count):", {int(k): int(c) for k, c in zip(*np.unique(diffs, return_counts=True))}, " CUs with != 2 workgroups:", sum(len(v) != 2 for v in d.values()))
