"""Diagnostic: phase shares of tfc_igemm_kernel from s_memtime stamps (needs a -DTFC_STAMP build of the library: TFC_SO_OVERRIDE).
Never quote this build's run time -- read the SHARES (cdna_hip_programming.md section 7, in-kernel stamps)."""
import ctypes, os, sys
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
    stamps = torch.zeros(1 << 22, dtype=torch.int64, device=DEV)
    for _ in range(3):
        ops.check(lib.tfc_conv_fwd(ops.stream_ptr(), dt, ops.OP_CONV, x.ptr, x.pitch, N, H, H, Cin, Cout, ops._p(pk), y.ptr, y.pitch, None, None,
                                   ops._p(stamps), None, 0), "conv")
    torch.cuda.synchronize()
    s = stamps.cpu().numpy().reshape(-1, 8)
    s = s[s[:, 0] != 0]
    t = s[:, :7].astype(np.int64)
    d = np.diff(t, axis=1)                         # prologue, main loop, epilogue VALU+LDS writes, barrier, stores issue, store drain
    life = t[:, 6] - t[:, 0]
    names = ["prologue", "mainloop", "ep:acc->lds", "ep:barrier", "ep:stores", "ep:drain"]
    print(f"H={H} {Cin}->{Cout}: {len(s)} waves, lifetime mean {life.mean():.0f} cyc(100MHz ticks?)  kernel span {(t[:,6].max()-t[:,0].min())}")
    for i, n in enumerate(names):
        print(f"   {n:12s} mean {d[:, i].mean():9.0f}  share {d[:, i].sum() / life.sum():6.1%}  p10 {np.percentile(d[:, i], 10):8.0f} p90 {np.percentile(d[:, i], 90):8.0f}")
    # how synchronised are the workgroups of one CU? spread of main-loop END times among co-resident waves of the first round
    hw = s[:, 7] & 0xFFFFFFFF
    xcc = s[:, 7] >> 32
    cu = ((hw >> 8) & 0xF) | (((hw >> 12) & 0x7) << 4) | (((hw >> 16) & 0x7) << 7) | (xcc << 10)     # CU_ID | SH_ID | SE_ID | XCC
    first = t[:, 0] < np.percentile(t[:, 0], 15)
    sp = []
    for c in np.unique(cu[first]):
        m = first & (cu == c)
        if m.sum() >= 8:
            sp.append(t[m, 2].max() - t[m, 2].min())
    if sp:
        print(f"   first-round spread of main-loop END among co-resident waves of a CU: mean {np.mean(sp):.0f} cycles over {len(sp)} CUs")
