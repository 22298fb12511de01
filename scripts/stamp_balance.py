"""Diagnostic (-DTFC_STAMP build): distribution of workgroup run times of the persistent gather GEMM (load balance across CUs / XCDs)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import tfc_gan_amd as T
from tfc_gan_amd import ops, _lib
DEV = "cuda:0"; dt = ops.DT_BF16; N = 32
lib = _lib.load()
for H, Cin, Cout in [(128, 64, 128), (64, 128, 256), (32, 256, 512)]:
    x = ops.View(torch.randn(N, H, H, Cin, device=DEV).to(torch.bfloat16), Cin)
    w = torch.randn(Cout, Cin, 4, 4, device=DEV) * 0.03
    y = ops.new_act(N, H - 1, H - 1, Cout, dt, DEV)
    pk = ops.pack_weight(dt, ops.OP_CONV, 0, w, Cin, Cout)
    stamps = torch.zeros(1 << 20, dtype=torch.int64, device=DEV)
    for rep in range(3):
        stamps.zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        ops.check(lib.tfc_conv_fwd(ops.stream_ptr(), dt, ops.OP_CONV, x.ptr, x.pitch, N, H, H, Cin, Cout, ops._p(pk), y.ptr, y.pitch, None, None, ops._p(stamps), None, 0), "conv")
        e1.record()
        torch.cuda.synchronize()
    s = stamps.cpu().numpy().reshape(-1, 8)
    s = s[s[:, 0] != 0].astype(np.float64)
    rt = s[::4, 6] / 100.0            # wave 0 of each workgroup: run time in us (100 MHz ticks)
    cyc = s[::4, 0]
    print(f"H={H} {Cin}->{Cout}: event time {e0.elapsed_time(e1) * 1e3:.1f} us; workgroup run time us: mean {rt.mean():.1f} min {rt.min():.1f} p50 {np.median(rt):.1f} p90 {np.percentile(rt, 90):.1f} max {rt.max():.1f}; "
          f"cycles mean {cyc.mean():.0f} max {cyc.max():.0f}; clock mean {(cyc / rt).mean() / 1e3:.3f} GHz min {(cyc / rt).min() / 1e3:.3f} max {(cyc / rt).max() / 1e3:.3f}")
    nb = len(rt)
    for xg in range(8):
        m = np.arange(nb) % 8 == xg
        print(f"   blockIdx%8={xg}: run time mean {rt[m].mean():7.1f} max {rt[m].max():7.1f} us   clock {(cyc[m] / rt[m]).mean() / 1e3:.3f} GHz")
    first, second = rt[:256], rt[256:512]
    if len(second):
        print(f"   first-dispatched 256: mean {first.mean():.1f} us; second 256: mean {second.mean():.1f} us")
