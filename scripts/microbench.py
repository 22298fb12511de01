"""Micro-benchmark of individual kernels at BASELINE sizes (batch 32) for rocprofv3 --kernel-trace / --pmc runs.
usage: python scripts/microbench.py [which ...]   which in {act, igemm, wgrad}"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import tfc_gan_amd as T
from tfc_gan_amd import ops
DEV = "cuda:0"
dt = ops.DT_BF16
N = 32
which = sys.argv[1:] or ["act", "igemm", "wgrad"]
REP = 3
def rnd(*s):
    return ops.View(torch.randn(*s, device=DEV).to(torch.bfloat16), s[-1])
if "act" in which:
    # Discriminator block 1: raw [32,255,255,64] -> pooled [32,128,128,64]
    raw = rnd(N, 255, 255, 64); out = ops.new_act(N, 128, 128, 64, dt, DEV); g = rnd(N, 128, 128, 64); dx = ops.new_act(N, 255, 255, 64, dt, DEV)
    stats = torch.zeros(N, 64, 2, device=DEV); stats[..., 1] = 255 * 255
    for _ in range(REP):
        ops.act_fwd(dt, raw, out, stats=None, slope=0.2, pool=2)                      # D / down1 forward
        ops.act_fwd(dt, raw, out, stats=stats, slope=0.2, pool=2)                     # normalised variant
        ops.act_bwd(dt, 0, g, raw, N, 255, 255, 64, dx, slope=0.2, pool=2)            # D backward
    # up5: blur s1 transpose on [32,128,128,64]
    a = rnd(N, 128, 128, 64); b = ops.new_act(N, 128, 128, 64, dt, DEV)
    for _ in range(REP):
        ops.act_bwd(dt, 0, a, None, N, 128, 128, 64, b, slope=1.0, pool=1)
if "cfg" in which:
    import time
    lib = T._lib.load()
    shapes = [(128, 64, 128), (64, 128, 256), (32, 256, 512)]          # (H, Cin, Cout): down2 / down3 / down4
    for H, Cin, Cout in shapes:
        x = rnd(N, H, H, Cin); w = torch.randn(Cout, Cin, 4, 4, device=DEV) * 0.03; y = ops.new_act(N, H - 1, H - 1, Cout, dt, DEV)
        pk = ops.pack_weight(dt, ops.OP_CONV, 0, w, Cin, Cout)
        for cfg in (0, 3, 1):
            lib.tfc_debug_set_igemm_config(cfg)
            for _ in range(2):
                ops.conv_fwd(dt, ops.OP_CONV, x, Cin, Cout, pk, y)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(10):
                ops.conv_fwd(dt, ops.OP_CONV, x, Cin, Cout, pk, y)
            torch.cuda.synchronize(); dtm = (time.perf_counter() - t0) / 10
            fl = 2.0 * N * (H - 1) ** 2 * Cin * Cout * 16
            print(f"H={H} {Cin}->{Cout} cfg {cfg}: {dtm*1e6:7.1f} us  {fl/dtm/1e12:7.1f} TFLOP/s")
    lib.tfc_debug_set_igemm_config(-1)
if "deep" in which:
    # the small-plane layers of the generator: every tile configuration, forward (and dgrad for the transposed ones)
    import time
    lib = T._lib.load()
    cases = [(ops.OP_CONV, 16, 512, 512), (ops.OP_CONV, 8, 512, 512), (ops.OP_CONVT, 4, 512, 512), (ops.OP_CONVT, 8, 1024, 512), (ops.OP_CONVT, 16, 1024, 256),
             (ops.OP_CONVT, 32, 512, 128), (ops.OP_CONVT, 64, 256, 64), (ops.OP_CONV, 32, 256, 512)]
    for op, H, Cin, Cout in cases:
        x = rnd(N, H, H, Cin)
        w = torch.randn((Cin, Cout, 4, 4) if op == ops.OP_CONVT else (Cout, Cin, 4, 4), device=DEV) * 0.03
        oh = ops.OUT_HW[op](H)
        y = ops.new_act(N, oh, oh, Cout, dt, DEV)
        pk = ops.pack_weight(dt, op, 0, w, Cin, Cout)
        for cfg in (-1, 0, 1, 2, 3):
            lib.tfc_debug_set_igemm_config(cfg)
            for _ in range(2):
                ops.conv_fwd(dt, op, x, Cin, Cout, pk, y)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(20):
                ops.conv_fwd(dt, op, x, Cin, Cout, pk, y)
            torch.cuda.synchronize(); dtm = (time.perf_counter() - t0) / 20
            fl = 2.0 * N * (oh * oh if op == ops.OP_CONV else H * H * 4) * Cin * Cout * (16 if op == ops.OP_CONV else 4)
            print(f"op {op} H={H} {Cin}->{Cout} cfg {cfg:2d}: {dtm*1e6:7.1f} us  {fl/dtm/1e12:7.1f} TFLOP/s")
    lib.tfc_debug_set_igemm_config(-1)
if "wgt" in which:
    # timed weight gradients of the three big conv layers and the transposed ones (hipEvents around 10 calls each)
    shapes = [(ops.OP_CONV, 128, 64, 128), (ops.OP_CONV, 64, 128, 256), (ops.OP_CONV, 32, 256, 512), (ops.OP_CONV, 16, 512, 512),
              (ops.OP_CONVT, 64, 256, 64), (ops.OP_CONVT, 32, 512, 128), (ops.OP_CONVT, 16, 1024, 256), (ops.OP_CONVT, 8, 1024, 512)]
    ws = None
    for op, H, Cin, Cout in shapes:
        x = rnd(N, H, H, Cin)
        oh = ops.OUT_HW[op](H)
        gy = rnd(N, oh, oh, Cout)
        dw = torch.empty((Cin, Cout, 4, 4) if op == ops.OP_CONVT else (Cout, Cin, 4, 4), device=DEV)
        for _ in range(2):
            ws = ops.conv_wgrad(dt, op, x, gy, Cin, Cout, dw, False, ws)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10):
            ws = ops.conv_wgrad(dt, op, x, gy, Cin, Cout, dw, False, ws)
        b.record(); torch.cuda.synchronize()
        us = a.elapsed_time(b) * 100
        fl = 2.0 * N * (oh * oh * 16 if op == ops.OP_CONV else H * H * 4 * 4) * Cin * Cout
        print(f"wgrad op {op} H={H} {Cin}->{Cout}: {us:7.1f} us  {fl / us / 1e6:7.1f} TFLOP/s")
if "stream" in which:
    import time
    def timeit(fn, reps=10):
        for _ in range(2): fn()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(reps): fn()
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps
    raw = rnd(N, 255, 255, 64); dxv = ops.new_act(N, 255, 255, 64, dt, DEV); gfull = rnd(N, 255, 255, 64); gpool = rnd(N, 128, 128, 64)
    outp = ops.new_act(N, 128, 128, 64, dt, DEV)
    MB = 1e6
    t = timeit(lambda: ops.act_bwd(dt, 0, gfull, raw, N, 255, 255, 64, dxv, slope=0.2, pool=0))
    print(f"act_bwd pool0 (read 2x266MB, write 266MB): {t*1e6:7.1f} us  {(3*266.3*MB)/t/1e12:5.2f} TB/s")
    t = timeit(lambda: ops.act_bwd(dt, 0, gpool, raw, N, 255, 255, 64, dxv, slope=0.2, pool=2))
    print(f"act_bwd pool2 (read 266+67MB, write 266MB): {t*1e6:7.1f} us  {((2*266.3+67.1)*MB)/t/1e12:5.2f} TB/s")
    t = timeit(lambda: ops.act_fwd(dt, raw, outp, stats=None, slope=0.2, pool=2))
    print(f"act_fwd pool2 (read 266MB, write 67MB): {t*1e6:7.1f} us  {((266.3+67.1)*MB)/t/1e12:5.2f} TB/s")
    t = timeit(lambda: ops.act_fwd(dt, raw, dxv, stats=None, slope=0.2, pool=0))
    print(f"act_fwd pool0 (read 266MB, write 266MB): {t*1e6:7.1f} us  {((2*266.3)*MB)/t/1e12:5.2f} TB/s")
    a32 = torch.randn(N * 255 * 255 * 64 // 2, device=DEV); b32 = torch.empty_like(a32)
    t = timeit(lambda: b32.copy_(a32))
    print(f"torch copy 266MB->266MB: {t*1e6:7.1f} us  {(2*a32.numel()*4)/t/1e12:5.2f} TB/s")
if "igemm" in which:
    # down2 / D2: 64 -> 128 @ 128x128 ; D1: 8 -> 64 @ 256x256
    x = rnd(N, 128, 128, 64); w = torch.randn(128, 64, 4, 4, device=DEV) * 0.03; y = ops.new_act(N, 127, 127, 128, dt, DEV)
    pk = ops.pack_weight(dt, ops.OP_CONV, 0, w, 64, 128); pkd = ops.pack_weight(dt, ops.OP_CONV, 1, w, 64, 128)
    gy = rnd(N, 127, 127, 128); gx = ops.new_act(N, 128, 128, 64, dt, DEV)
    x1 = rnd(N, 256, 256, 8); w1 = torch.randn(64, 6, 4, 4, device=DEV) * 0.1; y1 = ops.new_act(N, 255, 255, 64, dt, DEV)
    pk1 = ops.pack_weight(dt, ops.OP_CONV, 0, w1, 6, 64)
    for _ in range(REP):
        ops.conv_fwd(dt, ops.OP_CONV, x, 64, 128, pk, y)
        ops.conv_dgrad(dt, ops.OP_CONV, gy, N, 128, 128, 64, 128, pkd, gx)
        ops.conv_fwd(dt, ops.OP_CONV, x1, 6, 64, pk1, y1)
if "wgrad" in which:
    x = rnd(N, 128, 128, 64); gy = rnd(N, 127, 127, 128); dw = torch.empty(128, 64, 4, 4, device=DEV)
    x3 = rnd(N, 32, 32, 256); gy3 = rnd(N, 31, 31, 512); dw3 = torch.empty(512, 256, 4, 4, device=DEV)
    for _ in range(REP):
        ops.conv_wgrad(dt, ops.OP_CONV, x, gy, 64, 128, dw)
        ops.conv_wgrad(dt, ops.OP_CONV, x3, gy3, 256, 512, dw3)
torch.cuda.synchronize()
print("done")
if "cfg4" in which:
    # probe of a 128-pixel x 256-channel workgroup tile (wave = 128 x 64; needs a library built with -DTFC_PROBE_W64) against <4,1,1,4>
    import time
    lib = T._lib.load()
    shapes = [(64, 128, 256), (32, 256, 512), (16, 512, 512)]
    for H, Cin, Cout in shapes:
        x = rnd(N, H, H, Cin); w = torch.randn(Cout, Cin, 4, 4, device=DEV) * 0.03
        pk = ops.pack_weight(dt, ops.OP_CONV, 0, w, Cin, Cout)
        ys = {}
        for cfg in (3, 4):
            y = ops.new_act(N, H - 1, H - 1, Cout, dt, DEV)
            lib.tfc_debug_set_igemm_config(cfg)
            for _ in range(2):
                ops.conv_fwd(dt, ops.OP_CONV, x, Cin, Cout, pk, y)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(10):
                ops.conv_fwd(dt, ops.OP_CONV, x, Cin, Cout, pk, y)
            torch.cuda.synchronize(); dtm = (time.perf_counter() - t0) / 10
            fl = 2.0 * N * (H - 1) ** 2 * Cin * Cout * 16
            ys[cfg] = y.t.float()
            print(f"H={H} {Cin}->{Cout} cfg {cfg}: {dtm*1e6:7.1f} us  {fl/dtm/1e12:7.1f} TFLOP/s", flush=True)
        print("   max |cfg4 - cfg3| =", (ys[3] - ys[4]).abs().max().item(), flush=True)
    lib.tfc_debug_set_igemm_config(-1)
if "overlap" in which:
    # do two latency-bound MFMA kernels co-run faster than back to back? conv fwd/dgrad chain on one stream, wgrad on another
    import time
    x = rnd(N, 128, 128, 64); w = torch.randn(128, 64, 4, 4, device=DEV) * 0.03; y = ops.new_act(N, 127, 127, 128, dt, DEV)
    pk = ops.pack_weight(dt, ops.OP_CONV, 0, w, 64, 128); pkd = ops.pack_weight(dt, ops.OP_CONV, 1, w, 64, 128)
    gy = rnd(N, 127, 127, 128); gx = ops.new_act(N, 128, 128, 64, dt, DEV)
    dw = torch.empty(128, 64, 4, 4, device=DEV)
    raw = rnd(N, 255, 255, 64); dxv = ops.new_act(N, 255, 255, 64, dt, DEV); gpool = rnd(N, 128, 128, 64)
    ws = ops.conv_wgrad(dt, ops.OP_CONV, x, gy, 64, 128, dw)
    s2 = torch.cuda.Stream()
    def seq():
        ops.conv_dgrad(dt, ops.OP_CONV, gy, N, 128, 128, 64, 128, pkd, gx)
        ops.act_bwd(dt, 0, gpool, raw, N, 255, 255, 64, dxv, slope=0.2, pool=2)
        ops.conv_wgrad(dt, ops.OP_CONV, x, gy, 64, 128, dw, ws=ws)
    def par():
        ev = torch.cuda.Event(); ev.record()
        with torch.cuda.stream(s2):
            s2.wait_event(ev)
            ops.conv_wgrad(dt, ops.OP_CONV, x, gy, 64, 128, dw, ws=ws)
            ev2 = torch.cuda.Event(); ev2.record()
        ops.conv_dgrad(dt, ops.OP_CONV, gy, N, 128, 128, 64, 128, pkd, gx)
        ops.act_bwd(dt, 0, gpool, raw, N, 255, 255, 64, dxv, slope=0.2, pool=2)
        torch.cuda.current_stream().wait_event(ev2)
    for name, fn in (("sequential", seq), ("two streams", par), ("sequential", seq), ("two streams", par)):
        for _ in range(3): fn()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20): fn()
        torch.cuda.synchronize(); print(f"{name}: {(time.perf_counter() - t0) / 20 * 1e6:8.1f} us per (dgrad + act_bwd + wgrad)")
