"""Experiment: discriminator block 1 forward (first conv -> blur-pool) on the whole batch vs in image chunks, streaming vs default-policy stores
(does the 266 MB intermediate stay in the 256 MiB Infinity Cache when it is produced and consumed chunk by chunk?)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import tfc_gan_amd as T
from tfc_gan_amd import ops
DEV, dt, N = "cuda:0", ops.DT_BF16, 32
x = ops.View(torch.randn(N, 256, 256, 8, device=DEV).to(torch.bfloat16), 6)
w = torch.randn(64, 6, 4, 4, device=DEV) * 0.1
pk = ops.pack_weight(dt, ops.OP_CONV, 0, w, 6, 64)
bias = torch.randn(64, device=DEV) * 0.1
raw = ops.new_act(N, 255, 255, 64, dt, DEV)
out = ops.new_act(N, 128, 128, 64, dt, DEV)
mask = torch.empty(N, 255, 255, 8, dtype=torch.uint8, device=DEV)
def run(chunk):
    for n0 in range(0, N, chunk):
        xs, rs, os_ = ops.View(x.t[n0:n0 + chunk], 6), ops.View(raw.t[n0:n0 + chunk], 64), ops.View(out.t[n0:n0 + chunk], 64)
        ops.conv_first_fwd(dt, xs, 6, 64, pk, rs, bias=bias, flags=ops.EP_LEAKY, sign_mask=mask[n0:n0 + chunk])
        ops.act_fwd(dt, rs, os_, stats=None, slope=1.0, pool=2)
for plain in ("0", "1"):
    os.environ["TFC_C8_PLAIN"] = plain
    for chunk in (32, 16, 8, 4):
        for _ in range(3):
            run(chunk)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            run(chunk)
        e1.record()
        torch.cuda.synchronize()
        print(f"stores {'default' if plain == '1' else 'streaming'}, chunk {chunk:2d} images: {e0.elapsed_time(e1) / 20 * 1e3:7.1f} us per batch (conv + blur-pool)", flush=True)
