"""Diagnostic: phase shares of tfc_first_block_fwd_kernel from s_memtime stamps; needs a -DTFC_STAMP build (bash scripts/build_variant.sh fbstamp -DTFC_STAMP;
TFC_SO_OVERRIDE=$PWD/diag/lib_fbstamp.so). Read the SHARES, never the run time of this build."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from tfc_gan_amd import ops
DEV, dt, N = "cuda:0", ops.DT_BF16, 32
x = ops.View(torch.randn(N, 256, 256, 8, device=DEV).to(torch.bfloat16), 6)
w = torch.randn(64, 6, 4, 4, device=DEV) * 0.1
pk = ops.pack_weight(dt, ops.OP_CONV, 0, w, 6, 64)
out = ops.new_act(N, 128, 128, 64, dt, DEV)
names = ["conv MFMA", "barrier 1", "halo store + epilogue", "tap matrices + barrier 2", "mask store + blur GEMM", "output stores", "barrier 3 + tile decode", "tap-matrix builds"]
for gform in (False, True):
    st = torch.zeros(N * 255 * 255, dtype=torch.int64, device=DEV)
    for _ in range(2):
        ops.first_block_fwd(dt, x, 6, 64, pk, out, slope=0.2, act_after_rounding=gform, sign_mask=st)
    torch.cuda.synchronize()
    s = st.cpu().numpy()[: 512 * 4 * 8].reshape(-1, 8)
    s = s[s[:, 0] != 0]
    tiles = np.full(len(s), 18.0)                                   # 32 x 32 x 9 tiles over 512 workgroups
    tot = s.sum(1)
    print(f"form {'G' if gform else 'D'}: {len(s)} waves, {tiles.mean():.1f} tiles per wave, {(tot / tiles).mean():.0f} clocks (100 MHz units x shader ratio) per tile")
    for i, n in enumerate(names):
        print(f"   {n:32s} {(s[:, i] / tiles).mean():8.1f} per tile  {s[:, i].sum() / tot.sum():6.1%}")
