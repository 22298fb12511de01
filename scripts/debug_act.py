import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import torch.nn.functional as F
import tfc_gan_amd as T
from oracle import tfcgan_oracle as O
from tfc_gan_amd import ops
from tests.test_gpu_00_kernels import rnd, q, to_view, from_view, oracle_act, DEV
C,H,W,norm,slope,pool,drop = 256,3,3,True,0.2,2,0.0
for dt in (ops.DT_F32, ops.DT_BF16):
    N=2
    x = q(rnd((N, C, H, W), C + H) * 1.5 + 0.3, dt).requires_grad_(True)
    Ho, Wo = 2, 2
    y = oracle_act(x, norm, slope, pool)
    go = q(rnd(tuple(y.shape), 3), dt)
    (gx,) = torch.autograd.grad(y, x, go)
    xv = to_view(x.detach(), dt)
    xs = xv.t.float()
    stats = torch.stack((xs.sum((1, 2)), (xs * xs).sum((1, 2))), -1).contiguous()
    gov = to_view(go, dt)
    dxv = ops.new_act(N, H, W, C, dt, DEV)
    rstats = torch.zeros((N, C, 2), dtype=torch.float32, device=DEV)
    ops.act_bwd(dt, 1, gov, xv, N, H, W, C, None, stats=stats, slope=slope, pool=pool, rstats=rstats)
    ops.act_bwd(dt, 2, gov, xv, N, H, W, C, dxv, stats=stats, slope=slope, pool=pool, rstats=rstats)
    got = from_view(dxv)
    err = (got - gx).abs()
    print("dt", dt, "max err", err.max().item())
    xh = F.instance_norm(x.detach(), eps=1e-5)
    idx = torch.topk(err.flatten(), 6).indices
    for i in idx:
        n, c, yy, xx = np.unravel_index(i.item(), err.shape)
        print((n,c,yy,xx), "got", got[n,c,yy,xx].item(), "want", gx[n,c,yy,xx].item(), "xhat", xh[n,c,yy,xx].item(), "x", x[n,c,yy,xx].item(),
              "plane x", x[n,c].flatten().tolist())
