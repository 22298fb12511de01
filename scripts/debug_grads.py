"""per-parameter gradient error of the fp32 / bf16 HIP train step against the CPU oracle (N=1)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import tfc_gan_amd as T
from oracle import tfcgan_oracle as O
DEV = "cuda:0"
neg = [3, 3, 7, 0, 4, 9, 15, 2, 8, 8, 1, 12, 5, 13, 6, 10]
A, B = O.synthetic_pairs(1, seed=63)
Gc = O.init_weights_portable(O.GeneratorUNet((3, 256, 256)), seed=61).eval()
Dc = O.init_weights_portable(O.Discriminator1((3, 256, 256)), seed=62).train()
gsd = {k: v.clone() for k, v in Gc.state_dict().items()}
dsd = {k: v.clone() for k, v in Dc.state_dict().items()}
# oracle G-step gradients
fake = Gc(A)
pf = Dc(fake, A); pr = Dc(B, A)
lg = 0.5 * O.loss_gan_generator(pf, pr) + O.patch_triplet_loss(fake, B, neg)
lg.backward()
og = {k: p.grad.clone() for k, p in Gc.named_parameters()}
for cfgname, cfg in (("auto", -1), ("128x32", 2), ("128x64", 1)):
    for dtype in (torch.float32, torch.bfloat16):
        T._lib.load().tfc_debug_set_igemm_config(cfg)
        T.set_compute_dtype(dtype)
        G = T.GeneratorUNet((3, 256, 256)); G.load_state_dict(gsd); G = G.to(DEV).eval()
        D = T.Discriminator1((3, 256, 256)); D.load_state_dict(dsd); D = D.to(DEV).train()
        ts = T.TrainStep(G, D, compute_dtype=dtype, lr=0.0)
        ts.step(A.to(DEV), B.to(DEV), neg_idx=neg)
        torch.cuda.synchronize()
        print(f"--- {cfgname} {dtype}")
        for k in T.nets.g_backward_order():
            got = ts.gflat.grad_views[k].cpu().double(); want = og[k].double()
            print(f"  {k:28s} rel-L2 {((got-want).norm()/want.norm()).item():.3e}")
