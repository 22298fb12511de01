"""localise the fp32 gradient error between `final` and `up5`: intermediates vs oracle autograd, plus a determinism check"""
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
keep = {}
def fh(name):
    def hook(mod, inp, out):
        out.retain_grad(); keep[name] = out
    return hook
Gc.up5.register_forward_hook(fh("u5"))
Gc.up5.model[0].register_forward_hook(fh("up5.rawT"))
Gc.up5.model[1].register_forward_hook(fh("up5.blur"))
fake = Gc(A); fake.retain_grad()
pf = Dc(fake, A); pr = Dc(B, A)
lg = 0.5 * O.loss_gan_generator(pf, pr) + O.patch_triplet_loss(fake, B, neg)
lg.backward()
og = {k: p.grad.clone() for k, p in Gc.named_parameters()}
def rel(got, want):
    got, want = got.double(), want.double()
    return ((got - want).norm() / want.norm()).item()
def nchw(v, c=None):
    c = c or v.C
    return v.t[..., v.coff:v.coff + c].float().cpu().permute(0, 3, 1, 2)
T.set_compute_dtype(torch.float32)
res = []
for run in range(2):
    G = T.GeneratorUNet((3, 256, 256)); G.load_state_dict(gsd); G = G.to(DEV).eval()
    D = T.Discriminator1((3, 256, 256)); D.load_state_dict(dsd); D = D.to(DEV).train()
    ts = T.TrainStep(G, D, compute_dtype=torch.float32, lr=0.0)
    ts.G.debug = {}
    out = ts.step(A.to(DEV), B.to(DEV), neg_idx=neg)
    torch.cuda.synchronize()
    d = ts.G.debug
    print(f"run {run}: fake {rel(out['fake_B'].cpu(), fake.detach()):.2e}  dyf {rel(nchw(d['dyf'],3), (fake.grad*(1-fake.detach()**2))):.2e}  g_u5 {rel(nchw(d['g_u5'],128), keep['u5'].grad):.2e}"
          f"  up5.d_blur {rel(nchw(d['up5.d_blur']), keep['up5.blur'].grad):.2e}  up5.d_rawT {rel(nchw(d['up5.d_rawT']), keep['up5.rawT'].grad):.2e}")
    g5 = nchw(d['g_u5'],128); w5 = keep['u5'].grad
    print("    g_u5 up-part", rel(g5[:, :64], w5[:, :64]), " skip-part", rel(g5[:, 64:], w5[:, 64:]), " |g| max", w5.abs().max().item(), " mean|g|", w5.abs().mean().item())
    e = (g5 - w5).abs(); i = e.flatten().argmax().item(); idx = np.unravel_index(i, e.shape)
    print("    worst g_u5 element", idx, "got", g5[idx].item(), "want", w5[idx].item())
    res.append({k: ts.gflat.grad_views[k].clone() for k in T.nets.g_backward_order()})
    for k in ("final.2.weight", "up5.model.0.weight", "down1.model.0.weight"):
        print(f"    {k} vs oracle {rel(res[-1][k].cpu(), og[k]):.3e}")
print("determinism: run0 vs run1")
for k in ("final.2.weight", "up5.model.0.weight", "down1.model.0.weight"):
    print(f"    {k} {rel(res[0][k], res[1][k]):.3e}")
