"""Diagnostic: where does the bf16 engine leave the oracle's bf16-storage model? Compares the generator's stored tensors layer by layer."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
import tfc_gan_amd as T
from oracle import tfcgan_oracle as O
DEV = "cuda:0"
T.set_compute_dtype(torch.bfloat16)
Gc = O.init_weights_portable(O.GeneratorUNet((3, 256, 256)), seed=61).eval()
A, B = O.synthetic_pairs(1, seed=63)
core = T.nets.GeneratorCore(T.ops.DT_BF16)
core.set_params({k: v.to(DEV) for k, v in Gc.state_dict().items() if k in T.nets.g_param_names()})
fake, ctx = core.forward(A.to(DEV), seed=0, train=False, save=True)
torch.cuda.synchronize()
rb = O._RoundBoth.apply; rw = O._RoundFwd.apply
def nchw(v): return v.t[..., v.coff:v.coff + v.C].float().cpu().permute(0, 3, 1, 2)
with torch.no_grad():
    h = O._bf(A)
    skips = []
    for i, (name, _, _, norm, _) in enumerate(O._DOWNS):
        w = getattr(Gc, name).model[0].weight
        z = O._bf(F.conv2d(h, O._bf(w), padding=1))
        e = nchw(ctx.raw[i])
        print(f"{name}: raw  rel-L2 {((e - z).norm() / z.norm()).item():.3e}  frac != {(e != z).float().mean().item():.4f}")
        zz = F.instance_norm(z, eps=1e-5) if norm else z
        h = O._bf(O._blur(F.leaky_relu(zz, 0.2), 2))
        if i < 5:
            up = {4: "up1", 3: "up2", 2: "up3", 1: "up4", 0: "up5"}[i]
            cat = ctx.cat[up]
            cout = h.shape[1]
            e = cat.t[..., cat.t.shape[3] - cout:].float().cpu().permute(0, 3, 1, 2)
        else:
            e = nchw(ctx.d6)
        print(f"{name}: pool rel-L2 {((e - h).norm() / h.norm()).item():.3e}  frac != {(e != h).float().mean().item():.4f}")
        skips.append(h)
    h = skips.pop()
    for j, (name, *_rest) in enumerate(O._UPS):
        w = getattr(Gc, name).model[0].weight
        z = O._bf(F.conv_transpose2d(h, O._bf(w), stride=2, padding=1))
        z = O._bf(O._blur(z, 1))
        e = nchw(ctx.blur[j])
        print(f"{name}: blur rel-L2 {((e - z).norm() / z.norm()).item():.3e}  frac != {(e != z).float().mean().item():.4f}")
        z = O._bf(F.relu(F.instance_norm(z, eps=1e-5)))
        cat = ctx.cat[name]
        e = cat.t[..., :z.shape[1]].float().cpu().permute(0, 3, 1, 2)
        print(f"{name}: out  rel-L2 {((e - z).norm() / z.norm()).item():.3e}  frac != {(e != z).float().mean().item():.4f}")
        h = torch.cat((z, skips.pop()), dim=1)
    conv = Gc.final[2]
    pre = F.conv2d(F.pad(F.interpolate(h, scale_factor=2), (1, 0, 1, 0)), O._bf(conv.weight), conv.bias, padding=1)
    out = torch.tanh(pre)
    print(f"final: L1 {(fake.cpu() - out).abs().mean().item():.3e}")
