"""STN21 configuration (BASELINE.json configs[4]; reference TFC-STN/TFCGAN_STN21_Original_NewModel3_Official.py, "STN") as a runnable step.

The script's generators and discriminators are the SAME classes as the PATCH-16 path (GeneratorUNet1/2 = STN:276-357 and Discriminator1/2 =
STN:360-420 match P16:136-211 layer for layer), so `GeneratorUNet` / `Discriminator1` of this package serve as all four; what the
configuration adds is

    Net (STN:170-231)          localiser (kornia VisionTransformer + MLP) -> theta -> per-sample bicubic warp            -> `Net` below
    morph_triplet (STN:444-459)                                                                                           -> stn.morph_triplet
    the training step (STN:609-672): fake_B = G1(A); fake_A1 = G2(B); warped_B = Net(A, fake_A1, src=B); fake_A2 = G2(warped_B)
        loss_G = mean(GAN1 + GAN2) + 0.01 * L1(fake_A2, A) + mean(LPIPS(fake_A2, A) + LPIPS(fake_B, B)) + morph_triplet(A, B, warped_B)
        loss_D = 0.5 * (0.25 * (...D1...) + 0.25 * (...D2...))                                                         -> `STN21Step`

Everything heavy runs on the package's HIP kernels through the modules' autograd Functions (both generators, both discriminators, the warp,
the morphological gradient, LPIPS); the generator now returns its INPUT gradient, which is how the warp and the localiser train through
`G2(warped_B)`. The localiser is 17 tokens x 768 channels: plain torch layers (library GEMMs) -- kornia is absent from this image, so
`VisionTransformer` is restated from its published architecture (patch embedding, class token, learned positions, 12 pre-norm encoder blocks
with 12 heads and a 4x GELU MLP, final LayerNorm) with matmul / softmax / LayerNorm only (no MIOpen convolution, no fused attention: their
first-use compilation takes minutes on a fresh box): PARITY UNPINNED. The step keeps torch autograd across the five modules (their heavy
parts are the package's own hand-written forward / backward chains) but owns the parameters the way the PATCH-16 engine does: flat fp32
buffers, bucketed all-reduce from gradient hooks, `tfc_adam_step` -- so configuration C5 shards over GPUs like C4. A reference `model`
(STN) checkpoint does NOT load into `Net`: kornia's VisionTransformer key names are not reproduced.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import stn
from .models import Discriminator1, GeneratorUNet, weights_init_normal


class _Attention(nn.Module):
    """multi-head self-attention over 17 tokens as plain matmuls + softmax (library GEMMs; no fused-attention or MIOpen kernels, whose first-use
    compilation takes minutes on a fresh box)"""

    def __init__(self, dim, heads):
        super().__init__()
        self.heads = heads
        self.qkv = nn.Linear(dim, 3 * dim)
        self.proj = nn.Linear(dim, dim)

    def forward(self, x):
        n, t, d = x.shape
        q, k, v = self.qkv(x).reshape(n, t, 3, self.heads, d // self.heads).permute(2, 0, 3, 1, 4)
        att = torch.softmax((q @ k.transpose(-2, -1)) * (d // self.heads) ** -0.5, dim=-1)
        return self.proj((att @ v).transpose(1, 2).reshape(n, t, d))


class _EncoderBlock(nn.Module):
    def __init__(self, dim, heads, mlp_ratio=4):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim, eps=1e-6)
        self.attn = _Attention(dim, heads)
        self.norm2 = nn.LayerNorm(dim, eps=1e-6)
        self.mlp = nn.Sequential(nn.Linear(dim, mlp_ratio * dim), nn.GELU(), nn.Linear(mlp_ratio * dim, dim))

    def forward(self, x):
        x = x + self.attn(self.norm1(x))
        return x + self.mlp(self.norm2(x))


class _PatchEmbed(nn.Module):
    """Conv2d(in_channels, embed_dim, patch, stride=patch) computed as unfold + GEMM (the parameter keeps the convolution's shape and name)"""

    def __init__(self, in_channels, embed_dim, patch):
        super().__init__()
        self.patch_size = patch
        self.weight = nn.Parameter(torch.randn(embed_dim, in_channels, patch, patch) * 0.02)
        self.bias = nn.Parameter(torch.zeros(embed_dim))

    def forward(self, x):
        n, c, h, w = x.shape
        p = self.patch_size
        t = x.reshape(n, c, h // p, p, w // p, p).permute(0, 2, 4, 1, 3, 5).reshape(n, (h // p) * (w // p), c * p * p)
        return F.linear(t, self.weight.reshape(self.weight.shape[0], -1), self.bias)


class VisionTransformer(nn.Module):
    """kornia.contrib.VisionTransformer(image_size, patch_size, in_channels) restated (defaults embed_dim 768, depth 12, 12 heads): returns
    the encoded tokens [N, 1 + (image_size / patch_size)^2, 768] (class token first). Parity unpinned."""

    def __init__(self, image_size=256, patch_size=64, in_channels=6, embed_dim=768, depth=12, num_heads=12):
        super().__init__()
        self.patch = _PatchEmbed(in_channels, embed_dim, patch_size)
        self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
        n = (image_size // patch_size) ** 2 + 1
        self.positions = nn.Parameter(torch.randn(n, embed_dim) * 0.02)
        self.blocks = nn.Sequential(*[_EncoderBlock(embed_dim, num_heads) for _ in range(depth)])
        self.norm = nn.LayerNorm(embed_dim, eps=1e-6)

    def forward(self, x):
        t = self.patch(x)
        t = torch.cat((self.cls_token.expand(t.shape[0], -1, -1), t), 1) + self.positions
        return self.norm(self.blocks(t))


class LocalizerVIT(nn.Module):
    """STN:150-163"""

    def __init__(self, img_shape):
        super().__init__()
        channels, self.h, self.w = img_shape
        self.vit = nn.Sequential(VisionTransformer(image_size=self.h, patch_size=64, in_channels=channels * 2))

    def forward(self, x):
        return self.vit(x)


class Net(nn.Module):
    """STN:170-231. forward(img_A, img_B, src): theta = identity + fc_loc(ViT(cat(img_A, img_B))), every sample of `src` warped with its own
    matrix (F.affine_grid + F.grid_sample(bicubic, border, align_corners=True), fused in tfc_affine_warp_fwd/bwd)."""

    def __init__(self, img_shape=(3, 256, 256)):
        super().__init__()
        channels, h, w = img_shape
        self.localization = LocalizerVIT(img_shape)
        self.theta_emb = nn.Linear(1, h * w)                      # declared and never used by the reference (STN:178); kept for the state_dict
        ntok = (h // 64) * (w // 64) + 1
        self.fc_loc = nn.Sequential(nn.Linear(ntok * 768, 1024), nn.ReLU(True), nn.Linear(1024, 512), nn.ReLU(True), nn.Linear(512, 256), nn.Sigmoid(),
                                    nn.Linear(256, 3 * 2))
        self.fc_loc[2].bias.data.zero_()                          # STN:193
        self.warp = stn.Warp()

    def stn_phi(self, x):
        xs = self.localization(x)
        return self.fc_loc(xs.reshape(xs.shape[0], -1)).view(-1, 2, 3)

    def forward(self, img_A, img_B, src):
        dtheta = self.stn_phi(torch.cat((img_A, img_B), 1))
        return self.warp(dtheta, src)


def _bce_rel(a, b, target):
    """criterion_GAN(a - b, target) with a constant target (STN:482-505: BCEWithLogitsLoss, mean over all logits)"""
    x = a - b
    return F.binary_cross_entropy_with_logits(x, torch.full_like(x, target))


class STN21Step:
    """The batch-loop body of STN:609-672 as an engine: the five networks' parameters live in TWO flat fp32 buffers in gradient-completion order
    (generator side: localiser + MLP, generator 2, generator 1 -- optimizer_G of STN:546; discriminator side: both discriminators -- optimizer_D),
    module parameters are views of them, autograd accumulates straight into the flat gradient buffers, `tfc_adam_step` updates each side in one
    launch, and -- one process per GPU, as for PATCH-16 -- the gradient buffers are sum-all-reduced in buckets that are issued from
    post-accumulate hooks while the rest of the backward still runs (the reference wraps all five modules in nn.DataParallel, STN:536-540).
    Every loss of the step is a batch mean except LPIPS, a batch SUM in lpips_pytorch: it is scaled by the world size so that the rank-averaged
    gradient equals the reference's on the gathered batch. lpips: a module with the reference's call surface (tfc_gan_amd.LPIPS) or None."""

    def __init__(self, img_shape=(3, 256, 256), lpips=None, lr=2e-4, b1=0.5, b2=0.999, device="cuda:0", alpha2=0.01, eps=1e-8, bucket_bytes=32 << 20,
                 seed=0):
        from . import ops, parallel
        dev = torch.device(device)
        self.dev = dev
        self.G1, self.G2 = GeneratorUNet(img_shape).to(dev), GeneratorUNet(img_shape).to(dev)
        self.D1, self.D2 = Discriminator1(img_shape).to(dev), Discriminator1(img_shape).to(dev)
        self.net = Net(img_shape).to(dev)
        for m in (self.G1, self.G2, self.D1, self.D2, self.net):
            m.apply(weights_init_normal)                          # STN:539-543
        self.lpips, self.alpha2, self._bucket_bytes = lpips, alpha2, bucket_bytes
        self.lr, self.b1, self.b2, self.eps = lr, b1, b2, eps
        self.step_no = 0
        for i, m in enumerate((self.G1, self.G2)):                # dropout streams: a function of (seed, rank, module), not of id()
            object.__setattr__(m, "_drop_seed", (seed * 7919 + parallel.rank() * 1299709 + i * 104729 + 0x5EED) & 0x7FFFFFFF)
        self._flatten()

    def _flatten(self):
        """(re)build the flat buffers from the modules' current parameters (call again after load_state_dict on a module)"""
        from . import parallel
        gnamed, dnamed = {}, {}
        # gradient-completion order of loss_G.backward(): fake_A2 = G2(warped_B) is the last forward and the first backward, the localiser follows,
        # then G2's first use finishes its gradients, G1 (the first forward) comes last
        for pre, m in (("net.", self.net), ("G2.", self.G2), ("G1.", self.G1)):
            items = list(m.named_parameters())
            for k, p in (reversed(items) if pre == "net." else items):
                gnamed[pre + k] = p
        for pre, m in (("D1.", self.D1), ("D2.", self.D2)):
            for k, p in m.named_parameters():
                dnamed[pre + k] = p
        self.gflat = parallel.FlatParams(gnamed, list(gnamed), self.dev)
        self.dflat = parallel.FlatParams(dnamed, list(dnamed), self.dev)
        self._hooks = []
        for flat, named, side in ((self.gflat, gnamed, "g"), (self.dflat, dnamed, "d")):
            for k, p in named.items():
                p.data = flat.views[k]
                p.grad = flat.grad_views[k]
                self._hooks.append(p.register_post_accumulate_grad_hook(lambda _p, k=k, side=side: self._ready(side, k)))
        parallel.broadcast_flat(self.gflat)
        parallel.broadcast_flat(self.dflat)
        parallel.broadcast_tensors([b for D in (self.D1, self.D2) for b in D.named_core_buffers().values()])
        self.gm, self.gv = torch.zeros_like(self.gflat.data), torch.zeros_like(self.gflat.data)
        self.dm, self.dv = torch.zeros_like(self.dflat.data), torch.zeros_like(self.dflat.data)
        self.g_reduce = parallel.BucketReducer(self.gflat, self._bucket_bytes if hasattr(self, "_bucket_bytes") else 32 << 20)
        self.d_reduce = parallel.BucketReducer(self.dflat, 8 << 20)
        self._bump()

    def _ready(self, side, name):
        (self.g_reduce if side == "g" else self.d_reduce).ready(name)

    def _bump(self):
        """the raw-pointer Adam kernel moves neither data_ptr nor _version of a parameter: tell the modules' operand-stream caches"""
        for m in (self.G1, self.G2, self.D1, self.D2):
            m._weights_gen += 1

    def _d_requires_grad(self, on):
        for D in (self.D1, self.D2):
            for p in D.parameters():
                p.requires_grad_(on)

    def step(self, real_A, real_B):
        from . import ops, parallel
        valid, fake_t = 0.9, 0.0                                  # STN:613-615
        self.step_no += 1
        world = parallel.world_size()
        # ---------------- generators + STN (STN:620-662) ----------------
        self.gflat.grad.zero_()                                   # autograd ACCUMULATES into the flat gradient views
        self._d_requires_grad(False)                              # the reference computes (and then zeroes) discriminator gradients here: skipped
        fake_B = self.G1(real_A)
        fake_A1 = self.G2(real_B)
        warped_B = self.net(real_A, fake_A1, real_B)
        fake_A2 = self.G2(warped_B)
        recon = F.l1_loss(fake_A2, real_A)
        if self.lpips is not None:
            perc = (self.lpips(fake_A2, real_A) + self.lpips(fake_B, real_B)).mean()      # a batch SUM inside lpips_pytorch (STN:641-643)
        else:
            perc = fake_B.new_zeros(())
        morph = stn.morph_triplet(real_A, real_B, warped_B)
        gan1 = _bce_rel(self.D1(fake_B, real_A), self.D1(real_B, real_A).detach(), valid)
        gan2 = _bce_rel(self.D2(fake_A2, real_B), self.D2(real_A, real_B).detach(), valid)
        loss_gan = gan1 + gan2                                    # (loss_GAN1 + loss_GAN2).mean() of two scalars
        loss_G = loss_gan + self.alpha2 * recon + perc + morph
        (loss_gan + self.alpha2 * recon + perc * world + morph).backward()       # hooks issue the bucket all-reduces as gradients become final
        gscale = self.g_reduce.finish()
        ops.adam_step(self.gflat.data, self.gflat.grad, self.gm, self.gv, self.lr, self.b1, self.b2, self.eps, self.step_no, gscale)
        # ---------------- discriminators (STN:668-676) ----------------
        self._d_requires_grad(True)
        self.dflat.grad.zero_()

        def disc(D, real, fake_img, cond):
            pr, pf = D(real, cond), D(fake_img.detach(), cond)
            return 0.25 * (_bce_rel(pr, pf, valid) + _bce_rel(pf, pr, fake_t))
        loss_D1, loss_D2 = disc(self.D1, real_B, fake_B, real_A), disc(self.D2, real_A, fake_A2, real_B)
        loss_D = 0.5 * (loss_D1 + loss_D2)
        loss_D.backward()
        dscale = self.d_reduce.finish()
        ops.adam_step(self.dflat.data, self.dflat.grad, self.dm, self.dv, self.lr, self.b1, self.b2, self.eps, self.step_no, dscale)
        self._bump()
        out = {"loss_G": loss_G.detach(), "loss_GAN": loss_gan.detach(), "recon_loss": recon.detach(), "perc_loss": perc.detach(),
               "morph_loss": morph.detach(), "loss_D": loss_D.detach(), "loss_D1": loss_D1.detach(), "loss_D2": loss_D2.detach()}
        if world > 1:                                             # batch means: the mean over ranks is the global-batch value (LPIPS: the sum)
            keys = list(out)
            packed = torch.stack([out[k].reshape(()).float() for k in keys])
            parallel.all_reduce_mean(packed)
            out = {k: packed[i] * (world if k == "perc_loss" else 1.0) for i, k in enumerate(keys)}
            out["loss_G"] = out["loss_GAN"] + self.alpha2 * out["recon_loss"] + out["perc_loss"] + out["morph_loss"]
        out.update({"fake_B": fake_B.detach(), "fake_A2": fake_A2.detach(), "warped_B": warped_B.detach()})
        return out
