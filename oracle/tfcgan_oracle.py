"""CPU ORACLE for the PATCH-16 training hot path of nudro/TFC-GAN  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg may import this module; nothing under tfc-gan_amd/
does (the product path has no CPU fallback and fails loudly without the HIP library).

What it is: a plain torch-CPU fp32 / numpy float64 restatement of the reference's algorithm, function by function, each
citing the reference lines it follows ("P16" = TFC-GAN-FFT/TFCGAN_multigpu_patchFFT_16P.py, "G16" =
TFC-GAN-FFT/TFCGAN_multigpu_globalFFT_16P.py).

How it is pinned: the reference has no tests, golden vectors or fixtures of its own, and the script
cannot be imported (module-level argparse / .cuda() / missing torchvision, lpips_pytorch, antialiased_cnns, cv2).  The
oracle is therefore pinned by OUTPUTS OF THE REFERENCE'S OWN DEFINITIONS executed in the build container:
tests/golden/make_golden.py lifts the reference's class / function definitions by `ast` from /root/reference, runs them
on seeded inputs and stores small fixtures under tests/golden/*.npz; tests/test_oracle_golden.py checks every function
below against those fixtures.  Third-party pieces that are NOT in /root/reference stay "parity unpinned":
  * antialiased_cnns.BlurPool (pip antialiased-cnns, version unpinned by the reference): restated from its published
    algorithm (filt_size 4 -> reflect pad (1,2,1,2), binomial [1,3,3,1]^2/64, depthwise conv, buffer `filt`);
  * torchvision.transforms.ToPILImage on a float tensor (= mul(255).byte(), CHW->HWC, mode RGB);
  * lpips_pytorch.LPIPS (weights unfetchable offline) -- not part of this path.
"""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

# ---------------------------------------------------------------------------------------------------------------
# networks
# ---------------------------------------------------------------------------------------------------------------
BLUR_TAPS = (1.0, 3.0, 3.0, 1.0)


class BlurPool(nn.Module):
    """antialiased_cnns.BlurPool(channels, stride) as used at P16:109, :123, :192 (third party; parity unpinned)."""

    def __init__(self, channels, stride=2):
        super().__init__()
        k = torch.tensor(BLUR_TAPS)
        k2 = torch.outer(k, k)
        self.register_buffer("filt", (k2 / k2.sum())[None, None].repeat(channels, 1, 1, 1))
        self.stride = stride
        self.channels = channels

    def forward(self, x):
        return F.conv2d(F.pad(x, (1, 2, 1, 2), mode="reflect"), self.filt, stride=self.stride, groups=self.channels)


class MaskedDropout(nn.Module):
    """nn.Dropout(p) (P16:110-111, :127-128) with an optional EXPLICIT keep-mask so that a HIP run and the oracle can share
    the same random draw: `mask_fn(tag, shape) -> bool tensor` when set, torch's own RNG otherwise."""

    def __init__(self, p, tag):
        super().__init__()
        self.p, self.tag, self.mask_fn = p, tag, None

    def forward(self, x):
        if not self.training or self.p == 0.0:
            return x
        if self.mask_fn is None:
            return F.dropout(x, self.p, True)
        return x * self.mask_fn(self.tag, x.shape).to(x.dtype) / (1.0 - self.p)


class UNetDown(nn.Module):
    """P16:102-115: Conv2d(k4,s1,p1,no bias) -> [InstanceNorm2d] -> LeakyReLU(0.2) -> BlurPool(s2) -> [Dropout]."""

    def __init__(self, in_size, out_size, normalize=True, dropout=0.0, tag=""):
        super().__init__()
        seq = [nn.Conv2d(in_size, out_size, kernel_size=4, stride=1, padding=1, bias=False)]
        seq += [nn.InstanceNorm2d(out_size)] if normalize else []
        seq += [nn.LeakyReLU(0.2), BlurPool(out_size, stride=2)]
        seq += [MaskedDropout(dropout, tag)] if dropout else []
        self.model = nn.Sequential(*seq)

    def forward(self, x):
        return self.model(x)


class UNetUp(nn.Module):
    """P16:118-134: ConvTranspose2d(k4,s2,p1,no bias) -> BlurPool(s1) -> InstanceNorm2d -> ReLU -> [Dropout]; cat(skip)."""

    def __init__(self, in_size, out_size, dropout=0.0, tag=""):
        super().__init__()
        seq = [nn.ConvTranspose2d(in_size, out_size, kernel_size=4, stride=2, padding=1, bias=False), BlurPool(out_size, stride=1),
               nn.InstanceNorm2d(out_size), nn.ReLU()]
        seq += [MaskedDropout(dropout, tag)] if dropout else []
        self.model = nn.Sequential(*seq)

    def forward(self, x, skip_input):
        return torch.cat((self.model(x), skip_input), dim=1)


_DOWNS = (("down1", None, 64, False, 0.0), ("down2", 64, 128, True, 0.0), ("down3", 128, 256, True, 0.5),
          ("down4", 256, 512, True, 0.5), ("down5", 512, 512, False, 0.0), ("down6", 512, 512, True, 0.0))
_UPS = (("up1", 512, 512, 0.0), ("up2", 1024, 512, 0.5), ("up3", 1024, 256, 0.5), ("up4", 512, 128, 0.0), ("up5", 256, 64, 0.0))


class GeneratorUNet(nn.Module):
    """P16:136-174 (fp32: autocast is inert on CPU; the final `.type(HalfTensor)` is dropped)."""

    def __init__(self, img_shape):
        super().__init__()
        channels = img_shape[0]
        for name, cin, cout, norm, drop in _DOWNS:
            setattr(self, name, UNetDown(channels if cin is None else cin, cout, normalize=norm, dropout=drop, tag=name))
        for name, cin, cout, drop in _UPS:
            setattr(self, name, UNetUp(cin, cout, dropout=drop, tag=name))
        self.final = nn.Sequential(nn.Upsample(scale_factor=2), nn.ZeroPad2d((1, 0, 1, 0)), nn.Conv2d(128, channels, 4, padding=1), nn.Tanh())

    def forward(self, x):
        skips = []
        for name, *_ in _DOWNS:
            x = getattr(self, name)(x)
            skips.append(x)
        x = skips.pop()
        for name, *_ in _UPS:
            x = getattr(self, name)(x, skips.pop())
        return self.final(x)

    def set_mask_fn(self, fn):
        for m in self.modules():
            if isinstance(m, MaskedDropout):
                m.mask_fn = fn


class Discriminator1(nn.Module):
    """P16:182-211: cat(img_A, img_B) -> 4 x [spectral_norm(Conv2d k4 s1 p1, bias) -> LeakyReLU(0.2) -> BlurPool(s2)]
    -> ZeroPad2d((1,0,1,0)) -> Conv2d(512,1,4,p1,no bias)."""

    def __init__(self, img_shape):
        super().__init__()
        widths = (img_shape[0] * 2, 64, 128, 256, 512)
        seq = []
        for cin, cout in zip(widths[:-1], widths[1:]):
            seq += [torch.nn.utils.parametrizations.spectral_norm(nn.Conv2d(cin, cout, 4, stride=1, padding=1)),
                    nn.LeakyReLU(0.2), BlurPool(cout, stride=2)]
        seq += [nn.ZeroPad2d((1, 0, 1, 0)), nn.Conv2d(512, 1, 4, padding=1, bias=False)]
        self.model = nn.Sequential(*seq)

    def forward(self, img_A, img_B):
        return self.model(torch.cat((img_A, img_B), dim=1))


def weights_init_normal(m):
    """P16:218-224."""
    name = type(m).__name__
    if "Conv" in name:
        nn.init.normal_(m.weight.data, 0.0, 0.02)
    elif "BatchNorm2d" in name:
        nn.init.normal_(m.weight.data, 1.0, 0.02)
        nn.init.constant_(m.bias.data, 0.0)


def spectral_norm_step(W, u, v, power_iter=True, eps=1e-12):
    """torch.nn.utils.parametrizations._SpectralNorm (used at P16:188): one power iteration u <- normalize(W v),
    v <- normalize(W^T u), then sigma = u . (W v). W: [R,K]. Returns (u, v, sigma)."""
    W = W.reshape(W.shape[0], -1)
    if power_iter:
        u = F.normalize(W @ v, dim=0, eps=eps)
        v = F.normalize(W.t() @ u, dim=0, eps=eps)
    return u, v, torch.dot(u, W @ v)


# ---------------------------------------------------------------------------------------------------------------
# 16 patches + triplet head
# ---------------------------------------------------------------------------------------------------------------
def make_16_patches(B):
    """P16:227-253: patch k (0-based, row-major 4x4 grid) = B[:, :, 64*(k//4):+64, 64*(k%4):+64] (views)."""
    return tuple(B[:, :, 64 * (k // 4):64 * (k // 4) + 64, 64 * (k % 4):64 * (k % 4) + 64] for k in range(16))


def patch_triplet_loss(fake_B, real_B, neg_idx):
    """P16:558-583 with the 16 random indices given explicitly: (1/16) sum_k TripletMarginLoss(margin=1,p=2)(fake_k, B_k, B_{r_k}).
    F.triplet_margin_loss reduces ||x - y + 1e-6||_2 over the LAST dim only (64-pixel rows) and averages the rest."""
    fp, rp = make_16_patches(fake_B), make_16_patches(real_B)
    total = 0.0
    for k in range(16):
        total = total + F.triplet_margin_loss(fp[k], rp[k], rp[int(neg_idx[k])], margin=1.0, p=2)
    return total / 16.0


# ---------------------------------------------------------------------------------------------------------------
# FFT head (forward-only: the reference detaches through PIL / numpy)
# ---------------------------------------------------------------------------------------------------------------
def to_pil_uint8(t):
    """torchvision ToPILImage on a float CHW tensor (P16:263, :300): pic.mul(255).byte() -> HWC uint8 (negative values wrap
    mod 256 after truncation toward zero)."""
    a = (t.detach().to(torch.float32) * 255.0).numpy()
    return (np.trunc(a).astype(np.int64) & 255).astype(np.uint8).transpose(1, 2, 0)


def pil_luma(rgb):
    """PIL Image.convert("L") (P16:300): L = (19595 R + 38470 G + 7471 B + 32768) >> 16 in integer arithmetic."""
    r, g, b = (rgb[..., i].astype(np.int64) for i in range(3))
    return ((19595 * r + 38470 * g + 7471 * b + 32768) >> 16).astype(np.uint8)


def spectrum_components(luma):
    """FFT_Components.make_components (P16:276-282): rfft2 (float64) -> fftshift(all axes) -> abs, arctan2(imag, real)."""
    f = np.fft.fftshift(np.fft.rfft2(luma))
    return np.abs(f), np.arctan2(f.imag, f.real)


def make_spectra(luma):
    """FFT_Components.make_spectra (P16:284-289): log|fftshift(fft2(uint8 luma))| (float64), the FULL S x S magnitude spectrum."""
    return np.log(np.abs(np.fft.fftshift(np.fft.fft2(luma))))


def sample_spectra(thermal_tensor):
    """P16:378-388: per sample ToPILImage -> L -> make_spectra -> torch.Tensor (float32); [N,1,H,W]."""
    out = [torch.tensor(make_spectra(pil_luma(to_pil_uint8(thermal_tensor[t]))), dtype=torch.float32) for t in range(thermal_tensor.shape[0])]
    return torch.stack(out)[:, None]


def fft_components(thermal_tensor, patch=True):
    """P16:293-319 / G16:294-313: per sample ToPILImage -> L -> components -> float32; [N,1,S,S//2+1]."""
    amps, phas = [], []
    for t in range(thermal_tensor.shape[0]):
        a, p = spectrum_components(pil_luma(to_pil_uint8(thermal_tensor[t])))
        amps.append(torch.tensor(a, dtype=torch.float32))
        phas.append(torch.tensor(p, dtype=torch.float32))
    return torch.stack(amps)[:, None], torch.stack(phas)[:, None]


def calculate_ffts(fake_patches, real_patches):
    """P16:323-375: loss_Amp = (1/16) sum_k L1mean(A_k^fake, A_k^real), same for phase; loss_FFT = (loss_Amp + loss_Pha)/2."""
    la = lp = 0.0
    for fk, rk in zip(fake_patches, real_patches):
        af, pf = fft_components(fk)
        ar, pr = fft_components(rk)
        la = la + F.l1_loss(af, ar)
        lp = lp + F.l1_loss(pf, pr)
    la, lp = la / 16.0, lp / 16.0
    return 0.5 * (la + lp), la, lp


def patch_fft_loss(fake_B, real_B):
    return calculate_ffts(make_16_patches(fake_B), make_16_patches(real_B))


def global_fft_loss(fake_B, real_B):
    """G16:524-529: loss_FFT = 0.5 * (L1(amp) + L1(phase)) on the whole 256x256 image (rfft2 -> 256x129)."""
    af, pf = fft_components(fake_B, patch=False)
    ar, pr = fft_components(real_B, patch=False)
    la, lp = F.l1_loss(af, ar), F.l1_loss(pf, pr)
    return 0.5 * (la + lp), la, lp


def mse_spec(real_gray, fake_gray):
    """TFC-GAN-FFT/Devcom_MagMSE.py:91-118: mean squared error of log|fftshift(fft2(.))| between two float32 gray images."""
    mag = lambda im: np.log(np.abs(np.fft.fftshift(np.fft.fft2(np.asarray(im, dtype=np.float32)))))  # noqa: E731
    return float(np.mean((mag(real_gray) - mag(fake_gray)) ** 2))


def other_spec(real_gray, fake_gray):
    """TFC-GAN-FFT/eval/Eurecom/Eurecom_MagOther.py:90-118: sklearn mean_absolute_error of the same two log-magnitude spectra (uniform average over the
    columns of equal length = the mean absolute difference)."""
    mag = lambda im: np.log(np.abs(np.fft.fftshift(np.fft.fft2(np.asarray(im, dtype=np.float32)))))  # noqa: E731
    return float(np.mean(np.abs(mag(real_gray) - mag(fake_gray))))


# ---------------------------------------------------------------------------------------------------------------
# temperature head (forward-only; P16:255-268, :587-595 over datasets_temp.py:14-35)
# ---------------------------------------------------------------------------------------------------------------
TEMP_T = np.linspace(24, 38, num=256)       # P16:256 / datasets_temp.py:43: Celsius per uint8 code


def vectorize_temps(fake_B):
    """P16:260-268: per sample ToPILImage(...).convert("RGB") -> red channel (datasets_temp.py:33) -> dict lookup
    (np.searchsorted over the sorted keys 0..255 = plain indexing) -> torch.Tensor (float32); [N,1,H,W]."""
    out = [torch.tensor(TEMP_T[to_pil_uint8(fake_B[t])[:, :, 0]], dtype=torch.float32) for t in range(fake_B.shape[0])]
    return torch.stack(out)[:, None]


def temp_triplet_loss(fake_B, TB, B_tf, lambda_t=10.0):
    """P16:587-595: nn.TripletMarginLoss(margin=1, p=2)(TFB_, TB[N,1,H,W], TBTF) * lambda_t."""
    tb = TB.reshape(TB.size(0), 1, TB.size(-2), TB.size(-1)).to(torch.float32)
    return F.triplet_margin_loss(vectorize_temps(fake_B), tb, vectorize_temps(B_tf), margin=1.0, p=2) * lambda_t


def temp_head_case(seed=71, n=2):
    """Seeded inputs for the temperature-head fixture: fake_B (with negatives -> uint8 wrap), dataset temperatures TB near the fake
    ones, and a perturbed negative source, so that the hinge is active on part of the rows."""
    fake, _ = synthetic_pairs(n, seed=seed)
    fake = torch.tanh(fake * 1.5)
    rng = np.random.default_rng(seed + 1)
    neg_src = torch.tanh(fake * 1.5 + torch.from_numpy(rng.standard_normal(tuple(fake.shape)).astype(np.float32)) * 0.35)
    TB = vectorize_temps(fake)[:, 0] + torch.from_numpy(rng.standard_normal((n, 256, 256)).astype(np.float32)) * 4.0
    return fake, TB, neg_src


def color_jitter_thermal(real_B, params):
    """torchvision ColorJitter restricted to R=G=B inputs, explicit parameters (parity unpinned: torchvision absent). Mirrors
    the product helper so the negatives of the temperature term can be regenerated on both sides."""
    x = real_B.float()
    for op in params["order"]:
        if op == 0:
            x = (x * params["brightness"]).clamp(0.0, 1.0)
        elif op == 1:
            f = params["contrast"]
            mean = (0.9999 * x[:, :1]).mean(dim=(1, 2, 3), keepdim=True)
            x = (f * x + (1.0 - f) * mean).clamp(0.0, 1.0)
        elif op == 2:
            f = params["saturation"]
            x = (f * x + (1.0 - f) * 0.9999 * x).clamp(0.0, 1.0)
    return x


# ---------------------------------------------------------------------------------------------------------------
# adversarial losses, optimiser, the step
# ---------------------------------------------------------------------------------------------------------------
# ---------------------------------------------------------------------------------------------------------------
# STN21 configuration (TFC-STN/TFCGAN_STN21_Original_NewModel3_Official.py, "STN"): the pieces the script takes from kornia, restated.
# kornia is absent from the image, so these follow kornia's published behaviour (PARITY UNPINNED); the composition around them
# (morph_triplet, the loss functions, Net.forward, the step) is lifted from the script itself by tests/golden/make_golden.py.
# ---------------------------------------------------------------------------------------------------------------
def morph_gradient(x):
    """kornia.morphology.gradient(x, kernel = 3 x 3 cross) (STN:444-449) = dilation - erosion with geodesic borders: the maximum / minimum over the
    in-image neighbours {centre, up, down, left, right} (a neighbour outside the image never wins)."""
    def shifts(fill):
        pad = F.pad(x, (1, 1, 1, 1), value=fill)
        h, w = x.shape[-2:]
        return torch.stack((pad[..., 1:h + 1, 1:w + 1], pad[..., 0:h, 1:w + 1], pad[..., 2:h + 2, 1:w + 1], pad[..., 1:h + 1, 0:w], pad[..., 1:h + 1, 2:w + 2]))
    return shifts(float("-inf")).max(0).values - shifts(float("inf")).min(0).values


def morph_triplet(real_A, real_B, reg_B):
    """STN:444-459 with criterion_morph = nn.TripletMarginLoss(margin=1.0, p=2) (STN:99): anchor = 1 - grad(reg_B), positive = 1 - grad(real_A),
    negative = 1 - grad(real_B)"""
    crit = nn.TripletMarginLoss(margin=1.0, p=2)
    return crit(1.0 - morph_gradient(reg_B), 1.0 - morph_gradient(real_A), 1.0 - morph_gradient(real_B))


def loss_gan_generator(pred_fake, real_pred, valid=0.9):
    """P16:554: BCEWithLogits(pred_fake - real_pred.detach(), 0.9)."""
    x = pred_fake - real_pred.detach()
    return F.binary_cross_entropy_with_logits(x, torch.full_like(x, valid))


def loss_discriminator(pred_real, pred_fake, valid=0.9):
    """P16:628-630."""
    a = F.binary_cross_entropy_with_logits(pred_real - pred_fake, torch.full_like(pred_real, valid))
    b = F.binary_cross_entropy_with_logits(pred_fake - pred_real, torch.zeros_like(pred_real))
    return 0.5 * (a + b)


def adam_step(p, g, m, v, step, lr=2e-4, b1=0.5, b2=0.999, eps=1e-8):
    """torch.optim.Adam (P16:461-462) on raw tensors; returns updated (p, m, v)."""
    m = b1 * m + (1 - b1) * g
    v = b2 * v + (1 - b2) * g * g
    denom = v.sqrt() / np.sqrt(1 - b2 ** step) + eps
    return p - (lr / (1 - b1 ** step)) * m / denom, m, v


class TrainStep:
    """P16:545-638 without LPIPS (:598) and the temperature head (:587-595), bf16-free, no GradScaler:
       loss_G = 0.5*GAN + triplet + 0.01*FFT;  loss_D = 0.5*(real + fake)."""

    def __init__(self, generator, discriminator, lr=2e-4, b1=0.5, b2=0.999, fft_mode="patch"):
        self.G, self.D = generator, discriminator
        self.opt_G = torch.optim.Adam(generator.parameters(), lr=lr, betas=(b1, b2))
        self.opt_D = torch.optim.Adam(discriminator.parameters(), lr=lr, betas=(b1, b2))
        self.fft_mode = fft_mode

    def step(self, real_A, real_B, neg_idx):
        self.opt_G.zero_grad()
        fake_B = self.G(real_A)
        pred_fake = self.D(fake_B, real_A)
        real_pred = self.D(real_B, real_A)
        loss_gan = loss_gan_generator(pred_fake, real_pred)
        loss_trip = patch_triplet_loss(fake_B, real_B, neg_idx)
        with torch.no_grad():
            loss_fft, la, lp = (patch_fft_loss if self.fft_mode == "patch" else global_fft_loss)(fake_B, real_B)
        loss_G = 0.5 * loss_gan + loss_trip + 0.01 * loss_fft
        loss_G.backward()
        self.opt_G.step()
        self.opt_D.zero_grad()
        pred_real = self.D(real_B, real_A)
        pred_fake = self.D(fake_B.detach(), real_A)
        loss_D = loss_discriminator(pred_real, pred_fake)
        loss_D.backward()
        self.opt_D.step()
        return {"loss_G": loss_G.detach(), "loss_GAN_g": loss_gan.detach(), "loss_triplet_patch": loss_trip.detach(),
                "loss_FFT": loss_fft, "loss_Amp": la, "loss_Pha": lp, "loss_D": loss_D.detach(), "fake_B": fake_B.detach()}


# ---------------------------------------------------------------------------------------------------------------
# bf16-STORAGE mode of the step: the same algorithm with a round-to-bf16 at exactly the points where the HIP engine stores a bf16
# tensor (tfc-gan_amd/nets.py), fp32 arithmetic in between -- what "bf16 storage, fp32 accumulate" means.  Against the plain fp32
# oracle a bf16 run differs by ~20 % in generator gradients (activation masks flip on bf16-rounded pre-activations); against THIS
# mode only accumulation order and rare 1-ulp rounding ties are left, so the bf16 parity test can be tight.
#   forward rounding points : inputs (cat(A,B) / A as bf16 NHWC), packed weights, every conv / convT output, every pooled / blurred /
#                             normalised tensor, the PatchGAN logits; NOT the generator output (fp32 NCHW from the head kernel)
#   backward rounding points: every gradient tensor the engine stores in bf16 = the gradient AT each forward rounding point of an
#                             activation (weight gradients and the gradient w.r.t. fake_B are fp32)
# ---------------------------------------------------------------------------------------------------------------
def _bf(x):
    return x.to(torch.bfloat16).to(torch.float32)


class _RoundBoth(torch.autograd.Function):                       # stored activation: value and its gradient are bf16 tensors
    @staticmethod
    def forward(ctx, x):
        return _bf(x)

    @staticmethod
    def backward(ctx, g):
        return _bf(g)


class _RoundFwd(torch.autograd.Function):                        # bf16 operand whose gradient stays fp32 (weights, fake_B into D)
    @staticmethod
    def forward(ctx, x):
        return _bf(x)

    @staticmethod
    def backward(ctx, g):
        return g


class _RoundBwd(torch.autograd.Function):                        # fp32 value whose gradient is stored in bf16
    @staticmethod
    def forward(ctx, x):
        return x.clone()

    @staticmethod
    def backward(ctx, g):
        return _bf(g)


def _blur(x, stride):
    c = x.shape[1]
    k = torch.tensor(BLUR_TAPS)
    filt = (torch.outer(k, k) / 64.0)[None, None].repeat(c, 1, 1, 1)
    return F.conv2d(F.pad(x, (1, 2, 1, 2), mode="reflect"), filt, stride=stride, groups=c)


def generator_forward_bf16(G, x):
    """GeneratorUNet.forward (P16:160-174) with the engine's bf16 storage points (nets.GeneratorCore.forward); eval-mode dropout"""
    rb, rw = _RoundBoth.apply, _RoundFwd.apply
    h = _bf(x)
    skips = []
    for name, _, _, norm, _ in _DOWNS:
        w = getattr(G, name).model[0].weight
        z = rb(F.conv2d(h, rw(w), padding=1))                       # raw conv output (bf16), statistics of the stored values
        if norm:
            z = F.instance_norm(z, eps=1e-5)
        h = rb(_blur(F.leaky_relu(z, 0.2), 2))                      # pooled output (bf16): next input and skip window
        skips.append(h)
    h = skips.pop()
    for name, *_ in _UPS:
        w = getattr(G, name).model[0].weight
        z = rb(F.conv_transpose2d(h, rw(w), stride=2, padding=1))   # rawT
        z = rb(_blur(z, 1))                                         # blur
        z = rb(F.relu(F.instance_norm(z, eps=1e-5)))                # up-path window of the concat buffer
        # the skip window's gradient is the convT dgrad (stored bf16) PLUS the down-path conv dgrad accumulated on top of it
        h = torch.cat((z, _RoundBwd.apply(skips.pop())), dim=1)
    conv = G.final[2]
    u = F.pad(F.interpolate(h, scale_factor=2), (1, 0, 1, 0))
    pre = _RoundBwd.apply(F.conv2d(u, rw(conv.weight), conv.bias, padding=1))   # d(pre-tanh) is stored bf16 (tfc_tanh_bwd_pack)
    return torch.tanh(pre)


def discriminator_forward_bf16(D, img_a, img_b, state, power_iter=True):
    """Discriminator1.forward (P16:205-211) with the engine's storage points; `state`: list of [u, v] per SN block, advanced in place
    (one power iteration per forward in training mode, as torch.nn.utils.parametrizations.spectral_norm does)"""
    rb, rw = _RoundBoth.apply, _RoundFwd.apply
    h = torch.cat((rw(img_a), _bf(img_b)), dim=1)
    bi = 0
    for m in D.model:
        if isinstance(m, nn.Conv2d) and hasattr(m, "parametrizations"):
            W = m.parametrizations.weight.original
            u, v = state[bi]
            with torch.no_grad():
                u, v, _ = spectral_norm_step(W.detach(), u, v, power_iter=power_iter)
            state[bi] = [u, v]
            sigma = torch.dot(u, W.flatten(1) @ v)
            z = _RoundBwd.apply(F.conv2d(h, rw(W), padding=1) / sigma + m.bias.view(1, -1, 1, 1))   # d z is stored bf16; z itself never is
            h = _RoundFwd.apply(F.leaky_relu(z, 0.2))               # the conv epilogue stores the ACTIVATED tensor
            h = rb(_blur(h, 2))
            bi += 1
    head = D.model[-1]
    return rb(F.conv2d(F.pad(h, (1, 0, 1, 0)), head.weight, padding=1))   # the head kernel reads fp32 weights; logits stored bf16


def bf16_storage_step(G, D, real_A, real_B, neg_idx, fft_mode="patch"):
    """one step (P16:545-638 minus LPIPS / temperature) in bf16-storage arithmetic WITHOUT the optimiser update: returns the losses, fake_B and
    the parameter gradients of both networks (dicts by state_dict key)"""
    state = [[m.parametrizations.weight[0]._u.clone(), m.parametrizations.weight[0]._v.clone()] for m in D.model
             if isinstance(m, nn.Conv2d) and hasattr(m, "parametrizations")]
    gp = dict(G.named_parameters())
    dp = dict(D.named_parameters())
    fake = generator_forward_bf16(G, real_A)
    pf = discriminator_forward_bf16(D, fake, real_A, state)
    pr = discriminator_forward_bf16(D, real_B, real_A, state)
    l_gan = loss_gan_generator(pf, pr)
    l_trip = patch_triplet_loss(fake, real_B, neg_idx)
    with torch.no_grad():
        l_fft, la, lp = (patch_fft_loss if fft_mode == "patch" else global_fft_loss)(fake, real_B)
    l_G = 0.5 * l_gan + l_trip + 0.01 * l_fft
    gg = torch.autograd.grad(l_G, list(gp.values()), allow_unused=True)
    pr2 = discriminator_forward_bf16(D, real_B, real_A, state)
    pf2 = discriminator_forward_bf16(D, fake.detach(), real_A, state)
    l_D = loss_discriminator(pr2, pf2)
    dg = torch.autograd.grad(l_D, list(dp.values()), allow_unused=True)
    return {"loss_G": l_G.detach(), "loss_GAN_g": l_gan.detach(), "loss_triplet_patch": l_trip.detach(), "loss_FFT": l_fft, "loss_D": l_D.detach(),
            "fake_B": fake.detach(), "g_grads": {k: g for k, g in zip(gp, gg)}, "d_grads": {k: g for k, g in zip(dp, dg)}}


# ---------------------------------------------------------------------------------------------------------------
# portable synthetic inputs / weights -- regenerated identically on both sides of a parity test
# ---------------------------------------------------------------------------------------------------------------
def synthetic_pairs(n, seed=1234, size=256):
    """A (visible): uint8 U{0..255} per channel; B (thermal): one uint8 channel replicated to 3 (R=G=B, datasets_temp.py:33);
    both mapped to [-1,1] by x/127.5 - 1 (ToTensor + Normalize(.5,.5), P16:479-482)."""
    rng = np.random.default_rng(seed)
    a = rng.integers(0, 256, size=(n, 3, size, size), dtype=np.uint8)
    b = np.repeat(rng.integers(0, 256, size=(n, 1, size, size), dtype=np.uint8), 3, axis=1)
    f = lambda u: torch.from_numpy(u.astype(np.float32) / 127.5 - 1.0)  # noqa: E731
    return f(a), f(b)


def init_weights_portable(module, seed=0, std=0.02):
    """Deterministic N(0, std) weights from a numpy stream, in state_dict order; spectral-norm u/v get unit random vectors."""
    rng = np.random.default_rng(seed)
    sd = module.state_dict()
    with torch.no_grad():
        for k, t in sd.items():
            if k.endswith("filt"):
                continue
            x = torch.from_numpy(rng.standard_normal(tuple(t.shape)).astype(np.float32))
            if k.endswith("._u") or k.endswith("._v"):
                x = x / x.norm()
            else:
                x = x * std
            t.copy_(x)
    return module


# ---------------------------------------------------------------------------------------------------------------
# the dropout mask generator of the HIP kernels, restated in numpy (csrc/common.h tfc_hash32 / tfc_keep) so CPU tests
# can hand the oracle the very mask a GPU run uses
# ---------------------------------------------------------------------------------------------------------------
def hip_keep_mask(seed, n, drop_p):
    idx = np.arange(n, dtype=np.uint64)
    M = np.uint64(0xFFFFFFFF)
    seed = np.uint64(seed & 0xFFFFFFFF)
    x = (idx * np.uint64(0x9E3779B1) + seed) & M
    x ^= x >> np.uint64(16); x = (x * np.uint64(0x85EBCA6B)) & M
    x ^= x >> np.uint64(13); x = (x * np.uint64(0xC2B2AE35)) & M
    x ^= x >> np.uint64(16)
    x = (x + ((seed * np.uint64(0x27D4EB2F)) & M)) & M
    x ^= x >> np.uint64(15); x = (x * np.uint64(0x2C1B3C6D)) & M
    x ^= x >> np.uint64(12)
    thresh = np.uint64(int(round(drop_p * 16777216.0)))
    return (x >> np.uint64(8)) >= thresh


def hip_mask_fn(step_seed):
    """mask_fn for GeneratorUNet.set_mask_fn matching tfc-gan_amd/nets.py: per-layer seed = step_seed*64 + i (down_i, i =
    0..5) or + 16 + j (up_j), element index over the layer output in NHWC order."""
    layer = {"down1": 0, "down2": 1, "down3": 2, "down4": 3, "down5": 4, "down6": 5, "up1": 16, "up2": 17, "up3": 18, "up4": 19, "up5": 20}

    def fn(tag, shape):
        n, c, h, w = shape
        keep = hip_keep_mask(step_seed * 64 + layer[tag], n * c * h * w, 0.5).reshape(n, h, w, c)
        return torch.from_numpy(np.ascontiguousarray(keep.transpose(0, 3, 1, 2)))
    return fn
