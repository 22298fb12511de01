"""Loss heads of the PATCH-16 step with the reference's call surface (TFCGAN_multigpu_patchFFT_16P.py):

  make_16_patches(B)                     :227-253   16 views, patch k -> rows 64*(k//4).., cols 64*(k%4)..
  ContrastiveLoss / patch_triplet_loss   :75, :558-583   16x nn.TripletMarginLoss(margin=1, p=2) with given negatives
  FFT_Components / fft_components / calculate_ffts   :271-375
  relativistic BCE                       :554, :628-630 (engine.py uses the fused kernel directly)

The reference has no class called ContrastiveLoss: the "16-patch contrastive head" named by the project brief IS the
16-fold triplet mean above; `ContrastiveLoss` here is defined as exactly that.
"""
import numpy as np
import torch
import torch.nn as nn

from . import ops

PATCH = 64
GRID = 4


def make_16_patches(B):
    """16 zero-copy views of B[N,C,256,256], row-major 4x4 grid of 64x64 patches (reference order B1..B16)."""
    assert B.shape[-1] == 256 and B.shape[-2] == 256, "the reference hard-codes offsets 64/128/192 (256x256 images only)"
    return tuple(B[:, :, 64 * (k // GRID):64 * (k // GRID) + PATCH, 64 * (k % GRID):64 * (k % GRID) + PATCH] for k in range(16))


def patch_first_flat_index(k, width=256):
    """flat NCHW offset (within one channel plane) of the first element of patch k: 0,64,128,192,16384,..."""
    return 64 * (k // GRID) * width + 64 * (k % GRID)


class _Triplet16Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, fake, real, neg_idx):
        loss, dfake = ops.patch16_triplet(fake.detach(), real.detach(), neg_idx, want_grad=fake.requires_grad)
        ctx.save_for_backward(dfake) if dfake is not None else None
        ctx.has = dfake is not None
        ctx.in_dtype = fake.dtype
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        if not ctx.has:
            return None, None, None
        (dfake,) = ctx.saved_tensors
        return (dfake * g).to(ctx.in_dtype), None, None


def patch_triplet_loss(fake_B, real_B, neg_idx):
    """(1/16) * sum_k TripletMarginLoss(fake patch k, real patch k, real patch neg_idx[k]) -- one fused kernel."""
    return _Triplet16Fn.apply(fake_B, real_B, [int(i) for i in neg_idx])


class ContrastiveLoss(nn.Module):
    """The 16-patch triplet head. Two call forms:
         loss(fake_B, real_B, neg_idx=None)          whole images [N,C,256,256]; neg_idx: 16 ints (drawn with
                                                     np.random.randint(16) per patch like the reference when None)
         loss(anchor, positive, negative)            three [N,C,64,64] patch tensors = nn.TripletMarginLoss(margin=1, p=2)
                                                     on one patch (composed from the same kernel by embedding the patch)."""

    def __init__(self, margin=1.0, p=2):
        super().__init__()
        assert margin == 1.0 and p == 2, "the reference uses margin=1.0, p=2 (:75)"

    def forward(self, a, b, c=None):
        if torch.is_tensor(c):
            return _single_patch_triplet(a, b, c)
        if c is None:
            c = [int(np.random.randint(16, size=1).item()) for _ in range(16)]
        return patch_triplet_loss(a, b, c)


def _single_patch_triplet(anchor, positive, negative):
    """nn.TripletMarginLoss on one 64x64 patch through the 16-patch kernel: put anchor / positive in patch 0 and the negative
    in patch 1 of zero images; patches 1..15 use themselves as negatives and contribute exactly 1.0 each."""
    N, C = anchor.shape[:2]
    dev = anchor.device
    fake = torch.zeros((N, C, 256, 256), dtype=torch.float32, device=dev)
    real = torch.zeros_like(fake)
    fake[:, :, :64, :64] = anchor.float()
    real[:, :, :64, :64] = positive.detach().float()
    real[:, :, :64, 64:128] = negative.detach().float()
    fake = fake.detach().requires_grad_(anchor.requires_grad)
    neg = [1] + list(range(1, 16))
    total = patch_triplet_loss(fake, real, neg)
    return total * 16.0 - 15.0


class FFT_Components(object):
    """reference :271-289. `image`: 2-D uint8 array-like (a PIL "L" image in the reference) or a [1|3,H,W] tensor in [-1,1]."""

    def __init__(self, image):
        self.image = image

    def _as_tensor(self):
        img = self.image
        if torch.is_tensor(img):
            return img
        arr = np.asarray(img).astype(np.float32)                  # uint8 luma -> value whose *255 truncation gives it back
        t = torch.from_numpy((arr + 0.5) / 255.0)[None]
        return t

    def make_components(self):
        t = self._as_tensor().cuda()
        S = t.shape[-1]
        amp, pha = ops.fft_spectrum(t[None].expand(1, 3 if t.shape[0] == 3 else 1, S, S).contiguous(), S, 1, 1, shift=True)
        return amp[0], pha[0]

    def make_spectra(self):
        """reference :284-289: log|fftshift(fft2(image))|, the full S x S magnitude spectrum (S = 64 or 256), fp32 on the GPU."""
        t = self._as_tensor().cuda()
        S = t.shape[-1]
        return _full_log_spectrum(t[None].expand(1, 3 if t.shape[0] == 3 else 1, S, S).contiguous(), S)[0]


def _full_log_spectrum(img, S):
    """log-magnitude of the fftshifted FULL spectrum from the half spectrum of tfc_fft_spectrum: for real input |F[ky][kx]| = |F[-ky][-kx]|, so
    columns S/2+1 .. S-1 are the point reflection of columns S/2-1 .. 1 (pure indexing, no arithmetic beyond the kernel's)."""
    amp, _ = ops.fft_spectrum(img, S, 1, 1, shift=False)          # [N][S][S/2+1], unshifted
    nb = S // 2 + 1
    ky = torch.arange(S, device=amp.device)
    neg_ky = (-ky) % S
    right = amp[:, neg_ky][:, :, 1:nb - 1].flip(-1)               # kx = S/2+1 .. S-1  <-  conj partner (S-ky, S-kx)
    full = torch.cat((amp, right), dim=-1)                        # [N][S][S], kx = 0 .. S-1
    return torch.log(torch.fft.fftshift(full, dim=(-2, -1)))


def sample_spectra(thermal_tensor):
    """reference :378-388: thermal_tensor [N,3,S,S] in [-1,1] -> log-magnitude spectra [N,1,S,S] fp32 (S = 64 or 256) for the sample grids."""
    S = thermal_tensor.shape[-1]
    assert thermal_tensor.shape[-2] == S and S in (64, 256)
    return _full_log_spectrum(thermal_tensor.detach(), S)[:, None]


def fft_components(thermal_tensor, patch=True):
    """reference :293-319. thermal_tensor [N,3,S,S] in [-1,1] -> (AMP, PHA) each [N,1,S,S//2+1] fp32, fftshifted.
    patch=True: S=64 (aspect 33); patch=False: S=256 (129)."""
    S = 64 if patch else 256
    assert thermal_tensor.shape[-1] == S and thermal_tensor.shape[-2] == S
    N = thermal_tensor.shape[0]
    amp, pha = ops.fft_spectrum(thermal_tensor, S, 1, 1, shift=True)
    return amp.reshape(N, 1, S, S // 2 + 1), pha.reshape(N, 1, S, S // 2 + 1)


def patch_fft_loss(fake_B, real_B):
    """loss_FFT of calculate_ffts on whole images: 0.5 * (mean_k L1(amp) + mean_k L1(phase)) over the 16 patches.
    Carries no gradient, exactly like the reference (tensor -> PIL -> numpy round trip, :300-302)."""
    N = fake_B.shape[0]
    af, pf = ops.fft_spectrum(fake_B.detach(), 64, 4, 4, shift=False)
    ar, pr = ops.fft_spectrum(real_B.detach(), 64, 4, 4, shift=False)
    out = torch.zeros(3, dtype=torch.float32, device=fake_B.device)
    scale = 1.0 / (16.0 * N * 64 * 33)
    ops.l1_sum(af, ar, scale, out[0:1])
    ops.l1_sum(pf, pr, scale, out[1:2])
    return 0.5 * (out[0] + out[1]), out[0], out[1]


def global_fft_loss(fake_B, real_B):
    """GLO-16 variant (TFCGAN_multigpu_globalFFT_16P.py:294-313, :524-529): rfft2 of the whole 256x256 image."""
    N = fake_B.shape[0]
    af, pf = ops.fft_spectrum(fake_B.detach(), 256, 1, 1, shift=False)
    ar, pr = ops.fft_spectrum(real_B.detach(), 256, 1, 1, shift=False)
    out = torch.zeros(2, dtype=torch.float32, device=fake_B.device)
    scale = 1.0 / (N * 256 * 129)
    ops.l1_sum(af, ar, scale, out[0:1])
    ops.l1_sum(pf, pr, scale, out[1:2])
    return 0.5 * (out[0] + out[1]), out[0], out[1]


def calculate_ffts(*patches):
    """reference :323-375: calculate_ffts(fake_B1..fake_B16, B1..B16) -> loss_FFT (scalar, no gradient)."""
    assert len(patches) == 32, "expects 16 fake patches followed by 16 real patches"
    dev = patches[0].device
    N = patches[0].shape[0]
    out = torch.zeros(2, dtype=torch.float32, device=dev)
    scale = 1.0 / (16.0 * N * 64 * 33)
    for k in range(16):
        af, pf = ops.fft_spectrum(patches[k].detach(), 64, 1, 1, shift=True)
        ar, pr = ops.fft_spectrum(patches[16 + k].detach(), 64, 1, 1, shift=True)
        ops.l1_sum(af, ar, scale, out[0:1])
        ops.l1_sum(pf, pr, scale, out[1:2])
    return 0.5 * (out[0] + out[1])


# ---- temperature head (rank 1 of SURVEY.md section 8f; reference :255-268, :587-595) -------------------------------------------
T = np.linspace(24, 38, num=256)                      # reference :256 -- Celsius per uint8 code
_LUT = {}


def _temp_lut(dev):
    if dev not in _LUT:
        _LUT[dev] = torch.from_numpy(T.astype(np.float32)).to(dev)
    return _LUT[dev]


def vectorize_temps(fake_B):
    """reference :260-268: per sample ToPILImage -> red channel uint8 -> temperature LUT; returns [N,1,H,W] fp32 (no gradient)."""
    t = ops.vectorize_temps(fake_B.detach(), _temp_lut(fake_B.device))
    return t.reshape(t.shape[0], 1, t.shape[1], t.shape[2])


def temperature_triplet_loss(fake_B, TB, B_tf, lambda_t=10.0):
    """loss_temp_g of reference :587-595: criterion_temp(vectorize_temps(fake_B), TB, vectorize_temps(B_tf)) * lambda_t.
    TB: [N,H,W] (or [N,1,H,W]) ground-truth temperatures from the dataset (datasets_temp.py:66-67); B_tf: the augmented real_B
    that serves as negative (the reference draws it with torchvision ColorJitter, :591-592 -- see color_jitter_thermal)."""
    tfb = ops.vectorize_temps(fake_B.detach(), _temp_lut(fake_B.device))
    tneg = ops.vectorize_temps(B_tf.detach(), _temp_lut(fake_B.device))
    tb = TB.reshape(tfb.shape).to(tfb.device, torch.float32)
    return ops.row_triplet(tfb, tb, tneg, 1.0).reshape(()) * lambda_t


def color_jitter_params(rng, brightness=0.5, contrast=0.75, saturation=1.5, hue=0.5):
    """The random draw of transforms.ColorJitter(brightness=0.5, contrast=0.75, saturation=1.5, hue=0.5) (reference :591) as
    explicit values: op order (permutation of 0..3 = brightness, contrast, saturation, hue) and the four factors."""
    return {"order": [int(i) for i in rng.permutation(4)],
            "brightness": float(rng.uniform(max(0.0, 1 - brightness), 1 + brightness)),
            "contrast": float(rng.uniform(max(0.0, 1 - contrast), 1 + contrast)),
            "saturation": float(rng.uniform(max(0.0, 1 - saturation), 1 + saturation)),
            "hue": float(rng.uniform(-hue, hue))}


def color_jitter_thermal(real_B, params):
    """ColorJitter restricted to thermal images (R = G = B, datasets_temp.py:33): brightness and contrast are blends clamped to
    [0,1]; saturation blends with the luma 0.2989R+0.587G+0.114B (= 0.9999 v on grey); hue leaves grey pixels unchanged.
    torchvision is not installed here, so this restates its published tensor semantics -- PARITY UNPINNED; it only prepares
    the negatives of the (gradient-free) temperature term and is plain elementwise torch, not a kernel."""
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


def other_spec(real_gray, fake_gray):
    """Evaluation metric of TFC-GAN-FFT/eval/Eurecom/Eurecom_MagOther.py:90-118: per image pair sklearn mean_absolute_error(log|fftshift(fft2(real))|,
    log|fftshift(fft2(fake))|) (= the mean absolute difference over the whole spectrum). Same inputs and return convention as mse_spec."""
    return mse_spec(real_gray, fake_gray, absolute=True)


def mse_spec(real_gray, fake_gray, absolute=False):
    """Evaluation metric of TFC-GAN-FFT/Devcom_MagMSE.py:91-118: per image pair MSE(log|fftshift(fft2(real))|, log|fftshift(fft2(fake))|).
    real_gray / fake_gray: uint8 grayscale images [N,256,256] (tensor or array, as cv2.imread(path, 0) yields). Returns [N] fp32."""
    def prep(g):
        t = torch.as_tensor(np.asarray(g) if not torch.is_tensor(g) else g)
        assert t.dtype == torch.uint8 and t.shape[-2:] == (256, 256), "mse_spec expects uint8 [N,256,256] gray images"
        t = t.reshape(-1, 1, 256, 256).to("cuda", non_blocking=True).float()
        return (t + 0.5) / 255.0                      # the spectrum kernel truncates x*255 back to the same uint8
    a, _ = ops.fft_spectrum(prep(real_gray), 256, 1, 1, shift=False)
    b, _ = ops.fft_spectrum(prep(fake_gray), 256, 1, 1, shift=False)
    return ops.logmag_mse(a, b, absolute=absolute)
