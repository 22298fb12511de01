"""The LPIPS term of loss_G (reference TFCGAN_multigpu_patchFFT_16P.py:17, :69-73, :434, :598, :607) on the HIP kernels of this package.

    criterion_lpips = LPIPS(net_type='vgg', version='0.1')          # P16:70-73
    loss_pix_g = criterion_lpips(fake_B, real_B)                    # P16:598
    loss_G = ... + 0.5 * loss_pix_g + ...                           # P16:607

`lpips_pytorch` is a pip dependency that is absent from the reference tree and from this image, and its constructor downloads the
torchvision VGG16 weights and the five linear heads: its published algorithm is restated here -- PARITY UNPINNED (tests compare against
a torch-CPU restatement with random weights). What the package computes:

    x' = (x - mean) / std                        mean = [-.030, -.088, -.188], std = [.458, .448, .450]
    VGG16 `features`, activations after relu1_2, relu2_2, relu3_3, relu4_3, relu5_3
    per tap: f / (sqrt(sum_c f^2) + 1e-10), squared difference, 1x1 conv (no bias) to one channel, mean over H x W  -> [N,1,1,1]
    torch.sum(torch.cat(res, 0), 0, True)        -> [1,1,1,1]: summed over the five taps AND over the batch

The 13 3x3 convolutions run on the gather GEMM (TFC_OP_CONV3, bias + ReLU in its epilogue, bf16 or fp32 as `set_compute_dtype` says);
pooling, the ReLU backward, the input z-score and the heads are the HBM-bound kernels of csrc/lpips.hip. All weights are frozen
(`requires_grad=False` as in the package): the backward produces the gradient w.r.t. the first image only.

Weights never come from the network. `weights=` names local file(s) read with `torch.load(..., weights_only=True)`:
  * a state_dict of this class / of lpips_pytorch.LPIPS   (`net.layers.{k}.weight|bias`, `lin.{i}.1.weight`), or
  * a torchvision vgg16 state_dict                        (`features.{k}.weight|bias`), and / or
  * the original LPIPS linear heads                       (`lin{i}.model.1.weight`)
Without `weights` the module is seeded-random and says so (`self.pretrained is False`): the term is then a smooth perceptual-style
regulariser but NOT the published LPIPS metric.
"""
import warnings

import torch
import torch.nn as nn

from . import models, ops, parallel
from ._lib import EP_RELU, OP_CONV3, TfcError, check

VGG16_CFG = (64, 64, "M", 128, 128, "M", 256, 256, 256, "M", 512, 512, 512, "M", 512, 512, 512)
TAP_AFTER_CONV = (1, 3, 6, 9, 12)                    # relu1_2, relu2_2, relu3_3, relu4_3, relu5_3 (index into the 13 convolutions)
N_CHANNELS = (64, 128, 256, 512, 512)


def _feature_indices():
    """torchvision's vgg16.features numbering: conv, relu, [pool]"""
    idx, k = [], 0
    for c in VGG16_CFG:
        if c == "M":
            k += 1
        else:
            idx.append(k)
            k += 2
    return idx


CONV_INDEX = tuple(_feature_indices())               # (0, 2, 5, 7, 10, 12, 14, 17, 19, 21, 24, 26, 28)


class _Conv(nn.Module):
    def __init__(self, cin, cout):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(cout, cin, 3, 3), requires_grad=False)
        self.bias = nn.Parameter(torch.zeros(cout), requires_grad=False)


class _Vgg(nn.Module):
    def __init__(self):
        super().__init__()
        layers, cin = [nn.Identity() for _ in range(CONV_INDEX[-1] + 1)], 3
        convs = [c for c in VGG16_CFG if c != "M"]
        for k, cout in zip(CONV_INDEX, convs):
            layers[k] = _Conv(cin, cout)
            cin = cout
        self.layers = nn.ModuleList(layers)
        self.register_buffer("mean", torch.tensor([-.030, -.088, -.188]).view(1, 3, 1, 1))
        self.register_buffer("std", torch.tensor([.458, .448, .450]).view(1, 3, 1, 1))

    def convs(self):
        return [self.layers[k] for k in CONV_INDEX]


class _Lin(nn.Module):
    def __init__(self, nc):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(1, nc, 1, 1), requires_grad=False)


class LPIPS(nn.Module):
    """Drop-in for lpips_pytorch.LPIPS as the reference uses it. forward(x, y) -> [1,1,1,1], differentiable w.r.t. x."""

    def __init__(self, net_type="vgg", version="0.1", weights=None, seed=0):
        super().__init__()
        if net_type != "vgg" or version != "0.1":
            raise TfcError(f"LPIPS(net_type={net_type!r}, version={version!r}): the reference path uses net_type='vgg', version='0.1' only (P16:70-73)")
        self.net = _Vgg()
        self.lin = nn.ModuleList([nn.Sequential(nn.Identity(), _Lin(nc)) for nc in N_CHANNELS])
        g = torch.Generator().manual_seed(seed)
        for c in self.net.convs():
            fan_in = c.weight.shape[1] * 9
            c.weight.data = torch.randn(c.weight.shape, generator=g) * (2.0 / fan_in) ** 0.5
        for seq in self.lin:
            seq[1].weight.data = torch.rand(seq[1].weight.shape, generator=g) / seq[1].weight.shape[1]    # the published heads are non-negative
        self.pretrained = False
        self._packed = None
        if weights is not None:
            self.load_pretrained(*([weights] if isinstance(weights, (str, bytes)) or hasattr(weights, "__fspath__") else list(weights)))
        else:
            warnings.warn("LPIPS constructed without local weights: VGG16 and the linear heads are seeded-random, the value is not the "
                          "published LPIPS metric (there is no network access to fetch them; pass weights=<path>)", stacklevel=2)

    # ---- weights ----------------------------------------------------------------------------------------------------
    def load_pretrained(self, *paths):
        """read local checkpoint file(s) with a loader that executes nothing from them; see the module docstring for the accepted layouts"""
        own = self.state_dict()
        got_vgg = got_lin = 0
        for path in paths:
            sd = torch.load(path, map_location="cpu", weights_only=True)
            if isinstance(sd, dict) and "state_dict" in sd and isinstance(sd["state_dict"], dict):
                sd = sd["state_dict"]
            for k, v in sd.items():
                k2 = k[7:] if k.startswith("module.") else k
                if k2.startswith("features."):
                    k2 = "net.layers." + k2[len("features."):]
                elif k2.startswith("lin") and ".model.1." in k2:             # lin0.model.1.weight -> lin.0.1.weight
                    k2 = "lin." + k2[3:k2.index(".")] + ".1." + k2.split(".model.1.")[1]
                if k2 in own:
                    if tuple(own[k2].shape) != tuple(v.shape):
                        raise TfcError(f"{path}: {k} has shape {tuple(v.shape)}, expected {tuple(own[k2].shape)}")
                    own[k2].copy_(v.to(own[k2].dtype))
                    got_vgg += k2.startswith("net.layers.")
                    got_lin += k2.startswith("lin.")
        if got_vgg not in (0, 26) or got_lin not in (0, 5) or got_vgg + got_lin == 0:
            raise TfcError(f"LPIPS weights: found {got_vgg}/26 VGG16 tensors and {got_lin}/5 linear heads in {list(paths)}")
        self.pretrained = got_vgg == 26 and got_lin == 5
        self._packed = None
        return self

    def _apply(self, fn, *a, **k):
        self._packed = None
        return super()._apply(fn, *a, **k)

    def load_state_dict(self, *a, **k):
        self._packed = None
        return super().load_state_dict(*a, **k)

    def _streams(self, dt, dev):
        """operand streams of the 13 convolutions (forward + dgrad), packed once per (dtype, device): the weights are frozen"""
        key = (dt, str(dev), tuple(p._version for p in self.parameters()))
        if self._packed is not None and self._packed[0] == key:
            return self._packed[1]
        cin0 = 64 // (2 if dt == ops.DT_BF16 else 4)                         # the image is padded to one 64-byte channel chunk
        jobs, keep = [], []
        for i, c in enumerate(self.net.convs()):
            cout, cin = c.weight.shape[:2]
            cp = cin0 if i == 0 else cin
            w4 = torch.zeros((cout, cp, 4, 4), dtype=torch.float32, device=dev)   # 3x3 filter in rows / cols 0..2 of the 4x4 slot
            w4[:, :cin, :3, :3] = c.weight.detach().to(dev, torch.float32)
            keep.append(w4)
            jobs.append((OP_CONV3, 0, w4, cp, cout))
            jobs.append((OP_CONV3, 1, w4, cp, cout))
        plan = ops.PackPlan(dt, jobs)
        plan.run()
        layers = []
        for i, c in enumerate(self.net.convs()):
            layers.append({"cin": jobs[2 * i][3], "cout": jobs[2 * i][4], "fwd": plan.streams[2 * i], "dgrad": plan.streams[2 * i + 1],
                           "bias": c.bias.detach().to(dev, torch.float32).contiguous()})
        heads = [seq[1].weight.detach().to(dev, torch.float32).reshape(-1).contiguous() for seq in self.lin]
        shift = self.net.mean.detach().to(dev, torch.float32).reshape(-1).contiguous()
        scale = self.net.std.detach().to(dev, torch.float32).reshape(-1).contiguous()
        packed = {"layers": layers, "heads": heads, "shift": shift, "scale": scale, "cin0": cin0, "plan": plan}
        self._packed = (key, packed)
        return packed

    # ---- compute ----------------------------------------------------------------------------------------------------
    def _features(self, img, P, dt, keep_all):
        """VGG16 features of an fp32 NCHW image. Returns (taps, saved): the five tap activations, and (keep_all) every activation the
        backward needs: a list of ("conv", layer index, input View, output View) / ("pool", input View, output View)."""
        lib, st = ops.lib(), ops.stream_ptr()
        N, C, H, W = img.shape
        x = ops.new_act(N, H, W, P["cin0"], dt, img.device, zero=True)
        check(lib.tfc_lpips_input_fwd(st, dt, ops._p(img), ops._p(P["shift"]), ops._p(P["scale"]), x.ptr, N, C, H, W, x.pitch), "tfc_lpips_input_fwd")
        taps, saved, ci = [], [], 0
        for c in VGG16_CFG:
            if c == "M":
                y = ops.new_act(N, x.H // 2, x.W // 2, x.C, dt, img.device)
                check(lib.tfc_maxpool2_fwd(st, dt, x.ptr, y.ptr, N, x.H, x.W, x.C), "tfc_maxpool2_fwd")
                if keep_all:
                    saved.append(("pool", x, y))
            else:
                L = P["layers"][ci]
                y = ops.new_act(N, x.H, x.W, L["cout"], dt, img.device)
                ops.conv_fwd(dt, OP_CONV3, x, L["cin"], L["cout"], L["fwd"], y, bias=L["bias"], flags=EP_RELU)
                if keep_all:
                    saved.append(("conv", ci, x, y))
                if ci in TAP_AFTER_CONV:
                    taps.append(y)
                ci += 1
            x = y
        return taps, saved

    def value_and_grad(self, x, y, weight=1.0, want_grad=True):
        """(sum over batch and taps [1,1,1,1] * weight, d / d x fp32 NCHW or None). x, y: fp32 NCHW [N,3,H,W] on the GPU, H and W multiples of 16."""
        ops.require_gpu(x, y)
        if x.shape != y.shape or x.dim() != 4 or x.shape[1] != 3 or x.shape[2] % 16 or x.shape[3] % 16:
            raise TfcError(f"LPIPS: expected two [N,3,H,W] images with H, W multiples of 16, got {tuple(x.shape)} and {tuple(y.shape)}")
        dt = ops.dt_of(models.get_compute_dtype())
        dev = x.device
        P = self._streams(dt, dev)
        lib, st = ops.lib(), ops.stream_ptr()
        x32, y32 = x.detach().contiguous().float(), y.detach().contiguous().float()
        N = x32.shape[0]
        taps_y, _ = self._features(y32, P, dt, keep_all=False)
        taps_x, saved = self._features(x32, P, dt, keep_all=want_grad)
        per_img = torch.zeros(N, dtype=torch.float32, device=dev)
        dtaps = []
        for fx, fy, w in zip(taps_x, taps_y, P["heads"]):
            d = ops.new_act(N, fx.H, fx.W, fx.C, dt, dev) if want_grad else None
            check(lib.tfc_lpips_head(st, dt, fx.ptr, fy.ptr, ops._p(w), ops._p(per_img), None if d is None else d.ptr, N, fx.H, fx.W, fx.C,
                                     float(weight)), "tfc_lpips_head")
            dtaps.append(d)
        value = (per_img.sum() * weight).reshape(1, 1, 1, 1)
        if not want_grad:
            return value, None
        del taps_y
        g = None                                                     # gradient w.r.t. the current activation (None above relu5_3)
        for rec in reversed(saved):
            if rec[0] == "pool":
                _, xin, yout = rec
                gin = ops.new_act(N, xin.H, xin.W, xin.C, dt, dev)
                check(lib.tfc_maxpool2_bwd(st, dt, xin.ptr, g.ptr, gin.ptr, N, xin.H, xin.W, xin.C), "tfc_maxpool2_bwd")
                g = gin
                continue
            _, ci, xin, yout = rec
            L = P["layers"][ci]
            head = dtaps[TAP_AFTER_CONV.index(ci)] if ci in TAP_AFTER_CONV else None
            if g is None:
                g, head = head, None
            n = yout.t.numel()
            check(lib.tfc_relu_bwd(st, dt, g.ptr, yout.ptr, None if head is None else head.ptr, g.ptr, n), "tfc_relu_bwd")     # in place
            gin = ops.new_act(N, xin.H, xin.W, xin.pitch, dt, dev)
            ops.conv_dgrad(dt, OP_CONV3, g, N, xin.H, xin.W, L["cin"], L["cout"], L["dgrad"], gin)
            g = gin
        dx = torch.empty_like(x32)
        check(lib.tfc_lpips_input_bwd(st, dt, g.ptr, ops._p(P["scale"]), ops._p(dx), N, 3, x32.shape[2], x32.shape[3], g.pitch, 1.0, 0), "tfc_lpips_input_bwd")
        return value, dx

    def forward(self, x, y):
        return _LpipsFn.apply(x, y, self)

    def as_extra_loss(self, weight=0.5):
        """adapter for TrainStep.step(extra_loss_G=...): `weight` is the 0.5 of P16:607. The package SUMS over the batch, and the reference
        evaluates it on the gathered global batch (nn.DataParallel gathers fake_B on device 0); with one process per GPU and averaged
        gradients the local sum is therefore scaled by the world size, so that value and gradient equal the reference's global-batch sum."""
        def term(fake, real_B):
            w = weight * parallel.world_size()
            value, dx = self.value_and_grad(fake, real_B, weight=w)
            return value.reshape(()), dx
        return term


class _LpipsFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, y, mod):
        value, dx = mod.value_and_grad(x, y, want_grad=x.requires_grad)
        ctx.save_for_backward(dx)
        return value

    @staticmethod
    def backward(ctx, g):
        (dx,) = ctx.saved_tensors
        return (None if dx is None else dx * g.reshape(()), None, None)


def lpips(x, y, net_type="vgg", version="0.1", weights=None):
    """function form the reference imports beside the class (P16:17)"""
    return LPIPS(net_type, version, weights).to(x.device)(x, y)
