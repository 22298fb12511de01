"""nn.Module mirrors of the reference's network classes, running on the HIP kernels.

Same names, constructor signatures, child names (`down1..down6, up1..up5, final`, `.model` nn.Sequential indices) and
therefore the same state_dict keys / shapes as TFC-GAN-FFT/TFCGAN_multigpu_patchFFT_16P.py:102-211, so the reference's
train / test scripts (`.apply(weights_init_normal)`, `load_state_dict`, `.cuda()`, test_TFCGAN_16Patches.py's
`load_clean_state`) work unchanged.  `nn.DataParallel` (P16:444-445) is accepted as a wrapper on ONE device only: its
multi-device replicas hold no Parameters and would share one operand-stream cache across threads, so a replica's forward
raises TfcError pointing at the one-process-per-GPU path (TrainStep + parallel.py), which is how this engine scales.  The stock torch modules inside `.model` only HOLD parameters and buffers; `forward`
never calls them -- it runs the gather-GEMM / fused kernels through torch.autograd.Function wrappers.

There is no `ContrastiveLoss` / `Discriminator` class in the reference script; both names are exported as documented
aliases (see losses.py and the bottom of this file).
"""
import numpy as np
import torch
import torch.nn as nn

from . import nets, ops
from .ops import DT_BF16, DT_F32

_DEFAULT_DTYPE = torch.bfloat16


def set_compute_dtype(dtype):
    """torch.bfloat16 (default: bf16 storage, fp32 MFMA accumulate) or torch.float32 (exact-fp32 parity mode)."""
    global _DEFAULT_DTYPE
    ops.dt_of(dtype)
    _DEFAULT_DTYPE = dtype


def get_compute_dtype():
    return _DEFAULT_DTYPE


class BlurPool(nn.Module):
    """antialiased_cnns.BlurPool(channels, stride) stand-in (third-party, un-vendored in the reference; call sites :109,
    :123, :192): reflect pad (1,2,1,2), depthwise [1,3,3,1] x [1,3,3,1] / 64, buffer `filt` [C,1,4,4]."""

    def __init__(self, channels, stride=2):
        super().__init__()
        self.channels, self.stride = channels, stride
        a = np.array([1.0, 3.0, 3.0, 1.0])
        filt = torch.tensor(a[:, None] * a[None, :], dtype=torch.float32)
        filt = filt / filt.sum()
        self.register_buffer("filt", filt[None, None].repeat(channels, 1, 1, 1))

    def forward(self, x):
        return _BlurFn.apply(x, self.stride, ops.dt_of(get_compute_dtype()))


def _to_nhwc(x, dt):
    """module-boundary plumbing for stand-alone blocks: NCHW any float dtype -> NHWC View in the compute dtype."""
    t = x.detach().permute(0, 2, 3, 1).contiguous().to(ops.torch_dtype(dt))
    return ops.View(t, t.shape[3])


def _to_nchw(v, like):
    return v.t[..., v.coff:v.coff + v.C].permute(0, 3, 1, 2).to(like.dtype)


class _BlurFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, stride, dt):
        ops.require_gpu(x)
        xv = _to_nhwc(x, dt)
        N, H, W, C = xv.t.shape
        Ho = nets.pooled(H) if stride == 2 else H
        y = ops.new_act(N, Ho, nets.pooled(W) if stride == 2 else W, C, dt, x.device)
        ops.act_fwd(dt, xv, y, stats=None, slope=1.0, pool=stride)
        ctx.meta = (stride, dt, N, H, W, C)
        return _to_nchw(y, x)

    @staticmethod
    def backward(ctx, g):
        stride, dt, N, H, W, C = ctx.meta
        gv = _to_nhwc(g, dt)
        dx = ops.new_act(N, H, W, C, dt, g.device)
        ops.act_bwd(dt, 0, gv, None, N, H, W, C, dx, stats=None, slope=1.0, pool=stride)
        return _to_nchw(dx, g), None, None


def _next_seed(module):
    """per-MODULE dropout seed stream (an LCG on an attribute of the module: no process-global mutable state)"""
    s = (getattr(module, "_drop_seed", 0x5EED ^ (id(module) & 0xFFFF)) * 1103515245 + 12345) & 0x7FFFFFFF
    object.__setattr__(module, "_drop_seed", s)
    return s


def _refuse_replica(module):
    """nn.DataParallel's replicate() marks its per-device copies `_is_replica`; they carry plain tensors instead of Parameters and share
    this module's operand-stream cache between threads -- refuse loudly instead of computing with a mix of weights"""
    if getattr(module, "_is_replica", False):
        raise ops._lib.TfcError(f"{type(module).__name__} was called inside a multi-device nn.DataParallel replica. This engine scales as one "
                                "process per GPU: launch with torch.distributed.run, build the modules on cuda:LOCAL_RANK and use "
                                "tfc_gan_amd.TrainStep (bucketed RCCL all-reduce, tfc_gan_amd.parallel) instead of nn.DataParallel (INTEGRATION.md).")


class _DownFn(torch.autograd.Function):
    """Conv2d(k4,s1,p1,no bias) -> [InstanceNorm2d] -> LeakyReLU(0.2) -> BlurPool(s2) -> [Dropout]  (reference :102-115)"""

    @staticmethod
    def forward(ctx, x, w, normalize, drop_p, seed, dt):
        ops.require_gpu(x, w)
        cout, cin = w.shape[:2]
        xv = _to_nhwc(x, dt)
        if xv.C % 8:
            t = torch.zeros((*xv.t.shape[:3], ops.pad8(xv.C)), dtype=xv.t.dtype, device=x.device)
            t[..., :xv.C] = xv.t
            xv = ops.View(t, t.shape[3])
        N, H, W, _ = xv.t.shape
        wf = w.detach().float().contiguous()
        raw = ops.new_act(N, H - 1, W - 1, cout, dt, x.device)
        stats = torch.zeros((N, cout, 2), dtype=torch.float32, device=x.device) if normalize else None
        ops.conv_fwd(dt, ops.OP_CONV, xv, cin, cout, ops.pack_weight(dt, ops.OP_CONV, 0, wf, cin, cout), raw, stats=stats)
        y = ops.new_act(N, nets.pooled(H - 1), nets.pooled(W - 1), cout, dt, x.device)
        ops.act_fwd(dt, raw, y, stats=stats, slope=0.2, pool=2, drop_p=drop_p, seed=seed)
        ctx.saved = (xv, wf, raw, stats, normalize, drop_p, seed, dt, cin, cout)
        ctx.x_needs = x.requires_grad
        return _to_nchw(y, x)

    @staticmethod
    def backward(ctx, g):
        xv, wf, raw, stats, normalize, drop_p, seed, dt, cin, cout = ctx.saved
        N, Hc, Wc = raw.N, raw.H, raw.W
        gv = _to_nhwc(g, dt)
        d_raw = ops.new_act(N, Hc, Wc, cout, dt, g.device)
        if normalize:
            rstats = torch.zeros((N, cout, 2), dtype=torch.float32, device=g.device)
            ops.act_bwd(dt, 1, gv, raw, N, Hc, Wc, cout, None, stats=stats, slope=0.2, pool=2, drop_p=drop_p, seed=seed, rstats=rstats)
            ops.act_bwd(dt, 2, gv, raw, N, Hc, Wc, cout, d_raw, stats=stats, slope=0.2, pool=2, drop_p=drop_p, seed=seed, rstats=rstats)
        else:
            ops.act_bwd(dt, 0, gv, raw, N, Hc, Wc, cout, d_raw, stats=None, slope=0.2, pool=2, drop_p=drop_p, seed=seed)
        gw = torch.empty_like(wf)
        ops.conv_wgrad(dt, ops.OP_CONV, xv, d_raw, cin, cout, gw)
        gx = None
        if ctx.x_needs:
            dx = ops.new_act(N, xv.H, xv.W, ops.pad8(cin), dt, g.device)
            ops.conv_dgrad(dt, ops.OP_CONV, d_raw, N, xv.H, xv.W, cin, cout, ops.pack_weight(dt, ops.OP_CONV, 1, wf, cin, cout), dx)
            gx = _to_nchw(ops.View(dx.t, cin, 0), g)
        return gx, gw, None, None, None, None


class _UpFn(torch.autograd.Function):
    """ConvTranspose2d(k4,s2,p1,no bias) -> BlurPool(s1) -> InstanceNorm2d -> ReLU -> [Dropout]  (reference :118-130)"""

    @staticmethod
    def forward(ctx, x, w, drop_p, seed, dt):
        ops.require_gpu(x, w)
        cin, cout = w.shape[:2]
        xv = _to_nhwc(x, dt)
        N, H, W, _ = xv.t.shape
        wf = w.detach().float().contiguous()
        rawT = ops.new_act(N, 2 * H, 2 * W, cout, dt, x.device)
        ops.conv_fwd(dt, ops.OP_CONVT, xv, cin, cout, ops.pack_weight(dt, ops.OP_CONVT, 0, wf, cin, cout), rawT)
        blur = ops.new_act(N, 2 * H, 2 * W, cout, dt, x.device)
        bstats = torch.zeros((N, cout, 2), dtype=torch.float32, device=x.device)
        ops.act_fwd(dt, rawT, blur, stats=None, slope=1.0, pool=1, stats_out=bstats)
        y = ops.new_act(N, 2 * H, 2 * W, cout, dt, x.device)
        ops.act_fwd(dt, blur, y, stats=bstats, slope=0.0, pool=0, drop_p=drop_p, seed=seed)
        ctx.saved = (xv, wf, blur, bstats, drop_p, seed, dt, cin, cout)
        return _to_nchw(y, x)

    @staticmethod
    def backward(ctx, g):
        xv, wf, blur, bstats, drop_p, seed, dt, cin, cout = ctx.saved
        N, H, W = blur.N, blur.H, blur.W
        gv = _to_nhwc(g, dt)
        rstats = torch.zeros((N, cout, 2), dtype=torch.float32, device=g.device)
        ops.act_bwd(dt, 1, gv, blur, N, H, W, cout, None, stats=bstats, slope=0.0, pool=0, drop_p=drop_p, seed=seed, rstats=rstats)
        d_blur = ops.new_act(N, H, W, cout, dt, g.device)
        ops.act_bwd(dt, 2, gv, blur, N, H, W, cout, d_blur, stats=bstats, slope=0.0, pool=0, drop_p=drop_p, seed=seed, rstats=rstats)
        d_rawT = ops.new_act(N, H, W, cout, dt, g.device)
        ops.act_bwd(dt, 0, d_blur, None, N, H, W, cout, d_rawT, stats=None, slope=1.0, pool=1)
        gw = torch.empty_like(wf)
        ops.conv_wgrad(dt, ops.OP_CONVT, xv, d_rawT, cin, cout, gw)
        dx = ops.new_act(N, xv.H, xv.W, cin, dt, g.device)
        ops.conv_dgrad(dt, ops.OP_CONVT, d_rawT, N, xv.H, xv.W, cin, cout, ops.pack_weight(dt, ops.OP_CONVT, 1, wf, cin, cout), dx)
        return _to_nchw(dx, g), gw, None, None, None


class UNetDown(nn.Module):
    def __init__(self, in_size, out_size, normalize=True, dropout=0.0):
        super().__init__()
        layers = [nn.Conv2d(in_size, out_size, 4, 1, 1, bias=False)]
        if normalize:
            layers.append(nn.InstanceNorm2d(out_size))
        layers.append(nn.LeakyReLU(0.2))
        layers.append(BlurPool(out_size, stride=2))
        if dropout:
            layers.append(nn.Dropout(dropout))
        self.model = nn.Sequential(*layers)
        self.normalize, self.dropout = bool(normalize), float(dropout)

    def forward(self, x):
        p = self.dropout if self.training else 0.0
        return _DownFn.apply(x, self.model[0].weight, self.normalize, p, _next_seed(self), ops.dt_of(get_compute_dtype()))


class UNetUp(nn.Module):
    def __init__(self, in_size, out_size, dropout=0.0):
        super().__init__()
        layers = [nn.ConvTranspose2d(in_size, out_size, 4, 2, 1, bias=False), BlurPool(out_size, stride=1),
                  nn.InstanceNorm2d(out_size), nn.ReLU(inplace=True)]
        if dropout:
            layers.append(nn.Dropout(dropout))
        self.model = nn.Sequential(*layers)
        self.dropout = float(dropout)

    def forward(self, x, skip_input):
        p = self.dropout if self.training else 0.0
        y = _UpFn.apply(x, self.model[0].weight, p, _next_seed(self), ops.dt_of(get_compute_dtype()))
        return torch.cat((y, skip_input.to(y.dtype)), 1)


class _GeneratorFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, module, x, seed, *params):
        core = module._core_for(x.device)
        fake, gctx = core.forward(x.detach().float().contiguous(), seed=seed, train=module.training)
        ctx.core, ctx.gctx, ctx.module = core, gctx, module
        return fake

    @staticmethod
    def backward(ctx, g):
        core = ctx.core
        names = nets.g_param_names()
        grads = {k: torch.empty_like(core.params[k]) for k in names}
        gx = core.backward(ctx.gctx, g, grads, need_input_grad=ctx.needs_input_grad[1])
        ctx.gctx = None
        return (None, gx, None) + tuple(grads[k] for k in names)


class GeneratorUNet(nn.Module):
    """reference :136-174. forward(x[N,3,S,S]) -> fake_B in (-1,1), fp32 NCHW (the reference casts to HalfTensor, :173)."""

    def __init__(self, img_shape):
        super().__init__()
        channels, self.h, self.w = img_shape
        self.down1 = UNetDown(channels, 64, normalize=False)
        self.down2 = UNetDown(64, 128)
        self.down3 = UNetDown(128, 256, dropout=0.5)
        self.down4 = UNetDown(256, 512, dropout=0.5)
        self.down5 = UNetDown(512, 512, normalize=False)
        self.down6 = UNetDown(512, 512)
        self.up1 = UNetUp(512, 512)
        self.up2 = UNetUp(1024, 512, dropout=0.5)
        self.up3 = UNetUp(1024, 256, dropout=0.5)
        self.up4 = UNetUp(512, 128)
        self.up5 = UNetUp(256, 64)
        self.final = nn.Sequential(nn.Upsample(scale_factor=2), nn.ZeroPad2d((1, 0, 1, 0)), nn.Conv2d(128, channels, 4, padding=1), nn.Tanh())
        self.channels = channels
        self.compute_dtype = None            # None -> package default
        self._core = None
        self._core_key = None
        self._weights_gen = 0                # bumped by TrainStep after each raw-pointer Adam update (neither data_ptr nor _version moves)

    def named_core_params(self):
        sd = dict(self.named_parameters())
        return {k: sd[k] for k in nets.g_param_names()}

    def _core_for(self, device):
        dt = ops.dt_of(self.compute_dtype or get_compute_dtype())
        params = self.named_core_params()
        key = (dt, str(device), self._weights_gen) + tuple((p.data_ptr(), p._version) for p in params.values())
        if self._core is None or self._core.dt != dt:
            self._core = nets.GeneratorCore(dt, self.channels)
        if key != self._core_key:
            for k, p in params.items():
                if p.dtype != torch.float32 or not p.is_cuda:
                    raise ops._lib.TfcError(f"GeneratorUNet parameter {k} must be an fp32 CUDA tensor (got {p.dtype}, {p.device})")
            self._core.set_params({k: p.detach() for k, p in params.items()})
            self._core.repack()
            self._core_key = key
        return self._core

    def forward(self, x):
        _refuse_replica(self)
        params = self.named_core_params()
        return _GeneratorFn.apply(self, x, _next_seed(self), *params.values())


class _DiscriminatorFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, module, img_a, img_b, *params):
        core = module._core_for(img_a.device)
        logits, dctx = core.forward(img_a.detach().float().contiguous(), img_b.detach().float().contiguous(),
                                    power_iter=module.training)
        ctx.core, ctx.dctx = core, dctx
        ctx.need_a = img_a.requires_grad
        ctx.need_w = any(p.requires_grad for p in params)
        N, H, W = logits.N, logits.H, logits.W
        return logits.t[..., 0].reshape(N, 1, H, W).float()

    @staticmethod
    def backward(ctx, g):
        core, dctx = ctx.core, ctx.dctx
        dt = core.dt
        N, _, H, W = g.shape
        gl = ops.new_act(N, H, W, 8, dt, g.device, zero=True)
        gl.t[..., 0] = g.reshape(N, H, W).to(gl.t.dtype)
        names = nets.d_param_names()
        grads = {k: torch.empty_like(core.params[k]) for k in names} if ctx.need_w else None
        ga = core.backward(dctx, gl, grads, need_input_grad=ctx.need_a)
        ctx.dctx = None
        return (None, ga, None) + (tuple(grads[k] for k in names) if grads else (None,) * len(names))


class Discriminator1(nn.Module):
    """reference :182-211 (spectral-norm PatchGAN). forward(img_A, img_B) -> logits [N,1,S/16,S/16] fp32."""

    def __init__(self, img_shape):
        super().__init__()
        channels, self.h, self.w = img_shape

        def discriminator_block(in_filters, out_filters):
            return [torch.nn.utils.parametrizations.spectral_norm(nn.Conv2d(in_filters, out_filters, 4, stride=1, padding=1)),
                    nn.LeakyReLU(0.2, inplace=True), BlurPool(out_filters, stride=2)]

        self.model = nn.Sequential(*discriminator_block(channels * 2, 64), *discriminator_block(64, 128),
                                   *discriminator_block(128, 256), *discriminator_block(256, 512),
                                   nn.ZeroPad2d((1, 0, 1, 0)), nn.Conv2d(512, 1, 4, padding=1, bias=False))
        self.channels = channels
        self.compute_dtype = None
        self._core = None
        self._core_key = None
        self._weights_gen = 0

    def named_core_params(self):
        sd = dict(self.named_parameters())
        return {k: sd[k] for k in nets.d_param_names()}

    def named_core_buffers(self):
        sd = dict(self.named_buffers())
        out = {}
        for i, _, _ in nets.D_BLOCKS:
            for s in ("_u", "_v"):
                k = f"model.{i}.parametrizations.weight.0.{s}"
                out[k] = sd[k]
        return out

    def _core_for(self, device):
        dt = ops.dt_of(self.compute_dtype or get_compute_dtype())
        params, bufs = self.named_core_params(), self.named_core_buffers()
        key = (dt, str(device), self._weights_gen) + tuple((p.data_ptr(), p._version) for p in params.values()) + tuple(b.data_ptr() for b in bufs.values())
        if self._core is None or self._core.dt != dt:
            self._core = nets.DiscriminatorCore(dt, self.channels)
        if key != self._core_key:
            for k, p in list(params.items()) + list(bufs.items()):
                if p.dtype != torch.float32 or not p.is_cuda:
                    raise ops._lib.TfcError(f"Discriminator1 tensor {k} must be an fp32 CUDA tensor (got {p.dtype}, {p.device})")
            self._core.set_params({k: p.detach() for k, p in params.items()}, dict(bufs))
            self._core.repack()
            self._core_key = key
        return self._core

    def forward(self, img_A, img_B):
        _refuse_replica(self)
        params = self.named_core_params()
        return _DiscriminatorFn.apply(self, img_A, img_B, *params.values())


# The reference script names the class Discriminator1; older scripts of the same repo (TFC-STN/archive/*) call the identical
# class Discriminator -- north_star uses that name.
Discriminator = Discriminator1


def weights_init_normal(m):
    """reference :218-224: N(0, 0.02) on every module whose class name contains "Conv" (BatchNorm2d: N(1, 0.02), bias 0)."""
    classname = m.__class__.__name__
    if classname.find("Conv") != -1:
        torch.nn.init.normal_(m.weight.data, 0.0, 0.02)
    elif classname.find("BatchNorm2d") != -1:
        torch.nn.init.normal_(m.weight.data, 1.0, 0.02)
        torch.nn.init.constant_(m.bias.data, 0.0)
