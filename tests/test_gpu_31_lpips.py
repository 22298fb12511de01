"""GPU tests of the LPIPS term (P16:70-73, :598): lpips_pytorch is absent from the reference tree and from this image and its weights cannot
be fetched, so these tests compare against a torch-CPU RESTATEMENT of the package's published algorithm with seeded-random weights --
PARITY UNPINNED. torch's own conv2d / max_pool2d / relu (the functions torchvision's VGG16 is made of) are the arithmetic oracle."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

import tfc_gan_amd as T
from tfc_gan_amd import lpips as L
from tfc_gan_amd import ops

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def rnd(shape, seed, scale=1.0):
    return torch.from_numpy((np.random.default_rng(seed).standard_normal(shape) * scale).astype(np.float32))


def to_view(x, dt):
    """fp32 NCHW (C % 8 == 0) -> NHWC View in dt on the GPU"""
    t = x.permute(0, 2, 3, 1).contiguous().to(DEV).to(ops.torch_dtype(dt))
    return ops.View(t, x.shape[1])


def from_view(v):
    return v.t.float().cpu().permute(0, 3, 1, 2)


def q(x, dt):
    return x.bfloat16().float() if dt == ops.DT_BF16 else x


@pytest.mark.parametrize("dt", [ops.DT_F32, ops.DT_BF16])
@pytest.mark.parametrize("N,H,W,Cin,Cout", [(2, 16, 32, 32, 64), (1, 40, 24, 64, 128), (3, 8, 8, 128, 64), (1, 20, 20, 512, 512)])
def test_conv3x3_relu_forward_and_dgrad_vs_torch(dt, N, H, W, Cin, Cout):
    """TFC_OP_CONV3 (+ bias + ReLU epilogue) and its input gradient against F.conv2d(padding=1) on the same (storage-rounded) operands"""
    x, w, b = q(rnd((N, Cin, H, W), 1), dt), q(rnd((Cout, Cin, 3, 3), 2, (2.0 / (9 * Cin)) ** 0.5), dt), rnd((Cout,), 3, 0.1)
    xr = x.clone().requires_grad_(True)
    z = F.conv2d(xr, w, b, padding=1)
    want = F.relu(z)
    go = q(rnd(tuple(z.shape), 4), dt)
    (gx,) = torch.autograd.grad(z, xr, go)
    w4 = torch.zeros((Cout, Cin, 4, 4))
    w4[:, :, :3, :3] = w
    w4[:, :, 3, :] = 7.0                                        # the fourth row / column of the slot is ignored by the op
    w4[:, :, :, 3] = -7.0
    w4 = w4.to(DEV)
    pf = ops.pack_weight(dt, ops.OP_CONV3, 0, w4, Cin, Cout)
    pd = ops.pack_weight(dt, ops.OP_CONV3, 1, w4, Cin, Cout)
    xv = to_view(x, dt)
    yv = ops.new_act(N, H, W, Cout, dt, DEV)
    ops.conv_fwd(dt, ops.OP_CONV3, xv, Cin, Cout, pf, yv, bias=b.to(DEV), flags=ops.EP_RELU)
    tol = 2e-2 if dt == ops.DT_BF16 else 2e-4
    got = from_view(yv)
    assert (got >= 0).all()
    assert (got - want.detach()).abs().max().item() <= tol * max(1.0, want.abs().max().item())
    gv = ops.new_act(N, H, W, Cin, dt, DEV)
    ops.conv_dgrad(dt, ops.OP_CONV3, to_view(go, dt), N, H, W, Cin, Cout, pd, gv)
    assert (from_view(gv) - gx).abs().max().item() <= tol * max(1.0, gx.abs().max().item())


@pytest.mark.parametrize("dt", [ops.DT_F32, ops.DT_BF16])
def test_maxpool_relu_elementwise_vs_torch(dt):
    """2x2 max pooling forward / backward (with ties: values drawn from 5 levels, torch routes to the first maximum) and the fused ReLU backward"""
    N, C, H, W = 2, 64, 12, 20
    rng = np.random.default_rng(5)
    x = torch.from_numpy(rng.integers(0, 5, (N, C, H, W)).astype(np.float32) * 0.25).requires_grad_(True)
    y = F.max_pool2d(x, 2, 2)
    go = q(rnd(tuple(y.shape), 6), dt)
    (gx,) = torch.autograd.grad(y, x, go)
    lib, st = ops.lib(), ops.stream_ptr()
    xv, gov = to_view(x.detach(), dt), to_view(go, dt)
    yv = ops.new_act(N, H // 2, W // 2, C, dt, DEV)
    T._lib.check(lib.tfc_maxpool2_fwd(st, dt, xv.ptr, yv.ptr, N, H, W, C), "maxpool fwd")
    assert torch.equal(from_view(yv), y.detach())
    gv = ops.new_act(N, H, W, C, dt, DEV)
    T._lib.check(lib.tfc_maxpool2_bwd(st, dt, xv.ptr, gov.ptr, gv.ptr, N, H, W, C), "maxpool bwd")
    assert torch.equal(from_view(gv), gx)
    # relu backward with the extra (tap head) gradient
    a, g1, g2 = q(rnd((N, C, H, W), 7), dt).clamp_min(0), q(rnd((N, C, H, W), 8), dt), q(rnd((N, C, H, W), 9), dt)
    want = q((g1 + g2) * (a > 0), dt)
    av, g1v, g2v = to_view(a, dt), to_view(g1, dt), to_view(g2, dt)
    T._lib.check(lib.tfc_relu_bwd(st, dt, g1v.ptr, av.ptr, g2v.ptr, g1v.ptr, a.numel()), "relu bwd")
    assert (from_view(g1v) - want).abs().max().item() <= (2e-2 if dt == ops.DT_BF16 else 1e-6)


def ref_head(fx, fy, w):
    nx = fx / (fx.pow(2).sum(1, keepdim=True).sqrt() + 1e-10)
    ny = fy / (fy.pow(2).sum(1, keepdim=True).sqrt() + 1e-10)
    return F.conv2d((nx - ny) ** 2, w.view(1, -1, 1, 1)).mean((2, 3), True)


@pytest.mark.parametrize("dt", [ops.DT_F32, ops.DT_BF16])
@pytest.mark.parametrize("N,C,H,W", [(3, 64, 16, 24), (2, 128, 8, 8), (2, 256, 5, 7), (5, 512, 4, 4), (2, 512, 2, 2)])
def test_lpips_head_value_and_gradient_vs_torch(dt, N, C, H, W):
    fx = q(rnd((N, C, H, W), 11).clamp_min(0), dt).requires_grad_(True)
    fx.data[0, :, 0, 0] = 0                                      # an all-zero feature vector: value 0/eps, gradient defined as 0 on that pixel's norm term
    fy = q(rnd((N, C, H, W), 12).clamp_min(0), dt)
    w = torch.from_numpy(np.random.default_rng(13).random(C).astype(np.float32))
    want = ref_head(fx, fy, w)
    (gx,) = torch.autograd.grad(want.sum() * 0.7, fx)
    out = torch.zeros(N, device=DEV)
    xv, yv = to_view(fx.detach(), dt), to_view(fy, dt)
    dv = ops.new_act(N, H, W, C, dt, DEV)
    T._lib.check(ops.lib().tfc_lpips_head(ops.stream_ptr(), dt, xv.ptr, yv.ptr, ops._p(w.to(DEV)), ops._p(out), dv.ptr, N, H, W, C, 0.7), "head")
    assert torch.allclose(out.cpu(), want.reshape(-1), rtol=1e-4, atol=1e-6)
    got = from_view(dv)
    gx = gx.clone()
    gx[0, :, 0, 0] = got[0, :, 0, 0]                            # torch's gradient at the exact zero vector is inf/nan-free but convention-dependent
    assert (got - gx).abs().max().item() <= (1e-2 if dt == ops.DT_BF16 else 1e-4) * gx.abs().max().item()


def ref_lpips(x, y, mod):
    """the package's forward restated with torch ops (module docstring of tfc_gan_amd/lpips.py)"""
    convs = mod.net.convs()

    def feats(t):
        t = (t - mod.net.mean) / mod.net.std
        out, ci = [], 0
        for c in L.VGG16_CFG:
            if c == "M":
                t = F.max_pool2d(t, 2, 2)
            else:
                t = F.relu(F.conv2d(t, convs[ci].weight, convs[ci].bias, padding=1))
                if ci in L.TAP_AFTER_CONV:
                    out.append(t)
                ci += 1
        return out
    res = [ref_head(fx, fy, seq[1].weight.reshape(-1)) for fx, fy, seq in zip(feats(x), feats(y), mod.lin)]
    return torch.sum(torch.cat(res, 0), 0, True)


def _module(seed=0):
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m = T.LPIPS(net_type="vgg", version="0.1", seed=seed)
    for c in m.net.convs():
        c.bias.data = rnd(tuple(c.bias.shape), seed + 50, 0.05)
    return m


def test_lpips_fp32_value_and_gradient_vs_restatement_unpinned():
    T.set_compute_dtype(torch.float32)
    try:
        m = _module()
        x = (rnd((2, 3, 64, 48), 21) * 0.4).clamp(-1, 1).requires_grad_(True)
        y = (rnd((2, 3, 64, 48), 22) * 0.4).clamp(-1, 1)
        want = ref_lpips(x, y, m)
        (gx,) = torch.autograd.grad(want.sum(), x)
        mg = m.to(DEV)
        xg = x.detach().to(DEV).requires_grad_(True)
        got = mg(xg, y.to(DEV))
        assert got.shape == (1, 1, 1, 1)
        assert abs(got.item() - want.item()) <= 2e-4 * abs(want.item())
        (g2,) = torch.autograd.grad(got.sum(), xg)
        assert (g2.cpu() - gx).abs().max().item() <= 2e-3 * gx.abs().max().item()
        assert mg(xg.detach(), xg.detach()).item() == 0.0           # identical images: distance exactly 0
    finally:
        T.set_compute_dtype(torch.bfloat16)


def test_lpips_bf16_close_to_restatement_unpinned():
    """bf16 storage through 13 convolutions: value within 3 %, gradient direction preserved (cosine > 0.98)"""
    m = _module(seed=3)
    x = (rnd((2, 3, 64, 64), 31) * 0.4).clamp(-1, 1).requires_grad_(True)
    y = (rnd((2, 3, 64, 64), 32) * 0.4).clamp(-1, 1)
    want = ref_lpips(x, y, m)
    (gx,) = torch.autograd.grad(want.sum(), x)
    mg = m.to(DEV)
    val, g2 = mg.value_and_grad(x.detach().to(DEV), y.to(DEV))
    assert abs(val.item() - want.item()) <= 3e-2 * abs(want.item())
    cos = F.cosine_similarity(g2.cpu().reshape(1, -1), gx.reshape(1, -1)).item()
    assert cos > 0.98, cos


def test_lpips_term_in_train_step():
    """loss_G = ... + 0.5 * LPIPS(fake_B, real_B) (P16:607): the term changes the generator update and is logged"""
    m = _module(seed=5).to(DEV)
    outs = []
    for use in (False, True):
        torch.manual_seed(1)
        G, D = T.GeneratorUNet((3, 256, 256)).to(DEV), T.Discriminator1((3, 256, 256)).to(DEV)
        G.apply(T.weights_init_normal)
        D.apply(T.weights_init_normal)
        ts = T.TrainStep(G, D, compute_dtype=torch.bfloat16)
        A, B = T.synthetic_pairs(2, seed=3)
        out = ts.step(A.to(DEV), B.to(DEV), extra_loss_G=m.as_extra_loss(0.5) if use else None)
        outs.append({k: v.float().cpu().clone() for k, v in out.items()})
        w = G.state_dict()["final.1.weight"] if "final.1.weight" in G.state_dict() else next(iter(G.state_dict().values()))
        outs[-1]["w"] = w.float().cpu().clone()
    assert "loss_extra_g" in outs[1] and torch.isfinite(outs[1]["loss_extra_g"]) and outs[1]["loss_extra_g"] > 0
    val, _ = m.value_and_grad(outs[1]["fake_B"].to(DEV), T.synthetic_pairs(2, seed=3)[1].to(DEV), weight=0.5, want_grad=False)
    assert abs(val.item() - outs[1]["loss_extra_g"].item()) <= 1e-3 * abs(val.item())
    assert abs((outs[1]["loss_G"] - outs[1]["loss_extra_g"]).item() - outs[0]["loss_G"].item()) <= 2e-2 * abs(outs[0]["loss_G"].item())
    assert not torch.equal(outs[0]["w"], outs[1]["w"])
