"""GPU parity tests of the STN21 kernels (config C5, TFC-STN/TFCGAN_STN21_Original_NewModel3_Official.py) against torch on the CPU:
F.affine_grid + F.grid_sample(bicubic, border, align_corners=True) is torch's own code (importable here), so the warp is PINNED by the very
functions the reference calls (STN:228-229); kornia is absent, so the morphological gradient is checked against a torch restatement of
kornia.morphology.gradient's published algorithm (geodesic borders) -- parity unpinned, the test name says so."""
import numpy as np
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

import tfc_gan_amd as T

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def rnd(shape, seed, scale=1.0):
    return torch.from_numpy((np.random.default_rng(seed).standard_normal(shape) * scale).astype(np.float32))


def ref_warp(src, theta):
    """Net.forward of the reference, STN:220-231: per-sample affine_grid / grid_sample, concatenated"""
    outs = []
    for i in range(src.shape[0]):
        grid = F.affine_grid(theta[i:i + 1], src[i:i + 1].size(), align_corners=True)
        outs.append(F.grid_sample(src[i:i + 1], grid, mode="bicubic", padding_mode="border", align_corners=True))
    return torch.cat(outs)


@pytest.mark.parametrize("N,C,H,W,amp", [(3, 3, 64, 48, 0.05), (2, 3, 256, 256, 0.15), (2, 1, 33, 70, 0.6), (1, 3, 128, 128, 0.0)])
def test_affine_warp_forward_backward_vs_torch(N, C, H, W, amp):
    """forward + gradients w.r.t. theta AND src against torch autograd; amp = size of the affine perturbation (0.6: large parts of the grid
    leave the image -> border clipping of every tap is on the path; 0.0: the identity, every sample lands exactly on a pixel)"""
    src = rnd((N, C, H, W), 1).requires_grad_(True)
    ident = torch.tensor([[1.0, 0, 0], [0, 1.0, 0]])
    theta = (ident[None] + rnd((N, 2, 3), 2, amp)).requires_grad_(True)
    want = ref_warp(src, theta)
    go = rnd(tuple(want.shape), 3)
    gsrc, gth = torch.autograd.grad(want, (src, theta), go)
    s2 = src.detach().to(DEV).requires_grad_(True)
    t2 = theta.detach().to(DEV).requires_grad_(True)
    got = T.affine_warp(s2, t2)
    assert (got.cpu() - want.detach()).abs().max().item() <= 2e-4 * max(1.0, want.abs().max().item())
    g2s, g2t = torch.autograd.grad(got, (s2, t2), go.to(DEV))
    assert (g2s.cpu() - gsrc).abs().max().item() <= 1e-4 * max(1.0, gsrc.abs().max().item())
    # theta gradient: a sum over H*W*C pixels of O(1) terms times (W-1)/2
    scale = max(1.0, gth.abs().max().item())
    assert (g2t.cpu() - gth).abs().max().item() <= 2e-3 * scale, (g2t.cpu(), gth)
    # the module form: identity + delta (STN:208-211)
    w = T.Warp()(t2.detach() - ident.to(DEV), s2.detach())
    assert torch.allclose(w, got.detach(), atol=1e-5)


def ref_morph_gradient(x):
    """kornia.morphology.gradient(x, cross) restated: dilation - erosion over the in-image neighbours of the 3x3 cross (geodesic border:
    padding never wins)"""
    big = 1e4
    xp_max = F.pad(x, (1, 1, 1, 1), value=-big)
    xp_min = F.pad(x, (1, 1, 1, 1), value=big)
    H, W = x.shape[-2:]
    offs = [(1, 1), (0, 1), (2, 1), (1, 0), (1, 2)]
    stack_max = torch.stack([xp_max[..., a:a + H, b:b + W] for a, b in offs])
    stack_min = torch.stack([xp_min[..., a:a + H, b:b + W] for a, b in offs])
    return stack_max.max(0).values - stack_min.min(0).values


def test_morph_gradient_and_triplet_vs_torch_restatement_unpinned():
    """forward, backward (arg-max / arg-min routing) and the whole morph_triplet loss of STN:444-459; continuous random images have no ties"""
    x = rnd((2, 3, 40, 56), 5).requires_grad_(True)
    want = ref_morph_gradient(x)
    go = rnd(tuple(want.shape), 6)
    (gx,) = torch.autograd.grad(want, x, go)
    x2 = x.detach().to(DEV).requires_grad_(True)
    got = T.morph_gradient(x2)
    assert torch.equal(got.cpu(), want.detach())
    (g2,) = torch.autograd.grad(got, x2, go.to(DEV))
    assert (g2.cpu() - gx).abs().max().item() <= 1e-6
    # constant image: gradient zero everywhere, borders included
    assert T.morph_gradient(torch.full((1, 1, 8, 9), 0.3, device=DEV)).abs().max().item() == 0.0
    A, B, R = rnd((2, 3, 64, 64), 7), rnd((2, 3, 64, 64), 8), rnd((2, 3, 64, 64), 9).requires_grad_(True)
    crit = nn.TripletMarginLoss(margin=1.0, p=2)
    lw = crit(1 - ref_morph_gradient(R), 1 - ref_morph_gradient(A), 1 - ref_morph_gradient(B))
    (gR,) = torch.autograd.grad(lw, R)
    R2 = R.detach().to(DEV).requires_grad_(True)
    lg = T.morph_triplet(A.to(DEV), B.to(DEV), R2)
    assert abs(lg.item() - lw.item()) <= 1e-5 * max(1.0, abs(lw.item()))
    (g2R,) = torch.autograd.grad(lg, R2)
    assert (g2R.cpu() - gR).abs().max().item() <= 1e-6 + 1e-4 * gR.abs().max().item()


def test_row_triplet_with_anchor_gradient_vs_torch():
    a, p, n = rnd((3, 2, 17, 50), 11).requires_grad_(True), rnd((3, 2, 17, 50), 12), rnd((3, 2, 17, 50), 13)
    want = nn.TripletMarginLoss(margin=1.0, p=2)(a, p, n)
    (ga,) = torch.autograd.grad(want, a)
    a2 = a.detach().to(DEV).requires_grad_(True)
    got = T.triplet_margin_rows(a2, p.to(DEV), n.to(DEV))
    assert abs(got.item() - want.item()) <= 1e-6 * max(1.0, abs(want.item()))
    (g2,) = torch.autograd.grad(got, a2)
    assert (g2.cpu() - ga).abs().max().item() <= 1e-7 + 1e-5 * ga.abs().max().item()


def test_stn_warp_full_size_properties_batch32():
    """config C5 size (batch 32, 256x256): the identity warp returns the source; a pure translation by k pixels equals a shifted copy in the
    interior; the theta gradient of a loss that does not depend on position is zero"""
    N = 32
    src = torch.rand((N, 3, 256, 256), device=DEV)
    ident = torch.tensor([[1.0, 0, 0], [0, 1.0, 0]], device=DEV).repeat(N, 1, 1)
    out = T.affine_warp(src, ident)
    assert (out - src).abs().max().item() <= 2e-5
    k = 7
    th = ident.clone()
    th[:, 0, 2] = 2.0 * k / 255.0                                   # x_src = x_out + k pixels
    sh = T.affine_warp(src, th)
    assert (sh[..., :, : 256 - k - 2] - src[..., :, k: 256 - 2]).abs().max().item() <= 1e-4
    const = torch.full((2, 3, 256, 256), 0.25, device=DEV)
    t2 = ident[:2].clone().requires_grad_(True)
    T.affine_warp(const, t2).sum().backward()
    # analytically zero (the derivative weights of a cubic kernel sum to zero); 196,608 fp32 terms of size 0.25 * 127.5 * 1e-7 each remain
    assert t2.grad.abs().max().item() <= 0.2
