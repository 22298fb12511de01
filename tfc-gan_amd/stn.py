"""STN21 configuration (BASELINE.json configs[4]; reference TFC-STN/TFCGAN_STN21_Original_NewModel3_Official.py, "STN"): the pieces of that
script that are NOT the shared GeneratorUNet / Discriminator classes, on HIP kernels with the reference's call surface.

  affine_warp(src, theta)            STN:220-229  per-sample F.affine_grid(align_corners=True) + F.grid_sample(bicubic, border, align_corners=True)
  Warp.forward(theta_delta, src)     STN:204-231  `Net.forward` after the localiser: adds the identity, warps every sample, concatenates
  morph_gradient(x)                  STN:444-449  kornia.morphology.gradient with the 3 x 3 cross (kornia is absent here: restated from its
                                                  published algorithm -- geodesic borders -- PARITY UNPINNED; tested against a torch restatement)
  morph_triplet(real_A, real_B, reg_B)  STN:444-459  criterion_morph(1 - grad(reg_B), 1 - grad(real_A), 1 - grad(real_B)), differentiable w.r.t. reg_B

The localiser itself (kornia.contrib.VisionTransformer + MLP, STN:150-198) is plain dense layers over 17 tokens and is left to the caller
(kornia is not installed; any module that returns theta [N, 2, 3] plugs in): `Warp` takes its output.
"""
import ctypes

import torch
import torch.nn as nn

from . import ops
from ._lib import check


def _c(t):
    return t.detach().contiguous().float()


class _AffineWarpFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, src, theta):
        ops.require_gpu(src, theta)
        s, th = _c(src), _c(theta).reshape(-1, 6)
        N, C, H, W = s.shape
        assert th.shape[0] == N, "one 2x3 matrix per sample"
        out = torch.empty_like(s)
        check(ops.lib().tfc_affine_warp_fwd(ops.stream_ptr(), ops._p(s), ops._p(th), ops._p(out), N, C, H, W), "tfc_affine_warp_fwd")
        ctx.save_for_backward(s, th)
        ctx.need_src = src.requires_grad
        ctx.theta_shape = theta.shape
        return out

    @staticmethod
    def backward(ctx, g):
        s, th = ctx.saved_tensors
        N, C, H, W = s.shape
        g = _c(g)
        dth = torch.empty((N, 6), dtype=torch.float32, device=s.device)
        dsrc = torch.empty_like(s) if ctx.need_src else None
        check(ops.lib().tfc_affine_warp_bwd(ops.stream_ptr(), ops._p(s), ops._p(th), ops._p(g), ops._p(dth), ops._p(dsrc), N, C, H, W, ops.part_ws(s.device)),
              "tfc_affine_warp_bwd")
        return dsrc, dth.reshape(ctx.theta_shape)


def affine_warp(src, theta):
    """src [N,C,H,W], theta [N,2,3] -> warped [N,C,H,W] (fp32); differentiable w.r.t. theta and src."""
    return _AffineWarpFn.apply(src, theta)


class Warp(nn.Module):
    """The tail of the reference's `Net.forward` (STN:204-231): theta = identity + delta, every sample warped with its own matrix."""

    def forward(self, theta_delta, src):
        ident = torch.tensor([1.0, 0.0, 0.0, 0.0, 1.0, 0.0], dtype=torch.float32, device=src.device)
        theta = theta_delta.reshape(src.shape[0], 6).float() + ident
        return affine_warp(src, theta.reshape(-1, 2, 3))


class _MorphGradFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        ops.require_gpu(x)
        xc = _c(x)
        H, W = xc.shape[-2:]
        planes = xc.numel() // (H * W)
        out = torch.empty_like(xc)
        arg = torch.empty(xc.shape, dtype=torch.uint8, device=xc.device)
        check(ops.lib().tfc_morph_grad_fwd(ops.stream_ptr(), ops._p(xc), ops._p(out), ops._p(arg), planes, H, W), "tfc_morph_grad_fwd")
        ctx.save_for_backward(arg)
        return out

    @staticmethod
    def backward(ctx, g):
        (arg,) = ctx.saved_tensors
        g = _c(g)
        H, W = g.shape[-2:]
        dx = torch.empty_like(g)
        check(ops.lib().tfc_morph_grad_bwd(ops.stream_ptr(), ops._p(g), ops._p(arg), ops._p(dx), g.numel() // (H * W), H, W), "tfc_morph_grad_bwd")
        return dx


def morph_gradient(x):
    """kornia.morphology.gradient(x, cross 3x3) = dilation(x) - erosion(x); any [..., H, W] fp32 tensor; differentiable."""
    return _MorphGradFn.apply(x)


class _RowTripletFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, p, n, margin):
        ops.require_gpu(a, p, n)
        ac, pc, nc = _c(a), _c(p), _c(n)
        W = ac.shape[-1]
        loss = torch.empty(1, dtype=torch.float32, device=ac.device)
        da = torch.empty_like(ac) if a.requires_grad else None
        check(ops.lib().tfc_row_triplet_grad(ops.stream_ptr(), ops._p(ac), ops._p(pc), ops._p(nc), ac.numel() // W, W, float(margin), 1.0,
                                             ops._p(loss), ops._p(da)), "tfc_row_triplet_grad")
        ctx.da = da
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        return (None if ctx.da is None else ctx.da * g), None, None, None


def triplet_margin_rows(anchor, positive, negative, margin=1.0):
    """nn.TripletMarginLoss(margin, p=2) (distance over the last dim, mean over the rest); differentiable w.r.t. the anchor."""
    return _RowTripletFn.apply(anchor, positive, negative, margin)


def morph_triplet(real_A, real_B, reg_B):
    """reference STN:444-459: all three images cast into the "gradient modality" (1 - morphological gradient), then
    criterion_morph(anchor = reg_B, positive = real_A, negative = real_B)."""
    m_A = 1.0 - morph_gradient(real_A.detach())
    m_B = 1.0 - morph_gradient(real_B.detach())
    m_GB = 1.0 - morph_gradient(reg_B)
    return triplet_margin_rows(m_GB, m_A, m_B, 1.0)
