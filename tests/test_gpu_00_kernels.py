"""GPU parity tests, kernel level: every HIP entry point of include/tfc_gan.h against the CPU oracle / torch-CPU fp32 on the
same seeded inputs.  fp32 mode (TFC_DT_F32) must agree to fp32 round-off; bf16 mode is compared against the same
computation done on bf16-rounded operands, with a tolerance of a few bf16 ulps of the result scale (stated per test)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

import tfc_gan_amd as T
from oracle import tfcgan_oracle as O
from tfc_gan_amd import _lib, ops
from tfc_gan_amd.ops import DT_BF16, DT_F32, View

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def rnd(shape, seed, scale=1.0):
    return torch.from_numpy((np.random.default_rng(seed).standard_normal(shape) * scale).astype(np.float32))


def q(x, dt):
    """round to the storage dtype (identity in fp32 mode)"""
    return x.to(torch.bfloat16).float() if dt == DT_BF16 else x


def to_view(x_nchw, dt, pad_to=None):
    """NCHW fp32 CPU -> NHWC View on the GPU in dtype dt (channels zero-padded to a multiple of 8)."""
    n, c, h, w = x_nchw.shape
    cp = pad_to or ops.pad8(c)
    t = torch.zeros((n, h, w, cp), dtype=torch.float32)
    t[..., :c] = x_nchw.permute(0, 2, 3, 1)
    return View(t.to(DEV).to(ops.torch_dtype(dt)).contiguous(), c)


def from_view(v, c=None):
    c = c or v.C
    return v.t[..., v.coff:v.coff + c].float().cpu().permute(0, 3, 1, 2).contiguous()


def tol(dt, scale):
    return (2e-5 if dt == DT_F32 else 1.2e-2) * max(scale, 1e-6)


def test_probe_lane_maps():
    """the MFMA / transposing-read lane maps every kernel is written against (integer data, exact)"""
    out = torch.zeros(2560, dtype=torch.float32, device=DEV)
    _lib.check(_lib.load().tfc_probe_mfma(ops.stream_ptr(), ops._p(out)), "probe")
    o = out.cpu().numpy()
    i = np.arange(32)[:, None]
    j = np.arange(32)[None, :]
    for sec, K in ((0, 16), (1, 2)):
        k = np.arange(K)
        A = (3 * i[:, :, None] + k[None, None, :]) % 7          # A[i][k]
        B = (k[:, None] + 5 * j[0][None, :]) % 5                # B[k][j]
        D = (A[:, 0, :].astype(np.float64) @ B.astype(np.float64))
        got = np.zeros((32, 32))
        for l in range(64):
            for r in range(16):
                got[(r & 3) + 8 * (r >> 2) + 4 * (l >> 5), l & 31] = o[sec * 1024 + l * 16 + r]
        np.testing.assert_array_equal(got, D)
    for l in range(64):
        for e in range(8):
            assert o[2048 + l * 8 + e] == (8 * (l >> 5) + e) * 32 + 16 * ((l >> 4) & 1) + (l & 15)


def ref_conv(op, x, w, bias=None):
    if op == ops.OP_CONV:
        return F.conv2d(x, w, bias, padding=1)
    if op == ops.OP_PADCONV:
        return F.conv2d(F.pad(x, (1, 0, 1, 0)), w, bias, padding=1)
    if op == ops.OP_CONVT:
        return F.conv_transpose2d(x, w, bias, stride=2, padding=1)
    return F.conv2d(F.pad(F.interpolate(x, scale_factor=2), (1, 0, 1, 0)), w, bias, padding=1)


CONV_CASES = [  # op, N, H, W, Cin, Cout
    (ops.OP_CONV, 2, 20, 20, 64, 128), (ops.OP_CONV, 1, 33, 18, 3, 64), (ops.OP_CONV, 2, 9, 9, 128, 64),
    (ops.OP_CONV, 1, 16, 16, 6, 64), (ops.OP_CONV, 1, 8, 8, 512, 512), (ops.OP_PADCONV, 2, 16, 16, 512, 1),
    (ops.OP_CONVT, 2, 4, 4, 512, 512), (ops.OP_CONVT, 1, 17, 9, 256, 64), (ops.OP_CONVT, 1, 8, 8, 1024, 256),
    (ops.OP_UPCONV, 1, 16, 24, 128, 3),
]


@pytest.fixture
def force_cfg():
    def setter(cfg):
        _lib.check(_lib.load().tfc_debug_set_igemm_config(cfg), "set cfg")
    yield setter
    setter(-1)


@pytest.mark.parametrize("cfg", [-1, 0, 1, 2, 3, 16, 17, 18, 31])
@pytest.mark.parametrize("dt", [DT_F32, DT_BF16])
@pytest.mark.parametrize("op,N,H,W,Cin,Cout", CONV_CASES)
def test_conv_family(op, N, H, W, Cin, Cout, dt, cfg, force_cfg):
    if dt == DT_F32 and cfg >= 16:
        pytest.skip("fp32 always runs the one-tile-per-workgroup kernel: covered by cfg < 16")
    force_cfg(cfg)                  # every workgroup-tile variant of the gather GEMM; + 16 = the one-tile-per-workgroup kernel in bf16
    seed = op * 1000 + Cin + Cout
    K = Cin * (4 if op == ops.OP_CONVT else 16)
    x = q(rnd((N, Cin, H, W), seed), dt).requires_grad_(True)
    wshape = (Cin, Cout, 4, 4) if op == ops.OP_CONVT else (Cout, Cin, 4, 4)
    w = rnd(wshape, seed + 1, 1.0 / np.sqrt(K))
    wq = q(w, dt).requires_grad_(True)
    bias = rnd((Cout,), seed + 2, 0.5)
    y = ref_conv(op, x, wq, bias)
    go = q(rnd(tuple(y.shape), seed + 3), dt)
    gx, gw = torch.autograd.grad(y, (x, wq), go)
    OH, OW = y.shape[2:]
    wd = w.to(DEV)
    xv = to_view(x.detach(), dt)
    # forward (+ bias, + InstanceNorm statistics epilogue)
    yv = ops.new_act(N, OH, OW, ops.pad8(Cout), dt, DEV, zero=True)
    stats = torch.zeros((N, Cout, 2), dtype=torch.float32, device=DEV)
    pk = ops.pack_weight(dt, op, 0, wd, Cin, Cout)
    ops.conv_fwd(dt, op, xv, Cin, Cout, pk, View(yv.t, Cout), bias=bias.to(DEV), stats=stats)
    got = from_view(View(yv.t, Cout))
    assert (got - y.detach()).abs().max().item() <= tol(dt, y.abs().max().item())
    # InstanceNorm statistics (P16:107 normalises the conv OUTPUT tensor): the persistent bf16 kernel takes them from the bf16 values it
    # stores -- the tensor the normaliser reads -- and must match their sums to fp32 summation order; the one-tile-per-workgroup kernel
    # (fp32 mode, cfg + 16) sums its fp32 accumulators, which differ from the stored values by the bf16 rounding noise
    persistent = dt == DT_BF16 and cfg < 16 and Cout % 8 == 0 and (ops.pad8(Cin) * 2) % 64 == 0   # 64-byte channel chunks
    ys = got if persistent else y.detach()
    s1, s2 = ys.sum((2, 3)), (ys ** 2).sum((2, 3))
    sa = (1e-5 if persistent else 2e-4 if dt == DT_F32 else 2e-3) * OH * OW * max(1.0, y.abs().max().item())
    assert torch.allclose(stats[..., 0].cpu(), s1, rtol=2e-3, atol=sa)
    assert torch.allclose(stats[..., 1].cpu(), s2, rtol=2e-3, atol=sa * max(1.0, y.abs().max().item()))
    # dgrad (+ accumulate)
    gov = to_view(go, dt)
    dxv = ops.new_act(N, H, W, ops.pad8(Cin), dt, DEV, zero=True)
    pkd = ops.pack_weight(dt, op, 1, wd, Cin, Cout)
    ops.conv_dgrad(dt, op, gov, N, H, W, Cin, Cout, pkd, View(dxv.t, Cin))
    got = from_view(View(dxv.t, Cin))
    assert (got - gx).abs().max().item() <= tol(dt, gx.abs().max().item())
    ops.conv_dgrad(dt, op, gov, N, H, W, Cin, Cout, pkd, View(dxv.t, Cin), accumulate=True)
    got2 = from_view(View(dxv.t, Cin))
    assert (got2 - 2 * gx).abs().max().item() <= 2.5 * tol(dt, gx.abs().max().item())
    # wgrad (+ accumulate)
    dw = torch.zeros(wshape, dtype=torch.float32, device=DEV)
    ops.conv_wgrad(dt, op, xv, gov, Cin, Cout, dw)
    wtol = (1e-4 if dt == DT_F32 else 2e-3) * gw.abs().max().item() + 1e-5
    assert (dw.cpu() - gw).abs().max().item() <= wtol
    ops.conv_wgrad(dt, op, xv, gov, Cin, Cout, dw, accumulate=True)
    assert (dw.cpu() - 2 * gw).abs().max().item() <= 2 * wtol


@pytest.mark.parametrize("op,N,H,W,Cin,Cout,cfg", [(ops.OP_CONV, 12, 70, 70, 64, 128, 0), (ops.OP_CONVT, 6, 40, 40, 64, 64, 2),
                                                    (ops.OP_CONV, 9, 50, 66, 128, 64, 1), (ops.OP_UPCONV, 5, 48, 40, 64, 32, 2)])
def test_persistent_gather_gemm_many_tiles(op, N, H, W, Cin, Cout, cfg, force_cfg):
    """more work items than resident workgroups (2 per CU): every workgroup walks several tiles, so the cross-tile pipeline (next tile's halo
    and weight fragments requested during the current tile's last stage, staged-tile / bias reuse in LDS, a partial last round) is on the
    path; forward with bias + statistics, input gradient with accumulation, against torch on bf16-rounded operands"""
    dt = DT_BF16
    force_cfg(cfg)
    seed = 4000 + Cin + Cout
    K = Cin * (4 if op == ops.OP_CONVT else 16)
    x = q(rnd((N, Cin, H, W), seed), dt).requires_grad_(True)
    wshape = (Cin, Cout, 4, 4) if op == ops.OP_CONVT else (Cout, Cin, 4, 4)
    w = rnd(wshape, seed + 1, 1.0 / np.sqrt(K))
    wq = q(w, dt).requires_grad_(True)
    bias = rnd((Cout,), seed + 2, 0.5)
    y = ref_conv(op, x, wq, bias)
    go = q(rnd(tuple(y.shape), seed + 3), dt)
    (gx,) = torch.autograd.grad(y, x, go)
    OH, OW = y.shape[2:]
    xv = to_view(x.detach(), dt)
    yv = ops.new_act(N, OH, OW, Cout, dt, DEV, zero=True)
    stats = torch.zeros((N, Cout, 2), dtype=torch.float32, device=DEV)
    ops.conv_fwd(dt, op, xv, Cin, Cout, ops.pack_weight(dt, op, 0, w.to(DEV), Cin, Cout), yv, bias=bias.to(DEV), stats=stats)
    got = from_view(yv)
    if op == ops.OP_UPCONV:                                      # collapsed taps: weights summed in fp32, rounded once (see the head test)
        y = ref_conv(op, x, w, bias)
    assert (got - y.detach()).abs().max().item() <= tol(dt, y.abs().max().item())
    assert torch.allclose(stats[..., 0].cpu(), got.sum((2, 3)), rtol=1e-3, atol=1e-5 * OH * OW * y.abs().max().item())
    assert torch.allclose(stats[..., 1].cpu(), (got ** 2).sum((2, 3)), rtol=1e-3, atol=1e-5 * OH * OW * y.abs().max().item() ** 2)
    if op != ops.OP_UPCONV:
        base = q(rnd((N, Cin, H, W), seed + 4), dt)
        dxv = to_view(base, dt)
        ops.conv_dgrad(dt, op, to_view(go, dt), N, H, W, Cin, Cout, ops.pack_weight(dt, op, 1, w.to(DEV), Cin, Cout), dxv, accumulate=True)
        want = base + gx
        assert (from_view(dxv) - want).abs().max().item() <= 1.5 * tol(dt, want.abs().max().item())


@pytest.mark.parametrize("op,N,H,W,Cin,Cout,cfg", [(ops.OP_CONV, 12, 70, 70, 64, 128, 0), (ops.OP_CONVT, 6, 40, 40, 64, 64, 2), (ops.OP_CONV, 16, 50, 66, 128, 64, 1),
                                                    (ops.OP_UPCONV, 8, 48, 40, 64, 32, 2), (ops.OP_CONV, 5, 128, 128, 128, 256, 3), (ops.OP_CONVT, 9, 16, 16, 256, 128, 3),
                                                    (ops.OP_CONV, 11, 70, 70, 64, 128, 0)])
def test_two_half_gather_gemm_bit_identical(op, N, H, W, Cin, Cout, cfg, force_cfg):
    """round 3: the persistent gather GEMM runs as ONE 512-thread workgroup per CU whose two 4-wave halves share the barrier sequence, half 1 running
    (stages + 1) / 2 barrier slots behind half 0 (igemm.hip, HALVES == 2). Same tiles, same arithmetic, same order inside a tile: outputs, InstanceNorm
    statistics (fixed-order partials) and accumulated input gradients must be BIT-IDENTICAL to the two-independent-workgroups form, including ragged last
    rounds (work items not a multiple of 2 x CUs: one half of some workgroups pads with dummy barriers) and a half with nothing to do."""
    dt = DT_BF16
    seed = 4400 + Cin + Cout
    K = Cin * (4 if op == ops.OP_CONVT else 16)
    x = q(rnd((N, Cin, H, W), seed), dt)
    wshape = (Cin, Cout, 4, 4) if op == ops.OP_CONVT else (Cout, Cin, 4, 4)
    w = rnd(wshape, seed + 1, 1.0 / np.sqrt(K)).to(DEV)
    bias = rnd((Cout,), seed + 2, 0.5).to(DEV)
    OH, OW = ops.OUT_HW[op](H), ops.OUT_HW[op](W)
    go = q(rnd((N, Cout, OH, OW), seed + 3), dt)
    base = q(rnd((N, Cin, H, W), seed + 4), dt)
    xv, gov = to_view(x, dt), to_view(go, dt)
    pk, pkd = ops.pack_weight(dt, op, 0, w, Cin, Cout), (ops.pack_weight(dt, op, 1, w, Cin, Cout) if op != ops.OP_UPCONV else None)
    res = []
    for c in (cfg, cfg | 32):
        force_cfg(c)
        yv = ops.new_act(N, OH, OW, Cout, dt, DEV, zero=True)
        stats = torch.zeros((N, Cout, 2), dtype=torch.float32, device=DEV)
        ops.conv_fwd(dt, op, xv, Cin, Cout, pk, yv, bias=bias, stats=stats, flags=ops.EP_LEAKY)
        dxv = to_view(base, dt)
        if pkd is not None:
            ops.conv_dgrad(dt, op, gov, N, H, W, Cin, Cout, pkd, dxv, accumulate=True)
        torch.cuda.synchronize()
        res.append((yv.t.clone(), stats.clone(), dxv.t.clone()))
    assert res[0][0].float().abs().max().item() > 0 and res[0][1].abs().max().item() > 0
    for a, b, what in zip(res[0], res[1], ("output", "statistics", "input gradient")):
        assert torch.equal(a, b), (what, (a.float() - b.float()).abs().max().item())


@pytest.mark.parametrize("N,H,W,Cin,Cout,with_bias", [(2, 33, 18, 3, 64, False), (1, 40, 70, 6, 64, True), (6, 256, 256, 6, 64, True),
                                                       (5, 200, 256, 3, 64, False)])
def test_first_layer_weights_stationary_kernel(N, H, W, Cin, Cout, with_bias):
    """bf16, 8 padded input channels, no statistics: the persistent weights-stationary kernel (several tiles per workgroup at the
    big sizes) against torch on bf16-rounded operands; 1/sigma (oscale) and bias in the epilogue"""
    dt = DT_BF16
    x = q(rnd((N, Cin, H, W), 5), dt)
    w = rnd((Cout, Cin, 4, 4), 6, 1.0 / np.sqrt(Cin * 16))
    b = rnd((Cout,), 7, 0.5) if with_bias else None
    osc = 0.37
    want = F.conv2d(x, q(w, dt), padding=1) * osc + (b.view(1, -1, 1, 1) if with_bias else 0.0)
    if with_bias:                                                            # discriminator block 1: + LeakyReLU(0.2) in the epilogue
        want = F.leaky_relu(want, 0.2)
    yv = ops.new_act(N, H - 1, W - 1, Cout, dt, DEV, zero=True)
    pk = ops.pack_weight(dt, ops.OP_CONV, 0, w.to(DEV), Cin, Cout)
    ops.conv_fwd(dt, ops.OP_CONV, to_view(x, dt), Cin, Cout, pk, yv, bias=None if b is None else b.to(DEV),
                 oscale=torch.tensor([osc], device=DEV), flags=ops.EP_LEAKY if with_bias else 0)
    got = from_view(yv)
    assert (got - want).abs().max().item() <= tol(dt, want.abs().max().item())
    # weight gradient of the same layer: the first-layer kernel (four taps of a filter row share one MFMA tile; split-K over <= 512 workgroups)
    wq = q(w, dt).requires_grad_(True)
    z = F.conv2d(x, wq, padding=1)
    go = q(rnd(tuple(z.shape), 8), dt)
    (gw,) = torch.autograd.grad(z, wq, go)
    dw = torch.full((Cout, Cin, 4, 4), 3.0, dtype=torch.float32, device=DEV)
    ops.conv_wgrad(dt, ops.OP_CONV, to_view(x, dt), to_view(go, dt), Cin, Cout, dw)
    wtol = 2e-3 * gw.abs().max().item() + 1e-5
    assert (dw.cpu() - gw).abs().max().item() <= wtol
    ops.conv_wgrad(dt, ops.OP_CONV, to_view(x, dt), to_view(go, dt), Cin, Cout, dw, accumulate=True)
    assert (dw.cpu() - 2 * gw).abs().max().item() <= 2 * wtol


@pytest.mark.parametrize("valu", [True, False])
@pytest.mark.parametrize("N,S,Cin,disc", [(2, 40, 3, False), (1, 70, 6, True), (3, 256, 6, True), (2, 256, 3, False), (2, 251, 6, True), (1, 18, 3, True)])
def test_fused_first_block_backward_equals_unfused_chain(N, S, Cin, disc, valu, monkeypatch):
    """[BlurPool]^T -> LeakyReLU' -> weight / bias gradient of the first block in one kernel (the 266 MB gradient of the conv output is never
    written) against the chain it replaces, tfc_act_bwd(mode 0, pool 2) + tfc_conv_wgrad. Two forms: the VALU form (TFC_FIRST_BWD_VALU=1) puts the
    same d_raw bits into the MFMAs, only the split-K summation order differs (fp32 round-off: 2e-5); the product form runs the transposed blur as
    a GEMM against the tile's tap matrix on the matrix core (round 3) -- the <= 9 exact products per element are added in another order, so an
    element of d_raw may round to the neighbouring bf16 (<= 1 ulp on rare elements): 2e-4 on the weight gradient. Reflect aliases on all four
    borders, partial tiles and sizes whose last TWO tile rows are border rows (S = 251, 18), with and without accumulate."""
    monkeypatch.setenv("TFC_FIRST_BWD_VALU", "1" if valu else "0")
    rtol = 2e-5 if valu else 2e-4
    dt = DT_BF16
    assert ops.first_block_bwd_supported(dt, Cin, 64)
    H = S - 1
    Po = (H - 1) // 2 + 1
    xv = to_view(q(rnd((N, Cin, S, S), 5), dt), dt)
    yv = to_view(q(rnd((N, 64, H, H), 6), dt), dt)                                 # the stored conv output (only its sign matters)
    gv = to_view(q(rnd((N, 64, Po, Po), 7), dt), dt)
    d_raw = ops.new_act(N, H, H, 64, dt, DEV, zero=True)
    r0 = torch.zeros((N, 64), device=DEV)
    ops.act_bwd(dt, 0, gv, yv, N, H, H, 64, d_raw, stats=None, slope=0.2, pool=2, rstats=r0)
    dw0 = torch.zeros((64, Cin, 4, 4), device=DEV)
    ws = ops.conv_wgrad(dt, ops.OP_CONV, xv, d_raw, Cin, 64, dw0)
    dw1 = torch.full((64, Cin, 4, 4), 5.0, device=DEV)
    r1 = torch.zeros((N, 64), device=DEV)
    ws = ops.first_block_bwd_wgrad(dt, xv, yv, gv, Cin, 64, dw1, slope=0.2, ws=ws, bias_sums=r1 if disc else None)
    scale = dw0.abs().max().item()
    assert (dw1 - dw0).abs().max().item() <= rtol * scale + 1e-6, (dw1 - dw0).abs().max().item() / scale
    if disc:
        assert torch.allclose(r1, r0, rtol=1e-4, atol=1e-3 * r0.abs().max().item())
    ops.first_block_bwd_wgrad(dt, xv, yv, gv, Cin, 64, dw1, slope=0.2, ws=ws, accumulate=True)
    assert (dw1 - 2 * dw0).abs().max().item() <= 2 * rtol * scale + 1e-6
    # and the generic path still finds its accumulator zeroed
    dw2 = torch.zeros_like(dw0)
    ops.conv_wgrad(dt, ops.OP_CONV, xv, d_raw, Cin, 64, dw2, ws=ws)
    assert torch.allclose(dw2, dw0, rtol=1e-5, atol=1e-6 * scale)
    if not valu:
        # the sign word the first convolution can leave (tfc_conv_first_fwd): same bits as (y > 0), and the backward that reads it instead of y
        # gives the SAME weight / bias gradient bits as the one that derives the signs from y itself
        w1 = rnd((64, Cin, 4, 4), 8, 0.2).to(DEV)
        pk = ops.pack_weight(dt, ops.OP_CONV, 0, w1, Cin, 64)
        y2 = ops.new_act(N, H, H, 64, dt, DEV)
        mask = torch.zeros((N, H, H, 8), dtype=torch.uint8, device=DEV)
        bias = rnd((64,), 9, 0.3).to(DEV)
        ops.conv_first_fwd(dt, xv, Cin, 64, pk, y2, bias=bias, flags=ops.EP_LEAKY, sign_mask=mask)
        y3 = ops.new_act(N, H, H, 64, dt, DEV)
        ops.conv_fwd(dt, ops.OP_CONV, xv, Cin, 64, pk, y3, bias=bias, flags=ops.EP_LEAKY)
        assert torch.equal(y2.t, y3.t)
        want_bits = (y2.t.float() > 0).reshape(N, H, H, 8, 8).to(torch.int32)
        want = (want_bits << torch.arange(8, device=DEV, dtype=torch.int32)).sum(-1).to(torch.uint8)
        assert torch.equal(mask, want)
        dwa, dwb = torch.zeros_like(dw0), torch.zeros_like(dw0)
        ra, rb2 = torch.zeros((N, 64), device=DEV), torch.zeros((N, 64), device=DEV)
        ops.first_block_bwd_wgrad(dt, xv, y2, gv, Cin, 64, dwa, slope=0.2, ws=ws, bias_sums=ra)
        ops.first_block_bwd_wgrad(dt, xv, None, gv, Cin, 64, dwb, slope=0.2, ws=ws, bias_sums=rb2, sign_mask=mask)
        assert torch.equal(dwa, dwb) and torch.equal(ra, rb2)


@pytest.mark.parametrize("N,S,Cin,gform", [(2, 256, 6, False), (2, 256, 3, True), (3, 70, 6, False), (1, 41, 3, True), (2, 130, 6, False), (1, 18, 3, True), (1, 32, 6, False), (2, 34, 3, True),
                                               (1, 62, 6, True)])
def test_fused_first_block_forward_equals_unfused_chain(N, S, Cin, gform):
    """round 3: conv -> (+bias, x 1/sigma) -> LeakyReLU -> BlurPool(stride 2) of the first block in ONE kernel (tfc_first_block_fwd: the 266 MB conv output is
    never written, the BlurPool runs as a GEMM against the tile's tap matrix on the matrix core) against the chain it replaces, tfc_conv_first_fwd +
    tfc_act_fwd(pool = 2). Both forms: D (activation before the bf16 rounding) and G (raw output rounded, activation in fp32 inside the pooling:
    leaky(y) = max(y, 0) + 0.2 min(y, 0), both parts blurred separately). The conv values are the same bits (same MFMA order), the 16 exact tap products are
    added in another order: the pooled outputs may differ by one bf16 ulp on rare elements; the sign words must be IDENTICAL. Sizes: full (ragged last tiles
    in both directions: Wo = 128 = 8 x 15 + 8), odd sizes whose last pooled row / column uses the reflect aliases -- including those where that row (S = 70, 18) or column (S = 32, 62) is the FIRST
    of its tile, whose alias lies in front of the tile's usual conv window --, a plane smaller than one tile."""
    dt = DT_BF16
    H = S - 1
    Po = (H - 1) // 2 + 1
    xv = to_view(q(rnd((N, Cin, S, S), 31), dt), dt)
    w = rnd((64, Cin, 4, 4), 32, 0.3).to(DEV)
    pk = ops.pack_weight(dt, ops.OP_CONV, 0, w, Cin, 64)
    bias = None if gform else rnd((64,), 33, 0.3).to(DEV)
    osc = None if gform else torch.tensor([0.61], device=DEV)
    raw = ops.new_act(N, H, H, 64, dt, DEV)
    m0 = torch.zeros((N, H, H, 8), dtype=torch.uint8, device=DEV)
    ops.conv_first_fwd(dt, xv, Cin, 64, pk, raw, bias=bias, oscale=osc, flags=0 if gform else ops.EP_LEAKY, sign_mask=m0)
    want = ops.new_act(N, Po, Po, 64, dt, DEV)
    ops.act_fwd(dt, raw, want, stats=None, slope=0.2 if gform else 1.0, pool=2)
    wide = torch.full((N, Po, Po, 128), 7.0, dtype=torch.bfloat16, device=DEV)          # the generator writes into a channel window of a concat buffer
    got = View(wide, 64, 64)
    m1 = torch.zeros((N, H, H, 8), dtype=torch.uint8, device=DEV)
    ops.first_block_fwd(dt, xv, Cin, 64, pk, got, bias=bias, oscale=osc, slope=0.2, act_after_rounding=gform, sign_mask=m1)
    torch.cuda.synchronize()
    assert torch.equal(m0, m1), int((m0 != m1).sum())
    assert (wide[..., :64] == 7.0).all()                                                # the neighbouring window is untouched
    a, b = wide[..., 64:].float(), want.t.float()
    err = (a - b).abs()
    scale = b.abs().clamp_min(1e-3 * b.abs().max().item())
    assert (err <= 2.0 ** -7 * scale).all(), (err / scale).max().item()               # <= 1 bf16 ulp
    frac = (err > 0).float().mean().item()
    print(f"  elements that differ: {frac:.2e}")
    assert frac <= 2e-2, frac


@pytest.mark.parametrize("N,S,with_bias", [(2, 256, True), (1, 70, False), (3, 41, True)])
def test_blurpool_backward_from_sign_words_equals_the_stored_tensor_form(N, S, with_bias):
    """tfc_act_bwd_signs (LeakyReLU' from the sign words a first block leaves instead of its conv output) gives the bits of tfc_act_bwd(mode 0, pool 2) on the
    stored tensor, incl. the per-image bias-gradient sums"""
    dt = DT_BF16
    H = S - 1
    Po = (H - 1) // 2 + 1
    xv = to_view(q(rnd((N, 6, S, S), 41), dt), dt)
    w = rnd((64, 6, 4, 4), 42, 0.3).to(DEV)
    pk = ops.pack_weight(dt, ops.OP_CONV, 0, w, 6, 64)
    raw = ops.new_act(N, H, H, 64, dt, DEV)
    mask = torch.zeros((N, H, H, 8), dtype=torch.uint8, device=DEV)
    ops.conv_first_fwd(dt, xv, 6, 64, pk, raw, flags=ops.EP_LEAKY, sign_mask=mask)
    g = to_view(q(rnd((N, 64, Po, Po), 43), dt), dt)
    da, db = ops.new_act(N, H, H, 64, dt, DEV), ops.new_act(N, H, H, 64, dt, DEV)
    ra = torch.zeros((N, 64), device=DEV) if with_bias else None
    rb = torch.zeros((N, 64), device=DEV) if with_bias else None
    ops.act_bwd(dt, 0, g, raw, N, H, H, 64, da, stats=None, slope=0.2, pool=2, rstats=ra)
    ops.act_bwd_signs(dt, g, mask, N, H, H, 64, db, slope=0.2, rstats=rb)
    torch.cuda.synchronize()
    assert torch.equal(da.t, db.t)
    if with_bias:
        assert torch.equal(ra, rb)


def test_planned_pack_equals_single_pack():
    """tfc_conv_pack_planned (one launch for every operand stream of a network; single-slot taps take a scan-free path) writes the same bytes
    as tfc_conv_pack, for every op and pass of the path -- plain, flipped (dgrad), phase-split (transposed conv), collapsed (upsample head), 3x3"""
    for dt in (DT_BF16, DT_F32):
        jobs, refs = [], []
        cases = [(ops.OP_CONV, 64, 128), (ops.OP_CONV, 3, 64), (ops.OP_CONV, 512, 512), (ops.OP_PADCONV, 512, 1), (ops.OP_CONVT, 256, 64),
                 (ops.OP_CONVT, 1024, 512), (ops.OP_UPCONV, 128, 3)] + ([(ops.OP_CONV3, 64, 128)] if dt == DT_BF16 else [(ops.OP_CONV3, 32, 64)])
        for k, (op, cin, cout) in enumerate(cases):
            shape = (cin, cout, 4, 4) if op == ops.OP_CONVT else (cout, cin, 4, 4)
            w = rnd(shape, 100 + k, 0.1).to(DEV)
            for pas in (0, 1):
                jobs.append((op, pas, w, cin, cout))
                pre = torch.full((ops.packed_bytes(dt, op, pas, cin, cout),), 0x5A, dtype=torch.uint8, device=DEV)   # slack k-substeps stay untouched
                refs.append(ops.pack_weight(dt, op, pas, w, cin, cout, out=pre))
        plan = ops.PackPlan(dt, jobs)
        for st in plan.streams:
            st.fill_(0x5A)
        plan.run()
        torch.cuda.synchronize()
        for (op, pas, _, cin, cout), got, want in zip(jobs, plan.streams, refs):
            assert torch.equal(got, want), (dt, op, pas, cin, cout, int((got != want).sum()))
            assert (got != 0x5A).any()


def test_patchgan_head_kernel_matches_padconv():
    for dt in (DT_F32, DT_BF16):
        x = q(rnd((2, 512, 16, 16), 21), dt)
        w = rnd((1, 512, 4, 4), 22, 0.02)
        want = ref_conv(ops.OP_PADCONV, x, w if dt == DT_F32 else w)          # the head kernel reads the fp32 weights directly
        yv = ops.new_act(2, 16, 16, 8, dt, DEV, zero=True)
        ops.patchgan_head_fwd(dt, to_view(x, dt), w.to(DEV), View(yv.t, 1))
        got = from_view(View(yv.t, 1))
        assert (got - want).abs().max().item() <= (2e-5 if dt == DT_F32 else 1e-2) * want.abs().max().item()


def test_upconv_tanh_nchw_head():
    """generator head: Upsample + ZeroPad + Conv + Tanh written as fp32 NCHW (P16:153-158)"""
    for dt in (DT_F32, DT_BF16):
        x = q(rnd((2, 128, 16, 16), 5), dt)
        w = rnd((3, 128, 4, 4), 6, 0.02)
        b = rnd((3,), 7, 0.1)
        # duplicated taps of the upsampled input are collapsed: their weights are summed in fp32 and rounded to the storage dtype
        # ONCE, so the fair reference uses the unrounded weights (bf16 tolerance: 2^-9 relative on ~N(0, 0.9) pre-activations)
        want = torch.tanh(ref_conv(ops.OP_UPCONV, x, w, b))
        out = torch.empty((2, 3, 32, 32), dtype=torch.float32, device=DEV)
        ops.conv_fwd(dt, ops.OP_UPCONV, to_view(x, dt), 128, 3, ops.pack_weight(dt, ops.OP_UPCONV, 0, w.to(DEV), 128, 3), None,
                     bias=b.to(DEV), out_nchw=out)
        assert (out.cpu() - want).abs().max().item() <= (1e-5 if dt == DT_F32 else 1e-2)


@pytest.mark.parametrize("N,H,W,Cout,with_bias", [(2, 16, 16, 3, True), (1, 9, 10, 3, True), (3, 21, 40, 1, False), (5, 64, 64, 4, True)])
def test_generator_head_kernel(N, H, W, Cout, with_bias):
    """tfc_upconv_head_fwd (phases as columns of a 16-wide MFMA tile, channel chunks over the waves, persistent) against torch and
    against the gather-GEMM path it replaces; ragged tiles, several tiles per workgroup"""
    dt = DT_BF16
    x = q(rnd((N, 128, H, W), 5), dt)
    w = rnd((Cout, 128, 4, 4), 6, 0.02)
    b = rnd((Cout,), 7, 0.1) if with_bias else None
    want = torch.tanh(ref_conv(ops.OP_UPCONV, x, w, b))
    xv = to_view(x, dt)
    out = torch.full((N, Cout, 2 * H, 2 * W), 7.0, dtype=torch.float32, device=DEV)
    ops.upconv_head_fwd(dt, xv, w.to(DEV), None if b is None else b.to(DEV), out)
    assert (out.cpu() - want).abs().max().item() <= 1e-2            # collapsed weights are summed in fp32, rounded to bf16 once
    old = torch.empty_like(out)
    ops.conv_fwd(dt, ops.OP_UPCONV, xv, 128, Cout, ops.pack_weight(dt, ops.OP_UPCONV, 0, w.to(DEV), 128, Cout), None,
                 bias=None if b is None else b.to(DEV), out_nchw=old)
    assert (out - old).abs().max().item() <= 2e-5                    # same bf16 operands, fp32 accumulation in a different order


@pytest.mark.parametrize("N,H,W,Cout", [(2, 16, 16, 3), (1, 9, 10, 3), (3, 21, 40, 1), (4, 128, 128, 3)])
def test_generator_head_dgrad_kernel(N, H, W, Cout):
    """tfc_upconv_head_dgrad (collapsed 5 x 5 stride-2 window of dy, weights-stationary) against torch autograd on bf16-rounded operands and
    against the gather-GEMM path it replaces; ragged tiles, several tiles per workgroup, a destination window of a wider buffer"""
    dt = DT_BF16
    x = q(rnd((N, 128, H, W), 5), dt).requires_grad_(True)
    w = rnd((Cout, 128, 4, 4), 6, 0.05)
    y = ref_conv(ops.OP_UPCONV, x, w)
    go = q(rnd(tuple(y.shape), 7), dt)
    (gx,) = torch.autograd.grad(y, x, go)
    gov = to_view(go, dt)
    buf = torch.full((N, H, W, 160), 3.0, dtype=torch.bfloat16, device=DEV)
    ops.upconv_head_dgrad(dt, gov, N, H, W, w.to(DEV), View(buf, 128, 0))
    got = buf[..., :128].float().cpu().permute(0, 3, 1, 2)
    assert (buf[..., 128:] == 3.0).all()
    assert (got - gx).abs().max().item() <= tol(dt, gx.abs().max().item())
    old = ops.new_act(N, H, W, 128, dt, DEV, zero=True)
    ops.conv_dgrad(dt, ops.OP_UPCONV, gov, N, H, W, 128, Cout, ops.pack_weight(dt, ops.OP_UPCONV, 1, w.to(DEV), 128, Cout), old)
    assert (got - from_view(old)).abs().max().item() <= 2e-2 * gx.abs().max().item()     # both round the collapsed taps once, sums in different order


@pytest.mark.parametrize("N,H,W", [(2, 16, 16), (1, 37, 50), (5, 96, 64)])
def test_first_conv_dgrad_image(N, H, W):
    """tfc_conv_dgrad_image (four output rows packed into the columns of a 16-wide MFMA tile) against autograd on bf16-rounded operands"""
    dt, Cin, Cout, nch, osc = DT_BF16, 6, 64, 3, 0.41
    x = rnd((N, Cin, H, W), 21).requires_grad_(True)
    w = rnd((Cout, Cin, 4, 4), 22, 0.1)
    wq = q(w, dt)
    y = F.conv2d(x, wq, padding=1)
    go = q(rnd(tuple(y.shape), 23), dt)
    (gx,) = torch.autograd.grad(y, x, go)
    got = ops.conv_dgrad_image(dt, to_view(go, dt), N, H, W, Cin, w.to(DEV), torch.tensor([osc], device=DEV), nch).cpu()
    want = gx[:, :nch] * osc
    assert got.shape == want.shape
    assert (got - want).abs().max().item() <= tol(dt, want.abs().max().item())


def oracle_act(x, norm, slope, pool, mask=None, drop_p=0.0):
    y = F.instance_norm(x, eps=1e-5) if norm else x
    y = torch.where(y > 0, y, y * slope)
    if pool:
        y = O.BlurPool(x.shape[1], stride=pool)(y)
    if mask is not None:
        y = y * mask / (1.0 - drop_p)
    return y


def knife_free(shape, seed, lo=0.25, hi=2.0):
    """[N,C,H,W] fp32: every plane = a random permutation of +v / -v pairs, |v| in [lo, hi] (odd size: plus the zero-sum triple 1, 1.5, -2.5):
    plane mean exactly representable as ~0, no element within `lo` of zero"""
    n, c, h, w = shape
    rng = np.random.default_rng(seed)
    m = h * w
    out = np.empty((n, c, m), dtype=np.float32)
    for i in range(n):
        for j in range(c):
            k = (m - 3) // 2 if m % 2 else m // 2
            v = rng.uniform(lo, hi, size=k).astype(np.float32)
            vals = np.concatenate([v, -v] + ([np.array([1.0, 1.5, -2.5], dtype=np.float32)] if m % 2 else []))
            out[i, j] = vals[rng.permutation(m)]
    return torch.from_numpy(out.reshape(shape))


ACT_CASES = [  # C, H, W, norm, slope, pool, drop
    (64, 15, 15, True, 0.2, 2, 0.0), (128, 7, 7, True, 0.2, 2, 0.5), (64, 31, 30, False, 0.2, 2, 0.0), (512, 8, 8, True, 0.0, 0, 0.5),
    (64, 14, 14, False, 1.0, 1, 0.0), (256, 3, 3, True, 0.2, 2, 0.0), (8, 9, 9, False, 0.2, 2, 0.0),
    (128, 37, 50, False, 1.0, 1, 0.0), (64, 4, 5, False, 1.0, 1, 0.0), (64, 8, 33, False, 1.0, 1, 0.0), (64, 19, 67, False, 0.2, 2, 0.0),
    (8, 6, 7, False, 1.0, 1, 0.0),
]


@pytest.mark.parametrize("dt", [DT_F32, DT_BF16])
@pytest.mark.parametrize("C,H,W,norm,slope,pool,drop", ACT_CASES)
def test_fused_act_fwd_bwd(C, H, W, norm, slope, pool, drop, dt):
    N, seed = 2, 77
    # LeakyReLU'(0) / ReLU'(0) are knife edges, and through the InstanceNorm backward one flipped element changes its whole plane. Instead of
    # masking planes out of the comparison, the input is built so that NO element is near the edge: every (n, c) plane is a random permutation
    # of +-v pairs with |v| >= 0.25 (an odd plane gets the zero-sum triple 1, 1.5, -2.5), so its mean is zero up to storage rounding (~1e-4)
    # and |xhat| >= 0.1 everywhere, before and after the bf16 rounding of x. EVERY plane is compared.
    x = q(knife_free((N, C, H, W), C + H), dt).requires_grad_(True)
    Ho, Wo = ((H - 1) // 2 + 1, (W - 1) // 2 + 1) if pool == 2 else (H, W)
    mask = None
    if drop:
        keep = O.hip_keep_mask(seed, N * Ho * Wo * C, drop).reshape(N, Ho, Wo, C)
        mask = torch.from_numpy(np.ascontiguousarray(keep.transpose(0, 3, 1, 2))).float()
        dev_mask = ops.dropout_mask(N * Ho * Wo * C, drop, seed, DEV).cpu().numpy().astype(bool)
        np.testing.assert_array_equal(dev_mask, keep.reshape(-1))          # kernel RNG == numpy restatement
    y = oracle_act(x, norm, slope, pool, mask, drop)
    go = q(rnd(tuple(y.shape), 3), dt)
    (gx,) = torch.autograd.grad(y, x, go)
    xv = to_view(x.detach(), dt)
    stats = None
    if norm:
        xs = xv.t.float()
        stats = torch.stack((xs.sum((1, 2)), (xs * xs).sum((1, 2))), -1).contiguous()      # [N,C,2]
    yv = ops.new_act(N, Ho, Wo, C, dt, DEV)
    so = torch.zeros((N, C, 2), dtype=torch.float32, device=DEV)
    ops.act_fwd(dt, xv, yv, stats=stats, slope=slope, pool=pool, drop_p=drop, seed=seed, stats_out=so)
    got = from_view(yv)
    t = 2e-5 if dt == DT_F32 else 2e-2
    assert (got - y.detach()).abs().max().item() <= t * max(1.0, y.abs().max().item())
    assert torch.allclose(so[..., 0].cpu(), y.detach().sum((2, 3)), rtol=2e-3, atol=(2e-2 if dt == DT_BF16 else 2e-4) * Ho * Wo)
    assert torch.allclose(so[..., 1].cpu(), (y.detach() ** 2).sum((2, 3)), rtol=2e-2 if dt == DT_BF16 else 2e-3, atol=1e-3 * Ho * Wo)
    gov = to_view(go, dt)
    dxv = ops.new_act(N, H, W, C, dt, DEV)
    if norm:
        rstats = torch.zeros((N, C, 2), dtype=torch.float32, device=DEV)
        ops.act_bwd(dt, 1, gov, xv, N, H, W, C, None, stats=stats, slope=slope, pool=pool, drop_p=drop, seed=seed, rstats=rstats)
        ops.act_bwd(dt, 2, gov, xv, N, H, W, C, dxv, stats=stats, slope=slope, pool=pool, drop_p=drop, seed=seed, rstats=rstats)
    else:
        ops.act_bwd(dt, 0, gov, xv if slope != 1.0 else None, N, H, W, C, dxv, stats=None, slope=slope, pool=pool, drop_p=drop, seed=seed)
    gotg = from_view(dxv)
    err = (gotg - gx).abs()
    xh = F.instance_norm(x.detach(), eps=1e-5) if norm else x.detach()
    assert xh.abs().min().item() >= 0.1                            # the construction holds: nothing to mask
    assert err.max().item() <= (5e-5 if dt == DT_F32 else 3e-2) * max(1.0, gx.abs().max().item())


def test_pack_unpack_tanh_colsum():
    for dt in (DT_F32, DT_BF16):
        a, b = rnd((2, 3, 16, 24), 1), rnd((2, 3, 16, 24), 2)
        v = ops.pack_nhwc8(dt, a.to(DEV), b.to(DEV))
        want = torch.cat((a, b, torch.zeros(2, 2, 16, 24)), 1)
        assert torch.allclose(from_view(v), q(want, dt), atol=0)
        back = ops.unpack_nchw(dt, v, 3, c0=3)
        assert torch.allclose(back.cpu(), q(b, dt), atol=0)
        acc = ops.unpack_nchw(dt, v, 3, out=back.clone(), alpha=2.0, beta=1.0, c0=0)
        assert torch.allclose(acc.cpu(), q(b, dt) + 2 * q(a, dt), atol=1e-6)
        g, y = rnd((2, 3, 16, 24), 3), torch.tanh(rnd((2, 3, 16, 24), 4))
        db = torch.zeros(3, dtype=torch.float32, device=DEV)
        dv = ops.tanh_bwd_pack(dt, g.to(DEV), y.to(DEV), dbias=db)
        want = g * (1 - y * y)
        assert (from_view(dv, 3) - want).abs().max().item() <= (1e-6 if dt == DT_F32 else 2e-2)
        assert torch.allclose(db.cpu(), want.sum((0, 2, 3)), atol=1e-3)
        assert from_view(dv, 8)[:, 3:].abs().max().item() == 0
        x = q(rnd((2, 64, 9, 11), 5), dt)
        cs = torch.zeros(64, dtype=torch.float32, device=DEV)
        ops.colsum(dt, to_view(x, dt), cs)
        assert torch.allclose(cs.cpu(), x.sum((0, 2, 3)), atol=1e-3, rtol=1e-4)


def test_spectral_norm_step_and_backward():
    R, K = 128, 1024
    W = rnd((R, K), 1, 0.05)
    u = F.normalize(rnd((R,), 2), dim=0)
    v = F.normalize(rnd((K,), 3), dim=0)
    Wd, ud, vd = W.to(DEV), u.to(DEV), v.to(DEV)
    s2 = torch.zeros(2, dtype=torch.float32, device=DEV)
    uu, vv = u, v
    for it in range(3):
        ops.spectral_norm_step(Wd, ud, vd, s2, power_iter=True)
        uu, vv, sigma = O.spectral_norm_step(W, uu, vv)
        assert torch.allclose(ud.cpu(), uu, atol=2e-6) and torch.allclose(vd.cpu(), vv, atol=2e-6)
        assert abs(s2[0].item() - sigma.item()) < 1e-5 * sigma.item() and abs(s2[1].item() * sigma.item() - 1) < 1e-5
    ops.spectral_norm_step(Wd, ud, vd, s2, power_iter=False)              # eval mode: sigma only
    assert torch.allclose(ud.cpu(), uu, atol=2e-6) and abs(s2[0].item() - sigma.item()) < 1e-5 * sigma.item()
    Wr = W.clone().requires_grad_(True)
    sig = torch.dot(uu, Wr @ vv)
    G = rnd((R, K), 9)
    ((Wr / sig) * G).sum().backward()
    gout = torch.zeros((R, K), dtype=torch.float32, device=DEV)
    ops.spectral_norm_bwd(G.to(DEV), Wd, ud, vd, s2, gout)
    assert (gout.cpu() - Wr.grad).abs().max().item() <= 1e-4 * Wr.grad.abs().max().item()


def test_spectral_norm_batched_matches_oracle():
    shapes = [(64, 96), (128, 1024), (512, 8192), (256, 2048)]
    Ws = [rnd(sh, 10 + i, 0.05) for i, sh in enumerate(shapes)]
    us = [F.normalize(rnd((sh[0],), 20 + i), dim=0) for i, sh in enumerate(shapes)]
    vs = [F.normalize(rnd((sh[1],), 30 + i), dim=0) for i, sh in enumerate(shapes)]
    Wd, ud, vd = [w.to(DEV) for w in Ws], [u.to(DEV) for u in us], [v.to(DEV) for v in vs]
    sig = [torch.zeros(2, device=DEV) for _ in shapes]
    usn = [torch.zeros_like(u) for u in ud]
    vsn = [torch.zeros_like(v) for v in vd]
    ws = None
    for it in range(3):
        ws = ops.spectral_norm_step_batched(Wd, ud, vd, sig, power_iter=True, u_snaps=usn, v_snaps=vsn, ws=ws)
        for i in range(len(shapes)):
            us[i], vs[i], sigma = O.spectral_norm_step(Ws[i], us[i], vs[i])
            assert torch.allclose(ud[i].cpu(), us[i], atol=3e-6) and torch.allclose(vd[i].cpu(), vs[i], atol=3e-6), (it, i)
            assert torch.equal(usn[i], ud[i]) and torch.equal(vsn[i], vd[i])
            assert abs(sig[i][0].item() - sigma.item()) < 2e-5 * sigma.item() and abs(sig[i][1].item() * sigma.item() - 1) < 2e-5
    ops.spectral_norm_step_batched(Wd, ud, vd, sig, power_iter=False, ws=ws)          # eval: sigma only, u / v untouched
    for i in range(len(shapes)):
        assert torch.allclose(ud[i].cpu(), us[i], atol=3e-6)
        sigma = torch.dot(us[i], Ws[i] @ vs[i])
        assert abs(sig[i][0].item() - sigma.item()) < 2e-5 * sigma.item()


def test_triplet16_vs_oracle_and_golden(golden):
    g = golden("triplet16")
    fk, rl = O.synthetic_pairs(2, seed=31)
    fk = torch.tanh(fk * 1.5).requires_grad_(True)
    neg = g["neg_idx"].tolist()
    want = O.patch_triplet_loss(fk, rl, neg)
    want.backward()
    loss, dfake = ops.patch16_triplet(fk.detach().to(DEV), rl.to(DEV), neg)
    assert abs(loss.item() - float(g["loss"])) < 2e-6 and abs(loss.item() - want.item()) < 2e-6
    assert (dfake.cpu() - fk.grad).abs().max().item() < 1e-9 + 1e-5 * fk.grad.abs().max().item()
    assert torch.allclose(dfake.cpu()[:, :, ::4, ::4], torch.from_numpy(g["gfake_sub"]), atol=1e-9, rtol=1e-4)
    # r_k == k for every k: each hinge is exactly the margin and the gradient vanishes
    l1, d1 = ops.patch16_triplet(fk.detach().to(DEV), rl.to(DEV), list(range(16)))
    assert abs(l1.item() - 1.0) < 1e-6 and d1.abs().max().item() == 0.0
    # autograd surface
    f2 = fk.detach().to(DEV).requires_grad_(True)
    T.ContrastiveLoss()(f2, rl.to(DEV), neg).backward()
    assert torch.allclose(f2.grad.cpu(), fk.grad, atol=1e-9, rtol=1e-4)


def test_fft_spectrum_vs_oracle_and_golden(golden):
    gp, gg = golden("fft_patch"), golden("fft_global")
    ff, rr = O.synthetic_pairs(1, seed=41)
    ff = torch.tanh(ff * 2.0) * 0.999
    amp, pha = T.fft_components(T.make_16_patches(ff.to(DEV))[5])
    a_ref, p_ref = torch.from_numpy(gp["amp5"]), torch.from_numpy(gp["pha5"])
    assert amp.shape == (1, 1, 64, 33)
    assert (amp.cpu() - a_ref).abs().max().item() <= 2e-6 * a_ref.max().item() + 2e-2         # fp32 DFT of 8-bit data
    dphi = (pha.cpu() - p_ref).abs()
    dphi = torch.minimum(dphi, 2 * np.pi - dphi)                                              # +-pi are the same angle
    assert (dphi * a_ref).max().item() <= 0.05                                                # |dphi| <= 0.05 / amplitude
    loss, la, lp = T.patch_fft_loss(ff.to(DEV), rr.to(DEV))
    want, wa, wp = O.patch_fft_loss(ff, rr)
    assert abs(loss.item() - float(gp["loss_fft"])) <= 2e-4 * float(gp["loss_fft"])
    assert abs(la.item() - wa.item()) <= 1e-4 * wa.item() and abs(lp.item() - wp.item()) <= 2e-3
    assert abs(T.calculate_ffts(*T.make_16_patches(ff.to(DEV)), *T.make_16_patches(rr.to(DEV))).item() - want.item()) <= 2e-4 * want.item()
    # GLO-16: 256 x 129
    ga, gph = T.fft_components(ff.to(DEV), patch=False)
    assert ga.shape == (1, 1, 256, 129)
    ar = torch.from_numpy(gg["amp_sub"])
    assert (ga.cpu()[:, :, ::4, ::3] - ar).abs().max().item() <= 2e-6 * ar.max().item() + 0.2
    gl, _, _ = T.global_fft_loss(ff.to(DEV), rr.to(DEV))
    assert abs(gl.item() - float(gg["loss_fft"])) <= 3e-4 * float(gg["loss_fft"])
    # identical images -> exactly zero loss
    z, _, _ = T.patch_fft_loss(ff.to(DEV), ff.to(DEV))
    assert z.item() == 0.0


def test_mse_spec_eval_metric():
    rng = np.random.default_rng(5)
    real = rng.integers(0, 256, size=(3, 256, 256), dtype=np.uint8)
    fake = np.clip(real.astype(np.int32) + rng.integers(-40, 41, size=real.shape), 0, 255).astype(np.uint8)
    got = T.mse_spec(real, fake).cpu().numpy()
    want = np.array([O.mse_spec(real[i], fake[i]) for i in range(3)])
    np.testing.assert_allclose(got, want, rtol=2e-3)


def test_temperature_head(golden):
    g = golden("temp_head")
    tf_, TB, b_tf = O.temp_head_case(71)
    tfb = T.vectorize_temps(tf_.to(DEV)).cpu()
    assert torch.equal(tfb, O.vectorize_temps(tf_))                             # LUT of uint8 codes: bit-exact
    assert torch.equal(tfb[:, :, ::8, ::8], torch.from_numpy(g["tfb_sub"]))
    loss = T.temperature_triplet_loss(tf_.to(DEV), TB.to(DEV), b_tf.to(DEV)).item()
    assert abs(loss - float(g["loss_temp_g"])) < 1e-5 * abs(float(g["loss_temp_g"]))
    assert abs(loss - float(O.temp_triplet_loss(tf_, TB, b_tf))) < 1e-5 * abs(loss)
    # row triplet on ragged widths against torch
    for W, rows in ((5, 7), (64, 33), (300, 9)):
        a, p, n = rnd((rows, W), 1), rnd((rows, W), 2), rnd((rows, W), 3)
        got = ops.row_triplet(a.to(DEV), p.to(DEV), n.to(DEV)).item()
        assert abs(got - F.triplet_margin_loss(a, p, n, margin=1.0, p=2).item()) < 1e-5


def test_color_jitter_self_consistency_parity_unpinned():
    """torchvision is absent and the reference holds no ColorJitter fixture: the oracle's helper MIRRORS the product helper, so this is a
    CPU-vs-GPU self-consistency check of one restatement, not parity with torchvision -- PARITY UNPINNED (the jitter only prepares the
    negatives of the gradient-free temperature term)."""
    _, _, b_tf = O.temp_head_case(71)
    prm = T.color_jitter_params(np.random.default_rng(3))
    got = T.color_jitter_thermal(b_tf.to(DEV), prm).cpu()
    assert torch.allclose(got, O.color_jitter_thermal(b_tf, prm), atol=1e-6)
    assert torch.isfinite(got).all() and got.shape == b_tf.shape


def test_bce_relativistic_golden(golden):
    g = golden("bce_relativistic")
    a, b = torch.from_numpy(g["a"]), torch.from_numpy(g["b"])
    for dt in (DT_F32,):
        av, bv = to_view(a, dt), to_view(b, dt)
        da = ops.new_act(2, 16, 16, 8, dt, DEV, zero=True)
        db = ops.new_act(2, 16, 16, 8, dt, DEV, zero=True)
        lg = ops.bce_relativistic(dt, View(av.t, 1), View(bv.t, 1), 0, 0.9, da=View(da.t, 1))
        assert abs(lg.item() - float(g["loss_g"])) < 1e-6
        ar = a.clone().requires_grad_(True)
        br = b.clone().requires_grad_(True)
        O.loss_gan_generator(ar, b).backward()
        assert torch.allclose(from_view(View(da.t, 1)), ar.grad, atol=1e-8)
        ld = ops.bce_relativistic(dt, View(av.t, 1), View(bv.t, 1), 1, 0.9, 0.0, da=View(da.t, 1), db=View(db.t, 1))
        assert abs(ld.item() - float(g["loss_d"])) < 1e-6
        ar.grad = None
        O.loss_discriminator(ar, br).backward()
        assert torch.allclose(from_view(View(da.t, 1)), ar.grad, atol=1e-8) and torch.allclose(from_view(View(db.t, 1)), br.grad, atol=1e-8)


@pytest.mark.parametrize("dt", [DT_BF16, DT_F32])
def test_bce_gradient_writes_whole_pixels(dt):
    """the logit gradient lives in channel 0 of an 8-channel pixel whose other channels the head's input-gradient GEMM reads as zeros: the kernel
    must write all eight (the engine hands it UNINITIALISED buffers)"""
    N, H = 3, 16
    a = ops.View(torch.randn((N, H, H, 8), device=DEV).to(ops.torch_dtype(dt)), 1)
    b = ops.View(torch.randn((N, H, H, 8), device=DEV).to(ops.torch_dtype(dt)), 1)
    da = torch.full((N, H, H, 8), 7.0, dtype=ops.torch_dtype(dt), device=DEV)
    db = torch.full((N, H, H, 8), -7.0, dtype=ops.torch_dtype(dt), device=DEV)
    ops.bce_relativistic(dt, a, b, 1, 0.9, 0.0, da=View(da, 1, 0), db=View(db, 1, 0))
    assert (da[..., 1:] == 0).all() and (db[..., 1:] == 0).all()
    assert (da[..., 0].float().abs() > 0).any() and torch.equal(da[..., 0].float(), -db[..., 0].float())


def test_adam_vs_oracle():
    p, m, v = rnd((1000,), 1), torch.zeros(1000), torch.zeros(1000)
    pd, md, vd = p.to(DEV), m.to(DEV), v.to(DEV)
    for step in (1, 2, 3):
        g = rnd((1000,), 10 + step)
        ops.adam_step(pd, g.to(DEV), md, vd, 2e-4, 0.5, 0.999, 1e-8, step)
        p, m, v = O.adam_step(p, g, m, v, step)
        assert torch.allclose(pd.cpu(), p, atol=1e-7) and torch.allclose(md.cpu(), m, atol=1e-7)


@pytest.mark.parametrize("S,wx,wy,N,shift", [(64, 4, 4, 3, False), (64, 1, 1, 2, True), (256, 1, 1, 3, False), (256, 1, 1, 2, True)])
def test_fft_path_matches_direct_dft_and_numpy(S, wx, wy, N, shift):
    """the LDS radix-4 FFT (rows as packed real pairs, then columns) against the direct-DFT kernel it replaces and against numpy's float64
    rfft2 of the same uint8 luma (the reference's FFT_Components.make_components, P16:276-282 / G16)"""
    x, _ = O.synthetic_pairs(N, seed=300 + S)
    x = (torch.tanh(x * 1.7) * 0.999).to(DEV)
    a1, p1 = ops.fft_spectrum(x, S, wx, wy, shift=shift)
    a0, p0 = ops.fft_spectrum(x, S, wx, wy, shift=shift, direct=True)
    scale = a0.max().item()
    assert (a1 - a0).abs().max().item() <= 2e-6 * scale
    big = a0 > 1e-3 * scale                                     # phase of a near-zero bin is noise in any arithmetic
    dp = (p1 - p0).abs()
    dp = torch.minimum(dp, 2 * np.pi - dp)
    assert dp[big].max().item() <= 2e-3
    # numpy float64 on window 0 of image 0
    luma = O.pil_luma(O.to_pil_uint8(x[0, :, :S, :S].cpu()))
    f = np.fft.rfft2(luma)
    if shift:
        f = np.fft.fftshift(f)
    want = torch.from_numpy(np.abs(f).astype(np.float32))
    assert (a1[0].cpu() - want).abs().max().item() <= 4e-6 * scale
