"""GPU parity tests, network level: the nn.Module mirrors (reference call surface) and the fused TrainStep against the CPU
oracle and against the fixtures generated from the reference's own definitions (tests/golden).

fp32 parity mode: generator L1 vs oracle <= 1e-4 (BASELINE.json north_star), typically ~1e-6.
bf16 mode       : stated per test (bf16 keeps 8 significant bits; 12 layers of InstanceNorm-separated convolutions)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

import tfc_gan_amd as T
from oracle import tfcgan_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def t(a):
    return torch.from_numpy(np.asarray(a))


@pytest.fixture(autouse=True)
def _fp32_default():
    T.set_compute_dtype(torch.float32)
    yield
    T.set_compute_dtype(torch.bfloat16)


def test_blocks_vs_reference_goldens(golden):
    rng = np.random.default_rng(7)
    rn = lambda *s: torch.from_numpy(rng.standard_normal(s).astype(np.float32))  # noqa: E731
    cases = [("block_down_norm", T.UNetDown(8, 16), [rn(2, 8, 16, 16)], 21),
             ("block_down_nonorm", T.UNetDown(8, 16, normalize=False), [rn(2, 8, 15, 15)], 22),
             ("block_up", T.UNetUp(16, 8), [rn(2, 16, 7, 7), rn(2, 8, 14, 14)], 23)]
    for tag, mod, xs, seed in cases:
        g = golden(tag)
        O.init_weights_portable(mod, seed)
        mod = mod.to(DEV).eval()
        xs = [x.to(DEV).requires_grad_(True) for x in xs]
        y = mod(*xs)
        y.backward(t(g["go"]).to(DEV))
        w = next(mod.parameters())
        assert (y.cpu() - t(g["y"])).abs().max().item() < 2e-5, tag
        assert (xs[0].grad.cpu() - t(g["gx"])).abs().max().item() < 5e-5, tag
        assert (w.grad.cpu() - t(g["gw"])).abs().max().item() < 2e-4, tag


def test_generator_l1_vs_golden_fp32(golden):
    g = golden("generator_fwd")
    A, _ = O.synthetic_pairs(1, seed=11)
    G = O.init_weights_portable(T.GeneratorUNet((3, 256, 256)), seed=3).to(DEV).eval()
    with torch.no_grad():
        fake = G(A.to(DEV)).cpu()
    assert fake.shape == (1, 3, 256, 256) and fake.dtype == torch.float32
    l1 = (fake[:, :, ::8, ::8] - t(g["fake_sub"])).abs().mean().item()
    assert l1 <= 1e-4, l1                                                  # north_star: generator L1 vs reference <= 1e-4
    assert (fake[0, :, 0, :] - t(g["fake_row0"])).abs().max().item() <= 5e-5
    assert abs(fake.abs().mean().item() - float(g["fake_absmean"])) <= 1e-5


def test_discriminator_logits_vs_golden_fp32(golden):
    gg, gd = golden("generator_fwd"), golden("discriminator_fwd")
    A, _ = O.synthetic_pairs(1, seed=11)
    Gc = O.init_weights_portable(O.GeneratorUNet((3, 256, 256)), seed=3).eval()
    with torch.no_grad():
        fake = Gc(A)
    D = O.init_weights_portable(T.Discriminator1((3, 256, 256)), seed=4).to(DEV).eval()
    with torch.no_grad():
        logits = D(fake.to(DEV), A.to(DEV)).cpu()
    assert logits.shape == (1, 1, 16, 16)
    want = t(gd["logits"])                       # portable N(0,0.02) weights / sigma make |logits| ~ 1e5: compare relatively
    assert (logits - want).abs().max().item() <= 2e-5 * want.abs().max().item()


def test_generator_bf16_vs_oracle():
    """bf16 mode: generator output (tanh range) within 2e-2 mean abs of the fp32 oracle on the same weights / inputs"""
    T.set_compute_dtype(torch.bfloat16)
    A, _ = O.synthetic_pairs(2, seed=12)
    Gc = O.init_weights_portable(O.GeneratorUNet((3, 256, 256)), seed=3).eval()
    with torch.no_grad():
        want = Gc(A)
    G = T.GeneratorUNet((3, 256, 256))
    G.load_state_dict(Gc.state_dict())
    G = G.to(DEV).eval()
    with torch.no_grad():
        got = G(A.to(DEV)).cpu()
    l1 = (got - want).abs().mean().item()
    rel = l1 / want.abs().mean().item()
    print(f"bf16 generator L1 {l1:.3e} (relative {rel:.3e})")
    assert rel <= 5e-2, (l1, rel)


def _train_step_compare(golden, dtype, tol_loss, tol_grad):
    g = golden("train_step")
    T.set_compute_dtype(dtype)
    G = O.init_weights_portable(T.GeneratorUNet((3, 256, 256)), seed=61).to(DEV).eval()     # eval: dropout off (as in the golden)
    D = O.init_weights_portable(T.Discriminator1((3, 256, 256)), seed=62).to(DEV).train()
    gb = {k: v.clone() for k, v in G.state_dict().items()}
    db = {k: v.clone() for k, v in D.state_dict().items()}
    A, B = O.synthetic_pairs(1, seed=63)
    ts = T.TrainStep(G, D, compute_dtype=dtype)
    out = ts.step(A.to(DEV), B.to(DEV), neg_idx=g["neg_idx"].tolist())
    torch.cuda.synchronize()
    for k in ("loss_G", "loss_GAN_g", "loss_triplet_patch", "loss_FFT", "loss_D"):
        want = float(g[k])
        assert abs(float(out[k]) - want) <= tol_loss * max(1.0, abs(want)), (k, float(out[k]), want)
    assert (out["fake_B"].cpu()[:, :, ::8, ::8] - t(g["fake_sub"])).abs().mean().item() <= (1e-4 if dtype == torch.float32 else 3e-2)

    def close(got, want, tol):
        want = t(want).double()
        rel = ((got.cpu().double() - want).norm() / want.norm()).item()        # relative L2 error of the gradient tensor
        print(f"  grad rel-L2 error {rel:.3e} (tol {tol})")
        return rel <= tol

    # gradients left in the flat buffers by the step (sum over the single rank)
    assert close(ts.gflat.grad_views["down1.model.0.weight"], g["g_grad_down1"], tol_grad)
    assert close(ts.gflat.grad_views["up3.model.0.weight"][::16, ::16], g["g_grad_up3"], tol_grad)
    assert close(ts.dflat.grad_views["model.13.weight"], g["d_grad_head"], tol_grad)
    assert close(ts.dflat.grad_views["model.0.bias"], g["d_grad_b0"], tol_grad)
    assert close(ts.dflat.grad_views["model.3.parametrizations.weight.original"][::8, ::8], g["d_grad_w3"], tol_grad)
    if dtype == torch.float32:
        # Adam deltas: |delta| ~ lr; sign flips of ~zero gradients are excluded by comparing where the reference moved clearly
        for key, ref in (("final.2.weight", g["g_delta_final_w"]), ("down1.model.0.weight", g["g_delta_down1"])):
            got = (G.state_dict()[key] - gb[key]).cpu()
            assert (got - t(ref)).abs().mean().item() <= 2e-6, key
        got = (D.state_dict()["model.13.weight"] - db["model.13.weight"]).cpu()
        assert (got - t(g["d_delta_head"])).abs().mean().item() <= 2e-6
        assert torch.allclose(D.state_dict()["model.3.parametrizations.weight.0._u"].cpu(), t(g["d_u3"]), atol=1e-4)


def test_train_step_fp32_vs_reference_golden(golden):
    # Gradient tolerance: ReLU / LeakyReLU masks are knife edges. At N=1 one max-magnitude element carries ~5e-3 of a gradient
    # tensor's L2 norm, so a single element with |xhat| < 1e-6 that lands on the other side of zero (summation order differs
    # from torch's) moves every downstream gradient by ~1e-3 (round 2, with float atomics: 6e-4 .. 5e-3 from run to run). Since round 3 the engine's sums
    # have a fixed order, so the number no longer moves from run to run (test_step_is_bit_deterministic_on_one_and_two_streams); the bound stays at the
    # knife-edge scale because any legitimate change of a kernel's summation order lands somewhere in that range. Typical agreement is 6e-4.
    _train_step_compare(golden, torch.float32, 2e-4, 1e-2)


def test_train_step_bf16_vs_reference_golden(golden):
    # bf16 pre-activations flip ~0.3 % of the activation masks per layer relative to the fp32 reference (sqrt(0.003) ~ 5.5 % of the
    # gradient norm per layer, ~18-20 % through the 11 masked generator layers); the discriminator (5 layers) stays at ~1 %.
    _train_step_compare(golden, torch.bfloat16, 3e-2, 0.25)


def test_dropout_train_mode_matches_oracle_with_shared_masks():
    """training-mode generator (dropout p=0.5 in down3/down4/up2/up3): the oracle consumes the very masks the kernels draw"""
    A, _ = O.synthetic_pairs(1, seed=21)
    Gc = O.init_weights_portable(O.GeneratorUNet((3, 256, 256)), seed=8).train()
    core = T.nets.GeneratorCore(T.ops.DT_F32)
    core.set_params({k: v.to(DEV) for k, v in Gc.state_dict().items() if k in T.nets.g_param_names()})
    seed = 4242
    fake, _ = core.forward(A.to(DEV), seed=seed, train=True, save=False)
    Gc.set_mask_fn(O.hip_mask_fn(seed))
    with torch.no_grad():
        want = Gc(A)
    assert (fake.cpu() - want).abs().mean().item() <= 1e-4


def test_full_size_properties_batch32():
    """BASELINE config (batch 32, 256x256, bf16): size-independent properties instead of a (minutes-long) CPU oracle run"""
    T.set_compute_dtype(torch.bfloat16)
    torch.manual_seed(0)
    N = 32
    A, B = O.synthetic_pairs(N, seed=99)
    A, B = A.to(DEV), B.to(DEV)
    G = T.GeneratorUNet((3, 256, 256)).to(DEV)
    D = T.Discriminator1((3, 256, 256)).to(DEV)
    G.apply(T.weights_init_normal)
    D.apply(T.weights_init_normal)
    ts = T.TrainStep(G, D)
    TB = 24 + 14 * torch.randint(0, 256, (N, 256, 256), device=DEV).float() / 255
    B_tf = T.color_jitter_thermal(B, T.color_jitter_params(np.random.default_rng(1)))
    out = ts.step(A, B, T_B=TB, B_tf=B_tf)
    torch.cuda.synchronize()
    fake = out["fake_B"]
    assert fake.shape == (N, 3, 256, 256) and torch.isfinite(fake).all() and fake.abs().max().item() <= 1.0
    # temperature term (gradient-free, P16:587-595 / :607): the engine's value == the oracle's on the same fake_B
    want_t = float(O.temp_triplet_loss(fake[:4].cpu(), TB[:4].cpu(), B_tf[:4].cpu()))
    got_t = T.temperature_triplet_loss(fake[:4], TB[:4], B_tf[:4]).item()
    assert abs(got_t - want_t) <= 1e-4 * abs(want_t) + 1e-6            # fp32 row norms, different summation order
    lg = 0.5 * out["loss_GAN_g"] + out["loss_triplet_patch"] + 0.01 * out["loss_FFT"] + 0.5 * out["loss_temp_g"]
    assert abs(float(lg) - float(out["loss_G"])) <= 1e-4 * abs(float(lg))
    for k, v in out.items():
        if k != "fake_B":
            assert np.isfinite(float(v)), k
    # per-sample independence (InstanceNorm, per-sample losses): sample 5 alone == sample 5 inside the batch (eval: no dropout).
    # The only batch-dependent thing is the ORDER of the fp32 statistics atomics; in bf16 a flipped rounding is amplified by the
    # 12 layers, in fp32 mode the two runs agree to round-off.
    G.eval()
    with torch.no_grad():
        full = G(A)
        one = G(A[5:6])
    assert (full[5:6] - one).abs().mean().item() <= 1e-2
    G.compute_dtype = torch.float32
    with torch.no_grad():
        full = G(A[:6])
        one = G(A[5:6])
    assert (full[5:6] - one).abs().max().item() <= 2e-5
    G.compute_dtype = None
    # adjoint identity  <conv(x), g> == <x, dgrad(g)> == <w, wgrad(x, g)>  at the size of down2 (64 -> 128 @ 128x128, batch 32)
    ops = T.ops
    dt = ops.DT_BF16
    x = ops.new_act(N, 128, 128, 64, dt, DEV)
    x.t.copy_(torch.randn(x.t.shape, device=DEV))
    go = ops.new_act(N, 127, 127, 128, dt, DEV)
    go.t.copy_(torch.randn(go.t.shape, device=DEV))
    w = (torch.randn(128, 64, 4, 4, device=DEV) * 0.03).to(torch.bfloat16).float()
    y = ops.new_act(N, 127, 127, 128, dt, DEV)
    ops.conv_fwd(dt, ops.OP_CONV, x, 64, 128, ops.pack_weight(dt, ops.OP_CONV, 0, w, 64, 128), y)
    dx = ops.new_act(N, 128, 128, 64, dt, DEV)
    ops.conv_dgrad(dt, ops.OP_CONV, go, N, 128, 128, 64, 128, ops.pack_weight(dt, ops.OP_CONV, 1, w, 64, 128), dx)
    dw = torch.empty_like(w)
    ops.conv_wgrad(dt, ops.OP_CONV, x, go, 64, 128, dw)
    a = (y.t.double() * go.t.double()).sum().item()
    b = (x.t.double() * dx.t.double()).sum().item()
    c = (w.double() * dw.double()).sum().item()
    scale = (y.t.double().pow(2).sum().sqrt() * go.t.double().pow(2).sum().sqrt()).item()
    assert abs(a - b) <= 2e-3 * scale and abs(a - c) <= 2e-3 * scale, (a, b, c, scale)
    # the same identity for the two first-layer kernels at full size: weights-stationary conv (6 -> 64 @ 256x256) against the rows-packed
    # input-gradient kernel (gradient w.r.t. the first 3 channels; the other input channels are zero here)
    x6 = torch.zeros((N, 256, 256, 8), device=DEV)
    x6[..., :3] = torch.randn((N, 256, 256, 3), device=DEV)
    x6v = ops.View(x6.to(torch.bfloat16), 6)
    w1 = (torch.randn(64, 6, 4, 4, device=DEV) * 0.1).to(torch.bfloat16).float()
    y1 = ops.new_act(N, 255, 255, 64, dt, DEV)
    ops.conv_fwd(dt, ops.OP_CONV, x6v, 6, 64, ops.pack_weight(dt, ops.OP_CONV, 0, w1, 6, 64), y1)
    g1 = ops.new_act(N, 255, 255, 64, dt, DEV)
    g1.t.copy_(torch.randn(g1.t.shape, device=DEV))
    dx1 = ops.conv_dgrad_image(dt, g1, N, 256, 256, 6, w1, None, 3)                   # fp32 NCHW [N,3,256,256]
    a1 = (y1.t.double() * g1.t.double()).sum().item()
    b1 = (x6v.t[..., :3].double().permute(0, 3, 1, 2) * dx1.double()).sum().item()
    scale1 = (y1.t.double().pow(2).sum().sqrt() * g1.t.double().pow(2).sum().sqrt()).item()
    assert abs(a1 - b1) <= 2e-3 * scale1, (a1, b1, scale1)
    # triplet with r_k = k: exactly the margin; FFT loss of identical images: exactly 0
    l, d = ops.patch16_triplet(fake, B, list(range(16)))
    assert abs(l.item() - 1.0) < 1e-6 and d.abs().max().item() == 0.0
    z, _, _ = T.patch_fft_loss(fake, fake)
    assert z.item() == 0.0


@pytest.mark.parametrize("cdt", [torch.bfloat16, torch.float32])
def test_step_is_bit_deterministic_on_one_and_two_streams(cdt):
    """no float atomics on the training path -- sums that cross workgroups (InstanceNorm statistics and their backward reductions,
    bias gradients, spectral-norm products, split-K weight gradients) are added in a fixed order. Two steps from the same weights and inputs therefore
    give the SAME BITS: both flat gradient buffers, the post-Adam weights, the spectral-norm vectors and the logged losses, run to run on two streams
    (the product default) and against the one-stream schedule (nets.py: the side stream changes WHEN kernels run, never what they compute).
    Round 2 could only assert a cosine here (bf16 generator gradients agreed run to run to cos 0.991)."""
    T.set_compute_dtype(cdt)
    runs = []
    prev = T.set_wgrad_stream(True)
    try:
        for on in (True, True, False):
            T.set_wgrad_stream(on)
            G = O.init_weights_portable(T.GeneratorUNet((3, 256, 256)), seed=71).to(DEV)
            D = O.init_weights_portable(T.Discriminator1((3, 256, 256)), seed=72).to(DEV)
            A, B = O.synthetic_pairs(2, seed=73)
            A, B = A.to(DEV), B.to(DEV)
            ts = T.TrainStep(G, D, compute_dtype=cdt)
            out = ts.step(A, B)
            out2 = ts.step(A, B)                                  # a second step: Adam state, re-packed operand streams, evolved u / v
            torch.cuda.synchronize()
            runs.append({"g_grad": ts.gflat.grad.clone(), "d_grad": ts.dflat.grad.clone(), "g_w": ts.gflat.data.clone(), "d_w": ts.dflat.data.clone(),
                         "fake": out2["fake_B"].clone(), "sn": torch.cat([b.flatten() for b in ts.dbufs.values()]).clone(),
                         "losses": torch.stack([out2[k].reshape(()).float() for k in sorted(out2) if k != "fake_B"]).clone(),
                         "views": {k: v.clone() for k, v in list(ts.gflat.grad_views.items()) + list(ts.dflat.grad_views.items())}})
    finally:
        T.set_wgrad_stream(prev)
        T.set_compute_dtype(torch.float32)
    ref = runs[0]
    assert torch.isfinite(ref["g_grad"]).all() and ref["g_grad"].abs().max().item() > 0 and ref["d_grad"].abs().max().item() > 0
    for what, other in (("two streams, run to run", runs[1]), ("two streams vs one stream", runs[2])):
        bad = [k for k in ref["views"] if not torch.equal(ref["views"][k], other["views"][k])]
        assert not bad, (what, "parameter gradients differ", bad)
        for k in ("g_grad", "d_grad", "g_w", "d_w", "fake", "sn", "losses"):
            assert torch.equal(ref[k], other[k]), (what, k, (ref[k].double() - other[k].double()).abs().max().item())


def test_module_forward_sees_weights_updated_by_trainstep():
    """the module's operand-stream cache is keyed on (data_ptr, _version); the raw-pointer Adam kernel moves neither.
    G(x); ts.step(); G(x) must equal a FRESH module loaded from the same state_dict (P16:395 sample_images calls generator(real_A)
    between optimiser steps), same for D."""
    T.set_compute_dtype(torch.float32)
    G = O.init_weights_portable(T.GeneratorUNet((3, 256, 256)), seed=61).to(DEV).eval()
    D = O.init_weights_portable(T.Discriminator1((3, 256, 256)), seed=62).to(DEV).eval()
    A, B = O.synthetic_pairs(1, seed=63)
    A, B = A.to(DEV), B.to(DEV)
    ts = T.TrainStep(G, D, compute_dtype=torch.float32)
    with torch.no_grad():
        y0, l0 = G(A), D(B, A)                               # packs the module's own operand streams
    ts.step(A, B)
    ts.step(A, B)
    with torch.no_grad():
        y1, l1 = G(A), D(B, A)
    assert (y1 - y0).abs().max().item() > 1e-4               # the weights did move
    G2 = T.GeneratorUNet((3, 256, 256)).to(DEV).eval()
    G2.load_state_dict(G.state_dict())
    D2 = T.Discriminator1((3, 256, 256)).to(DEV).eval()
    D2.load_state_dict(D.state_dict())
    with torch.no_grad():
        y2, l2 = G2(A), D2(B, A)
    assert (y1 - y2).abs().max().item() <= 1e-5, (y1 - y2).abs().max().item()
    assert (l1 - l2).abs().max().item() <= 1e-5 * l2.abs().max().item()


def test_trainstep_sees_load_state_dict():
    """the reverse direction: load_state_dict AFTER TrainStep construction writes the flat buffer; the next step must re-pack its
    operand streams (a resumed run, P16:447-450). Two engines stepping from the same loaded weights must agree."""
    T.set_compute_dtype(torch.float32)
    A, B = O.synthetic_pairs(1, seed=64)
    A, B = A.to(DEV), B.to(DEV)
    neg = list(range(1, 16)) + [0]
    Gs = O.init_weights_portable(T.GeneratorUNet((3, 256, 256)), seed=71)
    Ds = O.init_weights_portable(T.Discriminator1((3, 256, 256)), seed=72)
    gsd = {k: v.clone() for k, v in Gs.state_dict().items()}
    dsd = {k: v.clone() for k, v in Ds.state_dict().items()}
    # engine 1: built on OTHER weights, then loaded
    G1 = O.init_weights_portable(T.GeneratorUNet((3, 256, 256)), seed=5).to(DEV).eval()
    D1 = O.init_weights_portable(T.Discriminator1((3, 256, 256)), seed=6).to(DEV).eval()
    ts1 = T.TrainStep(G1, D1, compute_dtype=torch.float32)
    ts1.step(A, B, neg_idx=neg)                               # streams packed from the seed-5/6 weights
    G1.load_state_dict(gsd)
    D1.load_state_dict(dsd)
    # engine 2: built directly on the loaded weights
    G2 = T.GeneratorUNet((3, 256, 256)); G2.load_state_dict(gsd); G2 = G2.to(DEV).eval()
    D2 = T.Discriminator1((3, 256, 256)); D2.load_state_dict(dsd); D2 = D2.to(DEV).eval()
    ts2 = T.TrainStep(G2, D2, compute_dtype=torch.float32)
    ts1.step_no = 0
    ts1.gm.zero_(); ts1.gv.zero_(); ts1.dm.zero_(); ts1.dv.zero_()
    o1 = ts1.step(A, B, neg_idx=neg)
    o2 = ts2.step(A, B, neg_idx=neg)
    assert (o1["fake_B"] - o2["fake_B"]).abs().max().item() <= 1e-4      # fp32 statistics atomics: summation order differs run to run
    for k in ("loss_G", "loss_D"):
        assert abs(float(o1[k]) - float(o2[k])) <= 1e-4 * max(1.0, abs(float(o2[k]))), k


def test_glo16_train_step_fp32_vs_reference_golden(golden):
    """config C3 (GLO-16, TFCGAN_multigpu_globalFFT_16P.py): TrainStep(fft_mode="global") at N = 1 in exact-fp32 mode against one step of the
    networks and whole-image fft_components lifted from the G16 script itself (G16:294-313, :524-534)"""
    g = golden("train_step_glo16")
    T.set_compute_dtype(torch.float32)
    G = O.init_weights_portable(T.GeneratorUNet((3, 256, 256)), seed=61).to(DEV).eval()
    D = O.init_weights_portable(T.Discriminator1((3, 256, 256)), seed=62).to(DEV).train()
    w0 = G.state_dict()["final.2.weight"].clone()
    A, B = O.synthetic_pairs(1, seed=64)
    ts = T.TrainStep(G, D, compute_dtype=torch.float32, fft_mode="global")
    out = ts.step(A.to(DEV), B.to(DEV), neg_idx=g["neg_idx"].tolist())
    torch.cuda.synchronize()
    for k in ("loss_G", "loss_GAN_g", "loss_triplet_patch", "loss_FFT", "loss_Amp", "loss_Pha", "loss_D"):
        want = float(g[k])
        # phase L1: the phase of a near-zero bin is round-off in ANY arithmetic (fp32 radix-4 FFT here, float64 pocketfft in the reference);
        # on this image that moves the mean |phase difference| by 2e-4 relative -- stated tolerance 5e-4 for the phase term and what contains it
        tol = 5e-4 if k in ("loss_Pha", "loss_FFT") else 2e-4
        assert abs(float(out[k]) - want) <= tol * max(1.0, abs(want)), (k, float(out[k]), want)
    assert (out["fake_B"].cpu()[:, :, ::8, ::8] - t(g["fake_sub"])).abs().mean().item() <= 1e-4
    want = t(g["g_grad_down1"]).double()
    rel = ((ts.gflat.grad_views["down1.model.0.weight"].cpu().double() - want).norm() / want.norm()).item()
    assert rel <= 1e-2, rel
    got = (G.state_dict()["final.2.weight"] - w0).cpu()
    assert (got - t(g["g_delta_final_w"])).abs().mean().item() <= 2e-6


def test_glo16_step_full_size_batch32_bf16():
    """config C3 at BASELINE size (batch 32, 256x256, bf16): the global-FFT step runs on the HIP path; size-independent properties: finite
    losses, loss_G = 0.5 GAN + triplet + 0.01 FFT (G16:534), FFT loss of identical images = 0, the whole-image spectrum of a constant image is
    a single DC bin, and the batched loss equals the mean of per-sample losses (every term is a batch mean)"""
    T.set_compute_dtype(torch.bfloat16)
    torch.manual_seed(0)
    N = 32
    A, B = O.synthetic_pairs(N, seed=98)
    A, B = A.to(DEV), B.to(DEV)
    G = T.GeneratorUNet((3, 256, 256)).to(DEV)
    D = T.Discriminator1((3, 256, 256)).to(DEV)
    G.apply(T.weights_init_normal)
    D.apply(T.weights_init_normal)
    ts = T.TrainStep(G, D, fft_mode="global")
    out = ts.step(A, B)
    torch.cuda.synchronize()
    for k, v in out.items():
        if k != "fake_B":
            assert np.isfinite(float(v)), k
    lg = 0.5 * out["loss_GAN_g"] + out["loss_triplet_patch"] + 0.01 * out["loss_FFT"]
    assert abs(float(lg) - float(out["loss_G"])) <= 1e-4 * abs(float(lg))
    fake = out["fake_B"]
    z, _, _ = T.global_fft_loss(fake, fake)
    assert z.item() == 0.0
    full, la, lp = T.global_fft_loss(fake, B)
    per = torch.stack([T.global_fft_loss(fake[i:i + 1], B[i:i + 1])[0] for i in range(0, N, 8)])
    sub, _, _ = T.global_fft_loss(fake[0:N:8], B[0:N:8])
    assert abs(per.mean().item() - sub.item()) <= 1e-4 * abs(sub.item())
    assert abs(float(out["loss_FFT"]) - full.item()) <= 1e-5 * abs(full.item())
    # the oracle on two of the 32 samples (CPU rfft2 of 256x256 is cheap): amplitude / phase L1 of the engine's own fake_B
    want, wa, wp = O.global_fft_loss(fake[:2].cpu(), B[:2].cpu())
    got, ga, gp = T.global_fft_loss(fake[:2], B[:2])
    assert abs(ga.item() - float(wa)) <= 2e-4 * float(wa) and abs(gp.item() - float(wp)) <= 2e-3 * float(wp) + 1e-4
    const = torch.full((1, 3, 256, 256), 0.5, device=DEV)
    amp, pha = T.fft_components(const, patch=False)
    assert amp.shape == (1, 1, 256, 129)
    dc = amp[0, 0, 128, 64].item()                                 # fftshift over both axes puts DC at (128, 64)
    assert abs(dc - 127 * 256 * 256) < 1 and (amp.sum().item() - dc) < 1e-2 * dc


def test_sample_spectra_and_mse_spec_vs_reference_goldens(golden):
    """FFT_Components.make_spectra / sample_spectra (P16:284-289, :378-388) and the evaluation metric mse_spec (Devcom_MagMSE.py:91-118),
    both against outputs of the reference's own (lifted) functions"""
    g = golden("spectra")
    x, _ = O.synthetic_pairs(2, seed=81)
    x = torch.tanh(x * 1.2) * 0.999 + 1e-3
    spec = T.sample_spectra(x.to(DEV)).cpu()
    assert spec.shape == (2, 1, 256, 256)
    assert (spec[:, :, ::4, ::4] - t(g["spec_sub"])).abs().max().item() <= 2e-3      # log of an fp32 direct-DFT magnitude vs float64 pocketfft
    assert (spec[1, 0, 7, :] - t(g["spec_row7"])).abs().max().item() <= 2e-3
    assert abs(spec.mean().item() - float(g["spec_mean"])) <= 1e-4
    one = T.FFT_Components(x[0].to(DEV)).make_spectra().cpu()
    assert torch.equal(one, spec[0, 0])
    gm = golden("mse_spec")
    rng = np.random.default_rng(91)
    base = rng.integers(1, 256, size=(3, 256, 256)).astype(np.uint8)
    other = np.clip(base.astype(np.int32) + rng.integers(-40, 41, size=base.shape), 1, 255).astype(np.uint8)
    got = T.mse_spec(base, other).cpu().numpy()
    np.testing.assert_allclose(got, gm["values"], rtol=2e-3)
    got_mae = T.other_spec(base, other).cpu().numpy()              # eval/Eurecom/Eurecom_MagOther.py:90-118, lifted the same way
    np.testing.assert_allclose(got_mae, gm["mae_values"], rtol=2e-3)


def _nchw(v):
    return v.t[..., v.coff:v.coff + v.C].float().cpu().permute(0, 3, 1, 2).contiguous()


def _rel(got, want):
    return ((got.double() - want.double()).norm() / want.double().norm().clamp_min(1e-30)).item()


class _EngineStatsNorm(torch.autograd.Function):
    """nn.InstanceNorm2d (P16:107, :124) evaluated with the ENGINE's statistics: x_hat = (x - mean) * rstd with (mean, rstd) derived from the engine's
    (sum, sum of squares) exactly as its kernels derive them (fp32: m = s1 * (1 / HW), var = max(s2 / HW - m^2, 0), rstd = rsqrt(var + eps)); backward =
    the InstanceNorm backward with those same constants. Teacher-forcing the statistics takes the one knife edge out of the per-layer comparison:
    sign(x - mean) -- the LeakyReLU / ReLU mask -- is then decided by identical fp32 numbers on both sides (the 5e-3..8e-3 tail of the
    7 x 7 layers was ONE mask flip, 0.8 |g| / sqrt(25088), caused by a last-bit difference of the mean)."""

    @staticmethod
    def forward(ctx, x, stats, eps):
        hw = x.shape[2] * x.shape[3]
        inv = torch.tensor(1.0 / hw, dtype=torch.float32)
        m = (stats[..., 0] * inv)[:, :, None, None]
        var = (stats[..., 1] * inv)[:, :, None, None] - m * m
        r = torch.rsqrt(var.clamp_min(0.0) + eps)
        xh = (x - m) * r
        ctx.save_for_backward(xh, r)
        return xh

    @staticmethod
    def backward(ctx, g):
        xh, r = ctx.saved_tensors
        mg = g.mean(dim=(2, 3), keepdim=True)
        mgx = (g * xh).mean(dim=(2, 3), keepdim=True)
        return r * (g - mg - xh * mgx), None, None


@pytest.mark.parametrize("net", ["generator", "discriminator"])
def test_fused_first_block_forward_vs_storage_oracle(net):
    """The per-layer tests below run with debug taps, i.e. on the UNFUSED first block; the product runs tfc_first_block_fwd (conv + LeakyReLU + BlurPool in one
    kernel, the conv output never stored). Here that kernel itself is held against the oracle's bf16-storage model of the block: the pooled output within the
    forward bound of the per-layer tests (5e-4 rel-L2), and its sign words against the sign of the oracle's own (bf16-rounded) conv output -- they may differ
    only where the two fp32 summation orders round a value across zero (none observed; a handful allowed). Generator form: LeakyReLU after the rounding of
    the raw conv output (P16:105-109); discriminator form: bias, 1/sigma and LeakyReLU before it (P16:187-190)."""
    from tfc_gan_amd import ops
    T.set_compute_dtype(torch.bfloat16)
    torch.set_num_threads(16)
    dt = ops.DT_BF16
    N, S = 2, 256
    A, B = O.synthetic_pairs(N, seed=71)
    rb, rw = O._RoundBoth.apply, O._RoundFwd.apply
    if net == "generator":
        Gc = O.init_weights_portable(O.GeneratorUNet((3, S, S)), seed=61).eval()
        w = Gc.down1.model[0].weight.detach()
        x = O._bf(A)
        cin, bias, inv_sigma = 3, None, None
        with torch.no_grad():
            z = rb(F.conv2d(x, rw(w), padding=1))
            want = rb(O._blur(F.leaky_relu(z, 0.2), 2))
        xin = ops.pack_nhwc8(dt, A.to(DEV))
    else:
        Dc = O.init_weights_portable(O.Discriminator1((3, S, S)), seed=62).train()
        w = Dc.model[0].parametrizations.weight.original.detach()
        bias = Dc.model[0].bias.detach()
        inv_sigma = 0.37                                            # any positive scale: the kernel takes 1/sigma as a device scalar
        x = O._bf(torch.cat((A, B), 1))
        cin = 6
        with torch.no_grad():
            z = rb(F.leaky_relu(F.conv2d(x, rw(w), padding=1) * inv_sigma + bias.view(1, -1, 1, 1), 0.2))
            want = rb(O._blur(z, 2))
        xin = ops.pack_nhwc8(dt, A.to(DEV), B.to(DEV))
    pk = ops.pack_weight(dt, ops.OP_CONV, 0, w.to(DEV), cin, 64)
    Po = want.shape[2]
    got = ops.new_act(N, Po, Po, 64, dt, DEV)
    mask = torch.zeros((N, S - 1, S - 1, 8), dtype=torch.uint8, device=DEV)
    ops.first_block_fwd(dt, xin, cin, 64, pk, got, bias=None if bias is None else bias.to(DEV),
                        oscale=None if inv_sigma is None else torch.tensor([inv_sigma], device=DEV), slope=0.2, act_after_rounding=(net == "generator"),
                        sign_mask=mask)
    torch.cuda.synchronize()
    r = _rel(_nchw(got), want)
    print(f"  {net} first block (fused forward) rel-L2 {r:.3e}")
    assert r <= 5e-4, r
    bits = np.unpackbits(mask.cpu().numpy().reshape(N, S - 1, S - 1, 8), axis=3, bitorder="little")       # bit c of the pixel's 64-bit word = channel c
    want_bits = (z > 0).permute(0, 2, 3, 1).numpy().astype(np.uint8)
    nbad = int((bits != want_bits).sum())
    print(f"  sign words: {nbad} of {bits.size} bits differ from the oracle's")
    assert nbad <= 1e-5 * bits.size, nbad


def test_bf16_layers_teacher_forced_vs_storage_oracle():
    """bf16 is the benchmarked dtype, and end to end a bf16 network is chaotic: a 1-ulp tie broken differently (different fp32 summation order)
    changes 1 % of the next layer's roundings, 18 % after two more layers, everything after six -- so NO independent implementation, however
    faithful, agrees end to end to better than ~1e-2 (scripts/debug_bf16_layers.py prints the growth). What CAN be tight is every layer on its
    own: each layer of the bf16 engine is fed to the oracle's bf16-STORAGE model of that layer (round to bf16 where the engine stores bf16, fp32 in
    between: oracle._RoundBoth / _RoundFwd / _RoundBwd) with the ENGINE's stored input AND the engine's InstanceNorm statistics (_EngineStatsNorm: the
    activation mask is then decided by identical numbers on both sides), forward and backward: outputs, input gradients and weight gradients of all
    12 generator layers must agree to 1-ulp ties (rel-L2 <= 5e-4 forward, <= 5e-3 backward, NO per-layer exceptions). A 10 % error in any bf16-only
    kernel (first-layer conv, pooled activation, up-conv head, transposed-conv phases, fused wgrads) fails here by an order of magnitude. Every
    weight-gradient line also reports how many elements of the engine's stored conv-output gradient differ from the oracle's, and by how much, so a
    miss is explained by the one run that shows it."""
    T.set_compute_dtype(torch.bfloat16)
    torch.set_num_threads(16)
    Gc = O.init_weights_portable(O.GeneratorUNet((3, 256, 256)), seed=61).eval()
    A, B = O.synthetic_pairs(1, seed=63)
    core = T.nets.GeneratorCore(T.ops.DT_BF16)
    params = {k: v.to(DEV) for k, v in Gc.state_dict().items() if k in T.nets.g_param_names()}
    core.set_params(params)
    core.debug = {}
    fake, ctx = core.forward(A.to(DEV), seed=0, train=False, save=True)
    rng = np.random.default_rng(5)
    g_fake = torch.from_numpy(rng.standard_normal((1, 3, 256, 256)).astype(np.float32)) * 1e-3
    grads = {k: torch.zeros_like(v) for k, v in params.items()}
    core.backward(ctx, g_fake.to(DEV), grads)
    torch.cuda.synchronize()
    dbg = core.debug
    rb, rw, rg = O._RoundBoth.apply, O._RoundFwd.apply, O._RoundBwd.apply
    FWD, BWD = 5e-4, 5e-3           # observed: forward <= 9.4e-5, backward <= 3.0e-3 (the dgrads that round an accumulated skip gradient)
    worst = [0.0, 0.0]

    def check(tag, got, want, tol, slot):
        r = _rel(got, want)
        worst[slot] = max(worst[slot], r)
        print(f"  {tag:34s} rel-L2 {r:.3e}")
        assert r <= tol, (tag, r)

    def draw_report(tag, got, want):
        """engine's stored conv-output gradient vs the oracle's (bf16-rounded): count / size of the differences"""
        d = (got.double() - want.double()).abs()
        scale = want.double().abs().max().item() + 1e-30
        n_diff = int((d > 0).sum())
        n_big = int((d > 0.02 * want.double().abs().clamp_min(1e-3 * scale)).sum())      # more than a few bf16 ulps: a flipped mask, not a tie
        print(f"  {tag:34s} d_raw: {n_diff} of {d.numel()} elements differ, {n_big} by more than 2 %, max |diff| / max |want| = {d.max().item() / scale:.2e}")
        return n_big

    # ---- down path: layer i maps its stored input to raw (conv) and to the pooled output; backward from the engine's own g_out ----
    skip_of = {4: "up1", 3: "up2", 2: "up3", 1: "up4", 0: "up5"}
    x_in = _nchw(ctx.x8)[:, :3]
    pooled = []
    for i, (name, _, cout, norm, _) in enumerate(O._DOWNS):
        w = getattr(Gc, name).model[0].weight.detach().clone().requires_grad_(True)
        x = x_in.clone().requires_grad_(i > 0)
        z = rb(F.conv2d(x, rw(w), padding=1))
        check(f"{name} conv fwd", _nchw(ctx.raw[i]), z.detach(), FWD, 0)
        raw_e = _nchw(ctx.raw[i])
        zt = z + (raw_e - z).detach()                               # teacher forcing: continue from the ENGINE's stored conv output
        zz = _EngineStatsNorm.apply(zt, ctx.stats[i].float().cpu(), 1e-5) if norm else zt
        y = rb(O._blur(F.leaky_relu(zz, 0.2), 2))
        if i < 5:
            cat = ctx.cat[skip_of[i]]
            y_e = cat.t[..., cat.t.shape[3] - cout:].float().cpu().permute(0, 3, 1, 2).contiguous()
        else:
            y_e = _nchw(ctx.d6)
        check(f"{name} norm/act/pool fwd", y_e, y.detach(), FWD, 0)
        g_out = _nchw(dbg[f"{name}.g_out"])
        ins = [w] + ([x] if i > 0 else [])
        gr = torch.autograd.grad(y, ins + [zt], g_out)
        if f"{name}.d_raw" in dbg:
            assert draw_report(f"{name}", _nchw(dbg[f"{name}.d_raw"]), O._bf(gr[-1])) == 0, name
        check(f"{name} wgrad", grads[f"{name}.model.0.weight"].cpu(), gr[0], BWD, 1)
        if i > 0:
            # the engine ACCUMULATES this input gradient onto the up path's part of the skip window: g_out(prev) = bf16(g_in(up)[skip] + dgrad)
            up = skip_of[i - 1]
            prev_cout = O._DOWNS[i - 1][2]
            # gradient of the concat buffer cat[up] (= output of `up`, input of the NEXT up block): the next block's g_in; cat[up5] feeds the head
            nxt_up = {"up1": "up2", "up2": "up3", "up3": "up4", "up4": "up5"}
            g_up = _nchw(dbg[f"{nxt_up[up]}.g_in"]) if up != "up5" else _nchw(dbg["g_u5"])
            skip_part = g_up[:, g_up.shape[1] - prev_cout:]
            want_prev = O._bf(skip_part + gr[1])
            check(f"{name} dgrad (+skip accumulate)", _nchw(dbg[f"{O._DOWNS[i - 1][0]}.g_out"]), want_prev, BWD, 1)
        pooled.append(y_e)
        x_in = y_e
    # ---- up path ----
    h_in = pooled[5]
    for j, (name, _, cout, _) in enumerate(O._UPS):
        w = getattr(Gc, name).model[0].weight.detach().clone().requires_grad_(True)
        x = h_in.clone().requires_grad_(True)
        zT = rb(F.conv_transpose2d(x, rw(w), stride=2, padding=1))
        zb = rb(O._blur(zT, 1))
        check(f"{name} convT+blur fwd", _nchw(ctx.blur[j]), zb.detach(), FWD, 0)
        zt = zb + (_nchw(ctx.blur[j]) - zb).detach()
        y = rb(F.relu(_EngineStatsNorm.apply(zt, ctx.bstats[j].float().cpu(), 1e-5)))
        cat = ctx.cat[name]
        y_e = cat.t[..., :cout].float().cpu().permute(0, 3, 1, 2).contiguous()
        check(f"{name} norm/relu fwd", y_e, y.detach(), FWD, 0)
        g_cat = _nchw(dbg[f"{name}.g_out"])
        gw, gx, gzb = torch.autograd.grad(y, [w, x, zt], g_cat[:, :cout])
        assert draw_report(f"{name} (d_blur)", _nchw(dbg[f"{name}.d_blur"]), O._bf(gzb)) == 0, name
        # (until the blur kernels took the InstanceNorm sums from the STORED bf16 values, up1 -- 8 x 8 planes -- sat at 0.5..1.2e-2 here)
        check(f"{name} wgrad", grads[f"{name}.model.0.weight"].cpu(), gw, BWD, 1)
        check(f"{name} dgrad", _nchw(dbg[f"{name}.g_in"]), O._bf(gx), BWD, 1)
        h_in = cat.t.float().cpu().permute(0, 3, 1, 2).contiguous()   # whole concat buffer = next input
    # ---- head: upsample + pad + conv + tanh (weights of collapsed taps are summed in fp32 and rounded ONCE: compare in that arithmetic) ----
    conv = Gc.final[2]
    w = conv.weight.detach().clone().requires_grad_(True)
    b = conv.bias.detach().clone().requires_grad_(True)
    x = h_in.clone().requires_grad_(True)
    pre = rg(F.conv2d(F.pad(F.interpolate(x, scale_factor=2), (1, 0, 1, 0)), w, b, padding=1))
    out = torch.tanh(pre)
    l1 = (fake.cpu() - out.detach()).abs().mean().item()
    print(f"  final (fp32 weights, taps rounded after collapsing in the engine): L1 {l1:.3e}")
    assert l1 <= 1.5e-3
    gw, gb, gx = torch.autograd.grad(out, [w, b, x], g_fake)
    check("final wgrad", grads["final.2.weight"].cpu(), gw, BWD, 1)
    check("final bias grad", grads["final.2.bias"].cpu(), gb, BWD, 1)
    check("final dgrad", _nchw(dbg["g_u5"]), O._bf(gx), 1e-2, 1)
    print("worst forward", worst[0], "worst backward", worst[1])


def test_bf16_discriminator_layers_teacher_forced_vs_storage_oracle():
    """the discriminator (P16:182-211) is 4 of the 5 network forwards of a step and runs bf16-only kernels the generator never touches:
    the TFC_EP_LEAKY epilogue with 1/sigma and bias, the 6 -> 64 first-layer kernel (tfc_conv_c8_kernel), the pure-blur pooling, the wave-per-pixel head
    (tfc_head_fwd_kernel), the rows-packed image gradient (tfc_dgrad_rows4_kernel), the per-image bias-gradient sums, the spectral-norm backward, and --
    in the discriminator step, where nothing needs the image gradient -- the fused first-block backward (tfc_wgrad_c8_fused_kernel). Every one of the 4
    spectrally normalised blocks and the head is fed to the oracle's bf16-storage model of that block with the ENGINE's stored input, the engine's
    (u, v) snapshot and the engine's incoming gradient: forward <= 5e-4, backward <= 5e-3 rel-L2, the stored conv-output gradients element by element."""
    T.set_compute_dtype(torch.bfloat16)
    torch.set_num_threads(16)
    Dc = O.init_weights_portable(O.Discriminator1((3, 256, 256)), seed=62).train()
    A, B = O.synthetic_pairs(1, seed=63)
    fake = torch.tanh(B * 0.7 + 0.3 * A)                              # a "generated" image: the argument whose gradient the generator step needs
    names = T.nets.d_param_names()
    sd = Dc.state_dict()
    params = {k: sd[k].clone().to(DEV) for k in names}
    bufs = {k: sd[k].clone().to(DEV) for k in sd if k.endswith("._u") or k.endswith("._v")}
    rb, rw, rg = O._RoundBoth.apply, O._RoundFwd.apply, O._RoundBwd.apply
    FWD, BWD = 5e-4, 5e-3
    rng = np.random.default_rng(9)
    g_log = torch.from_numpy(rng.standard_normal((1, 16, 16)).astype(np.float32)) * 1e-2

    def run(debug, need_input_grad):
        core = T.nets.DiscriminatorCore(T.ops.DT_BF16)
        core.set_params({k: v.clone() for k, v in params.items()}, {k: v.clone() for k, v in bufs.items()})
        if debug:
            core.debug = {}
        logits, ctx = core.forward(fake.to(DEV), A.to(DEV), power_iter=True, save=True)
        gl = T.ops.new_act(1, 16, 16, 8, T.ops.DT_BF16, DEV, zero=True)
        gl.t[..., 0] = g_log.to(DEV).to(torch.bfloat16)
        grads = {k: torch.full_like(v, 7.0) for k, v in params.items()}          # overwritten, not accumulated
        gimg = core.backward(ctx, gl, grads, need_input_grad=need_input_grad)
        torch.cuda.synchronize()
        return core, logits, ctx, grads, gimg

    core, logits, ctx, grads, gimg = run(True, True)
    dbg = core.debug
    worst = [0.0, 0.0]

    def check(tag, got, want, tol, slot):
        r = _rel(got, want)
        worst[slot] = max(worst[slot], r)
        print(f"  {tag:40s} rel-L2 {r:.3e}")
        assert r <= tol, (tag, r)

    glb = g_log.to(torch.bfloat16).float().reshape(1, 1, 16, 16)
    convs = [m for m in Dc.model if isinstance(m, torch.nn.Conv2d) and hasattr(m, "parametrizations")]
    want_first = {}
    for bi, (i, cin, cout) in enumerate(T.nets.D_BLOCKS):
        m = convs[bi]
        W = m.parametrizations.weight.original.detach().clone().requires_grad_(True)
        b = m.bias.detach().clone().requires_grad_(True)
        u, v, sig2 = (t.float().cpu() for t in ctx.sn[bi])
        x = _nchw(ctx.ins[bi])[:, :cin].clone().requires_grad_(True)
        sigma = torch.dot(u, W.flatten(1) @ v)                        # the engine's (u, v): sigma as torch's parametrization evaluates it (u, v constants)
        assert abs(sigma.item() - sig2[0].item()) <= 2e-6 * abs(sigma.item()), (sigma.item(), sig2[0].item())
        z = rg(F.conv2d(x, rw(W), padding=1) / sigma + b.view(1, -1, 1, 1))
        h = rw(F.leaky_relu(z, 0.2))
        raw_e = _nchw(ctx.raw[bi])
        check(f"block{bi} SN-conv+bias+leaky fwd", raw_e, h.detach(), FWD, 0)
        ht = h + (raw_e - h).detach()                                 # teacher forcing: continue from the ENGINE's stored activation
        y = rb(O._blur(ht, 2))
        y_e = _nchw(ctx.ins[bi + 1]) if bi < 3 else _nchw(ctx.p4)
        check(f"block{bi} blur-pool fwd", y_e, y.detach(), FWD, 0)
        g_out = _nchw(dbg[f"b{bi}.g_out"])
        gW, gb, gx, gz = torch.autograd.grad(y, [W, b, x, z], g_out)
        d = (_nchw(dbg[f"b{bi}.d_raw"]).double() - O._bf(gz).double()).abs()
        n_big = int((d > 0.02 * gz.double().abs().clamp_min(1e-3 * gz.abs().max().item())).sum())
        print(f"  block{bi} d_raw: {int((d > 0).sum())} of {d.numel()} elements differ, {n_big} by more than 2 %")
        assert n_big == 0, (bi, n_big)
        check(f"block{bi} wgrad (+ spectral-norm bwd)", grads[f"model.{i}.parametrizations.weight.original"].cpu(), gW, BWD, 1)
        check(f"block{bi} bias grad", grads[f"model.{i}.bias"].cpu(), gb, BWD, 1)
        if bi > 0:
            check(f"block{bi} dgrad", _nchw(dbg[f"b{bi - 1}.g_out"]), O._bf(gx), BWD, 1)
        else:
            check("block0 image gradient (rows-packed kernel)", gimg.cpu(), gx[:, :3], BWD, 1)
            want_first = {"w": gW, "b": gb}
    head = Dc.model[-1]
    Wh = head.weight.detach().clone().requires_grad_(True)
    xh = _nchw(ctx.p4).clone().requires_grad_(True)
    lg = rb(F.conv2d(F.pad(xh, (1, 0, 1, 0)), Wh, padding=1))
    check("head fwd", logits.t[..., 0].float().cpu().reshape(1, 1, 16, 16), lg.detach(), FWD, 0)
    gWh, gxh = torch.autograd.grad(lg, [Wh, xh], glb)
    check("head wgrad", grads["model.13.weight"].cpu(), gWh, BWD, 1)
    check("head dgrad", _nchw(dbg["b3.g_out"]), O._bf(gxh), BWD, 1)
    # the discriminator step's form: no image gradient -> block 0 runs the FUSED first-block backward (transposed blur + LeakyReLU' + weight / bias gradient)
    _, _, _, grads2, none = run(False, False)
    assert none is None
    check("block0 wgrad, fused first-block backward", grads2["model.0.parametrizations.weight.original"].cpu(), want_first["w"], BWD, 1)
    check("block0 bias grad, fused first-block backward", grads2["model.0.bias"].cpu(), want_first["b"], BWD, 1)
    for k in grads:                                                    # and the two forms agree with each other to round-off
        assert _rel(grads2[k].cpu(), grads[k].cpu()) <= 2e-5, k
    print("worst forward", worst[0], "worst backward", worst[1])


def test_train_step_bf16_vs_bf16_storage_oracle():
    """whole step, bf16 engine against the oracle's bf16-storage mode: the losses (2e-3) and the end-to-end numbers, stated for what they are: the
    generator output and the gradients carry the chaotic 1-ulp divergence of twelve bf16 layers (see the teacher-forced test above for the tight,
    per-layer statement), the 5-layer discriminator stays within a few %"""
    T.set_compute_dtype(torch.bfloat16)
    Gc = O.init_weights_portable(O.GeneratorUNet((3, 256, 256)), seed=61).eval()
    Dc = O.init_weights_portable(O.Discriminator1((3, 256, 256)), seed=62).train()
    A, B = O.synthetic_pairs(1, seed=63)
    neg = [3, 3, 7, 0, 4, 9, 15, 2, 8, 8, 1, 12, 5, 13, 6, 10]
    torch.set_num_threads(16)
    want = O.bf16_storage_step(Gc, Dc, A, B, neg)
    G = T.GeneratorUNet((3, 256, 256)); G.load_state_dict(Gc.state_dict()); G = G.to(DEV).eval()
    D = T.Discriminator1((3, 256, 256)); D.load_state_dict(Dc.state_dict()); D = D.to(DEV).train()
    ts = T.TrainStep(G, D, compute_dtype=torch.bfloat16)
    out = ts.step(A.to(DEV), B.to(DEV), neg_idx=neg)
    torch.cuda.synchronize()
    for k in ("loss_G", "loss_GAN_g", "loss_triplet_patch", "loss_FFT", "loss_D"):
        w = float(want[k])
        assert abs(float(out[k]) - w) <= 3e-3 * max(1.0, abs(w)), (k, float(out[k]), w)
    l1 = (out["fake_B"].cpu() - want["fake_B"]).abs().mean().item()
    print(f"bf16 engine vs bf16-storage oracle: generator L1 {l1:.3e}")
    assert l1 <= 8e-3
    for k, g in want["d_grads"].items():
        rel = _rel(ts.dflat.grad_views[k].cpu(), g)
        print(f"  D {k:48s} rel-L2 {rel:.3e}")
        assert rel <= 2e-2, (k, rel)                            # observed <= 5.7e-3
