"""The CPU oracle against the fixtures produced by the reference's own (ast-lifted) definitions -- tests/golden/make_golden.py.
This is what pins the oracle; the GPU parity tests then compare the HIP path with the oracle and with the same fixtures."""
import numpy as np
import torch
import torch.nn as nn

from oracle import tfcgan_oracle as O


def t(a):
    return torch.from_numpy(np.asarray(a))


def test_patch_index_map_bit_exact(golden):
    g = golden("patch_index_map")
    B = torch.arange(3 * 256 * 256, dtype=torch.float32).reshape(1, 3, 256, 256)
    first = [int(p[0, 0, 0, 0]) for p in O.make_16_patches(B)]
    assert first == g["first_flat"].tolist()
    assert first[:5] == [0, 64, 128, 192, 16384] and first[-1] == 49344
    assert [f // 256 for f in first] == g["row"].tolist() and [f % 256 for f in first] == g["col"].tolist()


def _block_case(golden, tag, mod, shapes, seed):
    g = golden(tag)
    O.init_weights_portable(mod, seed)
    mod.eval()
    rng = np.random.default_rng(7)
    return g, rng


def test_blocks_match_lifted_reference(golden):
    rng = np.random.default_rng(7)
    rn = lambda *s: torch.from_numpy(rng.standard_normal(s).astype(np.float32))  # noqa: E731
    cases = [("block_down_norm", O.UNetDown(8, 16), [rn(2, 8, 16, 16)], 21),
             ("block_down_nonorm", O.UNetDown(8, 16, normalize=False), [rn(2, 8, 15, 15)], 22),
             ("block_up", O.UNetUp(16, 8), [rn(2, 16, 7, 7), rn(2, 8, 14, 14)], 23)]
    for tag, mod, xs, seed in cases:
        g = golden(tag)
        O.init_weights_portable(mod, seed)
        mod.eval()
        xs = [x.clone().requires_grad_(True) for x in xs]
        y = mod(*xs)
        y.backward(t(g["go"]))
        w = next(mod.parameters())
        assert torch.allclose(y, t(g["y"]), atol=1e-6), tag
        assert torch.allclose(xs[0].grad, t(g["gx"]), atol=1e-6), tag
        assert torch.allclose(w.grad, t(g["gw"]), atol=1e-5), tag


def test_state_dict_contract(golden):
    g = golden("state_dict_keys")
    G, D = O.GeneratorUNet((3, 256, 256)), O.Discriminator1((3, 256, 256))
    assert list(G.state_dict().keys()) == g["g_keys"].tolist()
    assert list(D.state_dict().keys()) == g["d_keys"].tolist()
    assert [str(tuple(v.shape)) for v in G.state_dict().values()] == g["g_shapes"].tolist()
    assert [str(tuple(v.shape)) for v in D.state_dict().values()] == g["d_shapes"].tolist()
    assert sum(p.numel() for p in G.parameters()) == 29238275 and sum(p.numel() for p in D.parameters()) == 2767808


def test_networks_forward(golden):
    gg, gd = golden("generator_fwd"), golden("discriminator_fwd")
    A, _ = O.synthetic_pairs(1, seed=11)
    G = O.init_weights_portable(O.GeneratorUNet((3, 256, 256)), seed=3).eval()
    with torch.no_grad():
        fake = G(A)
    assert torch.allclose(fake[:, :, ::8, ::8], t(gg["fake_sub"]), atol=2e-6)
    assert torch.allclose(fake[0, :, 0, :], t(gg["fake_row0"]), atol=2e-6)
    assert abs(float(fake.abs().mean()) - float(gg["fake_absmean"])) < 1e-6
    D = O.init_weights_portable(O.Discriminator1((3, 256, 256)), seed=4).eval()
    with torch.no_grad():
        logits = D(fake, A)
    assert logits.shape == (1, 1, 16, 16)                                       # P16:95-96
    assert torch.allclose(logits, t(gd["logits"]), atol=1e-5)


def test_triplet_head(golden):
    g = golden("triplet16")
    fk, rl = O.synthetic_pairs(2, seed=31)
    fk = torch.tanh(fk * 1.5).requires_grad_(True)
    neg = g["neg_idx"].tolist()
    loss = O.patch_triplet_loss(fk, rl, neg)
    loss.backward()
    assert abs(float(loss) - float(g["loss"])) < 1e-6
    assert torch.allclose(fk.grad[:, :, ::4, ::4], t(g["gfake_sub"]), atol=1e-9)
    for k in (4, 8, 13):                                                        # r_k == k -> exactly the margin
        assert abs(float(g["per_patch"][k]) - 1.0) < 1e-6


def test_fft_head(golden):
    gp, gg = golden("fft_patch"), golden("fft_global")
    ff, rr = O.synthetic_pairs(1, seed=41)
    ff = torch.tanh(ff * 2.0) * 0.999
    amp, pha = O.fft_components(O.make_16_patches(ff)[5])
    assert amp.shape == (1, 1, 64, 33)
    assert torch.equal(amp, t(gp["amp5"])) and torch.equal(pha, t(gp["pha5"]))
    loss, _, _ = O.patch_fft_loss(ff, rr)
    assert abs(float(loss) - float(gp["loss_fft"])) < 1e-3 * abs(float(gp["loss_fft"])) * 1e-3 + 1e-4
    ga, gph = O.fft_components(ff, patch=False)
    assert torch.equal(ga[:, :, ::4, ::3], t(gg["amp_sub"])) and torch.equal(gph[:, :, ::4, ::3], t(gg["pha_sub"]))
    gl, _, _ = O.global_fft_loss(ff, rr)
    assert abs(float(gl) - float(gg["loss_fft"])) < 1e-4
    # uint8 wrap-around of ToPILImage on negatives
    q = O.to_pil_uint8(torch.tensor([-1.0, -0.5, -0.004, 0.999, 0.5]).reshape(1, 1, 5).repeat(3, 1, 1))
    assert q[0, :, 0].tolist() == [1, 129, 255, 254, 127]


def test_temperature_head(golden):
    g = golden("temp_head")
    assert np.array_equal(g["lut"], O.TEMP_T)                                   # P16:256-257 dict values, float64 bit-exact
    tf_, TB, b_tf = O.temp_head_case(71)
    tfb = O.vectorize_temps(tf_)
    assert tfb.shape == (2, 1, 256, 256)
    assert torch.equal(tfb[:, :, ::8, ::8], t(g["tfb_sub"]))
    assert abs(float(tfb.mean()) - float(g["tfb_mean"])) < 1e-5
    assert abs(float(O.temp_triplet_loss(tf_, TB, b_tf)) - float(g["loss_temp_g"])) < 1e-5 * abs(float(g["loss_temp_g"]))


def test_spectral_norm(golden):
    g = golden("spectral_norm")
    D = O.init_weights_portable(O.Discriminator1((3, 256, 256)), seed=5)
    m = D.model[3]
    W = m.parametrizations.weight.original.detach()
    u = m.parametrizations.weight[0]._u.clone()
    v = m.parametrizations.weight[0]._v.clone()
    for it in range(1, 5):
        u, v, sigma = O.spectral_norm_step(W, u, v)
        if it in (1, 4):
            assert torch.allclose(u, t(g[f"u{it}"]), atol=1e-6) and torch.allclose(v, t(g[f"v{it}"]), atol=1e-6)
            assert abs(float(sigma) - float(g[f"sigma{it}"])) < 1e-6


def test_bce(golden):
    g = golden("bce_relativistic")
    a, b = t(g["a"]), t(g["b"])
    assert abs(float(O.loss_gan_generator(a, b)) - float(g["loss_g"])) < 1e-6
    assert abs(float(O.loss_discriminator(a, b)) - float(g["loss_d"])) < 1e-6


def test_adam_matches_torch():
    rng = np.random.default_rng(0)
    p = torch.from_numpy(rng.standard_normal(100).astype(np.float32))
    q = nn.Parameter(p.clone())
    opt = torch.optim.Adam([q], lr=2e-4, betas=(0.5, 0.999))
    m = torch.zeros(100); v = torch.zeros(100); pp = p.clone()
    for step in (1, 2, 3):
        g = torch.from_numpy(rng.standard_normal(100).astype(np.float32))
        q.grad = g.clone()
        opt.step()
        pp, m, v = O.adam_step(pp, g, m, v, step)
        assert torch.allclose(pp, q.detach(), atol=1e-7)


def test_train_step(golden):
    g = golden("train_step")
    G = O.init_weights_portable(O.GeneratorUNet((3, 256, 256)), seed=61).eval()
    D = O.init_weights_portable(O.Discriminator1((3, 256, 256)), seed=62).train()
    gb = {k: v.clone() for k, v in G.state_dict().items()}
    db = {k: v.clone() for k, v in D.state_dict().items()}
    A, B = O.synthetic_pairs(1, seed=63)
    out = O.TrainStep(G, D).step(A, B, g["neg_idx"].tolist())
    for k in ("loss_G", "loss_GAN_g", "loss_triplet_patch", "loss_FFT", "loss_D"):
        assert abs(float(out[k]) - float(g[k])) <= 1e-5 * max(1.0, abs(float(g[k]))), k
    assert torch.allclose(out["fake_B"][:, :, ::8, ::8], t(g["fake_sub"]), atol=2e-6)
    assert torch.allclose(G.state_dict()["final.2.weight"] - gb["final.2.weight"], t(g["g_delta_final_w"]), atol=1e-6)
    assert torch.allclose(G.state_dict()["down1.model.0.weight"] - gb["down1.model.0.weight"], t(g["g_delta_down1"]), atol=1e-6)
    assert torch.allclose(D.state_dict()["model.13.weight"] - db["model.13.weight"], t(g["d_delta_head"]), atol=1e-6)
    assert torch.allclose(D.state_dict()["model.3.parametrizations.weight.0._u"], t(g["d_u3"]), atol=1e-5)


def test_glo16_train_step_matches_lifted_reference(golden):
    """config C3: the oracle's TrainStep(fft_mode="global") against one step of the networks / fft_components lifted from
    TFCGAN_multigpu_globalFFT_16P.py itself (G16:294-313, :524-534)"""
    g = golden("train_step_glo16")
    torch.set_num_threads(8)
    G = O.init_weights_portable(O.GeneratorUNet((3, 256, 256)), seed=61).eval()
    D = O.init_weights_portable(O.Discriminator1((3, 256, 256)), seed=62).train()
    A, B = O.synthetic_pairs(1, seed=64)
    w0 = G.state_dict()["final.2.weight"].clone()
    out = O.TrainStep(G, D, fft_mode="global").step(A, B, g["neg_idx"].tolist())
    for k in ("loss_G", "loss_GAN_g", "loss_triplet_patch", "loss_FFT", "loss_Amp", "loss_Pha", "loss_D"):
        assert abs(float(out[k]) - float(g[k])) <= 1e-5 * max(1.0, abs(float(g[k]))), k
    assert torch.allclose(out["fake_B"][:, :, ::8, ::8], t(g["fake_sub"]), atol=1e-6)
    assert torch.allclose(G.state_dict()["final.2.weight"] - w0, t(g["g_delta_final_w"]), atol=1e-7)


def test_sample_spectra_matches_lifted_reference(golden):
    """FFT_Components.make_spectra / sample_spectra (P16:284-289, :378-388)"""
    g = golden("spectra")
    x, _ = O.synthetic_pairs(2, seed=81)
    x = torch.tanh(x * 1.2) * 0.999 + 1e-3
    spec = O.sample_spectra(x)
    assert spec.shape == (2, 1, 256, 256)
    assert torch.allclose(spec[:, :, ::4, ::4], t(g["spec_sub"]), atol=1e-5)
    assert torch.allclose(spec[1, 0, 7, :], t(g["spec_row7"]), atol=1e-5)
    assert abs(float(spec.mean()) - float(g["spec_mean"])) <= 1e-5


def test_mse_spec_matches_lifted_reference(golden):
    """evaluation metric Devcom_MagMSE.py:91-118 (lifted with scipy.fft / sklearn as the script imports them)"""
    g = golden("mse_spec")
    rng = np.random.default_rng(91)
    base = rng.integers(1, 256, size=(3, 256, 256)).astype(np.uint8)
    other = np.clip(base.astype(np.int32) + rng.integers(-40, 41, size=base.shape), 1, 255).astype(np.uint8)
    got = [O.mse_spec(base[i], other[i]) for i in range(3)]
    np.testing.assert_allclose(got, g["values"], rtol=1e-5)
    got_mae = [O.other_spec(base[i], other[i]) for i in range(3)]          # eval/Eurecom/Eurecom_MagOther.py:90-118 (other_spec)
    np.testing.assert_allclose(got_mae, g["mae_values"], rtol=1e-5)
