"""GPU tests of the STN21 configuration as a runnable step (TFC-STN/TFCGAN_STN21_Original_NewModel3_Official.py, STN:170-231, :609-672).
The generator's INPUT gradient (new: STN21 trains the warp and the localiser through generator2(warped_B)) is pinned against the oracle;
the localiser's ViT is a restatement of kornia's (absent): parity unpinned, the warp path around it is checked against torch's own
affine_grid / grid_sample."""
import warnings

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import tfc_gan_amd as T
from oracle import tfcgan_oracle as O
from tfc_gan_amd import stn21

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def rnd(shape, seed, scale=1.0):
    return torch.from_numpy((np.random.default_rng(seed).standard_normal(shape) * scale).astype(np.float32))


def test_generator_input_gradient_vs_oracle():
    """d loss / d x of GeneratorUNet: fp32 parity mode against torch autograd on the oracle (eval mode: no dropout), bf16 mode against that
    result by direction (the network is 12 bf16 layers deep)"""
    Gc = O.init_weights_portable(O.GeneratorUNet((3, 128, 128)), seed=21).eval()
    x = (rnd((1, 3, 128, 128), 1) * 0.5).clamp(-1, 1)
    gw = rnd((1, 3, 128, 128), 2)
    xc = x.clone().requires_grad_(True)
    (Gc(xc) * gw).sum().backward()
    want = xc.grad
    try:
        for dtype, check in ((torch.float32, "tight"), (torch.bfloat16, "direction")):
            T.set_compute_dtype(dtype)
            G = T.GeneratorUNet((3, 128, 128))
            G.load_state_dict(Gc.state_dict())
            G = G.to(DEV).eval()
            xg = x.to(DEV).requires_grad_(True)
            (G(xg) * gw.to(DEV)).sum().backward()
            got = xg.grad.cpu()
            assert got.shape == want.shape
            if check == "tight":
                # same bar as the fp32 train-step gradients (ReLU / LeakyReLU knife-edge flips on an N = 1 input: 3e-3 observed)
                assert (got - want).norm().item() <= 2e-2 * want.norm().item(), (got - want).norm().item() / want.norm().item()
            else:
                cos = F.cosine_similarity(got.reshape(1, -1), want.reshape(1, -1)).item()     # 0.96-0.99 run to run (atomics order -> 1-ulp ties -> chaos)
                assert cos > 0.9, cos
            assert all(p.grad is not None for p in G.parameters())
    finally:
        T.set_compute_dtype(torch.bfloat16)


def test_net_warp_path_vs_torch_reference():
    """Net.forward (STN:204-231) on the GPU against the same modules on the CPU with torch's affine_grid / grid_sample: warped image and the
    gradient that reaches the last localiser layer"""
    torch.manual_seed(3)
    net = stn21.Net((3, 128, 128))
    with torch.no_grad():
        net.fc_loc[6].weight.mul_(0.05)
        net.fc_loc[6].bias.copy_(torch.tensor([0.02, -0.03, 0.05, 0.04, 0.01, -0.02]))
    A, B, src = rnd((2, 3, 128, 128), 4, 0.4), rnd((2, 3, 128, 128), 5, 0.4), rnd((2, 3, 128, 128), 6, 0.4)
    go = rnd((2, 3, 128, 128), 7)
    dth = net.stn_phi(torch.cat((A, B), 1))
    theta = dth.reshape(2, 6) + torch.tensor([1.0, 0, 0, 0, 1.0, 0])
    outs = []
    for i in range(2):
        grid = F.affine_grid(theta[i].view(1, 2, 3), src[i:i + 1].size(), align_corners=True)
        outs.append(F.grid_sample(src[i:i + 1], grid, mode="bicubic", padding_mode="border", align_corners=True))
    want = torch.cat(outs)
    (want * go).sum().backward()
    gref = net.fc_loc[6].weight.grad.clone()
    net.zero_grad()
    ng = net.to(DEV)
    got = ng(A.to(DEV), B.to(DEV), src.to(DEV))
    assert (got.cpu() - want.detach()).abs().max().item() <= 2e-3
    (got * go.to(DEV)).sum().backward()
    g2 = ng.fc_loc[6].weight.grad.cpu()
    assert (g2 - gref).norm().item() <= 2e-2 * gref.norm().item(), (g2 - gref).norm().item() / gref.norm().item()
    assert ng.localization.vit[0].patch.weight.grad is not None and ng.localization.vit[0].patch.weight.grad.abs().sum().item() > 0


def test_stn21_step_runs_and_trains_everything():
    """two steps of the STN21 batch-loop body at batch 2, 256 x 256: every loss finite, every network (both generators, the localiser and its
    MLP, both discriminators) moves, the second generator receives gradients from BOTH of its uses"""
    torch.manual_seed(5)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        crit = T.LPIPS().to(DEV)
    st = stn21.STN21Step((3, 256, 256), lpips=crit, device=DEV)
    probes = {"G1": st.G1.down4.model[0].weight, "G2": st.G2.up2.model[0].weight, "G2.down1": st.G2.down1.model[0].weight,
              "net.fc": st.net.fc_loc[6].weight, "net.vit": st.net.localization.vit[0].patch.weight,
              "D1": st.D1.model[3].parametrizations.weight.original, "D2": st.D2.model[9].parametrizations.weight.original}
    before = {k: v.detach().clone() for k, v in probes.items()}
    A, B = T.synthetic_pairs(2, seed=9)
    for _ in range(2):
        out = st.step(A.to(DEV), B.to(DEV))
    for k in ("loss_G", "loss_GAN", "recon_loss", "perc_loss", "morph_loss", "loss_D"):
        assert torch.isfinite(out[k]).all(), k
    assert out["loss_D"].item() > 0 and out["perc_loss"].item() > 0 and out["recon_loss"].item() > 0
    assert out["fake_B"].shape == (2, 3, 256, 256) and out["warped_B"].shape == (2, 3, 256, 256)
    for k, v in probes.items():
        assert not torch.equal(before[k], v.detach()), f"{k} did not move"
    # bicubic interpolation of values in [-1, 1] overshoots by at most the kernel's negative lobes
    assert torch.isfinite(out["warped_B"]).all() and out["warped_B"].abs().max().item() < 1.6
