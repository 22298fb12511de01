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
    # bicubic (A = -0.75) interpolation of values in [-1, 1]: per axis the taps sum to 1 and their negative lobes to at most 2 * 0.09375 at t = 0.5 ... the
    # absolute tap sum is <= 1.25 per axis, so |warped| <= 1.25^2 = 1.5625 for ANY theta
    assert torch.isfinite(out["warped_B"]).all() and out["warped_B"].abs().max().item() <= 1.5625 + 1e-5


def _portable_stn21(dev, dtype):
    """STN21Step with the weights of tests/golden/train_step_stn21.npz: init_weights_portable seeds 101..105 in the fixture's module order (the
    modules' state_dict order equals the lifted classes': asserted on the localiser's 160 keys), last localiser layer scaled for a visible warp"""
    T.set_compute_dtype(dtype)
    st = stn21.STN21Step((3, 256, 256), lpips=None, device=dev)
    for i, m in enumerate((st.G1, st.G2, st.D1, st.D2, st.net)):
        O.init_weights_portable(m, seed=101 + i)                  # writes through the parameter views into the flat buffers
    with torch.no_grad():
        st.net.fc_loc[6].weight.mul_(4.0)
        st.net.fc_loc[6].bias.copy_(torch.tensor([0.03, -0.02, 0.04, 0.02, -0.03, -0.05], device=dev))
    st._bump()
    st.G1.eval(); st.G2.eval(); st.net.eval(); st.D1.train(); st.D2.train()
    return st


def test_stn21_step_vs_reference_golden(golden):
    """one STN21 step against the step composed from the reference's OWN definitions (tests/golden/make_golden.py section xiii lifts
    Net / LocalizerVIT, both generators and discriminators, morph_triplet, global_pixel_loss, global_gen_loss, global_disc_loss from STN:150-505 and
    composes them as STN:609-672; morph.gradient and K.VisionTransformer are stand-ins, LPIPS is off on both sides): the seven losses, the warped and
    generated images, gradients of both generators and of the localiser's MLP, and the Adam deltas -- fp32 parity mode, N = 1."""
    g = golden("train_step_stn21")
    try:
        st = _portable_stn21(DEV, torch.float32)
        assert [k for k, _ in st.net.state_dict().items()] == [str(k) for k in g["net_keys"]]
        before = {"G2.final.2.weight": st.G2.final[2].weight.detach().clone(), "net.fc6": st.net.fc_loc[6].weight.detach().clone(),
                  "D1.head": st.D1.model[13].weight.detach().clone()}
        A, B = O.synthetic_pairs(1, seed=105)
        out = st.step(A.to(DEV), B.to(DEV))
        torch.cuda.synchronize()
    finally:
        T.set_compute_dtype(torch.bfloat16)
    for k in ("loss_G", "loss_GAN", "recon_loss", "morph_loss", "loss_D", "loss_D1", "loss_D2"):
        w = float(g[k])
        assert abs(float(out[k]) - w) <= 3e-4 * max(1.0, abs(w)), (k, float(out[k]), w)
    for k, got in (("warped_sub", out["warped_B"]), ("fake_A2_sub", out["fake_A2"]), ("fake_B_sub", out["fake_B"])):
        err = (got.cpu()[:, :, ::8, ::8] - torch.from_numpy(g[k])).abs().max().item()
        assert err <= 2e-3, (k, err)

    def rel(a, b):
        return ((a.double() - b.double()).norm() / b.double().norm()).item()
    gv = st.gflat.grad_views
    checks = [("g_G1_down1", gv["G1.down1.model.0.weight"]), ("g_G2_down1", gv["G2.down1.model.0.weight"]), ("g_G2_up3", gv["G2.up3.model.0.weight"][::16, ::16]),
              ("g_fc6", gv["net.fc_loc.6.weight"]), ("g_fc6_bias", gv["net.fc_loc.6.bias"]), ("g_fc0", gv["net.fc_loc.0.weight"][::64, ::256]),
              ("g_D1_head", st.dflat.grad_views["D1.model.13.weight"]), ("g_D2_b0", st.dflat.grad_views["D2.model.0.bias"])]
    for k, got in checks:
        r = rel(got.cpu(), torch.from_numpy(g[k]))
        print(f"  {k:12s} rel-L2 {r:.3e}")
        assert r <= 2e-2, (k, r)                                  # fp32 gradients of an N = 1 step: activation knife-edge flips (the PATCH-16 fp32 bar)
    for k, now, want in (("d_G2_final", st.G2.final[2].weight, "G2.final.2.weight"), ("d_fc6", st.net.fc_loc[6].weight, "net.fc6"),
                         ("d_D1_head", st.D1.model[13].weight, "D1.head")):
        delta = (now.detach() - before[want]).cpu()
        # the first Adam step moves every weight by ~lr * sign(g): only entries whose gradient is at round-off level may differ
        frac = ((delta - torch.from_numpy(g[k])).abs() > 2e-5).float().mean().item()
        assert frac <= 2e-2, (k, frac)


def test_stn21_batch32_bf16_properties():
    """configuration C5 at its batch (TFC-STN/0302_STN21_Devcom_NewModel.sh:3: 32) in the benchmarked arithmetic, with the LPIPS term: losses finite and in
    range, every network moves, the step is repeatable run to run (no order-dependent sums on its gradient path)"""
    def run():
        torch.manual_seed(7)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            crit = T.LPIPS().to(DEV)
        st = stn21.STN21Step((3, 256, 256), lpips=crit, device=DEV, seed=3)
        A, B = T.synthetic_pairs(32, seed=21)
        out = st.step(A.to(DEV), B.to(DEV))
        torch.cuda.synchronize()
        return st, out
    T.set_compute_dtype(torch.bfloat16)
    st, out = run()
    for k in ("loss_G", "loss_GAN", "recon_loss", "perc_loss", "morph_loss", "loss_D"):
        assert torch.isfinite(out[k]).all(), k
    assert 0.5 < out["loss_GAN"].item() < 4.0 and 0.1 < out["loss_D"].item() < 1.0 and 0.0 < out["recon_loss"].item() < 2.0
    assert out["morph_loss"].item() > 0 and out["perc_loss"].item() > 0
    assert out["warped_B"].abs().max().item() <= 1.5625 + 1e-5 and out["fake_B"].abs().max().item() <= 1.0
    gg, dg = st.gflat.grad.clone(), st.dflat.grad.clone()
    assert torch.isfinite(gg).all() and torch.isfinite(dg).all()
    for k in ("net.fc_loc.6.weight", "net.localization.vit.0.patch.weight", "G2.down1.model.0.weight", "G1.up3.model.0.weight"):
        assert st.gflat.grad_views[k].abs().sum().item() > 0, k
    st2, _ = run()
    # the custom kernels are order-independent; the localiser's library GEMMs / softmax are the only foreign code on the path
    r = ((st2.gflat.grad.double() - gg.double()).norm() / gg.double().norm()).item()
    assert r <= 1e-6, r
    assert torch.equal(st2.dflat.grad, dg)



def test_stn21_step_under_a_real_one_rank_rccl_group(tmp_path):
    """STN21Step's exchange (post-accumulate hooks -> bucket all-reduces -> tfc_adam_step) through the real ProcessGroupNCCL in a one-rank group
    (TFC_FORCE_COLLECTIVES=1): the sums are identities, so the step must end bit-equal to the run without a process group"""
    import os
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    worker = os.path.join(root, "tests", "stn21_ddp_worker.py")
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    one, two = str(tmp_path / "plain.pt"), str(tmp_path / "rccl.pt")
    env = dict(os.environ, TFC_TEST_PORT=str(port))
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "TFC_FORCE_COLLECTIVES", "TFC_TEST_RCCL1"):
        env.pop(k, None)
    subprocess.run([sys.executable, worker, one], check=True, env=env, timeout=600)
    r = subprocess.run([sys.executable, worker, two], env=dict(env, TFC_TEST_RCCL1="1"), timeout=600)
    if r.returncode == 77:
        pytest.skip("a one-rank RCCL process group cannot be created on this box (worker exit code 77)")
    assert r.returncode == 0
    a, b = torch.load(one, weights_only=True), torch.load(two, weights_only=True)
    for k in ("losses", "gg", "dg", "g", "d"):
        assert torch.equal(a[k], b[k]), (k, (a[k].double() - b[k].double()).abs().max().item())


def test_stn21_two_ranks_match_one_rank(tmp_path):
    """configuration C5 is an 8-GPU configuration (BASELINE.json configs[4]): STN21Step shards the batch over one process per GPU with the same flat
    buffers + bucketed all-reduce as the PATCH-16 engine. Two ranks (sharing the one card, gloo), each stepping one image of a global batch of 2, must
    end where one rank stepping both images ends: every loss of the step is a batch mean, so the rank-averaged gradient is the global-batch gradient up
    to fp32 summation order (reference semantics: nn.DataParallel gathers the batch, STN:536-540)."""
    import os
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    worker = os.path.join(root, "tests", "stn21_ddp_worker.py")
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    one, two = str(tmp_path / "one.pt"), str(tmp_path / "two.pt")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    subprocess.run([sys.executable, worker, one], check=True, env=env, timeout=600)
    subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                    "--master-port", str(port), worker, two], check=True, env=env, timeout=600)
    a, b = torch.load(one, weights_only=True), torch.load(two, weights_only=True)
    assert torch.allclose(a["losses"], b["losses"], rtol=2e-5, atol=1e-6), (a["losses"], b["losses"])
    for k in ("gg", "dg"):
        rel = ((a[k] - b[k]).norm() / a[k].norm()).item()
        # unlike PATCH-16 (test_gpu_20_ddp.py: <= 1e-5) the per-image work is NOT bit-identical here: the localiser runs library GEMMs whose
        # algorithm depends on the batch size (2 images vs 1), theta moves by ~1e-7, and in fp32 mode a handful of ReLU / LeakyReLU decisions of
        # near-zero pre-activations behind the warp flip (1.1e-3 observed; the same knife edge as the fp32 golden tests, bar 1e-2)
        assert rel <= 5e-3, (k, rel)
    for k in ("g", "d"):
        frac = ((a[k] - b[k]).abs() > 2e-5).float().mean().item()
        assert frac <= 2e-2, (k, frac)
