"""Host-side logic that needs no GPU: the C-ABI library loads and exports every declared symbol, and the table-driven
gather model (descriptors + packed operand stream, evaluated by the library's host emulator) reproduces torch's
convolutions for every op / pass of the path -- conv, pad+conv, transposed conv, upsample+pad+conv; fwd, dgrad, wgrad."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import tfc_gan_amd as T
from tfc_gan_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    header = open(os.path.join(ROOT, "include", "tfc_gan.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = set(re.findall(r"\b(tfc_[a-z0-9_]+)\s*\(", header))
    assert len(declared) >= 28
    assert declared == set(_lib.PROTOTYPES), declared ^ set(_lib.PROTOTYPES)
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.tfc_abi_version() == 2


def test_build_refuses_spilling_hand_scheduled_kernels():
    """the asm-volatile loads of the gather GEMM / weight-gradient K loops rely on hipcc never spilling between a load and its hand-counted
    wait; build() parses the compiler's resource remarks and fails on scratch or VGPR spills of those kernels (and only of those)."""
    rem = ("a.hip:1:1: remark: Function Name: _Z17tfc_igemm2_kernelILi2ELi2ELi2ELi2ELi1ELi1EEv [-Rpass-analysis=kernel-resource-usage]\n"
           "a.hip:1:1: remark:     VGPRs: 195 [-Rpass-analysis=kernel-resource-usage]\n"
           "a.hip:1:1: remark:     ScratchSize [bytes/lane]: 0 [-Rpass-analysis=kernel-resource-usage]\n"
           "a.hip:1:1: remark:     SGPRs Spill: 27 [-Rpass-analysis=kernel-resource-usage]\n"
           "a.hip:1:1: remark:     VGPRs Spill: 0 [-Rpass-analysis=kernel-resource-usage]\n"
           "a.hip:9:1: remark: Function Name: _Z15tfc_adam_kernelPf [-Rpass-analysis=kernel-resource-usage]\n"
           "a.hip:9:1: remark:     ScratchSize [bytes/lane]: 64 [-Rpass-analysis=kernel-resource-usage]\n")
    assert _lib.check_no_spills(rem) == []                          # SGPR spills go to VGPRs (no memory); other kernels are not guarded
    bad = rem.replace("ScratchSize [bytes/lane]: 0", "ScratchSize [bytes/lane]: 16").replace("VGPRs Spill: 0", "VGPRs Spill: 3")
    got = _lib.check_no_spills(bad)
    assert len(got) == 2 and all("tfc_igemm2_kernel" in g for g in got), got


def test_errors_are_loud():
    lib = _lib.load()
    rc = lib.tfc_conv_fwd(None, 7, 0, None, 0, 1, 8, 8, 8, 8, None, None, 0, None, None, None, None, 0, None)
    assert rc != 0 and b"dtype" in lib.tfc_last_error()
    with pytest.raises(T.TfcError):
        _lib.check(rc, "tfc_conv_fwd")
    with pytest.raises(T.TfcError):                      # CPU tensors are refused, never silently computed elsewhere
        T.ops.pack_nhwc8(T.ops.DT_BF16, torch.zeros(1, 3, 8, 8))


def _ref(op, x, w):
    if op == _lib.OP_CONV:
        return F.conv2d(x, w, padding=1)
    if op == _lib.OP_PADCONV:
        return F.conv2d(F.pad(x, (1, 0, 1, 0)), w, padding=1)
    if op == _lib.OP_CONVT:
        return F.conv_transpose2d(x, w, stride=2, padding=1)
    if op == _lib.OP_CONV3:                              # 3x3 / stride 1 / pad 1 filter in rows/cols 0..2 of the 4x4 slot; the rest is ignored
        return F.conv2d(x, w[:, :, :3, :3], padding=1)
    return F.conv2d(F.pad(F.interpolate(x, scale_factor=2), (1, 0, 1, 0)), w, padding=1)


def _nhwc8(x):
    n, c, h, w = x.shape
    out = np.zeros((n, h, w, (c + 7) // 8 * 8), dtype=np.float32)
    out[..., :c] = x.detach().numpy().transpose(0, 2, 3, 1)
    return np.ascontiguousarray(out)


def _emulate(op, pas, es, a, b, out_shape, N, H, W, Cin, Cout):
    lib = _lib.load()
    y = np.zeros(out_shape, dtype=np.float32)
    rc = lib.tfc_host_emulate_conv(op, pas, es, a.ctypes.data_as(ctypes.c_void_p), b.ctypes.data_as(ctypes.c_void_p),
                                   y.ctypes.data_as(ctypes.c_void_p), N, H, W, Cin, Cout)
    _lib.check(rc, "tfc_host_emulate_conv")
    return y


CASES = [(_lib.OP_CONV, 2, 9, 19, 3, 16), (_lib.OP_CONV, 1, 20, 9, 32, 5), (_lib.OP_CONV, 1, 8, 8, 16, 40),
         (_lib.OP_PADCONV, 2, 9, 17, 32, 1), (_lib.OP_CONVT, 2, 5, 9, 16, 8), (_lib.OP_CONVT, 1, 9, 17, 32, 16),
         (_lib.OP_UPCONV, 1, 9, 10, 32, 3), (_lib.OP_UPCONV, 2, 8, 17, 16, 3),
         (_lib.OP_CONV3, 2, 9, 19, 32, 64), (_lib.OP_CONV3, 1, 8, 8, 64, 32), (_lib.OP_CONV3, 1, 6, 10, 16, 32)]


@pytest.mark.parametrize("op,N,H,W,Cin,Cout", CASES)
@pytest.mark.parametrize("es", [2, 4])
def test_gather_model_matches_torch(op, N, H, W, Cin, Cout, es):
    if (((Cin + 7) // 8 * 8) * es > 64 and (((Cin + 7) // 8 * 8) * es) % 64) or (((Cout + 7) // 8 * 8) * es > 64 and (((Cout + 7) // 8 * 8) * es) % 64):
        pytest.skip("channel count not representable in this chunk geometry")
    if op == _lib.OP_CONV3 and ((Cin * es) % 64 or (Cout * es) % 64):
        pytest.skip("TFC_OP_CONV3 takes whole 64-byte channel chunks only")
    rng = np.random.default_rng(op * 100 + Cin)
    x = torch.from_numpy(rng.standard_normal((N, Cin, H, W)).astype(np.float32)).requires_grad_(True)
    wshape = (Cin, Cout, 4, 4) if op == _lib.OP_CONVT else (Cout, Cin, 4, 4)
    w = torch.from_numpy((rng.standard_normal(wshape) * 0.1).astype(np.float32)).requires_grad_(True)
    y = _ref(op, x, w)
    go = torch.from_numpy(rng.standard_normal(tuple(y.shape)).astype(np.float32))
    gx, gw = torch.autograd.grad(y, (x, w), go)
    OH, OW = y.shape[2:]
    c8 = lambda c: (c + 7) // 8 * 8  # noqa: E731
    wn = np.ascontiguousarray(w.detach().numpy())
    # forward
    got = _emulate(op, 0, es, _nhwc8(x), wn, (N, OH, OW, c8(Cout)), N, H, W, Cin, Cout)
    np.testing.assert_allclose(got[..., :Cout], y.detach().numpy().transpose(0, 2, 3, 1), atol=2e-4)
    # dgrad
    got = _emulate(op, 1, es, _nhwc8(go), wn, (N, H, W, c8(Cin)), N, H, W, Cin, Cout)
    np.testing.assert_allclose(got[..., :Cin], gx.numpy().transpose(0, 2, 3, 1), atol=2e-4)
    # wgrad
    got = _emulate(op, 2, es, _nhwc8(x), _nhwc8(go), wshape, N, H, W, Cin, Cout)
    np.testing.assert_allclose(got, gw.numpy(), atol=2e-3, rtol=1e-4)


def test_flat_params_and_buckets():
    from tfc_gan_amd import nets, parallel
    from oracle import tfcgan_oracle as O
    G = O.GeneratorUNet((3, 256, 256))
    named = {k: v for k, v in G.named_parameters()}
    flat = parallel.FlatParams(named, nets.g_backward_order(), torch.device("cpu"))
    assert flat.numel >= 29238275 and set(flat.order) == set(nets.g_param_names())
    for k, p in named.items():
        assert torch.equal(flat.views[k], p.detach()) and flat.views[k].data_ptr() % 16 == 0
    red = parallel.BucketReducer(flat, bucket_bytes=32 << 20)
    assert 2 <= len(red.buckets) <= 5
    assert red.buckets[0][0] == 0 and red.buckets[-1][1] == flat.numel
    assert all(a[1] == b[0] for a, b in zip(red.buckets, red.buckets[1:]))
    assert red.finish() == 1.0
    a, b = parallel.shared_neg_idx(3), parallel.shared_neg_idx(3)
    assert a == b and len(a) == 16 and all(0 <= i < 16 for i in a) and a != parallel.shared_neg_idx(4)


def test_patch_views_and_index_map(golden):
    g = golden("patch_index_map")
    B = torch.arange(3 * 256 * 256, dtype=torch.float32).reshape(1, 3, 256, 256)
    ps = T.make_16_patches(B)
    assert [int(p[0, 0, 0, 0]) for p in ps] == g["first_flat"].tolist()
    assert [T.patch_first_flat_index(k) for k in range(16)] == g["first_flat"].tolist()
    assert all(p.data_ptr() == B.data_ptr() + 4 * f for p, f in zip(ps, g["first_flat"].tolist()))    # views, no copies


def test_module_state_dict_contract(golden):
    g = golden("state_dict_keys")
    G, D = T.GeneratorUNet((3, 256, 256)), T.Discriminator1((3, 256, 256))
    assert list(G.state_dict().keys()) == g["g_keys"].tolist()
    assert list(D.state_dict().keys()) == g["d_keys"].tolist()
    assert [str(tuple(v.shape)) for v in G.state_dict().values()] == g["g_shapes"].tolist()
    assert [str(tuple(v.shape)) for v in D.state_dict().values()] == g["d_shapes"].tolist()
    G.apply(T.weights_init_normal)
    D.apply(T.weights_init_normal)
    assert abs(float(G.down2.model[0].weight.std()) - 0.02) < 2e-3
    assert T.Discriminator is T.Discriminator1
    with pytest.raises(T.TfcError):                      # CPU modules fail loudly: there is no CPU fallback
        G(torch.zeros(1, 3, 256, 256))


def test_dataparallel_prefixed_checkpoint_roundtrip():
    """reference checkpoints carry nn.DataParallel's 'module.' prefix (P16:692-695); test_TFCGAN_16Patches.py strips it (T16:153-163)"""
    G = T.GeneratorUNet((3, 256, 256))
    sd = {"module." + k: v.clone() + 1 for k, v in G.state_dict().items()}
    clean = {k[7:]: v for k, v in sd.items()}                       # load_clean_state
    missing = G.load_state_dict(clean)
    assert not missing.missing_keys and not missing.unexpected_keys
    assert torch.equal(G.state_dict()["final.2.bias"], sd["module.final.2.bias"])


def test_checkpoint_helpers_and_stitch(tmp_path):
    """save_checkpoint writes the 'module.'-prefixed file of P16:692-695; load_clean_state (T16:153-163) reads it back; the
    stitch grid of T16:217-263 puts fake patch k over real patch k in channel group k."""
    G = T.GeneratorUNet((3, 256, 256))
    with torch.no_grad():
        for p_ in G.parameters():
            p_.add_(0.5)
    path = str(tmp_path / "generator_0.pth")
    T.save_checkpoint(G, path)
    raw = torch.load(path, weights_only=True)
    assert all(k.startswith("module.") for k in raw) and len(raw) == len(G.state_dict())
    G2 = T.load_clean_state(T.GeneratorUNet((3, 256, 256)), path)
    for (k, a), (_, b) in zip(G.state_dict().items(), G2.state_dict().items()):
        assert torch.equal(a, b), k
    f = torch.arange(2 * 3 * 256 * 256, dtype=torch.float32).reshape(2, 3, 256, 256)
    r = -f
    grid = T.stitch_16_patches(f, r)
    assert grid.shape == (2, 48, 128, 64)
    for k in (0, 5, 15):
        y0, x0 = 64 * (k // 4), 64 * (k % 4)
        assert torch.equal(grid[:, 3 * k:3 * k + 3, :64], f[:, :, y0:y0 + 64, x0:x0 + 64])
        assert torch.equal(grid[:, 3 * k:3 * k + 3, 64:], r[:, :, y0:y0 + 64, x0:x0 + 64])
    assert T.global_grid(f, r, f).shape == (2, 3, 768, 256)


def test_zero_arena_hands_out_zeroed_disjoint_views():
    """the per-step scratch arena: every view is zero when taken, views never overlap, begin() re-zeroes what was handed out"""
    from tfc_gan_amd import ops
    a = ops.ZeroArena(torch.device("cpu"), nfloats=1024)
    a.begin()
    v1, v2 = a.take((3, 5)), a.take((64,))
    assert v1.abs().sum() == 0 and v2.abs().sum() == 0
    v1.fill_(1.0); v2.fill_(2.0)
    assert v1.sum() == 15 and v2.sum() == 128                   # disjoint
    assert v1.data_ptr() % 16 == 0 and v2.data_ptr() % 16 == 0
    big = a.take((4096,))                                        # does not fit: falls back to a fresh zero tensor
    assert big.numel() == 4096 and big.abs().sum() == 0
    a.begin()
    w = a.take((3, 5))
    assert w.data_ptr() == v1.data_ptr() and w.abs().sum() == 0


def test_synthetic_input_recipe_matches_the_oracle():
    """bench.py feeds the product's own synthetic_pairs (the timed path never imports oracle/); the recipe is the oracle's, bit for bit"""
    from oracle import tfcgan_oracle as O
    a, b = T.synthetic_pairs(2, seed=77)
    oa, ob = O.synthetic_pairs(2, seed=77)
    assert torch.equal(a, oa) and torch.equal(b, ob)
    assert a.min() >= -1 and a.max() <= 1 and torch.equal(b[:, 0], b[:, 1]) and torch.equal(b[:, 0], b[:, 2])
    t = T.synthetic_temperatures(2, seed=77)
    assert t.shape == (2, 256, 256) and t.min() >= 24 and t.max() <= 38


def test_dataparallel_replica_is_refused_loudly():
    """nn.DataParallel on several devices replicates the module (`_is_replica`): its copies hold no Parameters and would share the
    operand-stream cache between threads, so forward raises and points at the one-process-per-GPU path (P16:444-445 replaced)"""
    for mod, args in ((T.GeneratorUNet((3, 256, 256)), (torch.zeros(1, 3, 256, 256),)),
                      (T.Discriminator1((3, 256, 256)), (torch.zeros(1, 3, 256, 256), torch.zeros(1, 3, 256, 256)))):
        mod._is_replica = True
        with pytest.raises(T.TfcError, match="one process per GPU|one\\s+process per GPU"):
            mod(*args)


def test_dropout_seed_streams_are_per_module():
    """no process-global RNG counter: two modules draw independent, reproducible seed streams"""
    from tfc_gan_amd import models
    a, b = T.UNetDown(8, 8), T.UNetDown(8, 8)
    a._drop_seed = b._drop_seed = 123
    sa = [models._next_seed(a) for _ in range(3)]
    sb = [models._next_seed(b) for _ in range(3)]
    assert sa == sb and len(set(sa)) == 3
    assert models._next_seed(a) != sb[0]


def test_profiling_state_is_per_thread():
    """tfc_prof_enable / tfc_debug_set_igemm_config act on the calling thread only (no process-global mutable state in the library)"""
    import threading
    lib = _lib.load()
    assert lib.tfc_prof_enable(1) == 0
    seen = {}

    def other():
        n = lib.tfc_prof_records(4, None, None, None, None)
        seen["n"] = n
        seen["rc"] = lib.tfc_debug_set_igemm_config(2)
    t = threading.Thread(target=other)
    t.start()
    t.join()
    assert seen == {"n": 0, "rc": 0}
    assert lib.tfc_prof_enable(0) == 0


def test_lpips_local_weight_files(tmp_path):
    """LPIPS weights come from caller-supplied local files only (weights-only loader): torchvision's `features.N.*` layout plus the original
    `lin{i}.model.1.weight` heads, or this class's own state_dict; anything partial is refused"""
    import warnings
    from tfc_gan_amd import lpips as L
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        src = T.LPIPS(seed=7)
    assert any("seeded-random" in str(x.message) for x in w) and not src.pretrained
    assert L.CONV_INDEX == (0, 2, 5, 7, 10, 12, 14, 17, 19, 21, 24, 26, 28)
    vgg = {f"features.{k}.{n}": getattr(src.net.layers[k], n).detach().clone() for k in L.CONV_INDEX for n in ("weight", "bias")}
    vgg["classifier.0.weight"] = torch.zeros(2, 2)                                # ignored, as torchvision's checkpoint carries it
    lin = {f"lin{i}.model.1.weight": seq[1].weight.detach().clone() for i, seq in enumerate(src.lin)}
    torch.save(vgg, tmp_path / "vgg16.pth")
    torch.save(lin, tmp_path / "lin.pth")
    torch.save(src.state_dict(), tmp_path / "own.pth")
    a = T.LPIPS(weights=[tmp_path / "vgg16.pth", tmp_path / "lin.pth"], seed=1)
    b = T.LPIPS(weights=str(tmp_path / "own.pth"), seed=2)
    assert a.pretrained and b.pretrained
    for k, v in src.state_dict().items():
        assert torch.equal(a.state_dict()[k], v) and torch.equal(b.state_dict()[k], v)
    assert all(not p.requires_grad for p in a.parameters())
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        half = T.LPIPS(seed=1).load_pretrained(tmp_path / "lin.pth")              # heads only: accepted, but not "pretrained"
        assert not half.pretrained
        torch.save({k: v for k, v in list(vgg.items())[:6]}, tmp_path / "partial.pth")
        with pytest.raises(T.TfcError):
            T.LPIPS(weights=str(tmp_path / "partial.pth"))
        with pytest.raises(T.TfcError):
            T.LPIPS(net_type="alex")
        with pytest.raises(T.TfcError):
            T.LPIPS()(torch.zeros(1, 3, 32, 32), torch.zeros(1, 3, 32, 32))        # CPU tensors: no fallback


@pytest.mark.parametrize("h,w", [(60, 200), (75, 131), (300, 1000)])
def test_resize_plan_tables_reproduce_pil(h, w):
    """tfc_resize_plan_build (host): its tap tables, evaluated with numpy integers exactly as the kernels do (22-bit taps, int32 accumulation
    from 1 << 21, uint8 after each pass), reproduce PIL's Image.crop + Image.resize(BICUBIC) of both halves bit for bit"""
    from PIL import Image
    lib = _lib.load()
    out = 64
    n = lib.tfc_resize_plan_bytes(h, w, out)
    buf = (ctypes.c_uint8 * n)()
    _lib.check(lib.tfc_resize_plan_build(h, w, out, ctypes.cast(buf, ctypes.c_void_p)), "plan")
    plan = np.frombuffer(bytes(buf), dtype=np.int32)
    axes = [dict(ksize=plan[4 * i], b=plan[4 * i + 1], c=plan[4 * i + 2], n=plan[4 * i + 3]) for i in range(3)]
    assert plan[12] == out and plan[14] == h and plan[15] == w
    xs = int(plan[13])
    assert xs == int(round(w / 2)) and axes[0]["n"] == xs and axes[1]["n"] == w - xs and axes[2]["n"] == h
    img = np.random.default_rng(h).integers(0, 256, (h, w, 3)).astype(np.uint8)

    def run(src, ax, axis):
        src = np.moveaxis(src.astype(np.int64), axis, 0)
        res = np.empty((out,) + src.shape[1:], dtype=np.uint8)
        for o in range(out):
            x0, cnt = plan[ax["b"] + 2 * o], plan[ax["b"] + 2 * o + 1]
            k = plan[ax["c"] + o * ax["ksize"]: ax["c"] + o * ax["ksize"] + cnt].astype(np.int64)
            acc = (1 << 21) + np.tensordot(k, src[x0:x0 + cnt], axes=(0, 0))
            res[o] = np.clip(acc >> 22, 0, 255)
        return np.moveaxis(res, 0, axis)
    pil = Image.fromarray(img, "RGB")
    for half, (x0, x1) in enumerate(((0, xs), (xs, w))):
        got = run(run(img[:, x0:x1], axes[half], 1), axes[2], 0)
        box = (0, 0, w / 2, h) if half == 0 else (w / 2, 0, w, h)
        want = np.array(pil.crop(box).resize((out, out), Image.Resampling.BICUBIC))
        assert np.array_equal(got, want)


def _one_rank_worker(port, out_path):
    """one-rank gloo group on CPU: the collectives are skipped by default and issued under TFC_FORCE_COLLECTIVES=1 (the one-GPU rehearsal switch)"""
    import json
    import os
    import torch
    import torch.distributed as dist
    from tfc_gan_amd import parallel
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1)
    named = {"a": torch.ones(1000), "b": torch.full((3000,), 2.0)}
    flat = parallel.FlatParams(named, ["b", "a"], torch.device("cpu"))
    flat.grad.fill_(3.0)
    calls = {"n": 0}
    real = dist.all_reduce

    def counting(*a, **k):
        calls["n"] += 1
        return real(*a, **k)
    parallel.dist.all_reduce = counting
    res = {}
    for mode in ("0", "1"):
        os.environ["TFC_FORCE_COLLECTIVES"] = mode
        calls["n"] = 0
        red = parallel.BucketReducer(flat, bucket_bytes=4096)
        for k in flat.order:
            red.ready(k)
        scale = red.finish()
        res[mode] = {"active": parallel.collectives_active(), "calls": calls["n"], "scale": scale, "grad": float(flat.grad.sum())}
    dist.destroy_process_group()
    with open(out_path, "w") as f:
        json.dump(res, f)


def test_collectives_switch_in_a_one_rank_group(tmp_path):
    """parallel.collectives_active(): off in a one-rank group (nothing to exchange), on under TFC_FORCE_COLLECTIVES=1 -- then every bucket is all-reduced
    (an identity) and finish() still returns 1 / world = 1"""
    import json
    import socket
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    out = str(tmp_path / "r.json")
    p = mp.get_context("spawn").Process(target=_one_rank_worker, args=(port, out))
    p.start()
    p.join(120)
    assert p.exitcode == 0
    res = json.load(open(out))
    assert res["0"] == {"active": False, "calls": 0, "scale": 1.0, "grad": 3.0 * 4000}
    assert res["1"]["active"] is True and res["1"]["calls"] == 2 and res["1"]["scale"] == 1.0 and res["1"]["grad"] == 3.0 * 4000
