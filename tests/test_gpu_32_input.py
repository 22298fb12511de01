"""GPU tests of the input pipeline (datasets_temp.py:38-123) against PIL itself: Image.crop / Image.resize(BICUBIC) are the very functions the
reference calls and PIL is installed here and on the GPU box, so this row is PINNED -- and integer arithmetic, so the bar is bit-exact."""
import os

import numpy as np
import pytest
import torch
from PIL import Image

import tfc_gan_amd as T

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def ref_item(arr):
    """ImageDataset.__getitem__ of the reference restated on PIL + torch (datasets_temp.py:52-123; transforms = ToTensor, Normalize(0.5, 0.5))"""
    img = Image.fromarray(arr, "RGB")
    w, h = img.size
    A = img.crop((0, 0, w / 2, h)).resize((256, 256), Image.Resampling.BICUBIC)
    B = img.crop((w / 2, 0, w, h)).resize((256, 256), Image.Resampling.BICUBIC)
    lut = np.linspace(24, 38, num=256)
    TB = torch.Tensor(lut[np.array(B)[:, :, 0]])

    def tf(im):
        t = torch.from_numpy(np.array(im)).permute(2, 0, 1).contiguous().float().div(255)
        return (t - 0.5) / 0.5
    crops = {"B1": (0, 0, 128, 128), "B2": (128, 0, 256, 128), "B3": (0, 128, 128, 256), "B4": (128, 128, 256, 256)}
    out = {"A": tf(A), "B": tf(B), "T_B": TB, "A8": torch.from_numpy(np.array(A)), "B8": torch.from_numpy(np.array(B))}
    out.update({k: tf(B.crop(box)) for k, box in crops.items()})
    return out


def synth(h, w, seed):
    rng = np.random.default_rng(seed)
    base = rng.integers(0, 256, (h // 8 + 2, w // 8 + 2, 3)).astype(np.uint8)
    img = np.array(Image.fromarray(base, "RGB").resize((w, h), Image.Resampling.BILINEAR))
    img[::7, ::5] = rng.integers(0, 256, img[::7, ::5].shape)        # sharp pixels: exercises the clipping of the negative bicubic lobes
    img[:, w // 2:, 1] = img[:, w // 2:, 0]                           # thermal half: R = G = B
    img[:, w // 2:, 2] = img[:, w // 2:, 0]
    return img


@pytest.mark.parametrize("h,w,n", [(240, 640, 3), (256, 512, 2), (300, 513, 2), (301, 515, 1), (100, 200, 2), (1000, 2050, 1), (480, 1280, 4)])
def test_pair_resize_bit_exact_vs_pil(h, w, n):
    """even / odd widths (crop rounds half to even), no-op size, upscaling (support not widened), 4x reduction, the datasets' 1280 x 480"""
    raw = np.stack([synth(h, w, 10 * h + i) for i in range(n)])
    got = T.pair_resize_normalize(torch.from_numpy(raw).to(DEV), want_uint8=True)
    for i in range(n):
        want = ref_item(raw[i])
        for k in ("A8", "B8", "A", "B", "T_B"):
            assert torch.equal(got[k][i].cpu(), want[k]), (k, i, (got[k][i].cpu().float() - want[k].float()).abs().max())
        q = T.data.quadrants(got["B"][i])
        for k in ("B1", "B2", "B3", "B4"):
            assert torch.equal(q[k].cpu(), want[k])


def test_device_loader_matches_reference_items(tmp_path):
    """files on disk -> DeviceLoader batches with the reference's keys; order, shuffling per epoch, last short batch, mixed geometries"""
    root = tmp_path / "data"
    os.makedirs(root / "train")
    os.makedirs(root / "test")
    arrs = []
    for i in range(7):
        h, w = (240, 640) if i != 4 else (200, 500)
        a = synth(h, w, 100 + i)
        Image.fromarray(a, "RGB").save(root / "train" / f"img_{i:03d}.png")
        arrs.append(a)
    for i in range(3):
        Image.fromarray(synth(240, 640, 200 + i), "RGB").save(root / "test" / f"t_{i}.png")
    ds = T.ImageDataset(str(root), mode="train")
    assert len(ds) == 7 and [os.path.basename(f) for f in ds.files] == [f"img_{i:03d}.png" for i in range(7)]
    batches = list(T.DeviceLoader(ds, batch_size=3, shuffle=False, device=DEV))
    assert [b["A"].shape[0] for b in batches] == [3, 3, 1] and len(T.DeviceLoader(ds, batch_size=3, drop_last=True)) == 2
    assert set(batches[0].keys()) == {"A", "B", "B1", "B2", "B3", "B4", "T_B"}
    flat = {k: torch.cat([b[k] for b in batches]).cpu() for k in batches[0]}
    for i in range(7):
        want = ref_item(arrs[i])
        for k in flat:
            assert torch.equal(flat[k][i], want[k]), (k, i)
    e1 = torch.cat([b["A"] for b in T.DeviceLoader(ds, batch_size=4, shuffle=True, device=DEV, seed=1)]).cpu()
    # a shuffled epoch is a permutation of the same seven items: a different order, and every item bit-identical to exactly one unshuffled item
    assert e1.shape[0] == 7 and not torch.equal(e1, flat["A"])
    match = [[j for j in range(7) if torch.equal(e1[i], flat["A"][j])] for i in range(7)]
    assert all(len(m) == 1 for m in match) and sorted(m[0] for m in match) == list(range(7)), match
    assert len(T.ImageDataset(str(root), mode="test")) == 6          # the reference lists the test files twice in "test" mode (:46-47)
    tb = list(T.DeviceLoader(T.TestImageDataset(str(root)), batch_size=2, device=DEV))
    assert [b["A"].shape[0] for b in tb] == [2, 1] and set(tb[0].keys()) == {"A", "B"}
