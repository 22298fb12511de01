"""Input pipeline of the training scripts (reference TFC-GAN-FFT/datasets_temp.py:38-123 `ImageDataset`, :125-158 `TestImageDataset`;
used at P16:478-500, :524-530) with the per-pixel work on the GPU.

The reference does, per file and on DataLoader worker processes: PIL open -> crop the left / right half -> `resize((256, 256), BICUBIC)` each
-> temperature map of B through a dict lookup -> four 128 x 128 crops of B -> ToTensor + Normalize on all six images; the batch then crosses
PCIe as fp32 (6.3 MB per 32 pairs + crops). Here the host only DECODES the file (PIL, a thread pool); the uint8 pixels cross PCIe once
(59 MB per 32 files of 1280 x 480 -- or 1.2 MB per file) and `tfc_pair_resize_normalize` produces A, B, T_B on the device, bit-exact against PIL's
resampler; B1..B4 are views of B. The batch dict has the reference's keys.

    ds = ImageDataset(root, mode="train")                  # same file discovery as the reference (sorted glob of root/mode/*.*)
    for batch in DeviceLoader(ds, batch_size=32, shuffle=True, device="cuda:0"):
        ts.step(batch["A"], batch["B"], T_B=batch["T_B"])

`transforms_` is accepted for call compatibility; the reference always passes [ToTensor(), Normalize((0.5,)*3, (0.5,)*3)] (P16:479-482) and
that is what the kernel applies (torchvision is not installed here; anything else is refused).
"""
import ctypes
import glob
import os
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch

from . import ops
from ._lib import TfcError, check


class ResizePlan:
    """tap tables of one file geometry (H, W) -> 2 x (out x out), host + device copies"""

    def __init__(self, H, W, out, device):
        lib = ops.lib()
        nbytes = lib.tfc_resize_plan_bytes(H, W, out)
        if nbytes == 0:
            raise TfcError(f"bad image geometry {W} x {H}")
        self.host = (ctypes.c_uint8 * nbytes)()
        check(lib.tfc_resize_plan_build(H, W, out, ctypes.cast(self.host, ctypes.c_void_p)), "tfc_resize_plan_build")
        self.dev = torch.frombuffer(bytearray(self.host), dtype=torch.uint8).to(device)
        self.H, self.W, self.out = H, W, out


_PLANS = {}
_LUT = {}


def temperature_lut(device):
    """self.T = np.linspace(24, 38, num=256) (datasets_temp.py:41), as the float32 values torch.Tensor(...) makes of it (:70)"""
    dev = torch.device(device)
    if dev not in _LUT:
        _LUT[dev] = torch.from_numpy(np.linspace(24, 38, num=256).astype(np.float32)).to(dev)
    return _LUT[dev]


def pair_resize_normalize(raw, out=256, want_temps=True, want_uint8=False):
    """raw: uint8 [N, H, W, 3] on the GPU (N decoded A|B files of one geometry). Returns dict A, B [N,3,out,out] fp32 in [-1,1], T_B [N,out,out]
    (and A8 / B8 [N,out,out,3] uint8 when asked)."""
    ops.require_gpu(raw)
    if raw.dtype != torch.uint8 or raw.dim() != 4 or raw.shape[3] != 3:
        raise TfcError(f"expected uint8 [N, H, W, 3], got {raw.dtype} {tuple(raw.shape)}")
    raw = raw.contiguous()
    N, H, W, _ = raw.shape
    key = (H, W, out, raw.device)
    plan = _PLANS.get(key)
    if plan is None:
        plan = _PLANS[key] = ResizePlan(H, W, out, raw.device)
    lib = ops.lib()
    ws = torch.empty(lib.tfc_pair_resize_ws_bytes(N, H, out), dtype=torch.uint8, device=raw.device)
    A = torch.empty((N, 3, out, out), dtype=torch.float32, device=raw.device)
    B = torch.empty_like(A)
    TB = torch.empty((N, out, out), dtype=torch.float32, device=raw.device) if want_temps else None
    A8 = torch.empty((N, out, out, 3), dtype=torch.uint8, device=raw.device) if want_uint8 else None
    B8 = torch.empty_like(A8) if want_uint8 else None
    check(lib.tfc_pair_resize_normalize(ops.stream_ptr(), ops._p(raw), H * W * 3, W * 3, N, ctypes.cast(plan.host, ctypes.c_void_p), ops._p(plan.dev),
                                        ops._p(ws), ops._p(temperature_lut(raw.device)), ops._p(A), ops._p(B), ops._p(TB), ops._p(A8), ops._p(B8)),
          "tfc_pair_resize_normalize")
    res = {"A": A, "B": B}
    if want_temps:
        res["T_B"] = TB
    if want_uint8:
        res["A8"], res["B8"] = A8, B8
    return res


def quadrants(B):
    """B1..B4 of datasets_temp.py:80-110: crop boxes (0,0,128,128), (128,0,256,128), (0,128,128,256), (128,128,256,256) -- views, no copies"""
    h, w = B.shape[-2] // 2, B.shape[-1] // 2
    return {"B1": B[..., :h, :w], "B2": B[..., :h, w:], "B3": B[..., h:, :w], "B4": B[..., h:, w:]}


class ImageDataset:
    """File discovery and decoding of the reference's ImageDataset (datasets_temp.py:38-50): items are the DECODED files (uint8 HWC), the
    per-pixel pipeline runs in DeviceLoader on the GPU."""

    test_extends = True

    def __init__(self, root, transforms_=None, mode="train"):
        if transforms_ is not None and len(transforms_) not in (0, 2):
            raise TfcError("ImageDataset applies the reference's own transform list [ToTensor(), Normalize((0.5,)*3, (0.5,)*3)] (P16:479-482) on the "
                           "GPU; other transform lists are not supported")
        self.files = sorted(glob.glob(os.path.join(root, mode) + "/*.*"))
        if mode == "test" and self.test_extends:                  # the reference appends the test files a second time (:46-47)
            self.files.extend(sorted(glob.glob(os.path.join(root, "test") + "/*.*")))
        self.T = np.linspace(24, 38, num=256)
        self.d = dict(enumerate(self.T.flatten(), 0))

    def __len__(self):
        return len(self.files)

    def __getitem__(self, index):
        from PIL import Image
        with Image.open(self.files[index % len(self.files)]) as img:
            return np.asarray(img.convert("RGB"))


class TestImageDataset(ImageDataset):
    """datasets_temp.py:125-158: same files once, items carry A and B only"""
    test_extends = False

    def __init__(self, root, transforms_=None, mode="test"):
        super().__init__(root, transforms_, mode)


class DeviceLoader:
    """DataLoader(ImageDataset(...), batch_size, shuffle, num_workers) of P16:484-490 with the batch assembled on the GPU. Yields dicts with the
    reference's keys: A, B, B1..B4, T_B (TestImageDataset: A, B). Files of one batch must share their geometry (the reference's datasets do);
    a batch with mixed sizes is processed per geometry group and concatenated in order."""

    def __init__(self, dataset, batch_size=1, shuffle=False, drop_last=False, num_workers=8, device="cuda:0", seed=0, out=256):
        self.ds, self.bs, self.shuffle, self.drop_last = dataset, batch_size, shuffle, drop_last
        self.device, self.out = torch.device(device), out
        self.pool = ThreadPoolExecutor(max_workers=max(1, num_workers))
        self.epoch, self.seed = 0, seed
        self.full = not isinstance(dataset, TestImageDataset)

    def __len__(self):
        n = len(self.ds)
        return n // self.bs if self.drop_last else (n + self.bs - 1) // self.bs

    def close(self):
        """stop the decode threads (also runs when the loader is garbage-collected)"""
        self.pool.shutdown(wait=False, cancel_futures=True)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _assemble(self, arrays):
        groups = {}
        for i, a in enumerate(arrays):
            groups.setdefault(a.shape, []).append(i)
        if len(groups) == 1:                                       # the usual case: one geometry, the kernel output IS the batch
            host = torch.from_numpy(np.stack(arrays))
            if self.device.type == "cuda":
                host = host.pin_memory()
            batch = pair_resize_normalize(host.to(self.device, non_blocking=True), self.out, want_temps=self.full)
            if self.full:
                batch.update(quadrants(batch["B"]))
            return batch
        parts = [None] * len(arrays)
        for shape, idx in groups.items():
            host = torch.from_numpy(np.stack([arrays[i] for i in idx]))
            if self.device.type == "cuda":
                host = host.pin_memory()
            res = pair_resize_normalize(host.to(self.device, non_blocking=True), self.out, want_temps=self.full)
            for j, i in enumerate(idx):
                parts[i] = {k: v[j] for k, v in res.items()}
        batch = {k: torch.stack([p[k] for p in parts]) for k in parts[0]}
        if self.full:
            batch.update(quadrants(batch["B"]))
        return batch

    def __iter__(self):
        n = len(self.ds)
        order = list(range(n))
        if self.shuffle:
            g = torch.Generator().manual_seed(self.seed + self.epoch)
            order = torch.randperm(n, generator=g).tolist()
        self.epoch += 1
        batches = [order[i:i + self.bs] for i in range(0, n, self.bs)]
        if self.drop_last and batches and len(batches[-1]) < self.bs:
            batches.pop()
        nxt = None
        for bi, idx in enumerate(batches):
            cur = nxt if nxt is not None else [self.pool.submit(self.ds.__getitem__, i) for i in idx]
            nxt = [self.pool.submit(self.ds.__getitem__, i) for i in batches[bi + 1]] if bi + 1 < len(batches) else None   # decode ahead
            yield self._assemble([f.result() for f in cur])
