"""Synthetic PATCH-16 inputs of the benchmark contract: visible image A = uint8 U{0..255} per channel, thermal
image B = one uint8 channel replicated to three (thermal frames are R = G = B, datasets_temp.py:33), both mapped to [-1, 1] by
x / 127.5 - 1 exactly as ToTensor + Normalize((.5,.5,.5), (.5,.5,.5)) does (P16:479-482); T_B = 24 + 14 * u8 / 255 (datasets_temp.py:43-44)."""
import numpy as np
import torch


def synthetic_pairs(n, seed=1234, size=256):
    """-> (real_A, real_B) fp32 NCHW [n,3,size,size] on the CPU; a numpy Generator stream, identical on every platform"""
    rng = np.random.default_rng(seed)
    a = rng.integers(0, 256, size=(n, 3, size, size), dtype=np.uint8)
    b = np.repeat(rng.integers(0, 256, size=(n, 1, size, size), dtype=np.uint8), 3, axis=1)
    f = lambda u: torch.from_numpy(u.astype(np.float32) / 127.5 - 1.0)  # noqa: E731
    return f(a), f(b)


def synthetic_temperatures(n, seed=1234, size=256):
    """-> T_B fp32 [n,size,size]: the dataset's per-pixel temperatures of a random thermal frame"""
    u8 = np.random.default_rng(seed + 7919).integers(0, 256, size=(n, size, size))
    return torch.from_numpy((24.0 + 14.0 * u8 / 255.0).astype(np.float32))
