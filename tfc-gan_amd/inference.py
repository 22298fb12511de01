"""Checkpoint / inference helpers with the call surface of the reference's test script (TFC-GAN-FFT/test_TFCGAN_16Patches.py).
Pure tensor plumbing (views, concatenation, state-dict key handling) -- the kernels are in models.py / losses.py.

  save_checkpoint(model, path)            P16:692-695  torch.save(model.state_dict()) of an nn.DataParallel-wrapped module: every
                                          key carries the 'module.' prefix
  load_clean_state(model, path)           T16:153-163  strips the 7-character prefix and load_state_dict()s
  stitch_16_patches(fake_B, real_B)       T16:217-263  per patch cat((fake_k, real_k), -2), then cat over the 16 patches on dim 1
  global_grid(real_A, fake_B, real_B)     T16:270      cat((real_A, fake_B, real_B), -2)
"""
from collections import OrderedDict

import torch

from .losses import make_16_patches


def save_checkpoint(model, path):
    """state_dict with the reference's DataParallel key prefix, so the reference's own test script can load it."""
    sd = model.state_dict()
    if not all(k.startswith("module.") for k in sd):
        sd = OrderedDict(("module." + k, v) for k, v in sd.items())
    torch.save(OrderedDict((k, v.detach().cpu()) for k, v in sd.items()), path)


def load_clean_state(model_name, checkpoint_path):
    """T16:153-163. Loads with weights_only=True (a checkpoint is data, never code)."""
    state_dict = torch.load(checkpoint_path, map_location="cpu", weights_only=True)
    new_state_dict = OrderedDict()
    for k, v in state_dict.items():
        new_state_dict[k[7:] if k.startswith("module.") else k] = v
    model_name.load_state_dict(new_state_dict)
    return model_name


def stitch_16_patches(fake_B, real_B):
    """[N,3,256,256] x2 -> [N,48,128,64]: channel group k holds fake patch k stacked over real patch k (T16:217-263)."""
    fb, rb = make_16_patches(fake_B), make_16_patches(real_B)
    return torch.cat([torch.cat((f.data, r.data), -2) for f, r in zip(fb, rb)], 1)


def global_grid(real_A, fake_B, real_B):
    """T16:270: the three images stacked vertically, [N,3,768,256]."""
    return torch.cat((real_A.data, fake_B.data, real_B.data), -2)
