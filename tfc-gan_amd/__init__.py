"""tfc-gan_amd: MI355X-native (gfx950) engine for the PATCH-16 training hot path of nudro/TFC-GAN.

Import name: ``tfc_gan_amd`` (the directory name carries a hyphen; the one-file loader ``tfc_gan_amd.py`` at the repository
root registers this package under the importable name).

Reference surface mirrored here (TFC-GAN-FFT/TFCGAN_multigpu_patchFFT_16P.py): UNetDown, UNetUp, GeneratorUNet,
Discriminator1 (alias Discriminator), weights_init_normal, make_16_patches, the 16-patch triplet head (ContrastiveLoss),
FFT_Components / fft_components / calculate_ffts, and the fused TrainStep + data-parallel layer.
"""
import os as _os

# The step runs on two HIP streams (nets.py) and, with several GPUs, beside RCCL's stream. ROCm maps a process's streams onto GPU_MAX_HW_QUEUES hardware
# queues (default 4): with a process group alive the side stream came to share a hardware queue with busy work and the two-stream step ran 1.0 ms (10 %)
# SLOWER than without collectives -- measured with a one-rank "nccl" group; 8 queues bring it back to +0.2 ms. The runtime reads the
# variable when it initialises (first GPU call), so it is set here, before anything of this package touches the GPU; an explicit setting wins.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

from . import _lib, data, engine, inference, losses, lpips, models, nets, ops, parallel, stn, stn21, synthetic  # noqa: F401
from ._lib import TfcError, build  # noqa: F401
from .engine import TrainStep  # noqa: F401
from .nets import set_wgrad_stream  # noqa: F401
from .data import DeviceLoader, ImageDataset, TestImageDataset, pair_resize_normalize  # noqa: F401
from .lpips import LPIPS  # noqa: F401
from .stn21 import STN21Step  # noqa: F401
from .stn import Warp, affine_warp, morph_gradient, morph_triplet, triplet_margin_rows  # noqa: F401
from .synthetic import synthetic_pairs, synthetic_temperatures  # noqa: F401
from .inference import global_grid, load_clean_state, save_checkpoint, stitch_16_patches  # noqa: F401
from .losses import (ContrastiveLoss, FFT_Components, calculate_ffts, color_jitter_params, color_jitter_thermal,  # noqa: F401
                     fft_components, global_fft_loss, make_16_patches, mse_spec, other_spec, patch_fft_loss, patch_first_flat_index,
                     patch_triplet_loss, sample_spectra, temperature_triplet_loss, vectorize_temps)
from .models import (BlurPool, Discriminator, Discriminator1, GeneratorUNet, UNetDown, UNetUp, get_compute_dtype,  # noqa: F401
                     set_compute_dtype, weights_init_normal)

__all__ = ["UNetDown", "UNetUp", "GeneratorUNet", "Discriminator1", "Discriminator", "BlurPool", "weights_init_normal",
           "make_16_patches", "ContrastiveLoss", "patch_triplet_loss", "FFT_Components", "fft_components", "calculate_ffts",
           "patch_fft_loss", "global_fft_loss", "mse_spec", "other_spec", "sample_spectra", "vectorize_temps", "temperature_triplet_loss", "color_jitter_thermal",
           "color_jitter_params", "synthetic_pairs", "synthetic_temperatures", "load_clean_state", "save_checkpoint", "stitch_16_patches", "global_grid", "TrainStep", "STN21Step", "LPIPS", "ImageDataset", "TestImageDataset", "DeviceLoader", "pair_resize_normalize", "Warp", "affine_warp", "morph_gradient", "morph_triplet", "triplet_margin_rows", "set_compute_dtype", "get_compute_dtype", "build", "TfcError"]
