"""ctypes binding of libtfcgan_hip.so (C ABI declared in include/tfc_gan.h).

The library is built in-tree with hipcc for gfx950 (``build()``); nothing here falls back to a CPU or PyTorch
implementation: if the shared object is missing or a call fails, a RuntimeError is raised.
"""
import ctypes
import os
import subprocess
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_HERE)
CSRC = os.path.join(_HERE, "csrc")
SO_PATH = os.path.join(_HERE, "libtfcgan_hip.so")
SOURCES = ["api.hip", "igemm.hip", "elementwise.hip", "losses.hip", "stn.hip", "lpips.hip", "input.hip", "probe.hip"]
HEADERS = ["common.h", "tfc_desc.h", "pack_math.h"]
PUBLIC_HEADER = os.path.join(_ROOT, "include", "tfc_gan.h")

DT_BF16, DT_F32 = 0, 1
OP_CONV, OP_PADCONV, OP_CONVT, OP_UPCONV, OP_CONV3 = 0, 1, 2, 3, 4
EP_BIAS, EP_STATS, EP_ACCUM, EP_TANH_NCHW, EP_LEAKY, EP_RELU = 1, 2, 4, 8, 16, 32

_lock = threading.Lock()
_lib = None


def _stale():
    if not os.path.exists(SO_PATH):
        return True
    t = os.path.getmtime(SO_PATH)
    deps = [os.path.join(CSRC, f) for f in SOURCES + HEADERS] + [PUBLIC_HEADER]
    return any(os.path.exists(d) and os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    """Compile the HIP sources for gfx950 into tfc-gan_amd/libtfcgan_hip.so (cross-compiles without a GPU). Safe under `torch.distributed.run`:
    ranks that find the library stale take a file lock, ONE of them compiles, the others wait and load its result."""
    if not force and not _stale():
        return SO_PATH
    import fcntl
    try:
        lk = open(SO_PATH + ".lock", "w")
    except OSError:                                               # read-only tree: nothing to coordinate with
        return _build_locked(verbose)
    with lk:
        fcntl.flock(lk, fcntl.LOCK_EX)
        try:
            if not force and not _stale():                        # another rank built it while this one waited
                return SO_PATH
            return _build_locked(verbose)
        finally:
            fcntl.flock(lk, fcntl.LOCK_UN)


def _build_locked(verbose):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not os.path.exists(hipcc):
        hipcc = "hipcc"
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wno-unused-value", "-Rpass-analysis=kernel-resource-usage",
           "-Wl,-rpath,/opt/rocm/lib", "-o", SO_PATH + ".tmp"] + [os.path.join(CSRC, f) for f in SOURCES]
    if verbose:
        print(" ".join(cmd))
    res = subprocess.run(cmd, stderr=subprocess.PIPE, text=True)
    if res.returncode != 0:
        raise RuntimeError("hipcc failed:\n" + "\n".join(l for l in res.stderr.splitlines() if "remark:" not in l)[-8000:])
    bad = check_no_spills(res.stderr)
    if bad:
        raise RuntimeError("hand-scheduled kernels must not spill (their asm loads carry hand-counted waits the compiler cannot see): " + "; ".join(bad))
    os.replace(SO_PATH + ".tmp", SO_PATH)
    return SO_PATH


# The K loops of tfc_igemm2_kernel and the weight-gradient kernels issue global_load / ds_read through asm volatile with hand-counted s_waitcnt: correct only
# while hipcc keeps their destination registers in place (no scratch, no VGPR spill) between the load and the wait. The compiler's resource remarks
# (-Rpass-analysis=kernel-resource-usage) are checked at every build, so a toolchain or flag change that makes one of them spill fails loudly here
# instead of silently producing wrong convolutions.
GUARDED_KERNELS = ("tfc_igemm2_kernel", "tfc_wgrad")


def check_no_spills(remarks):
    """remarks: hipcc stderr with kernel-resource-usage remarks. Returns a list of complaints about guarded kernels (empty = fine)."""
    bad, name = [], None
    for line in remarks.splitlines():
        if "Function Name:" in line:
            name = line.split("Function Name:")[1].split("[")[0].strip()
        elif name and any(g in name for g in GUARDED_KERNELS):
            for key in ("ScratchSize [bytes/lane]:", "VGPRs Spill:"):
                if key in line:
                    val = int(line.split(key)[1].split("[")[0].strip())
                    if val != 0:
                        bad.append(f"{name[:80]}: {key} {val}")
    return bad


_c = ctypes
_vp, _i, _f, _ll, _u32, _sz = _c.c_void_p, _c.c_int, _c.c_float, _c.c_longlong, _c.c_uint32, _c.c_size_t

# name -> (restype, argtypes); mirrors include/tfc_gan.h one to one
PROTOTYPES = {
    "tfc_last_error": (_c.c_char_p, []),
    "tfc_abi_version": (_i, []),
    "tfc_part_ws_floats": (_sz, []),
    "tfc_conv_packed_bytes": (_sz, [_i, _i, _i, _i, _i]),
    "tfc_conv_pack": (_i, [_vp, _i, _i, _i, _vp, _vp, _vp, _i, _i]),
    "tfc_pack_plan_bytes": (_sz, [_i]),
    "tfc_pack_plan_build": (_i, [_i, _i, _c.POINTER(_i), _c.POINTER(_i), _c.POINTER(_vp), _c.POINTER(_vp), _c.POINTER(_i), _c.POINTER(_i),
                                 _vp, _c.POINTER(_i)]),
    "tfc_conv_pack_planned": (_i, [_vp, _i, _vp, _i, _i]),
    "tfc_conv_fwd": (_i, [_vp, _i, _i, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp, _i, _vp, _vp, _vp, _vp, _i, _vp]),
    "tfc_conv_dgrad_image": (_i, [_vp, _i, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp, _i, _vp]),
    "tfc_upconv_head_fwd": (_i, [_vp, _i, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp]),
    "tfc_upconv_head_dgrad": (_i, [_vp, _i, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp, _i]),
    "tfc_patchgan_head_fwd": (_i, [_vp, _i, _vp, _i, _i, _i, _i, _i, _vp, _vp, _i]),
    "tfc_conv_dgrad": (_i, [_vp, _i, _i, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp, _i, _vp, _i]),
    "tfc_conv_wgrad_ws_bytes": (_sz, [_i, _i, _i]),
    "tfc_conv_wgrad": (_i, [_vp, _i, _i, _vp, _i, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp, _i]),
    "tfc_act_fwd": (_i, [_vp, _i, _vp, _i, _i, _i, _i, _i, _vp, _i, _f, _i, _f, _u32, _vp, _i, _vp, _vp]),
    "tfc_act_bwd_signs": (_i, [_vp, _i, _vp, _i, _vp, _i, _i, _i, _i, _f, _vp, _vp, _i, _vp]),
    "tfc_act_bwd": (_i, [_vp, _i, _i, _vp, _i, _vp, _i, _i, _i, _i, _i, _vp, _i, _f, _i, _f, _u32, _vp, _vp, _i, _vp]),
    "tfc_dropout_mask": (_i, [_vp, _vp, _ll, _f, _u32]),
    "tfc_pack_nhwc8": (_i, [_vp, _i, _vp, _i, _vp, _i, _vp, _i, _i, _i]),
    "tfc_unpack_nchw": (_i, [_vp, _i, _vp, _i, _i, _i, _vp, _i, _i, _i, _f, _f]),
    "tfc_tanh_bwd_pack": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "tfc_colsum": (_i, [_vp, _i, _vp, _ll, _i, _i, _vp, _vp]),
    "tfc_cast": (_i, [_vp, _i, _i, _vp, _vp, _ll]),
    "tfc_axpby": (_i, [_vp, _vp, _vp, _vp, _ll, _f, _f]),
    "tfc_spectral_norm_step": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i]),
    "tfc_spectral_norm_batched_ws_floats": (_sz, [_i, _c.POINTER(_i), _c.POINTER(_i)]),
    "tfc_spectral_norm_step_batched": (_i, [_vp, _i, _c.POINTER(_vp), _c.POINTER(_vp), _c.POINTER(_vp), _c.POINTER(_vp), _c.POINTER(_vp),
                                            _c.POINTER(_vp), _c.POINTER(_i), _c.POINTER(_i), _vp, _i]),
    "tfc_spectral_norm_bwd": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i]),
    "tfc_patch16_triplet": (_i, [_vp, _vp, _vp, _c.POINTER(_i), _i, _i, _vp, _vp, _f]),
    "tfc_fft_spectrum": (_i, [_vp, _vp, _ll, _ll, _i, _i, _i, _i, _i, _i, _vp, _vp, _i, _vp]),
    "tfc_fft_spectrum_ws_bytes": (_sz, [_i, _i]),
    "tfc_logmag_mse": (_i, [_vp, _vp, _vp, _i, _i, _vp]),
    "tfc_logmag_mae": (_i, [_vp, _vp, _vp, _i, _i, _vp]),
    "tfc_vectorize_temps": (_i, [_vp, _vp, _ll, _i, _i, _i, _i, _vp, _vp]),
    "tfc_row_triplet": (_i, [_vp, _vp, _vp, _vp, _ll, _i, _f, _vp]),
    "tfc_l1_sum": (_i, [_vp, _vp, _vp, _ll, _f, _vp, _i]),
    "tfc_affine_warp_fwd": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i]),
    "tfc_affine_warp_bwd": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "tfc_morph_grad_fwd": (_i, [_vp, _vp, _vp, _vp, _ll, _i, _i]),
    "tfc_morph_grad_bwd": (_i, [_vp, _vp, _vp, _vp, _ll, _i, _i]),
    "tfc_row_triplet_grad": (_i, [_vp, _vp, _vp, _vp, _ll, _i, _f, _f, _vp, _vp]),
    "tfc_first_block_bwd_supported": (_i, [_i, _i, _i]),
    "tfc_first_block_bwd_wgrad": (_i, [_vp, _i, _vp, _i, _vp, _i, _vp, _i, _i, _i, _i, _i, _i, _f, _vp, _vp, _i, _vp, _vp, _vp]),
    "tfc_conv_first_fwd": (_i, [_vp, _i, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp, _i, _vp, _vp, _i, _vp]),
    "tfc_first_block_fwd": (_i, [_vp, _i, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _f, _i, _vp, _i, _vp]),
    "tfc_resize_plan_bytes": (_sz, [_i, _i, _i]),
    "tfc_resize_plan_build": (_i, [_i, _i, _i, _vp]),
    "tfc_pair_resize_ws_bytes": (_sz, [_i, _i, _i]),
    "tfc_pair_resize_normalize": (_i, [_vp, _vp, _ll, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "tfc_lpips_input_fwd": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i]),
    "tfc_lpips_input_bwd": (_i, [_vp, _i, _vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _i]),
    "tfc_maxpool2_fwd": (_i, [_vp, _i, _vp, _vp, _i, _i, _i, _i]),
    "tfc_maxpool2_bwd": (_i, [_vp, _i, _vp, _vp, _vp, _i, _i, _i, _i]),
    "tfc_relu_bwd": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _ll]),
    "tfc_lpips_head": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _f]),
    "tfc_bce_relativistic": (_i, [_vp, _i, _vp, _vp, _i, _i, _f, _f, _i, _vp, _vp, _vp, _f]),
    "tfc_adam_step": (_i, [_vp, _vp, _vp, _vp, _vp, _ll, _f, _f, _f, _f, _i, _f]),
    "tfc_prof_enable": (_i, [_i]),
    "tfc_prof_collect": (_i, [_i, _c.POINTER(_c.c_double), _c.POINTER(_c.c_double), _c.POINTER(_ll)]),
    "tfc_prof_records": (_i, [_i, _c.POINTER(_i), _c.POINTER(_c.c_double), _c.POINTER(_c.c_double), _c.POINTER(_i)]),
    "tfc_host_emulate_conv": (_i, [_i, _i, _i, _vp, _vp, _vp, _i, _i, _i, _i, _i]),
    "tfc_debug_set_igemm_config": (_i, [_i]),
    "tfc_probe_mfma": (_i, [_vp, _vp]),
}


def load():
    """Load the shared object (building it when stale) and attach prototypes. torch must be imported first so that the
    process already holds the HIP runtime torch ships; our library then binds to that same libamdhip64.so.7."""
    global _lib
    with _lock:
        if _lib is not None:
            return _lib
        import torch  # noqa: F401  (HIP runtime first)
        so = os.environ.get("TFC_SO_OVERRIDE")                    # development only: A/B a differently compiled library
        if not so:
            build()
            so = SO_PATH
        lib = ctypes.CDLL(so, mode=ctypes.RTLD_GLOBAL)
        for name, (res, args) in PROTOTYPES.items():
            fn = getattr(lib, name)      # AttributeError if the library does not export a declared symbol
            fn.restype = res
            fn.argtypes = args
        _lib = lib
        return lib


class TfcError(RuntimeError):
    pass


def check(rc, what=""):
    if rc != 0:
        msg = load().tfc_last_error().decode("utf-8", "replace")
        raise TfcError(f"libtfcgan_hip: {what} failed ({rc}): {msg}")
