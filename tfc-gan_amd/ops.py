"""Thin torch-tensor wrappers over the C ABI (include/tfc_gan.h). torch is used for device memory and streams only.

Activations are torch tensors of physical shape [N, H, W, pitch] (NHWC); a `View` names a channel window
[coff, coff + C) of such a buffer, which is how skip connections share one concat buffer without copies.
"""
import ctypes
import os
from dataclasses import dataclass

import torch

from . import _lib
from ._lib import (DT_BF16, DT_F32, EP_ACCUM, EP_BIAS, EP_LEAKY, EP_RELU, EP_STATS, EP_TANH_NCHW, OP_CONV, OP_CONV3, OP_CONVT, OP_PADCONV, OP_UPCONV,
                   check)


def lib():
    return _lib.load()


def torch_dtype(dt):
    return torch.bfloat16 if dt == DT_BF16 else torch.float32


def dt_of(dtype):
    if dtype == torch.bfloat16:
        return DT_BF16
    if dtype == torch.float32:
        return DT_F32
    raise ValueError(f"compute dtype must be torch.bfloat16 or torch.float32, got {dtype}")


def pad8(c):
    return (c + 7) // 8 * 8


def stream_ptr():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def require_gpu(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise _lib.TfcError("libtfcgan_hip kernels need CUDA/HIP tensors; got a CPU tensor "
                                "(there is no CPU fallback in this package)")


@dataclass
class View:
    """Channel window of an NHWC buffer."""
    t: torch.Tensor      # [N, H, W, pitch]
    C: int               # logical channels
    coff: int = 0

    @property
    def N(self):
        return self.t.shape[0]

    @property
    def H(self):
        return self.t.shape[1]

    @property
    def W(self):
        return self.t.shape[2]

    @property
    def pitch(self):
        return self.t.shape[3]

    @property
    def ptr(self):
        return ctypes.c_void_p(self.t.data_ptr() + self.coff * self.t.element_size())

    def sub(self, coff, C):
        return View(self.t, C, self.coff + coff)


class ZeroArena:
    """Small zero-initialised fp32 scratch (InstanceNorm statistics, bias-gradient partials, loss scalars) for one training step:
    ONE fill per step instead of one torch.zeros launch per buffer (~33 per step). `begin()` re-zeroes the part handed out since
    the previous begin(); views stay valid until the next begin() -- i.e. for the rest of the step that took them."""

    def __init__(self, device, nfloats=1 << 20):
        self.buf = torch.zeros(nfloats, dtype=torch.float32, device=device)
        self.used = 0
        self.active = False

    def begin(self):
        if self.used:
            self.buf[:self.used].zero_()
        self.used = 0
        self.active = True

    def take(self, shape):
        n = 1
        for d in shape:
            n *= int(d)
        n4 = (n + 63) // 64 * 64                                  # 256-byte granules: every view stays 16-byte aligned
        if self.used + n4 > self.buf.numel():
            return torch.zeros(shape, dtype=torch.float32, device=self.buf.device)
        v = self.buf[self.used:self.used + n].view(shape)
        self.used += n4
        return v


_ARENA = {}


def arena_begin(device):
    """start a step: from now on zeros_f32() on this device is served from the step arena"""
    dev = torch.device(device)
    a = _ARENA.get(dev)
    if a is None:
        a = _ARENA[dev] = ZeroArena(dev)
    a.begin()
    return a


def arena_end(device):
    """end of the step: later zeros_f32() calls (module-level forward / backward outside TrainStep) get buffers of their own"""
    a = _ARENA.get(torch.device(device))
    if a is not None:
        a.active = False


def zeros_f32(shape, device):
    a = _ARENA.get(torch.device(device))
    return a.take(tuple(shape)) if a is not None and a.active else torch.zeros(shape, dtype=torch.float32, device=device)


def new_act(N, H, W, C, dt, device, zero=False):
    f = torch.zeros if zero else torch.empty
    return View(f((N, H, W, C), dtype=torch_dtype(dt), device=device), C)


def _p(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


# Partial-sum scratch of the reducing entry points (include/tfc_gan.h, "DETERMINISM"): sums that cross workgroups leave each workgroup as a partial
# in a fixed slot of this buffer and are added in a fixed order behind the kernel -- no float atomics anywhere on the training path. One buffer per
# (device, stream): launches on one stream are ordered, and a kernel's partials are consumed by the reduction queued right behind it.
_PART_WS = {}


def part_ws(device):
    key = (torch.device(device), torch.cuda.current_stream(device).cuda_stream)
    ws = _PART_WS.get(key)
    if ws is None:
        ws = _PART_WS[key] = torch.empty(lib().tfc_part_ws_floats(), dtype=torch.float32, device=device)
    return ctypes.c_void_p(ws.data_ptr())


OUT_HW = {OP_CONV: lambda h: h - 1, OP_PADCONV: lambda h: h, OP_CONVT: lambda h: 2 * h, OP_UPCONV: lambda h: 2 * h, OP_CONV3: lambda h: h}


# ---- convolution family ---------------------------------------------------------------------------------------
def packed_bytes(dt, op, pas, Cin, Cout):
    return lib().tfc_conv_packed_bytes(dt, op, pas, Cin, Cout)


def pack_weight(dt, op, pas, w, Cin, Cout, scale=None, out=None):
    """w: torch-layout fp32 weight on the GPU; returns a uint8 tensor holding the MFMA operand stream."""
    require_gpu(w)
    assert w.dtype == torch.float32 and w.is_contiguous()
    nbytes = packed_bytes(dt, op, pas, Cin, Cout)
    if out is None:
        out = torch.empty(nbytes, dtype=torch.uint8, device=w.device)
    assert out.numel() >= nbytes
    check(lib().tfc_conv_pack(stream_ptr(), dt, op, pas, _p(w), _p(scale), _p(out), Cin, Cout), "tfc_conv_pack")
    return out


class PackPlan:
    """All operand streams of one network packed by a single launch. jobs: list of (op, pass, weight fp32 tensor, Cin, Cout);
    the weight tensors and the returned stream buffers must stay where they are (flat parameter buffer, persistent streams)."""

    def __init__(self, dt, jobs):
        self.dt = dt
        dev = jobs[0][2].device
        n = len(jobs)
        self.streams = [torch.empty(packed_bytes(dt, op, pas, cin, cout), dtype=torch.uint8, device=dev) for op, pas, w, cin, cout in jobs]
        self.keep = [w for _, _, w, _, _ in jobs]
        host = (ctypes.c_uint8 * lib().tfc_pack_plan_bytes(n))()
        nblk = ctypes.c_int(0)
        ci = lambda xs: (ctypes.c_int * n)(*xs)  # noqa: E731
        cp = lambda ts: (ctypes.c_void_p * n)(*[t.data_ptr() for t in ts])  # noqa: E731
        nj = lib().tfc_pack_plan_build(dt, n, ci([j[0] for j in jobs]), ci([j[1] for j in jobs]), cp(self.keep), cp(self.streams),
                                       ci([j[3] for j in jobs]), ci([j[4] for j in jobs]), ctypes.cast(host, ctypes.c_void_p), ctypes.byref(nblk))
        if nj < 0:
            check(nj, "tfc_pack_plan_build")
        self.njobs, self.nblocks = nj, nblk.value
        self.plan = torch.frombuffer(bytearray(host), dtype=torch.uint8).to(dev)

    def run(self):
        check(lib().tfc_conv_pack_planned(stream_ptr(), self.dt, _p(self.plan), self.njobs, self.nblocks), "tfc_conv_pack_planned")


def conv_fwd(dt, op, x: View, Cin, Cout, packed, y: View = None, bias=None, stats=None, out_nchw=None, flags=0, oscale=None):
    if bias is not None:
        flags |= EP_BIAS
    if stats is not None:
        flags |= EP_STATS
    if out_nchw is not None:
        flags |= EP_TANH_NCHW
    check(lib().tfc_conv_fwd(stream_ptr(), dt, op, x.ptr, x.pitch, x.N, x.H, x.W, Cin, Cout, _p(packed),
                             None if y is None else y.ptr, 0 if y is None else y.pitch, _p(bias), _p(stats), _p(out_nchw), _p(oscale), flags,
                             part_ws(x.t.device) if stats is not None else None), "tfc_conv_fwd")


def first_block_bwd_supported(dt, Cin, Cout):
    if os.environ.get("TFC_NO_FUSED_FIRST_BWD"):                  # A/B knob for profiling
        return False
    return bool(lib().tfc_first_block_bwd_supported(dt, Cin, Cout))


def first_block_fwd_supported():
    return not os.environ.get("TFC_NO_FUSED_FIRST_FWD")           # A/B knob for profiling (read per call)


def conv_first_fwd(dt, x: View, Cin, Cout, packed, y: View, bias=None, oscale=None, flags=0, sign_mask=None):
    """first convolution of a network (8 padded input channels -> 64) on the weights-stationary kernel; sign_mask: uint8 [N, H-1, W-1, 8] that receives
    one bit per stored value (> 0) for the fused backward of the block"""
    if bias is not None:
        flags |= EP_BIAS
    check(lib().tfc_conv_first_fwd(stream_ptr(), dt, x.ptr, x.pitch, x.N, x.H, x.W, Cin, Cout, _p(packed), y.ptr, y.pitch, _p(bias), _p(oscale), flags,
                                   _p(sign_mask)), "tfc_conv_first_fwd")


def first_block_fwd(dt, x: View, Cin, Cout, packed, out: View, bias=None, oscale=None, slope=0.2, act_after_rounding=False, sign_mask=None):
    """conv -> [+bias, x 1/sigma] -> LeakyReLU -> BlurPool(stride 2) of a first block in one kernel (the conv output is never written); `out` may be a
    channel window of a wider buffer"""
    check(lib().tfc_first_block_fwd(stream_ptr(), dt, x.ptr, x.pitch, x.N, x.H, x.W, Cin, Cout, _p(packed), _p(bias), _p(oscale), slope,
                                    1 if act_after_rounding else 0, out.ptr, out.pitch, _p(sign_mask)), "tfc_first_block_fwd")


def first_block_bwd_wgrad(dt, x: View, y: View, dy_pooled: View, Cin, Cout, dw, slope=0.2, accumulate=False, ws=None, bias_sums=None, sign_mask=None):
    """[BlurPool]^T -> LeakyReLU' -> weight (+ bias) gradient of the first block in one kernel: the gradient of the conv output is never written.
    sign_mask (from conv_first_fwd): read instead of the stored activation y (8 bytes per pixel instead of 128)"""
    nbytes = lib().tfc_conv_wgrad_ws_bytes(OP_CONV, Cin, Cout)
    if ws is None or ws.numel() * ws.element_size() < nbytes:
        nbig = lib().tfc_conv_wgrad_ws_bytes(OP_CONV, 1024, 512)
        ws = torch.zeros(max(nbytes, nbig), dtype=torch.uint8, device=x.t.device)
    assert dw.dtype == torch.float32 and dw.is_contiguous()
    check(lib().tfc_first_block_bwd_wgrad(stream_ptr(), dt, x.ptr, x.pitch, None if y is None else y.ptr, 0 if y is None else y.pitch, dy_pooled.ptr,
                                          dy_pooled.pitch, x.N, x.H, x.W, Cin, Cout, slope, _p(ws), _p(dw), 1 if accumulate else 0, _p(bias_sums),
                                          part_ws(x.t.device) if bias_sums is not None else None, _p(sign_mask)), "tfc_first_block_bwd_wgrad")
    return ws


def conv_dgrad_image(dt, dy: View, N, H, W, Cin, w, oscale, nch):
    """first discriminator conv, gradient w.r.t. its first `nch` input channels as fp32 NCHW [N,nch,H,W] (bf16 path)"""
    out = torch.empty((N, nch, H, W), dtype=torch.float32, device=dy.t.device)
    check(lib().tfc_conv_dgrad_image(stream_ptr(), dt, dy.ptr, dy.pitch, N, H, W, Cin, w.shape[0], _p(w), _p(oscale), nch, _p(out)),
          "tfc_conv_dgrad_image")
    return out


def upconv_head_fwd(dt, x: View, w, bias, out_nchw):
    """generator head (upsample + pad + conv(128 -> C<=4) + tanh), bf16: x NHWC View with 128 channels, w torch-layout fp32"""
    Cout = w.shape[0]
    check(lib().tfc_upconv_head_fwd(stream_ptr(), dt, x.ptr, x.pitch, x.N, x.H, x.W, 128, Cout, _p(w), _p(bias), _p(out_nchw)),
          "tfc_upconv_head_fwd")


def upconv_head_dgrad(dt, dy: View, N, H, W, w, dx: View):
    """input gradient of the generator head (bf16, 128 input channels, <= 8 output channels): dy NHWC8 at 2H x 2W -> dx [N,H,W,128]"""
    check(lib().tfc_upconv_head_dgrad(stream_ptr(), dt, dy.ptr, dy.pitch, N, H, W, 128, w.shape[0], _p(w), dx.ptr, dx.pitch), "tfc_upconv_head_dgrad")


def patchgan_head_fwd(dt, x: View, w, y: View):
    check(lib().tfc_patchgan_head_fwd(stream_ptr(), dt, x.ptr, x.pitch, x.N, x.H, x.W, x.C, _p(w), y.ptr, y.pitch), "tfc_patchgan_head_fwd")


def conv_dgrad(dt, op, dy: View, N, H, W, Cin, Cout, packed, dx: View, accumulate=False, oscale=None):
    check(lib().tfc_conv_dgrad(stream_ptr(), dt, op, dy.ptr, dy.pitch, N, H, W, Cin, Cout, _p(packed), dx.ptr, dx.pitch,
                               _p(oscale), EP_ACCUM if accumulate else 0), "tfc_conv_dgrad")


def conv_wgrad(dt, op, x: View, dy: View, Cin, Cout, dw, accumulate=False, ws=None):
    nbytes = lib().tfc_conv_wgrad_ws_bytes(op, Cin, Cout)
    if ws is None or ws.numel() * ws.element_size() < nbytes:
        nbig = lib().tfc_conv_wgrad_ws_bytes(op, 1024, 512)       # the largest layer of the path, so one buffer serves every call
        ws = torch.zeros(max(nbytes, nbig), dtype=torch.uint8, device=x.t.device)   # zero ONCE: the kernels re-zero the accumulator
    assert dw.dtype == torch.float32 and dw.is_contiguous()
    check(lib().tfc_conv_wgrad(stream_ptr(), dt, op, x.ptr, x.pitch, dy.ptr, dy.pitch, x.N, x.H, x.W, Cin, Cout, _p(ws), _p(dw),
                               1 if accumulate else 0), "tfc_conv_wgrad")
    return ws


# ---- fused norm / activation / blur-pool ----------------------------------------------------------------------
def act_fwd(dt, x: View, y: View, stats=None, slope=0.2, pool=0, drop_p=0.0, seed=0, stats_out=None):
    check(lib().tfc_act_fwd(stream_ptr(), dt, x.ptr, x.pitch, x.N, x.H, x.W, x.C, _p(stats), 0 if stats is None else 1, slope, pool,
                            drop_p, seed & 0xFFFFFFFF, y.ptr, y.pitch, _p(stats_out), part_ws(x.t.device) if stats_out is not None else None), "tfc_act_fwd")


def act_bwd(dt, mode, dy: View, x: View, N, H, W, C, dx: View = None, stats=None, slope=0.2, pool=0, drop_p=0.0, seed=0, rstats=None):
    check(lib().tfc_act_bwd(stream_ptr(), dt, mode, dy.ptr, dy.pitch, None if x is None else x.ptr, 0 if x is None else x.pitch,
                            N, H, W, C, _p(stats), 0 if stats is None else 1, slope, pool, drop_p, seed & 0xFFFFFFFF, _p(rstats),
                            None if dx is None else dx.ptr, 0 if dx is None else dx.pitch, part_ws(dy.t.device) if rstats is not None else None),
          "tfc_act_bwd")


def act_bwd_signs(dt, dy: View, sign_mask, N, H, W, C, dx: View, slope=0.2, rstats=None):
    """act_bwd(mode 0, pool 2) of a first block whose conv output was never stored: LeakyReLU' from the sign words of first_block_fwd / conv_first_fwd"""
    check(lib().tfc_act_bwd_signs(stream_ptr(), dt, dy.ptr, dy.pitch, _p(sign_mask), N, H, W, C, slope, _p(rstats), dx.ptr, dx.pitch,
                                  part_ws(dy.t.device) if rstats is not None else None), "tfc_act_bwd_signs")


def dropout_mask(n, drop_p, seed, device):
    out = torch.empty(n, dtype=torch.uint8, device=device)
    check(lib().tfc_dropout_mask(stream_ptr(), _p(out), n, drop_p, seed & 0xFFFFFFFF), "tfc_dropout_mask")
    return out


# ---- layout plumbing ------------------------------------------------------------------------------------------
def pack_nhwc8(dt, a, b=None):
    """NCHW fp32 a [N,Ca,H,W] (and b) -> NHWC with 8 channels."""
    require_gpu(a, b)
    a = a.contiguous().float()
    N, Ca, H, W = a.shape
    Cb = 0
    if b is not None:
        b = b.contiguous().float()
        Cb = b.shape[1]
    out = new_act(N, H, W, 8, dt, a.device)
    check(lib().tfc_pack_nhwc8(stream_ptr(), dt, _p(a), Ca, _p(b), Cb, out.ptr, N, H, W), "tfc_pack_nhwc8")
    return out


def unpack_nchw(dt, v: View, C, out=None, alpha=1.0, beta=0.0, c0=0):
    N, H, W = v.N, v.H, v.W
    if out is None:
        out = torch.empty((N, C, H, W), dtype=torch.float32, device=v.t.device)
    check(lib().tfc_unpack_nchw(stream_ptr(), dt, v.ptr, v.pitch, c0, C, _p(out), N, H, W, alpha, beta), "tfc_unpack_nchw")
    return out


def tanh_bwd_pack(dt, g, y, dbias=None):
    N, C, H, W = g.shape
    out = new_act(N, H, W, 8, dt, g.device)
    check(lib().tfc_tanh_bwd_pack(stream_ptr(), dt, _p(g), _p(y), out.ptr, _p(dbias), N, C, H, W, part_ws(g.device) if dbias is not None else None),
          "tfc_tanh_bwd_pack")
    return out


def colsum(dt, v: View, out):
    rows = v.N * v.H * v.W
    check(lib().tfc_colsum(stream_ptr(), dt, v.ptr, rows, v.pitch, v.C, _p(out), part_ws(v.t.device) if rows >= 4096 else None), "tfc_colsum")


def cast_from_f32(dt, x):
    y = torch.empty(x.shape, dtype=torch_dtype(dt), device=x.device)
    check(lib().tfc_cast(stream_ptr(), dt, 0, _p(x.contiguous()), _p(y), x.numel()), "tfc_cast")
    return y


def cast_to_f32(dt, x):
    y = torch.empty(x.shape, dtype=torch.float32, device=x.device)
    check(lib().tfc_cast(stream_ptr(), dt, 1, _p(x.contiguous()), _p(y), x.numel()), "tfc_cast")
    return y


def axpby(out, x, y, a, b):
    check(lib().tfc_axpby(stream_ptr(), _p(out), _p(x), _p(y), out.numel(), a, b), "tfc_axpby")
    return out


# ---- spectral norm --------------------------------------------------------------------------------------------
def spectral_norm_step(W, u, v, sigma2, power_iter=True, ws=None):
    R = W.shape[0]
    K = W.numel() // R
    need = lib().tfc_spectral_norm_batched_ws_floats(1, (ctypes.c_int * 1)(R), (ctypes.c_int * 1)(K))
    if ws is None or ws.numel() < need:
        ws = torch.empty(need, dtype=torch.float32, device=W.device)
    check(lib().tfc_spectral_norm_step(stream_ptr(), _p(W), _p(u), _p(v), _p(sigma2), _p(ws), R, K, 1 if power_iter else 0),
          "tfc_spectral_norm_step")


def spectral_norm_step_batched(Ws, us, vs, sigma2s, power_iter=True, u_snaps=None, v_snaps=None, ws=None):
    """one power iteration (u <- norm(W v), v <- norm(W^T u), sigma = u.W v) for up to 4 layers in 3 launches"""
    n = len(Ws)
    R = [int(W.shape[0]) for W in Ws]
    K = [int(W.numel() // W.shape[0]) for W in Ws]
    Ra, Ka = (ctypes.c_int * n)(*R), (ctypes.c_int * n)(*K)
    need = lib().tfc_spectral_norm_batched_ws_floats(n, Ra, Ka)
    if ws is None or ws.numel() < need:
        ws = torch.empty(need, dtype=torch.float32, device=Ws[0].device)

    def arr(ts):
        return None if ts is None else (ctypes.c_void_p * n)(*[None if t is None else t.data_ptr() for t in ts])
    check(lib().tfc_spectral_norm_step_batched(stream_ptr(), n, arr(Ws), arr(us), arr(vs), arr(sigma2s), arr(u_snaps), arr(v_snaps),
                                               Ra, Ka, _p(ws), 1 if power_iter else 0), "tfc_spectral_norm_step_batched")
    return ws


def spectral_norm_bwd(G, W, u, v, sigma2, gout, accumulate=False):
    R = W.shape[0]
    K = W.numel() // R
    ws = torch.empty(256, dtype=torch.float32, device=W.device)    # per-workgroup partials of <G, W> (doubles), added in a fixed order
    check(lib().tfc_spectral_norm_bwd(stream_ptr(), _p(G), _p(W), _p(u), _p(v), _p(sigma2), _p(ws), _p(gout), R, K,
                                      1 if accumulate else 0), "tfc_spectral_norm_bwd")


# ---- loss heads -----------------------------------------------------------------------------------------------
def patch16_triplet(fake, real, neg_idx, want_grad=True, gscale=1.0):
    """fake/real: fp32 NCHW [N,C,256,256]; neg_idx: 16 ints. Returns (loss[1], dfake or None)."""
    require_gpu(fake, real)
    assert fake.shape == real.shape and fake.shape[2:] == (256, 256), "make_16_patches hard-codes 256x256 (reference :233-251)"
    fake = fake.contiguous().float()
    real = real.contiguous().float()
    N, C = fake.shape[:2]
    loss = torch.empty(1, dtype=torch.float32, device=fake.device)
    dfake = torch.empty_like(fake) if want_grad else None
    idx = (ctypes.c_int * 16)(*[int(i) for i in neg_idx])
    check(lib().tfc_patch16_triplet(stream_ptr(), _p(fake), _p(real), idx, N, C, _p(loss), _p(dfake), gscale), "tfc_patch16_triplet")
    return loss, dfake


_FFT_WS = {}


def fft_spectrum(img, S, wins_x, wins_y, shift=True, direct=False):
    """img: fp32 [N,C,H,W] (any strides on N/C/H, unit stride on W). Returns amp, pha [N*wins_x*wins_y, S, S//2+1].
    direct=True: the direct-DFT kernel (no scratch) instead of the LDS radix-4 FFT -- the tests cross-check the two."""
    require_gpu(img)
    if img.dtype != torch.float32:
        img = img.float()
    if img.stride(3) != 1:
        img = img.contiguous()
    N, C = img.shape[:2]
    assert img.shape[2] >= wins_y * S and img.shape[3] >= wins_x * S
    nwin = N * wins_x * wins_y
    amp = torch.empty((nwin, S, S // 2 + 1), dtype=torch.float32, device=img.device)
    pha = torch.empty_like(amp)
    ws = None
    if not direct:
        need = lib().tfc_fft_spectrum_ws_bytes(S, nwin)
        key = (img.device, torch.cuda.current_stream().cuda_stream)   # one scratch per (device, stream): calls on a stream are ordered
        ws = _FFT_WS.get(key)
        if ws is None or ws.numel() < need:
            ws = _FFT_WS[key] = torch.empty(need, dtype=torch.uint8, device=img.device)
    check(lib().tfc_fft_spectrum(stream_ptr(), _p(img), img.stride(0), img.stride(1), img.stride(2), C, S, wins_x, wins_y, N,
                                 _p(amp), _p(pha), 1 if shift else 0, _p(ws)), "tfc_fft_spectrum")
    return amp, pha


def logmag_mse(amp_a, amp_b, absolute=False):
    """per window mean squared (or absolute) difference of the log-magnitude spectra over the FULL S x S spectrum"""
    nwin, S = amp_a.shape[0], amp_a.shape[1]
    out = torch.empty(nwin, dtype=torch.float32, device=amp_a.device)
    fn = lib().tfc_logmag_mae if absolute else lib().tfc_logmag_mse
    check(fn(stream_ptr(), _p(amp_a), _p(amp_b), S, nwin, _p(out)), "tfc_logmag_mae" if absolute else "tfc_logmag_mse")
    return out


def vectorize_temps(img, lut):
    """img: fp32 [N,C,H,W] (unit stride on W); lut: fp32 [256] on the device. Returns [N,H,W] fp32 (channel 0 -> uint8 -> lut)."""
    require_gpu(img)
    if img.dtype != torch.float32:
        img = img.float()
    if img.stride(3) != 1:
        img = img.contiguous()
    N, _, H, W = img.shape
    out = torch.empty((N, H, W), dtype=torch.float32, device=img.device)
    check(lib().tfc_vectorize_temps(stream_ptr(), _p(img), img.stride(0), img.stride(2), N, H, W, _p(lut), _p(out)), "tfc_vectorize_temps")
    return out


def row_triplet(anchor, positive, negative, margin=1.0):
    """nn.TripletMarginLoss(margin, p=2) over the last dim of three equal-shape contiguous fp32 tensors -> scalar [1]."""
    require_gpu(anchor)
    assert anchor.shape == positive.shape == negative.shape
    a, p_, n = (t.float().contiguous() for t in (anchor, positive, negative))
    W = a.shape[-1]
    out = torch.empty(1, dtype=torch.float32, device=a.device)
    check(lib().tfc_row_triplet(stream_ptr(), _p(a), _p(p_), _p(n), a.numel() // W, W, float(margin), _p(out)), "tfc_row_triplet")
    return out


def l1_sum(a, b, scale, out, zero_first=False):
    check(lib().tfc_l1_sum(stream_ptr(), _p(a), _p(b), a.numel(), scale, _p(out), 1 if zero_first else 0), "tfc_l1_sum")


def bce_relativistic(dt, a: View, b: View, mode, t1, t2=0.0, da: View = None, db: View = None, gscale=1.0):
    n = a.N * a.H * a.W
    loss = torch.empty(1, dtype=torch.float32, device=a.t.device)
    check(lib().tfc_bce_relativistic(stream_ptr(), dt, a.ptr, b.ptr, n, a.pitch, t1, t2, mode, _p(loss),
                                     None if da is None else da.ptr, None if db is None else db.ptr, gscale), "tfc_bce_relativistic")
    return loss


def adam_step(p, g, m, v, lr, b1, b2, eps, step, gscale=1.0):
    check(lib().tfc_adam_step(stream_ptr(), _p(p), _p(g), _p(m), _p(v), p.numel(), lr, b1, b2, eps, step, gscale), "tfc_adam_step")


# ---- measurement ----------------------------------------------------------------------------------------------
def prof_enable(on):
    check(lib().tfc_prof_enable(1 if on else 0), "tfc_prof_enable")


def prof_records(max_records=4096):
    """per-call records since the last collect: list of dicts {kclass, ms, flop, op, pass, N, H, W, Cin, Cout} (synchronise first)"""
    kc = (ctypes.c_int * max_records)()
    ms = (ctypes.c_double * max_records)()
    fl = (ctypes.c_double * max_records)()
    meta = (ctypes.c_int * (7 * max_records))()
    n = lib().tfc_prof_records(max_records, kc, ms, fl, meta)
    if n < 0:
        check(n, "tfc_prof_records")
    keys = ("op", "pass", "N", "H", "W", "Cin", "Cout")
    return [dict(kclass=kc[i], ms=ms[i], flop=fl[i], **{k: meta[7 * i + j] for j, k in enumerate(keys)}) for i in range(n)]


def prof_collect(kclass):
    ms, fl, n = ctypes.c_double(), ctypes.c_double(), ctypes.c_longlong()
    check(lib().tfc_prof_collect(kclass, ctypes.byref(ms), ctypes.byref(fl), ctypes.byref(n)), "tfc_prof_collect")
    return ms.value, fl.value, n.value
