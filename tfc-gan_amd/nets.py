"""Forward / backward orchestration of the two networks of the PATCH-16 path on the HIP kernels.

`GeneratorCore` mirrors GeneratorUNet (reference TFCGAN_multigpu_patchFFT_16P.py:136-174), `DiscriminatorCore` mirrors
Discriminator1 (:182-211).  Both work on raw NHWC buffers and hand-written backward chains (no autograd inside); the
nn.Module mirrors in models.py wrap them in torch.autograd.Function, and engine.py calls them directly.

Skip connections never copy: each UNetUp output and its skip tensor are two channel windows of ONE concat buffer,
the down path writes its pooled output straight into the skip window, and the transposed convolution reads the whole
buffer (reference torch.cat, :133).  Gradients mirror that: the convT dgrad writes the whole concat gradient, the
down path accumulates into its window.
"""
import os

import torch

from . import ops
from .ops import (DT_BF16, OP_CONV, OP_CONVT, OP_PADCONV, OP_UPCONV, View, new_act)

# Weight gradients on a second HIP stream. The weight gradient of layer i needs only the gradient of its output, which the main chain (input gradient ->
# activation backward -> next layer) has finished with, so it runs beside that chain and fills the tails of its persistent launches (same-box A/B:
# 10.22 -> 10.05 ms per step). Everything that touches a layer's parameter gradients (wgrad, spectral-norm backward, bias column sum, the all-reduce
# hook) stays together on the side stream, in program order; backward() joins before it returns. Results are bit-identical either way (no atomics
# cross the streams). TFC_WGRAD_STREAM=0 or set_wgrad_stream(False) serialises everything on the caller's stream (bench.py does that in its
# instrumented steps, so that a launch's duration is kernel time).
_SIDE = {"on": os.environ.get("TFC_WGRAD_STREAM", "1") not in ("", "0"), "streams": {}}


def set_wgrad_stream(on):
    """weight gradients on a second stream (default) or in line on the caller's stream; returns the previous setting"""
    prev, _SIDE["on"] = _SIDE["on"], bool(on)
    return prev


def _side_active():
    """side stream in use? Not under a gloo process group: its collectives on device tensors block the host until the stream they were issued on has
    drained, which serialises the two streams the hard way (one-card rehearsal, 2 ranks: 53 ms per step on one stream, 180-500 ms on two). RCCL's are
    stream-ordered and return at once."""
    if not _SIDE["on"]:
        return False
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_backend() == "gloo":
        return False
    return True


def _side_stream(dev):
    st = _SIDE["streams"].get(dev)
    if st is None:
        st = _SIDE["streams"][dev] = torch.cuda.Stream(dev)
    return st


def _on_side(dev, fn, *tensors):
    """run fn() on the side stream behind everything queued on the current stream so far; `tensors` are temporaries fn reads that the caller drops
    before the join (the caching allocator must not hand them out again until the side stream is done with them)"""
    if not _side_active():
        return fn()
    cur, st = torch.cuda.current_stream(dev), _side_stream(dev)
    st.wait_stream(cur)
    with torch.cuda.stream(st):
        out = fn()
    for t in tensors:
        (t.t if isinstance(t, View) else t).record_stream(st)
    return out


def _join_side(dev):
    if _side_active():
        torch.cuda.current_stream(dev).wait_stream(_side_stream(dev))


# the engine's names for the same three
def side_stream_on():
    return _side_active()


on_side, join_side = _on_side, _join_side


def side_mark(dev):
    """an event behind everything queued on the side stream so far (None when the side stream is not in use); wait_mark() makes the caller's stream wait
    for it -- a partial join: what was queued on the side stream AFTER the mark keeps running beside the caller"""
    if not _side_active():
        return None
    ev = torch.cuda.Event()
    ev.record(_side_stream(dev))
    return ev


def wait_mark(dev, ev):
    if ev is not None:
        torch.cuda.current_stream(dev).wait_event(ev)

# (name, Cin, Cout, normalize, dropout)  -- reference :140-145
G_DOWN = [("down1", 3, 64, False, 0.0), ("down2", 64, 128, True, 0.0), ("down3", 128, 256, True, 0.5),
          ("down4", 256, 512, True, 0.5), ("down5", 512, 512, False, 0.0), ("down6", 512, 512, True, 0.0)]
# (name, Cin, Cout, dropout, skip = index of the down block concatenated)  -- reference :147-151
G_UP = [("up1", 512, 512, 0.0, 4), ("up2", 1024, 512, 0.5, 3), ("up3", 1024, 256, 0.5, 2), ("up4", 512, 128, 0.0, 1),
        ("up5", 256, 64, 0.0, 0)]
# (index in nn.Sequential, Cin, Cout)  -- reference :194-202
D_BLOCKS = [(0, 6, 64), (3, 64, 128), (6, 128, 256), (9, 256, 512)]


def down_weight_key(name):
    return f"{name}.model.0.weight"


def g_param_names():
    names = [down_weight_key(n) for n, *_ in G_DOWN] + [f"{n}.model.0.weight" for n, *_ in G_UP]
    return names + ["final.2.weight", "final.2.bias"]


def g_backward_order():
    """Parameter names in the order their gradients become final during backward (bucket order for DDP)."""
    return (["final.2.weight", "final.2.bias"] + [f"{n}.model.0.weight" for n, *_ in reversed(G_UP)]
            + [down_weight_key(n) for n, *_ in reversed(G_DOWN)])


def d_param_names():
    out = []
    for i, _, _ in D_BLOCKS:
        out += [f"model.{i}.bias", f"model.{i}.parametrizations.weight.original"]
    return out + ["model.13.weight"]


def d_backward_order():
    out = ["model.13.weight"]
    for i, _, _ in reversed(D_BLOCKS):
        out += [f"model.{i}.parametrizations.weight.original", f"model.{i}.bias"]
    return out


def pooled(h):
    return (h - 1) // 2 + 1


class _Ctx:
    pass


class GeneratorCore:
    def __init__(self, dt=DT_BF16, channels=3):
        self.dt = dt
        self.channels = channels
        self.params = None
        self.packed = {}
        self._ws = None

    # ---- weights ----
    def set_params(self, params):
        """params: dict state_dict-key -> fp32 CUDA tensor in torch layout (may be views into a flat buffer)."""
        self.params = params
        self.packed = {}
        self._plan = None

    def repack(self):
        """re-pack every operand stream (fwd + dgrad) from the current weights: ONE launch over a device-resident plan"""
        if getattr(self, "_plan", None) is None:
            P, jobs, slots = self.params, [], []
            for name, cin, cout, _, _ in G_DOWN:
                jobs.append((OP_CONV, 0, P[down_weight_key(name)], cin, cout)); slots.append((name, "fwd"))
                if name != "down1":
                    jobs.append((OP_CONV, 1, P[down_weight_key(name)], cin, cout)); slots.append((name, "dgrad"))
            for name, cin, cout, _, _ in G_UP:
                for pas, key in ((0, "fwd"), (1, "dgrad")):
                    jobs.append((OP_CONVT, pas, P[f"{name}.model.0.weight"], cin, cout)); slots.append((name, key))
            for pas, key in ((0, "fwd"), (1, "dgrad")):
                jobs.append((OP_UPCONV, pas, P["final.2.weight"], 128, self.channels)); slots.append(("final", key))
            self._plan = ops.PackPlan(self.dt, jobs)
            for (name, key), buf in zip(slots, self._plan.streams):
                self.packed.setdefault(name, {})[key] = buf
        self._plan.run()

    # ---- forward ----
    def forward(self, x, seed=0, train=True, save=True):
        """x: fp32 NCHW [N,3,S,S] (S multiple of 64, >= 128). Returns (fake fp32 NCHW in (-1,1), ctx). down1 runs as ONE kernel (conv + LeakyReLU +
        BlurPool, ops.first_block_fwd): its 266 MB conv output is never written, the backward needs its signs only and gets them as sign words."""
        ops.require_gpu(x)
        if not self.packed:
            self.repack()
        dt, dev = self.dt, x.device
        N, C, S, S2 = x.shape
        assert C == self.channels and S == S2 and S % 64 == 0 and S >= 128, "GeneratorUNet needs square inputs, S % 64 == 0, S >= 128"
        ctx = _Ctx()
        ctx.N, ctx.S, ctx.seed, ctx.train = N, S, seed, train
        x8 = ops.pack_nhwc8(dt, x)
        # concat buffers: cat[k] = (up_k output | skip) at the resolution of the skip
        sizes = [S >> (i + 1) for i in range(6)]                 # pooled sizes of down1..down6: 128,64,32,16,8,4
        cat = {}
        for name, cin, cout, _, skip in G_UP:
            sc = G_DOWN[skip][2]
            cat[name] = new_act(N, sizes[skip], sizes[skip], cout + sc, dt, dev)
        d6 = new_act(N, sizes[5], sizes[5], 512, dt, dev)
        skip_of = {4: "up1", 3: "up2", 2: "up3", 1: "up4", 0: "up5"}
        ctx.x8, ctx.cat, ctx.d6 = x8, cat, d6
        ctx.raw, ctx.stats, ctx.dins = [], [], []
        cur = x8
        for i, (name, cin, cout, normalize, drop) in enumerate(G_DOWN):
            h = cur.H
            if i < 5:
                up = skip_of[i]
                upc = cat[up].C - cout
                dst = cat[up].sub(upc, cout)
            else:
                dst = d6
            first = i == 0 and not normalize and ops.first_block_bwd_supported(dt, cin, cout)
            if first and drop == 0.0 and getattr(self, "debug", None) is None and ops.first_block_fwd_supported():
                ctx.mask1 = torch.empty((N, h - 1, h - 1, 8), dtype=torch.uint8, device=dev) if save else None
                ops.first_block_fwd(dt, cur, cin, cout, self.packed[name]["fwd"], dst, slope=0.2, act_after_rounding=True, sign_mask=ctx.mask1)
                ctx.raw.append(None)
                ctx.stats.append(None)
                ctx.dins.append(cur)
                cur = dst
                continue
            raw = new_act(N, h - 1, h - 1, cout, dt, dev)
            stats = ops.zeros_f32((N, cout, 2), dev) if normalize else None
            if first and save:
                # the first convolution also leaves one sign bit per output value: all its fused backward (weight gradient only) reads of `raw`
                ctx.mask1 = torch.empty((N, h - 1, h - 1, 8), dtype=torch.uint8, device=dev)
                ops.conv_first_fwd(dt, cur, cin, cout, self.packed[name]["fwd"], raw, sign_mask=ctx.mask1)
            else:
                ops.conv_fwd(dt, OP_CONV, cur, cin, cout, self.packed[name]["fwd"], raw, stats=stats)
            ops.act_fwd(dt, raw, dst, stats=stats, slope=0.2, pool=2, drop_p=drop if train else 0.0, seed=seed * 64 + i)
            ctx.raw.append(raw)
            ctx.stats.append(stats)
            ctx.dins.append(cur)
            cur = dst
        ctx.blur, ctx.bstats, ctx.uins = [], [], []
        for j, (name, cin, cout, drop, skip) in enumerate(G_UP):
            h = cur.H
            rawT = new_act(N, 2 * h, 2 * h, cout, dt, dev)
            ops.conv_fwd(dt, OP_CONVT, cur, cin, cout, self.packed[name]["fwd"], rawT)
            blur = new_act(N, 2 * h, 2 * h, cout, dt, dev)
            bstats = ops.zeros_f32((N, cout, 2), dev)
            ops.act_fwd(dt, rawT, blur, stats=None, slope=1.0, pool=1, stats_out=bstats)
            ops.act_fwd(dt, blur, cat[name].sub(0, cout), stats=bstats, slope=0.0, pool=0, drop_p=drop if train else 0.0,
                        seed=seed * 64 + 16 + j)
            ctx.blur.append(blur)
            ctx.bstats.append(bstats)
            ctx.uins.append(cur)
            cur = View(cat[name].t, cat[name].t.shape[3], 0)
        fake = torch.empty((N, self.channels, S, S), dtype=torch.float32, device=dev)
        if dt == DT_BF16 and self.channels <= 4:                 # phases as columns of one 16-wide MFMA tile, filter in registers
            ops.upconv_head_fwd(dt, cur, self.params["final.2.weight"], self.params["final.2.bias"], fake)
        else:
            ops.conv_fwd(dt, OP_UPCONV, cur, 128, self.channels, self.packed["final"]["fwd"], None, bias=self.params["final.2.bias"],
                         out_nchw=fake)
        ctx.u5 = cur
        ctx.fake = fake
        if not save:
            return fake, None
        return fake, ctx

    # ---- backward ----
    def backward(self, ctx, g_fake, grads, hook=None, accumulate=False, need_input_grad=False):
        try:
            return self._backward(ctx, g_fake, grads, hook, accumulate, need_input_grad)
        finally:
            _join_side(g_fake.device)                             # the weight gradients may have run on the side stream

    def _backward(self, ctx, g_fake, grads, hook=None, accumulate=False, need_input_grad=False):
        """g_fake: fp32 NCHW gradient of the loss wrt fake. grads: dict key -> fp32 tensor (torch layout) that receives the
        parameter gradients (overwritten, or accumulated when `accumulate`). hook(key) fires when a gradient is final.
        need_input_grad: also return d loss / d x as fp32 NCHW (STN21: fake_A2 = generator2(warped_B), STN:629 -- the warp and the localiser
        train through the generator's input); the PATCH-16 step never asks for it."""
        dt, N = self.dt, ctx.N
        dev = g_fake.device
        ch = self.channels
        gb = grads["final.2.bias"]
        if not accumulate:
            gb.zero_()
        dyf = ops.tanh_bwd_pack(dt, g_fake.contiguous().float(), ctx.fake, dbias=gb)
        def _head_wgrad():
            self._ws = ops.conv_wgrad(dt, OP_UPCONV, ctx.u5, dyf, 128, ch, grads["final.2.weight"], accumulate, self._ws)
            if hook:
                hook("final.2.weight")
                hook("final.2.bias")
        _on_side(dev, _head_wgrad, dyf)
        g_cat = new_act(N, ctx.u5.H, ctx.u5.W, ctx.u5.pitch, dt, dev)
        if dt == DT_BF16 and ch <= 8 and ctx.u5.pitch == 128:
            ops.upconv_head_dgrad(dt, dyf, N, ctx.u5.H, ctx.u5.W, self.params["final.2.weight"], g_cat)    # weights-stationary head kernel
        else:
            ops.conv_dgrad(dt, OP_UPCONV, dyf, N, ctx.u5.H, ctx.u5.W, 128, ch, self.packed["final"]["dgrad"], g_cat)
        dbg = getattr(self, "debug", None)
        if dbg is not None:
            dbg["dyf"], dbg["g_u5"] = dyf, ops.View(g_cat.t.clone(), g_cat.C)   # clone: the skip window is accumulated into later
        g_skip = [None] * 6                                       # gradient window of d1..d5 (views), g_d6 separately
        for j in range(4, -1, -1):
            name, cin, cout, drop, skip = G_UP[j]
            blur, bstats, uin = ctx.blur[j], ctx.bstats[j], ctx.uins[j]
            H = blur.H
            g_out = g_cat.sub(0, cout)
            g_skip[skip] = g_cat.sub(cout, g_cat.pitch - cout)
            rstats = ops.zeros_f32((N, cout, 2), dev)
            dp = drop if ctx.train else 0.0
            sd = ctx.seed * 64 + 16 + j
            ops.act_bwd(dt, 1, g_out, blur, N, H, H, cout, None, stats=bstats, slope=0.0, pool=0, drop_p=dp, seed=sd, rstats=rstats)
            d_blur = new_act(N, H, H, cout, dt, dev)
            ops.act_bwd(dt, 2, g_out, blur, N, H, H, cout, d_blur, stats=bstats, slope=0.0, pool=0, drop_p=dp, seed=sd, rstats=rstats)
            d_rawT = new_act(N, H, H, cout, dt, dev)
            ops.act_bwd(dt, 0, d_blur, None, N, H, H, cout, d_rawT, stats=None, slope=1.0, pool=1)
            key = f"{name}.model.0.weight"
            if dbg is not None:
                dbg[f"{name}.g_out"] = ops.View(g_cat.t.clone(), g_cat.C)       # gradient of the whole concat buffer (up window | skip window)
                dbg[f"{name}.d_blur"], dbg[f"{name}.d_rawT"] = d_blur, d_rawT
            def _up_wgrad(uin=uin, d_rawT=d_rawT, cin=cin, cout=cout, key=key):
                self._ws = ops.conv_wgrad(dt, OP_CONVT, uin, d_rawT, cin, cout, grads[key], accumulate, self._ws)
                if hook:
                    hook(key)
            _on_side(dev, _up_wgrad, d_rawT)
            g_in = new_act(N, uin.H, uin.W, uin.pitch, dt, dev)
            ops.conv_dgrad(dt, OP_CONVT, d_rawT, N, uin.H, uin.W, cin, cout, self.packed[name]["dgrad"], g_in)
            if dbg is not None:
                dbg[f"{name}.g_in"] = ops.View(g_in.t.clone(), g_in.C)          # before the down path accumulates into its skip window
            g_cat = g_in                                          # gradient of the next concat buffer (or of d6 when j == 0)
        g_cur = g_cat                                             # = gradient of d6
        for i in range(5, -1, -1):
            name, cin, cout, normalize, drop = G_DOWN[i]
            raw, stats, din = ctx.raw[i], ctx.stats[i], ctx.dins[i]
            Hc = din.H - 1
            dp = drop if ctx.train else 0.0
            sd = ctx.seed * 64 + i
            key = down_weight_key(name)
            if (i == 0 and not need_input_grad and not normalize and dp == 0.0 and dbg is None and ops.first_block_bwd_supported(dt, cin, cout)
                    and pooled(Hc) >= 2):
                # nothing but this weight gradient needs the 266 MB gradient of the first convolution's output: it is never written (igemm.hip:
                # tfc_wgrad_c8_fused_kernel = tfc_act_bwd(mode 0, pool 2) + tfc_conv_wgrad in one kernel, same d_raw bits)
                def _first_wgrad(din=din, raw=raw, g_cur=g_cur, cin=cin, cout=cout, key=key):
                    self._ws = ops.first_block_bwd_wgrad(dt, din, raw, g_cur, cin, cout, grads[key], slope=0.2, accumulate=accumulate, ws=self._ws,
                                                         sign_mask=getattr(ctx, "mask1", None))
                    if hook:
                        hook(key)
                _on_side(dev, _first_wgrad, g_cur)
                continue
            d_raw = new_act(N, Hc, Hc, cout, dt, dev)
            if raw is None:                                       # down1 of a fused forward: LeakyReLU' from the sign words
                ops.act_bwd_signs(dt, g_cur, ctx.mask1, N, Hc, Hc, cout, d_raw, slope=0.2)
            elif normalize:
                rstats = ops.zeros_f32((N, cout, 2), dev)
                ops.act_bwd(dt, 1, g_cur, raw, N, Hc, Hc, cout, None, stats=stats, slope=0.2, pool=2, drop_p=dp, seed=sd, rstats=rstats)
                ops.act_bwd(dt, 2, g_cur, raw, N, Hc, Hc, cout, d_raw, stats=stats, slope=0.2, pool=2, drop_p=dp, seed=sd, rstats=rstats)
            else:
                ops.act_bwd(dt, 0, g_cur, raw, N, Hc, Hc, cout, d_raw, stats=None, slope=0.2, pool=2, drop_p=dp, seed=sd)
            key = down_weight_key(name)
            if dbg is not None:
                dbg[f"{name}.g_out"] = ops.View(g_cur.t[..., g_cur.coff:g_cur.coff + g_cur.C].clone(), g_cur.C)   # incl. the accumulated skip part
                dbg[f"{name}.d_raw"] = d_raw
            def _down_wgrad(din=din, d_raw=d_raw, cin=cin, cout=cout, key=key):
                self._ws = ops.conv_wgrad(dt, OP_CONV, din, d_raw, cin, cout, grads[key], accumulate, self._ws)
                if hook:
                    hook(key)
            _on_side(dev, _down_wgrad, d_raw)
            if i > 0:
                tgt = g_skip[i - 1]                               # window of d_{i} gradient already holding the skip-path part
                ops.conv_dgrad(dt, OP_CONV, d_raw, N, din.H, din.W, cin, cout, self.packed[name]["dgrad"], tgt, accumulate=True)
                g_cur = tgt
            elif need_input_grad:
                w1 = self.params[key]
                if dt == DT_BF16 and cout == 64 and self.channels <= 4:
                    return ops.conv_dgrad_image(dt, d_raw, N, din.H, din.W, cin, w1, None, self.channels)   # rows-packed kernel, fp32 NCHW out
                gx8 = new_act(N, din.H, din.W, din.pitch, dt, dev)
                ops.conv_dgrad(dt, OP_CONV, d_raw, N, din.H, din.W, cin, cout, ops.pack_weight(dt, OP_CONV, 1, w1.contiguous(), cin, cout), gx8)
                return ops.unpack_nchw(dt, gx8, self.channels, c0=0)
        return None


class DiscriminatorCore:
    def __init__(self, dt=DT_BF16, channels=3):
        self.dt = dt
        self.channels = channels
        self.params = None
        self.buffers = None
        self.head_packed = {}

    def set_params(self, params, buffers):
        """params: dict key -> fp32 tensor ('model.{0,3,6,9}.bias', '...parametrizations.weight.original', 'model.13.weight');
        buffers: dict 'model.{i}.parametrizations.weight.0._u' / '._v' -> fp32 tensors (updated in place)."""
        self.params, self.buffers = params, buffers
        self.head_packed = {}
        self._plan = None

    def repack(self):
        """operand streams of the UN-normalised weights, once per weight update (one launch); 1/sigma is applied in the GEMM epilogue"""
        if getattr(self, "_plan", None) is None:
            jobs, slots = [], []
            w = self.params["model.13.weight"]
            jobs.append((OP_PADCONV, 0, w, 512, 1)); slots.append("fwd")
            jobs.append((OP_PADCONV, 1, w, 512, 1)); slots.append("dgrad")
            for i, cin, cout in D_BLOCKS:
                W = self.params[f"model.{i}.parametrizations.weight.original"]
                jobs.append((OP_CONV, 0, W, cin, cout)); slots.append(f"f{i}")
                jobs.append((OP_CONV, 1, W, cin, cout)); slots.append(f"d{i}")
            self._plan = ops.PackPlan(self.dt, jobs)
            for key, buf in zip(slots, self._plan.streams):
                self.head_packed[key] = buf
        self._plan.run()

    def forward(self, img_a, img_b, power_iter=True, save=True, after_sn=None):
        """img_a, img_b: fp32 NCHW [N,3,S,S]. Returns (logits View [N,S/16,S/16,pitch 8] channel 0, ctx).
        = sn_snapshot() + chain(). after_sn: called once the power iteration of this call is queued -- everything after it reads only this call's
        snapshots of u, v, sigma, so a second forward may start from there on another stream (forward_pair)."""
        ops.require_gpu(img_a, img_b)
        snapshot = self.sn_snapshot(img_a.device, power_iter, save)
        if after_sn is not None:
            after_sn()
        return self.chain(img_a, img_b, snapshot, save)

    def sn_snapshot(self, dev, power_iter=True, save=True):
        """one power iteration of the four spectrally normalised blocks (u, v updated in place as the reference's training-mode forward does) and this
        call's snapshots (u, v, [sigma, 1/sigma]) -- all a forward / backward pass of this call reads afterwards. Depends on the weights only, so the
        generator step takes both of its calls' snapshots up front, in order, and runs the second call's chain beside the generator."""
        if not self.head_packed:
            self.repack()
        # spectral norm of all four blocks in one batched power iteration; per-call snapshots of u, v, sigma for the backward
        Ws = [self.params[f"model.{i}.parametrizations.weight.original"] for i, _, _ in D_BLOCKS]
        us = [self.buffers[f"model.{i}.parametrizations.weight.0._u"] for i, _, _ in D_BLOCKS]
        vs = [self.buffers[f"model.{i}.parametrizations.weight.0._v"] for i, _, _ in D_BLOCKS]
        nu, nv = [u.numel() for u in us], [v.numel() for v in vs]
        snap = torch.empty(sum(nu) + sum(nv) + 2 * len(Ws), dtype=torch.float32, device=dev)
        offs, o = [], 0
        for a in nu + nv + [2] * len(Ws):
            offs.append(o)
            o += a
        L = len(Ws)
        usn = [snap[offs[k]:offs[k] + nu[k]] for k in range(L)]
        vsn = [snap[offs[L + k]:offs[L + k] + nv[k]] for k in range(L)]
        sig = [snap[offs[2 * L + k]:offs[2 * L + k] + 2] for k in range(L)]
        self._sn_ws = ops.spectral_norm_step_batched(Ws, us, vs, sig, power_iter=power_iter, u_snaps=usn if save else None,
                                                     v_snaps=vsn if save else None, ws=getattr(self, "_sn_ws", None))
        return usn, vsn, sig

    def chain(self, img_a, img_b, snapshot, save=True):
        """the convolution chain of one call, given its sn_snapshot(). Block 1 runs as one kernel (ops.first_block_fwd) that keeps sign words
        instead of its 266 MB conv output (all any backward reads of it)."""
        usn, vsn, sig = snapshot
        dt, dev = self.dt, img_a.device
        N, C, S, _ = img_a.shape
        ctx = _Ctx()
        ctx.N, ctx.S = N, S
        x8 = ops.pack_nhwc8(dt, img_a, img_b)
        ctx.ins, ctx.raw, ctx.sn = [], [], []
        cur = x8
        for bi, (i, cin, cout) in enumerate(D_BLOCKS):
            sigma2 = sig[bi]
            h = cur.H
            first = bi == 0 and ops.first_block_bwd_supported(dt, cin, cout)
            if first and getattr(self, "debug", None) is None and ops.first_block_fwd_supported():
                ctx.mask1 = torch.empty((N, h - 1, h - 1, 8), dtype=torch.uint8, device=dev) if save else None
                out = new_act(N, pooled(h - 1), pooled(h - 1), cout, dt, dev)
                ops.first_block_fwd(dt, cur, cin, cout, self.head_packed[f"f{i}"], out, bias=self.params[f"model.{i}.bias"], oscale=sigma2[1:],
                                    slope=0.2, act_after_rounding=False, sign_mask=ctx.mask1)
                ctx.ins.append(cur)
                ctx.raw.append(None)
                ctx.sn.append((usn[bi], vsn[bi], sigma2) if save else None)
                cur = out
                continue
            raw = new_act(N, h - 1, h - 1, cout, dt, dev)
            # SN-conv + bias + LeakyReLU(0.2) in one kernel (P16:188-190): `raw` holds the ACTIVATED tensor; the backward's slope test
            # (y > 0) is the same on y = LeakyReLU(z) as on z, so act_bwd below keeps slope = 0.2 on this tensor
            if first and save:
                ctx.mask1 = torch.empty((N, h - 1, h - 1, 8), dtype=torch.uint8, device=dev)       # sign bits for the fused backward of block 1
                ops.conv_first_fwd(dt, cur, cin, cout, self.head_packed[f"f{i}"], raw, bias=self.params[f"model.{i}.bias"], oscale=sigma2[1:],
                                   flags=ops.EP_LEAKY, sign_mask=ctx.mask1)
            else:
                ops.conv_fwd(dt, OP_CONV, cur, cin, cout, self.head_packed[f"f{i}"], raw, bias=self.params[f"model.{i}.bias"], oscale=sigma2[1:],
                             flags=ops.EP_LEAKY)
            out = new_act(N, pooled(h - 1), pooled(h - 1), cout, dt, dev)
            ops.act_fwd(dt, raw, out, stats=None, slope=1.0, pool=2)
            ctx.ins.append(cur)
            ctx.raw.append(raw)
            ctx.sn.append((usn[bi], vsn[bi], sigma2) if save else None)
            cur = out
        logits = new_act(N, cur.H, cur.W, 8, dt, dev)             # only channel 0 is ever written or read (bce: stride 8; modules: [..., 0])
        ops.patchgan_head_fwd(dt, cur, self.params["model.13.weight"], View(logits.t, 1, 0))
        ctx.p4 = cur
        return View(logits.t, 1, 0), (ctx if save else None)

    def forward_pair(self, a1, b1, a2, b2, power_iter=True, save=True):
        """forward(a1, b1) then forward(a2, b2), same results as the two calls in that order (the second power iteration follows the first). With the
        side stream on, the second call's convolution chain runs beside the first's: two independent chains of MFMA-bound GEMMs and HBM-bound
        blur-pools that fill each other's gaps."""
        if not _side_active() or os.environ.get("TFC_NO_FWD_PAIR", "0") not in ("", "0"):    # (A/B knob)
            return self.forward(a1, b1, power_iter, save), self.forward(a2, b2, power_iter, save)
        dev = a1.device
        if not self.head_packed:
            self.repack()
        box = {}

        def second():
            box["r"] = _on_side(dev, lambda: self.forward(a2, b2, power_iter, save))
        first = self.forward(a1, b1, power_iter, save, after_sn=second)
        _join_side(dev)
        return first, box["r"]

    def backward(self, ctx, g_logits, grads=None, need_input_grad=True, accumulate=False, hook=None, ws=None):
        try:
            return self._backward(ctx, g_logits, grads, need_input_grad, accumulate, hook, ws)
        finally:
            _join_side(g_logits.t.device)                         # the weight gradients may have run on the side stream

    def _backward(self, ctx, g_logits, grads=None, need_input_grad=True, accumulate=False, hook=None, ws=None):
        """g_logits: View [N,h,w,8] (channel 0 = gradient, channels 1..7 zero). grads: dict key -> fp32 tensor or None
        (skip all weight gradients: generator step). Returns fp32 NCHW gradient of img_a (first argument) or None."""
        dt, N = self.dt, ctx.N
        dev = g_logits.t.device
        wsh = [getattr(self, "_ws", None) if ws is None else ws]  # persistent wgrad scratch (zeroed once, re-zeroed by the kernels)
        gl = View(g_logits.t, 8, 0)
        if grads is not None:
            def _head_wgrad():
                wsh[0] = ops.conv_wgrad(dt, OP_PADCONV, ctx.p4, gl, 512, 1, grads["model.13.weight"], accumulate, wsh[0])
                if hook:
                    hook("model.13.weight")
            _on_side(dev, _head_wgrad)
        g_cur = new_act(N, ctx.p4.H, ctx.p4.W, 512, dt, dev)
        ops.conv_dgrad(dt, OP_PADCONV, gl, N, ctx.p4.H, ctx.p4.W, 512, 1, self.head_packed["dgrad"], g_cur)
        g_in = None
        for bi in range(3, -1, -1):
            i, cin, cout = D_BLOCKS[bi]
            raw, xin = ctx.raw[bi], ctx.ins[bi]
            u, v, sigma2 = ctx.sn[bi]
            Hc = xin.H - 1
            gb_img = ops.zeros_f32((N, cout), dev) if grads is not None else None
            fuse = (bi == 0 and grads is not None and not need_input_grad and getattr(self, "debug", None) is None
                    and ops.first_block_bwd_supported(dt, cin, cout))
            if fuse:
                # discriminator step: block 1's conv-output gradient (266 MB) feeds only its weight / bias gradients -> never written
                def _first_wgrad(i=i, cin=cin, cout=cout, xin=xin, raw=raw, g_cur=g_cur, gb_img=gb_img, u=u, v=v, sigma2=sigma2):
                    W = self.params[f"model.{i}.parametrizations.weight.original"]
                    gsn = torch.empty_like(W)
                    wsh[0] = ops.first_block_bwd_wgrad(dt, xin, raw, g_cur, cin, cout, gsn, slope=0.2, ws=wsh[0], bias_sums=gb_img,
                                                       sign_mask=getattr(ctx, "mask1", None))
                    gbias = grads[f"model.{i}.bias"]
                    if not accumulate:
                        gbias.zero_()
                    ops.colsum(ops.DT_F32, View(gb_img.view(N, 1, 1, cout), cout), gbias)
                    ops.spectral_norm_bwd(gsn, W, u, v, sigma2, grads[f"model.{i}.parametrizations.weight.original"], accumulate)
                    if hook:
                        hook(f"model.{i}.parametrizations.weight.original")
                        hook(f"model.{i}.bias")
                _on_side(dev, _first_wgrad, g_cur, gb_img)
                continue
            d_raw = new_act(N, Hc, Hc, cout, dt, dev)
            if raw is None:                                       # block 1 of a fused forward: LeakyReLU' from the sign words
                ops.act_bwd_signs(dt, g_cur, ctx.mask1, N, Hc, Hc, cout, d_raw, slope=0.2, rstats=gb_img)
            else:
                ops.act_bwd(dt, 0, g_cur, raw, N, Hc, Hc, cout, d_raw, stats=None, slope=0.2, pool=2, rstats=gb_img)   # + per-image bias gradient
            ddbg = getattr(self, "debug", None)
            if ddbg is not None:
                ddbg[f"b{bi}.g_out"], ddbg[f"b{bi}.d_raw"] = g_cur, d_raw
            W = self.params[f"model.{i}.parametrizations.weight.original"]
            if grads is not None:
                def _block_wgrad(i=i, cin=cin, cout=cout, xin=xin, d_raw=d_raw, gb_img=gb_img, W=W, u=u, v=v, sigma2=sigma2):
                    gbias = grads[f"model.{i}.bias"]
                    if not accumulate:
                        gbias.zero_()
                    ops.colsum(ops.DT_F32, View(gb_img.view(N, 1, 1, cout), cout), gbias)
                    gsn = torch.empty_like(W)
                    wsh[0] = ops.conv_wgrad(dt, OP_CONV, xin, d_raw, cin, cout, gsn, False, wsh[0])
                    ops.spectral_norm_bwd(gsn, W, u, v, sigma2, grads[f"model.{i}.parametrizations.weight.original"], accumulate)
                    if hook:
                        hook(f"model.{i}.parametrizations.weight.original")
                        hook(f"model.{i}.bias")
                _on_side(dev, _block_wgrad, d_raw, gb_img)
            if bi == 0 and need_input_grad and dt == DT_BF16 and self.channels <= 4 and cout == 64:
                # gradient w.r.t. the generated image only (3 of the 6 input channels): rows-packed 16-wide MFMA kernel, fp32 NCHW out
                self._ws = wsh[0]
                return ops.conv_dgrad_image(dt, d_raw, N, xin.H, xin.W, cin, W, sigma2[1:], self.channels)
            if bi > 0 or need_input_grad:
                g_in = new_act(N, xin.H, xin.W, xin.pitch, dt, dev)
                ops.conv_dgrad(dt, OP_CONV, d_raw, N, xin.H, xin.W, cin, cout, self.head_packed[f"d{i}"], g_in, oscale=sigma2[1:])
                g_cur = g_in
        self._ws = wsh[0]
        if need_input_grad:
            return ops.unpack_nchw(dt, g_cur, self.channels, c0=0)
        return None
