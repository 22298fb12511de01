"""The PATCH-16 training step (reference TFCGAN_multigpu_patchFFT_16P.py:545-638) as one engine object.

    G step : fake = G(A); pf = D(fake, A); pr = D(B, A)
             loss_G = 0.5 * BCE(pf - pr.detach(), 0.9) + triplet16(fake, B, neg_idx) + 0.01 * patchFFT(fake, B)   (:554-607)
             backward through D (input gradient only) and G; Adam(G)
    D step : pr = D(B, A); pf = D(fake.detach(), A); loss_D = 0.5 * [BCE(pr - pf, 0.9) + BCE(pf - pr, 0)]          (:623-630)
             backward; Adam(D)

Scope notes: LPIPS (:598) needs VGG weights that cannot be fetched offline; `extra_loss_G` lets a caller
plug it in.  The temperature head (:587-595, zero gradient) is computed when the caller passes `T_B` (and the augmented
negatives `B_tf`): it adds 0.5 * loss_temp_g to the logged loss_G exactly as :607 does.  The FFT term carries no gradient in the reference (tensor -> PIL -> numpy, :300-302) and none here.  bf16 needs no
GradScaler (:518); the reference's wasted D weight-gradients during the G step (:619 zeroes them) are simply not computed.
"""
import math
import os

import torch

from . import nets, ops, parallel
from .losses import global_fft_loss, patch_fft_loss, temperature_triplet_loss
from .ops import DT_BF16


class TrainStep:
    def __init__(self, generator, discriminator, lr=2e-4, b1=0.5, b2=0.999, eps=1e-8, compute_dtype=torch.bfloat16,
                 fft_mode="patch", seed=0, bucket_bytes=16 << 20, lambda_gan=0.5, lambda_fft=0.01, lambda_trip=1.0, d_bucket_bytes=4 << 20):
        dev = next(generator.parameters()).device
        if dev.type != "cuda":
            raise ops._lib.TfcError("TrainStep needs the modules on a CUDA/HIP device (no CPU fallback)")
        self.dev, self.dt = dev, ops.dt_of(compute_dtype)
        self.G_mod, self.D_mod = generator, discriminator
        self.lr, self.b1, self.b2, self.eps = lr, b1, b2, eps
        self.lambda_gan, self.lambda_fft, self.lambda_trip = lambda_gan, lambda_fft, lambda_trip
        self.fft_mode = fft_mode
        self.seed = seed
        self.step_no = 0
        # flat fp32 parameter / gradient / Adam buffers in backward order; module parameters become views of them, so
        # state_dict()/load_state_dict() keep working and stay in the reference's layout.
        self.gflat = parallel.FlatParams(generator.named_core_params(), nets.g_backward_order(), dev)
        self.dflat = parallel.FlatParams(discriminator.named_core_params(), nets.d_backward_order(), dev)
        for mod, flat in ((generator, self.gflat), (discriminator, self.dflat)):
            named = dict(mod.named_parameters())
            for k in flat.order:
                named[k].data = flat.views[k]
                named[k].grad = flat.grad_views[k]
        parallel.broadcast_flat(self.gflat)
        parallel.broadcast_flat(self.dflat)
        self.dbufs = discriminator.named_core_buffers()
        parallel.broadcast_tensors(list(self.dbufs.values()))
        self.gm, self.gv = torch.zeros_like(self.gflat.data), torch.zeros_like(self.gflat.data)
        self.dm, self.dv = torch.zeros_like(self.dflat.data), torch.zeros_like(self.dflat.data)
        self.g_reduce = parallel.BucketReducer(self.gflat, bucket_bytes)
        # the discriminator's 11 MB of gradients: its largest layer (model.9, 8.4 MB) is final first, so a 4 MiB cut lets that part of the exchange
        # run under the rest of the D backward; only the last ~2.6 MB (model.6, .3, .0) are exposed
        self.d_reduce = parallel.BucketReducer(self.dflat, min(bucket_bytes, d_bucket_bytes))
        self.G = nets.GeneratorCore(self.dt, generator.channels)
        self.G.set_params(self.gflat.views)
        self.D = nets.DiscriminatorCore(self.dt, discriminator.channels)
        self.D.set_params(self.dflat.views, self.dbufs)
        self.G.repack()
        self.D.repack()
        self._pversions = self._param_versions()
        self.last = {}

    def _param_versions(self):
        """autograd version counters of the module parameters: they move when the CALLER writes the weights (load_state_dict,
        optimizer of its own, .apply(init)); the raw-pointer Adam kernel of this class does not touch them."""
        return tuple(p._version for m in (self.G_mod, self.D_mod) for p in m.parameters())

    def _invalidate_module_cores(self):
        """the modules' own forward (sample_images P16:391-416, eval) caches operand streams keyed on (data_ptr, _version); the Adam
        kernel changes neither, so bump the modules' weight generation after every update"""
        self.G_mod._weights_gen += 1
        self.D_mod._weights_gen += 1

    def _gl(self, like):
        return ops.new_act(like.N, like.H, like.W, 8, self.dt, self.dev)      # tfc_bce_relativistic writes whole 8-channel pixels (logit gradient, 7 zeros)

    def step(self, real_A, real_B, neg_idx=None, extra_loss_G=None, T_B=None, B_tf=None):
        """real_A, real_B: fp32 NCHW [N,3,256,256] in [-1,1] on the GPU (this rank's shard). Returns a dict of device scalars.
        T_B [N,256,256] + B_tf [N,3,256,256] (augmented real_B) switch the gradient-free temperature term on.
        extra_loss_G(fake, real_B) -> (loss, dfake) runs on the SIDE stream beside the discriminator chain (with the triplet / FFT heads): it may use any
        kernel of the package except tfc_bce_relativistic, whose scalar-reduction slot belongs to the main stream's call (csrc/common.h: TfcRedSlot)."""
        dt = self.dt
        self.step_no += 1
        t = self.step_no
        if neg_idx is None:
            neg_idx = parallel.shared_neg_idx(t, self.seed)
        drop_seed = (self.seed * 7919 + t * 104729 + parallel.rank() * 1299709) & 0x3FFFFFF
        train = self.G_mod.training
        pv = self._param_versions()
        if pv != self._pversions:                                 # weights written from outside since the last step (load_state_dict): re-pack
            self.G.repack()
            self.D.repack()
            self._invalidate_module_cores()
            self._pversions = pv
        ops.arena_begin(self.dev)                                 # one fill for all the small zero-initialised buffers of this step
        # ---------------- generator step ----------------
        def pixel_losses():                                       # everything that needs only the generated image (beside the discriminator chain)
            lt, gt = ops.patch16_triplet(fake, real_B, neg_idx, want_grad=True, gscale=self.lambda_trip)
            lf = patch_fft_loss(fake, real_B) if self.fft_mode == "patch" else global_fft_loss(fake, real_B)
            ex = extra_loss_G(fake, real_B) if extra_loss_G is not None else None     # optional pluggable term (LPIPS, P16:598): (loss, dfake), already weighted
            return lt, gt, lf, ex
        if nets.side_stream_on() and os.environ.get("TFC_NO_GSTEP_OVERLAP", "0") in ("", "0"):
            # Two-stream form of the same program (nets.py: side stream). Both power iterations of this step's two discriminator calls come first, in
            # call order (they read the weights only); the chain of the SECOND call (real pair, no gradient) then runs beside the generator forward,
            # and the triplet / FFT losses of the generated image beside the first call's chain.
            snap_f = self.D.sn_snapshot(self.dev, True, True)
            snap_r = self.D.sn_snapshot(self.dev, True, False)
            pr = nets.on_side(self.dev, lambda: self.D.chain(real_B, real_A, snap_r, save=False), snap_r[2][0])[0]
            pr_ready = nets.side_mark(self.dev)
            fake, gctx = self.G.forward(real_A, seed=drop_seed, train=train)
            loss_trip, g_trip, (loss_fft, loss_amp, loss_pha), extra_pair = nets.on_side(self.dev, pixel_losses)
            pf, dctx_f = self.D.chain(fake, real_A, snap_f, save=True)
            nets.wait_mark(self.dev, pr_ready)                    # the logits of the real pair; the pixel losses (LPIPS: 6 ms) run on, D.backward below joins
        else:
            fake, gctx = self.G.forward(real_A, seed=drop_seed, train=train)
            pf, dctx_f = self.D.forward(fake, real_A, power_iter=True, save=True)
            pr, _ = self.D.forward(real_B, real_A, power_iter=True, save=False)
            loss_trip, g_trip, (loss_fft, loss_amp, loss_pha), extra_pair = pixel_losses()
        g_pf = self._gl(pf)
        loss_gan = ops.bce_relativistic(dt, pf, pr, 0, 0.9, da=ops.View(g_pf.t, 1, 0), gscale=self.lambda_gan)
        g_fake = self.D.backward(dctx_f, g_pf, grads=None, need_input_grad=True)
        ops.axpby(g_fake, g_fake, g_trip, 1.0, 1.0)
        extra = None
        if extra_pair is not None:
            extra, g_extra = extra_pair
            ops.axpby(g_fake, g_fake, g_extra, 1.0, 1.0)
        self.G.backward(gctx, g_fake, self.gflat.grad_views, hook=self.g_reduce.ready)
        def g_update():
            gscale = self.g_reduce.finish()
            ops.adam_step(self.gflat.data, self.gflat.grad, self.gm, self.gv, self.lr, self.b1, self.b2, self.eps, t, gscale)
            self.G.repack()
        if nets.side_stream_on() and os.environ.get("TFC_NO_GUPDATE_OVERLAP", "0") in ("", "0"):
            nets.on_side(self.dev, g_update)                      # HBM-bound Adam + re-pack beside the discriminator step's first chain (joined by D.backward)
        else:
            g_update()
        # ---------------- discriminator step ----------------
        (pr2, dctx_r), (pf2, dctx_f2) = self.D.forward_pair(real_B, real_A, fake, real_A, power_iter=True, save=True)
        g_pr, g_pf2 = self._gl(pr2), self._gl(pf2)
        loss_d = ops.bce_relativistic(dt, pr2, pf2, 1, 0.9, 0.0, da=ops.View(g_pr.t, 1, 0), db=ops.View(g_pf2.t, 1, 0))
        self.D.backward(dctx_r, g_pr, grads=self.dflat.grad_views, need_input_grad=False, accumulate=False)
        self.D.backward(dctx_f2, g_pf2, grads=self.dflat.grad_views, need_input_grad=False, accumulate=True, hook=self.d_reduce.ready)
        dscale = self.d_reduce.finish()
        ops.adam_step(self.dflat.data, self.dflat.grad, self.dm, self.dv, self.lr, self.b1, self.b2, self.eps, t, dscale)
        self.D.repack()
        ops.arena_end(self.dev)
        self._invalidate_module_cores()
        loss_g = self.lambda_gan * loss_gan + self.lambda_trip * loss_trip + self.lambda_fft * loss_fft
        if extra is not None:
            loss_g = loss_g + extra.reshape(loss_g.shape).to(loss_g.dtype)
        loss_temp = None
        if T_B is not None:
            loss_temp = temperature_triplet_loss(fake, T_B, B_tf if B_tf is not None else real_B)
            loss_g = loss_g + 0.5 * loss_temp
        self.last = {"loss_G": loss_g.reshape(()), "loss_GAN_g": loss_gan.reshape(()), "loss_triplet_patch": loss_trip.reshape(()),
                     "loss_FFT": loss_fft.reshape(()), "loss_Amp": loss_amp, "loss_Pha": loss_pha, "loss_D": loss_d.reshape(()),
                     "fake_B": fake}
        if loss_temp is not None:
            self.last["loss_temp_g"] = loss_temp
        if extra is not None:
            self.last["loss_extra_g"] = extra.reshape(())
        if parallel.collectives_active():                          # every logged loss is a batch mean: mean over ranks = the global-batch value
            keys = [k for k in self.last if k != "fake_B"]
            packed = torch.stack([self.last[k].reshape(()).float() for k in keys])
            parallel.all_reduce_mean(packed)
            for i, k in enumerate(keys):
                self.last[k] = packed[i]
        return self.last

    # algorithmic work of one step per image: conv / convT MACs x 2
    G_FWD_GFLOP = 23.574
    D_FWD_GFLOP = 13.224
    STEP_GFLOP = 3 * 23.574 + 4 * 13.224 + 13.224 + 2 * 2 * 13.224
