"""Data parallelism for the PATCH-16 step: one process per GPU, RCCL (torch.distributed backend "nccl") over xGMI.

The reference uses single-process nn.DataParallel (TFCGAN_multigpu_patchFFT_16P.py:444-445): weights are re-broadcast on
every forward and losses are computed on GPU 0 over the gathered batch.  Here every rank keeps resident weights and Adam
state, processes its own shard of the batch (every op of the path is per-sample and every loss is a batch mean), and the
only exchange is a sum all-reduce of the flat gradient buffers, cut into buckets that follow the order in which the
backward pass finalises gradients so that each bucket's all-reduce overlaps the rest of the backward.

Backend-agnostic (tested with gloo on CPU tensors, world size 2).
"""
import os

import torch
import torch.distributed as dist


def dist_ready():
    return dist.is_available() and dist.is_initialized()


def world_size():
    return dist.get_world_size() if dist_ready() else 1


def rank():
    return dist.get_rank() if dist_ready() else 0


def collectives_active():
    """issue the gradient / loss / weight collectives? Yes with more than one rank -- and, for the one-GPU rehearsal of the product default (RCCL +
    the side stream), in a ONE-rank group under TFC_FORCE_COLLECTIVES=1: the all-reduces are then identities, but they travel through
    ProcessGroupNCCL's stream hand-over exactly as with eight ranks (tests/test_gpu_20_ddp.py: bit-equal to the run without a group)."""
    return dist_ready() and (dist.get_world_size() > 1 or os.environ.get("TFC_FORCE_COLLECTIVES", "0") not in ("", "0"))


# ---- measurement: how long a stream waits for gradient buckets (bench.py: exposed_allreduce_ms) -------------------------------------------
_EXPOSED = {"on": False, "pairs": []}


def exposed_wait_begin():
    """from now on every BucketReducer.finish() brackets its waits with an event pair on the stream it runs on"""
    _EXPOSED["on"], _EXPOSED["pairs"] = True, []


def exposed_wait_collect():
    """total milliseconds between the event pairs recorded since exposed_wait_begin() (synchronises); stops the recording"""
    _EXPOSED["on"] = False
    pairs, _EXPOSED["pairs"] = _EXPOSED["pairs"], []
    if not pairs:
        return 0.0
    torch.cuda.synchronize()
    return float(sum(a.elapsed_time(b) for a, b in pairs))


class FlatParams:
    """Parameters of one network flattened into ONE fp32 buffer in gradient-completion order, plus a same-shaped gradient
    buffer.  `views[name]` / `grad_views[name]` are torch-layout windows into them."""

    def __init__(self, named_tensors, order, device=None):
        self.order = [k for k in order if k in named_tensors]
        missing = set(named_tensors) - set(self.order)
        assert not missing, f"parameters without a bucket position: {missing}"
        self.shapes = {k: tuple(named_tensors[k].shape) for k in self.order}
        self.offsets, off = {}, 0
        for k in self.order:
            self.offsets[k] = off
            off += (named_tensors[k].numel() + 3) // 4 * 4            # keep every window 16-byte aligned
        self.numel = off
        device = device or named_tensors[self.order[0]].device
        self.data = torch.zeros(off, dtype=torch.float32, device=device)
        self.grad = torch.zeros(off, dtype=torch.float32, device=device)
        self.views = {k: self._win(self.data, k) for k in self.order}
        self.grad_views = {k: self._win(self.grad, k) for k in self.order}
        for k in self.order:
            self.views[k].copy_(named_tensors[k].detach().to(torch.float32))

    def _win(self, buf, k):
        n = 1
        for s in self.shapes[k]:
            n *= s
        return buf[self.offsets[k]:self.offsets[k] + n].view(self.shapes[k])

    def span(self, k):
        n = self.views[k].numel()
        return self.offsets[k], self.offsets[k] + (n + 3) // 4 * 4


class BucketReducer:
    """Sum-all-reduce of a FlatParams.grad in buckets of ~bucket_bytes, launched as soon as every gradient of the bucket is
    final.  xGMI is point-to-point (7 links x ~153 GB/s per GPU), so a ring is bound by one link: buckets are sized large
    (default 32 MiB) to stay bandwidth- rather than latency-bound, and there are few of them (the generator needs 4)."""

    def __init__(self, flat: FlatParams, bucket_bytes=32 << 20, group=None):
        self.flat, self.group = flat, group
        self.buckets = []                                        # (start, end, set(names))
        start, names = 0, []
        for k in flat.order:
            s, e = flat.span(k)
            names.append(k)
            if (e - start) * 4 >= bucket_bytes:
                self.buckets.append((start, e, set(names)))
                start, names = e, []
        if names:
            self.buckets.append((start, flat.numel, set(names)))
        self.bucket_of = {k: i for i, (_, _, ns) in enumerate(self.buckets) for k in ns}
        self.reset()

    def reset(self):
        self.pending = [set(ns) for _, _, ns in self.buckets]
        self.works = []

    def ready(self, name):
        """gradient `name` is final (call in backward order)."""
        if not collectives_active():
            return
        i = self.bucket_of[name]
        self.pending[i].discard(name)
        if not self.pending[i]:
            s, e, _ = self.buckets[i]
            self.works.append(dist.all_reduce(self.flat.grad[s:e], op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def finish(self):
        """wait for every bucket (flushing buckets whose hooks never fired); returns the factor that turns the summed
        gradient into the mean (folded into the Adam kernel, so no extra pass over the gradients)."""
        ws = world_size()
        if collectives_active():
            for i, p in enumerate(self.pending):
                if p:
                    s, e, _ = self.buckets[i]
                    self.works.append(dist.all_reduce(self.flat.grad[s:e], op=dist.ReduceOp.SUM, group=self.group, async_op=True))
                    p.clear()
            timed = _EXPOSED["on"] and self.flat.grad.is_cuda
            if timed:
                ev0 = torch.cuda.Event(enable_timing=True)
                ev0.record()
            for w in self.works:
                w.wait()
            if timed:
                ev1 = torch.cuda.Event(enable_timing=True)
                ev1.record()
                _EXPOSED["pairs"].append((ev0, ev1))
        self.reset()
        return 1.0 / ws


def all_reduce_mean(t, group=None):
    """in-place mean over the ranks of a small tensor (the logged losses: one collective of ~10 floats per step)"""
    ws = world_size()
    if collectives_active():
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
        t /= ws
    return t


def broadcast_flat(flat: FlatParams, src=0, group=None):
    if collectives_active():
        dist.broadcast(flat.data, src=src, group=group)


def broadcast_tensors(tensors, src=0, group=None):
    if collectives_active():
        for t in tensors:
            dist.broadcast(t, src=src, group=group)


def shared_neg_idx(step, seed=1234):
    """The 16 negative-patch indices of one step (reference: np.random.randint(16) per patch, unseeded, :565-580):
    a deterministic function of (seed, step), hence identical on every rank without communication."""
    import numpy as np
    return [int(v) for v in np.random.default_rng([seed, step]).integers(16, size=16)]


def shard_slice(global_batch, r=None, ws=None):
    r = rank() if r is None else r
    ws = world_size() if ws is None else ws
    assert global_batch % ws == 0, "global batch must divide evenly (drop_last=True in the reference, :490)"
    per = global_batch // ws
    return slice(r * per, (r + 1) * per)
