"""The N>1 path on CPU: two gloo ranks, each with its own shard, exchange gradients through FlatParams + BucketReducer
(the code the RCCL path runs); the averaged result must equal the single-process gradient of the batch-mean loss, which is
what the reference's gather-then-loss nn.DataParallel computes (TFCGAN_multigpu_patchFFT_16P.py:444-445)."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import tfcgan_oracle as O


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _loss(G, A, B, neg):
    return O.patch_triplet_loss(G(A), B, neg)


def _worker(rank, world, port, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(4)
    import tfc_gan_amd as T
    from tfc_gan_amd import nets, parallel
    G = O.init_weights_portable(O.GeneratorUNet((3, 256, 256)), seed=70 + rank).eval()     # ranks start DIFFERENT on purpose
    named = {k: v for k, v in G.named_parameters()}
    flat = parallel.FlatParams(named, nets.g_backward_order(), torch.device("cpu"))
    for k, p in named.items():
        p.data = flat.views[k]
    parallel.broadcast_flat(flat, src=0)                                                   # ... and are made identical here
    red = parallel.BucketReducer(flat, bucket_bytes=32 << 20)
    A, B = O.synthetic_pairs(world, seed=5)
    sl = parallel.shard_slice(world)
    neg = parallel.shared_neg_idx(7)
    loss = _loss(G, A[sl], B[sl], neg)
    grads = torch.autograd.grad(loss, [named[k] for k in flat.order])
    for k, g in zip(flat.order, grads):                                                   # backward order = bucket order
        flat.grad_views[k].copy_(g)
        red.ready(k)
    scale = red.finish()
    avg = flat.grad * scale
    if rank == 0:
        torch.save({"avg": avg, "w0": flat.data.clone(), "neg": neg, "nbuckets": len(red.buckets), "order": flat.order}, out_path)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gradient_allreduce_equals_full_batch(tmp_path):
    world = 2
    out = str(tmp_path / "r0.pt")
    mp.spawn(_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    got = torch.load(out, weights_only=False)
    from tfc_gan_amd import nets, parallel
    assert got["neg"] == parallel.shared_neg_idx(7) and got["nbuckets"] >= 2
    G = O.init_weights_portable(O.GeneratorUNet((3, 256, 256)), seed=70).eval()             # rank 0's weights were broadcast
    named = {k: v for k, v in G.named_parameters()}
    A, B = O.synthetic_pairs(world, seed=5)
    loss = _loss(G, A, B, got["neg"])
    grads = torch.autograd.grad(loss, [named[k] for k in got["order"]])
    flat = parallel.FlatParams(named, nets.g_backward_order(), torch.device("cpu"))
    assert torch.equal(flat.data, got["w0"])
    for k, g in zip(got["order"], grads):
        s, _ = flat.span(k)
        ref = g.reshape(-1)
        mine = got["avg"][s:s + ref.numel()]
        assert (mine - ref).abs().max().item() <= 1e-6 + 1e-4 * ref.abs().max().item(), k


def _worker4(rank, world, port, out_path):
    """four ranks, deliberately awkward bucket boundaries (5 MB buckets cut the 29.2 M-parameter buffer into ~20 uneven pieces; one
    hook never fires, so finish() must flush its bucket), plus the logged-loss averaging collective"""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    from tfc_gan_amd import nets, parallel
    G = O.init_weights_portable(O.GeneratorUNet((3, 256, 256)), seed=90).eval()
    named = {k: v for k, v in G.named_parameters()}
    flat = parallel.FlatParams(named, nets.g_backward_order(), torch.device("cpu"))
    red = parallel.BucketReducer(flat, bucket_bytes=5_000_000)
    gen = torch.Generator().manual_seed(1000 + rank)
    flat.grad.copy_(torch.randn(flat.numel, generator=gen))
    mine = flat.grad.clone()
    skipped = flat.order[3]
    for k in flat.order:                                         # backward order; one gradient "forgets" its hook
        if k != skipped:
            red.ready(k)
    scale = red.finish()
    losses = torch.tensor([float(rank), 10.0 * rank, -1.0])
    parallel.all_reduce_mean(losses)
    gathered = [torch.empty_like(mine) for _ in range(world)] if rank == 0 else None
    dist.gather(mine, gathered, dst=0)
    if rank == 0:
        want = torch.stack(gathered).sum(0)
        torch.save({"ok": bool(torch.allclose(flat.grad, want, rtol=1e-5, atol=1e-5)), "scale": scale, "losses": losses,
                    "nbuckets": len(red.buckets), "sizes": [e - s for s, e, _ in red.buckets]}, out_path)
    dist.barrier()
    dist.destroy_process_group()


def test_four_ranks_uneven_buckets_and_loss_mean(tmp_path):
    world = 4
    out = str(tmp_path / "r0.pt")
    mp.spawn(_worker4, args=(world, _free_port(), out), nprocs=world, join=True)
    got = torch.load(out, weights_only=False)
    assert got["ok"] and got["scale"] == 0.25
    assert got["nbuckets"] >= 6 and len(set(got["sizes"])) > 2          # uneven pieces
    assert torch.allclose(got["losses"], torch.tensor([1.5, 15.0, -1.0]))
