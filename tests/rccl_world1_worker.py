"""Worker of tests/test_gpu_20_ddp.py::test_two_stream_step_under_a_real_one_rank_rccl_group. argv: out.pt mode, mode = "plain" (no process group) or
"rccl" (a ONE-rank group of backend "nccl" = RCCL, TFC_FORCE_COLLECTIVES=1: every bucket all-reduce, broadcast and loss average of the product path is
issued through ProcessGroupNCCL -- identities in value, real in stream ordering). Two bf16 PATCH-16 steps on two streams; saves the state."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tfc_gan_amd as T  # noqa: E402
from oracle import tfcgan_oracle as O  # noqa: E402  (seeded inputs / portable weights only)
from tfc_gan_amd import nets, parallel  # noqa: E402


def run(out_path, mode):
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    if mode == "rccl":
        os.environ["TFC_FORCE_COLLECTIVES"] = "1"
        try:
            dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{os.environ['TFC_TEST_PORT']}", rank=0, world_size=1)
            probe = torch.ones(4, device=dev)
            dist.all_reduce(probe)                                   # communicator creation happens here
            torch.cuda.synchronize()
        except Exception as e:                                       # no usable RCCL on this box: an environment problem, not a result (exit code 77 = skip)
            print("RCCL one-rank group unavailable:", repr(e), flush=True)
            sys.exit(77)
        assert parallel.collectives_active() and dist.get_backend() == "nccl"
    T.set_compute_dtype(torch.bfloat16)
    T.set_wgrad_stream(True)
    assert nets.side_stream_on()
    G = O.init_weights_portable(T.GeneratorUNet((3, 256, 256)), seed=61).to(dev)
    D = O.init_weights_portable(T.Discriminator1((3, 256, 256)), seed=62).to(dev)
    ts = T.TrainStep(G, D, compute_dtype=torch.bfloat16, bucket_bytes=16 << 20)
    issued = {"n": 0}
    if mode == "rccl":
        real = dist.all_reduce

        def counting(*a, **k):
            issued["n"] += 1
            return real(*a, **k)
        parallel.dist.all_reduce = counting
    A, B = O.synthetic_pairs(2, seed=63)
    A, B = A.to(dev), B.to(dev)
    for _ in range(2):
        out = ts.step(A, B)
    torch.cuda.synchronize()
    sn = torch.cat([b.flatten() for b in ts.dbufs.values()])
    torch.save({"g": ts.gflat.data.cpu(), "d": ts.dflat.data.cpu(), "gm": ts.gm.cpu(), "dm": ts.dm.cpu(), "sn": sn.cpu(),
                "loss": torch.stack([out["loss_G"].float(), out["loss_D"].float()]).cpu(), "collectives": torch.tensor(issued["n"])}, out_path)
    if mode == "rccl":
        dist.destroy_process_group()
    print("worker done", mode, issued["n"], flush=True)


if __name__ == "__main__":
    run(sys.argv[1], sys.argv[2])
