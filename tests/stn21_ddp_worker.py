"""Worker of tests/test_gpu_33_stn21.py::test_stn21_two_ranks_match_one_rank (under torch.distributed.run: 2 ranks sharing cuda:0, gloo): one STN21 step
on this rank's shard of a global batch of 2, fp32 parity mode, no LPIPS; rank 0 saves the updated flat buffers and the losses."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tfc_gan_amd as T  # noqa: E402
from oracle import tfcgan_oracle as O  # noqa: E402  (seeded inputs / portable weights only)
from tfc_gan_amd import parallel, stn21  # noqa: E402


def run(out_path, global_batch=2):
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    rccl1 = world == 1 and os.environ.get("TFC_TEST_RCCL1", "0") == "1"      # one-rank "nccl" group: the collectives are issued through RCCL, moving nothing
    if world > 1:
        dist.init_process_group("gloo")
    elif rccl1:
        os.environ["TFC_FORCE_COLLECTIVES"] = "1"
        try:
            dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{os.environ['TFC_TEST_PORT']}", rank=0, world_size=1)
            probe = torch.ones(4, device=dev)
            dist.all_reduce(probe)
            torch.cuda.synchronize()
        except Exception as e:                                       # environment problem, not a result: exit code 77 = skip
            print("RCCL one-rank group unavailable:", repr(e), flush=True)
            sys.exit(77)
        assert parallel.collectives_active()
    T.set_compute_dtype(torch.float32)
    torch.manual_seed(1)
    st = stn21.STN21Step((3, 256, 256), lpips=None, device=dev, bucket_bytes=64 << 20)
    for i, m in enumerate((st.G1, st.G2, st.D1, st.D2, st.net)):
        O.init_weights_portable(m, seed=201 + i)
    with torch.no_grad():
        st.net.fc_loc[6].weight.mul_(4.0)
    st._bump()
    st.G1.eval(); st.G2.eval(); st.net.eval()
    A, B = O.synthetic_pairs(global_batch, seed=77)
    sl = parallel.shard_slice(global_batch)
    out = st.step(A[sl].to(dev), B[sl].to(dev))
    torch.cuda.synchronize()
    if parallel.rank() == 0:
        torch.save({"g": st.gflat.data.cpu(), "d": st.dflat.data.cpu(), "gg": st.gflat.grad.cpu() / world, "dg": st.dflat.grad.cpu() / world,
                    "losses": torch.stack([out[k].float().reshape(()) for k in ("loss_G", "loss_GAN", "recon_loss", "morph_loss", "loss_D")]).cpu()}, out_path)
    if world > 1:
        dist.barrier()
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    run(sys.argv[1])
