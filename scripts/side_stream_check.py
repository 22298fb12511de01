"""Gradient agreement of one training step with the weight-gradient side stream on / off, against the run-to-run spread of each setting.
usage: python scripts/side_stream_check.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

import tfc_gan_amd as T  # noqa: E402

dev = torch.device("cuda", 0)


def run(on, cdt, batch):
    T.set_compute_dtype(cdt)
    T.set_wgrad_stream(on)
    torch.manual_seed(7)
    G = T.GeneratorUNet((3, 256, 256)).to(dev)
    D = T.Discriminator1((3, 256, 256)).to(dev)
    G.apply(T.weights_init_normal)
    D.apply(T.weights_init_normal)
    A, B = T.synthetic_pairs(batch, seed=73)
    ts = T.TrainStep(G, D, compute_dtype=cdt)
    ts.step(A.to(dev), B.to(dev))
    torch.cuda.synchronize()
    return ts.gflat.grad.clone().double(), ts.dflat.grad.clone().double()


for cdt in (torch.float32, torch.bfloat16):
    for batch in (2, 8):
        r = {k: run(on, cdt, batch) for k, on in (("off1", False), ("off2", False), ("on1", True), ("on2", True))}
        for a, b in (("off1", "off2"), ("on1", "on2"), ("on1", "off1"), ("on2", "off2")):
            cg = F.cosine_similarity(r[a][0], r[b][0], dim=0).item()
            cd = F.cosine_similarity(r[a][1], r[b][1], dim=0).item()
            mg = ((r[a][0] - r[b][0]).abs().max() / r[b][0].abs().max()).item()
            md = ((r[a][1] - r[b][1]).abs().max() / r[b][1].abs().max()).item()
            print(f"{str(cdt):15s} batch {batch}  {a} vs {b}: cos G {cg:.6f} D {cd:.6f}   max rel diff G {mg:.2e} D {md:.2e}", flush=True)
