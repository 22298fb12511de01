"""Stability soak: many consecutive training steps at the benchmark size; every loss must stay finite and the parameters bounded."""
import sys, time, torch
sys.path.insert(0, __file__.rsplit('/', 2)[0])
import tfc_gan_amd as T
dev = torch.device('cuda', 0)
T.set_compute_dtype(torch.bfloat16); torch.manual_seed(42)
G = T.GeneratorUNet((3,256,256)).to(dev); D = T.Discriminator1((3,256,256)).to(dev)
G.apply(T.weights_init_normal); D.apply(T.weights_init_normal)
ts = T.TrainStep(G, D, compute_dtype=torch.bfloat16)
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
t0 = time.perf_counter()
for i in range(steps):
    A, B = T.synthetic_pairs(32, seed=1000 + (i % 8))
    out = ts.step(A.to(dev), B.to(dev))
    if i % 50 == 49 or i == steps - 1:
        vals = {k: float(v) for k, v in out.items() if k != "fake_B"}
        assert all(v == v and abs(v) < 1e9 for v in vals.values()), (i, vals)
        print(i + 1, {k: round(v, 4) for k, v in vals.items() if k in ("loss_G", "loss_D", "loss_triplet_patch", "loss_FFT")}, flush=True)
torch.cuda.synchronize()
pg, pd = ts.gflat.data, ts.dflat.data
assert torch.isfinite(pg).all() and torch.isfinite(pd).all()
print(f"{steps} steps ok in {time.perf_counter() - t0:.1f} s; |G params| max {pg.abs().max().item():.3f}, |D params| max {pd.abs().max().item():.3f}")
