"""Stability soak: many consecutive training steps at the benchmark size. Asserted: every loss finite; the discriminator LEARNS (loss_D of the last
20 steps below its start and below 0.5 in the median) while staying positive and, isolated spikes apart, below 2 (no collapse of the relativistic game); parameters bounded; and the module's own forward (operand streams re-packed from the final weights) reproduces the engine's
last fake_B -- a stale packed-weight cache would show here."""
import sys, time, torch
sys.path.insert(0, __file__.rsplit('/', 2)[0])
import tfc_gan_amd as T
dev = torch.device('cuda', 0)
T.set_compute_dtype(torch.bfloat16); torch.manual_seed(int(sys.argv[2]) if len(sys.argv) > 2 else 42)
G = T.GeneratorUNet((3,256,256)).to(dev); D = T.Discriminator1((3,256,256)).to(dev)
G.apply(T.weights_init_normal); D.apply(T.weights_init_normal)
ts = T.TrainStep(G, D, compute_dtype=torch.bfloat16)
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
t0 = time.perf_counter()
trip, lossd = [], []
for i in range(steps):
    A, B = T.synthetic_pairs(32, seed=1000 + (i % 8))
    out = ts.step(A.to(dev), B.to(dev))
    trip.append(out["loss_triplet_patch"]); lossd.append(out["loss_D"])
    if i % 50 == 49 or i == steps - 1:
        vals = {k: float(v) for k, v in out.items() if k != "fake_B"}
        assert all(v == v and abs(v) < 1e9 for v in vals.values()), (i, vals)
        print(i + 1, {k: round(v, 4) for k, v in vals.items() if k in ("loss_G", "loss_D", "loss_triplet_patch", "loss_FFT")}, flush=True)
torch.cuda.synchronize()
pg, pd = ts.gflat.data, ts.dflat.data
assert torch.isfinite(pg).all() and torch.isfinite(pd).all()
trip = torch.stack([t.float() for t in trip]).cpu(); lossd = torch.stack([t.float() for t in lossd]).cpu()
if steps >= 100:                                                  # the discriminator learns to tell the pairs apart (0.69 at the start)
    # (median: an isolated spike of the relativistic game inside the window -- seed-dependent, seen with every kernel variant -- must not decide this)
    assert lossd[-20:].median() < lossd[:3].mean() and lossd[-20:].median() < 0.5, (lossd[:3].mean(), lossd[-20:].median())
print(f"loss_D: min {lossd.min().item():.4f}, max {lossd.max().item():.4f} at step {int(lossd.argmax())}, mean of the last 20 {lossd[-20:].mean().item():.4f}; "
      f"steps above 1.0: {int((lossd > 1.0).sum())}", flush=True)
# the relativistic game does not collapse: loss_D stays positive and, apart from isolated spikes right after the generator catches up (a handful of steps,
# largest seen 10.5 with the fused first block and 3.4 without it, on different seeds: six seeds x two variants, round 3), below 2
assert 0.0 < lossd.min() and lossd.max() < 50.0 and int((lossd > 2.0).sum()) <= 3, (lossd.min(), lossd.max(), int((lossd > 2.0).sum()))
assert pg.abs().max() < 10 and pd.abs().max() < 50
G.eval()
A, B = T.synthetic_pairs(2, seed=7)
with torch.no_grad():
    y_mod = G(A.to(dev))
y_eng, _ = ts.G.forward(A.to(dev), seed=0, train=False, save=False)
# two runs of the SAME bf16 network differ by the chaotic amplification of fp32 atomics order in the InstanceNorm sums (max |diff| ~0.05 observed); weights
# that were stale by even a few Adam steps would differ by an order of magnitude more in the mean
md = (y_mod - y_eng).abs().mean().item()
print(f"module forward vs engine forward after {steps} steps: mean |diff| {md:.2e}")
assert md < 5e-3, md
print(f"{steps} steps ok in {time.perf_counter() - t0:.1f} s; |G params| max {pg.abs().max().item():.3f}, |D params| max {pd.abs().max().item():.3f}")
