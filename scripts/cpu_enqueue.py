"""How far ahead of the GPU is the host? Time to ENQUEUE one step (no synchronisation) vs the synchronised step time."""
import sys, time, torch
sys.path.insert(0, __file__.rsplit('/', 2)[0])
import tfc_gan_amd as T
dev = torch.device('cuda', 0)
T.set_compute_dtype(torch.bfloat16); torch.manual_seed(42)
G = T.GeneratorUNet((3,256,256)).to(dev); D = T.Discriminator1((3,256,256)).to(dev)
G.apply(T.weights_init_normal); D.apply(T.weights_init_normal)
ts = T.TrainStep(G, D, compute_dtype=torch.bfloat16)
A, B = T.synthetic_pairs(32, seed=1234); A, B = A.to(dev), B.to(dev)
for _ in range(5): ts.step(A, B)
torch.cuda.synchronize()
enq = []
t0 = time.perf_counter()
for _ in range(20):
    t1 = time.perf_counter(); ts.step(A, B); enq.append(time.perf_counter() - t1)
torch.cuda.synchronize(); tot = (time.perf_counter() - t0) / 20
print(f"enqueue per step: median {sorted(enq)[10]*1e3:.2f} ms, first {enq[0]*1e3:.2f} ms; synchronised step {tot*1e3:.2f} ms")
