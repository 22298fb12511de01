import sys, time, torch
sys.path.insert(0, '/root/repo')
import tfc_gan_amd as T
dev = torch.device('cuda', 0)
T.set_compute_dtype(torch.bfloat16); torch.manual_seed(42)
G = T.GeneratorUNet((3,256,256)).to(dev); D = T.Discriminator1((3,256,256)).to(dev)
G.apply(T.weights_init_normal); D.apply(T.weights_init_normal)
ts = T.TrainStep(G, D, compute_dtype=torch.bfloat16)
A, B = T.synthetic_pairs(32, seed=1234); A, B = A.to(dev), B.to(dev)
for _ in range(5): ts.step(A, B)
for rep in range(3):
    for prof in (False, True):
        T.ops.prof_enable(prof)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20): ts.step(A, B)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
        T.ops.prof_enable(False)
        if prof: T.ops.prof_collect(0); T.ops.prof_collect(1)
        print(f"prof={prof}: {dt*1e3:.3f} ms/step  {32/dt:.1f} img/s")
