"""Where does the one-rank RCCL rehearsal lose its millisecond? One process, same box: the step (a) without collectives, (b) with a stream-ordered stand-in
for dist.all_reduce (torch streams + events only: the dependency structure of RCCL, none of ProcessGroupNCCL), (c) with the real one-rank "nccl" group,
(d) the same with the collectives issued but never waited for. Prints ms/step and the host's enqueue time per step."""
import os, socket, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist
import tfc_gan_amd as T
from tfc_gan_amd import parallel

dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
T.set_compute_dtype(torch.bfloat16)
A, B = T.synthetic_pairs(32, seed=1234)
A, B = A.to(dev), B.to(dev)


def timed(label):
    torch.manual_seed(42)
    G = T.GeneratorUNet((3, 256, 256)).to(dev); D = T.Discriminator1((3, 256, 256)).to(dev)
    G.apply(T.weights_init_normal); D.apply(T.weights_init_normal)
    ts = T.TrainStep(G, D, compute_dtype=torch.bfloat16)
    for _ in range(5):
        ts.step(A, B)
    torch.cuda.synchronize()
    enq = []
    t0 = time.perf_counter()
    for _ in range(20):
        t1 = time.perf_counter(); ts.step(A, B); enq.append(time.perf_counter() - t1)
    torch.cuda.synchronize()
    print(f"{label:58s} {1e3 * (time.perf_counter() - t0) / 20:7.3f} ms/step   host enqueue {1e3 * sorted(enq)[10]:6.2f} ms", flush=True)


timed("plain")
# (b) stand-in
comm = torch.cuda.Stream(dev)
class _Work:
    def __init__(self, ev): self.ev = ev
    def wait(self):
        torch.cuda.current_stream(dev).wait_event(self.ev); return True
def fake_all_reduce(t, op=None, group=None, async_op=False):
    comm.wait_stream(torch.cuda.current_stream(dev))
    ev = torch.cuda.Event(); ev.record(comm)
    w = _Work(ev)
    if async_op: return w
    w.wait(); return None
real_ar, real_active = dist.all_reduce, parallel.collectives_active
parallel.collectives_active = lambda: True
parallel.dist.all_reduce = fake_all_reduce
parallel.dist.broadcast = lambda *a, **k: None
timed("stand-in collectives (streams + events only)")
parallel.dist.all_reduce = real_ar
parallel.collectives_active = real_active
# (c) real one-rank group
os.environ["TFC_FORCE_COLLECTIVES"] = "1"
sk = socket.socket(); sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]; sk.close()
dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=dev)
import importlib
timed("real one-rank nccl group")
# (d) only the bucket all-reduces real, loss average skipped
orig_mean = parallel.all_reduce_mean
parallel.all_reduce_mean = lambda t, group=None: t
timed("real group, loss average skipped")
parallel.all_reduce_mean = orig_mean
# (e) bucket all-reduces replaced by the stand-in, loss average real
parallel.dist.all_reduce = lambda t, op=None, group=None, async_op=False: (fake_all_reduce(t, op, group, async_op) if async_op else real_ar(t, op=op, group=group))
timed("real group, bucket all-reduces by the stand-in")
parallel.dist.all_reduce = real_ar
dist.destroy_process_group()
