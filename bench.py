#!/usr/bin/env python3
"""Benchmark of the PATCH-16 training hot path (BASELINE.json metric) on N MI355X GPUs of one node.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python bench.py --gpus N ...        (launches its own N ranks: a child `torch.distributed.run`, rank 0's JSON line relayed, non-zero exit if any rank fails)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...   (the driver's form)

One process per GPU; each rank trains on its own 32 synthetic 256x256 thermal/visible pairs already resident in HBM
(weak scaling), gradients are all-reduced over RCCL.  A "step" = the full generator step + discriminator step of
TFC-GAN-FFT/TFCGAN_multigpu_patchFFT_16P.py:545-638 (without LPIPS / temperature head).  Rank 0 prints ONE
JSON line.  `roofline` is measured live with hipEvents around the launches of the dominant kernel (tfc_igemm2_kernel, the persistent
halo-staged implicit-GEMM convolution) in every 4th step of the timed region; `cpu_baseline` times the CPU oracle (torch fp32 restatement
of the same step) on this box's host cores on a bounded sample (rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PER_GPU_BATCH = 32
MFMA_BF16_PEAK_TFLOPS = 2500.0          # dense bf16 MFMA peak, /opt/skills/guides/MI355X_MICROARCH.md
MFMA_F32_PEAK_TFLOPS = 157.3            # fp32 MFMA (v_mfma_f32_32x32x2_f32): 1/16 of the bf16 rate, same guide


def _cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown CPU"


def cpu_baseline(budget_s=8.0):
    """The oracle's TrainStep (torch fp32 restatement of the same step, LPIPS / temperature head excluded as BASELINE.md section 3 defines it)
    on the host cores of this box. BASELINE.md section 3 asks for N=4 and N=32 with >= 5 warm-up + >= 20 timed steps; at ~3 images/s that is
    several minutes of CPU time, and this leg is bounded to ~30 s so that the default run still finishes in a few minutes: N=4 gets 2 warm-up
    steps and then as many timed steps as fit in `budget_s` (at least 5); N=32 (the metric's batch) gets ONE timed step after that warm-up.
    `value` is the N=32 figure. Threads: the box's CPU share for one GPU is 16 cores (more torch threads than that only oversubscribe)."""
    import torch
    from oracle import tfcgan_oracle as O
    cores = min(os.cpu_count() or 1, 16)
    torch.set_num_threads(cores)
    G = O.GeneratorUNet((3, 256, 256))
    D = O.Discriminator1((3, 256, 256))
    G.apply(O.weights_init_normal)
    D.apply(O.weights_init_normal)
    ts = O.TrainStep(G, D)
    neg = list(range(1, 16)) + [0]
    A, B = O.synthetic_pairs(4, seed=2)
    for _ in range(2):
        ts.step(A, B, neg)
    n4, t0 = 0, time.perf_counter()
    while n4 < 5 or time.perf_counter() - t0 < budget_s:
        ts.step(A, B, neg)
        n4 += 1
    dt4 = time.perf_counter() - t0
    A, B = O.synthetic_pairs(32, seed=3)
    ts.step(A, B, neg)                                            # one warm-up step at this batch (allocator, thread pools): round 2 timed a cold step
    t0 = time.perf_counter()
    ts.step(A, B, neg)
    dt32 = time.perf_counter() - t0
    return {"value": 32 / dt32, "unit": "images/sec", "cores": cores, "kind": "port", "cpu_model": _cpu_model(),
            "n4_images_per_sec": 4 * n4 / dt4,
            "sample": f"full PATCH-16 steps (G step + D step, triplet16 + patch-FFT, Adam; torch fp32, {cores} threads on {_cpu_model()}): "
                      f"batch 4: 2 warm-up + {n4} timed steps in {dt4:.1f} s; batch 32 (value): 1 warm-up + 1 timed step in {dt32:.1f} s"}


PMC_TRAFFIC_FILE = "r03_pmc_traffic.json"   # written by scripts/pmc_traffic.py from the two --pmc passes of THIS command
DOMINANT_KERNEL = "tfc_igemm2_kernel"
PROF_EVERY = int(os.environ.get("TFC_BENCH_PROF_EVERY", "4"))   # instrument every 4th timed step with hipEvents (see main; the knob is for A/B runs)


def pmc_traffic(kernel):
    """HBM bytes per launch of `kernel` from the rocprofv3 PMC passes of THIS command (FETCH_SIZE and WRITE_SIZE collected in separate
    runs; FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950) -- profiles/<PMC_TRAFFIC_FILE>, or None if absent"""
    path = os.path.join(ROOT, "profiles", PMC_TRAFFIC_FILE)
    if not os.path.exists(path):
        return None
    with open(path) as f:
        d = json.load(f).get(kernel)
    return None if d is None else d["hbm_bytes_per_launch_corrected"]


def generator_l1(dev):
    """generator L1 vs the oracle on one synthetic image, fp32 parity mode and bf16 mode (the 'gen L1 vs ref' half of the metric)"""
    import torch
    import tfc_gan_amd as T
    from oracle import tfcgan_oracle as O
    A, _ = O.synthetic_pairs(1, seed=11)
    Gc = O.init_weights_portable(O.GeneratorUNet((3, 256, 256)), seed=3).eval()
    with torch.no_grad():
        want = Gc(A)
    out = {}
    for name, dtype in (("fp32", torch.float32), ("bf16", torch.bfloat16)):
        T.set_compute_dtype(dtype)
        G = T.GeneratorUNet((3, 256, 256))
        G.load_state_dict(Gc.state_dict())
        G = G.to(dev).eval()
        with torch.no_grad():
            got = G(A.to(dev)).cpu()
        out[name] = float((got - want).abs().mean())
    T.set_compute_dtype(torch.bfloat16)
    return out


def self_launch(ngpus):
    """`python bench.py --gpus N` without an outer launcher: start the N ranks as a CHILD `python -m torch.distributed.run` (this parent has imported
    nothing that touches the GPU and never execs), let the child's stdout / stderr through (rank 0 prints the one JSON line) and exit with its code."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(ngpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=PER_GPU_BATCH, help="per-GPU batch (BASELINE config: 32)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--dtype", choices=["bf16", "fp32"], default="bf16",
                    help="bf16 = the benchmarked arithmetic (BASELINE.json); fp32 = the exact-fp32 parity mode (generator L1 vs oracle 1.9e-6), on the "
                         "legacy one-tile-per-workgroup kernels at 1/16 of the MFMA rate -- reported so that its cost is a number, not a guess")
    ap.add_argument("--lpips", action="store_true", help="include the LPIPS term (P16:598) in the timed step (seeded-random VGG16 weights)")
    ap.add_argument("--config", choices=["patch16", "glo16", "stn21"], default="patch16",
                    help="patch16 = BASELINE.json configs[1] (the metric's configuration); glo16 = configs[2] (TFCGAN_multigpu_globalFFT_16P.py: "
                         "whole-image FFT loss instead of the 16 patch FFTs); stn21 = configs[4] "
                         "(TFCGAN_STN21_Original_NewModel3_Official.py: two generators, two discriminators, localiser + warp, LPIPS with seeded-random VGG16 "
                         "weights, morphological triplet; tfc_gan_amd.STN21Step)")
    args = ap.parse_args()

    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")               # before the HIP runtime starts (tfc_gan_amd/__init__.py says why); inherited by self-launched ranks
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and "RANK" not in os.environ:               # no outer launcher: become one (before anything initialises the GPU)
        raise SystemExit(self_launch(args.gpus))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} under a launcher with WORLD_SIZE={world}: the two must agree")

    import torch
    import torch.distributed as dist
    import tfc_gan_amd as T
    from tfc_gan_amd import parallel
    assert torch.cuda.is_available(), "bench.py needs a GPU"
    if os.environ.get("TFC_BENCH_BACKEND", "nccl") != "nccl":
        local_rank = local_rank % max(torch.cuda.device_count(), 1)   # rehearsal: more ranks than cards
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("TFC_BENCH_BACKEND", "nccl")      # "gloo": rehearsal of the multi-rank path on ONE card (ranks share cuda:0)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
    elif os.environ.get("TFC_FORCE_COLLECTIVES", "0") not in ("", "0"):
        # one-GPU rehearsal of the RCCL path: a ONE-rank "nccl" group; every bucket all-reduce / broadcast / loss average of the product path is issued
        # (identities in value, real in stream ordering and launch cost): `exposed_allreduce_ms` is then the floor the exchange adds at any N
        import socket
        sk = socket.socket()
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
        sk.close()
        try:
            dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=dev)
        except Exception as e:                                    # an environment problem, not a measurement: exit code 77
            print("one-rank RCCL group unavailable:", repr(e), file=sys.stderr, flush=True)
            raise SystemExit(77)

    cdt = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    T.set_compute_dtype(cdt)
    torch.manual_seed(42)                                # reference: torch.manual_seed(42), P16:61
    G = T.GeneratorUNet((3, 256, 256)).to(dev)
    D = T.Discriminator1((3, 256, 256)).to(dev)
    G.apply(T.weights_init_normal)
    D.apply(T.weights_init_normal)
    bucket_kw = {"bucket_bytes": int(os.environ["TFC_BUCKET_MB"]) << 20} if os.environ.get("TFC_BUCKET_MB") else {}   # A/B knob
    stn = args.config == "stn21"
    if stn:
        import warnings
        assert not args.lpips, "--config stn21 always carries the LPIPS term (STN:640)"
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")                       # "seeded-random weights" -- stated in config.workload
            stn_crit = T.LPIPS().to(dev)
        del G, D
        ts = T.STN21Step((3, 256, 256), lpips=stn_crit, device=dev, **bucket_kw)
    else:
        ts = T.TrainStep(G, D, compute_dtype=cdt, fft_mode="patch" if args.config == "patch16" else "global", **bucket_kw)
    A, B = T.synthetic_pairs(args.batch, seed=1234 + rank)      # the product's own recipe: oracle/ is used by the checker legs only
    A, B = A.to(dev), B.to(dev)

    extra = None
    crit = None

    def make_lpips():
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")                       # "seeded-random weights" -- stated in config.workload instead
            return T.LPIPS(net_type="vgg", version="0.1").to(dev)
    if args.lpips:
        crit = make_lpips()
        extra = crit.as_extra_loss(0.5)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def run_step():
        return ts.step(A, B) if stn else ts.step(A, B, extra_loss_G=extra)

    for _ in range(args.warmup):
        run_step()
    barrier()
    # The roofline object is measured live, inside the timed region, with hipEvent pairs around the MFMA kernel launches on the launch
    # stream. An event pair costs ~2.3 us of stream time (A/B in scripts/ab_prof.py: 0.45 ms per fully instrumented step, 3.5 %), so
    # every PROF_EVERY-th timed step is instrumented, not all of them: `value` then carries < 1 % of instrumentation overhead.
    if os.environ.get("TFC_BENCH_NO_EXPOSED", "0") in ("", "0"):   # (A/B knob: what does the measurement itself cost?)
        parallel.exposed_wait_begin()                      # hipEvent pairs around the waits of BucketReducer.finish(): `exposed_allreduce_ms`
    t0 = time.perf_counter()
    # The product runs its weight gradients on a second stream beside the input-gradient chain (nets.py). A launch that shares the chip with another
    # kernel has no duration of its own, so the INSTRUMENTED steps (and only they) run everything in line on one stream: the roofline object then
    # prices exclusive launches, `value` is the mix actually run (every PROF_EVERY-th step serial, the rest overlapped).
    side_default = T.set_wgrad_stream(True)
    T.set_wgrad_stream(side_default)
    for i in range(args.steps):
        inst = i % PROF_EVERY == 0
        T.ops.prof_enable(inst)
        T.set_wgrad_stream(side_default and not inst)
        out = run_step()
    barrier()
    elapsed = time.perf_counter() - t0
    T.ops.prof_enable(False)
    T.set_wgrad_stream(side_default)
    recs = T.ops.prof_records(16384)
    ig_ms, ig_flop, ig_n = T.ops.prof_collect(0)
    wg_ms, wg_flop, wg_n = T.ops.prof_collect(1)
    T.ops.prof_collect(2)                                 # wgrad finish passes (no algorithmic FLOP of their own)
    fb_ms, fb_flop, fb_n = T.ops.prof_collect(3)          # fused first-block backward ([BlurPool]^T + LeakyReLU' + weight gradient in one launch)
    # the dominant kernel proper: class-0 calls that dispatch tfc_igemm2_kernel (bf16, whole 64-byte channel chunks, NHWC output); the rest of
    # class 0 (first-layer tfc_conv_c8_kernel, the two 3-channel heads) is reported beside it
    p8 = lambda c: (c + 7) // 8 * 8  # noqa: E731
    es = 2 if args.dtype == "bf16" else 4
    dom = [r for r in recs if r["kclass"] == 0 and (p8(r["Cin"]) * es) % 64 == 0 and (p8(r["Cout"]) * es) % 64 == 0]
    peak = MFMA_BF16_PEAK_TFLOPS if args.dtype == "bf16" else MFMA_F32_PEAK_TFLOPS
    kname = DOMINANT_KERNEL if args.dtype == "bf16" else "tfc_igemm_kernel"
    dom_ms, dom_flop = sum(r["ms"] for r in dom), sum(r["flop"] for r in dom)
    dom_launches = len(dom)                               # one launch per call: the four sub-pixel phases of a transposed convolution fold into one grid
    loss_g, loss_d = float(out["loss_G"]), float(out["loss_D"])
    exposed_ms = parallel.exposed_wait_collect() / args.steps    # stream time spent waiting for gradient buckets, per step (0 at one rank)
    tmax = torch.tensor([elapsed, exposed_ms], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    elapsed, exposed_ms = float(tmax[0].item()), float(tmax[1].item())

    if rank == 0:
        ms_per_step = 1e3 * elapsed / args.steps
        value = args.batch * world * args.steps / elapsed
        achieved = (dom_flop / 1e12) / (dom_ms / 1e3) if dom_ms > 0 else 0.0
        nprof = len(range(0, args.steps, PROF_EVERY))                       # timed steps that carried the hipEvent instrumentation
        step_s = elapsed / args.steps
        line = {
            "metric": ("training images/sec at 256x256 STN21_Original_NewModel3_Official (BASELINE.json configs[4]; not the headline metric)" if stn else
                       "training images/sec at 256x256 PATCH-16, 1/2/4/8 MI355X; gen L1 vs ref"),
            "value": value, "unit": "images/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": ("STN21_Original_NewModel3_Official 256x256 " + args.dtype + ", batch 32 per GPU (BASELINE.json configs[4]): localiser + affine warp, two "
                                    "generators, two discriminators, morphological triplet, 0.5 * LPIPS with seeded-random VGG16 weights (same work, not the "
                                    "published metric), Adam on flat buffers (STN:609-672)") if stn else
                                   (("PATCH-16 256x256 " + args.dtype + ", batch 32 per GPU (BASELINE.json configs[1]; configs[3] at 8 GPUs): "
                                     "G step + D step, 16-patch triplet + patch-FFT loss, Adam") if args.config == "patch16" else
                                    ("GLO-16 (TFCGAN_multigpu_globalFFT_16P.py) 256x256 " + args.dtype + ", batch 32 per GPU (BASELINE.json configs[2]): "
                                     "G step + D step, 16-patch triplet + whole-image FFT loss, Adam")) +
                                   (" + 0.5 * LPIPS(fake_B, real_B) (P16:598, seeded-random VGG16 weights: same work, not the published metric)" if args.lpips else
                                    " (step as BASELINE.md section 3 defines it: LPIPS and the temperature head excluded)"),
                       "global_batch": args.batch * world, "per_gpu_batch": args.batch, "parallelism": f"dp{world}",
                       "algorithmic_gflop_per_image": None if stn else T.TrainStep.STEP_GFLOP},
            "roofline": {"bound": "mfma", "kernel": kname + (" (persistent halo-staged implicit-GEMM conv: fwd / dgrad / convT / upconv, bf16 MFMA 32x32x16)" if args.dtype == "bf16"
                                                             else " (one-tile-per-workgroup halo-staged implicit GEMM, fp32 MFMA 32x32x2: exact fp32 parity mode)"),
                         "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak,
                         "traffic": pmc_traffic(DOMINANT_KERNEL) if args.dtype == "bf16" else None, "traffic_unit": "HBM bytes per launch (PMC)", "launches": dom_launches,
                         "avg_launch_ms": dom_ms / max(dom_launches, 1), "algorithmic_gflop_per_launch": dom_flop / 1e9 / max(dom_launches, 1),
                         "instrumented_steps": nprof, "share_of_step_time": (dom_ms / 1e3 / nprof) / step_s,
                         "instrumented_mode": ("one stream: exclusive launches (the other timed steps run the weight gradients on a second stream)"
                                               if side_default else "one stream (TFC_WGRAD_STREAM=0: every step)"),
                         "whole_conv_class": {"kernels": "tfc_igemm2_kernel + tfc_first_block_fwd_kernel (conv + LeakyReLU + BlurPool of the first blocks, priced at the conv FLOP) + 3-channel head kernels", "achieved": (ig_flop / 1e12) / (ig_ms / 1e3) if ig_ms > 0 else 0.0,
                                              "unit": "TFLOP/s", "calls": ig_n, "share_of_step_time": (ig_ms / 1e3 / nprof) / step_s},
                         "second_kernel": {"kernel": "tfc_wgrad_kernel family (incl. slab reduction)", "achieved": (wg_flop / 1e12) / (wg_ms / 1e3) if wg_ms > 0 else 0.0,
                                           "unit": "TFLOP/s", "launches": wg_n, "share_of_step_time": (wg_ms / 1e3 / nprof) / step_s},
                         "fused_first_block_backward": {"kernel": "tfc_wgrad_c8_fusedm_kernel (transposed blur as a GEMM on the matrix core + first-layer weight gradient; sign words instead of the stored activation)",
                                                        "calls": fb_n, "avg_call_ms": fb_ms / max(fb_n, 1), "share_of_step_time": (fb_ms / 1e3 / nprof) / step_s}},
            "whole_step_tflops": None if stn else T.TrainStep.STEP_GFLOP * value / 1e3,
            # stream time between "all buckets issued" and "all buckets arrived" in BucketReducer.finish(), summed over the generator's and the
            # discriminator's exchange, max over ranks (DESIGN section 6 predicts 0.09-0.18 ms at 8 GPUs; the generator's part sits on the side stream
            # beside the discriminator step, so this is an upper bound of what the exchange adds to the step)
            "exposed_allreduce_ms": exposed_ms, "allreduce_backend": (dist.get_backend() if parallel.collectives_active() else None),
            "final_losses": {"loss_G": loss_g, "loss_D": loss_d},
        }
        if world == 1 and not stn and not args.lpips and not args.no_cpu_baseline and args.dtype == "bf16":
            # the reference's full loss_G also carries 0.5 * LPIPS (P16:598, :607): the same step with that term, timed right after
            crit = make_lpips()
            term = crit.as_extra_loss(0.5)
            for _ in range(2):
                ts.step(A, B, extra_loss_G=term)
            torch.cuda.synchronize()
            k2 = max(5, args.steps // 2)
            t1 = time.perf_counter()
            for _ in range(k2):
                ts.step(A, B, extra_loss_G=term)
            torch.cuda.synchronize()
            dt2 = time.perf_counter() - t1
            line["with_lpips"] = {"value": args.batch * k2 / dt2, "unit": "images/sec", "ms_per_step": 1e3 * dt2 / k2, "steps": k2,
                                  "note": "same step + 0.5 * LPIPS(fake_B, real_B): VGG16 forward x2 + backward w.r.t. fake_B, seeded-random weights"}
        if world == 1 and not args.no_cpu_baseline and not stn:
            line["generator_l1_vs_oracle"] = generator_l1(dev)
            line["cpu_baseline"] = cpu_baseline()
        else:
            line["cpu_baseline"] = None                           # (stn21: the CPU port in oracle/ is the PATCH-16 / GLO-16 step)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
