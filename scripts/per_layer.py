#!/usr/bin/env python3
"""Per-layer roofline table of the convolution family from a rocprofv3 kernel trace.

    TFC_LAUNCH_LOG=gpurun_out/X/launch.log rocprofv3 --kernel-trace --output-format csv -d gpurun_out/X -o p -- \
        python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline
    python scripts/per_layer.py gpurun_out/X gpurun_out/X/launch.log profiles/r02_per_layer.md [skip_calls]

The library appends one line per convolution-class API call to $TFC_LAUNCH_LOG (api.hip: ProfScope):
    kclass op pass N H W Cin Cout flop nlaunch
in launch order; the kernel trace lists every dispatch in the same order.  Walking both in lockstep (only the kernels the conv-class
launchers issue are considered) attributes each dispatch to its call; calls with equal (kclass, op, pass, shape) are then averaged.
Peak: 2.5 PFLOP/s dense bf16 MFMA (MI355X_MICROARCH.md)."""
import collections
import csv
import glob
import sys

PEAK = 2500.0
CONV_KERNELS = ("tfc_igemm_kernel", "tfc_conv_c8_kernel", "tfc_upconv_head_kernel", "tfc_dgrad_rows4_kernel", "tfc_wgrad_kernel", "tfc_wgrad22_kernel",
                "tfc_wgradT_kernel", "tfc_wgradT2_kernel", "tfc_wgrad_reduce_kernel", "tfc_wgrad_finish_kernel", "tfc_igemm2_kernel", "tfc_wgrad_fin_kernel", "tfc_wgrad_reduce_fin_kernel", "tfc_wgrad_c8_kernel", "tfc_wgrad_c8_reduce_kernel", "tfc_wgrad_head_kernel", "tfc_wgrad_head_reduce_kernel", "tfc_wgrad_c8_fused_kernel", "tfc_wgrad_c8_fusedm_kernel",
                "tfc_wgrad_head_finish_kernel", "tfc_dgrad_head_kernel", "tfc_first_block_fwd_kernel")
OPN = {0: "conv", 1: "padconv", 2: "convT", 3: "upconv"}
PASSN = {0: "fwd", 1: "dgrad", 2: "wgrad", 3: "wgrad-finish"}
CLASSN = {0: "gather GEMM", 1: "weight gradient (+ slab reduce)", 2: "wgrad finish", 3: "fused first-block backward ([BlurPool]^T + LeakyReLU' + wgrad)"}


def short(name):
    n = name.replace("void ", "")
    return n.split("<")[0].split("(")[0]


def main():
    d, logp, dst = sys.argv[1], sys.argv[2], sys.argv[3]
    rows = []
    for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            if k in CONV_KERNELS:
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), k, r["Kernel_Name"]))
    rows.sort()
    calls = [l.split() for l in open(logp) if l.strip()]
    need = sum(int(c[9]) for c in calls)
    if need != len(rows):
        print(f"warning: log expects {need} dispatches, trace has {len(rows)} conv-class dispatches; aligning from the END", file=sys.stderr)
        if need > len(rows):
            sys.exit("trace is shorter than the launch log")
        rows = rows[len(rows) - need:]
    agg = collections.OrderedDict()
    i = 0
    for c in calls:
        kclass, op, pas, N, H, W, Cin, Cout = (int(x) for x in c[:8])
        flop, nl = float(c[8]), int(c[9])
        mine = rows[i:i + nl]
        i += nl
        dur = sum(e - s for s, e, _, _ in mine) / 1e3           # us of kernel time (sum over the call's dispatches)
        key = (kclass, op, pas, N, H, W, Cin, Cout)
        a = agg.setdefault(key, {"n": 0, "us": 0.0, "flop": flop, "kern": collections.Counter()})
        a["n"] += 1
        a["us"] += dur
        for _, _, k, full in mine:
            tmpl = full.replace("void ", "").split("(")[0]
            a["kern"][tmpl.replace("unsigned short", "bf16")] += 1
    # the first calls of a process are warm-up (cold caches / lazy module load): report per-call means, they are dominated by steady state
    out = ["# Per-layer table of the convolution family (rocprofv3 kernel trace joined with the library's launch log)", "",
           f"source: `{d}`; {len(calls)} calls, {len(rows)} dispatches. `us` = mean kernel time per call (all dispatches of the call), "
           f"`frac` = TFLOP/s / {PEAK:.0f}.", "",
           "| class | op | pass | N x H x W | Cin -> Cout | calls | us / call | GFLOP | TFLOP/s | frac | kernels |", "|---|---|---|---|---|---|---|---|---|---|---|"]
    tot = collections.defaultdict(lambda: [0.0, 0.0])
    for (kclass, op, pas, N, H, W, Cin, Cout), a in agg.items():
        us = a["us"] / a["n"]
        tf = a["flop"] / us / 1e6 if us > 0 else 0.0
        kern = ", ".join(f"{k} x{v // a['n']}" if v // a["n"] > 1 else k for k, v in a["kern"].items())
        out.append(f"| {kclass} | {OPN[op]} | {PASSN[pas]} | {N}x{H}x{W} | {Cin}->{Cout} | {a['n']} | {us:.1f} | {a['flop'] / 1e9:.1f} | {tf:.0f} | {tf / PEAK:.3f} | {kern} |")
        tot[kclass][0] += a["us"]
        tot[kclass][1] += a["flop"] * a["n"]
    out.append("")
    for kclass, (us, fl) in sorted(tot.items()):
        if us > 0:
            out.append(f"class {kclass} ({CLASSN.get(kclass, '?')}): {us / 1e3:.2f} ms total, {fl / us / 1e6:.0f} TFLOP/s = {fl / us / 1e6 / PEAK:.3f} of peak")
    # wgrad family including its finish pass
    us_w = tot[1][0] + tot[2][0]
    if us_w > 0:
        out.append(f"class 1+2 (wgrad incl. slab reduce and finish): {us_w / 1e3:.2f} ms, {tot[1][1] / us_w / 1e6:.0f} TFLOP/s = {tot[1][1] / us_w / 1e6 / PEAK:.3f} of peak")
    open(dst, "w").write("\n".join(out) + "\n")
    print("\n".join(out))


if __name__ == "__main__":
    main()
