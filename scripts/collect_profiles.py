"""Copy the summaries scripts/collect_profiles.sh left under gpurun_out/<tag>_prof/ into profiles/ (tracked).  usage: python scripts/collect_profiles.py r02"""
import csv
import glob
import os
import shutil
import sys

tag = sys.argv[1]
src = f"gpurun_out/{tag}_prof"
os.makedirs("profiles", exist_ok=True)
stats = glob.glob(src + "/trace/**/*kernel_stats.csv", recursive=True)[0]
shutil.copy(stats, f"profiles/{tag}_bench_n1_kernel_stats.csv")
rows = list(csv.DictReader(open(stats)))
steps = 7
tot = sum(float(r["TotalDurationNs"]) for r in rows)
with open(f"profiles/{tag}_bench_n1_kernel_stats.md", "w") as f:
    f.write("# TFC_WGRAD_STREAM=0 rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline  (MI355X, batch 32, bf16; 7 steps incl. warm-up; one stream, so that a traced duration is kernel time -- the product default overlaps the weight gradients on a second stream)\n\n```\n")
    f.write(f"total {tot / 1e6 / steps:.3f} ms/step over {steps} steps\n")
    for r in rows[:60]:
        f.write(f"{r['Name'][:100]:100s} {int(r['Calls']) / steps:6.1f}/step {float(r['TotalDurationNs']) / 1e6 / steps:7.3f} ms "
                f"{float(r['AverageNs']) / 1e3:8.1f} us {float(r['Percentage']):5.1f}%\n")
    f.write("```\n")
shutil.copy(src + "/per_layer.md", f"profiles/{tag}_per_layer.md")
shutil.copy(src + "/pmc_traffic.json", f"profiles/{tag}_pmc_traffic.json")
shutil.copy(src + "/bench.json", f"profiles/{tag}_bench_n1.json")
shutil.copy(src + "/bench_traced.json", f"profiles/{tag}_bench_n1_traced.json")
print(open(f"profiles/{tag}_bench_n1_kernel_stats.md").read()[:1500])
