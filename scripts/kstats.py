"""Print a rocprofv3 kernel_stats.csv as per-step milliseconds.  usage: python scripts/kstats.py <csv> <steps> [rows]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = int(sys.argv[2]); top = int(sys.argv[3]) if len(sys.argv) > 3 else 40
tot = sum(float(r['TotalDurationNs']) for r in rows)
print(f"total {tot/1e6/steps:.3f} ms/step over {steps} steps")
for r in rows[:top]:
    print(f"{r['Name'][:100]:100s} {int(r['Calls'])/steps:6.1f}/step {float(r['TotalDurationNs'])/1e6/steps:7.3f} ms {float(r['AverageNs'])/1e3:8.1f} us {float(r['Percentage']):5.1f}%")
