"""Summarise rocprofv3 --pmc counter_collection.csv files per kernel: mean counter value per dispatch (+ kernel time from the trace rows).
usage: python scripts/pmc_summary.py <dir> [<dir> ...]"""
import collections
import csv
import glob
import sys


def main():
    for d in sys.argv[1:]:
        vals = collections.defaultdict(lambda: collections.defaultdict(list))
        dur = collections.defaultdict(list)
        for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                name = r["Kernel_Name"].replace("void ", "").split("(")[0][:60]
                if not name.startswith("tfc_"):
                    continue
                vals[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
                dur[(name, r["Dispatch_Id"])] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        print(f"== {d}")
        for name, cs in vals.items():
            ds = [v for (n, _), v in dur.items() if n == name]
            print(f"{name}  ({len(ds)} dispatches, mean {sum(ds) / len(ds):.1f} us)")
            for c, v in sorted(cs.items()):
                print(f"    {c:28s} {sum(v) / len(v):16.0f}")


if __name__ == "__main__":
    main()
