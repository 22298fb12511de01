"""HBM traffic per launch from two rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE cannot share a pass on gfx950):
    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d <dir>/fetch -o p -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline
    rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d <dir>/write -o p -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline
    python scripts/pmc_traffic.py <dir> profiles/r01_pmc_traffic.json
Corrections as /opt/skills/guides/MI355X_MICROARCH.md (HBM section) prescribes: both counters are in KB; on gfx950 FETCH_SIZE reports half
of the bytes of wide coalesced reads, so it is doubled; WRITE_SIZE is exact for 16-byte-per-lane stores."""
import collections
import csv
import glob
import json
import sys


def per_kernel(d, counter):
    out = collections.defaultdict(lambda: [0.0, 0])
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            name = r["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0]
            out[name][0] += float(r["Counter_Value"])
            out[name][1] += 1
    return out


def main():
    d, dst = sys.argv[1], sys.argv[2]
    fetch, write = per_kernel(d + "/fetch", "FETCH_SIZE"), per_kernel(d + "/write", "WRITE_SIZE")
    res = {}
    for k in sorted(fetch, key=lambda k: -fetch[k][0]):
        if not k.startswith("tfc_") or fetch[k][1] == 0:
            continue
        n = fetch[k][1]
        fk = fetch[k][0] / n
        wk = write[k][0] / max(write[k][1], 1) if k in write else 0.0
        res[k] = {"launches_profiled": n, "FETCH_SIZE_KB_per_launch": fk, "WRITE_SIZE_KB_per_launch": wk,
                  "hbm_bytes_per_launch_corrected": (2.0 * fk + wk) * 1024.0}
    if not res:
        sys.exit(f"no counter data under {d}: {dst} left untouched")
    json.dump(res, open(dst, "w"), indent=1)
    for k, v in list(res.items())[:12]:
        print(f"{k:34s} {v['launches_profiled']:5d} launches  {v['hbm_bytes_per_launch_corrected'] / 1e6:9.1f} MB/launch")


if __name__ == "__main__":
    main()
