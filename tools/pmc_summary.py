#!/usr/bin/env python
"""Per-kernel totals of one rocprofv3 --pmc counter from its CSV output (one counter per pass).

    python tools/pmc_summary.py gpurun_out/pmc_fetch gpurun_out/pmc_write
prints a markdown table: kernel, dispatches, mean FETCH_SIZE and WRITE_SIZE per dispatch (raw counter units, KB).
"""
import csv
import glob
import os
import sys
from collections import defaultdict


def load(d):
    out = defaultdict(lambda: [0, 0.0])
    name = None
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            k = row.get("Kernel_Name") or row.get("kernel_name")
            name = row.get("Counter_Name") or name
            out[k][0] += 1
            out[k][1] += float(row.get("Counter_Value") or 0)
    return name, out


def main(dirs):
    tabs = [load(d) for d in dirs]
    kernels = sorted({k for _, t in tabs for k in t}, key=lambda k: -max(t[k][1] if k in t else 0 for _, t in tabs))
    print("| kernel | dispatches | " + " | ".join(f"{n} per dispatch (raw)" for n, _ in tabs) + " |")
    print("|---|---:|" + "---:|" * len(tabs))
    for k in kernels:
        n = max(t[k][0] for _, t in tabs if k in t)
        vals = [("%.0f" % (t[k][1] / t[k][0])) if k in t and t[k][0] else "-" for _, t in tabs]
        print(f"| `{k[:70]}` | {n} | " + " | ".join(vals) + " |")


if __name__ == "__main__":
    main(sys.argv[1:])
