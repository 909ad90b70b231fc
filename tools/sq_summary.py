#!/usr/bin/env python
"""Per-kernel sums of several rocprofv3 --pmc counters collected in ONE pass (CSV output).

    python tools/sq_summary.py gpurun_out/pmc_sq
prints a markdown table: kernel, dispatches, then one column per counter (sum over dispatches)."""
import csv
import glob
import os
import sys
from collections import defaultdict


def main(d):
    tot = defaultdict(lambda: defaultdict(float))
    disp = defaultdict(set)
    names = []
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"][:60]
            c = r["Counter_Name"]
            if c not in names:
                names.append(c)
            tot[k][c] += float(r["Counter_Value"])
            disp[k].add(r["Dispatch_Id"])
    print("| kernel | dispatches | " + " | ".join(names) + " |")
    print("|---|---:|" + "---:|" * len(names))
    for k in sorted(tot, key=lambda k: -tot[k].get("SQ_WAVE_CYCLES", 0)):
        if tot[k].get("SQ_WAVE_CYCLES", 0) < 1e8:
            continue
        print(f"| `{k}` | {len(disp[k])} | " + " | ".join(f"{tot[k].get(c, 0) / 1e9:.2f} G" for c in names) + " |")


if __name__ == "__main__":
    main(sys.argv[1])
