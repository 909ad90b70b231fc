#!/usr/bin/env python
"""Per-kernel sums of several rocprofv3 --pmc counters collected in ONE pass (CSV output).

    python tools/sq_summary.py gpurun_out/pmc_sq [--json out.json --tag r4 --workload s256 --fields 16000]
prints a markdown table: kernel, dispatches, then one column per counter (sum over dispatches).  With --json it also writes the
per-kernel sums per dispatch with the hash of the kernel sources they were measured on: profiles/sq_counters.json, the file
bench.py takes the eikonal kernel's VALU instruction count from (and refuses when the sources have changed since)."""
import csv
import glob
import os
import sys
from collections import defaultdict


def main(d, js=None, tag="?", workload="s256", fields=16000):
    tot = defaultdict(lambda: defaultdict(float))
    disp = defaultdict(set)
    names = []
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"][:120]
            c = r["Counter_Name"]
            if c not in names:
                names.append(c)
            tot[k][c] += float(r["Counter_Value"])
            disp[k].add(r["Dispatch_Id"])
    print("| kernel | dispatches | " + " | ".join(names) + " |")
    print("|---|---:|" + "---:|" * len(names))
    for k in sorted(tot, key=lambda k: -tot[k].get("SQ_WAVE_CYCLES", 0)):
        if tot[k].get("SQ_WAVE_CYCLES", 0) < 1e8 and "disp" not in k:
            continue
        print(f"| `{k[:60]}` | {len(disp[k])} | " + " | ".join(f"{tot[k].get(c, 0) / 1e9:.2f} G" for c in names) + " |")


    if js:
        import json
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        from bench import kernel_source_hash
        json.dump({"tag": tag, "workload": workload, "fields": fields, "source_sha": kernel_source_hash(),
                   "unit": "counter sums per dispatch (one rocprofv3 --pmc pass, counters only); SQ_ACTIVE_* / SQ_WAVE_CYCLES / "
                           "SQ_WAIT_* count quad-cycles",
                   "kernels": {k: {"dispatches": len(disp[k]), **{c: tot[k][c] / len(disp[k]) for c in names}}
                               for k in tot if tot[k].get("SQ_WAVE_CYCLES", 0) >= 1e8}}, open(js, "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    argv = sys.argv[1:]
    opts = {}
    while len(argv) >= 2 and argv[-2].startswith("--"):
        opts[argv[-2][2:]] = argv[-1]
        argv = argv[:-2]
    main(argv[0], opts.get("json"), opts.get("tag", "?"), opts.get("workload", "s256"), int(opts.get("fields", "16000")))
