#!/usr/bin/env python
"""Per-kernel totals of one rocprofv3 --pmc counter from its CSV output (one counter per pass).

    python tools/pmc_summary.py gpurun_out/pmc_fetch gpurun_out/pmc_write [--json out.json --tag r2 --workload s256 --fields 16000]
prints a markdown table: kernel, dispatches, mean FETCH_SIZE and WRITE_SIZE per dispatch (raw counter units, KB).  With --json it
also writes the per-kernel means (KiB per dispatch) together with the hash of the kernel sources they were measured on: this is
profiles/pmc_traffic.json, the file bench.py takes `traffic` from (and refuses when the sources have changed since).
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


def write_json(dirs, path, tag, workload, fields):
    import json
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench import kernel_source_hash
    (_, fetch), (_, write) = load(dirs[0]), load(dirs[1])
    kernels = {}
    for k in set(fetch) | set(write):
        n = max(fetch[k][0] if k in fetch else 0, write[k][0] if k in write else 0)
        kernels[k[:120]] = {"dispatches": n,
                            "fetch_kib": fetch[k][1] / fetch[k][0] if k in fetch and fetch[k][0] else 0.0,
                            "write_kib": write[k][1] / write[k][0] if k in write and write[k][0] else 0.0}
    json.dump({"tag": tag, "workload": workload, "fields": fields, "source_sha": kernel_source_hash(),
               "unit": "KiB per dispatch, raw FETCH_SIZE / WRITE_SIZE (two separate rocprofv3 --pmc passes)",
               "kernels": kernels}, open(path, "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    argv = sys.argv[1:]
    opts = {}
    while len(argv) >= 2 and argv[-2].startswith("--"):
        opts[argv[-2][2:]] = argv[-1]
        argv = argv[:-2]
    main(argv)
    if "json" in opts:
        write_json(argv, opts["json"], opts.get("tag", "?"), opts.get("workload", "s256"), int(opts.get("fields", "16000")))
