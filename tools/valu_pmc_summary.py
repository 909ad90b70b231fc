#!/usr/bin/env python
"""Counter pass of tools/valu_issue_calib.hip (tools/valu_calib.sh): per dispatch of the timed launches, the SQ counters beside the
instruction count the program is known to issue.

    python tools/valu_pmc_summary.py gpurun_out <tag>
Each directory gpurun_out/valu_pmc_<tag>_<op>/ holds one rocprofv3 --pmc CSV pass of `valu_issue_calib <op>`: 8 dispatches (a warm-up
and a timed launch for W = 1..4 wavefronts per SIMD).  SQ_ACTIVE_INST_VALU and SQ_WAVE_CYCLES / SQ_BUSY_CYCLES count quad-cycles
(4 shader cycles) summed over the SQs; GRBM_GUI_ACTIVE counts shader cycles of the launch."""
import csv
import glob
import os
import sys
from collections import defaultdict


def main(root, tag):
    print("| instruction | W | SQ_INSTS_VALU | SQ_ACTIVE_INST_VALU (quad-cycles) | active quad-cycles per instruction | SQ_WAVE_CYCLES | "
          "SQ_BUSY_CYCLES | GRBM_GUI_ACTIVE (cycles) | instructions per GUI cycle per SIMD (1 024 SIMDs) |")
    print("|---|---|---:|---:|---:|---:|---:|---:|---:|")
    for d in sorted(glob.glob(os.path.join(root, f"valu_pmc_{tag}_*"))):
        if not os.path.isdir(d):
            continue
        op = os.path.basename(d)[len(f"valu_pmc_{tag}_"):]
        per = defaultdict(dict)
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                per[int(r["Dispatch_Id"])][r["Counter_Name"]] = per[int(r["Dispatch_Id"])].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        ids = sorted(per)
        timed = ids[1::2]           # warm-up, timed, warm-up, timed ...
        for w, i in enumerate(timed, 1):
            c = per[i]
            n, a = c.get("SQ_INSTS_VALU", 0), c.get("SQ_ACTIVE_INST_VALU", 0)
            gui = c.get("GRBM_GUI_ACTIVE", 0)
            print(f"| `{op}` | {w} | {n:.4g} | {a:.4g} | {a / n if n else 0:.3f} | {c.get('SQ_WAVE_CYCLES', 0):.4g} | "
                  f"{c.get('SQ_BUSY_CYCLES', 0):.4g} | {gui:.4g} | {n / gui / 1024 if gui else 0:.3f} |")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
