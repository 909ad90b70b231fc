#!/bin/bash
# HBM-side traffic of the eikonal kernel alone (two counter passes, counters only): tools/pmc_fmm.sh [sources]
root=$PWD; out=$root/gpurun_out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmcf_$c
  timeout 600 rocprofv3 --pmc $c --output-format csv -d /tmp/pmcf_$c -- python $root/tools/fmm_only.py ${1:-1000} 1 > /tmp/pmcf_$c.log 2>&1
  f=$(find /tmp/pmcf_$c -name "*counter_collection.csv" | head -1)
  python - "$f" "$c" <<'PY'
import csv, sys, collections
tot = collections.defaultdict(float); n = collections.defaultdict(int)
for r in csv.DictReader(open(sys.argv[1])):
    if r["Counter_Name"] == sys.argv[2]:
        k = r["Kernel_Name"][:60]; tot[k] += float(r["Counter_Value"]); n[k] += 1
for k in sorted(tot, key=lambda k: -tot[k])[:4]:
    print(sys.argv[2], k, "dispatches", n[k], "per dispatch (raw KiB)", round(tot[k] / n[k]))
PY
done
