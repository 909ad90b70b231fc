#!/bin/bash
# LDS bank conflicts of the eikonal kernel with and without the skewed heap rows (fmm.hip: KSKEW / NSKEW; -DDZ_FMM_SKEW; default = the
# round-5 layout): one counters-only pass each of tools/fmm_only.py (S-256, 16 000 fields), then three timed same-box pairs.
root=$PWD; out=$root/gpurun_out; mkdir -p $out
i=0
for f in "" "-DDZ_FMM_SKEW"; do
  export DAZIM_LIB=/tmp/libdazim_sk$i.so
  DAZIM_HIPCC_EXTRA="$f" python -c "import dazimsurftomo_amd as dz; dz.build(force=True)" > /dev/null 2>&1 || echo "build failed: $f"
  i=$((i+1))
done
cd /tmp && export TMPDIR=/tmp
for i in 0 1; do
  rm -rf /tmp/pmcsk_$i
  DAZIM_LIB=/tmp/libdazim_sk$i.so timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU SQ_ACTIVE_INST_VALU \
    --output-format csv -d /tmp/pmcsk_$i -- python $root/tools/fmm_only.py 1000 1 > /tmp/pmcsk_$i.log 2>&1
  f=$(find /tmp/pmcsk_$i -name "*counter_collection.csv" | head -1)
  python - "$f" "$i" <<'PY'
import csv, sys, collections
tot = collections.defaultdict(float)
for r in csv.DictReader(open(sys.argv[1])):
    if "fmm_kernel" in r["Kernel_Name"]:
        tot[r["Counter_Name"]] += float(r["Counter_Value"])
lab = ["plain rows (default)", "skewed rows (-DDZ_FMM_SKEW)"][int(sys.argv[2])]
print(lab, {k: f"{v/1e9:.2f} G" for k, v in sorted(tot.items())}, "conflict / LDS-active =", round(tot["SQ_LDS_BANK_CONFLICT"] / max(tot["SQ_ACTIVE_INST_LDS"], 1), 3))
PY
done
cd $root
for rep in 1 2 3 4; do
  for i in 0 1; do
    echo -n "[$i] "; DAZIM_LIB=/tmp/libdazim_sk$i.so python tools/fmm_only.py 1000 1 2>&1 | grep kernel | awk '{print $7, $8, $9}'
  done
done
