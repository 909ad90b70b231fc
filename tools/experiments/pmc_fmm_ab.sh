#!/bin/bash
# counters of the eikonal kernel alone for several prebuilt libraries: tools/pmc_fmm_ab.sh <tag> <libA.so|-> <libB.so|-> ...
#   one rocprofv3 --pmc pass each (counters only) of tools/fmm_only.py ${SRC:-1000} 1  ->  gpurun_out/pmcab_<tag>.md
tag=$1; shift
root=$PWD; out=$root/gpurun_out; mkdir -p $out
export TMPDIR=/tmp
: > $out/pmcab_$tag.md
i=0
for l in "$@"; do
  if [ "$l" = "-" ]; then unset DAZIM_LIB; else export DAZIM_LIB=$root/$l; fi
  d=/tmp/pmcab_$i; rm -rf $d
  (cd /tmp && timeout 600 rocprofv3 --pmc ${CTRS:-SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE} \
     --output-format csv -d $d -- python $root/tools/fmm_only.py ${SRC:-1000} 1 > $d.log 2>&1)
  echo "## $l" >> $out/pmcab_$tag.md
  grep kernel $d.log | tail -1 >> $out/pmcab_$tag.md
  python $root/tools/sq_summary.py $d | grep -E "kernel \||fmm_kernel|---" >> $out/pmcab_$tag.md
  i=$((i+1))
done
cat $out/pmcab_$tag.md
