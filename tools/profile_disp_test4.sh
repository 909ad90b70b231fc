#!/bin/bash
# SQ counters of the dispersion kernel on test4_Yunnan's model (one counters-only rocprofv3 pass): tools/profile_disp_test4.sh <tag>
#   -> gpurun_out/disp_test4_<tag>.md  (copy to profiles/<round>_disp_test4.md)
tag=${1:-x}
root=$PWD
out=$root/gpurun_out
mkdir -p $out
python tools/disp_test4.py 3 > $out/disp_test4_time_$tag.log 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_SCA \
  --output-format csv -d $out/pmc_disp4_$tag -- python $root/tools/disp_test4.py 1 > $out/pmc_disp4_$tag.log 2>&1
cd $root
(echo "# Dispersion kernel on test4_Yunnan's model (1 596 columns x 109 curves x 36 periods, 86 layers): time and SQ counters ($tag)"; echo; echo '```'; cat $out/disp_test4_time_$tag.log; echo '```'; echo; python tools/sq_summary.py $out/pmc_disp4_$tag) > $out/disp_test4_$tag.md
rm -rf $out/pmc_disp4_$tag
cat $out/disp_test4_$tag.md
