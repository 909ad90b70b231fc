#!/bin/bash
# VALU issue-rate calibration on the GPU box (from the repo root): tools/valu_calib.sh <tag>
#   1. the timed pass of tools/valu_issue_calib.hip                                   -> gpurun_out/valu_issue_<tag>.md
#   2. one rocprofv3 counter pass (counters only) of the same program, v_fma_f32 / v_pk_fma_f32 / v_fma_f64 rows
#      (SQ_INSTS_VALU, SQ_ACTIVE_INST_VALU, SQ_BUSY_CYCLES, SQ_WAVE_CYCLES, GRBM_GUI_ACTIVE)  -> gpurun_out/valu_issue_pmc_<tag>.md
# Copy both into profiles/ (r5_valu_issue.md).
tag=${1:-x}
root=$PWD
out=$root/gpurun_out
mkdir -p $out
hipcc -O2 --offload-arch=gfx950 $root/tools/valu_issue_calib.hip -o $out/valu_issue_calib || exit 1
$out/valu_issue_calib > $out/valu_issue_$tag.md 2> $out/valu_issue_$tag.err
cd /tmp && export TMPDIR=/tmp
for op in v_fma_f32 v_pk_fma_f32 v_fma_f64 v_add_u32; do
  timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv \
    -d $out/valu_pmc_${tag}_$op -- $out/valu_issue_calib $op > $out/valu_pmc_${tag}_$op.log 2>&1
done
cd $root
python tools/valu_pmc_summary.py $out $tag > $out/valu_issue_pmc_$tag.md
rm -rf $out/valu_pmc_${tag}_v_*/ $out/valu_issue_calib
cat $out/valu_issue_$tag.md; cat $out/valu_issue_pmc_$tag.md
