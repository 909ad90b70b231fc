#!/bin/bash
# Issue / wait counters per kernel (one rocprofv3 --pmc pass of a 1-step bench run; counters only): tools/profile_sq.sh <tag>
#   -> gpurun_out/sq_<tag>.md (copy to profiles/<round>_sq_counters.md) + gpurun_out/sq_counters_<tag>.json (-> profiles/sq_counters_<workload>.json,
#      which bench.py quotes while the kernel sources are the ones it was measured on)
tag=${1:-x}
wl=${2:-s256}                       # workload (bench.py --workload): s256 (the metric), s128, s512
fields=$(python -c "print({'s128': 1600, 's256': 16000, 's512': 32000}['$wl'])")
root=$PWD
out=$root/gpurun_out
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
DAZIM_FMM_ASYNC=0 timeout 900 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT \
  --output-format csv -d $out/pmc_sq_$tag -- python $root/bench.py --workload $wl --no-cpu --steps 1 --warmup 0 > $out/pmc_sq_$tag.log 2>&1
cd $root
python tools/sq_summary.py $out/pmc_sq_$tag --json $out/sq_counters_$tag.json --tag $tag --workload $wl --fields $fields > $out/sq_$tag.md
cp $out/sq_counters_$tag.json $root/profiles/sq_counters_$wl.json
rm -rf $out/pmc_sq_$tag
head -12 $out/sq_$tag.md
