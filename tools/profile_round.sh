#!/bin/bash
# The measurement passes behind profiles/ (run on the GPU box from the repo root): tools/profile_round.sh <tag>
#   (counter passes run with DAZIM_FMM_ASYNC=0: one kernel at a time, per-kernel counters; the trace and the bench line with the default)
#   1. two counter passes (FETCH_SIZE, WRITE_SIZE; counters only)  -> gpurun_out/pmc_<tag>.md + profiles/pmc_traffic_<workload>.json
#      (bench.py quotes `traffic` from that json while the kernel sources are the ones it was measured on)
#   2. rocprofv3 --kernel-trace --stats of a 3-step run            -> gpurun_out/kstats_<tag>.md
#   3. the default bench.py run (with the CPU baseline)            -> gpurun_out/bench_<tag>.json
# Copy kstats_<tag>.md, pmc_<tag>.md, pmc_traffic_<tag>.json (as profiles/pmc_traffic_<workload>.json) and bench_<tag>.json into profiles/.
tag=${1:-x}
wl=${2:-s256}                       # workload (bench.py --workload): s256 (the metric), s128, s512
fields=$(python -c "print({'s128': 1600, 's256': 16000, 's512': 32000}['$wl'])")
root=$PWD
out=$root/gpurun_out
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
DAZIM_FMM_ASYNC=0 timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch_$tag -- python $root/bench.py --workload $wl --no-cpu --steps 1 --warmup 0 > $out/pmc_fetch_$tag.log 2>&1
DAZIM_FMM_ASYNC=0 timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_write_$tag -- python $root/bench.py --workload $wl --no-cpu --steps 1 --warmup 0 > $out/pmc_write_$tag.log 2>&1
cd $root
python tools/pmc_summary.py $out/pmc_fetch_$tag $out/pmc_write_$tag --json $out/pmc_traffic_$tag.json --tag $tag --workload $wl --fields $fields > $out/pmc_$tag.md
cp $out/pmc_traffic_$tag.json $root/profiles/pmc_traffic_$wl.json
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $out/prof_$tag -o bench -- python $root/bench.py --workload $wl --no-cpu --steps 3 --warmup 1 > $out/prof_$tag.log 2>&1
timeout 900 python $root/bench.py --workload $wl 2> $out/bench_$tag.log | tail -1 > $out/bench_$tag.json
cd $root
python tools/rocpd_summary.py $(find $out/prof_$tag -name "*results.db" | head -1) > $out/kstats_$tag.md
# keep the merged-back scratch small: the raw traces stay on the box
rm -rf $out/prof_$tag $out/pmc_fetch_$tag $out/pmc_write_$tag
tail -1 $out/bench_$tag.json | head -c 900; echo; head -14 $out/kstats_$tag.md; head -8 $out/pmc_$tag.md
