#!/bin/bash
# The measurement passes behind profiles/ (run on the GPU box from the repo root): tools/profile_round.sh <tag>
#   1. the default bench.py run (with the CPU baseline)            -> gpurun_out/bench_<tag>.json
#   2. rocprofv3 --kernel-trace --stats of a 3-step run            -> gpurun_out/kstats_<tag>.md
#   3. two counter passes (FETCH_SIZE, WRITE_SIZE; counters only)  -> gpurun_out/pmc_<tag>.md
tag=${1:-x}
root=$PWD
out=$root/gpurun_out
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout 900 python $root/bench.py 2> $out/bench_$tag.log | tail -1 > $out/bench_$tag.json
timeout 600 rocprofv3 --kernel-trace --stats -d $out/prof_$tag -o bench -- python $root/bench.py --no-cpu --steps 3 --warmup 1 > $out/prof_$tag.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch_$tag -- python $root/bench.py --no-cpu --steps 1 --warmup 0 > $out/pmc_fetch_$tag.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_write_$tag -- python $root/bench.py --no-cpu --steps 1 --warmup 0 > $out/pmc_write_$tag.log 2>&1
cd $root
python tools/rocpd_summary.py $(find $out/prof_$tag -name "*results.db" | head -1) > $out/kstats_$tag.md
python tools/pmc_summary.py $out/pmc_fetch_$tag $out/pmc_write_$tag > $out/pmc_$tag.md
# keep the merged-back scratch small: the raw traces stay on the box
rm -rf $out/prof_$tag $out/pmc_fetch_$tag $out/pmc_write_$tag
tail -1 $out/bench_$tag.json | head -c 600; echo; head -12 $out/kstats_$tag.md; head -8 $out/pmc_$tag.md
