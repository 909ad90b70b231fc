#!/bin/bash
# After `tools/profile_sq.sh <tag>_<wl> <wl>; tools/profile_round.sh <tag>_<wl> <wl>` for wl in s256 s128 s512 (on the GPU box) and the
# merge of gpurun_out/: assemble the round's summaries under profiles/.   tools/install_profiles.sh <tag> <round prefix, e.g. r5>
tag=$1; rp=$2; o=gpurun_out; p=profiles
for w in s256 s128 s512; do
  cp $o/sq_counters_${tag}_$w.json $p/sq_counters_$w.json
  cp $o/pmc_traffic_${tag}_$w.json $p/pmc_traffic_$w.json
done
cp $o/bench_${tag}_s256.json $p/${rp}_bench.json
cp $o/bench_${tag}_s128.json $p/${rp}_bench_s128.json
cp $o/bench_${tag}_s512.json $p/${rp}_bench_s512.json
{ echo "# SQ counters of the S-256, S-128 and S-512 bench steps (one counters-only rocprofv3 pass each: tools/profile_sq.sh)"; for w in s256 s128 s512; do echo; echo "## $w"; echo; grep "^|" $o/sq_${tag}_$w.md; done; } > $p/${rp}_sq_counters.md
{ echo "# HBM-side traffic of the bench steps (FETCH_SIZE / WRITE_SIZE, separate counter passes; tools/profile_round.sh)"; for w in s256 s128 s512; do echo; echo "## $w"; echo; tail -n +2 $o/pmc_${tag}_$w.md; done; } > $p/${rp}_pmc_hbm_traffic.md
{ for w in s256 s128 s512; do echo "## $w"; echo; cat $o/kstats_${tag}_$w.md; echo; done; } > $p/${rp}_bench_kernel_stats.md
