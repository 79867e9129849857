#!/bin/bash
# Kernel trace + separate FETCH_SIZE / WRITE_SIZE passes of one bench configuration.
#   tools/collect_config_profiles.sh <tag> <bench.py arguments...>
# leaves gpurun_out/<tag>/{stats,pmc_fetch,pmc_write}; tools/make_config_profile.py turns them into
# profiles/<tag>_kernel_stats.csv and profiles/<tag>_pmc.json.  Run on the GPU box from the repo root.
tag=$1; shift
R=$(pwd)
out=$R/gpurun_out/$tag
rm -rf $out && mkdir -p $out
common="--no-cpu-baseline --no-settle"
[ -z "$SURFACE" ] && common="$common --no-processor-surface"  # SURFACE=1: the dense [R, V] calls are traced as well
cd /tmp && export TMPDIR=/tmp
timeout -k 10 280 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o t -- \
    python3 $R/bench.py $common "$@" > $out/line.json 2> $out/stats.err || exit 1
small="--steps 10 --warmup 5 --no-kernel-timing"
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -o pmc -- \
    python3 $R/bench.py $common "$@" $small > $out/pmc_fetch.log 2>&1 || exit 1
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -o pmc -- \
    python3 $R/bench.py $common "$@" $small > $out/pmc_write.log 2>&1 || exit 1
cd $R
python3 tools/make_config_profile.py $tag "$@"
