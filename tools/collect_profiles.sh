#!/bin/bash
# Run on the GPU box from the repo root: the bench line, the kernel trace and the two counter passes that
# tools/make_profile_summary.py turns into profiles/<tag>_*.  Usage: tools/collect_profiles.sh <tag>
set -e
tag=${1:-rX}
R=$(pwd)
out=$R/gpurun_out/$tag
mkdir -p $out
timeout -k 10 400 python3 bench.py > $out/bench.json 2> $out/bench.err
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -o bench -- \
    python3 $R/bench.py --no-cpu-baseline --no-processor-surface --no-settle > $out/prof.log 2>&1
small="--steps 10 --warmup 5 --no-cpu-baseline --no-processor-surface --no-kernel-timing --no-settle"
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -o pmc -- python3 $R/bench.py $small > $out/pmc_fetch.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -o pmc -- python3 $R/bench.py $small > $out/pmc_write.log 2>&1
cd $R
ls $out $out/prof $out/pmc_fetch
