#!/bin/bash
# Per-depth kernel times of one bench configuration from a rocprofv3 kernel trace: tools/trace_shape.sh <tag> <bench args>
tag=$1; shift
R=$(pwd)
out=$R/gpurun_out/trace_$tag
rm -rf $out && mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $out -o t -- python3 $R/bench.py --steps 100 --no-cpu-baseline \
   --no-processor-surface --no-settle --no-kernel-timing "$@" > $out.log 2>&1 || exit 1
python3 $R/tools/trace_depths.py $(find $out -name '*kernel_trace.csv' | head -1) 100
