#!/bin/bash
# Per-depth kernel times of the evaluation shape (Arts, 50 beams x 160 prompts, 256-token rows) with and
# without the repetition penalty, from rocprofv3 kernel traces.  Run from the repo root on the GPU box.
R=$(pwd)
out=$R/gpurun_out/eval_depths
rm -rf $out && mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for pen in ${PENS:-1.1 1.0}; do
  timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $out/p$pen -o t -- python3 $R/bench.py --trie Arts --beams 50 \
    --batch 160 --prompt-len 253 --repetition-penalty $pen --steps 100 --no-cpu-baseline --no-processor-surface --no-settle \
    --no-kernel-timing "$@" > $out/p$pen.log 2>&1 || exit 1
  f=$(find $out/p$pen -name '*kernel_trace.csv' | head -1)
  echo "== penalty $pen"
  python3 $R/tools/trace_depths.py $f 100
done
