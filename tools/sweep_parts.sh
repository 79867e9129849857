#!/bin/bash
# Steps/s under a HIP graph for several workgroup budgets of split rows: tools/sweep_parts.sh "<bench args>" target ...
# (GRX_SPLIT_TARGET = workgroups per launch the split may use; parts per row = target / rows, at most GRX_SPLIT_PARTS.)
# SWEEP_LIBS="a b": the variant builds build/variants/lib_a.so, lib_b.so side by side (default: the shipped library).
args=$1; shift
for rep in 1 2; do
for t in "$@"; do
  for v in ${SWEEP_LIBS:-shipped}; do
    lib=""; [ $v != shipped ] && lib=$PWD/build/variants/lib_$v.so
    GENREC_BENCH_LIB=$lib GRX_SPLIT_TARGET=$t timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-processor-surface --steps 3000 --warmup 30 --graph $args > /tmp/sp_line.json || exit 1
    python3 - "$t" $v <<'PY'
import json, sys
d = json.load(open("/tmp/sp_line.json"))
print(f"target {sys.argv[1]:>6s} {sys.argv[2]:10s} {d['value']:9.1f} steps/s  {d['ms_per_step']*1e3:7.2f} us per step", flush=True)
PY
  done
done
done
