#!/bin/bash
# Round-robin bench runs of build variants at the two launcher shapes, HIP-graph mode: tools/ab_small.sh <rounds> lib...
rounds=$1; shift
for i in $(seq 1 $rounds); do
  for v in "$@"; do
    for shape in "--batch 4 --beams 16" "--batch 8 --beams 50 --trie Arts --repetition-penalty 1.1 --prompt-len 253"; do
      GENREC_BENCH_LIB=$PWD/build/variants/lib_$v.so timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-processor-surface --graph --steps 3000 --warmup 30 $shape > /tmp/ab_line.json || exit 1
      python3 - "$v" "$shape" <<'PY'
import json, sys
d = json.load(open("/tmp/ab_line.json"))
print(f"{sys.argv[1]:10s} {sys.argv[2][:22]:22s} {d['value']:9.1f} steps/s  {d['ms_per_step']*1e3:7.2f} us per step", flush=True)
PY
    done
  done
done
