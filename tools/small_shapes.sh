#!/bin/bash
# Eager against HIP-graph steps/s at the shapes the reference's launchers use, alternating, one box.
# Run from the repo root on the GPU box; prints one line per run.
line() {
  python3 - "$@" <<'PY'
import json, sys
d = json.load(open("/tmp/ss_line.json"))
print(f"{sys.argv[1]:10s} {sys.argv[2]:6s} {d['value']:9.1f} steps/s  {d['ms_per_step']*1e3:7.1f} us per step", flush=True)
PY
}
for i in 1 2 3; do
  for mode in eager graph; do
    flag=""; [ $mode = graph ] && flag="--graph"
    timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-processor-surface --batch 4 --beams 16 --steps 3000 --warmup 30 $flag > /tmp/ss_line.json || exit 1
    line "RL 4x16" $mode
    timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-processor-surface --batch 8 --beams 50 --trie Arts --repetition-penalty 1.1 --prompt-len 253 --steps 3000 --warmup 30 $flag > /tmp/ss_line.json || exit 1
    line "eval 8x50" $mode
  done
done
