#!/bin/bash
# Steps/s of the one-launch step against the two kernels for several row counts (16 beams, headline trie).
for b in "$@"; do
  for lim in 1000000 0; do
    GRX_FUSE_MAX_ROWS=$lim timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-processor-surface --batch $b --steps 200 --warmup 20 > /tmp/ab_line.json || exit 1
    python3 - "$b" "$lim" <<'PY'
import json, sys
d = json.load(open("/tmp/ab_line.json"))
r = d["roofline"]
print(f"groups {sys.argv[1]:>5s} {'one launch ' if sys.argv[2] != '0' else 'two kernels'} {d['value']:8.1f} steps/s  step {d['ms_per_step']*1e3:7.1f} us  kernel {r['kernel_ms']*1e3:7.1f} us", flush=True)
PY
  done
done
