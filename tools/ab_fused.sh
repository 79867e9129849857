#!/bin/bash
# Alternating whole-bench pairs on one box: the decode step as one launch (default) against the two kernels
# (--no-fused-step).  tools/ab_fused.sh <pairs> <bench.py arguments...>; run from the repo root on the GPU box.
pairs=$1; shift
for i in $(seq 1 $pairs); do
  for mode in "" "--no-fused-step"; do
    timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-processor-surface $mode "$@" > /tmp/ab_line.json || exit 1
    python3 - "${mode:-fused}" <<'PY'
import json, sys
d = json.load(open("/tmp/ab_line.json"))
r = d["roofline"]
print(f"{sys.argv[1]:18s} {d['value']:8.1f} steps/s  step {d['ms_per_step']*1e3:7.1f} us  kernel {r['kernel_ms']*1e3:7.1f} us  frac {r['frac']:.3f}  step_frac {r['step_frac']:.3f}  probe {r['read_probe']['GBps']:.0f}", flush=True)
PY
  done
done
