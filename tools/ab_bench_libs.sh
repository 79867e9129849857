#!/bin/bash
# Alternating whole-bench runs of library builds on one box: tools/ab_bench_libs.sh <pairs> <libA> <libB> -- <bench args>
# Prints steps/s and the stream kernel's time per run.  Run from the repo root on the GPU box.
pairs=$1; a=$2; b=$3; shift 4
for i in $(seq 1 $pairs); do
  for lib in $a $b; do
    GENREC_BENCH_LIB=$lib timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-processor-surface "$@" > /tmp/ab_line.json || exit 1
    python3 - "$lib" <<'PY'
import json, sys
d = json.load(open("/tmp/ab_line.json"))
r = d["roofline"]
print(f"{sys.argv[1]:40s} {d['value']:8.1f} steps/s  step {d['ms_per_step']*1e3:7.1f} us  kernel {r['kernel_ms']*1e3:7.1f} us  frac {r['frac']:.3f}  probe {r['read_probe']['GBps']:.0f}", flush=True)
PY
  done
done
