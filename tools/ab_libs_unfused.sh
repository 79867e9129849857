#!/bin/bash
# tools/ab_libs_unfused.sh <rounds> <bench args> -- lib names...: whole-bench runs of build variants, round robin
rounds=$1; shift
args=()
while [ "$1" != "--" ]; do args+=("$1"); shift; done
shift
for i in $(seq 1 $rounds); do
  for v in "$@"; do
    GENREC_BENCH_LIB=$PWD/build/variants/lib_$v.so timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-processor-surface "${args[@]}" > /tmp/ab_line.json || exit 1
    python3 - "$v" <<'PY'
import json, sys
d = json.load(open("/tmp/ab_line.json"))
r = d["roofline"]
print(f"{sys.argv[1]:18s} {d['value']:8.1f} steps/s  step {d['ms_per_step']*1e3:7.1f} us  kernel {r['kernel_ms']*1e3:7.1f} us  probe {r['read_probe']['GBps']:.0f}", flush=True)
PY
  done
done
