#!/bin/bash
# Variant builds side by side under a replayed graph at several shapes, default split rule:
#   SWEEP_LIBS="a b" tools/ab_rows.sh "<bench args>" ["<bench args>" ...]
for args in "$@"; do
  echo "== $args"
  for rep in 1 2; do
    for v in ${SWEEP_LIBS:-shipped}; do
      lib=""; [ $v != shipped ] && lib=$PWD/build/variants/lib_$v.so
      GENREC_BENCH_LIB=$lib timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-processor-surface --steps 3000 --warmup 30 --graph $args > /tmp/sp_line.json || exit 1
      python3 - $v <<'PY'
import json, sys
d = json.load(open("/tmp/sp_line.json"))
print(f"{sys.argv[1]:10s} {d['value']:9.1f} steps/s  {d['ms_per_step']*1e3:7.2f} us per step", flush=True)
PY
    done
  done
done
