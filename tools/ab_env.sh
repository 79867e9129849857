#!/bin/bash
# One environment knob of the library at several values, under a replayed graph at several shapes:
#   tools/ab_env.sh NAME "v1 v2 ..." "<bench args>" ["<bench args>" ...]
name=$1; vals=$2; shift 2
for args in "$@"; do
  echo "== $args"
  for rep in 1 2; do
    for v in $vals; do
      env $name=$v timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-processor-surface --steps 3000 --warmup 30 --graph $args > /tmp/sp_line.json || exit 1
      python3 - $name=$v <<'PY'
import json, sys
d = json.load(open("/tmp/sp_line.json"))
print(f"{sys.argv[1]:28s} {d['value']:9.1f} steps/s  {d['ms_per_step']*1e3:7.2f} us per step", flush=True)
PY
    done
  done
done
