#!/bin/bash
# Steps/s for several split settings: tools/sweep_split.sh "<bench args>" "rows:target" ...
args=$1; shift
for st in "$@"; do
  GRX_SPLIT_ROWS=${st%%:*} GRX_SPLIT_TARGET=${st##*:} timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-processor-surface --steps 200 --warmup 20 $args > /tmp/ab_line.json || exit 1
  python3 - "$st" <<'PY'
import json, sys
d = json.load(open("/tmp/ab_line.json"))
r = d["roofline"]
print(f"split rows:target {sys.argv[1]:>10s}  {d['value']:8.1f} steps/s  step {d['ms_per_step']*1e3:7.1f} us  kernel {r['kernel_ms']*1e3:7.1f} us", flush=True)
PY
done
