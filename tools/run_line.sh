#!/bin/bash
# one short bench run, printed as: <label> steps/s kernel_us probe_us
label=$1; shift
timeout -k 10 120 python3 bench.py --no-cpu-baseline --no-processor-surface "$@" | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$label', round(d['value'],1), round(d['roofline']['kernel_ms']*1e3,1), round(d['roofline']['read_probe']['ms']*1e3,1), d['roofline'].get('settle'))"
