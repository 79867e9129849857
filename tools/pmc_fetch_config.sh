#!/bin/bash
# FETCH_SIZE of the streaming kernel for one bench configuration (arguments are passed on to bench.py):
# prints launches, average KiB fetched per launch (x2 = bytes on gfx950 for wide coalesced streams).
set -e
R=$(pwd)
out=$R/gpurun_out/pmc_cfg
rm -rf $out && mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out -o pmc -- python3 $R/bench.py --steps 10 --warmup 5 \
    --no-cpu-baseline --no-processor-surface --no-kernel-timing --no-settle "$@" > $out/run.log 2>&1
cd $R
python3 - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/pmc_cfg/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if r["Counter_Name"] == "FETCH_SIZE" and "masked_lsm" in r["Kernel_Name"]:
        acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    print(k[:60], len(v), "launches, avg KiB", sum(v) / len(v), "-> bytes x2:", sum(v) / len(v) * 2048)
PY
