#!/bin/bash
# Shader clock while the LM-head GEMM kernels run: GRBM_GUI_ACTIVE (cycles) over the dispatch's duration.
#   tools/lm_head_clock.sh [library path]
R=$(pwd); out=$R/gpurun_out/lmh_clock; rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
LH_LIB=${1:-} timeout -s KILL 120 rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d $out -o pmc -- python3 $R/tools/lm_head_fused_probe.py 2048 > $out/run.log 2>&1 || exit 1
cd $R
python3 - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/lmh_clock/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"]
    if ("lm_head_lse" in k or "Cijk" in k) and r["Counter_Name"] == "GRBM_GUI_ACTIVE":
        dur = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        acc[k[:30]].append((float(r["Counter_Value"]), dur))
for k, v in acc.items():
    v = v[len(v) // 2:]
    cyc = sum(c for c, _ in v) / len(v); dur = sum(d for _, d in v) / len(v)
    print(f"{k:30s} {cyc / 1e6:8.2f} M cycles in {dur / 1e6:6.3f} ms = {cyc / dur * 1e3:6.0f} MHz (n={len(v)})")
PY
