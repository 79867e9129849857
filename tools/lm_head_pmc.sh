#!/bin/bash
# Where the LM-head kernel's cycles go, next to the library GEMM's: one --pmc pass over tools/lm_head_fused_probe.py 2048
R=$(pwd); out=$R/gpurun_out/lmh_pmc; rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
LH_ROUNDS=1 timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE \
  --output-format csv -d $out -o pmc -- python3 $R/tools/lm_head_fused_probe.py 2048 > $out/run.log 2>&1 || exit 1
cd $R
python3 - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/lmh_pmc/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"]
    if "lm_head_lse" in k or "Cijk" in k:
        acc[k[:40]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print(k)
    for c, v in d.items():
        print(f"   {c:28s} {sum(v)/len(v):16.0f}  (n={len(v)})")
PY
