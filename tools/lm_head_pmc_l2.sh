#!/bin/bash
# L2 behaviour of the LM-head kernel next to the library GEMM's: one --pmc pass (4 TCC counters) over tools/lm_head_fused_probe.py 2048
R=$(pwd); out=$R/gpurun_out/lmh_pmc_l2; rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
LH_ROUNDS=1 timeout -s KILL 120 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum \
  --output-format csv -d $out -o pmc -- python3 $R/tools/lm_head_fused_probe.py 2048 > $out/run.log 2>&1 || exit 1
cd $R
python3 - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/lmh_pmc_l2/**/*counter_collection.csv", recursive=True)[0]
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
