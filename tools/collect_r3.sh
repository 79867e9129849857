#!/bin/bash
# Everything the profiles/r3_* files are made from, in one go on a fresh GPU box (run from the repo root).
set -x
part=${1:-all}
if [ $part = all ] || [ $part = a ]; then
tools/collect_profiles.sh r3 || exit 1
python3 tools/make_profile_summary.py r3 gpurun_out/r3/bench.json $(find gpurun_out/r3/prof -name '*kernel_stats.csv' | head -1) \
   $(find gpurun_out/r3/pmc_fetch -name '*counter_collection.csv' | head -1) $(find gpurun_out/r3/pmc_write -name '*counter_collection.csv' | head -1) > gpurun_out/r3/summary.txt 2>&1 || exit 1
tools/collect_config_profiles.sh r3_eval_penalty --trie Arts --beams 50 --batch 160 --prompt-len 253 --repetition-penalty 1.1 --steps 100 --warmup 20 > gpurun_out/r3_configs.txt 2>&1 || exit 1
tools/collect_config_profiles.sh r3_eval_plain --trie Arts --beams 50 --batch 160 --prompt-len 253 --steps 100 --warmup 20 >> gpurun_out/r3_configs.txt 2>&1 || exit 1
tools/configs_r3.sh >> gpurun_out/r3_configs.txt 2>&1 || exit 1
SURFACE=1 tools/collect_config_profiles.sh r3_c2_dense_surface --steps 20 --warmup 5 >> gpurun_out/r3_configs.txt 2>&1 || exit 1
tools/collect_config_profiles.sh r3_rl_64rows --batch 4 --beams 16 --steps 300 --warmup 30 >> gpurun_out/r3_configs.txt 2>&1 || exit 1
tools/collect_config_profiles.sh r3_eval_400rows --batch 8 --beams 50 --trie Arts --repetition-penalty 1.1 --prompt-len 253 --steps 300 --warmup 30 >> gpurun_out/r3_configs.txt 2>&1 || exit 1
fi
if [ $part = all ] || [ $part = b ]; then
tools/small_shapes.sh > gpurun_out/r3_small_shapes.txt 2>&1 || exit 1
for a in "--batch 4 --beams 16" "--batch 4 --beams 16 --no-fused-step" "--batch 8 --beams 50 --trie Arts --repetition-penalty 1.1 --prompt-len 253"; do
  timeout -k 10 200 python3 tools/host_step_cost.py $a 2>&1 | grep "^rows" >> gpurun_out/r3_small_shapes.txt || exit 1
done
GRX_FUSE_MAX_ROWS=1000000 tools/ab_fused.sh 3 > gpurun_out/r3_fused_ab.txt 2>&1 || exit 1
echo "== C3 (Games, fixed hint, 32768 rows)" >> gpurun_out/r3_fused_ab.txt
GRX_FUSE_MAX_ROWS=1000000 tools/ab_fused.sh 2 --trie Games --processor fixed_hint --batch 2048 --steps 50 --warmup 10 >> gpurun_out/r3_fused_ab.txt 2>&1 || exit 1
echo "== codebook 512 (65536 rows)" >> gpurun_out/r3_fused_ab.txt
GRX_FUSE_MAX_ROWS=1000000 tools/ab_fused.sh 2 --trie cb512 --batch 4096 --steps 40 --warmup 10 >> gpurun_out/r3_fused_ab.txt 2>&1 || exit 1
echo "== row counts (16 beams)" >> gpurun_out/r3_fused_ab.txt
tools/ab_fuse_rows.sh 4 16 32 64 128 >> gpurun_out/r3_fused_ab.txt 2>&1 || exit 1
timeout -k 10 300 python3 tools/lm_head_fused_probe.py > gpurun_out/r3_lm_head.txt 2>&1 || exit 1
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r3_lmh -o t -- python3 $GRAFT_REPO_ROOT/tools/lm_head_fused_probe.py 2048 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
cp $(find gpurun_out/r3_lmh -name '*kernel_stats.csv' | head -1) profiles/r3_lm_head_fused_kernel_stats.csv
tools/lm_head_pmc.sh > gpurun_out/r3_lm_head_pmc.txt 2>&1
fi
echo done
