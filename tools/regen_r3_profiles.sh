#!/bin/bash
# profiles/r3_* from what tools/collect_r3.sh a left under gpurun_out/ (the GPU box only sends gpurun_out/ back, so the
# files under profiles/ are made again here, from the same inputs and by the same scripts).  Run from the repo root.
set -e
python3 tools/make_profile_summary.py r3 gpurun_out/r3/bench.json $(find gpurun_out/r3/prof -name '*kernel_stats.csv' | head -1) \
   $(find gpurun_out/r3/pmc_fetch -name '*counter_collection.csv' | head -1) $(find gpurun_out/r3/pmc_write -name '*counter_collection.csv' | head -1) | tail -3
m="python3 tools/make_config_profile.py"
$m r3_eval_penalty --trie Arts --beams 50 --batch 160 --prompt-len 253 --repetition-penalty 1.1 --steps 100 --warmup 20
$m r3_eval_plain --trie Arts --beams 50 --batch 160 --prompt-len 253 --steps 100 --warmup 20
$m r3_c1 --trie Industrial_and_Scientific --batch 64 --dtype fp32 --steps 200 --warmup 20
$m r3_c3 --trie Games --processor fixed_hint --batch 2048 --steps 50 --warmup 10
$m r3_c4_cb512 --trie cb512 --batch 4096 --steps 40 --warmup 10
$m r3_arts_k50 --trie Arts --beams 50 --batch 160 --steps 100 --warmup 20
$m r3_c2_fp32 --dtype fp32 --steps 100 --warmup 20
SURFACE=1 $m r3_c2_dense_surface --steps 20 --warmup 5
$m r3_rl_64rows --batch 4 --beams 16 --steps 300 --warmup 30
$m r3_eval_400rows --batch 8 --beams 50 --trie Arts --repetition-penalty 1.1 --prompt-len 253 --steps 300 --warmup 30
