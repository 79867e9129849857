#!/bin/bash
# rocprofv3 kernel trace + FETCH_SIZE / WRITE_SIZE passes for the configurations other than the headline one.
# Run on the GPU box from the repo root; leaves profiles/r3_<config>_{kernel_stats.csv,pmc.json}.
set -e
c=tools/collect_config_profiles.sh
$c r3_c1 --trie Industrial_and_Scientific --batch 64 --dtype fp32 --steps 200 --warmup 20
$c r3_c3 --trie Games --processor fixed_hint --batch 2048 --steps 50 --warmup 10
$c r3_c4_cb512 --trie cb512 --batch 4096 --steps 40 --warmup 10
$c r3_arts_k50 --trie Arts --beams 50 --batch 160 --steps 100 --warmup 20
$c r3_c2_fp32 --dtype fp32 --steps 100 --warmup 20
