#!/bin/bash
# The parity-test configurations other than the headline one, timed with the same bench.py
# (one JSON line each, appended to $1).  Run on the GPU box from the repo root.
set -e
out=${1:-gpurun_out/configs.jsonl}
: > "$out"
common="--no-cpu-baseline --no-processor-surface"
run() { echo "== $*" >&2; timeout -k 10 280 python3 bench.py $common "$@" >> "$out"; }
run --trie Industrial_and_Scientific --batch 64 --dtype fp32 --steps 200 --warmup 20          # C1
run --trie Games --processor fixed_hint --batch 2048 --steps 50 --warmup 10                  # C3
run --trie cb32 --batch 4096 --steps 40 --warmup 10                                           # C4, smallest codebook
run --trie cb512 --batch 4096 --steps 40 --warmup 10                                          # C4, largest codebook
run --trie Arts --beams 50 --batch 160 --steps 100 --warmup 20                                # evaluation-style K = 50
run --dtype fp32 --steps 100 --warmup 20                                                      # headline shape, fp32 logits
