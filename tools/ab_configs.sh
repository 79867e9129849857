#!/bin/bash
# A/B of a variant library against the shipped one over the bench configurations: alternating order per pair.
#   tools/ab_configs.sh build/variants/lib_x.so
other=$1
run() { label=$1; shift; bash tools/run_line.sh "$label" "$@" | cut -c1-48; }
cfgs=("" "--trie Industrial_and_Scientific --batch 64 --dtype fp32 --steps 200" "--trie Games --processor fixed_hint --batch 2048 --steps 50 --warmup 10" "--trie cb512 --batch 4096 --steps 40 --warmup 10" "--trie Arts --beams 50 --batch 160" "--dtype fp32" "--trie Arts --beams 50 --batch 160 --prompt-len 253 --repetition-penalty 1.1")
for c in "${cfgs[@]}"; do
  echo "== ${c:-headline}"
  run new $c && GENREC_BENCH_LIB=$other run old $c && GENREC_BENCH_LIB=$other run old $c && run new $c || exit 1
done
