#!/bin/bash
# Build tuning variants of libgenrec_hip.so into build/variants/ (not shipped; A/B timing only).
#   tools/build_variants.sh name1 "flags1" name2 "flags2" ...
set -e
cd "$(dirname "$0")/.."
mkdir -p build/variants
SRC="genrec_amd/csrc/trie_state.hip genrec_amd/csrc/masked_logsoftmax.hip genrec_amd/csrc/chunked_lse.hip genrec_amd/csrc/beam_step.hip genrec_amd/csrc/scoring.hip genrec_amd/csrc/exchange.hip genrec_amd/csrc/lm_head.hip"
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Iinclude -Igenrec_amd/csrc -fvisibility=hidden -DGRX_BUILDING -shared"
while [ $# -ge 2 ]; do
  /opt/rocm/bin/hipcc $FLAGS $2 $SRC -o build/variants/lib_$1.so &
  shift 2
done
wait
ls -la build/variants
