#!/bin/bash
# Build tuning variants of libgenrec_hip.so into build/variants/ (not shipped; A/B timing only).
set -e
cd "$(dirname "$0")/.."
mkdir -p build/variants
SRC="genrec_amd/csrc/trie_state.hip genrec_amd/csrc/masked_logsoftmax.hip genrec_amd/csrc/beam_step.hip genrec_amd/csrc/scoring.hip"
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Iinclude -Igenrec_amd/csrc -fvisibility=hidden -DGRX_BUILDING -shared"
build() { # name, extra flags
  /opt/rocm/bin/hipcc $FLAGS $2 $SRC -o build/variants/lib_$1.so &
}
build base ""
build ntfill "-DMLS_NTFILL=1"
build fpt4 "-DFILL_PER_THREAD=4"
build fpt16nt "-DFILL_PER_THREAD=16 -DMLS_NTFILL=1"
wait
ls -la build/variants
