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
build ntst "-DMLS_NTST=1"
build wg16 "-DMLS_WG_PER_CU=16"
build wg32 "-DMLS_WG_PER_CU=32"
build wg6 "-DMLS_WG_PER_CU=6"
build ntst_wg16 "-DMLS_NTST=1 -DMLS_WG_PER_CU=16"
wait
ls -la build/variants
