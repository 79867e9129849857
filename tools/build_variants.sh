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
build nt "-DMLS_NT=1"
build nt_t512 "-DMLS_NT=1 -DMLS_THREADS=512"
build nt_u2 "-DMLS_NT=1 -DMLS_UNROLL=2"
build buf0 "-DMLS_BUF=1 -DMLS_AUX=0"
build buf_nt "-DMLS_BUF=1 -DMLS_AUX=2"
build buf_sc0 "-DMLS_BUF=1 -DMLS_AUX=1"
build buf_sc1 "-DMLS_BUF=1 -DMLS_AUX=16"
build buf_sc1nt "-DMLS_BUF=1 -DMLS_AUX=18"
build buf_sc0nt "-DMLS_BUF=1 -DMLS_AUX=3"
build buf_all "-DMLS_BUF=1 -DMLS_AUX=19"
wait
ls -la build/variants
