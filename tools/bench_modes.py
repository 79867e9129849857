#!/usr/bin/env python3
"""Streaming-kernel timing across output modes, element types and row counts.

Uses the shipped library through the Python boundary; the kernel time is the
library's own dispatch-attached event timing.  Prints one line per case with
the algorithmic bytes (input once, plus the dense output when there is one).
"""
import ctypes as C
import os
import sys
import tempfile

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from genrec_amd import _lib  # noqa: E402
from genrec_amd.logit_processor import Candidates, ConstrainedLogitsProcessor  # noqa: E402
from genrec_amd.synth import ASSISTANT_PREFIX, SidTokenizer, sample_decode_rows  # noqa: E402
from genrec_amd.trie import build_trie_from_index, create_prefix_allowed_tokens_fn  # noqa: E402
from genrec_amd.synth import write_index_file  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    path = write_index_file(os.environ.get("BM_TRIE", "Instruments"), tempfile.mkdtemp())
    tok = SidTokenizer.from_index(path)
    trie, suffix, pidx = build_trie_from_index(path, tok, prefix=ASSISTANT_PREFIX)
    fn = create_prefix_allowed_tokens_fn(trie, suffix)
    V = tok.vocab_size_padded
    K = 16
    step = int(os.environ.get("BM_STEP", 1))
    reps = int(os.environ.get("BM_REPS", 20))
    row_counts = [int(x) for x in os.environ.get("BM_ROWS", "8192,2048,1024,512,256,64,16").split(",")]
    cases = os.environ.get("BM_CASES", "compact:bf16,compact:fp32,dense:fp32:fp32,dense:bf16:bf16").split(",")
    if os.environ.get("BM_LIB"):  # a tuning build instead of the shipped library
        _lib.LIB_PATH = os.path.abspath(os.environ["BM_LIB"])
    h = _lib.lib()
    proc = ConstrainedLogitsProcessor(fn, K, pidx, None, tok.eos_token_id)
    print(f"trie={os.environ.get('BM_TRIE', 'Instruments')} V={V} step={step}")
    for R in row_counts:
        ids = torch.from_numpy(sample_decode_rows(trie, R // K, K, step, seed=1)).to(dev)
        for case in cases:
            parts = case.split(":")
            dt = {"bf16": torch.bfloat16, "fp32": torch.float32}[parts[1]]
            bufs = []
            for _ in range(2):  # alternate buffers so nothing is served from a cache
                x = torch.empty((R, V), dtype=dt, device=dev)
                for r0 in range(0, R, 1024):
                    x[r0 : r0 + 1024] = (torch.randn((min(1024, R - r0), V), device=dev) * 4).to(dt)
                bufs.append(x)
            esz = bufs[0].element_size()
            nbytes = R * V * esz
            if parts[0] == "compact":
                cand = Candidates(R, trie.max_fanout, dev)

                def call(i):
                    proc.count = step
                    proc.candidates_into(ids, bufs[i & 1], cand)
            else:
                nbytes += R * V * esz

                def call(i):
                    proc.count = step
                    return proc(ids, bufs[i & 1])

            for i in range(3):
                call(i)
            torch.cuda.synchronize()
            h.grx_timing_enable(1)
            for i in range(reps):
                call(i)
            torch.cuda.synchronize()
            tot, n = C.c_double(), C.c_int32()
            h.grx_timing_read(C.byref(tot), C.byref(n))
            h.grx_timing_enable(0)
            ms = tot.value / max(n.value, 1)
            print(f"R={R:6d} {case:18s} {ms * 1e3:9.1f} us  {nbytes / 1e9:7.3f} GB  {nbytes / ms / 1e6:7.0f} GB/s "
                  f"({nbytes / ms / 1e6 / 8000 * 100:5.1f}% of 8 TB/s)", flush=True)
            del bufs


if __name__ == "__main__":
    main()
