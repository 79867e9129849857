#!/usr/bin/env python3
"""Fused LM head (grx_lm_head_candidates) against the library GEMM + streaming kernel, 8192 rows x V = 152,640.

    python tools/lm_head_fused_probe.py [H ...]

Per hidden size: torch.matmul alone (TFLOP/s), torch.matmul -> [R, V] bf16 -> candidates_into (the route the
reference's forward pass implies), and the fused call; times by CUDA events over interleaved rounds."""
import os
import sys
import tempfile

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from genrec_amd import _lib  # noqa: E402

if os.environ.get("LH_LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["LH_LIB"])
from genrec_amd.logit_processor import Candidates, ConstrainedLogitsProcessor  # noqa: E402
from genrec_amd.synth import ASSISTANT_PREFIX, SidTokenizer, sample_decode_rows, write_index_file  # noqa: E402
from genrec_amd.trie import build_trie_from_index, create_prefix_allowed_tokens_fn  # noqa: E402

dev = torch.device("cuda", 0)
path = write_index_file("Instruments", tempfile.mkdtemp())
tok = SidTokenizer.from_index(path)
trie, suffix, pidx = build_trie_from_index(path, tok, prefix=ASSISTANT_PREFIX)
fn = create_prefix_allowed_tokens_fn(trie, suffix)
V, K, R = tok.vocab_size_padded, 16, int(os.environ.get("LH_ROWS", 8192))
step = 1
ids = torch.from_numpy(sample_decode_rows(trie, R // K, K, step, prompt_len=61, seed=1)).to(dev)
proc = ConstrainedLogitsProcessor(fn, K, pidx, None, tok.eos_token_id)
cand_a = Candidates(R, trie.max_fanout, dev)
cand_b = Candidates(R, trie.max_fanout, dev)


def timed(fn_, n):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn_()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for H in [int(x) for x in (sys.argv[1:] or ["1536", "2048", "3584"])]:
    hidden = (torch.randn((R, H), device=dev) * 0.5).to(torch.bfloat16)
    weight = (torch.randn((V, H), device=dev) * 0.05).to(torch.bfloat16)
    logits = torch.empty((R, V), dtype=torch.bfloat16, device=dev)
    flops = 2.0 * R * V * H

    def gemm():
        torch.matmul(hidden, weight.T, out=logits)

    def library_route():
        torch.matmul(hidden, weight.T, out=logits)
        proc.count = step
        proc.candidates_into(ids, logits, cand_a)

    def fused():
        proc.count = step
        proc.lm_head_candidates_into(ids, hidden, weight, cand_b)

    for f in (gemm, library_route, fused):
        f()
    torch.cuda.synchronize()
    res = {"gemm": [], "library": [], "fused": []}
    for _ in range(5):
        res["gemm"].append(timed(gemm, 3))
        res["library"].append(timed(library_route, 3))
        res["fused"].append(timed(fused, 3))
    med = {k: sorted(v)[len(v) // 2] for k, v in res.items()}
    same = bool(torch.equal(cand_a.count, cand_b.count))
    print(f"H={H}: torch.matmul {med['gemm']:.3f} ms ({flops / med['gemm'] / 1e9:.0f} TFLOP/s) | matmul + step "
          f"{med['library']:.3f} ms | fused {med['fused']:.3f} ms ({flops / med['fused'] / 1e9:.0f} TFLOP/s incl. finish) "
          f"| fused / library = {med['fused'] / med['library']:.3f} | counts equal {same}", flush=True)
    del hidden, weight, logits
proc.check_errors()
