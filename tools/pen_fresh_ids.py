#!/usr/bin/env python3
"""Does the penalised streaming kernel run slower on token rows that another kernel has just written?
Times the call on rows rewritten right before it against the call on rows left alone (evaluation shape)."""
import os
import sys
import tempfile

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from genrec_amd.logit_processor import Candidates, ConstrainedLogitsProcessor  # noqa: E402
from genrec_amd.synth import ASSISTANT_PREFIX, SidTokenizer, sample_decode_rows, write_index_file  # noqa: E402
from genrec_amd.trie import build_trie_from_index, create_prefix_allowed_tokens_fn  # noqa: E402

dev = torch.device("cuda", 0)
path = write_index_file("Arts", tempfile.mkdtemp())
tok = SidTokenizer.from_index(path)
trie, suffix, pidx = build_trie_from_index(path, tok, prefix=ASSISTANT_PREFIX)
fn = create_prefix_allowed_tokens_fn(trie, suffix)
V, K, R, step = tok.vocab_size_padded, 50, 8000, 2
src = torch.from_numpy(sample_decode_rows(trie, R // K, K, step, prompt_len=253, seed=1)).to(dev)
ids = src.clone()
x = (torch.randn((R, V), device=dev) * 4).to(torch.bfloat16)
cand = Candidates(R, trie.max_fanout, dev)
for pen in (1.1, 1.0):
    proc = ConstrainedLogitsProcessor(fn, K, pidx, None, tok.eos_token_id, **({"repetition_penalty": pen} if pen != 1.0 else {}))
    for mode in ("kept", "rewritten", "kept", "rewritten", "rewritten, promise", "kept, promise", "rewritten, promise"):
        ts = []
        # "promise": the K rows of a group share their prompt, and the call says so (shared_prefix)
        hint = (256, K) if "promise" in mode else None
        for it in range(30):
            if mode.startswith("rewritten"):
                ids.copy_(src)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            proc.count = step
            e0.record()
            proc.candidates_into(ids, x, cand, keep_top=100, shared_prefix=hint)
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
        ts = sorted(ts[5:])
        print(f"penalty {pen} ids {mode:18s}: median {ts[len(ts) // 2]:7.1f} us  min {ts[0]:7.1f}", flush=True)
