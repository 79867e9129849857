#!/usr/bin/env python3
"""Phase timestamps of the beam steps that run inside the streaming kernel (needs a -DFUSE_STAMP build at
build/variants/lib_fstamp.so): how long a group's step takes while the other groups' rows stream."""
import ctypes as C
import os
import sys
import tempfile

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from genrec_amd import _lib  # noqa: E402

_lib.LIB_PATH = os.path.join(ROOT, "build/variants", os.environ.get("ST_LIB", "lib_fstamp.so"))
from genrec_amd.beam import BeamSearchState  # noqa: E402
from genrec_amd.logit_processor import Candidates, ConstrainedLogitsProcessor  # noqa: E402
from genrec_amd.synth import ASSISTANT_PREFIX, SidTokenizer, sample_decode_rows, write_index_file  # noqa: E402
from genrec_amd.trie import build_trie_from_index, create_prefix_allowed_tokens_fn  # noqa: E402

dev = torch.device("cuda", 0)
path = write_index_file("Instruments", tempfile.mkdtemp())
tok = SidTokenizer.from_index(path)
trie, suffix, pidx = build_trie_from_index(path, tok, prefix=ASSISTANT_PREFIX)
fn = create_prefix_allowed_tokens_fn(trie, suffix)
K, B = 16, int(os.environ.get("ST_GROUPS", 512))
V = tok.vocab_size_padded
proc = ConstrainedLogitsProcessor(fn, K, pidx, suffix, tok.eos_token_id)
prompts = torch.from_numpy(sample_decode_rows(trie, B, 1, 0, prompt_len=61, seed=1)).to(dev)
rows = prompts.repeat_interleave(K, dim=0).contiguous()
levels = 5
x = [torch.empty((B * K, V), dtype=torch.bfloat16, device=dev) for _ in range(2)]
for t in x:
    for r0 in range(0, B * K, 1024):
        t[r0:r0 + 1024] = (torch.randn((min(1024, B * K - r0), V), device=dev) * 4).to(torch.bfloat16)
h = _lib.lib()
h.grx_debug_read_fuse_stamps.argtypes = [C.c_void_p]
state = BeamSearchState(rows, K, rows.shape[1] + levels + 1, tok.eos_token_id, tok.pad_token_id, planner=proc)
keep = proc.keep_top_for(state.n_keep)
cand = Candidates(B * K, proc.candidate_width(state.n_keep), dev)
names = {1: "loads issued", 2: "loads arrived (barrier)", 3: "keys", 5: "threshold", 6: "survivors", 7: "ranked", 8: "barrier",
         9: "criteria", 10: "top-K ranks", 11: "barrier", 12: "results written", 13: "barrier", 14: "heuristic"}
for gen in range(2):
    proc.count = 0
    state.reset(rows, proc)
    for d in range(levels):
        state.fused_step(proc, x[d & 1], cand, keep_top=keep, planner=proc if d + 1 < levels else None)
        torch.cuda.synchronize()
        if gen == 0:
            continue
        buf = np.zeros(1024 * 16, dtype=np.uint64)
        assert h.grx_debug_read_fuse_stamps(buf.ctypes.data) == 0
        st = buf.reshape(1024, 16).astype(np.int64)[: min(B, 1024)]
        t0 = st[:, 0].min()
        print(f"depth {d}: first workgroup of the launch -> first beam step {(t0 - st[0, 15]) / 100:.1f} us, -> last beam step done "
              f"{(st[:, 14].max() - st[0, 15]) / 100:.1f} us")
        print(f"depth {d}: beam steps start {((st[:, 0] - t0) / 100).min():.1f} .. {((st[:, 0] - t0) / 100).max():.1f} us after the "
              f"first one; whole step median {np.median(st[:, 14] - st[:, 0]) / 100:.1f} us, p90 "
              f"{np.percentile(st[:, 14] - st[:, 0], 90) / 100:.1f}, max {(st[:, 14] - st[:, 0]).max() / 100:.1f}")
        prev = 0
        for n in sorted(names):
            dt = (st[:, n] - st[:, prev]) / 100.0
            print(f"    -> {names[n]:26s} median {np.median(dt):6.2f} us   p90 {np.percentile(dt, 90):6.2f}")
            prev = n
proc.check_errors()
