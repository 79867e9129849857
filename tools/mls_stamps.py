#!/usr/bin/env python3
"""Phase timestamps inside the streaming kernel (needs a -DMLS_STAMP build at build/variants/lib_stamp.so)."""
import ctypes as C
import os
import sys
import tempfile

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from genrec_amd import _lib  # noqa: E402

_lib.LIB_PATH = os.path.join(ROOT, "build/variants", os.environ.get("ST_LIB", "lib_stamp.so"))
from genrec_amd.logit_processor import Candidates, ConstrainedLogitsProcessor  # noqa: E402
from genrec_amd.synth import ASSISTANT_PREFIX, SidTokenizer, sample_decode_rows  # noqa: E402
from genrec_amd.trie import build_trie_from_index, create_prefix_allowed_tokens_fn  # noqa: E402
from genrec_amd.synth import write_index_file  # noqa: E402

dev = torch.device("cuda", 0)
path = write_index_file(os.environ.get("ST_TRIE", "Instruments"), tempfile.mkdtemp())
tok = SidTokenizer.from_index(path)
trie, suffix, pidx = build_trie_from_index(path, tok, prefix=ASSISTANT_PREFIX)
fn = create_prefix_allowed_tokens_fn(trie, suffix)
V, K = tok.vocab_size_padded, int(os.environ.get("ST_BEAMS", 16))
PEN = float(os.environ.get("ST_PENALTY", 1.0))
PROMPT = int(os.environ.get("ST_PROMPT", 61))
h = _lib.lib()
h.grx_debug_read_mls_stamps.argtypes = [C.c_void_p]
proc = ConstrainedLogitsProcessor(fn, K, pidx, None, tok.eos_token_id, **({"repetition_penalty": PEN} if PEN != 1.0 else {}))
names = ["start->head issued+resolve", "resolve->gather issued", "stream", "combine", "merge/logf", "emit+ack"]
for R in [int(x) for x in os.environ.get("ST_ROWS", "16,256,8192").split(",")]:
    step = int(os.environ.get("ST_STEP", 1))
    ids = torch.from_numpy(sample_decode_rows(trie, R // K, K, step, prompt_len=PROMPT, seed=1)).to(dev)
    x = (torch.randn((R, V), device=dev) * 4).to(torch.bfloat16)
    cand = Candidates(R, trie.max_fanout, dev)
    for _ in range(3):
        proc.count = step
        proc.candidates_into(ids, x, cand, keep_top=int(os.environ.get("ST_KEEP", 0)))
    torch.cuda.synchronize()
    buf = np.zeros(4096 * 8, dtype=np.uint64)
    assert h.grx_debug_read_mls_stamps(buf.ctypes.data) == 0
    n_wg = min(4096, R * max(1, int(h.grx_workspace_bytes(R) and 1)))
    st = buf.reshape(4096, 8).astype(np.int64)
    live = st[:, 0] > 0
    st = st[live]
    t0 = st[:, 0].min()
    print(f"penalty={PEN} step={step}", end=" ")
    print(f"R={R}: {len(st)} workgroups stamped; kernel span {(st[:, 6].max() - t0) / 100:.1f} us "
          f"(first start -> last end); start spread {(st[:, 0].max() - t0) / 100:.1f} us")
    for i, nm in enumerate(names):
        d = (st[:, i + 1] - st[:, i]) / 100.0
        d = d[st[:, i + 1] > 0]
        if len(d):
            print(f"   {nm:28s} median {np.median(d):7.2f} us   p90 {np.percentile(d, 90):7.2f}   max {d.max():7.2f}  (n={len(d)})")

    if os.environ.get("ST_TIMELINE") and R >= 4096:
        # how many of the stamped workgroups are streaming / between rows in every 4 us of the kernel
        bins = np.arange(0, (st[:, 6].max() - t0) / 100.0 + 4, 4.0)
        s_lo, s_hi = (st[:, 2] - t0) / 100.0, (st[:, 3] - t0) / 100.0
        b_lo, b_hi = (st[:, 0] - t0) / 100.0, (st[:, 6] - t0) / 100.0
        starts = np.histogram(b_lo, bins)[0]
        print("   t(us)  starts  streaming  resident(stamped)")
        for i in range(len(bins) - 1):
            mid = bins[i] + 2.0
            print(f"   {bins[i]:5.0f}  {starts[i]:6d}  {int(((s_lo <= mid) & (s_hi > mid)).sum()):9d}  {int(((b_lo <= mid) & (b_hi > mid)).sum()):8d}")
