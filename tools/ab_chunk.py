#!/usr/bin/env python3
"""Interleaved A/B timing of build variants of the slice kernel (build/variants/lib_*.so).

A library GEMM writes one [R, Vc] bf16 slice, then grx_constrained_partial of a variant consumes it,
as in the sliced LM head; HIP events bracket the partial call.  AB_PARTS lists the waves-per-row
values to try with every variant.
"""
import ctypes as C
import glob
import os
import sys
import tempfile

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from genrec_amd import _lib  # noqa: E402
from genrec_amd.logit_processor import Candidates  # noqa: E402
from genrec_amd.synth import ASSISTANT_PREFIX, SidTokenizer, sample_decode_rows, write_index_file  # noqa: E402
from genrec_amd.trie import build_trie_from_index  # noqa: E402


def main():
    R = int(os.environ.get("AB_ROWS", 8192))
    Vc = int(os.environ.get("AB_SLICE", 8192))
    H = int(os.environ.get("AB_HIDDEN", 2048))
    rounds = int(os.environ.get("AB_ROUNDS", 10))
    parts_list = [int(p) for p in os.environ.get("AB_PARTS", "1").split(",")]
    # gemm: a GEMM writes the slice right before; none: 40 slices in rotation (cold in every cache);
    # hot: one slice read over and over (whatever cache can hold it, does)
    producer = os.environ.get("AB_PRODUCER", "gemm")
    dev = torch.device("cuda", 0)
    path = write_index_file("Instruments", tempfile.mkdtemp())
    tok = SidTokenizer.from_index(path)
    trie, suffix, pidx = build_trie_from_index(path, tok, prefix=ASSISTANT_PREFIX)
    V = tok.vocab_size_padded
    ids = torch.from_numpy(sample_decode_rows(trie, R // 16, 16, 1, seed=1)).to(dev)
    h_ = torch.randn(R, H, device=dev, dtype=torch.bfloat16)
    w = torch.randn(Vc, H, device=dev, dtype=torch.bfloat16) * 0.02
    bufs = [torch.empty(R, Vc, device=dev, dtype=torch.bfloat16) for _ in range(40 if producer == "none" else 1)]
    for b in bufs:
        torch.matmul(h_, w.t(), out=b)
    cmax = trie.max_fanout
    cand = Candidates(R, cmax, dev)
    names = sys.argv[1:] or sorted(os.path.basename(p)[4:-3] for p in glob.glob(os.path.join(ROOT, "build/variants/lib_*.so")))
    libs = {}
    for name in names:
        h = C.CDLL(os.path.join(ROOT, "build/variants", f"lib_{name}.so"))
        _lib._declare(h)
        t, st = C.c_void_p(), C.c_void_p()
        assert h.grx_trie_create(trie.child_off.ctypes.data_as(C.c_void_p), trie.child_tok.ctypes.data_as(C.c_void_p),
                                 trie.n_nodes, trie.n_edges, C.byref(t)) == 0
        assert h.grx_state_create(C.byref(st)) == 0
        ws = torch.zeros(int(h.grx_workspace_bytes(R)), dtype=torch.uint8, device=dev)
        libs[name] = (h, t, st, ws)
    rows = _lib.GrxRows()
    rows.input_ids, rows.ids_stride, rows.R, rows.T = ids.data_ptr(), ids.stride(0), R, ids.shape[1]
    rows.scores_stride, rows.V, rows.dtype = Vc, Vc, _lib.GRX_BF16
    con = _lib.GrxConstraint()
    con.mode, con.prefix_index, con.n_prefix, con.eos_token_id = 0, pidx, -1, tok.eos_token_id
    stream = torch.cuda.current_stream().cuda_stream
    partials = torch.zeros(R * 64 * 2, device=dev)
    results = {}
    v_begin = (V - Vc) // 8 * 8 if os.environ.get("AB_LAST", "0") == "1" else 8192  # a slice with / without SID tokens
    for rnd in range(rounds + 2):
        for name, (h, t, st, ws) in libs.items():
            for parts in parts_list:
                ev = []
                for i in range(8):
                    buf = bufs[(rnd * 8 + i) % len(bufs)]
                    if producer == "gemm":
                        torch.matmul(h_, w.t(), out=buf)
                    rows.scores = buf.data_ptr()
                    ch = _lib.GrxChunk()
                    ch.v_begin, ch.v_total, ch.slot, ch.n_slots, ch.parts, ch.first = v_begin, V, 0, 64, parts, 1 if i == 0 else 0
                    if i == 0:
                        h.grx_state_set_count(st, 1, stream)
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    rc = h.grx_constrained_partial(t, st, C.byref(rows), C.byref(con), C.byref(ch), partials.data_ptr(),
                                                   cand.buf.data_ptr(), cand.stride, cmax, ws.data_ptr(), ws.numel(), stream)
                    e1.record()
                    assert rc == 0, h.grx_last_error()
                    if i > 0:
                        ev.append((e0, e1))
                torch.cuda.synchronize()
                if rnd >= 2:
                    results.setdefault((name, parts), []).extend(a.elapsed_time(b) for a, b in ev)
    nbytes = R * Vc * 2
    print(f"rows={R} slice={Vc} bytes={nbytes/1e6:.1f} MB producer={producer} v_begin={v_begin}")
    for key, v in sorted(results.items(), key=lambda kv: sorted(kv[1])[len(kv[1]) // 2]):
        v = sorted(v)
        med, mn = v[len(v) // 2], v[0]
        print(f"{key[0]:14s} parts={key[1]:2d} median {med*1e3:7.1f} us  min {mn*1e3:7.1f} us   {nbytes/med/1e6:7.0f} GB/s (median)")


if __name__ == "__main__":
    main()
