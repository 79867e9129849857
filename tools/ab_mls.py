#!/usr/bin/env python3
"""Interleaved A/B timing of masked-log-softmax build variants (build/variants/lib_*.so).

Each variant is a full libgenrec_hip.so; the same inputs go through
grx_constrained_candidates of every variant in round-robin order inside one
process, and the library's own event timing brackets the streaming kernel.
"""
import ctypes as C
import glob
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from genrec_amd import _lib  # noqa: E402
from genrec_amd.synth import ASSISTANT_PREFIX, SidTokenizer, sample_decode_rows  # noqa: E402
from genrec_amd.trie import build_trie_from_index  # noqa: E402
from genrec_amd.synth import write_index_file  # noqa: E402


def main():
    import tempfile

    R = int(os.environ.get("AB_ROWS", 8192))
    rounds = int(os.environ.get("AB_ROUNDS", 12))
    dtype = {"bf16": torch.bfloat16, "fp32": torch.float32}[os.environ.get("AB_DTYPE", "bf16")]
    step = int(os.environ.get("AB_STEP", 1))
    dev = torch.device("cuda", 0)
    trie_name = os.environ.get("AB_TRIE", "Instruments")
    if trie_name.startswith("cb"):
        from genrec_amd.synth import synthetic_index, write_index

        path = write_index(os.path.join(tempfile.mkdtemp(), "cb.index.json"),
                           synthetic_index(9922, 4, int(trie_name[2:]), seed=0))
    else:
        path = write_index_file(trie_name, tempfile.mkdtemp())
    tok = SidTokenizer.from_index(path)
    trie, suffix, pidx = build_trie_from_index(path, tok, prefix=ASSISTANT_PREFIX)
    V = tok.vocab_size_padded
    beams = int(os.environ.get("AB_BEAMS", 16))
    prompt_len = int(os.environ.get("AB_PROMPT", 61))
    penalty = float(os.environ.get("AB_PENALTY", 1.0))
    ids = torch.from_numpy(sample_decode_rows(trie, R // beams, beams, step, prompt_len=prompt_len, seed=1)).to(dev)
    bufs = []
    for _ in range(2):
        x = torch.empty((R, V), dtype=dtype, device=dev)
        for r0 in range(0, R, 1024):
            x[r0 : r0 + 1024] = (torch.randn((min(1024, R - r0), V), device=dev) * 4).to(dtype)
        bufs.append(x)
    from genrec_amd.logit_processor import Candidates

    keep = int(os.environ.get("AB_KEEP", 0))  # rows keep their AB_KEEP best (the bench uses 2 x beams)
    cmax = min(trie.max_fanout, keep + 64) if keep else trie.max_fanout
    cand = Candidates(R, cmax, dev)
    libs = {}
    names = sys.argv[1:] or sorted(os.path.basename(p)[4:-3] for p in glob.glob(os.path.join(ROOT, "build/variants/lib_*.so")))
    for name in names:
        h = C.CDLL(os.path.join(ROOT, "build/variants", f"lib_{name}.so"))
        _lib._declare(h)
        t = C.c_void_p()
        assert h.grx_trie_create(trie.child_off.ctypes.data_as(C.c_void_p), trie.child_tok.ctypes.data_as(C.c_void_p),
                                 trie.n_nodes, trie.n_edges, C.byref(t)) == 0
        st = C.c_void_p()
        assert h.grx_state_create(C.byref(st)) == 0
        ws = torch.zeros(int(h.grx_workspace_bytes(R)), dtype=torch.uint8, device=dev)
        libs[name] = (h, t, st, ws)
    rows = _lib.GrxRows()
    rows.input_ids, rows.ids_stride, rows.R, rows.T = ids.data_ptr(), ids.stride(0), R, ids.shape[1]
    rows.scores_stride, rows.V, rows.dtype = V, V, _lib.torch_dtype_code(dtype)
    con = _lib.GrxConstraint()
    con.mode, con.prefix_index, con.n_prefix, con.eos_token_id = 0, pidx, -1, tok.eos_token_id
    if penalty != 1.0:
        con.repetition_penalty = penalty
    con.keep_top = keep
    stream = torch.cuda.current_stream().cuda_stream
    results = {n: [] for n in libs}
    ref = None
    for rnd in range(rounds + 2):
        for name, (h, t, st, ws) in libs.items():
            h.grx_state_set_count(st, step, stream)
            rows.scores = bufs[rnd & 1].data_ptr()
            h.grx_timing_enable(1)
            for _ in range(3):
                h.grx_state_set_count(st, step, stream)
                rc = h.grx_constrained_candidates(t, st, C.byref(rows), C.byref(con), cand.buf.data_ptr(), cand.stride,
                                                  cmax, ws.data_ptr(), ws.numel(), stream)
                assert rc == 0, h.grx_last_error()
            tot, n = C.c_double(), C.c_int32()
            h.grx_timing_read(C.byref(tot), C.byref(n))
            h.grx_timing_enable(0)
            if rnd >= 2:
                results[name].append(tot.value / n.value)
            if rnd == 0:
                got = (cand.count.clone(), cand.tokens.clone(), cand.logp.clone())
                if ref is None:
                    ref = got
                else:
                    if "abl" in name:
                        continue
                    assert torch.equal(got[0], ref[0]), name
                    m = torch.arange(cmax, device=dev)[None, :] < ref[0][:, None]
                    if keep:  # kept entries come in no particular order
                        big = torch.iinfo(torch.int32).max
                        o_g = torch.where(m, got[1], big).argsort(dim=1)
                        o_r = torch.where(m, ref[1], big).argsort(dim=1)
                        got = (got[0], got[1].gather(1, o_g), got[2].gather(1, o_g))
                        ref_c = (ref[0], ref[1].gather(1, o_r), ref[2].gather(1, o_r))
                    else:
                        ref_c = ref
                    same = got[1][m] == ref_c[1][m]
                    print(f"{name}: tokens equal {bool(same.all())}, max |dlogp| {float((got[2][m] - ref_c[2][m]).abs().max()):.2e}")
                    assert (got[2][m] - ref_c[2][m])[same].abs().max() < 3e-5, name
    nbytes = R * V * (2 if dtype == torch.bfloat16 else 4)
    print(f"rows={R} V={V} dtype={dtype} step={step} T={ids.shape[1]} penalty={penalty} bytes={nbytes/1e9:.3f} GB")
    for name, v in sorted(results.items(), key=lambda kv: sorted(kv[1])[len(kv[1]) // 2]):
        v = sorted(v)
        med, mn = v[len(v) // 2], v[0]
        print(f"{name:14s} median {med*1e3:8.1f} us  min {mn*1e3:8.1f} us   {nbytes/med/1e6:7.0f} GB/s (median)  {nbytes/mn/1e6:7.0f} GB/s (best)")


if __name__ == "__main__":
    main()
