#!/usr/bin/env python3
"""LM head + constrained step, whole versus by column slices, on this GPU.

Three ways to get from the last hidden states [R, H] to the compact candidates of a decode step
(R = 8192 rows, V = 152,640, bf16, library GEMM through torch):

  whole    one GEMM writes the [R, V] logits (2.5 GB), grx_constrained_candidates reads them back
  sliced   the GEMM runs over slices of Vc columns into a small buffer that
           grx_constrained_partial consumes at once (one stream), then grx_constrained_finish
  overlap  as sliced, but GEMMs and partial kernels on two streams with three slice buffers, so the
           (memory-bound) consumer of slice i runs beside the (compute-bound) GEMM of slice i + 1

Prints one JSON line per (H, variant).  Run under `rocprofv3 --kernel-trace --stats` for the
per-kernel durations.
"""
import argparse
import json
import os
import sys
import tempfile

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from genrec_amd.logit_processor import Candidates, ConstrainedLogitsProcessor  # noqa: E402
from genrec_amd.synth import ASSISTANT_PREFIX, SidTokenizer, sample_decode_rows, write_index_file  # noqa: E402
from genrec_amd.trie import build_trie_from_index, create_prefix_allowed_tokens_fn  # noqa: E402


def timed(fn, n, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--hidden", type=int, nargs="*", default=[1536, 2048, 3584])
    ap.add_argument("--slices", type=int, nargs="*", default=[4096, 8192, 16384])
    ap.add_argument("--rows", type=int, default=8192)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--parts", type=int, default=0, help="waves per row of a slice (0: the library's choice)")
    ap.add_argument("--variants", nargs="*", default=["sliced", "overlap"])
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    path = write_index_file("Instruments", tempfile.mkdtemp())
    tok = SidTokenizer.from_index(path)
    trie, suffix, pidx = build_trie_from_index(path, tok, prefix=ASSISTANT_PREFIX)
    fn = create_prefix_allowed_tokens_fn(trie, suffix)
    K, V = 16, tok.vocab_size_padded
    R = args.rows
    ids = torch.from_numpy(sample_decode_rows(trie, R // K, K, 1, prompt_len=61, seed=3)).to(dev)
    proc = ConstrainedLogitsProcessor(fn, K, pidx, None, tok.eos_token_id)
    cand = Candidates(R, trie.max_fanout, dev)
    cand2 = Candidates(R, trie.max_fanout, dev)
    side = torch.cuda.Stream(device=dev)

    for H in args.hidden:
        torch.manual_seed(H)
        h = torch.randn(R, H, device=dev, dtype=torch.bfloat16)
        w = torch.randn(V, H, device=dev, dtype=torch.bfloat16) * 0.02
        out = torch.empty(R, V, device=dev, dtype=torch.bfloat16)
        flops = 2.0 * R * V * H

        def gemm_whole():
            torch.matmul(h, w.t(), out=out)

        def step_whole():
            proc.count = 1
            proc.candidates_into(ids, out, cand)

        def whole():
            gemm_whole()
            step_whole()

        t_gemm = timed(gemm_whole, args.iters)
        t_step = timed(step_whole, args.iters)
        t_whole = timed(whole, args.iters)
        print(json.dumps({"hidden": H, "variant": "whole", "ms": t_whole, "gemm_ms": t_gemm, "step_ms": t_step,
                          "gemm_TFLOPs": flops / t_gemm / 1e9}), flush=True)
        want_tokens, want_logp, want_count = cand.tokens.clone(), cand.logp.clone(), cand.count.clone()
        del out

        for Vc in args.slices:
            bounds = [(a, min(a + Vc, V)) for a in range(0, V, Vc)]
            flat = [torch.empty(R * Vc, device=dev, dtype=torch.bfloat16) for _ in range(3)]

            def gemm_slices():
                for a, b in bounds:
                    torch.matmul(h, w[a:b].t(), out=flat[0][: R * (b - a)].view(R, b - a))

            def sliced():
                proc.count = 1
                call = proc.chunked_step(ids, cand2, V, bounds, torch.bfloat16, parts=args.parts or None)
                for a, b in bounds:
                    buf = flat[0][: R * (b - a)].view(R, b - a)
                    torch.matmul(h, w[a:b].t(), out=buf)
                    call.feed(a, buf)
                call.finish()

            def overlap():
                proc.count = 1
                call = proc.chunked_step(ids, cand2, V, bounds, torch.bfloat16, parts=args.parts or None)
                main = torch.cuda.current_stream(dev)
                free = [None, None, None]  # event after which a buffer may be overwritten
                side.wait_stream(main)
                for i, (a, b) in enumerate(bounds):
                    buf = flat[i % 3][: R * (b - a)].view(R, b - a)
                    if free[i % 3] is not None:
                        main.wait_event(free[i % 3])
                    torch.matmul(h, w[a:b].t(), out=buf)
                    ready = torch.cuda.Event()
                    ready.record(main)
                    with torch.cuda.stream(side):
                        side.wait_event(ready)
                        call.feed(a, buf)
                        done = torch.cuda.Event()
                        done.record(side)
                    free[i % 3] = done
                main.wait_stream(side)
                call.finish()

            t_g = timed(gemm_slices, args.iters)
            for name, f in (("sliced", sliced), ("overlap", overlap)):
                if name not in args.variants:
                    continue
                t = timed(f, args.iters)
                ok = bool(torch.equal(cand2.count, want_count))
                valid = torch.arange(cand2.cmax, device=dev)[None, :] < want_count[:, None]
                ok = ok and bool(torch.equal(cand2.tokens[valid], want_tokens[valid]))
                err = float((cand2.logp[valid] - want_logp[valid]).abs().max())
                print(json.dumps({"hidden": H, "variant": name, "slice_columns": Vc, "slices": len(bounds), "ms": t,
                                  "gemm_only_ms": t_g, "vs_whole": t / t_whole, "same_tokens": ok,
                                  "max_logp_diff": err}), flush=True)
            del flat
        del h, w
    proc.check_errors()


if __name__ == "__main__":
    main()
