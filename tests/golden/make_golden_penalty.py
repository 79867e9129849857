"""Golden vectors for the fused repetition penalty, recorded from the REAL chain:
torch log_softmax -> transformers' RepetitionPenaltyLogitsProcessor -> the reference's
processors, i.e. what generate(repetition_penalty=p, logits_processor=[reference processor])
feeds to beam search (evaluate.py:218,363-371).

Run in the build container only (needs the reference checkout; it never ships):

    python tests/golden/make_golden_penalty.py [--reference /root/reference]

Writes tests/golden/penalty.npz: token rows, logit seeds, and per row the allowed ids with the
chain's fp32 values and a float64 evaluation of the same chain.  Asserts that the CPU
restatement in oracle/ reproduces the chain bit for bit.
"""

from __future__ import annotations

import argparse
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from make_golden import load_reference  # noqa: E402

from genrec_amd.synth import ASSISTANT_PREFIX, SidTokenizer, sample_decode_rows  # noqa: E402
from genrec_amd.trie import build_trie_from_index  # noqa: E402
from helpers import write_index_file  # noqa: E402
from oracle import processors as oproc  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reference", default="/root/reference")
    args = ap.parse_args()
    ref = load_reference(args.reference)
    from transformers.generation.logits_process import RepetitionPenaltyLogitsProcessor

    import tempfile

    path = write_index_file("Instruments", tempfile.mkdtemp())
    tok = SidTokenizer.from_index(path)
    trie, suffix, pidx = build_trie_from_index(path, tok, prefix=ASSISTANT_PREFIX)
    ref_trie = ref.trie.Trie(sorted(trie))
    o_trie = oproc.DictTrie(sorted(trie))
    V, K, eos = tok.vocab_size_padded, 4, tok.eos_token_id
    rng = np.random.default_rng(77)
    out = {}
    cases = []
    for ci, (proc_name, penalty, skip, step) in enumerate([
        ("count", 1.1, 0, 0), ("count", 1.1, 0, 1), ("count", 1.3, 0, 2), ("count", 0.8, 5, 1),
        ("fixed", 1.1, 0, 1), ("fixed", 2.0, 3, 3), ("count", 1.1, 0, 4),
    ]):
        ids = sample_decode_rows(trie, 3, K, step, prompt_len=29, seed=100 + ci)
        # interaction history: SID tokens inside the prompt, among them children of the rows' own nodes,
        # so that allowed tokens are penalised too; plus some repeated tokens
        n_sid = len(tok.new_tokens)
        for r in range(ids.shape[0]):
            hist = rng.integers(151665, 151665 + n_sid, size=12)
            ids[r, 3:15] = hist
            ids[r, 15] = ids[r, 3]
        if step == 4:
            ids[:, 16] = eos  # the forced token of the last step appears in the row as well
        ids_t = torch.from_numpy(ids)
        seed = 9000 + ci
        gen = torch.Generator().manual_seed(seed)
        logits = torch.randn(ids.shape[0], V, generator=gen) * 4.0
        if ci == 2:  # very peaked rows whose peak is a penalised token
            logits[torch.arange(ids.shape[0]), ids_t[:, 7]] += 40.0
        lp = torch.nn.functional.log_softmax(logits, dim=-1)
        pen = RepetitionPenaltyLogitsProcessor(penalty, prompt_ignore_length=skip or None)(ids_t, lp)
        if proc_name == "count":
            rp = ref.count.ConstrainedLogitsProcessor(ref_trie.get, K, pidx, None, eos)
            op = oproc.RefConstrainedLogitsProcessor(o_trie.get, K, pidx, None, eos)
            rp.count = op.count = step
        else:
            rp = ref.fixed.FixedHintConstrainedLogitsProcessor(ref_trie.get, K, pidx, suffix, eos)
            op = oproc.RefFixedHintConstrainedLogitsProcessor(o_trie.get, K, pidx, suffix, eos)
        want = rp(ids_t, pen)
        mine = op(ids_t, oproc.ref_repetition_penalty(ids_t, lp, penalty, skip))
        assert torch.equal(want, mine), "oracle chain differs from the reference chain"
        # float64 evaluation of the same chain
        lp64 = torch.log_softmax(logits.double(), dim=-1)
        pen64 = oproc.ref_repetition_penalty(ids_t, lp64, penalty, skip)
        exact = torch.log_softmax(pen64, dim=-1)
        finite = torch.isfinite(want)
        counts = finite.sum(dim=1)
        cmax = int(counts.max())
        tokm = np.zeros((ids.shape[0], cmax), dtype=np.int32)
        val = np.zeros((ids.shape[0], cmax), dtype=np.float32)
        val64 = np.zeros((ids.shape[0], cmax), dtype=np.float64)
        n_pen_allowed = 0
        for r in range(ids.shape[0]):
            cols = torch.nonzero(finite[r]).flatten()
            tokm[r, : len(cols)] = cols.numpy()
            val[r, : len(cols)] = want[r, cols].numpy()
            val64[r, : len(cols)] = exact[r, cols].numpy()
            n_pen_allowed += len(set(cols.tolist()) & set(ids[r, skip:].tolist()))
        out[f"{ci}.ids"] = ids
        out[f"{ci}.count"] = counts.numpy().astype(np.int32)
        out[f"{ci}.tok"] = tokm
        out[f"{ci}.val"] = val
        out[f"{ci}.val64"] = val64
        cases.append({"processor": proc_name, "penalty": penalty, "skip": skip, "step": step, "seed": seed,
                      "peaked": ci == 2, "penalised_allowed": n_pen_allowed})
        print(f"case {ci}: {proc_name} p={penalty} skip={skip} step={step}: {int(counts.sum())} allowed entries, "
              f"{n_pen_allowed} of them penalised; max |fp32 chain - fp64 chain| = "
              f"{float((want[finite].double() - exact[finite]).abs().max()):.2e}")
    meta = {"cases": cases, "K": K, "V": V, "eos": eos, "prefix": suffix, "prefix_index": pidx, "scale": 4.0}
    out["meta_json"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    path = os.path.join(HERE, "penalty.npz")
    np.savez_compressed(path, **out)
    print(f"wrote {os.path.relpath(path, ROOT)} ({os.path.getsize(path)} bytes)")


if __name__ == "__main__":
    main()
