"""Evaluation-style decoding recorded from the installed transformers ``generate`` driving the REAL
reference processor the way evaluate.py does (evaluate.py:283-288, 358-371): count-based processor
with ``prefix_ids=None``, ``count`` reset by hand before every batch, 50 beams all returned,
``length_penalty=0.0``, ``repetition_penalty=1.1`` (so transformers' own penalty processor runs ahead
of the reference's), two batches in a row.

Run in the build container only (needs the reference checkout; it never ships):

    python tests/golden/make_golden_eval50.py [--reference /root/reference]

Asserts that the CPU restatement (oracle/beam.py + oracle/processors.py) reproduces generate() on the
live beams, then writes tests/golden/eval50.npz.
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

from make_golden import load_reference, save_npz  # noqa: E402

from oracle import beam as obeam  # noqa: E402
from oracle.processors import DictTrie, RefConstrainedLogitsProcessor, ref_repetition_penalty  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reference", default="/root/reference")
    args = ap.parse_args()
    ref = load_reference(args.reference)
    import transformers
    from transformers import LogitsProcessorList, Qwen2Config, Qwen2ForCausalLM

    rng = np.random.default_rng(29)
    prefix, eos, pad, V, K, levels = [11, 12, 13], 2, 0, 1024, 50, 4
    penalty, lp = 1.1, 0.0
    items = set()
    while len(items) < 700:
        items.add(tuple(int(100 + 200 * l + rng.integers(0, 28 if l == 0 else 40)) for l in range(levels)))
    seqs = [prefix + list(it) + [eos] for it in sorted(items)]
    ref_trie = ref.trie.Trie(seqs)
    fn = ref.evaluate.create_prefix_allowed_tokens_fn(ref_trie, prefix)
    torch.manual_seed(31)
    cfg = Qwen2Config(vocab_size=V, hidden_size=32, intermediate_size=64, num_hidden_layers=2,
                      num_attention_heads=2, num_key_value_heads=2, max_position_embeddings=128,
                      eos_token_id=eos, pad_token_id=pad, bos_token_id=None, tie_word_embeddings=False)
    model = Qwen2ForCausalLM(cfg).eval()
    with torch.no_grad():
        for p in model.parameters():
            p.mul_(5.0)
    arrays = {"trie_sequences": np.asarray(seqs, dtype=np.int64)}
    meta = {"prefix": prefix, "eos": eos, "pad": pad, "V": V, "K": K, "length_penalty": lp,
            "repetition_penalty": penalty, "transformers": transformers.__version__, "batches": []}
    # the processor object lives across batches, as in evaluate.py; prompts contain tokens that are also
    # trie tokens, so the repetition penalty touches allowed candidates
    proc = ref.count.ConstrainedLogitsProcessor(fn, K, 3, None, eos)
    max_new = levels + 1
    for b in range(2):
        B = 3
        rows = []
        for _ in range(B):
            it = seqs[int(rng.integers(0, len(seqs)))]
            rows.append(rng.integers(20, 90, size=5).tolist() + [it[3], it[4]] + prefix)
        prompts = torch.tensor(rows, dtype=torch.int64)
        proc.count = 0  # evaluate.py:358-361
        with torch.no_grad():
            out = model.generate(
                prompts, attention_mask=torch.ones_like(prompts), logits_processor=LogitsProcessorList([proc]),
                num_beams=K, num_return_sequences=K, max_new_tokens=max_new, do_sample=False, length_penalty=lp,
                repetition_penalty=penalty, eos_token_id=eos, pad_token_id=pad, output_scores=True,
                output_logits=True, return_dict_in_generate=True)
        steps = len(out.scores)
        assert steps == max_new

        class Chain:  # generate()'s processor list: repetition penalty, then the reference processor
            def __init__(self):
                self.inner = RefConstrainedLogitsProcessor(DictTrie(seqs).get, K, 3, None, eos)

            def __call__(self, ids, scores):
                return self.inner(ids, ref_repetition_penalty(ids, scores, penalty))

        state, _ = obeam.ref_constrained_beam_search(lambda t, flat: out.logits[t], Chain(), prompts, K, max_new, eos,
                                                     pad, lp, False)
        seqs_o, scores_o, bidx_o = state.finalize()
        live = out.sequences_scores > -1.0e8
        assert torch.equal(live, scores_o > -1.0e8), "live sets differ"
        assert float((scores_o[live] - out.sequences_scores[live]).abs().max()) <= 2e-5
        w = min(seqs_o.shape[1], out.sequences.shape[1])
        assert torch.equal(seqs_o[:, :w][live], out.sequences[:, :w][live]), "sequences differ"
        assert torch.equal(bidx_o.long()[live], out.beam_indices.long()[live]), "parents differ"
        name = f"batch{b}"
        meta["batches"].append({"name": name, "steps": steps, "max_new_tokens": max_new, "live": int(live.sum())})
        arrays[f"{name}.prompts"] = prompts.numpy()
        arrays[f"{name}.sequences"] = out.sequences.numpy()
        arrays[f"{name}.sequences_scores"] = out.sequences_scores.numpy()
        arrays[f"{name}.beam_indices"] = out.beam_indices.numpy().astype(np.int32)
        for t in range(steps):
            arrays[f"{name}.logits.{t}"] = out.logits[t].numpy().astype(np.float32)
        print(f"  {name}: steps={steps} live={int(live.sum())}/{live.numel()} count after={proc.count}")
    arrays["meta_json"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    save_npz(os.path.join(HERE, "eval50.npz"), **arrays)
    print("oracle loop == generate() on all live beams of both batches")


if __name__ == "__main__":
    main()
