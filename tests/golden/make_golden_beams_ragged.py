"""Beam-search traces on a RAGGED trie (items of 1..4 levels, so EOS arrives at different steps and
finished beams compete with running ones), for every early_stopping mode and several length
penalties, recorded from the installed transformers ``generate`` driving the REAL reference processor.

Run in the build container only (needs the reference checkout; it never ships):

    python tests/golden/make_golden_beams_ragged.py [--reference /root/reference]

Asserts that the CPU restatement of the loop (oracle/beam.py) reproduces generate() on the live
beams of every case, then writes tests/golden/beams_ragged.npz (prompts, per-step raw logits and
processor outputs, returned sequences / scores / beam indices).
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


def ragged_sequences(rng, n_items, prefix, eos, codebook=10, max_levels=4):
    seqs = set()
    while len(seqs) < n_items:
        levels = int(rng.integers(1, max_levels + 1))
        item = tuple(int(100 + 30 * l + rng.integers(0, codebook)) for l in range(levels))
        seqs.add(tuple(prefix) + item + (eos,))
    return [list(s) for s in sorted(seqs)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reference", default="/root/reference")
    args = ap.parse_args()
    ref = load_reference(args.reference)
    import transformers
    from transformers import LogitsProcessorList, Qwen2Config, Qwen2ForCausalLM

    rng = np.random.default_rng(17)
    prefix, eos, pad, V = [11, 12, 13], 2, 0, 512
    seqs = ragged_sequences(rng, 260, prefix, eos)
    ref_trie = ref.trie.Trie(seqs)
    fn = ref.evaluate.create_prefix_allowed_tokens_fn(ref_trie, prefix)
    torch.manual_seed(23)
    cfg = Qwen2Config(vocab_size=V, hidden_size=32, intermediate_size=64, num_hidden_layers=2,
                      num_attention_heads=2, num_key_value_heads=2, max_position_embeddings=128,
                      eos_token_id=eos, pad_token_id=pad, bos_token_id=None, tie_word_embeddings=False)
    model = Qwen2ForCausalLM(cfg).eval()
    with torch.no_grad():
        for p in model.parameters():
            p.mul_(5.0)
    arrays = {"trie_sequences_flat": np.asarray([t for s in seqs for t in s], dtype=np.int64),
              "trie_sequence_lengths": np.asarray([len(s) for s in seqs], dtype=np.int64)}
    meta = {"prefix": prefix, "eos": eos, "pad": pad, "V": V, "transformers": transformers.__version__, "cases": {}}
    B = 4
    prompts = torch.tensor([rng.integers(20, 90, size=6).tolist() + prefix for _ in range(B)], dtype=torch.int64)
    arrays["prompts"] = prompts.numpy()
    n_ok = 0
    for K in (4, 8):
        for es in (False, True, "never"):
            for lp in (0.0, 1.0, 2.0, -0.5):
                name = f"k{K}_es{es}_lp{lp}"
                proc = ref.fixed.FixedHintConstrainedLogitsProcessor(fn, K, 3, prefix, eos)
                max_new = 7
                with torch.no_grad():
                    out = model.generate(
                        prompts, attention_mask=torch.ones_like(prompts),
                        logits_processor=LogitsProcessorList([proc]), num_beams=K, num_return_sequences=K,
                        max_new_tokens=max_new, do_sample=False, length_penalty=lp, early_stopping=es,
                        eos_token_id=eos, pad_token_id=pad, output_scores=True, output_logits=True,
                        return_dict_in_generate=True)
                steps = len(out.scores)
                state = obeam.RefBeamSearch(prompts.repeat_interleave(K, dim=0), K, prompts.shape[1] + max_new, eos,
                                            pad, lp, es)
                looping = []
                for t in range(steps):
                    state.step(out.scores[t])
                    looping.append(bool(state.has_unfinished()))
                seqs_o, scores_o, bidx_o = state.finalize()
                live = out.sequences_scores > -1.0e8
                width = min(seqs_o.shape[1], out.sequences.shape[1])
                assert torch.equal(live, scores_o > -1.0e8), f"{name}: live sets differ"
                assert torch.equal(scores_o[live], out.sequences_scores[live]), f"{name}: scores differ"
                assert torch.equal(seqs_o[:, :width][live], out.sequences[:, :width][live]), f"{name}: sequences differ"
                bw = min(bidx_o.shape[1], out.beam_indices.shape[1])
                assert torch.equal(bidx_o[:, :bw].long()[live], out.beam_indices[:, :bw].long()[live]), f"{name}: parents"
                # generate() stopped after `steps` iterations: the restated loop condition agrees at every step
                assert all(looping[:-1]) and (not looping[-1] or steps == max_new), f"{name}: loop condition {looping}"
                meta["cases"][name] = {"K": K, "early_stopping": es, "length_penalty": lp, "steps": steps,
                                       "max_new_tokens": max_new}
                arrays[f"{name}.sequences"] = out.sequences.numpy()
                arrays[f"{name}.sequences_scores"] = out.sequences_scores.numpy()
                arrays[f"{name}.beam_indices"] = out.beam_indices.numpy().astype(np.int32)
                for t in range(steps):
                    arrays[f"{name}.logits.{t}"] = out.logits[t].numpy().astype(np.float32)
                lens = sorted({int((r != pad).sum()) - prompts.shape[1] for r in out.sequences[live]})
                print(f"  {name}: steps={steps} live={int(live.sum())}/{live.numel()} returned lengths {lens}")
                n_ok += 1
    arrays["meta_json"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    save_npz(os.path.join(HERE, "beams_ragged.npz"), **arrays)
    print(f"{n_ok} cases: oracle loop == generate() on all live beams")


if __name__ == "__main__":
    main()
