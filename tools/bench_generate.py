#!/usr/bin/env python3
"""End-to-end decode of a randomly initialised Qwen2 on the GPU: transformers' generate() with our
processor in its LogitsProcessorList (the drop-in route) against constrained_generate (the fused
route).  Reports wall time per batch; results are compared on live beams."""
import os
import sys
import tempfile
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from transformers import LogitsProcessorList, Qwen2Config, Qwen2ForCausalLM  # noqa: E402

from genrec_amd.generate import constrained_generate  # noqa: E402
from genrec_amd.logit_processor import ConstrainedLogitsProcessor  # noqa: E402
from genrec_amd.synth import ASSISTANT_PREFIX, SidTokenizer, sample_decode_rows  # noqa: E402
from genrec_amd.trie import build_trie_from_index, create_prefix_allowed_tokens_fn  # noqa: E402
from genrec_amd.synth import write_index_file  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    B = int(os.environ.get("BG_BATCH", 64))
    K = int(os.environ.get("BG_BEAMS", 16))
    T = int(os.environ.get("BG_PROMPT", 253))
    hidden = int(os.environ.get("BG_HIDDEN", 1024))
    layers = int(os.environ.get("BG_LAYERS", 8))
    path = write_index_file("Instruments", tempfile.mkdtemp())
    tok = SidTokenizer.from_index(path)
    trie, suffix, pidx = build_trie_from_index(path, tok, prefix=ASSISTANT_PREFIX)
    fn = create_prefix_allowed_tokens_fn(trie, suffix)
    V = tok.vocab_size_padded
    torch.manual_seed(0)
    cfg = Qwen2Config(vocab_size=V, hidden_size=hidden, intermediate_size=hidden * 3, num_hidden_layers=layers,
                      num_attention_heads=hidden // 128, num_key_value_heads=max(1, hidden // 512),
                      max_position_embeddings=1024, eos_token_id=tok.eos_token_id, pad_token_id=tok.pad_token_id,
                      bos_token_id=None, tie_word_embeddings=False)
    model = Qwen2ForCausalLM(cfg).to(torch.bfloat16).to(dev).eval()
    prompts = torch.from_numpy(sample_decode_rows(trie, B, 1, 0, prompt_len=T, seed=1)).to(dev)
    mask = torch.ones_like(prompts)
    steps = 5
    print(f"Qwen2 random init: hidden {hidden}, {layers} layers, V {V}, bf16; batch {B} x {K} beams, prompt {prompts.shape[1]}")

    hf_procs = {}  # one processor per mode for the whole run, as the reference's launchers build theirs once

    def hf(vouched=False, kept=False):
        proc = hf_procs.get((vouched, kept))
        if proc is None:
            proc = hf_procs[(vouched, kept)] = ConstrainedLogitsProcessor(
                fn, K, pidx, suffix, tok.eos_token_id, assume_normalized=vouched, reuse_output=kept)
        with torch.no_grad():
            return model.generate(prompts, attention_mask=mask, logits_processor=LogitsProcessorList([proc]),
                                  num_beams=K, num_return_sequences=K, max_new_tokens=steps, do_sample=False,
                                  length_penalty=1.0, eos_token_id=tok.eos_token_id, pad_token_id=tok.pad_token_id,
                                  return_dict_in_generate=True, output_scores=True)

    def fused(share, cols=None, overlap=False):
        proc = ConstrainedLogitsProcessor(fn, K, pidx, suffix, tok.eos_token_id)
        return constrained_generate(model, prompts, proc, K, steps, attention_mask=mask, eos_token_id=tok.eos_token_id,
                                    pad_token_id=tok.pad_token_id, length_penalty=1.0, max_steps=steps,
                                    share_prefill=share, lm_head_slice=cols, lm_head_overlap=overlap)

    def timed(fn_, reps=3):
        fn_()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            out = fn_()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps * 1e3, out

    t_hf, a = timed(hf)
    t_hv, _ = timed(lambda: hf(True))
    t_hk, _ = timed(lambda: hf(True, True))
    t_f1, b = timed(lambda: fused(False))
    t_f2, c = timed(lambda: fused(True))
    live = a.sequences_scores > -1e8
    w = min(a.sequences.shape[1], c.sequences.shape[1])
    same = float((a.sequences[:, :w][live] == c.sequences[:, :w][live]).all(dim=1).float().mean())
    print(f"generate() + our processor in the list : {t_hf:8.1f} ms per batch")
    print(f"generate() + our processor, rows vouched normalised : {t_hv:8.1f} ms per batch  ({t_hf / t_hv:.2f}x)")
    print(f"generate() + our processor, rows vouched normalised, one kept result tensor : {t_hk:8.1f} ms per batch  "
          f"({t_hf / t_hk:.2f}x)")
    print(f"constrained_generate, prefill per beam : {t_f1:8.1f} ms per batch  ({t_hf / t_f1:.2f}x)")
    print(f"constrained_generate, shared prefill   : {t_f2:8.1f} ms per batch  ({t_hf / t_f2:.2f}x)")
    print(f"live beams with identical sequences (bf16 model, near-ties may flip): {same * 100:.1f} %")
    # LM head by column slices (the [rows, vocab] logits never written out); slice sized to ~128 MB
    cols = max(1024, (128 << 20) // (B * K * 2) // 1024 * 1024)
    for ov in (False, True):
        t_s, d = timed(lambda: fused(True, cols, ov))
        same = float((c.sequences[live] == d.sequences[live]).all(dim=1).float().mean())
        print(f"constrained_generate, shared prefill, LM head in slices of {cols} columns"
              f"{' on two streams' if ov else ''} : {t_s:8.1f} ms per batch  ({t_hf / t_s:.2f}x; "
              f"{same * 100:.1f} % of live beams equal to whole-logits decoding)")


if __name__ == "__main__":
    main()
