"""GPU: ``reuse_output=True`` -- the processor returns one kept [rows, vocab] tensor, and a call only
moves the finite positions (those of the previous call back to -inf, the new ones in) instead of
writing rows x vocab of -inf again (``grx_constrained_logsoftmax_reuse``).  What a call returns must
equal, bit for bit, what a processor with a new tensor per call returns -- over sequences of calls
whose rows move through the trie, on every instantiation of the dense kernel."""

import numpy as np
import pytest
import torch

from test_gpu_processors import FP32_TOL, _assert_dense_close, _dev, _make, _small_setup

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("case", ["count_rl", "count_eval", "count_seeded2", "count_midloop_reset", "fixed_hint"])
def test_kept_output_follows_the_reference_goldens_over_two_generations(case):
    data, meta, trie, fn = _small_setup()
    dev = _dev()
    proc = _make(case, fn, meta, reuse_output=True)
    info = meta["cases"][case]
    ptrs = set()
    for generation in range(2):
        if hasattr(proc, "reset"):
            proc.reset()
        for i in range(info["steps"]):
            ids = torch.from_numpy(data[f"{case}.ids.{i}"]).to(dev)
            scores = torch.from_numpy(data[f"{case}.scores.{i}"]).to(dev)
            out = proc(ids, scores)
            ptrs.add((tuple(out.shape), out.data_ptr()))
            _assert_dense_close(out, data[f"{case}.out.{i}"], FP32_TOL)
            if info["counts_after"][i] >= 0 and generation == 0:
                assert proc.count == info["counts_after"][i]
    shapes = {s for s, _ in ptrs}
    assert len(ptrs) == len(shapes) or len(shapes) > 1, "one tensor per shape"


def _walk_rows(rng, seqs, prefix, eos, R, T, depth_of_row, V):
    """Rows that end in prefix + the first depth tokens of a random item (depth per row)."""
    n_pre = len(prefix)
    rows = rng.integers(20, 90, size=(R, T))
    for r in range(R):
        item = seqs[int(rng.integers(0, len(seqs)))][n_pre:-1]
        d = min(int(depth_of_row[r]), len(item))
        body = list(prefix) + list(item[:d])
        rows[r, T - len(body):] = body
    return rows.astype(np.int64)


@pytest.mark.parametrize("V,R,dtype,normalised", [
    (2048, 64, torch.float32, False),     # 16-byte streams, one workgroup per row
    (2048, 64, torch.bfloat16, True),     # nothing streamed: the call is reset + gather
    (1001, 48, torch.float32, False),     # element-wise instantiation (rows not 16-byte aligned)
    (1001, 48, torch.bfloat16, True),
    (70016, 16, torch.float32, False),    # few long rows: a row is split over several workgroups
    (70016, 16, torch.bfloat16, False),
])
def test_kept_output_equals_a_new_tensor_per_call_while_rows_move_through_the_trie(V, R, dtype, normalised):
    from genrec_amd import _lib
    from genrec_amd.fixed_hint_logit_processor import FixedHintConstrainedLogitsProcessor
    from genrec_amd.trie import FlatTrie, create_prefix_allowed_tokens_fn

    data, meta, _, _ = _small_setup()
    dev = _dev()
    prefix, eos, K = meta["prefix"], meta["eos"], meta["K"]
    seqs = [s for s in data["trie_sequences"].tolist()]
    top = max(max(s) for s in seqs)
    assert top < V
    fn = create_prefix_allowed_tokens_fn(FlatTrie(seqs), prefix)
    if V > 60000:
        assert _lib.lib().grx_chunk_parts(R, V, _lib.torch_dtype_code(dtype)) >= 1  # (split factor is internal)
    kept = FixedHintConstrainedLogitsProcessor(fn, K, 3, prefix, eos, sync_errors=True, reuse_output=True,
                                               assume_normalized=normalised)
    new = FixedHintConstrainedLogitsProcessor(fn, K, 3, prefix, eos, sync_errors=True, assume_normalized=normalised)
    rng = np.random.default_rng(V + R)
    levels = max(len(s) for s in seqs) - len(prefix) - 1
    T = len(prefix) + levels + 6
    last_ptr = None
    for step in range(8):
        # every row at its own depth; now and then a finished row (ends in EOS: only EOS is allowed)
        depth = rng.integers(0, levels, size=R)
        ids = _walk_rows(rng, seqs, prefix, eos, R, T, depth, V)
        done = rng.random(R) < 0.15
        ids[done, -1] = eos
        x = (torch.randn(R, V, generator=torch.Generator().manual_seed(step)) * 4).to(dtype)
        if normalised:
            x = torch.log_softmax(x.float(), dim=-1).to(dtype)
        a = kept(torch.from_numpy(ids).to(dev), x.to(dev))
        b = new(torch.from_numpy(ids).to(dev), x.to(dev))
        assert last_ptr is None or a.data_ptr() == last_ptr
        last_ptr = a.data_ptr()
        fin = torch.isfinite(b)
        assert int(fin.sum()) >= R
        assert torch.equal(torch.isfinite(a), fin), step
        assert torch.equal(a[fin], b[fin]), step
        assert bool((a[~fin] == float("-inf")).all()), step
    # another row count: a new tensor, filled from scratch
    ids = _walk_rows(rng, seqs, prefix, eos, 2 * K, T, rng.integers(0, levels, size=2 * K), V)
    x = (torch.randn(2 * K, V) * 4).to(dtype)
    if normalised:
        x = torch.log_softmax(x.float(), dim=-1).to(dtype)
    a = kept(torch.from_numpy(ids).to(dev), x.to(dev))
    b = new(torch.from_numpy(ids).to(dev), x.to(dev))
    assert a.shape == b.shape and torch.equal(torch.isfinite(a), torch.isfinite(b))
    assert torch.equal(a[torch.isfinite(b)], b[torch.isfinite(b)])


def test_a_call_that_raises_leaves_the_kept_output_consistent():
    """Rows that leave the trie raise like the reference; the kept tensor still matches its record, so
    the next good call is right."""
    from genrec_amd.logit_processor import ConstrainedLogitsProcessor

    data, meta, trie, fn = _small_setup()
    dev = _dev()
    K, eos, V = meta["K"], meta["eos"], meta["V"]
    good = torch.from_numpy(data["count_rl.ids.1"]).to(dev)
    R = good.shape[0]
    bad = good.clone()
    bad[::3, -1] = 19  # no such child
    scores = torch.randn(R, V, device=dev)
    kept = ConstrainedLogitsProcessor(fn, K, 3, None, eos, reuse_output=True)
    new = ConstrainedLogitsProcessor(fn, K, 3, None, eos)
    for ids, ok in ((good, True), (bad, False), (good, True), (bad, False), (bad, False), (good, True)):
        kept.count = new.count = 1
        if ok:
            a, b = kept(ids, scores), new(ids, scores)
            assert torch.equal(a, b)
        else:
            with pytest.raises(AssertionError, match="No valid tokens"):
                kept(ids, scores)


def test_generate_with_a_kept_output_equals_generate_with_new_tensors():
    """transformers' beam loop reads the processor's result once per step and clones what it keeps
    (output_scores), so one kept tensor serves the whole generation."""
    from transformers import LogitsProcessorList, Qwen2Config, Qwen2ForCausalLM

    from genrec_amd.logit_processor import ConstrainedLogitsProcessor
    from genrec_amd.trie import FlatTrie, create_prefix_allowed_tokens_fn
    from helpers import load_npz

    dev = _dev()
    data, meta = load_npz("beams.npz")
    prefix, eos, pad, V = meta["prefix"], meta["eos"], meta["pad"], meta["V"]
    fn = create_prefix_allowed_tokens_fn(FlatTrie(data["trie_sequences"].tolist()), prefix)
    K = 8
    torch.manual_seed(5)
    cfg = Qwen2Config(vocab_size=V, hidden_size=32, intermediate_size=64, num_hidden_layers=2,
                      num_attention_heads=2, num_key_value_heads=2, max_position_embeddings=128,
                      eos_token_id=eos, pad_token_id=pad, bos_token_id=None, tie_word_embeddings=False)
    model = Qwen2ForCausalLM(cfg).eval()
    with torch.no_grad():
        for p in model.parameters():
            p.mul_(6.0)
    model = model.to(dev)
    g = torch.Generator().manual_seed(2)
    prompts = torch.tensor([torch.randint(20, 90, (7,), generator=g).tolist() + prefix for _ in range(3)], device=dev)
    outs = []
    for kw in ({}, {"reuse_output": True}, {"reuse_output": True, "assume_normalized": True}):
        proc = ConstrainedLogitsProcessor(fn, K, 3, prefix, eos, **kw)
        with torch.no_grad():
            outs.append(model.generate(prompts, attention_mask=torch.ones_like(prompts),
                                       logits_processor=LogitsProcessorList([proc]), num_beams=K,
                                       num_return_sequences=K, max_new_tokens=8, do_sample=False, length_penalty=1.0,
                                       eos_token_id=eos, pad_token_id=pad, output_scores=True,
                                       return_dict_in_generate=True))
    want = outs[0]
    live = want.sequences_scores > -1e8
    for got, tol in ((outs[1], 0.0), (outs[2], 2e-5)):
        assert torch.equal(got.sequences[live], want.sequences[live])
        assert float((got.sequences_scores[live] - want.sequences_scores[live]).abs().max()) <= tol
        assert len(got.scores) == len(want.scores)
        for a, b in zip(got.scores, want.scores):
            assert torch.equal(torch.isfinite(a), torch.isfinite(b))
            fin = torch.isfinite(b)
            assert float((a[fin] - b[fin]).abs().max()) <= tol
