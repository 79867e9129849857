"""GPU parity of the step over column slices of the logits (``grx_constrained_partial`` /
``grx_constrained_finish``): the same candidates as the whole-row call -- tokens bit for bit,
log-probabilities within 1e-5 of the float64 log-softmax -- against the reference-recorded goldens,
at full vocabulary width, and at the headline shape."""

import ctypes as C

import numpy as np
import pytest
import torch

from helpers import load_json_gz, load_npz, write_index_file
from test_gpu_processors import BF16_TOL, FP32_TOL, _dev, _make, _small_setup

pytestmark = pytest.mark.gpu


def _feed_all(proc, ids, scores, bounds, cand=None, order=None, **kw):
    from genrec_amd.logit_processor import Candidates

    if cand is None:
        cand = Candidates(ids.shape[0], max(proc.trie.max_fanout, 1), scores.device)
        cand.buf.fill_(-3)  # stale contents must not matter
    step = proc.chunked_step(ids, cand, scores.shape[1], bounds, scores.dtype, **kw)
    for i in (order if order is not None else range(len(bounds))):
        a, b = bounds[i]
        step.feed(a, scores[:, a:b])
    return step.finish()


@pytest.mark.parametrize("case", ["count_rl", "count_eval", "count_seeded2", "count_midloop_reset", "fixed_hint"])
def test_slices_give_the_reference_golden_results(case):
    data, meta, trie, fn = _small_setup()
    dev = _dev()
    proc = _make(case, fn, meta)
    whole = _make(case, fn, meta)
    info = meta["cases"][case]
    V = meta["V"]
    # ragged, unaligned slices, fed out of order (the opening call is the middle one)
    layouts = [([(0, 100), (100, 611), (611, V)], [1, 2, 0]), ([(0, 512), (512, V)], None), ([(0, V)], None)]
    for i in range(info["steps"]):
        ids = torch.from_numpy(data[f"{case}.ids.{i}"]).to(dev)
        scores = torch.from_numpy(data[f"{case}.scores.{i}"]).to(dev)
        want = data[f"{case}.out.{i}"]
        bounds, order = layouts[i % len(layouts)]
        cand = _feed_all(proc, ids, scores, bounds, order=order)
        ref = whole.candidates(ids, scores)
        count, tok, logp = cand.count.cpu().numpy(), cand.tokens.cpu().numpy(), cand.logp.cpu().numpy()
        assert torch.equal(cand.count, ref.count)
        lse = torch.logsumexp(scores.double(), dim=-1).cpu().numpy()
        assert np.max(np.abs(cand.logsumexp.cpu().numpy() - lse)) <= FP32_TOL
        for r in range(want.shape[0]):
            cols = np.nonzero(np.isfinite(want[r]))[0]
            assert count[r] == len(cols)
            assert np.array_equal(tok[r, : count[r]], cols)  # bit-exact token ids, ascending
            assert np.max(np.abs(logp[r, : count[r]] - want[r, cols])) <= FP32_TOL
        # rows are written in whole lines: zero padding up to the next multiple of 16 entries
        n_line = ((ref.count + 1 + 15) // 16) * 16
        e = torch.arange(cand.stride, device=dev)[None, :]
        pad = (e > ref.count[:, None]) & (e < n_line[:, None])
        assert not bool(cand.buf[pad].any())
        if info["counts_after"][i] >= 0:
            assert proc.count == info["counts_after"][i] == whole.count
    proc.check_errors()


def test_slices_at_full_vocabulary_width_fp32_and_bf16(tmp_path):
    """Instruments trie, V = 152,640, 19 slices of 8192 columns (the last one ragged)."""
    from genrec_amd.logit_processor import ConstrainedLogitsProcessor
    from genrec_amd.synth import ASSISTANT_PREFIX, SidTokenizer
    from genrec_amd.trie import build_trie_from_index, create_prefix_allowed_tokens_fn

    dev = _dev()
    data, meta = load_npz("full_vocab_rows.npz")
    path = write_index_file("Instruments", tmp_path)
    tok = SidTokenizer.from_index(path)
    trie, suffix, pidx = build_trie_from_index(path, tok, prefix=ASSISTANT_PREFIX)
    fn = create_prefix_allowed_tokens_fn(trie, suffix)
    K, V = meta["K"], meta["V"]
    bounds = [(a, min(a + 8192, V)) for a in range(0, V, 8192)]
    p32 = ConstrainedLogitsProcessor(fn, K, pidx, suffix, tok.eos_token_id)
    p16 = ConstrainedLogitsProcessor(fn, K, pidx, suffix, tok.eos_token_id)
    for step in range(meta["steps"]):
        ids = torch.from_numpy(data[f"ids.{step}"]).to(dev)
        gen = torch.Generator().manual_seed(meta["seed"] + step)
        logits = torch.randn(ids.shape[0], V, generator=gen) * meta["scale"]
        n = data[f"count.{step}"]
        for proc, x, want_lp, tol in ((p32, logits, data[f"logp.{step}"], FP32_TOL),
                                      (p16, logits.to(torch.bfloat16), data[f"logp_bf16.{step}"], FP32_TOL)):
            cand = _feed_all(proc, ids, x.to(dev), bounds)
            assert np.array_equal(cand.count.cpu().numpy(), n)
            exact = torch.log_softmax(x.double(), dim=-1).numpy()
            tk, lp = cand.tokens.cpu().numpy(), cand.logp.cpu().numpy()
            for r in range(ids.shape[0]):
                cols = data[f"tok.{step}"][r, : n[r]]
                assert np.array_equal(tk[r, : n[r]], cols)
                ex = exact[r, cols]
                ref_err = float(np.max(np.abs(want_lp[r, : n[r]] - ex)))
                assert np.max(np.abs(lp[r, : n[r]] - ex)) <= tol
                assert np.max(np.abs(lp[r, : n[r]] - want_lp[r, : n[r]])) <= tol + ref_err
    p32.check_errors()
    p16.check_errors()


@pytest.mark.parametrize("mode", ["count", "fixed_hint"])
def test_slices_equal_whole_rows_at_the_headline_shape(tmp_path, mode):
    """8192 rows x 152,640 bf16 columns: sliced == whole-row call (tokens bit for bit, log-probs and
    log-sum-exp within 1e-5), log-sum-exp of sampled rows against float64."""
    from genrec_amd.fixed_hint_logit_processor import FixedHintConstrainedLogitsProcessor
    from genrec_amd.logit_processor import ConstrainedLogitsProcessor
    from genrec_amd.synth import ASSISTANT_PREFIX, SidTokenizer, sample_decode_rows
    from genrec_amd.trie import build_trie_from_index, create_prefix_allowed_tokens_fn

    dev = _dev()
    path = write_index_file("Instruments", tmp_path)
    tok = SidTokenizer.from_index(path)
    trie, suffix, pidx = build_trie_from_index(path, tok, prefix=ASSISTANT_PREFIX)
    fn = create_prefix_allowed_tokens_fn(trie, suffix)
    B, K, V = 512, 16, tok.vocab_size_padded
    cls = ConstrainedLogitsProcessor if mode == "count" else FixedHintConstrainedLogitsProcessor
    sliced = cls(fn, K, pidx, suffix, tok.eos_token_id)
    whole = cls(fn, K, pidx, suffix, tok.eos_token_id)
    bounds = [(a, min(a + 8192, V)) for a in range(0, V, 8192)]
    gen = torch.Generator(device=dev).manual_seed(5)
    for depth in (0, 2):
        ids = torch.from_numpy(sample_decode_rows(trie, B, K, depth, prompt_len=24, seed=depth)).to(dev)
        logits = (torch.randn(B * K, V, generator=gen, device=dev) * 4).to(torch.bfloat16)
        if mode == "count":
            sliced.count = depth
            whole.count = depth
        got = _feed_all(sliced, ids, logits, bounds)
        want = whole.candidates(ids, logits)
        assert torch.equal(got.count, want.count) and int(want.count.min()) >= 1
        valid = torch.arange(want.cmax, device=dev)[None, :] < want.count[:, None]
        assert torch.equal(got.tokens[valid], want.tokens[valid])
        assert float((got.logp[valid] - want.logp[valid]).abs().max()) <= FP32_TOL
        assert float((got.logsumexp - want.logsumexp).abs().max()) <= FP32_TOL
        rows = torch.arange(0, B * K, 257, device=dev)
        exact = torch.logsumexp(logits[rows].double(), dim=-1)
        assert float((got.logsumexp[rows].double() - exact).abs().max()) <= FP32_TOL
    sliced.check_errors()


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
def test_few_rows_with_several_waves_per_slice_and_unaligned_slices(dtype):
    """16 rows: a slice is shared by several waves per row (one partial slot each); slice borders
    that break 16-byte alignment take the element-wise instantiation."""
    from oracle.processors import DictTrie, RefConstrainedLogitsProcessor

    from genrec_amd import _lib
    from genrec_amd.logit_processor import ConstrainedLogitsProcessor
    from genrec_amd.trie import FlatTrie, create_prefix_allowed_tokens_fn

    data, meta, _, _ = _small_setup()
    dev = _dev()
    seqs = data["trie_sequences"].tolist()
    fn = create_prefix_allowed_tokens_fn(FlatTrie(seqs), meta["prefix"])
    o_trie = DictTrie(seqs)
    K = meta["K"]
    ids = torch.from_numpy(data["count_rl.ids.1"])
    V = 70001
    torch.manual_seed(9)
    scores = (torch.randn(ids.shape[0], V) * 5).to(dtype)
    ref = RefConstrainedLogitsProcessor(o_trie.get, K, 3, meta["prefix"], meta["eos"])
    ref.count = 1
    want = ref(ids, scores.float())
    assert _lib.lib().grx_chunk_parts(ids.shape[0], 65536, _lib.torch_dtype_code(dtype)) > 1
    for bounds in ([(0, 65536), (65536, V)], [(0, 1001), (1001, 66537), (66537, V)]):
        proc = ConstrainedLogitsProcessor(fn, K, 3, meta["prefix"], meta["eos"])
        proc.count = 1
        cand = _feed_all(proc, ids.to(dev), scores.to(dev), bounds)
        for r in range(ids.shape[0]):
            cols = torch.nonzero(torch.isfinite(want[r])).flatten()
            assert cand.count[r].item() == cols.numel()
            assert torch.equal(cand.tokens[r, : cols.numel()].cpu().long(), cols)
            assert (cand.logp[r, : cols.numel()].cpu() - want[r, cols]).abs().max() <= FP32_TOL
        assert proc.count == 2
        proc.check_errors()


def test_sliced_step_reports_errors_and_rejects_what_it_cannot_do():
    from genrec_amd import _lib
    from genrec_amd.fixed_hint_logit_processor import FixedHintConstrainedLogitsProcessor
    from genrec_amd.logit_processor import Candidates, ConstrainedLogitsProcessor

    data, meta, trie, fn = _small_setup()
    dev = _dev()
    K, prefix, eos, V = meta["K"], meta["prefix"], meta["eos"], meta["V"]
    bad = torch.from_numpy(data["bad_rows"]).to(dev)
    scores = torch.randn(bad.shape[0], V, device=dev)
    bounds = [(0, 512), (512, V)]
    proc = ConstrainedLogitsProcessor(fn, K, 3, None, eos, sync_errors=True)
    proc.count = 1
    with pytest.raises(AssertionError, match="No valid tokens for prefix_ids"):
        _feed_all(proc, bad, scores, bounds)
    fixed = FixedHintConstrainedLogitsProcessor(fn, K, 3, prefix, eos, sync_errors=True)
    with pytest.raises(ValueError, match="Failed to find a valid trie prefix match"):
        _feed_all(fixed, bad, scores, bounds)
    # configuration the sliced step does not serve
    with pytest.raises(_lib.GenrecHipError, match="assume_normalized"):
        ConstrainedLogitsProcessor(fn, K, 3, None, eos, assume_normalized=True).chunked_step(
            bad, Candidates(bad.shape[0], 8, dev), V, bounds, torch.float32)
    ok = ConstrainedLogitsProcessor(fn, K, 3, None, eos)
    for wrong in ([(0, 512)], [(0, 512), (500, V)], [(0, 512), (512, V + 1)], [(1, V)]):
        with pytest.raises(ValueError, match="cover"):
            ok.chunked_step(bad, Candidates(bad.shape[0], 8, dev), V, wrong, torch.float32)
    step = ok.chunked_step(bad, Candidates(bad.shape[0], 8, dev), V, bounds, torch.float32)
    step.feed(0, scores[:, :512])
    with pytest.raises(RuntimeError, match="1 of 2"):
        step.finish()
    # the C ABI validates slice geometry before launching
    lib = _lib.lib()
    ids = torch.from_numpy(data["count_rl.ids.1"]).to(dev)
    R = ids.shape[0]
    cand = Candidates(R, 8, dev)
    cand.buf.zero_()
    ws = torch.zeros(int(lib.grx_workspace_bytes(R)), dtype=torch.uint8, device=dev)
    partials = torch.zeros(R * 4 * 2, device=dev)
    st = ok._state(0)
    rows = _lib.GrxRows()
    rows.input_ids, rows.ids_stride, rows.R, rows.T = ids.data_ptr(), ids.stride(0), R, ids.shape[1]
    sc = torch.zeros(R, 512, device=dev)
    rows.scores, rows.scores_stride, rows.V, rows.dtype = sc.data_ptr(), 512, 512, _lib.GRX_F32
    con = ok._constraint()

    def call(**kw):
        ch = _lib.GrxChunk()
        ch.v_begin, ch.v_total, ch.slot, ch.n_slots, ch.parts, ch.first = 0, V, 0, 4, 1, 1
        for k, v in kw.items():
            setattr(ch, k, v)
        return lib.grx_constrained_partial(trie.device_handle(dev), st.ptr, C.byref(rows), C.byref(con), C.byref(ch),
                                           partials.data_ptr(), cand.buf.data_ptr(), cand.stride, cand.cmax,
                                           ws.data_ptr(), ws.numel(), None)

    calls_before = ok._read_state(0)[3]
    assert call(v_begin=V - 100) == -1 and b"outside the vocabulary" in lib.grx_last_error()
    assert call(slot=4) == -1 and b"slots" in lib.grx_last_error()
    assert call(parts=0) == -1
    assert call(v_total=10) == -1
    assert lib.grx_constrained_finish(R, 0, partials.data_ptr(), cand.buf.data_ptr(), cand.stride, cand.cmax,
                                      ws.data_ptr(), ws.numel(), None) == -1
    # a penalised step without the buffer for the penalised tokens' logits, keep_top on slices
    con.repetition_penalty = 1.1
    assert call() == -1 and b"pen_logits" in lib.grx_last_error()
    assert lib.grx_constrained_finish_penalised(C.byref(rows), C.byref(con), V, 4, partials.data_ptr(), None, 0,
                                                cand.buf.data_ptr(), cand.stride, cand.cmax, ws.data_ptr(),
                                                ws.numel(), None) == -1
    con.repetition_penalty = 1.0
    con.keep_top = 4
    assert call() == -3 and b"keep_top" in lib.grx_last_error()
    con.keep_top = 0
    torch.cuda.synchronize()
    assert not bool(cand.buf.any()), "a refused call must not have written anything"
    assert ok._read_state(0)[3] == calls_before, "a refused call must not count as a processor call"


def test_sliced_first_step_on_one_row_per_group_and_plans_made_ahead():
    """The fused loop's use: the beam kernels plan the call ahead (``planned``), the first step sees one
    row per group.  Sliced and whole-row searches give identical beams."""
    from genrec_amd.beam import BeamSearchState
    from genrec_amd.fixed_hint_logit_processor import FixedHintConstrainedLogitsProcessor
    from genrec_amd.logit_processor import Candidates, ConstrainedLogitsProcessor

    data, meta, trie, fn = _small_setup()
    dev = _dev()
    K, prefix, eos, V = meta["K"], meta["prefix"], meta["eos"], meta["V"]
    prompts = torch.from_numpy(data["count_rl.ids.0"])[::K].contiguous().to(dev)
    B, T = prompts.shape
    torch.manual_seed(4)
    table = [torch.randn(B * K, V, device=dev) * 3 for _ in range(5)]
    bounds = [(0, 300), (300, 1000), (1000, V)]
    for cls in (ConstrainedLogitsProcessor, FixedHintConstrainedLogitsProcessor):
        results = []
        for sliced in (False, True):
            proc = cls(fn, K, 3, prefix, eos)
            if hasattr(proc, "reset"):
                proc.reset()
            state = BeamSearchState(prompts.repeat_interleave(K, dim=0), K, T + 6, eos, 0, planner=proc,
                                    first_step_shared=True)
            cmax = max(proc.trie.max_fanout, 1)
            first, cand = Candidates(B, cmax, dev), Candidates(B * K, cmax, dev)
            for step in range(4):
                shared = step == 0
                rows = state.prompt_rows() if shared else state.flat_running_sequences()
                logits = table[step][::K].contiguous() if shared else table[step]
                buf = first if shared else cand
                if sliced:
                    call = proc.chunked_step(rows, buf, V, bounds, logits.dtype, one_row_per_group=shared)
                    for a, b in bounds:
                        call.feed(a, logits[:, a:b])
                    call.finish()
                else:
                    proc.candidates_into(rows, logits, buf, one_row_per_group=shared)
                state.step(buf, V, planner=proc, shared_rows=shared)
            proc.check_errors()
            results.append(state.finalize())
        (s0, c0, b0), (s1, c1, b1) = results
        live = c0 > -1e8
        assert torch.equal(s0[live], s1[live]) and torch.equal(b0[live], b1[live])
        assert float((c0[live] - c1[live]).abs().max()) <= 1e-5


@pytest.mark.parametrize("kind,penalty", [("count", 1.0), ("fixed", 1.0), ("count", 1.3), ("fixed", 1.1)])
def test_generation_with_a_sliced_lm_head_equals_generation_from_whole_logits(kind, penalty):
    """constrained_generate(lm_head_slice=...) -- transformer body, LM head slice by slice into the
    sliced step, one stream or two -- returns the beams of the same loop fed whole logits, with or
    without the fused repetition penalty."""
    from transformers import Qwen2Config, Qwen2ForCausalLM

    from genrec_amd.fixed_hint_logit_processor import FixedHintConstrainedLogitsProcessor
    from genrec_amd.generate import constrained_generate
    from genrec_amd.logit_processor import ConstrainedLogitsProcessor
    from genrec_amd.trie import FlatTrie, create_prefix_allowed_tokens_fn

    dev = _dev()
    data, meta = load_npz("beams.npz")
    prefix, eos, pad, V = meta["prefix"], meta["eos"], meta["pad"], meta["V"]
    seqs = data["trie_sequences"].tolist()
    fn = create_prefix_allowed_tokens_fn(FlatTrie(seqs), prefix)
    K = 8
    torch.manual_seed(11)
    cfg = Qwen2Config(vocab_size=V, hidden_size=32, intermediate_size=64, num_hidden_layers=2,
                      num_attention_heads=2, num_key_value_heads=2, max_position_embeddings=128,
                      eos_token_id=eos, pad_token_id=pad, bos_token_id=None, tie_word_embeddings=False)
    model = Qwen2ForCausalLM(cfg).eval()
    with torch.no_grad():
        for p in model.parameters():
            p.mul_(6.0)
    model = model.to(dev)
    cls = ConstrainedLogitsProcessor if kind == "count" else FixedHintConstrainedLogitsProcessor
    g = torch.Generator().manual_seed(21)
    rows, lens = [], [9, 6, 11, 8]
    for n in lens:
        body = torch.randint(20, 90, (n,), generator=g).tolist()
        rows.append([pad] * (max(lens) - n) + body + prefix)
    prompts = torch.tensor(rows, device=dev)
    mask = (prompts != pad).long()
    kw = {} if penalty == 1.0 else {"repetition_penalty": penalty}
    want = constrained_generate(model, prompts, cls(fn, K, 3, prefix, eos, **kw), K, 8, attention_mask=mask,
                                eos_token_id=eos, pad_token_id=pad, length_penalty=0.0)
    live = want.sequences_scores > -1e8
    assert int(live.sum()) >= 12
    for cols, overlap, share in ((64, False, True), (64, True, True), (100, True, False), (V, False, True)):
        got = constrained_generate(model, prompts, cls(fn, K, 3, prefix, eos, **kw), K, 8, attention_mask=mask,
                                   eos_token_id=eos, pad_token_id=pad, length_penalty=0.0, share_prefill=share,
                                   lm_head_slice=cols, lm_head_overlap=overlap)
        assert got.steps == want.steps
        assert torch.equal(got.sequences[live], want.sequences[live]), (cols, overlap, share)
        assert torch.equal(got.beam_indices[live], want.beam_indices[live])
        assert float((got.sequences_scores[live] - want.sequences_scores[live]).abs().max()) <= 1e-4


def test_sliced_step_with_the_fused_repetition_penalty_matches_the_transformers_chain():
    """The recorded chain log_softmax -> transformers' RepetitionPenaltyLogitsProcessor -> reference
    processor (penalty.npz), through column slices: every slice leaves the raw logits of the rows' own
    tokens behind, the finishing kernel penalises and renormalises.  Tokens bit for bit; values within
    1e-5 of the float64 chain and of the whole-row kernel with the same penalty; bf16 slices against a
    float64 chain over the rounded logits."""
    import tempfile

    from oracle.processors import ref_repetition_penalty

    from genrec_amd.fixed_hint_logit_processor import FixedHintConstrainedLogitsProcessor
    from genrec_amd.logit_processor import ConstrainedLogitsProcessor
    from genrec_amd.synth import ASSISTANT_PREFIX, SidTokenizer, write_index_file
    from genrec_amd.trie import build_trie_from_index, create_prefix_allowed_tokens_fn
    from helpers import load_npz

    dev = _dev()
    data, meta = load_npz("penalty.npz")
    path = write_index_file("Instruments", tempfile.mkdtemp())
    tok = SidTokenizer.from_index(path)
    trie, suffix, pidx = build_trie_from_index(path, tok, prefix=ASSISTANT_PREFIX)
    fn = create_prefix_allowed_tokens_fn(trie, suffix)
    K, V, eos = meta["K"], meta["V"], meta["eos"]
    rng = np.random.default_rng(3)
    for ci, case in enumerate(meta["cases"]):
        ids = torch.from_numpy(data[f"{ci}.ids"])
        logits = torch.randn(ids.shape[0], V, generator=torch.Generator().manual_seed(case["seed"])) * meta["scale"]
        if case["peaked"]:
            logits[torch.arange(ids.shape[0]), ids[:, 7]] += 40.0
        kw = dict(repetition_penalty=case["penalty"], prompt_ignore_length=case["skip"], sync_errors=True)

        def make():
            if case["processor"] == "count":
                p = ConstrainedLogitsProcessor(fn, K, pidx, None, eos, **kw)
                p.count = case["step"]
                return p
            return FixedHintConstrainedLogitsProcessor(fn, K, pidx, suffix, eos, **kw)

        n = torch.from_numpy(data[f"{ci}.count"])
        valid = torch.arange(data[f"{ci}.tok"].shape[1])[None, :] < n[:, None]
        want_tok = torch.from_numpy(data[f"{ci}.tok"])[valid]
        exact = torch.from_numpy(data[f"{ci}.val64"])[valid]
        # slice borders: the SID tokens sit at the top of the vocabulary, the prompt tokens below 151,643,
        # so penalised tokens and candidates land in different slices, and some slices hold neither
        cuts = sorted({0, V, 65536, 131072, *rng.integers(151000, V, size=2).tolist()})
        bounds = list(zip(cuts[:-1], cuts[1:]))
        order = rng.permutation(len(bounds)).tolist()
        for dtype in (torch.float32, torch.bfloat16):
            x = logits.to(dtype).to(dev)
            whole = make().candidates(ids.to(dev), x)
            cand = _feed_all(make(), ids.to(dev), x, bounds, order=order)
            assert torch.equal(cand.count.cpu(), n)
            cvalid = torch.arange(cand.cmax)[None, :] < n[:, None]
            assert torch.equal(cand.tokens.cpu()[cvalid], want_tok)
            got = cand.logp.cpu()[cvalid]
            wvalid = torch.arange(whole.cmax)[None, :] < n[:, None]
            assert float((got - whole.logp.cpu()[wvalid]).abs().max()) <= FP32_TOL
            assert float((cand.logsumexp - whole.logsumexp).abs().max()) <= FP32_TOL
            if dtype == torch.float32:
                assert float((got.double() - exact).abs().max()) <= FP32_TOL
            else:
                lp64 = torch.log_softmax(logits.to(dtype).double(), dim=-1)
                e16 = torch.log_softmax(ref_repetition_penalty(ids, lp64, case["penalty"], case["skip"]), dim=-1)
                rows = torch.arange(ids.shape[0])[:, None].expand_as(cvalid)[cvalid]
                assert float((got.double() - e16[rows, want_tok.long()]).abs().max()) <= FP32_TOL
