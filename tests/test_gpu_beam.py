"""GPU parity of the fused beam step (through the C ABI) against traces of the
installed transformers ``generate`` driving the reference processors, and
against the CPU oracle of that loop."""

import numpy as np
import pytest
import torch

from helpers import load_npz

from genrec_amd.logit_processor import Candidates

pytestmark = pytest.mark.gpu

LIVE = -1.0e8      # slots below are -1e9 placeholders / -inf fillers (order unspecified in torch.topk)
SCORE_TOL = 1e-5


def _dev():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch.device("cuda", 0)


def _processor(name, fn, K, prefix, eos):
    from genrec_amd.fixed_hint_logit_processor import FixedHintConstrainedLogitsProcessor
    from genrec_amd.logit_processor import ConstrainedLogitsProcessor

    cls = FixedHintConstrainedLogitsProcessor if name.startswith("fixed") else ConstrainedLogitsProcessor
    return cls(fn, K, 3, prefix, eos, sync_errors=True)


@pytest.mark.parametrize("first_step_shared", [False, True])
@pytest.mark.parametrize("case", ["count_k4_lp1", "count_k16_lp0", "count_k16_lp1", "fixed_k8_lp1"])
def test_fused_loop_reproduces_hf_generate(case, first_step_shared):
    """Raw logits recorded from generate() go through processor.candidates + grx_beam_step;
    returned sequences / beam_indices are bit-exact and scores within 1e-5 on live beams.
    ``first_step_shared``: step 0 runs on one logit row per prompt instead of the K copies."""
    from genrec_amd.beam import constrained_beam_search
    from genrec_amd.trie import FlatTrie, create_prefix_allowed_tokens_fn

    dev = _dev()
    data, meta = load_npz("beams.npz")
    info = meta["cases"][case]
    K, prefix, eos, pad, V = info["K"], meta["prefix"], meta["eos"], meta["pad"], meta["V"]
    trie = FlatTrie(data["trie_sequences"].tolist())
    fn = create_prefix_allowed_tokens_fn(trie, prefix)
    proc = _processor(case, fn, K, prefix, eos)
    prompts = torch.from_numpy(data[f"{case}.prompts"]).to(dev)
    logits = [torch.from_numpy(data[f"{case}.logits.{t}"]).to(dev) for t in range(info["steps"])]
    want_scores_t = [data[f"{case}.scores.{t}"] for t in range(info["steps"])]
    seen = []

    def logits_fn(step, flat, next_parent):
        # the recorded logits are in HF's beam order of that step; parity on the live beams makes it ours too
        seen.append(flat.clone())
        if step == 0 and first_step_shared:
            assert flat.shape[0] == prompts.shape[0]
            return logits[0][::K].contiguous()
        return logits[step]

    state, seqs, scores, bidx = constrained_beam_search(
        logits_fn, proc, prompts, K, info["max_new_tokens"], eos, pad, info["length_penalty"], False,
        first_step_shared=first_step_shared,
    )
    assert len(seen) == info["steps"], "loop length differs from generate()"
    want_seqs = data[f"{case}.sequences"]
    want_scores = data[f"{case}.sequences_scores"]
    want_bidx = data[f"{case}.beam_indices"]
    scores = scores.cpu().numpy()
    live = want_scores > LIVE
    assert np.array_equal(live, scores > LIVE)
    assert np.array_equal(seqs.cpu().numpy()[live], want_seqs[live])          # item rankings, bit-exact
    assert np.array_equal(bidx.cpu().numpy()[live], want_bidx[live])          # beam parents, bit-exact
    assert np.max(np.abs(scores[live] - want_scores[live])) <= SCORE_TOL
    assert live.sum() >= live.size // 2


@pytest.mark.parametrize("case", ["count_k16_lp1", "fixed_k8_lp1"])
def test_beam_step_matches_oracle_stepwise(case):
    """Every state tensor after every step equals the CPU oracle's (live slots)."""
    from oracle.beam import RefBeamSearch
    from oracle.processors import compact_from_dense

    from genrec_amd.beam import BeamSearchState

    dev = _dev()
    data, meta = load_npz("beams.npz")
    info = meta["cases"][case]
    K, eos, pad, V = info["K"], meta["eos"], meta["pad"], meta["V"]
    prompts = torch.from_numpy(data[f"{case}.prompts"]).repeat_interleave(K, dim=0)
    max_len = prompts.shape[1] + info["max_new_tokens"]
    ref = RefBeamSearch(prompts, K, max_len, eos, pad, info["length_penalty"], False)
    for shared in (True, False):
        ref = RefBeamSearch(prompts, K, max_len, eos, pad, info["length_penalty"], False)
        st = BeamSearchState(prompts.to(dev), K, max_len, eos, pad, info["length_penalty"], False, shared_prompt=shared)
        for t in range(info["steps"]):
            dense = torch.from_numpy(data[f"{case}.scores.{t}"])  # processor output recorded from generate()
            cmax = int((dense > float("-inf")).sum(dim=1).max())
            cnt, tok, lp = compact_from_dense(dense, cmax)
            tr = ref.step(dense)
            res = st.step(Candidates.from_parts(cnt, tok, lp, dev), V)
            live = tr.topk_scores > LIVE
            assert torch.equal(res.topk_scores.cpu()[live], tr.topk_scores[live])   # same fp32 adds -> bit-exact
            assert torch.equal(res.topk_token.cpu().long()[live], tr.topk_token[live])
            assert torch.equal(res.topk_parent.cpu().long()[live], tr.topk_parent[live])
            assert torch.equal(res.topk_hit.cpu().bool()[live], tr.topk_hit[live])
            run_live = ref.running_beam_scores > LIVE
            assert torch.equal(st.running_beam_scores.cpu()[run_live], ref.running_beam_scores[run_live])
            assert torch.equal(st.running_sequences.cpu()[run_live], ref.running_sequences[run_live])
            assert torch.equal(res.next_parent.cpu().view(-1, K)[run_live].long(),
                               tr.next_parent.view(-1, K)[run_live].long())
            fin_live = ref.beam_scores > LIVE
            assert torch.equal(st.beam_scores.cpu()[fin_live], ref.beam_scores[fin_live])
            assert torch.equal(st.sequences.cpu()[fin_live], ref.sequences[fin_live])
            assert torch.equal(st.beam_indices.cpu()[fin_live], ref.beam_indices[fin_live])
            assert torch.equal(st.is_sent_finished.cpu()[fin_live], ref.is_sent_finished[fin_live])
            assert torch.equal(st.unsat.cpu().bool().view(-1, 1), ref.unsat)
        assert st.has_unfinished() == ref.has_unfinished()


def test_fused_loop_with_repetition_penalty_matches_the_oracle_loop():
    """evaluate.py decodes with repetition_penalty=1.1 (evaluate.py:218): generate() then runs
    RepetitionPenaltyLogitsProcessor ahead of the trie processor at every step.  The fused loop with
    repetition_penalty= on the processor returns the same rankings as the oracle loop with that chain."""
    from oracle.beam import ref_constrained_beam_search
    from oracle.processors import DictTrie, RefConstrainedLogitsProcessor, ref_repetition_penalty

    from genrec_amd.beam import constrained_beam_search
    from genrec_amd.logit_processor import ConstrainedLogitsProcessor
    from genrec_amd.trie import FlatTrie, create_prefix_allowed_tokens_fn

    dev = _dev()
    data, meta = load_npz("beams.npz")
    case = "count_k16_lp0"
    info = meta["cases"][case]
    K, prefix, eos, pad = info["K"], meta["prefix"], meta["eos"], meta["pad"]
    seqs = data["trie_sequences"].tolist()
    trie = FlatTrie(seqs)
    fn = create_prefix_allowed_tokens_fn(trie, prefix)
    prompts = torch.from_numpy(data[f"{case}.prompts"])
    # history: the prompt repeats tokens that the trie can propose
    sid = sorted({t for s in seqs for t in s[len(prefix):-1]})
    g = torch.Generator().manual_seed(5)
    prompts[:, :6] = torch.tensor(sid)[torch.randint(0, len(sid), (prompts.shape[0], 6), generator=g)]
    logits = [torch.from_numpy(data[f"{case}.logits.{t}"]) for t in range(info["steps"])]
    penalty = 1.1

    class Chain:
        def __init__(self):
            self.inner = RefConstrainedLogitsProcessor(DictTrie(seqs).get, K, 3, prefix, eos)

        def reset(self):
            self.inner.reset()

        def __call__(self, ids, lp):
            return self.inner(ids, ref_repetition_penalty(ids, lp, penalty))

    ref_state, _ = ref_constrained_beam_search(lambda step, flat: logits[step], Chain(), prompts, K,
                                               info["max_new_tokens"], eos, pad, info["length_penalty"])
    want_seqs, want_scores, want_bidx = ref_state.finalize()
    proc = ConstrainedLogitsProcessor(fn, K, 3, prefix, eos, sync_errors=True, repetition_penalty=penalty)
    dev_logits = [x.to(dev) for x in logits]
    _, got_seqs, got_scores, got_bidx = constrained_beam_search(
        lambda step, flat, parent: dev_logits[step], proc, prompts.to(dev), K, info["max_new_tokens"], eos, pad,
        info["length_penalty"])
    live = want_scores > LIVE
    assert torch.equal(live, got_scores.cpu() > LIVE)
    assert torch.equal(got_seqs.cpu()[live], want_seqs[live])
    assert torch.equal(got_bidx.cpu()[live].long(), want_bidx[live].long())
    assert float((got_scores.cpu()[live] - want_scores[live]).abs().max()) <= SCORE_TOL
    # the penalty changes the outcome (otherwise this test would not see it)
    plain = ConstrainedLogitsProcessor(fn, K, 3, prefix, eos, sync_errors=True)
    _, _, plain_scores, _ = constrained_beam_search(
        lambda step, flat, parent: dev_logits[step], plain, prompts.to(dev), K, info["max_new_tokens"], eos, pad,
        info["length_penalty"])
    assert not torch.equal(plain_scores.cpu()[live], got_scores.cpu()[live])


def test_fused_loop_on_a_ragged_trie_in_every_stopping_mode():
    """generate() traces on a trie whose items have 1..4 levels, early_stopping False / True / "never",
    length_penalty 0 / 1 / 2 / -0.5: the fused loop stops after the same number of steps and returns the
    same sequences and beam parents (bit-exact) and scores (1e-5) on live beams."""
    from helpers import ragged_beam_cases

    from genrec_amd.beam import constrained_beam_search
    from genrec_amd.fixed_hint_logit_processor import FixedHintConstrainedLogitsProcessor
    from genrec_amd.trie import FlatTrie, create_prefix_allowed_tokens_fn

    dev = _dev()
    data, meta, seqs = ragged_beam_cases()
    prefix, eos, pad = meta["prefix"], meta["eos"], meta["pad"]
    fn = create_prefix_allowed_tokens_fn(FlatTrie(seqs), prefix)
    prompts = torch.from_numpy(data["prompts"]).to(dev)
    for name, info in meta["cases"].items():
        K = info["K"]
        proc = FixedHintConstrainedLogitsProcessor(fn, K, 3, prefix, eos, sync_errors=True)
        logits = [torch.from_numpy(data[f"{name}.logits.{t}"]).to(dev) for t in range(info["steps"])]
        calls = []

        def logits_fn(step, flat, parent):
            calls.append(step)
            return logits[step]

        _, got_seqs, got_scores, got_bidx = constrained_beam_search(
            logits_fn, proc, prompts, K, info["max_new_tokens"], eos, pad, info["length_penalty"],
            info["early_stopping"])
        assert len(calls) == info["steps"], f"{name}: stopped after {len(calls)} steps, generate() after {info['steps']}"
        want_scores = torch.from_numpy(data[f"{name}.sequences_scores"])
        want_seqs = torch.from_numpy(data[f"{name}.sequences"])
        want_bidx = torch.from_numpy(data[f"{name}.beam_indices"])
        live = want_scores > LIVE
        got_seqs, got_scores, got_bidx = got_seqs.cpu(), got_scores.cpu(), got_bidx.cpu()
        assert torch.equal(live, got_scores > LIVE), name
        w, bw = min(got_seqs.shape[1], want_seqs.shape[1]), min(got_bidx.shape[1], want_bidx.shape[1])
        assert torch.equal(got_seqs[:, :w][live], want_seqs[:, :w][live]), name
        assert torch.equal(got_bidx[:, :bw].long()[live], want_bidx[:, :bw].long()[live]), name
        assert float((got_scores[live] - want_scores[live]).abs().max()) <= SCORE_TOL, name
        # generate() fills finished rows with `pad_token_id or eos_token_id` (utils.py:3319): with the
        # recorded pad id 0 the rows trail EOS, not 0 -- and so must ours
        assert pad == 0 and eos != 0
        short = live & (want_seqs[:, w - 1] == eos) & (want_seqs[:, w - 2] == eos)
        assert bool(short.any()) and not bool((want_seqs[live][:, prompts.shape[1]:] == pad).any()), name
        assert not bool((got_seqs[:, :w][live][:, prompts.shape[1]:] == pad).any()), name


@pytest.mark.parametrize("vouched", [False, True])
@pytest.mark.parametrize("kind", ["count", "fixed"])
def test_processors_drop_into_transformers_generate_on_the_gpu(kind, vouched):
    """The unmodified HF loop on cuda:0 (tiny random Qwen2) with our processor in its
    LogitsProcessorList, as evaluate.py:363-371 / MIMIGenRec.py:79 call it, against the same loop with
    the CPU restatement of the reference processor: same sequences and beam indices, scores within
    1e-5, over two consecutive generate() calls (counter reset between them as evaluate.py:358-361 does).
    ``vouched``: with ``assume_normalized=True`` -- generate() hands the lone processor log-softmax values,
    so skipping the read pass must give the same beams."""
    from transformers import LogitsProcessor, LogitsProcessorList, Qwen2Config, Qwen2ForCausalLM

    from oracle.processors import (DictTrie, RefConstrainedLogitsProcessor,
                                   RefFixedHintConstrainedLogitsProcessor)

    from genrec_amd.fixed_hint_logit_processor import FixedHintConstrainedLogitsProcessor
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

    class OnCpu(LogitsProcessor):
        """The oracle works on CPU tensors; the HF loop hands over device tensors."""

        def __init__(self, inner):
            self.inner = inner

        @property
        def count(self):
            return self.inner.count

        @count.setter
        def count(self, v):
            self.inner.count = v

        def __call__(self, input_ids, scores):
            return self.inner(input_ids.cpu(), scores.float().cpu()).to(scores.device)

    get = DictTrie(seqs).get
    if kind == "count":
        ours = ConstrainedLogitsProcessor(fn, K, 3, None, eos, assume_normalized=vouched)  # evaluate.py:283-288: no prefix_ids
        ref = OnCpu(RefConstrainedLogitsProcessor(get, K, 3, None, eos))
    else:
        ours = FixedHintConstrainedLogitsProcessor(fn, K, 3, prefix, eos, assume_normalized=vouched)
        ref = OnCpu(RefFixedHintConstrainedLogitsProcessor(get, K, 3, prefix, eos))
    g = torch.Generator().manual_seed(3)
    for round_ in range(2):
        prompts = torch.cat([torch.randint(20, 90, (3, 7), generator=g), torch.tensor([prefix] * 3)], dim=1).to(dev)
        outs = []
        for proc in (ours, ref):
            if hasattr(proc, "count"):
                proc.count = 0
            with torch.no_grad():
                outs.append(model.generate(
                    prompts, attention_mask=torch.ones_like(prompts), logits_processor=LogitsProcessorList([proc]),
                    num_beams=K, num_return_sequences=K, max_new_tokens=8, do_sample=False, length_penalty=0.0,
                    eos_token_id=eos, pad_token_id=pad, output_scores=True, return_dict_in_generate=True))
        a, b = outs
        live = b.sequences_scores > LIVE
        assert int(live.sum()) >= 3 * K // 2
        assert torch.equal(a.sequences[live], b.sequences[live])
        assert torch.equal(a.beam_indices[live], b.beam_indices[live])
        assert float((a.sequences_scores[live] - b.sequences_scores[live]).abs().max()) <= SCORE_TOL
        assert len(a.scores) == len(b.scores)
        for sa, sb in zip(a.scores, b.scores):  # the processor outputs HF recorded at every step
            fin = torch.isfinite(sb)
            assert torch.equal(torch.isfinite(sa), fin)
            assert float((sa[fin] - sb[fin]).abs().max()) <= SCORE_TOL
    ours.check_errors()


@pytest.mark.parametrize("kind", ["count", "count_none", "fixed"])
def test_plans_written_by_the_beam_kernels_equal_the_planning_kernel(kind):
    """Fused loop: beam init / beam step also prepare the next processor call (fixed-hint: the row
    plans; count-based: the prefix_ids match reduction and the counter bookkeeping).  In lockstep with
    a second processor that prepares for itself, the candidates -- and in the fixed-hint mode the plan
    words in the two workspaces -- are identical at every step, over two generations."""
    from genrec_amd.beam import BeamSearchState
    from genrec_amd.fixed_hint_logit_processor import FixedHintConstrainedLogitsProcessor
    from genrec_amd.logit_processor import ConstrainedLogitsProcessor
    from genrec_amd.trie import FlatTrie, create_prefix_allowed_tokens_fn

    dev = _dev()
    data, meta = load_npz("beams.npz")
    prefix, eos, pad, V = meta["prefix"], meta["eos"], meta["pad"], meta["V"]
    trie = FlatTrie(data["trie_sequences"].tolist())
    fn = create_prefix_allowed_tokens_fn(trie, prefix)
    K, B, steps = 8, 5, 4

    def make():
        if kind == "fixed":
            return FixedHintConstrainedLogitsProcessor(fn, K, 3, prefix, eos, sync_errors=False)
        return ConstrainedLogitsProcessor(fn, K, 3, prefix if kind == "count" else None, eos, sync_errors=False)

    ahead, alone = make(), make()
    g = torch.Generator().manual_seed(9)
    plan_off = 512 * 4 + 2048 * 16  # arrival counters of split rows, their partial pairs, then the plans  # counters + partials come first in the workspace

    def plans(proc, R):
        ws = next(iter(proc._workspaces.values()))
        return ws[plan_off : plan_off + 16 * R].view(torch.int32).view(R, 4).clone()

    for generation in range(2):
        prompts = torch.cat([torch.randint(20, 90, (B, 5), generator=g), torch.tensor([prefix] * B)], dim=1)
        rows = prompts.repeat_interleave(K, dim=0).to(dev)
        for p in (ahead, alone):
            # with prefix_ids the second generation relies on the reference's own reset: its rows end in the
            # prefix, which drops the shared counter (4 by then) back to 0 (logit_processor.py:39-40)
            if hasattr(p, "reset") and (kind != "count" or generation == 0):
                p.count = 0
        st_a = BeamSearchState(rows, K, rows.shape[1] + steps + 1, eos, pad, 1.0, False, planner=ahead)
        st_b = BeamSearchState(rows, K, rows.shape[1] + steps + 1, eos, pad, 1.0, False)
        cand_a = Candidates(B * K, trie.max_fanout, dev)
        cand_b = Candidates(B * K, trie.max_fanout, dev)
        for t in range(steps):
            logits = (torch.randn(B * K, V, generator=g) * 3).to(dev)
            assert ahead._planned is not None
            ahead.candidates_into(st_a.flat_running_sequences(), logits, cand_a)
            alone.candidates_into(st_b.flat_running_sequences(), logits, cand_b)
            if kind == "fixed":  # count-based calls keep no row plans: their rows are walked by the streaming kernel
                pa, pb = plans(ahead, B * K), plans(alone, B * K)
                used = pb[:, 0] >= 0
                assert torch.equal(pa[:, 0], pb[:, 0]), f"step {t}: nodes differ"
                assert torch.equal(pa[used][:, :3], pb[used][:, :3]), f"step {t}: child ranges differ"
            assert torch.equal(cand_a.count, cand_b.count)
            valid = torch.arange(cand_a.cmax, device=dev)[None, :] < cand_a.count[:, None]
            assert torch.equal(cand_a.tokens[valid], cand_b.tokens[valid])
            assert torch.equal(cand_a.logp[valid], cand_b.logp[valid])
            st_a.step(cand_a, V, planner=ahead)
            st_b.step(cand_b, V)
            assert torch.equal(st_a.running_sequences, st_b.running_sequences)
        ahead._planned = None  # the plans made by the last step are for rows nobody will process
    if kind != "fixed":
        assert ahead.count == alone.count
    ahead.check_errors()
    alone.check_errors()


@pytest.mark.parametrize("kind,penalty", [("count", 1.0), ("fixed", 1.0), ("count", 1.1)])
def test_constrained_generate_matches_transformers_generate(kind, penalty):
    """constrained_generate (our loop around a HF model: shared prefill, cache re-ordering, attention
    mask and position ids for left-padded prompts) against model.generate() with the same processor in
    its LogitsProcessorList: same sequences and beam indices, scores within 1e-4, with and without a
    repetition penalty, with the prefill shared across beams or repeated per beam."""
    from transformers import LogitsProcessorList, Qwen2Config, Qwen2ForCausalLM

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

    def make(**kw):
        if kind == "count":
            return ConstrainedLogitsProcessor(fn, K, 3, prefix, eos, **kw)
        return FixedHintConstrainedLogitsProcessor(fn, K, 3, prefix, eos, **kw)

    g = torch.Generator().manual_seed(21)
    sid = sorted({t for s in seqs for t in s[len(prefix):-1]})
    rows, lens = [], [9, 6, 11]
    for n in lens:  # left-padded prompts of different lengths, with trie tokens in their history
        body = torch.randint(20, 90, (n,), generator=g).tolist()
        body[: n // 2] = [sid[int(i)] for i in torch.randint(0, len(sid), (n // 2,), generator=g)]
        rows.append([pad] * (max(lens) - n) + body + prefix)
    prompts = torch.tensor(rows, device=dev)
    mask = (prompts != pad).long()
    with torch.no_grad():
        want = model.generate(prompts, attention_mask=mask, logits_processor=LogitsProcessorList([make()]),
                              num_beams=K, num_return_sequences=K, max_new_tokens=8, do_sample=False,
                              length_penalty=0.0, repetition_penalty=penalty, eos_token_id=eos, pad_token_id=pad,
                              output_scores=True, return_dict_in_generate=True)
    live = want.sequences_scores > LIVE
    assert int(live.sum()) >= 12
    for share in (True, False):
        got = constrained_generate(model, prompts, make(repetition_penalty=penalty), K, 8, attention_mask=mask,
                                   eos_token_id=eos, pad_token_id=pad, length_penalty=0.0, share_prefill=share)
        assert got.steps == len(want.scores)
        w = min(got.sequences.shape[1], want.sequences.shape[1])
        assert torch.equal(got.sequences[:, :w][live], want.sequences[:, :w][live]), f"share_prefill={share}"
        bw = min(got.beam_indices.shape[1], want.beam_indices.shape[1])
        assert torch.equal(got.beam_indices[:, :bw][live].long(), want.beam_indices[:, :bw][live].long())
        assert float((got.sequences_scores[live] - want.sequences_scores[live]).abs().max()) <= 1e-4
    if kind == "count" and penalty == 1.0:
        # the trainer-facing wrapper (MIMIGenRec.py:50-79): every prompt arrives K times, one token tensor comes back
        from genrec_amd.generate import make_fused_generate

        gen = make_fused_generate(model, make(), num_beams=K, max_completion_length=8, length_penalty=0.0)
        seqs_out = gen(input_ids=prompts.repeat_interleave(K, dim=0), attention_mask=mask.repeat_interleave(K, dim=0))
        w = min(seqs_out.shape[1], want.sequences.shape[1])
        assert seqs_out.shape[0] == prompts.shape[0] * K
        assert torch.equal(seqs_out[:, :w][live], want.sequences[:, :w][live])


def test_beam_step_large_random_against_oracle():
    """Synthetic candidates at rollout scale (K=16 and K=50, fan-out up to 242): exact top-2K order."""
    from oracle.beam import RefBeamSearch

    from genrec_amd.beam import BeamSearchState

    dev = _dev()
    torch.manual_seed(0)
    for K, B, cmax, V in ((16, 37, 242, 152704), (50, 5, 242, 152704), (16, 9, 700, 200000), (3, 11, 5, 97)):
        R = B * K
        prompts = torch.randint(5, 90, (B, 6)).repeat_interleave(K, dim=0)
        eos = 1
        ref = RefBeamSearch(prompts, K, 6 + 4, eos, 0, 1.0, False)
        st = BeamSearchState(prompts.to(dev), K, 6 + 4, eos, 0, 1.0, False, shared_prompt=True)
        for step in range(3):
            cnt = torch.randint(1, cmax + 1, (R,), dtype=torch.int32)
            if step == 2:
                cnt[::3] = 1
            tok = torch.zeros((R, cmax), dtype=torch.int32)
            lp = torch.zeros((R, cmax), dtype=torch.float32)
            dense = torch.full((R, V), float("-inf"))
            for r in range(R):
                n = int(cnt[r])
                cols = torch.sort(torch.randperm(V - 2)[:n] + 2).values
                if step == 2 and r % 3 == 0:
                    cols = torch.tensor([eos])
                vals = -torch.rand(n) * 20
                tok[r, :n] = cols.int()
                lp[r, :n] = vals
                dense[r, cols] = vals
            tr = ref.step(dense)
            res = st.step(Candidates.from_parts(cnt, tok, lp, dev), V)
            live = tr.topk_scores > LIVE
            assert torch.equal(res.topk_scores.cpu()[live], tr.topk_scores[live])
            assert torch.equal(res.topk_token.cpu().long()[live], tr.topk_token[live])
            assert torch.equal(res.topk_parent.cpu().long()[live], tr.topk_parent[live])
            run_live = ref.running_beam_scores > LIVE
            assert torch.equal(st.running_sequences.cpu()[run_live], ref.running_sequences[run_live])
            fin_live = ref.beam_scores > LIVE
            assert torch.equal(st.sequences.cpu()[fin_live], ref.sequences[fin_live])
            assert torch.equal(st.beam_scores.cpu()[fin_live], ref.beam_scores[fin_live])


@pytest.mark.parametrize("K,cmax", [(17, 256), (64, 256), (64, 257), (65, 100), (33, 1), (50, 164), (64, 64), (20, 300)])
def test_beam_step_at_the_edges_of_the_kernel_instantiations(K, cmax):
    """Groups of 17 .. 64 beams with rows of at most 256 candidates run the 1024-thread instantiation that holds
    its entries in registers; one beam or one candidate more takes the two-pass route.  Both against the oracle,
    three steps, the last with rows that end."""
    from oracle.beam import RefBeamSearch

    from genrec_amd.beam import BeamSearchState

    dev = _dev()
    g = torch.Generator().manual_seed(1000 * K + cmax)
    B, V, eos = 3, 3000, 1
    R = B * K
    prompts = torch.randint(5, 90, (B, 5), generator=g).repeat_interleave(K, dim=0)
    ref = RefBeamSearch(prompts, K, 5 + 4, eos, 0, 1.0, False)
    st = BeamSearchState(prompts.to(dev), K, 5 + 4, eos, 0, 1.0, False, shared_prompt=True)
    for step in range(3):
        cnt = torch.randint(1, cmax + 1, (R,), generator=g, dtype=torch.int32)
        cnt[1::5] = cmax  # full rows as well
        tok = torch.zeros((R, cmax), dtype=torch.int32)
        lp = torch.zeros((R, cmax), dtype=torch.float32)
        dense = torch.full((R, V), float("-inf"))
        for r in range(R):
            n = int(cnt[r])
            cols = torch.sort(torch.randperm(V - 2, generator=g)[:n] + 2).values
            if step == 2 and r % 3 == 0:
                cols = torch.tensor([eos])
                n = 1
                cnt[r] = 1
            vals = -torch.rand(n, generator=g) * 20
            tok[r, :n] = cols.int()
            lp[r, :n] = vals
            dense[r, cols] = vals
        tr = ref.step(dense)
        res = st.step(Candidates.from_parts(cnt, tok, lp, dev), V)
        live = tr.topk_scores > LIVE
        assert torch.equal(res.topk_scores.cpu()[live], tr.topk_scores[live]), step
        assert torch.equal(res.topk_token.cpu().long()[live], tr.topk_token[live]), step
        assert torch.equal(res.topk_parent.cpu().long()[live], tr.topk_parent[live]), step
        run_live = ref.running_beam_scores > LIVE
        assert torch.equal(st.running_sequences.cpu()[run_live], ref.running_sequences[run_live]), step
        assert torch.equal(st.running_beam_scores.cpu()[run_live], ref.running_beam_scores[run_live]), step
        fin_live = ref.beam_scores > LIVE
        assert torch.equal(st.sequences.cpu()[fin_live], ref.sequences[fin_live]), step
        assert torch.equal(st.beam_scores.cpu()[fin_live], ref.beam_scores[fin_live]), step


def test_tie_order_is_canonical():
    """Equal scores are ordered by beam * V + token ascending (documented rule; torch.topk leaves it open)."""
    from genrec_amd.beam import BeamSearchState

    dev = _dev()
    K, B, V = 4, 1, 100
    prompts = torch.zeros((K, 2), dtype=torch.int64, device=dev)
    st = BeamSearchState(prompts, K, 6, 1, 0, 1.0, False)
    st.run_score[st.cur].fill_(0.0)  # all beams live with equal score
    cnt = torch.full((K,), 3, dtype=torch.int32, device=dev)
    tok = torch.tensor([[10, 20, 30]] * K, dtype=torch.int32, device=dev)
    lp = torch.full((K, 3), -1.0, device=dev)
    res = st.step(Candidates.from_parts(cnt, tok, lp, dev), V)
    assert res.topk_parent.cpu().tolist() == [[0, 0, 0, 1, 1, 1, 2, 2]]
    assert res.topk_token.cpu().tolist() == [[10, 20, 30, 10, 20, 30, 10, 20]]


@pytest.mark.parametrize("kind", ["count", "count_none", "fixed"])
def test_first_step_from_one_row_per_group_equals_k_copies(kind):
    """The first step of a search whose K beams are copies of one prompt: a processor call on one row
    per group + a beam step with shared candidate rows gives, bit for bit and in every slot, what K
    identical logit rows give -- including the counter, over two generations (the second relies on
    the reference's own mid-call reset when prefix_ids is given)."""
    from genrec_amd.beam import BeamSearchState
    from genrec_amd.fixed_hint_logit_processor import FixedHintConstrainedLogitsProcessor
    from genrec_amd.logit_processor import ConstrainedLogitsProcessor
    from genrec_amd.trie import FlatTrie, create_prefix_allowed_tokens_fn

    dev = _dev()
    data, meta = load_npz("beams.npz")
    prefix, eos, pad, V = meta["prefix"], meta["eos"], meta["pad"], meta["V"]
    trie = FlatTrie(data["trie_sequences"].tolist())
    fn = create_prefix_allowed_tokens_fn(trie, prefix)
    K, B, steps = 8, 5, 3

    def make():
        if kind == "fixed":
            return FixedHintConstrainedLogitsProcessor(fn, K, 3, prefix, eos, sync_errors=False)
        return ConstrainedLogitsProcessor(fn, K, 3, prefix if kind == "count" else None, eos, sync_errors=False)

    one, full = make(), make()
    g = torch.Generator().manual_seed(21)
    for generation in range(2):
        prompts = torch.cat([torch.randint(20, 90, (B, 5), generator=g), torch.tensor([prefix] * B)], dim=1)
        rows = prompts.repeat_interleave(K, dim=0).to(dev)
        for p in (one, full):
            if hasattr(p, "reset") and (kind != "count" or generation == 0):
                p.count = 0
        L = rows.shape[1] + steps + 1
        st_one = BeamSearchState(rows, K, L, eos, pad, 1.0, False, planner=one, first_step_shared=True)
        st_full = BeamSearchState(rows, K, L, eos, pad, 1.0, False, planner=full)
        assert one._planned is not None and one._planned[1] == B  # B rows were planned, not B * K
        first = Candidates(B, trie.max_fanout, dev)
        cand_one = Candidates(B * K, trie.max_fanout, dev)
        cand_full = Candidates(B * K, trie.max_fanout, dev)
        for t in range(steps):
            if t == 0:
                logits_b = (torch.randn(B, V, generator=g) * 3).to(dev)
                one.candidates_into(st_one.prompt_rows(), logits_b, first, one_row_per_group=True)
                full.candidates_into(st_full.flat_running_sequences(), logits_b.repeat_interleave(K, dim=0), cand_full)
                assert torch.equal(first.buf, cand_full.buf[::K])
                res_one = st_one.step(first, V, planner=one, shared_rows=True)
            else:
                logits = (torch.randn(B * K, V, generator=g) * 3).to(dev)
                one.candidates_into(st_one.flat_running_sequences(), logits, cand_one)
                full.candidates_into(st_full.flat_running_sequences(), logits, cand_full)
                assert torch.equal(cand_one.count, cand_full.count)
                res_one = st_one.step(cand_one, V, planner=one)
            res_full = st_full.step(cand_full, V, planner=full)
            for name in ("topk_scores", "topk_parent", "topk_token", "topk_hit", "next_parent"):
                assert torch.equal(getattr(res_one, name), getattr(res_full, name)), (generation, t, name)
            assert torch.equal(st_one.running_sequences, st_full.running_sequences)
            assert torch.equal(st_one.running_beam_scores, st_full.running_beam_scores)
            assert torch.equal(st_one.sequences, st_full.sequences)
            assert torch.equal(st_one.beam_scores, st_full.beam_scores)
        for p in (one, full):
            p._planned = None
    if kind != "fixed":
        assert one.count == full.count
    one.check_errors()
    full.check_errors()


@pytest.mark.parametrize("K,cmax", [(16, 256), (40, 256)])
def test_adversarial_score_layout_takes_the_general_selection_route(K, cmax):
    """The fast top-2K selection ranks only the candidates above a threshold taken from per-thread
    maxima; when nearly all large scores sit with few threads, more candidates pass the threshold than
    the pairwise ranking holds and the kernel falls back to plain repeated max extraction (16 beams:
    a quarter of the lanes own every large score, > 1000 survivors at the second step).  With 40 beams
    the rotated dealing spreads the same layout over all threads.  Same answer either way."""
    from oracle.beam import RefBeamSearch

    from genrec_amd.beam import BeamSearchState

    dev = _dev()
    B, V, eos = 2, 4000, 1
    R = B * K
    g = torch.Generator().manual_seed(5)
    prompts = torch.randint(5, 90, (B, 4), generator=g).repeat_interleave(K, dim=0)
    ref = RefBeamSearch(prompts, K, 4 + 3, eos, 0, 1.0, False)
    st = BeamSearchState(prompts.to(dev), K, 4 + 3, eos, 0, 1.0, False)
    for step in range(2):
        cnt = torch.full((R,), cmax, dtype=torch.int32)
        tok = (torch.arange(cmax, dtype=torch.int32)[None, :] * 3 + 2).repeat(R, 1)  # ascending, no eos
        col = torch.arange(cmax)[None, :].expand(R, cmax)
        # entries in a quarter of the columns (mod 64) are large, the rest small
        high = (col % 64) < 16
        lp = torch.where(high, -torch.rand(R, cmax, generator=g), -10.0 - torch.rand(R, cmax, generator=g))
        dense = torch.full((R, V), float("-inf"))
        dense.scatter_(1, tok.long(), lp)
        tr = ref.step(dense)
        res = st.step(Candidates.from_parts(cnt, tok, lp, dev), V)
        live = tr.topk_scores > LIVE
        assert torch.equal(res.topk_scores.cpu()[live], tr.topk_scores[live])
        assert torch.equal(res.topk_token.cpu().long()[live], tr.topk_token[live])
        assert torch.equal(res.topk_parent.cpu().long()[live], tr.topk_parent[live])
        run_live = ref.running_beam_scores > LIVE
        assert torch.equal(st.running_sequences.cpu()[run_live], ref.running_sequences[run_live])


def test_shared_first_step_with_hinted_prompts_in_the_fixed_hint_mode(tmp_path):
    """Fixed-hint processor, prompts that already carry one ground-truth token (a different one per
    group), first step on one row per group, on a dense trie where a token continues almost any
    parent.  The plans of the B prompt rows and the plans the beam kernel writes for the B * K rows of
    the next call share the front of one buffer, so the kernel must not look up a parent's node there;
    entries a wrong lookup would hit are filled with a real but foreign node beforehand."""
    from genrec_amd.beam import BeamSearchState
    from genrec_amd.fixed_hint_logit_processor import FixedHintConstrainedLogitsProcessor
    from genrec_amd.synth import ASSISTANT_PREFIX, SidTokenizer, synthetic_index, write_index
    from genrec_amd.trie import build_trie_from_index, create_prefix_allowed_tokens_fn

    dev = _dev()
    path = write_index(str(tmp_path / "dense.index.json"), synthetic_index(300, 3, 8, seed=4))
    tok = SidTokenizer.from_index(path)
    trie, suffix, pidx = build_trie_from_index(path, tok, prefix=ASSISTANT_PREFIX)
    fn = create_prefix_allowed_tokens_fn(trie, suffix)
    V, eos, pad = tok.vocab_size_padded, tok.eos_token_id, tok.pad_token_id
    K, B, steps = 8, 6, 3
    items = sorted(trie)
    firsts = sorted({it[len(suffix)] for it in items})
    assert len(firsts) >= B
    g = torch.Generator().manual_seed(3)
    prompts = torch.tensor([[int(x) for x in torch.randint(100, 900, (4,), generator=g)] + list(suffix) + [firsts[b]]
                            for b in range(B)])
    rows = prompts.repeat_interleave(K, dim=0).to(dev)

    one = FixedHintConstrainedLogitsProcessor(fn, K, pidx, suffix, eos, sync_errors=True)
    full = FixedHintConstrainedLogitsProcessor(fn, K, pidx, suffix, eos, sync_errors=True)
    L = rows.shape[1] + steps + 1
    one._workspace(0, B * K)  # one buffer from the start, as a caller of the C ABI would hold it
    st_one = BeamSearchState(rows, K, L, eos, pad, 1.0, False, planner=one, first_step_shared=True)
    st_full = BeamSearchState(rows, K, L, eos, pad, 1.0, False, planner=full)
    # entries B .. B*K of the plan buffer: the node of group 0's prompt, a valid parent for most tokens
    ws = next(iter(one._workspaces.values()))
    plan_off = 512 * 4 + 2048 * 16  # arrival counters of split rows, their partial pairs, then the plans
    plans = ws[plan_off : plan_off + 16 * B * K].view(torch.int32).view(B * K, 4)
    torch.cuda.synchronize()
    assert int(plans[0, 0]) > 0
    plans[B:] = plans[0]
    first = Candidates(B, trie.max_fanout, dev)
    cand_one = Candidates(B * K, trie.max_fanout, dev)
    cand_full = Candidates(B * K, trie.max_fanout, dev)
    for t in range(steps):
        if t == 0:
            logits_b = (torch.randn(B, V, generator=g) * 3).to(dev)
            one.candidates_into(st_one.prompt_rows(), logits_b, first, one_row_per_group=True)
            full.candidates_into(st_full.flat_running_sequences(), logits_b.repeat_interleave(K, dim=0), cand_full)
            st_one.step(first, V, planner=one, shared_rows=True)
        else:
            logits = (torch.randn(B * K, V, generator=g) * 3).to(dev)
            one.candidates_into(st_one.flat_running_sequences(), logits, cand_one)
            full.candidates_into(st_full.flat_running_sequences(), logits, cand_full)
            valid = torch.arange(cand_full.cmax, device=dev)[None, :] < cand_full.count[:, None]
            assert torch.equal(cand_one.count, cand_full.count), f"step {t}: allowed sets differ"
            assert torch.equal(cand_one.tokens[valid], cand_full.tokens[valid])
            st_one.step(cand_one, V, planner=one)
        st_full.step(cand_full, V, planner=full)
        assert torch.equal(st_one.running_sequences, st_full.running_sequences)
    one._planned = full._planned = None


@pytest.mark.parametrize("first_step_shared", [False, True])
def test_evaluation_style_50_beams_with_repetition_penalty_reproduces_generate(first_step_shared):
    """tests/golden/eval50.npz (generate() as evaluate.py drives it: prefix_ids=None, count reset by hand,
    50 beams all returned, length_penalty 0, repetition_penalty 1.1 from transformers' own processor):
    the fused loop with the penalty folded into the streaming kernel returns the same item rankings
    (bit-exact sequences and beam parents) and scores within 2e-5, two batches on one processor."""
    from genrec_amd.beam import constrained_beam_search
    from genrec_amd.logit_processor import ConstrainedLogitsProcessor
    from genrec_amd.trie import FlatTrie, create_prefix_allowed_tokens_fn

    dev = _dev()
    data, meta = load_npz("eval50.npz")
    prefix, eos, pad, K = meta["prefix"], meta["eos"], meta["pad"], meta["K"]
    trie = FlatTrie(data["trie_sequences"].tolist())
    fn = create_prefix_allowed_tokens_fn(trie, prefix)
    proc = ConstrainedLogitsProcessor(fn, K, 3, None, eos, repetition_penalty=meta["repetition_penalty"])
    for batch in meta["batches"]:
        name = batch["name"]
        prompts = torch.from_numpy(data[f"{name}.prompts"]).to(dev)
        logits = [torch.from_numpy(data[f"{name}.logits.{t}"]).to(dev) for t in range(batch["steps"])]

        def logits_fn(step, flat, next_parent):
            if step == 0 and first_step_shared:
                return logits[0][::K].contiguous()
            return logits[step]

        # constrained_beam_search resets the processor itself, which is evaluate.py's `count = 0`
        _, seqs, scores, bidx = constrained_beam_search(
            logits_fn, proc, prompts, K, batch["max_new_tokens"], eos, pad, meta["length_penalty"], False,
            first_step_shared=first_step_shared)
        want_scores = data[f"{name}.sequences_scores"]
        live = want_scores > LIVE
        assert int(live.sum()) == batch["live"]
        assert np.array_equal(seqs.cpu().numpy()[live], data[f"{name}.sequences"][live])
        assert np.array_equal(bidx.cpu().numpy()[live], data[f"{name}.beam_indices"][live])
        assert np.max(np.abs(scores.cpu().numpy()[live] - want_scores[live])) <= 2e-5
        assert proc.count == batch["steps"]
    proc.check_errors()
