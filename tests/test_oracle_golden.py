"""CPU: the oracle (restatement of the reference) against the golden vectors captured
from the real reference, and against the known answers in the reference's tests/docs."""

import math

import numpy as np
import pytest
import torch

import os

from helpers import GOLDEN, INDEX_NAMES, load_json_gz, load_npz, write_index_file
from oracle import beam as obeam
from oracle import processors as oproc
from oracle import scoring as oscore


def test_trie_queries_match_reference(tmp_path):
    from genrec_amd.synth import ASSISTANT_PREFIX, SidTokenizer

    gold = load_json_gz("tries.json.gz")
    for name in INDEX_NAMES:
        g = gold[name]
        path = write_index_file(name, tmp_path)
        tok = SidTokenizer.from_index(path)
        trie, suffix, pidx = oproc.ref_build_trie_from_index(path, tok, prefix=ASSISTANT_PREFIX)
        assert (len(trie), suffix, pidx) == (g["n_unique_items"], g["prefix_ids"], g["prefix_index"])
        assert tok.vocab_size_padded == g["vocab_padded"] and len(tok.new_tokens) == g["n_sid_tokens"]
        for q, a in zip(g["queries"], g["answers"]):
            assert sorted(trie.get(q)) == a


def _oracle_processor(case, fn, meta):
    K, prefix, eos = meta["K"], meta["prefix"], meta["eos"]
    if case in ("count_rl", "count_midloop_reset"):
        return oproc.RefConstrainedLogitsProcessor(fn, K, 3, prefix, eos)
    if case == "count_eval":
        return oproc.RefConstrainedLogitsProcessor(fn, K, 3, None, eos)
    if case == "count_seeded2":
        return oproc.RefConstrainedLogitsProcessor(fn, K, 3, prefix, eos, initial_count=2)
    return oproc.RefFixedHintConstrainedLogitsProcessor(fn, K, 3, prefix, eos)


@pytest.mark.parametrize("case", ["count_rl", "count_eval", "count_seeded2", "count_midloop_reset", "fixed_hint"])
def test_processor_oracle_bit_exact_with_reference(case):
    data, meta = load_npz("processors.npz")
    trie = oproc.DictTrie(data["trie_sequences"].tolist())
    proc = _oracle_processor(case, trie.get, meta)
    info = meta["cases"][case]
    for i in range(info["steps"]):
        out = proc(torch.from_numpy(data[f"{case}.ids.{i}"]), torch.from_numpy(data[f"{case}.scores.{i}"]))
        assert np.array_equal(out.numpy(), data[f"{case}.out.{i}"])  # same torch ops in the same order
        if info["counts_after"][i] >= 0:
            assert proc.count == info["counts_after"][i]


def test_processor_oracle_error_cases():
    data, meta = load_npz("processors.npz")
    trie = oproc.DictTrie(data["trie_sequences"].tolist())
    K, prefix, eos, V = meta["K"], meta["prefix"], meta["eos"], meta["V"]
    bad = torch.from_numpy(data["bad_rows"])
    scores = torch.zeros(bad.shape[0], V)
    want = {e["processor"]: e for e in meta["errors"]}
    p = oproc.RefConstrainedLogitsProcessor(trie.get, K, 3, None, eos)
    p.count = 1
    with pytest.raises(AssertionError) as e:
        p(bad, scores)
    assert str(e.value)[:60] == want["count"]["message"]
    f = oproc.RefFixedHintConstrainedLogitsProcessor(trie.get, K, 3, prefix, eos)
    with pytest.raises(ValueError) as e:
        f(bad, scores)
    assert str(e.value)[:60] == want["fixed"]["message"]
    with pytest.raises(ValueError) as e:
        f(torch.from_numpy(data["noprefix_rows"]), scores)
    assert str(e.value)[:60] == want["fixed_noprefix"]["message"]


def test_reference_unit_test_known_answers():
    """tests/test_fixed_hint_runtime.py:75-105,185-192 and
    tests/test_fixed_hint_logit_processor_selection.py:36-65 of the reference."""
    calls = []

    def allowed(ids):
        calls.append(list(ids))
        return [0]

    p = oproc.RefConstrainedLogitsProcessor(allowed, 2, 3, [101, 102, 103], 0)
    p(torch.tensor([[9, 101, 102, 103], [9, 101, 102, 103]]), torch.zeros(2, 10))
    assert calls == [[101, 102, 103], [101, 102, 103]]

    def only_full(ids):
        return [7] if ids == [101, 102, 103, 9001, 101, 102, 103] else []

    assert oproc.fixed_hint_allowed_row([101, 102, 103, 9001, 101, 102, 103], [101, 102, 103], only_full, None) == [7]
    assert oproc.fixed_hint_allowed_row([101, 102, 103, 999], [101, 102, 103], lambda ids: [], 999) == [999]

    from genrec_amd.fixed_hint_logit_processor import (
        find_last_prefix_match_start,
        find_valid_prefix_match_start,
        resolve_allowed_tokens_for_sequence,
    )

    sent = [1, 2, 3, 101, 102, 103, 9001, 9002, 42]
    assert find_last_prefix_match_start(sent, [101, 102, 103]) == 3
    assert find_valid_prefix_match_start([101, 102, 103, 9001, 101, 102, 103], [101, 102, 103], only_full) == 0
    assert resolve_allowed_tokens_for_sequence([101, 102, 103, 999], [101, 102, 103], lambda ids: [], 999) == [999]
    with pytest.raises(ValueError):
        find_last_prefix_match_start([1, 2], [5])


@pytest.mark.parametrize("case", ["count_k4_lp1", "count_k16_lp0", "count_k16_lp1", "fixed_k8_lp1"])
def test_beam_oracle_reproduces_generate(case):
    """Processor outputs recorded from transformers.generate -> oracle loop -> same sequences,
    scores and beam indices (live beams)."""
    data, meta = load_npz("beams.npz")
    info = meta["cases"][case]
    K = info["K"]
    prompts = torch.from_numpy(data[f"{case}.prompts"])
    st = obeam.RefBeamSearch(prompts.repeat_interleave(K, dim=0), K, prompts.shape[1] + info["max_new_tokens"],
                             meta["eos"], meta["pad"], info["length_penalty"], False)
    for t in range(info["steps"]):
        assert st.has_unfinished()
        st.step(torch.from_numpy(data[f"{case}.scores.{t}"]))
    assert not st.has_unfinished()
    seqs, scores, bidx = st.finalize()
    want_scores = torch.from_numpy(data[f"{case}.sequences_scores"])
    live = want_scores > -1e8
    assert torch.equal(scores[live], want_scores[live])
    assert torch.equal(seqs[live], torch.from_numpy(data[f"{case}.sequences"])[live])
    assert torch.equal(bidx[live], torch.from_numpy(data[f"{case}.beam_indices"])[live])


@pytest.mark.parametrize("case", ["count_k16_lp1", "fixed_k8_lp1"])
def test_whole_oracle_pipeline_from_raw_logits(case):
    """Raw logits -> log_softmax -> oracle processor -> oracle beam loop == generate()."""
    data, meta = load_npz("beams.npz")
    info = meta["cases"][case]
    K, prefix, eos = info["K"], meta["prefix"], meta["eos"]
    trie = oproc.DictTrie(data["trie_sequences"].tolist())
    cls = oproc.RefFixedHintConstrainedLogitsProcessor if case.startswith("fixed") else oproc.RefConstrainedLogitsProcessor
    proc = cls(trie.get, K, 3, prefix, eos)
    prompts = torch.from_numpy(data[f"{case}.prompts"])
    logits = [torch.from_numpy(data[f"{case}.logits.{t}"]) for t in range(info["steps"])]
    st, traces = obeam.ref_constrained_beam_search(lambda step, flat: logits[step], proc, prompts, K,
                                                   info["max_new_tokens"], eos, meta["pad"], info["length_penalty"])
    assert len(traces) == info["steps"]
    seqs, scores, _ = st.finalize()
    want_scores = torch.from_numpy(data[f"{case}.sequences_scores"])
    live = want_scores > -1e8
    assert torch.equal(scores[live], want_scores[live])
    assert torch.equal(seqs[live], torch.from_numpy(data[f"{case}.sequences"])[live])


def test_scoring_oracle_matches_reference_golden():
    g = load_json_gz("scoring.json.gz")
    K = g["K"]
    assert g["token_prefix_statics_checked"] is True  # _group_normalize(_tokenwise) were compared with the reference
    assert oscore.ref_rule_reward(g["comps"], g["gts"], g["hints"]) == g["rule"]
    assert oscore.ref_prefix_reward(g["comps"], g["gts"], g["hints"], True) == g["prefix_norm"]
    assert oscore.ref_prefix_reward(g["comps"], g["gts"], g["hints"], False) == g["prefix_raw"]
    assert oscore.ref_ndcg_reward(g["comps"], g["gts"], K, g["hints"]) == g["ndcg"]
    assert oscore.ref_rule_reward(g["comps2"], g["gts2"]) == g["rule2"]
    assert oscore.ref_prefix_reward(g["comps2"], g["gts2"]) == g["prefix2"]
    assert oscore.ref_ndcg_reward(g["comps2"], g["gts2"], K) == g["ndcg2"]
    assert oscore.ref_compute_metrics(g["metrics_data"]) == g["metrics"]


def test_scoring_oracle_matches_reference_on_edge_cases():
    """Ragged prediction lists and numbers beyond 32 bits, recorded from the reference by
    tests/golden/make_golden_edge.py."""
    g = load_json_gz("scoring_edge.json.gz")
    for name, case in g["metrics"].items():
        assert oscore.ref_compute_metrics(case["data"]) == case["metrics"], name
    assert oscore.ref_rule_reward(g["comps"], g["gts"]) == g["rule"]
    assert oscore.ref_prefix_reward(g["comps"], g["gts"]) == g["prefix"]
    assert oscore.ref_ndcg_reward(g["comps"], g["gts"], g["K"]) == g["ndcg"]


def test_docs_worked_examples():
    """docs/2026-03-13-genrec-rl-reward-forms.md:119,216-221,283-301,358-380,407-423 of the reference."""
    assert [round(w, 4) for w in oscore.ndcg_rank_weights(4)] == [-0.3904, -0.2463, -0.1952, -0.1681]
    gts = ["<a_10><b_20><c_30><d_40>"] * 4
    comps = ["<a_10><b_20><c_30><d_40>", "<a_10><b_20><c_30><d_99>", "<a_10><b_99><c_88><d_77>",
             "<a_99><b_88><c_77><d_66>"]
    assert oscore.ref_rule_reward(comps, gts) == [1.0, 0.0, 0.0, 0.0]
    assert oscore.ref_prefix_reward(comps, gts) == [1.0, 0.75, 0.25, 0.0]
    ndcg = oscore.ref_ndcg_reward(comps, gts, 4)
    assert [round(x, 4) for x in ndcg] == [0.0, -0.2463, -0.1952, -0.1681]
    prefix_tok = torch.tensor([[0.25] * 4, [0.25, 0.25, 0.25, 0], [0.25, 0, 0, 0], [0.0] * 4])
    ones = torch.ones(4, 4)
    got = oscore.ref_group_normalize_tokenwise(prefix_tok, ones, 4)
    want = torch.tensor([[0.5749, 0.9968, 0.9968, 1.7247], [0.5749, 0.9968, 0.9968, -0.5749],
                         [0.5749, -0.9968, -0.9968, -0.5749], [-1.7247, -0.9968, -0.9968, -0.5749]])
    assert torch.allclose(got, want, atol=1e-4)
    total = prefix_tok + torch.tensor(ndcg, dtype=torch.float32).unsqueeze(1)
    got = oscore.ref_group_normalize_tokenwise(total, ones, 4)
    want = torch.tensor([[1.4386, 1.5613, 1.5613, 1.7124], [-0.2102, 0.1751, 0.1751, -0.7879],
                         [0.1320, -0.9443, -0.9443, -0.5304], [-1.3604, -0.7920, -0.7920, -0.3941]])
    assert torch.allclose(got, want, atol=1e-4)
    err = torch.tensor([[0, 0, 0, 0], [0, 0, 0, 1], [0, 1, 1, 1], [1, 1, 1, 1]], dtype=torch.float32)
    errtok = prefix_tok + torch.tensor(ndcg, dtype=torch.float32).unsqueeze(1) * err
    got = oscore.ref_group_normalize_tokenwise(errtok, ones, 4)
    want = torch.tensor([[0.5765, 0.9980, 0.9980, 1.7124], [0.5765, 0.9980, 0.9980, -0.7879],
                         [0.5765, -1.0605, -1.0605, -0.5304], [-1.7294, -0.9354, -0.9354, -0.3941]])
    assert torch.allclose(got, want, atol=1e-4)
    # prefix_token: sequence-normalised prefix advantage spread over matched tokens + broadcast ndcg advantage
    adv_p = oscore.ref_group_normalize(torch.tensor([1.0, 0.75, 0.25, 0.0]), 4)
    adv_n = oscore.ref_group_normalize(torch.tensor(ndcg, dtype=torch.float32), 4)
    mass = prefix_tok.sum(dim=1, keepdim=True)
    dist = torch.where(mass > 0, prefix_tok / (mass + 1e-8), torch.zeros_like(prefix_tok))
    tok_adv = adv_p.unsqueeze(1) * dist + adv_n.unsqueeze(1) * ones
    want = torch.tensor([[1.7015] * 4, [-0.6971, -0.6971, -0.6971, -0.8796], [-0.9484, -0.4008, -0.4008, -0.4008],
                         [-0.1473] * 4])
    assert torch.allclose(tok_adv, want, atol=1e-4)
    assert math.isclose(float(oscore.ref_ndcg_rank_penalties(8, 4, 1)[5]), -0.2463, abs_tol=1e-4)


def test_cascade_oracle_matches_reference_golden():
    """The dynamic-hint cascade restatement makes the reference's decisions on every recorded scenario
    (tests/golden/make_golden_cascade.py drove the reference's own method)."""
    from oracle import cascade as ocascade

    cases = load_json_gz("cascade.json.gz")["cases"]
    assert len(cases) >= 30
    multi = 0
    for case in cases:
        table = {(g, d): comps for g, d, comps in case["table"]}
        out = ocascade.run_cascade(case["gts"], case["overrides"], case["K"], case["max_depth"],
                                   lambda groups, depths: [table[(g, d)] for g, d in zip(groups, depths)],
                                   case["others"])
        want = case["expected"]
        assert out["selected_group_hint_depths"] == want["selected_group_hint_depths"]
        assert out["selected_group_rule_hits"] == want["selected_group_rule_hits"]
        assert [r for s in out["selected"] for r in s["rule_rewards"]] == want["selected_rule_rewards"]
        assert [c for s in out["selected"] for c in s["completions"]] == want["selected_completions"]
        assert [s["prompt_hint_depth"] for s in out["selected"]] == want["selected_prompt_hint_depths"]
        assert out["stage_stats"] == want["stage_stats"]
        assert [{"groups": c["groups"], "depths": c["depths"]} for c in out["calls"]] == want["calls"]
        multi += case["others"] is not None
    assert multi >= 8  # scenarios in which other ranks keep the cascade running


def test_penalty_chain_oracle_bit_exact_with_reference(tmp_path):
    """log_softmax -> repetition penalty -> processor, restated in oracle/, equals the values recorded
    from transformers' RepetitionPenaltyLogitsProcessor feeding the reference's processors."""
    from genrec_amd.synth import ASSISTANT_PREFIX, SidTokenizer
    from genrec_amd.trie import build_trie_from_index

    data, meta = load_npz("penalty.npz")
    path = write_index_file("Instruments", tmp_path)
    tok = SidTokenizer.from_index(path)
    trie, suffix, pidx = build_trie_from_index(path, tok, prefix=ASSISTANT_PREFIX)
    o_trie = oproc.DictTrie(sorted(trie))
    K, V, eos = meta["K"], meta["V"], meta["eos"]
    for ci, case in enumerate(meta["cases"]):
        ids = torch.from_numpy(data[f"{ci}.ids"])
        logits = torch.randn(ids.shape[0], V, generator=torch.Generator().manual_seed(case["seed"])) * meta["scale"]
        if case["peaked"]:
            logits[torch.arange(ids.shape[0]), ids[:, 7]] += 40.0
        lp = torch.nn.functional.log_softmax(logits, dim=-1)
        pen = oproc.ref_repetition_penalty(ids, lp, case["penalty"], case["skip"])
        if case["processor"] == "count":
            proc = oproc.RefConstrainedLogitsProcessor(o_trie.get, K, pidx, None, eos)
            proc.count = case["step"]
        else:
            proc = oproc.RefFixedHintConstrainedLogitsProcessor(o_trie.get, K, pidx, suffix, eos)
        out = proc(ids, pen)
        for r in range(ids.shape[0]):
            n = int(data[f"{ci}.count"][r])
            cols = torch.nonzero(torch.isfinite(out[r])).flatten()
            assert cols.tolist() == data[f"{ci}.tok"][r, :n].tolist()
            assert np.array_equal(out[r, cols].numpy(), data[f"{ci}.val"][r, :n])


def test_beam_oracle_reproduces_generate_on_a_ragged_trie_in_every_stopping_mode():
    """Items of 1..4 levels: EOS arrives at different steps, finished beams compete with running ones.
    The oracle loop (raw logits -> log_softmax -> fixed-hint processor -> beam step) returns what
    generate() returned for early_stopping False / True / "never" and four length penalties."""
    from helpers import ragged_beam_cases

    data, meta, seqs = ragged_beam_cases()
    prefix, eos, pad = meta["prefix"], meta["eos"], meta["pad"]
    prompts = torch.from_numpy(data["prompts"])
    get = oproc.DictTrie(seqs).get
    assert len(meta["cases"]) == 24
    for name, info in meta["cases"].items():
        K = info["K"]
        proc = oproc.RefFixedHintConstrainedLogitsProcessor(get, K, 3, prefix, eos)
        logits = [torch.from_numpy(data[f"{name}.logits.{t}"]) for t in range(info["steps"])]
        state, _ = obeam.ref_constrained_beam_search(
            lambda step, flat: logits[step], proc, prompts, K, info["max_new_tokens"], eos, pad,
            info["length_penalty"], info["early_stopping"])
        assert state.cur_len - prompts.shape[1] == info["steps"], f"{name}: loop length"
        seqs_o, scores_o, bidx_o = state.finalize()
        want_scores = torch.from_numpy(data[f"{name}.sequences_scores"])
        want_seqs = torch.from_numpy(data[f"{name}.sequences"])
        want_bidx = torch.from_numpy(data[f"{name}.beam_indices"])
        live = want_scores > -1.0e8
        w, bw = min(seqs_o.shape[1], want_seqs.shape[1]), min(bidx_o.shape[1], want_bidx.shape[1])
        assert torch.equal(scores_o[live], want_scores[live]), name
        assert torch.equal(seqs_o[:, :w][live], want_seqs[:, :w][live]), name
        assert torch.equal(bidx_o[:, :bw].long()[live], want_bidx[:, :bw].long()[live]), name


def test_oracle_reproduces_evaluation_style_generate_with_50_beams_and_repetition_penalty():
    """tests/golden/eval50.npz: generate() as evaluate.py drives it (count processor with prefix_ids=None,
    count reset by hand per batch, 50 beams, length_penalty 0, repetition_penalty 1.1 applied by
    transformers ahead of the reference processor), two batches with one processor object."""
    from helpers import load_npz

    data, meta = load_npz("eval50.npz")
    prefix, eos, pad, K = meta["prefix"], meta["eos"], meta["pad"], meta["K"]
    seqs = data["trie_sequences"].tolist()
    inner = oproc.RefConstrainedLogitsProcessor(oproc.DictTrie(seqs).get, K, 3, None, eos)

    def chain(ids, scores):
        return inner(ids, oproc.ref_repetition_penalty(ids, scores, meta["repetition_penalty"]))

    for batch in meta["batches"]:
        name = batch["name"]
        prompts = torch.from_numpy(data[f"{name}.prompts"])
        logits = [torch.from_numpy(data[f"{name}.logits.{t}"]) for t in range(batch["steps"])]
        inner.count = 0
        state, _ = obeam.ref_constrained_beam_search(lambda step, flat: logits[step], chain, prompts, K,
                                                     batch["max_new_tokens"], eos, pad, meta["length_penalty"], False)
        seqs_o, scores_o, bidx_o = state.finalize()
        want_scores = torch.from_numpy(data[f"{name}.sequences_scores"])
        live = want_scores > -1.0e8
        assert int(live.sum()) == batch["live"]
        assert float((scores_o[live] - want_scores[live]).abs().max()) <= 2e-5  # fp32 sums of 5 log-probs
        assert torch.equal(seqs_o[live], torch.from_numpy(data[f"{name}.sequences"])[live])
        assert torch.equal(bidx_o.long()[live], torch.from_numpy(data[f"{name}.beam_indices"]).long()[live])
        assert inner.count == batch["steps"]


def test_token_prefix_oracle_bit_exact_with_the_reference_trainer():
    """token_prefix.npz holds what the reference's TokenPrefixGRPOTrainer produced (reward builders and
    _generate_and_score_completions, token_prefix_grpo_trainer.py:107-149,174-287): the restatement must
    reproduce token rewards, sequence rewards and error masks bit for bit.  The advantages are fp32 means and
    variances taken by PyTorch's CPU reduction kernels, whose summation order follows the vector width of the
    host that runs them: bit-equal on the host that recorded the fixture (asserted by the generating script,
    restatement and reference side by side), within 4 ulp of a unit-scale value (5e-7) on any other."""
    d = np.load(os.path.join(GOLDEN, "token_prefix.npz"))
    n, eos = int(d["n_cases"][0]), int(d["n_cases"][1])
    assert n == 72
    for c in range(n):
        K, normalize, has_ndcg, total, err_pen = (int(v) for v in d[f"c{c}.cfg"])
        pw, nw, _ = (float(v) for v in d[f"c{c}.weights"])
        ids, mask = torch.from_numpy(d[f"c{c}.ids"]), torch.from_numpy(d[f"c{c}.mask"].astype(np.int64))
        gt = [row[:n_].tolist() for row, n_ in zip(d[f"c{c}.gt_ids"], d[f"c{c}.gt_len"])]
        tr, sr, em = oscore.ref_prefix_token_rewards(ids, mask, gt, eos, bool(normalize), True)
        assert torch.equal(tr, torch.from_numpy(d[f"c{c}.token_rewards"]))
        assert torch.equal(sr, torch.from_numpy(d[f"c{c}.seq_rewards"]))
        assert torch.equal(em, torch.from_numpy(d[f"c{c}.error_mask"].astype(np.float32)))
        ndcg = torch.from_numpy(d[f"c{c}.ndcg"]) if has_ndcg else None
        adv = oscore.ref_token_prefix_advantages(tr, sr, em, mask, K, pw, nw, bool(has_ndcg), ndcg, bool(total),
                                                 bool(err_pen))
        assert float((adv - torch.from_numpy(d[f"c{c}.advantages"])).abs().max()) <= 5e-7, c
