"""Randomised parity of both processors against the CPU oracle on small random tries: ragged item
lengths, prefix lengths 0..4, EOS present or None, rows that leave the trie, rows shorter than the
window, decoy prefix occurrences, arbitrary counter values.  Results, counter movement and exception
types must agree case by case."""

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

TOL = 1e-5


def _dev():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch.device("cuda", 0)


def _random_case(rng):
    V = int(rng.integers(64, 700))
    n_prefix = int(rng.integers(0, 5))
    prefix = rng.integers(3, 20, size=n_prefix).tolist()
    eos = int(rng.integers(0, 3)) if rng.random() < 0.8 else None
    levels = int(rng.integers(1, 5))
    items = set()
    for _ in range(int(rng.integers(1, 60))):
        depth = int(rng.integers(1, levels + 1))
        items.add(tuple(int(rng.integers(20, V)) for _ in range(depth)))
    tail = [eos] if eos is not None else [int(rng.integers(20, V))]
    seqs = [prefix + list(it) + tail for it in sorted(items)]
    K = int(rng.integers(1, 5))
    groups = int(rng.integers(1, 4))
    R = K * groups
    step = int(rng.integers(0, levels + 1))
    T = int(rng.integers(max(1, n_prefix + step), n_prefix + step + 12))
    rows = rng.integers(20, V, size=(R, T))
    for r in range(R):
        kind = rng.random()
        it = list(sorted(items)[int(rng.integers(0, len(items)))])
        gen = it[:step]
        body = prefix + gen
        if kind < 0.75 and len(body) <= T:
            rows[r, T - len(body):] = body           # a valid path (possibly shorter than `step`)
            if rng.random() < 0.3 and n_prefix and T - len(body) - n_prefix - 1 >= 0:
                at = int(rng.integers(0, T - len(body) - n_prefix))
                rows[r, at : at + n_prefix] = prefix  # a decoy occurrence earlier in the row
        elif kind < 0.85 and eos is not None:
            rows[r, -1] = eos                         # finished row
        # else: leave random tokens (leaves the trie / no prefix at all)
    count = int(rng.integers(0, levels + 2)) if rng.random() < 0.3 else step
    return dict(V=V, prefix=prefix, eos=eos, seqs=seqs, K=K, rows=rows.astype(np.int64), count=count, step=step,
                use_prefix_ids=bool(rng.random() < 0.6), dtype=[torch.float32, torch.bfloat16][int(rng.integers(0, 2))])


@pytest.mark.parametrize("block", range(6))
def test_random_tries_rows_and_counters(block):
    from oracle.processors import (DictTrie, RefConstrainedLogitsProcessor,
                                   RefFixedHintConstrainedLogitsProcessor)

    from genrec_amd.fixed_hint_logit_processor import FixedHintConstrainedLogitsProcessor
    from genrec_amd.logit_processor import ConstrainedLogitsProcessor
    from genrec_amd.trie import FlatTrie, create_prefix_allowed_tokens_fn

    dev = _dev()
    rng = np.random.default_rng(1000 + block)
    outcomes = {"ok": 0, "AssertionError": 0, "ValueError": 0}
    for _ in range(25):
        c = _random_case(rng)
        trie = FlatTrie(c["seqs"])
        fn = create_prefix_allowed_tokens_fn(trie, c["prefix"])
        get = DictTrie(c["seqs"]).get
        ids = torch.from_numpy(c["rows"])
        scores = (torch.randn(ids.shape[0], c["V"], generator=torch.Generator().manual_seed(int(rng.integers(1 << 30))))
                  * 4).to(c["dtype"])
        pidx, K, eos = len(c["prefix"]), c["K"], c["eos"]
        pids = c["prefix"] if c["use_prefix_ids"] else None
        for kind in ("count", "fixed"):
            if kind == "count":
                ours = ConstrainedLogitsProcessor(fn, K, pidx, pids, eos, sync_errors=True)
                ref = RefConstrainedLogitsProcessor(get, K, pidx, pids, eos)
                ours.count = ref.count = c["count"]
            else:
                ours = FixedHintConstrainedLogitsProcessor(fn, K, pidx, c["prefix"], eos, sync_errors=True)
                ref = RefFixedHintConstrainedLogitsProcessor(get, K, pidx, c["prefix"], eos)
            try:
                want = ref(ids, scores.float())
            except (AssertionError, ValueError) as e:
                with pytest.raises(type(e)):
                    ours(ids.to(dev), scores.to(dev))
                outcomes[type(e).__name__] += 1
                continue
            outcomes["ok"] += 1
            got = ours(ids.to(dev), scores.to(dev)).float().cpu()
            fin = torch.isfinite(want)
            assert torch.equal(torch.isfinite(got), fin), (kind, c["prefix"], c["count"])
            tol = TOL if c["dtype"] == torch.float32 else 0.07
            assert float((got[fin] - want[fin]).abs().max()) <= tol
            if kind == "count":
                assert ours.count == ref.count
                ours.count = c["count"]
            cand = ours.candidates(ids.to(dev), scores.to(dev))
            assert torch.equal(cand.count.cpu().long(), fin.sum(dim=1))
            valid = torch.arange(cand.cmax)[None, :] < cand.count.cpu()[:, None]
            cols = torch.nonzero(fin)[:, 1]
            assert torch.equal(cand.tokens.cpu()[valid].long(), cols)
            assert float((cand.logp.cpu()[valid] - want[fin]).abs().max()) <= TOL
    assert outcomes["ok"] >= 10 and outcomes["AssertionError"] + outcomes["ValueError"] >= 5, outcomes


@pytest.mark.parametrize("block", range(4))
def test_random_slicings_and_row_cuts_agree_with_whole_rows(block):
    """The same random tries, rows and counters through the sliced step (random slice borders, fed in
    random order) and through the row cut (`keep_top` with random running scores): whatever the oracle
    accepts must come out of the slices as it does from whole rows -- tokens and counts bit for bit,
    log-probs within 1e-5 of the oracle -- and the cut rows must hold exactly the best entries of the
    whole rows by (fp32(logp + running score) descending, token ascending)."""
    from oracle.processors import (DictTrie, RefConstrainedLogitsProcessor,
                                   RefFixedHintConstrainedLogitsProcessor)

    from genrec_amd.fixed_hint_logit_processor import FixedHintConstrainedLogitsProcessor
    from genrec_amd.logit_processor import Candidates, ConstrainedLogitsProcessor
    from genrec_amd.trie import FlatTrie, create_prefix_allowed_tokens_fn

    dev = _dev()
    rng = np.random.default_rng(5000 + block)
    n_ok = n_cut = 0
    for _ in range(25):
        c = _random_case(rng)
        V = c["V"]
        if V >= 200 and rng.random() < 0.6:
            # wide nodes where the rows stand (up to twelve parents at the rows' depth), so rows have entries to shed
            tail, n_pre, d = c["seqs"][0][-1:], len(c["prefix"]), c["step"]
            parents = sorted({tuple(q[n_pre : n_pre + d]) for q in c["seqs"] if len(q) - n_pre - 1 >= d})[:12]
            have = {tuple(q) for q in c["seqs"]}
            for par in parents:
                wide = rng.choice(np.arange(20, V), size=int(rng.integers(80, min(V - 20, 500))), replace=False)
                for t in wide:
                    q = c["prefix"] + list(par) + [int(t)] + tail
                    if tuple(q) not in have:
                        have.add(tuple(q))
                        c["seqs"].append(q)
        trie = FlatTrie(c["seqs"])
        fn = create_prefix_allowed_tokens_fn(trie, c["prefix"])
        get = DictTrie(c["seqs"]).get
        ids = torch.from_numpy(c["rows"])
        scores = (torch.randn(ids.shape[0], V, generator=torch.Generator().manual_seed(int(rng.integers(1 << 30))))
                  * 4).to(c["dtype"])
        pidx, K, eos = len(c["prefix"]), c["K"], c["eos"]
        pids = c["prefix"] if c["use_prefix_ids"] else None
        n_cuts = int(rng.integers(0, 5))
        borders = sorted({0, V, *rng.integers(1, V, size=n_cuts).tolist()})
        bounds = list(zip(borders[:-1], borders[1:]))
        order = rng.permutation(len(bounds)).tolist()
        for kind in ("count", "fixed"):
            def make():
                if kind == "count":
                    p = ConstrainedLogitsProcessor(fn, K, pidx, pids, eos, sync_errors=True)
                    p.count = c["count"]
                    return p
                return FixedHintConstrainedLogitsProcessor(fn, K, pidx, c["prefix"], eos, sync_errors=True)
            if kind == "count":
                ref = RefConstrainedLogitsProcessor(get, K, pidx, pids, eos)
                ref.count = c["count"]
            else:
                ref = RefFixedHintConstrainedLogitsProcessor(get, K, pidx, c["prefix"], eos)
            try:
                want = ref(ids, scores.float())
            except (AssertionError, ValueError):
                continue  # the error conventions of the sliced step have a test of their own
            n_ok += 1
            fin = torch.isfinite(want)
            whole = make().candidates(ids.to(dev), scores.to(dev))
            # ---- the sliced step
            sliced = make()
            cand = Candidates(ids.shape[0], max(trie.max_fanout, 1), dev)
            cand.buf.fill_(-7)
            step = sliced.chunked_step(ids.to(dev), cand, V, bounds, scores.dtype)
            dscores = scores.to(dev)
            for i in order:
                a, b = bounds[i]
                step.feed(a, dscores[:, a:b])
            got = step.finish()
            sliced.check_errors()
            tag = (kind, bounds, order, c["prefix"], c["count"])
            assert torch.equal(got.count.cpu().long(), fin.sum(dim=1)), tag
            valid = torch.arange(got.cmax)[None, :] < got.count.cpu()[:, None]
            assert torch.equal(got.tokens.cpu()[valid].long(), torch.nonzero(fin)[:, 1]), tag
            if bool(valid.any()):
                assert float((got.logp.cpu()[valid] - want[fin]).abs().max()) <= TOL, tag
            if kind == "count":
                assert sliced.count == ref.count
            # ---- the row cut
            keep = int(rng.integers(1, 9))
            if trie.max_fanout > 1024:
                continue
            bias = torch.from_numpy((rng.standard_normal(ids.shape[0]) * 2).astype(np.float32))
            cut = Candidates(ids.shape[0], max(trie.max_fanout, 1), dev)
            make().candidates_into(ids.to(dev), dscores, cut, keep_top=keep, row_bias=bias.to(dev))
            wc, wt, wl = whole.count.cpu().numpy(), whole.tokens.cpu().numpy(), whole.logp.cpu().numpy()
            gc, gt, gl = cut.count.cpu().numpy(), cut.tokens.cpu().numpy(), cut.logp.cpu().numpy()
            for r in range(ids.shape[0]):
                n = int(wc[r])
                t, v = wt[r, :n], wl[r, :n]
                if n > keep + 64:
                    n_cut += 1
                    ssum = (v + np.float32(bias[r].item())).astype(np.float32)
                    sel = np.lexsort((t, -ssum.astype(np.float64)))[:keep]
                    t, v = t[sel], v[sel]
                assert gc[r] == len(t), (tag, r)
                g, w = np.argsort(gt[r, : gc[r]], kind="stable"), np.argsort(t, kind="stable")
                assert np.array_equal(gt[r, : gc[r]][g], t[w]), (tag, r)
                assert np.array_equal(gl[r, : gc[r]][g].view(np.int32), v[w].view(np.int32)), (tag, r)
    assert n_ok >= 10 and n_cut >= 1, (n_ok, n_cut)


LIVE = -1.0e8  # scores above this belong to real hypotheses; placeholder beams sit near -1e9


def _random_beam_case(rng):
    K = int(rng.choice([1, 2, 3, 5, 8, 16, 17, 33, 65, 100]))
    B = int(rng.integers(1, 4))
    V = int(rng.integers(40, 2500))
    # long rows for many beams now and then: that is what sends the kernel down its by-column route
    widest = 300 if K <= 17 else (160 if rng.random() < 0.3 else 24)
    cmax = int(rng.integers(1, min(V - 2, widest) + 1))
    prompt_len = int(rng.choice([1, 4, 19]))
    max_new = int(rng.integers(2, 7))
    eos = int(rng.integers(0, 2)) if rng.random() < 0.75 else None
    return dict(K=K, B=B, V=V, cmax=cmax, prompt_len=prompt_len, max_new=max_new, eos=eos,
                shared=bool(rng.random() < 0.6),
                early_stopping=[False, True, "never"][int(rng.integers(0, 3))],
                length_penalty=float(rng.choice([0.0, 0.5, 1.0, 2.0])),
                p_eos=float(rng.choice([0.0, 0.2, 0.6])), p_empty=float(rng.choice([0.0, 0.1])))


@pytest.mark.parametrize("block", range(6))
def test_random_beam_searches_against_the_oracle(block):
    """Beam step fuzz: beam counts 1..100 (one wave, several waves, more beams than a wave), candidate
    rows from empty to full, every early_stopping mode, length penalties, EOS or none, prompts shared
    per group or not (sequence tails staged in LDS or copied from memory), run to max_length so the
    forced stop is hit.  Every live slot of every step must equal the CPU restatement of the
    transformers loop bit for bit."""
    from oracle.beam import RefBeamSearch

    from genrec_amd.beam import BeamSearchState
    from genrec_amd.logit_processor import Candidates

    dev = _dev()
    rng = np.random.default_rng(7000 + block)
    for _ in range(20):
        c = _random_beam_case(rng)
        K, B, V, cmax, P = c["K"], c["B"], c["V"], c["cmax"], c["prompt_len"]
        R = B * K
        if c["shared"]:
            prompts = torch.from_numpy(rng.integers(2, V, size=(B, P))).repeat_interleave(K, dim=0)
        else:
            prompts = torch.from_numpy(rng.integers(2, V, size=(R, P)))
        L = P + c["max_new"]
        ref = RefBeamSearch(prompts, K, L, c["eos"], 0, c["length_penalty"], c["early_stopping"])
        st = BeamSearchState(prompts.to(dev), K, L, c["eos"], 0, c["length_penalty"], c["early_stopping"],
                             shared_prompt=c["shared"])
        for step in range(c["max_new"]):
            # first step of groups that share a prompt: sometimes one candidate row per group, which all
            # K beams extend from (the oracle sees that row K times)
            one_per_group = step == 0 and c["shared"] and rng.random() < 0.5
            n_rows = B if one_per_group else R
            cnt = rng.integers(1, cmax + 1, size=n_rows).astype(np.int32)
            cnt[rng.random(n_rows) < c["p_empty"]] = 0
            tok = np.zeros((n_rows, cmax), dtype=np.int32)
            lp = np.zeros((n_rows, cmax), dtype=np.float32)
            dense = torch.full((n_rows, V), float("-inf"))
            for r in range(n_rows):
                n = int(cnt[r])
                cols = np.sort(rng.choice(V - 2, size=n, replace=False) + 2)
                if n and c["eos"] is not None and rng.random() < c["p_eos"]:
                    cols[0] = c["eos"]  # eos < 2 <= every other column: still ascending
                vals = (-rng.random(n) * 20).astype(np.float32)
                tok[r, :n], lp[r, :n] = cols, vals
                dense[r, torch.from_numpy(cols).long()] = torch.from_numpy(vals)
            tr = ref.step(dense.repeat_interleave(K, dim=0) if one_per_group else dense)
            res = st.step(Candidates.from_parts(torch.from_numpy(cnt), torch.from_numpy(tok), torch.from_numpy(lp), dev), V,
                          shared_rows=one_per_group)
            tag = (c, step, one_per_group)
            live = tr.topk_scores > LIVE
            assert torch.equal(res.topk_scores.cpu()[live], tr.topk_scores[live]), tag
            assert torch.equal(res.topk_token.cpu().long()[live], tr.topk_token[live]), tag
            assert torch.equal(res.topk_parent.cpu().long()[live], tr.topk_parent[live]), tag
            assert torch.equal(res.topk_hit.cpu().bool()[live], tr.topk_hit[live]), tag
            run_live = ref.running_beam_scores > LIVE
            assert torch.equal(st.running_beam_scores.cpu()[run_live], ref.running_beam_scores[run_live]), tag
            assert torch.equal(st.running_sequences.cpu()[run_live], ref.running_sequences[run_live]), tag
            assert torch.equal(st.run_bidx[st.cur].cpu()[run_live], ref.running_beam_indices[run_live]), tag
            assert torch.equal(res.next_parent.cpu().view(B, K)[run_live], tr.next_parent.view(B, K)[run_live]), tag
            fin_live = ref.beam_scores > LIVE
            assert torch.equal(st.beam_scores.cpu()[fin_live], ref.beam_scores[fin_live]), tag
            assert torch.equal(st.sequences.cpu()[fin_live], ref.sequences[fin_live]), tag
            assert torch.equal(st.beam_indices.cpu()[fin_live], ref.beam_indices[fin_live]), tag
            assert torch.equal(st.is_sent_finished.cpu()[fin_live], ref.is_sent_finished[fin_live]), tag
            # the heuristic flag only depends on placeholder slots when the best running beam is one
            decided = ref.running_beam_scores[:, 0] > LIVE
            assert torch.equal(st.unsat.cpu().bool()[decided], ref.unsat[:, 0][decided]), tag


@pytest.mark.parametrize("block", range(3))
def test_random_reward_rows_against_the_text_oracle(block):
    """Rule / prefix / NDCG rewards on random code rows: lengths 0..5 on either side, hints, group sizes
    1..20 with a trailing incomplete group, both prefix normalisations, code rows and token rows (ids
    without a code in between) -- against the restatement of the reference's text functions."""
    from oracle.scoring import ref_ndcg_reward, ref_prefix_reward, ref_rule_reward

    from genrec_amd.rewards.ranking_reward import SidRewardScorer, score_code_rows

    dev = _dev()
    rng = np.random.default_rng(300 + block)
    letters = "abcde"
    for _ in range(12):
        K = int(rng.integers(1, 21))
        R = int(rng.integers(1, 6)) * K + (int(rng.integers(0, K)) if rng.random() < 0.4 else 0)
        cb = int(rng.integers(2, 5))  # small codebook: plenty of exact and partial matches
        gts = [rng.integers(0, cb, size=int(rng.integers(0, 6))).tolist() for _ in range(R)]
        preds, hints = [], []
        for r in range(R):
            kind = rng.random()
            if kind < 0.35:
                full = list(gts[r])
            elif kind < 0.7:
                full = list(gts[r][: int(rng.integers(0, len(gts[r]) + 1))]) + rng.integers(0, cb, size=int(rng.integers(0, 3))).tolist()
            else:
                full = rng.integers(0, cb, size=int(rng.integers(0, 6))).tolist()
            cut = int(rng.integers(0, len(full) + 1)) if rng.random() < 0.5 else 0
            hints.append(full[:cut])
            preds.append(full[cut:])

        def text(codes, start=0):
            return "".join(f"<{letters[(start + i) % 5]}_{c}>" for i, c in enumerate(codes))

        completions = [text(p, len(h)) for p, h in zip(preds, hints)]
        hint_texts = [text(h) for h in hints]
        gt_texts = [text(g) for g in gts]
        normalize = bool(rng.integers(0, 2))
        want_rule = ref_rule_reward(completions, gt_texts, hint_texts)
        want_prefix = ref_prefix_reward(completions, gt_texts, hint_texts, normalize=normalize)
        want_ndcg = ref_ndcg_reward(completions, gt_texts, K, hint_texts)

        # (a) int32 code rows with lengths
        width = max(1, max(len(h) + len(p) for h, p in zip(hints, preds)), max(len(g) for g in gts))
        pc, pl = np.zeros((R, width), np.int32), np.zeros(R, np.int32)
        gc, gl = np.zeros((R, width), np.int32), np.zeros(R, np.int32)
        for r in range(R):
            full = hints[r] + preds[r]
            pc[r, : len(full)], pl[r] = full, len(full)
            gc[r, : len(gts[r])], gl[r] = gts[r], len(gts[r])
        got = score_code_rows(torch.from_numpy(pc).to(dev), torch.from_numpy(pl).to(dev), torch.from_numpy(gc).to(dev),
                              torch.from_numpy(gl).to(dev), K, normalize)
        n_full = (R // K) * K
        assert got["rule"].cpu().tolist() == want_rule
        assert np.allclose(got["prefix"].cpu().numpy(), np.asarray(want_prefix, dtype=np.float32), rtol=0, atol=1e-7)
        assert np.allclose(got["ndcg"].cpu().numpy()[:n_full], np.asarray(want_ndcg, dtype=np.float32), rtol=0, atol=1e-7)

        # (b) int64 token rows through a code table; ids >= 100 carry no code (padding, EOS) and are skipped
        table = np.full(200, -1, np.int32)
        table[:cb] = np.arange(cb)
        tw = width + 3

        def token_rows(rows):
            out = np.full((R, tw), 150, np.int64)
            for r, codes in enumerate(rows):
                slots = np.sort(rng.choice(tw, size=len(codes), replace=False))
                out[r, slots] = codes
            return torch.from_numpy(out).to(dev)

        scorer = SidRewardScorer(table, K, dev, prefix_normalize=normalize)
        hint_rows = token_rows(hints) if any(hints) else None
        got_t = scorer(token_rows(preds), token_rows(gts), hint_rows)
        assert got_t["rule"].cpu().tolist() == want_rule
        assert np.allclose(got_t["prefix"].cpu().numpy(), np.asarray(want_prefix, dtype=np.float32), rtol=0, atol=1e-7)
        assert np.allclose(got_t["ndcg"].cpu().numpy()[:n_full], np.asarray(want_ndcg, dtype=np.float32), rtol=0, atol=1e-7)
