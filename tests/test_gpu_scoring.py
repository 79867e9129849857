"""GPU parity of rewards, ranking metrics and advantages (through the C ABI)
against vectors captured from the reference and the reference docs' worked examples."""

import math

import numpy as np
import pytest
import torch

from helpers import load_json_gz

pytestmark = pytest.mark.gpu


def _dev():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch.device("cuda", 0)


def test_text_rewards_match_reference_golden():
    from genrec_amd.rewards import get_ndcg_rule_reward, get_prefix_rule_reward, rule_reward

    _dev()
    g = load_json_gz("scoring.json.gz")
    K = g["K"]
    kw = dict(reward_model=[{"ground_truth": s} for s in g["gts"]], oracle_hint_text=g["hints"])
    assert rule_reward(None, g["comps"], None, **kw) == g["rule"]
    assert get_prefix_rule_reward(True)(None, g["comps"], None, **kw) == g["prefix_norm"]
    assert get_prefix_rule_reward(False)(None, g["comps"], None, **kw) == g["prefix_raw"]
    got = get_ndcg_rule_reward(K)(None, g["comps"], None, **kw)
    assert np.allclose(got, g["ndcg"], rtol=0, atol=1e-7) and [x == 0 for x in got] == [x == 0 for x in g["ndcg"]]
    kw2 = dict(reward_model=[{"ground_truth": s} for s in g["gts2"]])
    assert rule_reward(None, g["comps2"], None, **kw2) == g["rule2"]
    assert np.allclose(get_prefix_rule_reward(True)(None, g["comps2"], None, **kw2), g["prefix2"], rtol=0, atol=1e-7)
    assert np.allclose(get_ndcg_rule_reward(K)(None, g["comps2"], None, **kw2), g["ndcg2"], rtol=0, atol=1e-7)
    # conversational completions and the positional fallback for reward_model
    conv = [[{"role": "assistant", "content": c}] for c in g["comps2"]]
    assert rule_reward(None, conv, None, **kw2) == g["rule2"]
    assert rule_reward(None, g["comps2"], None, a=1, b=2, rm=kw2["reward_model"], d=4, e=5) == g["rule2"]


def test_known_answers_from_reference_tests_and_docs():
    """tests/test_fixed_hint_runtime.py:143-182 and docs/2026-03-13-genrec-rl-reward-forms.md:119,216-221."""
    from genrec_amd.rewards import get_ndcg_rule_reward, get_prefix_rule_reward, rule_reward
    from genrec_amd.rewards.ranking_reward import ndcg_rank_weights

    _dev()
    gt = "<a_1><b_2><c_3><d_4>"
    assert rule_reward(["p"], ["<c_3><d_4>"], None, reward_model=[{"ground_truth": gt}],
                       oracle_hint_text=["<a_1><b_2>"]) == [1.0]
    assert get_prefix_rule_reward(True)(["p"], ["<c_3><d_9>"], None, reward_model=[{"ground_truth": gt}],
                                        oracle_hint_text=["<a_1><b_2>"]) == [0.75]
    r = get_ndcg_rule_reward(2)(["p", "p"], ["<c_3><d_4>", "<c_3><d_9>"], None,
                                reward_model=[{"ground_truth": gt}] * 2, oracle_hint_text=["<a_1><b_2>"] * 2)
    assert r[0] == 0.0 and r[1] < 0.0
    assert [round(w, 4) for w in ndcg_rank_weights(4)] == [-0.3904, -0.2463, -0.1952, -0.1681]
    gts = ["<a_10><b_20><c_30><d_40>"] * 4
    comps = ["<a_10><b_20><c_30><d_40>", "<a_10><b_20><c_30><d_99>", "<a_10><b_99><c_88><d_77>",
             "<a_99><b_88><c_77><d_66>"]
    kw = dict(reward_model=[{"ground_truth": s} for s in gts])
    assert rule_reward(None, comps, None, **kw) == [1.0, 0.0, 0.0, 0.0]
    assert get_prefix_rule_reward(True)(None, comps, None, **kw) == [1.0, 0.75, 0.25, 0.0]
    assert [round(x, 4) for x in get_ndcg_rule_reward(4)(None, comps, None, **kw)] == [0.0, -0.2463, -0.1952, -0.1681]
    # a group without an exact row gets all zeros
    assert get_ndcg_rule_reward(4)(None, comps[1:] + comps[1:2], None, **kw) == [0.0] * 4


def test_token_level_scorer_matches_text_rewards():
    from oracle.scoring import ref_ndcg_reward, ref_prefix_reward, ref_rule_reward

    from genrec_amd.rewards import SidRewardScorer
    from genrec_amd.synth import EOS_TOKEN_ID, SidTokenizer

    dev = _dev()
    rng = np.random.default_rng(3)
    letters = "abcd"
    toks = [f"<{letters[l]}_{c}>" for l in range(4) for c in range(8)]
    tk = SidTokenizer(toks)
    K, groups = 16, 40
    comps, gts, hints, comp_ids, gt_ids, hint_ids = [], [], [], [], [], []
    for g in range(groups):
        gt = rng.integers(0, 8, size=4)
        d = int(rng.integers(0, 3))
        for k in range(K):
            pred = gt.copy() if rng.random() < 0.2 else np.where(rng.random(4) < 0.6, gt, rng.integers(0, 8, size=4))
            pred[:d] = gt[:d]
            text = lambda codes, lo, hi: "".join(f"<{letters[i]}_{int(codes[i])}>" for i in range(lo, hi))  # noqa: E731
            gts.append(text(gt, 0, 4)); hints.append(text(gt, 0, d)); comps.append(text(pred, d, 4))
            gt_ids.append(tk.encode(gts[-1]))
            hint_ids.append([tk.pad_token_id] * (2 - d) + tk.encode(hints[-1]))
            comp_ids.append(tk.encode(comps[-1]) + [EOS_TOKEN_ID] + [tk.pad_token_id] * d)
    scorer = SidRewardScorer(tk.code_table(), K, dev)
    res = scorer(torch.tensor(comp_ids, device=dev), torch.tensor(gt_ids, device=dev), torch.tensor(hint_ids, device=dev))
    assert res["rule"].cpu().tolist() == ref_rule_reward(comps, gts, hints)
    assert res["prefix"].cpu().tolist() == ref_prefix_reward(comps, gts, hints, True)
    assert np.allclose(res["ndcg"].cpu().numpy(), ref_ndcg_reward(comps, gts, K, hints), rtol=0, atol=1e-7)


def test_metrics_match_reference_golden(tmp_path):
    from genrec_amd.metrics import compute_metrics, compute_metrics_from_ids

    dev = _dev()
    g = load_json_gz("scoring.json.gz")
    save = tmp_path / "run" / "metrics.json"
    got = compute_metrics(g["metrics_data"], save_path=str(save))
    assert set(got) == set(g["metrics"])
    for k, v in g["metrics"].items():
        assert abs(got[k] - v) <= 1e-12, k
    import json

    saved = json.loads(save.read_text())
    assert saved == {k: round(v, 4) for k, v in g["metrics"].items()}
    assert (tmp_path / "run" / "metrics.tsv").exists()
    # id-level interface, 50 beams, all cut-offs
    n, nb, L = 300, 50, 4
    rng = np.random.default_rng(0)
    beams = rng.integers(0, 5, size=(n, nb, L))
    target = rng.integers(0, 5, size=(n, L))
    m = compute_metrics_from_ids(torch.tensor(beams, device=dev), torch.tensor(target, device=dev))
    for k in (1, 3, 5, 10, 20, 50):
        hr, dcg = 0, 0.0
        for i in range(n):
            hits = np.nonzero((beams[i] == target[i]).all(axis=1))[0]
            if len(hits) and hits[0] < k:
                hr += 1
                dcg += 1.0 / math.log(hits[0] + 2)
        assert abs(m[f"HR@{k}"] - hr / n) <= 1e-12
        assert abs(m[f"NDCG@{k}"] - dcg / n / (1.0 / math.log(2))) <= 1e-12
    assert compute_metrics([]) == {}


def test_group_advantages_match_oracle_and_docs():
    from oracle.scoring import ref_group_advantage, ref_group_normalize, ref_group_normalize_tokenwise

    from genrec_amd.advantage import group_advantages, group_normalize, group_normalize_tokenwise, ndcg_rank_penalties

    dev = _dev()
    torch.manual_seed(5)
    for K, G, F in ((16, 64, 3), (4, 7, 1), (50, 3, 2)):
        rpf = torch.randn(G * K, F)
        rpf[3, 0] = float("nan")
        rpf[::5] = rpf[0]  # some repeated rewards
        rpf[K : 2 * K] = 1.0  # a zero-variance group
        w = torch.rand(F)
        for scale in ("group", "batch", "none"):
            want = ref_group_advantage(rpf, w, K, scale)
            got = group_advantages(rpf.to(dev), w.to(dev), K, scale)
            for a, b, name in zip(got, want, ("rewards", "adv", "mean", "std")):
                a = a.cpu()
                tol = 1e-6 * max(1.0, float(b.abs().max()))
                if name == "adv":
                    # (r - mean) / (std + 1e-4): in a zero-variance group the reference divides the rounding
                    # noise of its own fp32 mean (one ulp of the reward) by 1e-4; the kernel forms the mean in
                    # float64 and returns exactly 0 there, so allow ulp(reward) / 1e-4
                    tol = 1.2e-7 * max(1.0, float(want[0].abs().max())) / 1e-4 if scale != "none" else 1e-6
                b = b.expand_as(a) if b.dim() == 0 else b
                bad = ~torch.isclose(a, b, rtol=1e-5, atol=tol, equal_nan=True)
                assert not bad.any(), (K, scale, name, a[bad][:4], b[bad][:4], torch.nonzero(bad).flatten()[:4])
        v = torch.randn(G * K)
        assert torch.allclose(group_normalize(v.to(dev), K).cpu(), ref_group_normalize(v, K), rtol=1e-5, atol=1e-5)
        tv = torch.randn(G * K, 7)
        tm = (torch.rand(G * K, 7) > 0.3).float()
        tm[:K, 2] = 0  # a position nobody reaches
        got = group_normalize_tokenwise(tv.to(dev), tm.to(dev), K).cpu()
        assert torch.allclose(got, ref_group_normalize_tokenwise(tv, tm, K), rtol=1e-5, atol=1e-6)
    # worked example of docs/2026-03-13-genrec-rl-reward-forms.md:283-301 (prefix_token_only)
    raw = torch.tensor([[0.25, 0.25, 0.25, 0.25], [0.25, 0.25, 0.25, 0.0], [0.25, 0, 0, 0], [0, 0, 0, 0]])
    got = group_normalize_tokenwise(raw.to(dev), torch.ones(4, 4, device=dev), 4).cpu()
    want = torch.tensor([[0.5749, 0.9968, 0.9968, 1.7247], [0.5749, 0.9968, 0.9968, -0.5749],
                         [0.5749, -0.9968, -0.9968, -0.5749], [-1.7247, -0.9968, -0.9968, -0.5749]])
    assert torch.allclose(got, want, atol=1e-4)
    pen = ndcg_rank_penalties(8, 4, dev, process_index=1).cpu()
    assert [round(float(x), 4) for x in pen] == [-0.3904, -0.2463, -0.1952, -0.1681] * 2


def test_dynamic_hint_cascade_makes_the_reference_decisions(monkeypatch):
    """The device cascade (hinted prompts by indexing, rule hits from the reward kernel, masked
    selection) reproduces the decisions recorded from the reference's _run_dynamic_hint_cascade."""
    from helpers import cascade_case_tensors, load_json_gz

    import genrec_amd.cascade as cascade
    from genrec_amd.rewards.ranking_reward import SidRewardScorer

    dev = _dev()
    for n, case in enumerate(load_json_gz("cascade.json.gz")["cases"]):
        tok, prompts, gt, caps, table = cascade_case_tensors(case, dev, seed=n)
        K, L, want = case["K"], case["L"], case["expected"]
        scorer = SidRewardScorer(tok.code_table(), K, dev)
        calls, stage = [], [0]

        def generate(stage_prompts, depths, groups):
            T = prompts.shape[1]
            for row, g, d in zip(stage_prompts.tolist(), groups.tolist(), depths.tolist()):
                width = len(row)
                assert row[width - T - d:] == prompts[g].tolist() + gt[g, :d].tolist()
                assert all(t == tok.pad_token_id for t in row[: width - T - d])
            calls.append({"groups": groups.tolist(), "depths": depths.tolist()})
            return torch.cat([table[(g, d)] for g, d in zip(groups.tolist(), depths.tolist())], 0)

        def others(n_left, device, group=None):  # the scripted state of the other ranks
            flag = case["others"][stage[0]] if case["others"] is not None else False
            stage[0] += 1
            return n_left > 0 or flag

        monkeypatch.setattr(cascade, "any_rank_unresolved", others)
        got = cascade.run_dynamic_hint_cascade(generate, prompts, gt, K, case["max_depth"], scorer,
                                               tok.pad_token_id, depth_cap=caps)
        assert got.hint_depth.tolist() == want["selected_group_hint_depths"]
        assert got.prompt_hint_depth.tolist() == want["selected_prompt_hint_depths"]
        assert got.rule_hit_any.tolist() == want["selected_group_rule_hits"]
        assert got.rule_rewards.tolist() == want["selected_rule_rewards"]
        assert got.stage_stats == want["stage_stats"]
        assert calls == want["calls"]
        codes = tok.code_table()
        rows = [[int(codes[t]) for t in row if codes[t] >= 0] for row in got.completion_ids.tolist()]
        assert rows == want["selected_completions"]


def test_result_json_and_metrics_straight_from_beam_ids(tmp_path):
    """result.json written from beam token ids has the reference's layout (evaluate.py:382-389); the
    metrics computed from the ids equal the metrics the text path computes from that file and the
    oracle's restatement of evaluate.compute_metrics."""
    import json

    from oracle.scoring import ref_compute_metrics

    from genrec_amd.metrics import compute_metrics
    from genrec_amd.results import token_text_table, write_result_json
    from genrec_amd.synth import SidTokenizer

    dev = _dev()
    tok = SidTokenizer(f"<{l}_{c}>" for l in "abcd" for c in range(8))
    table = token_text_table(tok, tok.vocab_size_padded)
    rng = np.random.default_rng(3)
    n, K, L = 64, 20, 4
    items = rng.integers(0, 8, size=(n, K, L))
    items[:, 1:, 3] = (items[:, :1, 3] + np.arange(1, K)[None, :]) % 8 + 8 * 0  # keep beams of a sample distinct-ish
    target = items[np.arange(n), rng.integers(0, K, size=n)].copy()
    target[::5] = 7  # some targets nobody predicted (unless by accident)

    def ids_of(codes):
        return [tok.encode(f"<{'abcd'[l]}_{int(c)}>")[0] for l, c in enumerate(codes)]

    beams = np.array([[ids_of(b) + [tok.eos_token_id] for b in sample] for sample in items])
    beams[:, :, -1] = np.where(rng.random((n, K)) < 0.5, tok.eos_token_id, tok.pad_token_id)
    tgt = np.array([ids_of(t) for t in target])
    test_data = [{"instruction": f"q{i}", "output": "".join(f"<{'abcd'[l]}_{int(c)}>" for l, c in enumerate(t))}
                 for i, t in enumerate(target)]
    path = str(tmp_path / "run" / "result.json")
    metrics = write_result_json(test_data, torch.from_numpy(beams).to(dev), table, path,
                                target_ids=torch.from_numpy(tgt).to(dev))
    with open(path, encoding="utf-8") as f:
        text = f.read()
    records = json.loads(text)
    assert text == json.dumps(records, indent=4, ensure_ascii=False)
    assert [r["predict"] for r in records] == [[tok.decode(b) for b in sample] for sample in beams]
    assert all(r["output"] == d["output"] and r["instruction"] == d["instruction"] for r, d in zip(records, test_data))
    want = ref_compute_metrics(records)
    assert metrics.keys() == want.keys()
    assert all(abs(metrics[k] - want[k]) <= 1e-12 for k in want)
    from_text = compute_metrics(records)
    assert all(abs(from_text[k] - want[k]) <= 1e-12 for k in want)
    saved = json.load(open(tmp_path / "run" / "metrics.json"))
    assert saved == {k: round(v, 4) for k, v in want.items()}


def test_ragged_prediction_lists_and_numbers_beyond_32_bits_follow_the_reference():
    """Edge cases recorded from the reference (tests/golden/make_golden_edge.py): compute_metrics on
    records whose prediction lists differ in length -- cut-offs from the first sample, each sample
    skipped for every k beyond its own list (evaluate.py:137-143, 168-170) -- and the text rewards on
    numbers that do not fit int32 / int64 (the reference compares Python integers)."""
    from genrec_amd.metrics import compute_metrics
    from genrec_amd.rewards import ranking_reward as rr

    _dev()
    g = load_json_gz("scoring_edge.json.gz")
    for name, case in g["metrics"].items():
        got = compute_metrics(case["data"])
        assert list(got) == list(case["metrics"]), name
        for k, v in case["metrics"].items():
            assert abs(got[k] - v) <= 1e-12, (name, k)
    kw = dict(reward_model=[{"ground_truth": t} for t in g["gts"]])
    assert rr.rule_reward(None, g["comps"], None, **kw) == g["rule"]
    assert rr.get_ndcg_rule_reward(g["K"])(None, g["comps"], None, **kw) == pytest.approx(g["ndcg"], abs=1e-7)
    assert rr.get_prefix_rule_reward(True)(None, g["comps"], None, **kw) == pytest.approx(g["prefix"], abs=1e-7)


@pytest.mark.parametrize("K,want", [(16, ("rule", "ndcg")), (4, ("ndcg", "rule", "prefix")), (50, ("prefix",))])
def test_one_launch_scoring_equals_rewards_then_advantages(K, want):
    """grx_rollout_score_tokens: the reward columns and the local advantages of a rollout shard in one
    kernel, bit for bit what the reward kernel followed by the advantage kernel produce."""
    from genrec_amd.advantage import group_advantages, rollout_score
    from genrec_amd.rewards.ranking_reward import SidRewardScorer

    dev = _dev()
    g = torch.Generator().manual_seed(K)
    G, L = 37, 4
    table = torch.full((700,), -1, dtype=torch.int32)
    table[100:700] = torch.arange(600, dtype=torch.int32) % 41
    gt = torch.randint(100, 700, (G, L), generator=g).repeat_interleave(K, dim=0)
    comp = torch.randint(100, 700, (G * K, L + 1), generator=g)
    comp[:, L] = 5  # EOS-like token without a code
    hit = torch.rand(G * K, generator=g) < 0.25
    comp[hit, :L] = gt[hit]
    comp[K : 2 * K, :L] = gt[K : 2 * K]  # zero-variance group
    w = torch.rand(len(want), generator=g).to(dev)
    scorer = SidRewardScorer(table.numpy(), K, dev)
    for scale in ("group", "none"):
        cols = scorer(comp.to(dev), gt.to(dev), want=want)
        rpf = torch.stack([cols[k] for k in want], dim=1).contiguous()
        ref = group_advantages(rpf, w, K, scale)
        res = rollout_score(scorer, comp.to(dev), gt.to(dev), w, scale, want=want)
        assert torch.equal(res.gathered_rewards_per_func(), rpf)
        assert torch.equal(res.local_advantages, ref[1]) and torch.equal(res.local_rewards, ref[0])
        assert torch.equal(res.local_group_mean, ref[2]) and torch.equal(res.local_std, ref[3])
        for a, b in zip(res.wait_global(), ref):
            assert torch.equal(a, b)
    res = rollout_score(scorer, comp.to(dev), gt.to(dev), w, "batch", want=want)  # falls back to the two-step path
    ref = group_advantages(rpf, w, K, "batch")
    assert torch.equal(res.local_advantages, ref[1])
    with pytest.raises(ValueError):
        scorer.score_with_advantages(comp.to(dev), gt.to(dev), w, "batch", want=want)


@pytest.mark.parametrize("scale", ["group", "batch", "none"])
def test_logged_statistics_come_back_in_one_read(scale):
    """fixed_hint_grpo_trainer.py:410-417 restated with the torch operations the reference uses (TRL's nanstd:
    nan-aware mean of squared deviations, times count / (count - 1)), NaN rewards included."""
    from genrec_amd.advantage import rollout_advantages

    dev = _dev()
    K, G = 8, 12
    torch.manual_seed(4)
    rpf = torch.randn(G * K, 3)
    rpf[::7, 1] = float("nan")  # a reward function that returned None for some rows
    rpf[K : 2 * K] = 0.25       # a group without spread
    w = torch.tensor([1.0, 0.5, 2.0])
    res = rollout_advantages(rpf.to(dev), w.to(dev), K, scale)
    got = res.logged_statistics(["rule_reward", "prefix_rule_reward", "ndcg_rule_reward"])
    rewards = (rpf * w.unsqueeze(0)).nansum(dim=1)
    mean = rewards.view(-1, K).mean(dim=1).repeat_interleave(K)
    std = rewards.std().expand_as(rewards) if scale == "batch" else rewards.view(-1, K).std(dim=1).repeat_interleave(K)
    want = {"reward": mean.mean().item(), "reward_std": std.mean().item(),
            "frac_reward_zero_std": torch.isclose(std, torch.zeros_like(std)).float().mean().item()}
    for i, name in enumerate(["rule_reward", "prefix_rule_reward", "ndcg_rule_reward"]):
        col = rpf[:, i]
        want[f"rewards/{name}/mean"] = torch.nanmean(col).item()
        var = torch.nanmean((col - torch.nanmean(col, keepdim=True)) ** 2)
        n = torch.sum(~torch.isnan(col))
        want[f"rewards/{name}/std"] = torch.sqrt(var * n / (n - 1)).item()
    assert set(got) == set(want)
    for k in want:
        assert abs(got[k] - want[k]) <= 1e-5 * max(1.0, abs(want[k])), (k, got[k], want[k])
    assert got["frac_reward_zero_std"] > 0 or scale == "batch"
    with pytest.raises(ValueError):
        res.logged_statistics(["only_one"])
