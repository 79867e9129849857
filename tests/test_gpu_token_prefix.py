"""GPU: token-level prefix advantages (``grx_token_prefix_advantages``) against outputs recorded from the
reference's own ``TokenPrefixGRPOTrainer._generate_and_score_completions`` and reward builders
(tests/golden/make_golden_token_prefix.py -> token_prefix.npz): token rewards, sequence rewards and the
error-token mask bit for bit; advantages within 1e-5 (the group statistics are accumulated in fp64 here,
in fp32 there)."""

import numpy as np
import pytest
import torch

from helpers import GOLDEN as GOLDEN_DIR

pytestmark = pytest.mark.gpu


def _dev():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch.device("cuda", 0)


def _close(got, want, tag):
    err = (got - want).abs()
    tol = 1e-5 + 1e-5 * want.abs()
    assert bool((err <= tol).all()), (tag, float(err.max()))


def test_recorded_rollouts_of_the_reference_trainer():
    import os

    from genrec_amd.token_advantage import token_prefix_advantages

    dev = _dev()
    d = np.load(os.path.join(GOLDEN_DIR, "token_prefix.npz"))
    n, eos = int(d["n_cases"][0]), int(d["n_cases"][1])
    seen, n_ratio = set(), 0
    for c in range(n):
        K, normalize, has_ndcg, total, err_pen = (int(v) for v in d[f"c{c}.cfg"])
        pw, nw, ratio = (float(v) for v in d[f"c{c}.weights"])
        seen.add((has_ndcg, total, err_pen))
        ids = torch.from_numpy(d[f"c{c}.ids"]).to(dev)
        mask = torch.from_numpy(d[f"c{c}.mask"].astype(np.int64)).to(dev)
        ndcg = torch.from_numpy(d[f"c{c}.ndcg"]).to(dev) if has_ndcg else None
        res = token_prefix_advantages(
            ids, mask, torch.from_numpy(d[f"c{c}.gt_ids"]), torch.from_numpy(d[f"c{c}.gt_len"]), K, pw, nw, ndcg,
            has_ndcg=bool(has_ndcg), eos_token_id=eos, prefix_reward_normalize=bool(normalize),
            token_adv_total_token_normalize=bool(total), token_level_ndcg_error_token_penalty=bool(err_pen))
        tag = (c, K, normalize, has_ndcg, total, err_pen)
        assert torch.equal(res.token_rewards.cpu(), torch.from_numpy(d[f"c{c}.token_rewards"])), tag
        assert torch.equal(res.seq_rewards.cpu(), torch.from_numpy(d[f"c{c}.seq_rewards"])), tag
        assert torch.equal(res.error_token_mask.cpu(), torch.from_numpy(d[f"c{c}.error_mask"].astype(np.float32))), tag
        _close(res.token_advantages.cpu(), torch.from_numpy(d[f"c{c}.advantages"]), tag)
        # the logged share of non-zero advantages: comparable unless the reference's own fp32 rounding noise
        # (a group of equal rewards: noise / 1e-4, where the fp64 mean here gives exactly 0) counts as non-zero
        ref_adv = d[f"c{c}.advantages"]
        if not bool(((ref_adv != 0) & (np.abs(ref_adv) < 1e-4)).any()):
            assert abs(float(res.nonzero_ratio()) - ratio) <= 1e-6, tag
            n_ratio += 1
    assert len(seen) == 6 and n_ratio >= 40, n_ratio


@pytest.mark.parametrize("K,total,err_pen", [(16, False, False), (16, True, False), (16, True, True), (50, True, True),
                                             (50, False, False)])
def test_a_rollout_shard_against_the_restatement(K, total, err_pen):
    """8,000-8,192 rows (a rank's shard of the rollout configuration), ragged completions and masks,
    second rank's rank-penalty offset; the CPU restatement of the reference as the checker."""
    from oracle import scoring as oscore

    from genrec_amd.token_advantage import pack_ground_truth_ids, token_prefix_advantages

    dev = _dev()
    rng = np.random.default_rng(K + 2 * total + err_pen)
    groups = 8192 // 16 if K == 16 else 160
    R, levels, Tc, eos = groups * K, 4, 6, 2
    gts, ids, mask = [], np.zeros((R, Tc), dtype=np.int64), np.zeros((R, Tc), dtype=np.int64)
    for g in range(groups):
        gt = [1000 + 256 * l + int(rng.integers(0, 4)) for l in range(levels if rng.random() < 0.9 else int(rng.integers(0, 4)))]
        for k in range(K):
            r = g * K + k
            gts.append(gt)
            depth = int(rng.integers(0, levels + 1))
            toks = list(gt[:depth]) + [1000 + 256 * l + int(rng.integers(0, 4)) for l in range(min(depth, len(gt)), levels)]
            toks = toks[: int(rng.choice([0, 2, 3, 4], p=[0.03, 0.07, 0.1, 0.8]))]
            if rng.random() < 0.85:
                toks.append(eos)
            ids[r, : len(toks)] = toks
            mask[r, : len(toks)] = 1
            if rng.random() < 0.05 and len(toks) > 1:
                mask[r, int(rng.integers(0, len(toks)))] = 0
    ndcg = torch.from_numpy((rng.random(R) * -0.4 * (rng.random(R) < 0.6)).astype(np.float32))
    pw, nw, proc = 1.0, 0.5, 1
    tids, tmask = torch.from_numpy(ids), torch.from_numpy(mask)
    o_tr, o_sr, o_em = oscore.ref_prefix_token_rewards(tids, tmask, gts, eos, True, True)
    # the restatement's slice of a two-rank gather: this shard is rank 1, rank 0's rows are zeros
    pad = lambda t: torch.cat([torch.zeros_like(t), t])
    want = oscore.ref_token_prefix_advantages(o_tr, o_sr, o_em, tmask, K, pw, nw, True, ndcg, total, err_pen,
                                              process_index=proc, gather=pad, n_local=R)
    gt_ids, gt_len = pack_ground_truth_ids(gts, dev)
    res = token_prefix_advantages(tids.to(dev), tmask.to(dev), gt_ids, gt_len, K, pw, nw, ndcg.to(dev), has_ndcg=True,
                                  eos_token_id=eos, token_adv_total_token_normalize=total,
                                  token_level_ndcg_error_token_penalty=err_pen, process_index=proc)
    assert torch.equal(res.token_rewards.cpu(), o_tr) and torch.equal(res.seq_rewards.cpu(), o_sr)
    assert torch.equal(res.error_token_mask.cpu(), o_em)
    got = res.token_advantages.cpu()
    assert float(o_sr.max()) == 1.0 and int((want != 0).sum()) > R
    # groups whose rewards are all equal: the reference divides its own fp32 rounding noise by 1e-4
    err = (got - want).abs()
    assert bool((err <= 1e-5 + 1e-5 * want.abs() + 2e-3 * (want.abs() < 2e-3)).all()), float(err.max())


def test_operands_are_checked_before_anything_is_launched():
    from genrec_amd import _lib
    from genrec_amd.token_advantage import token_prefix_advantages

    dev = _dev()
    ids = torch.zeros((6, 4), dtype=torch.int64, device=dev)
    gt, gl = torch.zeros((6, 2), dtype=torch.int64), torch.zeros(6, dtype=torch.int32)
    with pytest.raises(_lib.GenrecHipError, match="groups of 4"):
        token_prefix_advantages(ids, torch.ones_like(ids), gt, gl, 4, 1.0)
    with pytest.raises(ValueError, match="ndcg_rewards"):
        token_prefix_advantages(ids, torch.ones_like(ids), gt, gl, 3, 1.0, 0.5, None, has_ndcg=True)
    with pytest.raises(ValueError, match="agree"):
        token_prefix_advantages(ids, torch.ones((6, 5), device=dev), gt, gl, 3, 1.0)
    empty = token_prefix_advantages(ids[:0], torch.ones_like(ids[:0]), gt[:0], gl[:0], 3, 1.0)
    assert empty.token_advantages.shape == (0, 4)
