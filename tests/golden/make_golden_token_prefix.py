"""Golden token-level advantages, recorded from the REAL reference.

Run in the build container only (needs the reference checkout; it never ships):

    python tests/golden/make_golden_token_prefix.py [--reference /root/reference]

Drives ``TokenPrefixGRPOTrainer._generate_and_score_completions`` (token_prefix_grpo_trainer.py:174-287)
and its two token-space reward builders (:65-149) on randomized rollouts.  `trl` is not installed here:
it gets a stand-in whose ``GRPOTrainer._generate_and_score_completions`` hands back the prepared
completion ids and mask (what TRL's generation would have produced); the tokenizer is a table; the
reward functions are the reference's own (rewards/ranking_reward.py); everything from the prefix match
to the token advantages is the reference's code, including ``accelerate.utils.gather``.  The CPU
restatement in oracle/scoring.py is asserted equal on every case before the fixture (inputs + the
reference's outputs) is written.
"""

from __future__ import annotations

import argparse
import collections
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

from make_golden import save_npz  # noqa: E402

from oracle import scoring as oscore  # noqa: E402

LETTERS = "abcd"
SID_BASE, EOS, PAD = 1000, 2, 0


class StubBase:
    """Stands in for trl.GRPOTrainer: generation already happened."""

    def __init__(self, *a, **k):
        pass

    def _generate_and_score_completions(self, inputs):
        return {"completion_ids": self._prepared_ids.clone(), "completion_mask": self._prepared_mask.clone()}


def load_reference(ref_root: str):
    for name in ("fire",):
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    trl = types.ModuleType("trl")
    trl.GRPOTrainer = StubBase
    trl.GRPOConfig = type("GRPOConfig", (), {})
    sys.modules["trl"] = trl
    sys.path.insert(0, ref_root)
    import rewards.ranking_reward as ref_rewards
    import token_prefix_grpo_trainer as ref_tp

    return ref_tp, ref_rewards


def sid_token(level: int, code: int) -> int:
    return SID_BASE + 256 * level + code


class Tok:
    """Tokenizer table: `<a_12><b_3>` <-> SID token ids; anything else decodes to nothing."""

    eos_token_id = EOS

    def __call__(self, text, add_special_tokens=False):
        import re

        ids = [sid_token(LETTERS.index(l), int(c)) for l, c in re.findall(r"<([a-d])_(\d+)>", text)]
        return types.SimpleNamespace(input_ids=ids)

    def batch_decode(self, ids, skip_special_tokens=True):
        out = []
        for row in ids.tolist():
            s = ""
            for t in row:
                if t >= SID_BASE:
                    level, code = divmod(t - SID_BASE, 256)
                    s += f"<{LETTERS[level]}_{code}>"
            out.append(s)
        return out


def make_rollout(rng, groups, K, levels, comp_len):
    """Completions that share prefixes of their ground truth to varying depth; some without EOS, some short,
    some rows masked out entirely, some with holes in the mask, some ground truths empty or short."""
    R = groups * K
    ids = np.full((R, comp_len), PAD, dtype=np.int64)
    mask = np.zeros((R, comp_len), dtype=np.int64)
    gts = []
    for g in range(groups):
        n_gt = int(rng.choice([0, 1, levels - 1, levels, levels], p=[0.05, 0.05, 0.1, 0.4, 0.4]))
        gt = [int(rng.integers(0, 6)) for _ in range(n_gt)]
        gt_text = "".join(f"<{LETTERS[l]}_{c}>" for l, c in enumerate(gt))
        for k in range(K):
            r = g * K + k
            gts.append(gt_text)
            depth = int(rng.integers(0, levels + 1))
            codes = list(gt[:depth]) + [int(rng.integers(0, 6)) for _ in range(levels - min(depth, len(gt)))]
            codes = codes[:levels]
            n = int(rng.choice([0, 1, levels - 1, levels], p=[0.05, 0.1, 0.15, 0.7]))
            toks = [sid_token(l, c) for l, c in enumerate(codes[:n])]
            if rng.random() < 0.8:
                toks.append(EOS)
            toks = toks[:comp_len]
            ids[r, : len(toks)] = toks
            mask[r, : len(toks)] = 1
            if rng.random() < 0.1 and len(toks) > 1:  # a hole in the mask
                mask[r, int(rng.integers(0, len(toks)))] = 0
    return ids, mask, gts


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reference", default="/root/reference")
    args = ap.parse_args()
    ref_tp, ref_rewards = load_reference(args.reference)
    rng = np.random.default_rng(77)
    arrays, n_cases = {}, 0
    tok = Tok()
    for K in (4, 8):
        for levels in (3, 4):
            for normalize in (True, False):
                for funcs in ("prefix_ndcg", "prefix_only_func", "prefix_rule_ndcg"):
                    for total_norm, err_pen in ((False, False), (True, False), (True, True)):
                        groups = int(rng.integers(2, 6))
                        comp_len = levels + int(rng.integers(1, 3))
                        ids, mask, gts = make_rollout(rng, groups, K, levels, comp_len)
                        R = groups * K
                        prefix_fn = ref_rewards.get_prefix_rule_reward(normalize=normalize)
                        ndcg_fn = ref_rewards.get_ndcg_rule_reward(K)
                        if funcs == "prefix_ndcg":
                            reward_funcs, weights = [prefix_fn, ndcg_fn], [1.0, 0.5]
                        elif funcs == "prefix_only_func":
                            reward_funcs, weights = [prefix_fn, ref_rewards.rule_reward], [0.7, 0.0]
                        else:
                            reward_funcs, weights = [prefix_fn, ref_rewards.rule_reward, ndcg_fn], [1.0, 0.0, 2.0]
                        t = ref_tp.TokenPrefixGRPOTrainer(prefix_reward_normalize=normalize,
                                                          token_adv_total_token_normalize=total_norm,
                                                          token_level_ndcg_error_token_penalty=err_pen)
                        t._prepared_ids = torch.from_numpy(ids)
                        t._prepared_mask = torch.from_numpy(mask)
                        t.reward_funcs = reward_funcs
                        t.reward_weights = torch.tensor(weights, dtype=torch.float32)
                        t.num_generations = K
                        t.processing_class = tok
                        t.accelerator = types.SimpleNamespace(process_index=0, device=torch.device("cpu"),
                                                              gather_for_metrics=lambda x: x.reshape(1))
                        t.control = types.SimpleNamespace(should_evaluate=False)
                        t._metrics = {"train": collections.defaultdict(list), "eval": collections.defaultdict(list)}
                        inputs = [{"prompt": "p", "reward_model": {"ground_truth": g}} for g in gts]
                        out = t._generate_and_score_completions(inputs)
                        adv = out["advantages"]
                        rm = [{"ground_truth": g} for g in gts]
                        tr, sr, em = t._build_prefix_token_rewards_with_error_mask(out["completion_ids"],
                                                                                   out["completion_mask"], rm)
                        tr0, sr0 = t._build_prefix_token_rewards(out["completion_ids"], out["completion_mask"], rm)
                        # ---- the restatement, on the same inputs
                        gt_tok = [tok(g).input_ids for g in gts]
                        o_tr, o_sr, o_em = oscore.ref_prefix_token_rewards(torch.from_numpy(ids), torch.from_numpy(mask),
                                                                           gt_tok, EOS, normalize, True)
                        assert torch.equal(o_tr, tr) and torch.equal(o_sr, sr) and torch.equal(o_em, em)
                        p_tr, p_sr, _ = oscore.ref_prefix_token_rewards(torch.from_numpy(ids), torch.from_numpy(mask),
                                                                        gt_tok, EOS, normalize, False)
                        assert torch.equal(p_tr, tr0) and torch.equal(p_sr, sr0)
                        has_ndcg = funcs != "prefix_only_func"
                        texts = tok.batch_decode(torch.from_numpy(ids))
                        ndcg_local = (torch.tensor(ndcg_fn(prompts=None, completions=texts, completion_ids=None,
                                                           reward_model=rm), dtype=torch.float32)
                                      if has_ndcg else None)
                        pw = weights[0]
                        nw = weights[-1] if has_ndcg else 0.0
                        o_adv = oscore.ref_token_prefix_advantages(o_tr, o_sr, o_em, torch.from_numpy(mask), K, pw, nw,
                                                                   has_ndcg, ndcg_local, total_norm, err_pen)
                        assert torch.equal(o_adv, adv), (K, levels, normalize, funcs, total_norm, err_pen)
                        ratio = float(t._metrics["train"]["token_adv_nonzero_ratio"][0])
                        c = f"c{n_cases}"
                        arrays[f"{c}.ids"], arrays[f"{c}.mask"] = ids, mask.astype(np.int8)
                        width = max(1, max(len(g) for g in gt_tok))
                        gt_arr = np.full((R, width), -1, dtype=np.int64)
                        for i, g in enumerate(gt_tok):
                            gt_arr[i, : len(g)] = g
                        arrays[f"{c}.gt_ids"] = gt_arr
                        arrays[f"{c}.gt_len"] = np.asarray([len(g) for g in gt_tok], dtype=np.int32)
                        arrays[f"{c}.ndcg"] = (ndcg_local.numpy() if has_ndcg else np.zeros(0, dtype=np.float32))
                        arrays[f"{c}.cfg"] = np.asarray([K, int(normalize), int(has_ndcg), int(total_norm), int(err_pen)],
                                                        dtype=np.int32)
                        arrays[f"{c}.weights"] = np.asarray([pw, nw, ratio], dtype=np.float64)
                        arrays[f"{c}.token_rewards"], arrays[f"{c}.seq_rewards"] = tr.numpy(), sr.numpy()
                        arrays[f"{c}.error_mask"] = em.numpy().astype(np.int8)
                        arrays[f"{c}.advantages"] = adv.numpy()
                        n_cases += 1
    arrays["n_cases"] = np.asarray([n_cases, EOS], dtype=np.int32)
    save_npz(os.path.join(HERE, "token_prefix.npz"), **arrays)
    print(f"{n_cases} cases: oracle == reference (builders and advantages, bit for bit)")


if __name__ == "__main__":
    main()
