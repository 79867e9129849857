"""Oracle (test infrastructure): rewards, ranking metrics and group advantages on the CPU.

Restates:
  * ``_extract_number``                       -- util.py:133-134
  * ``rule_reward`` / prefix / NDCG rewards   -- rewards/ranking_reward.py:25-110
  * ``compute_metrics`` (HR@k, NDCG@k)        -- evaluate.py:125-204 (without the file output)
  * sequence-level group advantage            -- fixed_hint_grpo_trainer.py:383-401
  * ``_group_normalize`` / ``_group_normalize_tokenwise`` / NDCG rank penalties
                                              -- token_prefix_grpo_trainer.py:41-63,151-161
  * token-space prefix rewards, error-token mask and the token advantages composed from them
                                              -- token_prefix_grpo_trainer.py:65-149,196-278
"""

from __future__ import annotations

import math
import re
from typing import Optional

import torch


def extract_numbers(text: str) -> list[int]:
    return [int(m) for m in re.findall(r"_(\d+)", text)]


def _completion_text(completion) -> str:  # ranking_reward.py:15-22
    if isinstance(completion, list) and completion and isinstance(completion[0], dict):
        return completion[0].get("content", "")
    if isinstance(completion, dict):
        return completion.get("content", "")
    return str(completion)


def full_completion_text(i: int, completion, hint_texts) -> str:  # ranking_reward.py:25-30
    text = _completion_text(completion)
    return f"{hint_texts[i]}{text}" if hint_texts is not None else text


def ref_rule_reward(completions, ground_truths, hint_texts=None) -> list[float]:  # ranking_reward.py:42-52
    out = []
    for i, c in enumerate(completions):
        same = extract_numbers(full_completion_text(i, c, hint_texts)) == extract_numbers(ground_truths[i])
        out.append(1.0 if same else 0.0)
    return out


def ref_prefix_reward(completions, ground_truths, hint_texts=None, normalize=True) -> list[float]:
    out = []  # ranking_reward.py:55-79
    for i, c in enumerate(completions):
        gt = extract_numbers(ground_truths[i])
        pred = extract_numbers(full_completion_text(i, c, hint_texts))
        if not gt:
            out.append(0.0)
            continue
        matched = 0
        for p, g in zip(pred, gt):
            if p != g:
                break
            matched += 1
        out.append(float(matched) / float(len(gt)) if normalize else float(matched))
    return out


def ndcg_rank_weights(num_beams: int) -> list[float]:  # ranking_reward.py:83-84
    w = [-1.0 / math.log2(i + 2) for i in range(num_beams)]
    total = sum(w)
    return [-x / total for x in w]


def ref_ndcg_reward(completions, ground_truths, num_beams: int, hint_texts=None) -> list[float]:
    weights = ndcg_rank_weights(num_beams)  # ranking_reward.py:86-108
    out: list[float] = []
    group: list[float] = []
    hit = False
    for i, c in enumerate(completions):
        exact = extract_numbers(full_completion_text(i, c, hint_texts)) == extract_numbers(ground_truths[i])
        hit = hit or exact
        group.append(0.0 if exact else weights[i % num_beams])
        if (i + 1) % num_beams == 0:
            out.extend(group if hit else [0.0] * num_beams)
            group, hit = [], False
    return out


def ref_compute_metrics(test_data: list, topk_list: Optional[list] = None) -> dict:  # evaluate.py:125-180
    if topk_list is None:
        topk_list = [1, 3, 5, 10, 20, 50]
    n = len(test_data)
    if n == 0:
        return {}
    first = test_data[0].get("predict", [])
    if isinstance(first, str):
        n_beam = 1 if first else 0
    else:
        n_beam = len(first) if first else 0
    valid = [k for k in topk_list if k <= n_beam] if n_beam > 0 else topk_list
    ndcg = dict.fromkeys(valid, 0.0)
    hr = dict.fromkeys(valid, 0)
    for sample in test_data:
        predict = sample.get("predict", "")
        target = sample.get("output", "")
        if isinstance(predict, str):
            preds = [predict.strip('"\n').strip()] if predict else [""]
        else:
            preds = [str(p).strip('"\n').strip() for p in predict]
        if isinstance(target, list):
            tgt = target[0].strip('"\n').strip(" ") if target else ""
        else:
            tgt = str(target).strip(' \n"')
        rank = len(preds)
        for i, p in enumerate(preds):
            if p == tgt:
                rank = i
                break
        for k in valid:
            if k > len(preds):
                continue
            if rank < k:
                hr[k] += 1
                ndcg[k] += 1.0 / math.log(rank + 2)
    out = {}
    for k in valid:
        out[f"HR@{k}"] = hr[k] / n
        out[f"NDCG@{k}"] = ndcg[k] / n / (1.0 / math.log(2))
    return out


def ref_group_advantage(rewards_per_func: torch.Tensor, weights: torch.Tensor, num_generations: int,
                        scale_rewards: str = "group"):
    """fixed_hint_grpo_trainer.py:383-401 -> (rewards, advantages, mean, std)."""
    rewards = (rewards_per_func * weights.unsqueeze(0)).nansum(dim=1)
    mean = rewards.view(-1, num_generations).mean(dim=1).repeat_interleave(num_generations, dim=0)
    adv = rewards - mean
    if scale_rewards in ("group", "none"):
        std = rewards.view(-1, num_generations).std(dim=1).repeat_interleave(num_generations, dim=0)
    elif scale_rewards == "batch":
        std = rewards.std().expand_as(rewards)
    else:
        raise ValueError(f"Invalid value for scale_rewards: {scale_rewards}")
    if scale_rewards != "none":
        adv = adv / (std + 1e-4)
    return rewards, adv, mean, std


def ref_group_normalize(values: torch.Tensor, group_size: int, eps: float = 1e-4) -> torch.Tensor:
    g = values.view(-1, group_size)  # token_prefix_grpo_trainer.py:41-45
    mean = g.mean(dim=1).repeat_interleave(group_size, dim=0)
    std = g.std(dim=1).repeat_interleave(group_size, dim=0)
    return (values - mean) / (std + eps)


def ref_group_normalize_tokenwise(values: torch.Tensor, mask: torch.Tensor, group_size: int, eps: float = 1e-4):
    bsz, seq_len = values.shape  # token_prefix_grpo_trainer.py:47-63
    v = values.view(-1, group_size, seq_len)
    m = mask.view(-1, group_size, seq_len).float()
    denom = m.sum(dim=1).clamp_min(1.0)
    mean = (v * m).sum(dim=1) / denom
    centered = (v - mean.unsqueeze(1)) * m
    var = (centered * centered).sum(dim=1) / denom
    std = torch.sqrt(var + eps)
    return (((v - mean.unsqueeze(1)) / std.unsqueeze(1)) * m).view(bsz, seq_len)


def ref_ndcg_rank_penalties(batch_size: int, num_generations: int, process_index: int = 0) -> torch.Tensor:
    w = torch.tensor([-1.0 / math.log2(i + 2) for i in range(num_generations)], dtype=torch.float32)
    w = -w / w.sum()  # token_prefix_grpo_trainer.py:151-161
    idx = (torch.arange(batch_size) + process_index * batch_size) % num_generations
    return w[idx]


# ----------------------------------------------------------------- token-level prefix advantages ---
def ref_prefix_token_rewards(completion_ids: torch.Tensor, completion_mask: torch.Tensor, gt_token_ids: list,
                             eos_token_id: Optional[int], normalize: bool = True, with_error_mask: bool = True):
    """(token_rewards [R, Tc], seq_rewards [R], error_token_mask [R, Tc] or None): the prefix match of every
    completion against its ground truth IN TOKEN SPACE -- token_prefix_grpo_trainer.py:107-149
    (``with_error_mask=False``: the older builder at :65-105, which differs only in skipping rows whose
    ground truth is empty before anything else).  ``gt_token_ids[i]`` is what the tokenizer returns for row
    i's ground-truth text (:123)."""
    bsz, comp_len = completion_ids.shape
    token_rewards = torch.zeros((bsz, comp_len), dtype=torch.float32)
    seq_rewards = torch.zeros((bsz,), dtype=torch.float32)
    error = torch.zeros((bsz, comp_len), dtype=torch.float32) if with_error_mask else None
    for i in range(bsz):
        gt = [int(t) for t in gt_token_ids[i]]
        if not with_error_mask and len(gt) == 0:
            continue
        valid = [p for p in range(comp_len) if float(completion_mask[i, p]) > 0]
        if not valid:
            continue
        ids = [int(completion_ids[i, p]) for p in valid]
        if ids and eos_token_id is not None and ids[-1] == int(eos_token_id):  # terminal EOS takes no part
            ids, valid = ids[:-1], valid[:-1]
        if with_error_mask and not valid:
            continue
        matched = 0
        while matched < min(len(ids), len(gt)) and ids[matched] == gt[matched]:
            matched += 1
        if len(gt) > 0 and matched > 0:
            per_tok = 1.0 / float(len(gt)) if normalize else 1.0
            for p in valid[:matched]:
                token_rewards[i, p] = per_tok
            seq_rewards[i] = float(matched) * float(per_tok)
        if with_error_mask:
            for p in valid[matched:]:
                error[i, p] = 1.0
    return token_rewards, seq_rewards, error


def ref_token_prefix_advantages(prefix_token_rewards: torch.Tensor, prefix_seq: torch.Tensor,
                                error_token_mask: torch.Tensor, completion_mask: torch.Tensor, num_generations: int,
                                prefix_weight: float, ndcg_weight: float, has_ndcg: bool,
                                ndcg_local: Optional[torch.Tensor], total_token_normalize: bool,
                                error_token_penalty: bool, process_index: int = 0, gather=lambda t: t,
                                n_local: Optional[int] = None) -> torch.Tensor:
    """Token advantages of the token-level trainer -- token_prefix_grpo_trainer.py:201-278.  ``has_ndcg``: an
    ``ndcg_rule_reward`` is among the reward functions (:163-172); ``ndcg_local`` is its output for the local
    rows, needed where the reference calls it (:226-236, :263-273); ``gather`` stands for
    ``accelerate.utils.gather`` (identity on one rank)."""
    n_local = prefix_seq.shape[0] if n_local is None else n_local
    mask = completion_mask.float()
    sl = slice(process_index * n_local, (process_index + 1) * n_local)
    K = num_generations
    if total_token_normalize:
        if not has_ndcg:
            ndcg_tok = torch.zeros_like(prefix_token_rewards)  # :210-211
        elif error_token_penalty:  # :212-223
            positive = gather((prefix_seq > 0).float())
            group_has = positive.view(-1, K).max(dim=1).values.repeat_interleave(K)[sl].unsqueeze(1)
            rank = ref_ndcg_rank_penalties(n_local, K, process_index)
            ndcg_tok = rank.unsqueeze(1) * error_token_mask * group_has
        else:  # :224-237
            ndcg_tok = ndcg_local.unsqueeze(1) * mask
        total = prefix_weight * prefix_token_rewards + ndcg_weight * ndcg_tok  # :240
        return ref_group_normalize_tokenwise(gather(total), gather(mask), K)[sl]  # :241-246
    adv_prefix = ref_group_normalize(gather(prefix_seq), K)[sl]  # :249-251
    mass = prefix_token_rewards.sum(dim=1, keepdim=True)
    dist = torch.where(mass > 0, prefix_token_rewards / (mass + 1e-8), torch.zeros_like(prefix_token_rewards))
    out = prefix_weight * adv_prefix.unsqueeze(1) * dist  # :255-261
    if has_ndcg:  # :262-277
        adv_ndcg = ref_group_normalize(gather(ndcg_local), K)[sl]
        out = out + ndcg_weight * adv_ndcg.unsqueeze(1) * mask
    return out
