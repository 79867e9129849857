"""Dynamic-hint cascade on token tensors.

The reference's ``DynamicHintRuleOnlyGRPOTrainer._run_dynamic_hint_cascade``
(fixed_hint_grpo_trainer.py:558-693) re-runs generation for the prompt groups
that found no exact item, each time with one more ground-truth SID token
appended to the prompt, until every group has a hit or reached its depth cap.
It works on text: hint strings, decoded completions, a regular expression per
row.  Here the same control flow runs on device tensors: hinted prompts are
built by indexing, rule hits come from the reward kernel, group selection is a
masked copy over the whole stage batch, and the host reads back one packed triple per stage --
groups hit, groups selected, "some rank still has unresolved groups" -- to size the next stage
(the reference synchronises there as well: ``accelerator.gather(flag).max().item()``,
fixed_hint_grpo_trainer.py:511-521).
"""

from __future__ import annotations

from dataclasses import dataclass, field
from typing import Callable, Optional

import torch


@dataclass
class CascadeResult:
    completion_ids: torch.Tensor      # [G * K, Lc] completions of the selected stage of every group
    prompt_ids: torch.Tensor          # [G, T + max_hint_depth] hinted prompts of that stage (left-padded)
    hint_depth: torch.Tensor          # [G] int32 stage index the group was selected at
    prompt_hint_depth: torch.Tensor   # [G] int32 hint tokens actually in the prompt (stage index capped per group)
    rule_hit_any: torch.Tensor        # [G] bool
    rule_rewards: torch.Tensor        # [G * K] fp32
    stage_stats: list = field(default_factory=list)


def any_rank_unresolved(n_local_unresolved: int, device, group=None) -> bool:
    """True while some rank still has unresolved groups (fixed_hint_grpo_trainer.py:511-521)."""
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return n_local_unresolved > 0
    backend = dist.get_backend(group)
    flag_device = device if backend == "nccl" else torch.device("cpu")
    flag = torch.tensor([1 if n_local_unresolved > 0 else 0], dtype=torch.int32, device=flag_device)
    dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=group)
    return bool(flag.item())


def _stage_sync(counts: torch.Tensor, n_local: int, device, group=None) -> tuple[int, int, bool]:
    """The one device-to-host read of a cascade stage: ``counts`` = [groups with a rule hit, groups
    selected] of this rank's stage batch of ``n_local`` groups, plus the consensus whether any rank
    still has unresolved groups (fixed_hint_grpo_trainer.py:511-521).  With RCCL the consensus is
    reduced on the device ahead of the read, so the three numbers come back together."""
    import torch.distributed as dist

    alone = not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1
    if not alone and counts.is_cuda and dist.get_backend(group) == "nccl":
        flag = ((n_local - counts[1]) > 0).to(torch.int32).reshape(1)
        dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=group)
        hit, picked, go_on = torch.cat([counts, flag]).tolist()
        return int(hit), int(picked), bool(go_on)
    hit, picked = (int(v) for v in counts.tolist())
    return hit, picked, any_rank_unresolved(n_local - picked, device, group)


def hinted_prompts(prompt_ids: torch.Tensor, gt_ids: torch.Tensor, depths: torch.Tensor, width: int,
                   pad_token_id: int) -> torch.Tensor:
    """[g, width] rows ``pad * (width - T - d) + prompt + gt[:d]`` (left-padded like a batch tokenizer does)."""
    g, T = prompt_ids.shape
    dev = prompt_ids.device
    extra = width - T
    cols = torch.arange(width, device=dev)[None, :]
    shift = (extra - depths.to(torch.int64))[:, None]          # leading pad columns
    src = cols - shift                                         # index into prompt ++ gt
    both = torch.cat([prompt_ids, gt_ids[:, :extra]], dim=1) if extra > 0 else prompt_ids
    rows = torch.gather(both, 1, src.clamp(min=0, max=both.shape[1] - 1))
    return torch.where(src >= 0, rows, torch.full_like(rows, pad_token_id))


def run_dynamic_hint_cascade(
    generate_stage: Callable[[torch.Tensor, torch.Tensor, torch.Tensor], torch.Tensor],
    prompt_ids: torch.Tensor,
    gt_ids: torch.Tensor,
    num_generations: int,
    max_hint_depth: int,
    scorer,
    pad_token_id: int,
    depth_cap: Optional[torch.Tensor] = None,
    group=None,
) -> CascadeResult:
    """``prompt_ids`` [G, T] (one row per prompt group, ending in the assistant prefix) and
    ``gt_ids`` [G, L] (the ground-truth item's SID tokens) live on the GPU.

    ``generate_stage(stage_prompts [g, T + max_hint_depth], depths [g] int32, group_index [g])``
    returns the completions ``[g * num_generations, Lc]`` (token ids, same Lc every stage; shorter
    completions padded with non-SID tokens) -- e.g. ``constrained_beam_search`` with the fixed-hint
    processor.  ``depth_cap`` [G] is each group's largest allowed depth (the group maximum of
    ``dynamic_hint_max_depth_override`` clamped to [0, max_hint_depth]; default: max_hint_depth).
    ``scorer`` is a ``SidRewardScorer``; rule hits compare hint + completion codes with the
    ground truth's (fixed_hint_grpo_trainer.py:33-44).
    """
    dev = prompt_ids.device
    G, T = prompt_ids.shape
    K = int(num_generations)
    width = T + int(max_hint_depth)
    cap = (torch.full((G,), int(max_hint_depth), dtype=torch.int32, device=dev) if depth_cap is None
           else depth_cap.to(device=dev, dtype=torch.int32).clamp(0, int(max_hint_depth)))
    unresolved = torch.ones(G, dtype=torch.bool, device=dev)
    order = torch.arange(G, device=dev)  # unresolved groups first, in index order
    n_local = G                          # how many of them: the one thing the host tracks
    res: Optional[CascadeResult] = None
    stats = []
    for depth in range(int(max_hint_depth) + 1):
        dummy = n_local == 0
        # a rank that is done keeps generating with the others on one dummy group (594-596)
        idx = torch.zeros(1, dtype=torch.int64, device=dev) if dummy else order[:n_local]
        d = torch.minimum(torch.full_like(cap[idx], depth), cap[idx])
        stage_prompts = hinted_prompts(prompt_ids[idx], gt_ids[idx], d, width, pad_token_id)
        comp = generate_stage(stage_prompts, d, idx)
        if comp.shape[0] != idx.numel() * K:
            raise ValueError(f"generate_stage returned {comp.shape[0]} rows for {idx.numel()} groups of {K}")
        if res is None:
            res = CascadeResult(
                completion_ids=torch.full((G * K, comp.shape[1]), pad_token_id, dtype=comp.dtype, device=dev),
                prompt_ids=torch.full((G, width), pad_token_id, dtype=prompt_ids.dtype, device=dev),
                hint_depth=torch.full((G,), -1, dtype=torch.int32, device=dev),
                prompt_hint_depth=torch.full((G,), -1, dtype=torch.int32, device=dev),
                rule_hit_any=torch.zeros(G, dtype=torch.bool, device=dev),
                rule_rewards=torch.zeros(G * K, dtype=torch.float32, device=dev),
            )
        counts = torch.zeros(2, dtype=torch.int32, device=dev)  # groups with a rule hit, groups selected
        if not dummy:
            # hint tokens beyond each group's depth are replaced by padding, which carries no SID code
            hint = gt_ids[idx][:, : max(int(max_hint_depth), 1)]
            hint = torch.where(torch.arange(hint.shape[1], device=dev)[None, :] < d[:, None].to(torch.int64), hint,
                               torch.full_like(hint, pad_token_id))
            rule = scorer(comp, gt_ids[idx].repeat_interleave(K, dim=0), hint.repeat_interleave(K, dim=0),
                          want=("rule",))["rule"]
            hit_any = rule.view(-1, K).max(dim=1).values > 0
            take = hit_any | (cap[idx] == depth)
            # masked copies over the whole stage batch: nothing here needs to know how many groups were taken
            rows = (idx[:, None] * K + torch.arange(K, device=dev)[None, :]).flatten()
            take_rows = take.repeat_interleave(K)
            res.completion_ids[rows] = torch.where(take_rows[:, None], comp, res.completion_ids[rows])
            res.rule_rewards[rows] = torch.where(take_rows, rule.to(torch.float32), res.rule_rewards[rows])
            res.prompt_ids[idx] = torch.where(take[:, None], stage_prompts, res.prompt_ids[idx])
            res.hint_depth[idx] = torch.where(take, torch.full_like(d, depth), res.hint_depth[idx])
            res.prompt_hint_depth[idx] = torch.where(take, d, res.prompt_hint_depth[idx])
            res.rule_hit_any[idx] = torch.where(take, hit_any, res.rule_hit_any[idx])
            unresolved[idx] = ~take
            order = torch.argsort((~unresolved).to(torch.int8), stable=True)
            counts = torch.stack([hit_any.sum(), take.sum()]).to(torch.int32)
        hit_groups, picked, go_on = _stage_sync(counts, n_local, dev, group)
        n_left = n_local - picked
        stats.append({"requested_hint_depth": depth, "evaluated_group_count": n_local,
                      "rule_hit_group_count": hit_groups, "selected_group_count": picked,
                      "remaining_group_count": n_left})
        n_local = n_left
        if not go_on:
            break
    if bool((res.hint_depth < 0).any()):
        raise RuntimeError("Dynamic hint cascade left prompt groups without a selected stage.")
    res.stage_stats = stats
    return res
