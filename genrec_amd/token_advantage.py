"""Token-level prefix advantages of a GRPO rollout shard.

Mirror of what ``TokenPrefixGRPOTrainer._generate_and_score_completions`` does after generation
(token_prefix_grpo_trainer.py:196-278) and of its token-space reward builders
(``_build_prefix_token_rewards_with_error_mask``, :107-149; ``_build_prefix_token_rewards``,
:65-105).  The reference walks the rows in Python and reads every compared token back from the
device (``int(comp_valid_ids[matched].item())``, :138); here one kernel launch (two with
``total_token_normalize``) does the match, the rewards, the error-token mask and the advantages
for all rows of the rank.

Prompt groups never straddle ranks, so the ``gather`` calls of the reference (:215, :241-242,
:249, :274) change no value of the local rows; none is issued here.
"""

from __future__ import annotations

import ctypes as C
import math
from dataclasses import dataclass
from typing import Optional, Sequence

import torch

from . import _lib


@dataclass
class TokenPrefixAdvantages:
    token_advantages: torch.Tensor   # [R, Tc] fp32: ``outputs["advantages"]`` of the reference (:278)
    token_rewards: torch.Tensor      # [R, Tc] fp32 prefix token rewards (:116, :143)
    seq_rewards: torch.Tensor        # [R] fp32 prefix sequence rewards (:117, :144)
    error_token_mask: torch.Tensor   # [R, Tc] fp32, 1 on wrong tokens (:118, :147)
    completion_mask: torch.Tensor    # [R, Tc] fp32

    def nonzero_ratio(self) -> torch.Tensor:
        """Share of valid completion tokens with a non-zero advantage (the logged
        ``token_adv_nonzero_ratio``, :281-283), a 0-d device tensor."""
        m = self.completion_mask
        return ((self.token_advantages.abs() > 0).float() * m).sum() / m.sum().clamp_min(1.0)


def pack_ground_truth_ids(ground_truth_ids: Sequence[Sequence[int]], device) -> tuple[torch.Tensor, torch.Tensor]:
    """``(gt_ids [R, Lmax] int64 padded with -1, gt_len [R] int32)`` on ``device`` from per-row token id
    lists -- ``tokenizer(gt_text, add_special_tokens=False).input_ids`` of every row's ground truth
    (token_prefix_grpo_trainer.py:122-123), built once per batch on the host."""
    width = max(1, max((len(g) for g in ground_truth_ids), default=0))
    ids = torch.full((len(ground_truth_ids), width), -1, dtype=torch.int64)
    for i, g in enumerate(ground_truth_ids):
        if len(g):
            ids[i, : len(g)] = torch.as_tensor(list(g), dtype=torch.int64)
    lens = torch.tensor([len(g) for g in ground_truth_ids], dtype=torch.int32)
    return ids.to(device), lens.to(device)


def rank_penalty_weights(num_generations: int) -> torch.Tensor:
    """The K rank weights of ``_build_ndcg_rank_penalties`` (:151-158), computed on the host with the
    reference's own fp32 operations so that the values are bit-identical to its."""
    w = torch.tensor([-1.0 / math.log2(i + 2) for i in range(num_generations)], dtype=torch.float32)
    return -w / w.sum()


_RANK_WEIGHTS: dict = {}


def _rank_weights_on(num_generations: int, device) -> torch.Tensor:
    key = (int(num_generations), str(device))
    w = _RANK_WEIGHTS.get(key)
    if w is None:
        w = _RANK_WEIGHTS[key] = rank_penalty_weights(num_generations).to(device)
    return w


def token_prefix_advantages(
    completion_ids: torch.Tensor,
    completion_mask: torch.Tensor,
    gt_ids: torch.Tensor,
    gt_len: torch.Tensor,
    num_generations: int,
    prefix_weight: float,
    ndcg_weight: float = 0.0,
    ndcg_rewards: Optional[torch.Tensor] = None,
    *,
    has_ndcg: Optional[bool] = None,
    eos_token_id: Optional[int] = None,
    prefix_reward_normalize: bool = True,
    token_adv_total_token_normalize: bool = False,
    token_level_ndcg_error_token_penalty: bool = False,
    process_index: int = 0,
    eps: float = 1e-4,
) -> TokenPrefixAdvantages:
    """``completion_ids`` [R, Tc] int and ``completion_mask`` [R, Tc] (valid where > 0) on the GPU;
    ``gt_ids`` / ``gt_len`` from :func:`pack_ground_truth_ids`; ``ndcg_rewards`` [R]: the output of
    ``ndcg_rule_reward`` for these rows (e.g. ``SidRewardScorer(...)["ndcg"]``), needed whenever the
    reference would call that function (:226-236, :263-273).  ``has_ndcg``: an ``ndcg_rule_reward`` is
    among the trainer's reward functions (default: ``ndcg_rewards is not None`` or the error-token
    penalty is on); the three flags are the trainer's constructor arguments (:21-23)."""
    _lib.require_cuda(completion_ids, "completion_ids")
    dev = completion_ids.device
    if completion_ids.dim() != 2 or completion_mask.shape != completion_ids.shape:
        raise ValueError("completion_ids and completion_mask must be [rows, completion length] and agree")
    R, Tc = completion_ids.shape
    K = int(num_generations)
    if has_ndcg is None:
        has_ndcg = ndcg_rewards is not None or bool(token_level_ndcg_error_token_penalty)
    total, err_pen = bool(token_adv_total_token_normalize), bool(token_level_ndcg_error_token_penalty)
    needs_ndcg = has_ndcg and not (total and err_pen)
    if needs_ndcg and ndcg_rewards is None:
        raise ValueError("this configuration uses the rows' NDCG rewards: pass ndcg_rewards")
    ids = completion_ids.to(torch.int64)
    if Tc > 0 and ids.stride(1) != 1:
        ids = ids.contiguous()
    mask = completion_mask.to(device=dev, dtype=torch.float32).contiguous()
    gt_ids = gt_ids.to(device=dev, dtype=torch.int64).contiguous()
    gt_len = gt_len.to(device=dev, dtype=torch.int32).contiguous()
    if gt_ids.shape[0] != R or gt_len.shape[0] != R:
        raise ValueError("one ground truth per completion row")
    nd = ndcg_rewards.to(device=dev, dtype=torch.float32).contiguous() if (needs_ndcg and ndcg_rewards is not None) else None
    rank = _rank_weights_on(K, dev) if (has_ndcg and total and err_pen) else None
    out = TokenPrefixAdvantages(
        token_advantages=torch.empty((R, Tc), dtype=torch.float32, device=dev),
        token_rewards=torch.empty((R, Tc), dtype=torch.float32, device=dev),
        seq_rewards=torch.empty((R,), dtype=torch.float32, device=dev),
        error_token_mask=torch.empty((R, Tc), dtype=torch.float32, device=dev),
        completion_mask=mask,
    )
    scratch = torch.empty((R, Tc), dtype=torch.float32, device=dev) if total else None
    a = _lib.GrxTokenPrefixArgs()
    a.completion_ids, a.ids_stride = ids.data_ptr(), ids.stride(0) if R > 1 else max(Tc, 1)
    a.completion_mask = mask.data_ptr()
    a.gt_ids, a.gt_stride, a.gt_len = gt_ids.data_ptr(), gt_ids.stride(0) if R > 1 else gt_ids.shape[1], gt_len.data_ptr()
    a.ndcg = nd.data_ptr() if nd is not None else None
    a.rank_weights = rank.data_ptr() if rank is not None else None
    a.R, a.Tc, a.K = R, Tc, K
    a.eos_token_id = -1 if eos_token_id is None else int(eos_token_id)
    a.normalize = 1 if prefix_reward_normalize else 0
    a.total_token_normalize, a.has_ndcg, a.error_token_penalty = int(total), int(bool(has_ndcg)), int(err_pen)
    a.rank_offset = (int(process_index) * R) % K  # only its remainder enters the rank index (:159-160)
    a.prefix_weight, a.ndcg_weight, a.eps = float(prefix_weight), float(ndcg_weight if has_ndcg else 0.0), float(eps)
    a.token_rewards, a.seq_rewards = out.token_rewards.data_ptr(), out.seq_rewards.data_ptr()
    a.error_mask = out.error_token_mask.data_ptr()
    a.scratch = scratch.data_ptr() if scratch is not None else None
    a.token_advantages = out.token_advantages.data_ptr()
    index = dev.index if dev.index is not None else torch.cuda.current_device()
    with _lib.on_device(index):
        _lib.check(_lib.lib().grx_token_prefix_advantages(C.byref(a), _lib.current_stream_ptr(dev)))
    return out
