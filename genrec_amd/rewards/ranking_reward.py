"""Rule / prefix / NDCG rewards scored on the device.

Mirror of the reference's ``rewards/ranking_reward.py`` (rule_reward 42-52,
get_prefix_rule_reward 55-79, get_ndcg_rule_reward 82-110, build_reward_setup
126-154): same function names (TRL matches reward functions by ``__name__``),
same ``(prompts, completions, completion_ids, **reward_kwargs) -> list[float]``
signature, same values.  The reference compares the integers that
``re.findall(r"_(\\d+)")`` finds in the text (util.py:133-134); here those
integers become int32 code rows and one kernel scores all rows of the local
batch (``grx_sid_rewards``).

``SidRewardScorer`` is the tensor-level interface for callers that already
hold token ids on the device: it maps token ids to codes through a lookup
table and never touches strings.
"""

from __future__ import annotations

import ctypes as C
import math
import re
from typing import Optional, Sequence

import numpy as np
import torch

from .. import _lib

_NUM_RE = re.compile(r"_(\d+)")


def _extract_number(text: str) -> list[int]:
    return [int(m) for m in _NUM_RE.findall(text)]


def _extract_reward_model_from_kwargs(reward_kwargs):
    reward_model = reward_kwargs.get("reward_model")
    if reward_model is not None:
        return reward_model
    # older callers pass (data_source, ability, reward_model, extra_info, trainer_state) positionally by key order
    values = list(reward_kwargs.values())
    if len(values) != 5:
        raise ValueError("reward_model missing from reward kwargs")
    return values[2]


def _extract_completion_text(completion) -> str:
    if isinstance(completion, list) and completion and isinstance(completion[0], dict):
        return completion[0].get("content", "")
    if isinstance(completion, dict):
        return completion.get("content", "")
    return str(completion)


def _full_texts(completions, reward_kwargs) -> list[str]:
    hints = reward_kwargs.get("oracle_hint_text")
    texts = [_extract_completion_text(c) for c in completions]
    if hints is not None:
        texts = [f"{hints[i]}{t}" for i, t in enumerate(texts)]
    return texts


def ndcg_rank_weights(num_beams: int) -> list[float]:
    """``-1/log2(i+2)`` normalised to sum -1, in float64 (reference: ranking_reward.py:83-84)."""
    raw = [-1.0 / math.log2(i + 2) for i in range(num_beams)]
    total = sum(raw)
    return [-w / total for w in raw]


def _pack_codes(rows: Sequence[Sequence[int]], ids: dict) -> tuple[np.ndarray, np.ndarray]:
    """int32 rows for the kernel.  The reference compares Python integers of any size
    (``re.findall(r"_(\\d+)")`` on free text can yield more digits than int32 holds), and only
    equality matters: every distinct value is replaced by a small id from ``ids``, which the
    prediction and ground-truth sides share."""
    lens = np.fromiter((len(r) for r in rows), dtype=np.int32, count=len(rows))
    width = max(int(lens.max()) if len(rows) else 0, 1)
    codes = np.zeros((len(rows), width), dtype=np.int32)
    for i, r in enumerate(rows):
        if r:
            codes[i, : len(r)] = [ids.setdefault(v, len(ids)) for v in r]
    return codes, lens


def _default_device() -> torch.device:
    if not torch.cuda.is_available():
        raise _lib.GenrecHipError("reward scoring runs on the GPU; no HIP device is visible and there is no CPU path")
    return torch.device("cuda", torch.cuda.current_device())


def _reward_columns(R: int, want, dev, num_beams: int, out, ndcg_weights):
    n = len(want)
    if out is None:
        out = torch.empty((R, n), dtype=torch.float32, device=dev)
    assert out.shape == (R, n) and out.dtype == torch.float32 and out.is_contiguous()
    cols = {k: out[:, i] for i, k in enumerate(want)}
    if "ndcg" in cols:
        if R % num_beams:
            cols["ndcg"].fill_(float("nan"))  # rows of an incomplete trailing group get no value
        if ndcg_weights is None:
            ndcg_weights = torch.tensor(ndcg_rank_weights(num_beams), dtype=torch.float64).to(torch.float32).to(dev)
    return cols, n, ndcg_weights


def score_code_rows(pred: torch.Tensor, pred_len: Optional[torch.Tensor], gt: torch.Tensor,
                    gt_len: Optional[torch.Tensor], num_beams: int, prefix_normalize: bool = True,
                    want=("rule", "prefix", "ndcg"), ndcg_weights: Optional[torch.Tensor] = None,
                    out: Optional[torch.Tensor] = None) -> dict[str, torch.Tensor]:
    """Launch ``grx_sid_rewards`` on contiguous int32 code rows that live on the device.

    ``out`` (fp32 [R, len(want)]) receives the rewards as columns in ``want``
    order and is allocated when absent; the returned dict holds its column views.
    """
    for t in (pred, gt):
        _lib.require_cuda(t, "reward inputs")
        assert t.dtype == torch.int32 and t.is_contiguous() and t.dim() == 2
    R, dev = pred.shape[0], pred.device
    cols, n, ndcg_weights = _reward_columns(R, want, dev, num_beams, out, ndcg_weights)

    def ptr(name):
        return cols[name].data_ptr() if name in cols else None

    index = dev.index if dev.index is not None else torch.cuda.current_device()
    with _lib.on_device(index):
        _lib.check(
            _lib.lib().grx_sid_rewards(
                pred.data_ptr(), pred_len.data_ptr() if pred_len is not None else None, pred.shape[1],
                gt.data_ptr(), gt_len.data_ptr() if gt_len is not None else None, gt.shape[1],
                R, int(num_beams), 1 if prefix_normalize else 0,
                ndcg_weights.data_ptr() if ndcg_weights is not None else None,
                ptr("rule"), ptr("prefix"), ptr("ndcg"), n, _lib.current_stream_ptr(dev),
            )
        )
    return cols


def score_token_rows(pred_tok: torch.Tensor, gt_tok: torch.Tensor, code_table: torch.Tensor, num_beams: int,
                     prefix_normalize: bool = True, want=("rule", "prefix", "ndcg"),
                     ndcg_weights: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None
                     ) -> dict[str, torch.Tensor]:
    """Launch ``grx_sid_rewards_tokens`` on int64 token rows (any row stride) on the device."""
    for t in (pred_tok, gt_tok):
        _lib.require_cuda(t, "reward inputs")
        assert t.dtype == torch.int64 and t.dim() == 2 and (t.shape[1] == 0 or t.stride(1) == 1)
    assert code_table.dtype == torch.int32 and code_table.is_contiguous()
    R, dev = pred_tok.shape[0], pred_tok.device
    cols, n, ndcg_weights = _reward_columns(R, want, dev, num_beams, out, ndcg_weights)

    def ptr(name):
        return cols[name].data_ptr() if name in cols else None

    def stride(t):
        return t.stride(0) if t.shape[0] > 1 else max(t.shape[1], 1)

    index = dev.index if dev.index is not None else torch.cuda.current_device()
    with _lib.on_device(index):
        _lib.check(
            _lib.lib().grx_sid_rewards_tokens(
                pred_tok.data_ptr(), stride(pred_tok), pred_tok.shape[1], gt_tok.data_ptr(), stride(gt_tok),
                gt_tok.shape[1], code_table.data_ptr(), code_table.shape[0], R, int(num_beams),
                1 if prefix_normalize else 0, ndcg_weights.data_ptr() if ndcg_weights is not None else None,
                ptr("rule"), ptr("prefix"), ptr("ndcg"), n, _lib.current_stream_ptr(dev),
            )
        )
    return cols


def _score_texts(completions, reward_kwargs, num_beams: int, prefix_normalize: bool, want) -> dict[str, list[float]]:
    reward_model = _extract_reward_model_from_kwargs(reward_kwargs)
    texts = _full_texts(completions, reward_kwargs)
    ids: dict = {}
    pred, pred_len = _pack_codes([_extract_number(t) for t in texts], ids)
    gt, gt_len = _pack_codes([_extract_number(reward_model[i]["ground_truth"]) for i in range(len(texts))], ids)
    dev = _default_device()
    res = score_code_rows(
        torch.from_numpy(pred).to(dev), torch.from_numpy(pred_len).to(dev), torch.from_numpy(gt).to(dev),
        torch.from_numpy(gt_len).to(dev), num_beams, prefix_normalize, want,
    )
    return {k: v.cpu().tolist() for k, v in res.items()}


def rule_reward(prompts, completions, completion_ids, **reward_kwargs):
    """1.0 where the completion's SID codes equal the ground truth's, else 0.0."""
    return _score_texts(completions, reward_kwargs, 1, True, ("rule",))["rule"]


def get_prefix_rule_reward(normalize: bool = True):
    """Matched-prefix length before the first mismatch, divided by ``len(gt)`` when ``normalize``."""

    def prefix_rule_reward(prompts, completions, completion_ids, **reward_kwargs):
        return _score_texts(completions, reward_kwargs, 1, normalize, ("prefix",))["prefix"]

    return prefix_rule_reward


def get_ndcg_rule_reward(num_beams):
    """Rank penalty ``w[i % num_beams]`` for non-exact rows of groups that contain an exact row."""

    def ndcg_rule_reward(prompts, completions, completion_ids, **reward_kwargs):
        vals = _score_texts(completions, reward_kwargs, num_beams, True, ("ndcg",))["ndcg"]
        return vals[: (len(vals) // num_beams) * num_beams]  # the reference emits whole groups only

    return ndcg_rule_reward


def _append_rule_reward_probe(reward_funcs, reward_weights):
    if any(getattr(f, "__name__", "") == "rule_reward" for f in reward_funcs):
        return reward_funcs, reward_weights
    reward_funcs = list(reward_funcs) + [rule_reward]
    reward_weights = [1.0] * (len(reward_funcs) - 1) if reward_weights is None else list(reward_weights)
    reward_weights.append(0.0)
    return reward_funcs, reward_weights


def build_reward_setup(reward_mode: str, num_beams: int, prefix_reward_normalize: bool = True,
                       probe_rule_with_zero_weight: bool = False):
    """(reward_funcs, reward_weights) per mode (reference: ranking_reward.py:126-154)."""
    mode = reward_mode.strip().lower()
    prefix_func = get_prefix_rule_reward(normalize=prefix_reward_normalize)
    ndcg_func = get_ndcg_rule_reward(num_beams)
    if mode == "rule_only":
        return [rule_reward], None
    if mode == "ranking_only":
        return _append_rule_reward_probe([ndcg_func], None)
    if mode == "prefix_rule_only":
        return _append_rule_reward_probe([prefix_func], None)
    if mode == "prefix_only":
        return _append_rule_reward_probe([prefix_func, ndcg_func], None)
    if mode == "prefix_ranking":
        return [prefix_func, rule_reward, ndcg_func], None
    if mode == "ranking":
        return [rule_reward, ndcg_func], None
    raise ValueError(
        f"Unsupported reward_mode={reward_mode}. "
        "Use one of: ranking, rule_only, ranking_only, prefix_rule_only, prefix_only, prefix_ranking."
    )


def build_reward_funcs(reward_mode: str, num_beams: int, prefix_reward_normalize: bool = True):
    return build_reward_setup(reward_mode, num_beams, prefix_reward_normalize, False)[0]


class SidRewardScorer:
    """Rewards straight from token ids on the device.

    ``code_table[token_id]`` is the integer after ``_`` in the token's text
    (-1 for tokens that carry no SID code, e.g. EOS / padding).  Tokens with
    code -1 are dropped, like text the regular expression does not match.
    """

    def __init__(self, code_table, num_beams: int, device=None, prefix_normalize: bool = True):
        self.device = torch.device(device) if device is not None else _default_device()
        self.table = torch.as_tensor(np.asarray(code_table), dtype=torch.int32).to(self.device)
        self.num_beams = int(num_beams)
        self.prefix_normalize = prefix_normalize
        self.ndcg_weights = torch.tensor(ndcg_rank_weights(self.num_beams), dtype=torch.float64).to(torch.float32).to(
            self.device
        )

    def __call__(self, completion_ids: torch.Tensor, gt_ids: torch.Tensor, hint_ids: Optional[torch.Tensor] = None,
                 want=("rule", "prefix", "ndcg"), out: Optional[torch.Tensor] = None) -> dict[str, torch.Tensor]:
        """Rewards of [R, L] completion token ids against [R, Lg] ground-truth token ids.

        ``hint_ids`` [R, Lh] (oracle hint tokens, padded with non-SID tokens) are
        prepended to the completion, as the reference prepends ``oracle_hint_text``
        (ranking_reward.py:25-30).  ``out`` [R, len(want)] fp32 may be supplied to
        receive the columns (e.g. the trainer's ``rewards_per_func`` block).
        """
        full = completion_ids if hint_ids is None else torch.cat([hint_ids.to(completion_ids.device), completion_ids], 1)

        def rows(t):
            t = t.to(device=self.device, dtype=torch.int64)
            return t if t.shape[1] == 0 or t.stride(1) == 1 else t.contiguous()

        return score_token_rows(rows(full), rows(gt_ids), self.table, self.num_beams, self.prefix_normalize, want,
                                self.ndcg_weights, out=out)

    _KIND = {"rule": 0, "prefix": 1, "ndcg": 2}

    def score_with_advantages(self, completion_ids: torch.Tensor, gt_ids: torch.Tensor, reward_weights: torch.Tensor,
                              scale_rewards: str = "group", hint_ids: Optional[torch.Tensor] = None,
                              want=("rule", "ndcg"), out: Optional[torch.Tensor] = None, eps: float = 1e-4):
        """One launch for a rollout shard's scoring: the reward columns of ``__call__`` (into ``out``
        [R, len(want)]) plus ``(rewards, advantages, group_mean, std)`` of the rows, equal bit for bit to
        ``group_advantages(out, reward_weights, num_beams, scale_rewards)`` -- for ``scale_rewards`` "group"
        or "none", which need no other rank's rows.  Returns ``(out, (rewards, advantages, mean, std))``."""
        if scale_rewards not in ("group", "none"):
            raise ValueError("score_with_advantages serves scale_rewards 'group' and 'none'; 'batch' needs the "
                             "rewards of every rank (rollout_advantages)")
        full = completion_ids if hint_ids is None else torch.cat([hint_ids.to(completion_ids.device), completion_ids], 1)
        full = full.to(device=self.device, dtype=torch.int64)
        gt = gt_ids.to(device=self.device, dtype=torch.int64)
        full = full if full.shape[1] == 0 or full.stride(1) == 1 else full.contiguous()
        gt = gt if gt.shape[1] == 0 or gt.stride(1) == 1 else gt.contiguous()
        R, n = full.shape[0], len(want)
        if out is None:
            out = torch.empty((R, n), dtype=torch.float32, device=self.device)
        assert out.shape == (R, n) and out.dtype == torch.float32 and out.is_contiguous()
        w = reward_weights.to(device=self.device, dtype=torch.float32).contiguous()
        assert w.numel() == n
        kinds = (C.c_int32 * n)(*[self._KIND[k] for k in want])
        stats = [torch.empty(R, dtype=torch.float32, device=self.device) for _ in range(4)]

        def stride(t):
            return t.stride(0) if t.shape[0] > 1 else max(t.shape[1], 1)

        index = self.device.index if self.device.index is not None else torch.cuda.current_device()
        with _lib.on_device(index):
            _lib.check(
                _lib.lib().grx_rollout_score_tokens(
                    full.data_ptr(), stride(full), full.shape[1], gt.data_ptr(), stride(gt), gt.shape[1],
                    self.table.data_ptr(), self.table.shape[0], R, self.num_beams, 1 if self.prefix_normalize else 0,
                    self.ndcg_weights.data_ptr(), out.data_ptr(), n, kinds, w.data_ptr(),
                    1 if scale_rewards == "group" else 0, float(eps), stats[0].data_ptr(), stats[1].data_ptr(),
                    stats[2].data_ptr(), stats[3].data_ptr(), _lib.current_stream_ptr(self.device),
                )
            )
        return out, tuple(stats)
