"""Group-relative advantages of a GRPO rollout shard.

Mirror of the reward aggregation the reference trainers run after scoring
(fixed_hint_grpo_trainer.py:382-408; TokenPrefixGRPOTrainer._group_normalize,
_group_normalize_tokenwise and _build_ndcg_rank_penalties,
token_prefix_grpo_trainer.py:41-63,151-161): gather the per-function rewards of
all ranks, weight and nansum them, normalise inside each group of
``num_generations`` consecutive rows, and keep the local slice.

Prompt groups never straddle ranks, so decoding needs no communication; the one
exchange is this all-gather of a few KB per rank, issued through
``torch.distributed`` (backend "nccl" is RCCL on ROCm: one hop over xGMI).
"""

from __future__ import annotations

import ctypes as C
import math
from typing import Optional

import torch
import torch.distributed as dist

from . import _lib

_SCALE = {"none": 0, "group": 1, "batch": 2}


def gather_rows(t: torch.Tensor) -> torch.Tensor:
    """All-gather along dim 0 (``accelerate.utils.gather``); identity without a process group."""
    if _native_comm is not None and t.is_cuda:
        return _native_comm.allgather(t)
    if not (dist.is_available() and dist.is_initialized()):
        return t
    if dist.get_world_size() == 1 and not _single_rank_collectives():
        return t
    t = t.contiguous()
    shape = (dist.get_world_size() * t.shape[0],) + tuple(t.shape[1:])
    if t.is_cuda and dist.get_backend() == "gloo":
        # gloo gathers host tensors only (debugging / rehearsal set-ups); RCCL takes the device tensor as is
        out = torch.empty(shape, dtype=t.dtype)
        dist.all_gather_into_tensor(out, t.cpu())
        return out.to(t.device)
    out = torch.empty(shape, dtype=t.dtype, device=t.device)
    dist.all_gather_into_tensor(out, t)
    return out


def _dev_index(t: torch.Tensor) -> int:
    return t.device.index if t.device.index is not None else torch.cuda.current_device()


def group_advantages(rewards_per_func: torch.Tensor, reward_weights: torch.Tensor, num_generations: int,
                     scale_rewards: str = "group", eps: float = 1e-4):
    """(rewards, advantages, group_mean, std) for [R, n_funcs] fp32 rewards on the device
    (reference: fixed_hint_grpo_trainer.py:383-401)."""
    if scale_rewards not in _SCALE:
        raise ValueError(
            f"Invalid value for scale_rewards: {scale_rewards}. Must be one of 'batch', 'group', or 'none'."
        )
    _lib.require_cuda(rewards_per_func, "rewards_per_func")
    rpf = rewards_per_func.to(torch.float32).contiguous()
    w = reward_weights.to(device=rpf.device, dtype=torch.float32).contiguous()
    R, n_funcs = rpf.shape
    outs = [torch.empty(R, dtype=torch.float32, device=rpf.device) for _ in range(4)]
    with _lib.on_device(_dev_index(rpf)):
        _lib.check(
            _lib.lib().grx_group_advantage(rpf.data_ptr(), w.data_ptr(), R, n_funcs, int(num_generations),
                                           _SCALE[scale_rewards], float(eps), outs[0].data_ptr(), outs[1].data_ptr(),
                                           outs[2].data_ptr(), outs[3].data_ptr(), _lib.current_stream_ptr(rpf.device))
        )
    return tuple(outs)


def group_normalize(values: torch.Tensor, group_size: int, eps: float = 1e-4) -> torch.Tensor:
    """``(v - mean_g) / (std_g + eps)`` with the unbiased group std
    (reference: token_prefix_grpo_trainer.py:41-45)."""
    ones = torch.ones(1, dtype=torch.float32, device=values.device)
    return group_advantages(values.reshape(-1, 1), ones, group_size, "group", eps)[1]


def group_normalize_tokenwise(values: torch.Tensor, mask: torch.Tensor, group_size: int, eps: float = 1e-4
                              ) -> torch.Tensor:
    """Mask-aware per-position group normalisation (reference: token_prefix_grpo_trainer.py:47-63)."""
    _lib.require_cuda(values, "values")
    v = values.to(torch.float32).contiguous()
    m = mask.to(device=v.device, dtype=torch.float32).contiguous()
    R, Tc = v.shape
    out = torch.empty_like(v)
    with _lib.on_device(_dev_index(v)):
        _lib.check(
            _lib.lib().grx_group_normalize_tokenwise(v.data_ptr(), m.data_ptr(), R, Tc, int(group_size), float(eps),
                                                     out.data_ptr(), _lib.current_stream_ptr(v.device))
        )
    return out


def ndcg_rank_penalties(batch_size: int, num_generations: int, device, process_index: int = 0) -> torch.Tensor:
    """Per-row rank penalties of the token-level trainer (reference: token_prefix_grpo_trainer.py:151-161)."""
    w = torch.tensor([-1.0 / math.log2(i + 2) for i in range(num_generations)], dtype=torch.float32, device=device)
    w = -w / w.sum()
    idx = (torch.arange(batch_size, device=device) + process_index * batch_size) % num_generations
    return w[idx]


class RolloutAdvantages:
    """Result of :func:`rollout_advantages`.

    ``local_advantages`` ([R_local] fp32) is what the trainer's loss consumes
    (``advantages[process_slice]``, fixed_hint_grpo_trainer.py:404-408) and is ready in stream order.
    The statistics over the rows of ALL ranks that the reference logs (``rewards``, ``advantages``,
    ``group_mean``, ``std``: fixed_hint_grpo_trainer.py:383-401, 410-430) are produced on demand:
    the all-gather behind them is started when the object is built, and only ``wait_global()``
    (or unpacking the object like the 5-tuple it used to be) waits for it.
    """

    def __init__(self, local, local_stats, pending, weights, num_generations, scale_rewards, eps):
        self.local_advantages = local
        self.local_rewards, self.local_group_mean, self.local_std = local_stats
        self._pending = pending  # (work handle or None, gathered tensor, host staging or None)
        self._args = (weights, num_generations, scale_rewards, eps)
        self._global = None

    def gathered_rewards_per_func(self) -> torch.Tensor:
        """[R_global, n_funcs] rewards of all ranks (waits for the all-gather)."""
        work, out, device = self._pending
        if work is not None:
            work.wait()
            self._pending = (None, out, device)
        if device is not None and out.device != device:  # gloo gathered on the host
            out = out.to(device)
            self._pending = (None, out, None)
        return out

    def wait_global(self):
        """``(rewards, advantages, group_mean, std)`` over the rows of all ranks."""
        if self._global is None:
            w, K, scale, eps = self._args
            self._global = group_advantages(self.gathered_rewards_per_func(), w, K, scale, eps)
        return self._global

    def logged_statistics(self, reward_func_names) -> dict:
        """The scalars the reference logs after a rollout (fixed_hint_grpo_trainer.py:410-417):
        ``rewards/<name>/mean`` and ``/std`` per reward function (nan-aware, unbiased -- TRL's ``nanstd``),
        ``reward``, ``reward_std`` and ``frac_reward_zero_std`` over the rows of all ranks, as Python
        floats -- computed on the device and fetched with ONE read-back where the reference issues
        2 x n_funcs + 3 ``.item()`` calls.  ``nanstd`` lives in TRL (``trl.trainer.utils``), which is not
        installed where this was written: its published formula (nan-aware mean of squared deviations
        times count / (count - 1), square root) is restated here -- parity unpinned on that one function."""
        rpf = self.gathered_rewards_per_func().to(torch.float32)
        _, _, mean, std = self.wait_global()
        col_mean = torch.nanmean(rpf, dim=0)
        present = ~torch.isnan(rpf)
        count = present.sum(dim=0).to(torch.float32)
        var = torch.nanmean((rpf - col_mean[None, :]) ** 2, dim=0) * (count / (count - 1))
        zero_std = torch.isclose(std, torch.zeros_like(std)).float().mean()
        packed = torch.cat([col_mean, torch.sqrt(var), torch.stack([mean.mean(), std.mean(), zero_std])]).tolist()
        n = rpf.shape[1]
        names = list(reward_func_names)
        if len(names) != n:
            raise ValueError(f"{len(names)} reward function names for {n} reward columns")
        out = {}
        for i, name in enumerate(names):
            out[f"rewards/{name}/mean"] = packed[i]
            out[f"rewards/{name}/std"] = packed[n + i]
        out["reward"], out["reward_std"], out["frac_reward_zero_std"] = packed[2 * n], packed[2 * n + 1], packed[2 * n + 2]
        return out

    def __iter__(self):  # (local_advantages, all_advantages, rewards, group_mean, std)
        rewards, adv, mean, std = self.wait_global()
        return iter((self.local_advantages, adv, rewards, mean, std))

    def __getitem__(self, i):
        if i == 0:
            return self.local_advantages
        return tuple(self)[i]


_native_comm = None


def use_native_comm(comm) -> None:
    """Route the exchange step's all-gather of device tensors through ``comm`` (a
    ``genrec_amd.comm.RcclComm``): RCCL on the compute stream instead of ``torch.distributed``.
    ``None`` switches back."""
    global _native_comm
    _native_comm = comm


def _single_rank_collectives() -> bool:
    """GENREC_AMD_SINGLE_RANK_COLLECTIVES=1: issue the collective even in a process group of one rank
    (lets a one-GPU box run the RCCL code path itself; off by default, where one rank gathers nothing)."""
    import os

    return os.environ.get("GENREC_AMD_SINGLE_RANK_COLLECTIVES", "0") not in ("", "0")


def _start_gather(t: torch.Tensor):
    """Begin ``gather_rows(t)`` without waiting: (work, output tensor, device to move it to or None)."""
    if _native_comm is not None and t.is_cuda:
        return None, _native_comm.allgather(t), None  # stream-ordered: nothing to wait for
    if not (dist.is_available() and dist.is_initialized()):
        return None, t, None
    if dist.get_world_size() == 1 and not _single_rank_collectives():
        return None, t, None
    t = t.contiguous()
    shape = (dist.get_world_size() * t.shape[0],) + tuple(t.shape[1:])
    if t.is_cuda and dist.get_backend() == "gloo":
        out = torch.empty(shape, dtype=t.dtype)
        return dist.all_gather_into_tensor(out, t.cpu(), async_op=True), out, t.device
    out = torch.empty(shape, dtype=t.dtype, device=t.device)
    return dist.all_gather_into_tensor(out, t, async_op=True), out, None


def rollout_advantages(local_rewards_per_func: torch.Tensor, reward_weights: torch.Tensor, num_generations: int,
                       scale_rewards: str = "group", process_index: Optional[int] = None, eps: float = 1e-4
                       ) -> RolloutAdvantages:
    """The exchange step of a rollout shard (reference: fixed_hint_grpo_trainer.py:382-408): the
    ``[R_local, n_funcs]`` rewards of all ranks are all-gathered, rewards are weighted and summed,
    advantages normalised per group of ``num_generations`` rows, and the local rows kept.

    Whole groups live on one rank, so for ``scale_rewards`` "group" and "none" a rank's advantages
    follow from its own rows alone: they are computed at once from the local rewards -- the same
    values the reference slices out of the gathered result -- while the all-gather (needed for the
    logged statistics only) runs beside them.  "batch" divides by the std over the rows of all
    ranks, so there the gather comes first.  Returns a :class:`RolloutAdvantages`.
    """
    if scale_rewards not in _SCALE:
        raise ValueError(
            f"Invalid value for scale_rewards: {scale_rewards}. Must be one of 'batch', 'group', or 'none'."
        )
    pending = _start_gather(local_rewards_per_func)
    n_local = local_rewards_per_func.shape[0]
    # a collective ran (or is running) whenever what comes back is not the local tensor itself -- with
    # torch.distributed there is a work handle to wait for, with the library's communicator there is none
    gathered = pending[1] is not local_rewards_per_func
    if scale_rewards == "batch" and gathered:
        res = RolloutAdvantages(None, (None, None, None), pending, reward_weights, num_generations, scale_rewards, eps)
        rewards, adv, mean, std = res.wait_global()
        if process_index is None:
            process_index = _native_comm.rank if _native_comm is not None else dist.get_rank()
        sl = slice(process_index * n_local, (process_index + 1) * n_local)
        res.local_advantages = adv[sl]
        res.local_rewards, res.local_group_mean, res.local_std = rewards[sl], mean[sl], std[sl]
        return res
    rewards, adv, mean, std = group_advantages(local_rewards_per_func, reward_weights, num_generations,
                                               scale_rewards, eps)
    res = RolloutAdvantages(adv, (rewards, mean, std), pending, reward_weights, num_generations, scale_rewards, eps)
    if not gathered:  # one process, no collective: the local statistics are the global ones
        res._global = (rewards, adv, mean, std)
    return res


def rollout_score(scorer, completion_ids: torch.Tensor, gt_ids: torch.Tensor, reward_weights: torch.Tensor,
                  scale_rewards: str = "group", hint_ids: Optional[torch.Tensor] = None, want=("rule", "ndcg"),
                  out: Optional[torch.Tensor] = None, eps: float = 1e-4) -> RolloutAdvantages:
    """Scoring and exchange of a rollout shard: rewards of the local completions (``scorer``: a
    ``SidRewardScorer``), local advantages, and the all-gather of the reward matrix started beside
    them -- one kernel launch where ``scale_rewards`` allows it ("group", "none"), else rewards,
    gather and ``rollout_advantages`` in sequence.  Returns a :class:`RolloutAdvantages`."""
    if scale_rewards == "batch":
        cols = scorer(completion_ids, gt_ids, hint_ids, want=want, out=out)
        rpf = out if out is not None else torch.stack([cols[k] for k in want], dim=1)
        return rollout_advantages(rpf, reward_weights, scorer.num_beams, scale_rewards, eps=eps)
    rpf, (rewards, adv, mean, std) = scorer.score_with_advantages(completion_ids, gt_ids, reward_weights,
                                                                  scale_rewards, hint_ids, want, out, eps)
    pending = _start_gather(rpf)
    res = RolloutAdvantages(adv, (rewards, mean, std), pending, reward_weights, scorer.num_beams, scale_rewards, eps)
    if pending[1] is rpf:  # no collective ran
        res._global = (rewards, adv, mean, std)
    return res
