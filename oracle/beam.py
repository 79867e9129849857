"""Oracle (test infrastructure): the beam-search step arithmetic on the CPU.

The reference does not own this code: it calls ``model.generate(num_beams=K)``
(evaluate.py:363-371, MIMIGenRec.py:79, analyze_rl_beam_hint.py:609-623) and the
arithmetic lives in ``transformers.GenerationMixin._beam_search``.  This module
restates that loop body from transformers 5.15.0, ``generation/utils.py``:
  * initial state                          -- 3316-3351
  * accumulate + flatten                   -- 3418-3420
  * ``_get_top_k_continuations``           -- 3077-3129
  * stopping criteria (EOS / max length)   -- 3438-3444
  * ``_get_running_beams_for_next_iteration`` -- 3131-3151
  * ``_update_finished_beams``             -- 3153-3204
  * ``_check_early_stop_heuristic``        -- 3008-3052
  * loop condition                         -- 3055-3075
  * output cropping                        -- 3512-3523
It is pinned by ``tests/golden/beam_*.npz``: traces captured from the installed
``generate`` driving the real reference processors.

One deliberate difference: ``torch.topk`` leaves the order of equal values
unspecified; here ties are ordered by ascending index (the rule the GPU kernel
implements).  On tie-free inputs both give the same answer.
"""

from __future__ import annotations

from dataclasses import dataclass
from typing import Optional

import torch

NEG = -1.0e9


def topk_canonical(x: torch.Tensor, k: int):
    """top-k along the last dim ordered by (value descending, index ascending)."""
    vals, idx = torch.sort(x, dim=-1, descending=True, stable=True)
    return vals[..., :k], idx[..., :k]


def _gather_beams(t: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
    while idx.dim() < t.dim():
        idx = idx.unsqueeze(-1)
    return torch.take_along_dim(t, idx, dim=1)


@dataclass
class BeamStepTrace:
    topk_scores: torch.Tensor   # [B, n_keep] accumulated log-probs of the kept candidates
    topk_parent: torch.Tensor   # [B, n_keep] beam index inside the group
    topk_token: torch.Tensor    # [B, n_keep]
    topk_hit: torch.Tensor      # [B, n_keep] bool
    next_parent: torch.Tensor   # [B*K] flat cache-reorder index


class RefBeamSearch:
    """State of one batched beam search; ``step(log_probs)`` advances it by one token."""

    def __init__(self, input_ids: torch.Tensor, num_beams: int, max_length: int, eos_token_id: Optional[int],
                 pad_token_id: Optional[int] = None, length_penalty: float = 1.0, early_stopping=False,
                 n_eos_tokens: Optional[int] = None):
        R, cur_len = input_ids.shape
        assert R % num_beams == 0
        self.K = num_beams
        self.B = R // num_beams
        self.max_length = max_length
        self.eos = eos_token_id
        self.length_penalty = length_penalty
        self.early_stopping = early_stopping
        self.prompt_len = cur_len
        self.cur_len = cur_len
        n_eos = (1 if eos_token_id is not None else 0) if n_eos_tokens is None else n_eos_tokens
        self.n_keep = max(2, 1 + n_eos) * num_beams
        fill = (pad_token_id or eos_token_id) if eos_token_id is not None else -1  # utils.py:3319
        B, K = self.B, self.K
        self.running_sequences = torch.full((B, K, max_length), fill, dtype=torch.int64)
        self.running_sequences[:, :, :cur_len] = input_ids.view(B, K, cur_len)
        self.sequences = self.running_sequences.clone()
        self.running_beam_scores = torch.zeros((B, K), dtype=torch.float32)
        self.running_beam_scores[:, 1:] = NEG
        self.beam_scores = torch.full((B, K), NEG, dtype=torch.float32)
        self.is_sent_finished = torch.zeros((B, K), dtype=torch.bool)
        self.unsat = torch.ones((B, 1), dtype=torch.bool)
        self.next_hits = torch.zeros((B, K), dtype=torch.bool)
        self.running_beam_indices = torch.full((B, K, max_length - cur_len), -1, dtype=torch.int32)
        self.beam_indices = self.running_beam_indices.clone()
        self.top_num_beam_mask = torch.cat(
            (torch.ones(K, dtype=torch.bool), torch.zeros(self.n_keep - K, dtype=torch.bool))
        )

    def flat_running_sequences(self) -> torch.Tensor:
        return self.running_sequences[:, :, : self.cur_len].reshape(self.B * self.K, self.cur_len)

    def step(self, log_probs: torch.Tensor) -> BeamStepTrace:
        """``log_probs``: [B*K, V] fp32 output of the logits processor."""
        B, K, n_keep = self.B, self.K, self.n_keep
        V = log_probs.shape[-1]
        cur_len, prompt_len = self.cur_len, self.prompt_len
        acc = log_probs.view(B, K, V) + self.running_beam_scores[:, :, None]
        acc = acc.reshape(B, K * V)

        topk_log_probs, topk_indices = topk_canonical(acc, n_keep)
        parent = topk_indices // V
        token = topk_indices % V
        topk_running_beam_indices = _gather_beams(self.running_beam_indices, parent)
        topk_running_sequences = _gather_beams(self.running_sequences, parent)
        topk_running_sequences[:, :, cur_len] = token
        topk_running_beam_indices[:, :, cur_len - prompt_len] = (
            parent + torch.arange(B).view(-1, 1) * K
        ).to(torch.int32)

        hits = torch.full((B, n_keep), cur_len + 1 >= self.max_length, dtype=torch.bool)
        if self.eos is not None:
            hits = hits | (token == self.eos)

        # running beams for the next iteration
        running_log_probs = topk_log_probs + hits.to(torch.float32) * NEG
        _, next_idx = topk_canonical(running_log_probs, K)
        self.running_sequences = _gather_beams(topk_running_sequences, next_idx)
        self.running_beam_scores = _gather_beams(running_log_probs, next_idx)
        self.running_beam_indices = _gather_beams(topk_running_beam_indices, next_idx)

        # finished beams
        just_finished = hits & self.top_num_beam_mask[None, :]
        fin = topk_log_probs / ((cur_len + 1 - prompt_len) ** self.length_penalty)
        full = torch.all(self.is_sent_finished, dim=-1, keepdim=True) & (self.early_stopping is True)
        fin = fin + full.to(torch.float32) * NEG
        fin = fin + (~self.unsat).to(torch.float32) * NEG
        fin = fin + (~just_finished) * NEG
        merged_scores = torch.cat((self.beam_scores, fin), dim=1)
        merged_seqs = torch.cat((self.sequences, topk_running_sequences), dim=1)
        merged_bidx = torch.cat((self.beam_indices, topk_running_beam_indices), dim=1)
        merged_done = torch.cat((self.is_sent_finished, just_finished), dim=1)
        _, keep = topk_canonical(merged_scores, K)
        self.sequences = _gather_beams(merged_seqs, keep)
        self.beam_scores = _gather_beams(merged_scores, keep)
        self.beam_indices = _gather_beams(merged_bidx, keep)
        self.is_sent_finished = _gather_beams(merged_done, keep)

        next_parent = self.running_beam_indices[..., cur_len - prompt_len].reshape(-1).clone()
        self.next_hits = hits
        self.cur_len = cur_len + 1

        # early-stop heuristic for the next iteration
        if self.early_stopping == "never" and self.length_penalty > 0.0:
            hyp_len = self.max_length - prompt_len
        else:
            hyp_len = self.cur_len - prompt_len
        best_possible = self.running_beam_scores[:, :1] / (hyp_len**self.length_penalty)
        worst_finished = torch.where(
            self.is_sent_finished, torch.min(self.beam_scores, dim=1, keepdim=True)[0], torch.tensor(NEG)
        )
        self.unsat = self.unsat & torch.any(best_possible > worst_finished, dim=-1, keepdim=True)
        return BeamStepTrace(topk_log_probs, parent, token, hits, next_parent)

    def has_unfinished(self) -> bool:
        improvement_possible = bool(torch.any(self.unsat))
        exists_open_beam = not (bool(torch.all(self.is_sent_finished)) and self.early_stopping is True)
        valid_continuations = not bool(torch.all(self.next_hits))
        return improvement_possible and exists_open_beam and valid_continuations

    def finalize(self, num_return_sequences: Optional[int] = None):
        n = self.K if num_return_sequences is None else num_return_sequences
        seqs = self.sequences[:, :n, :].reshape(self.B * n, -1)
        scores = self.beam_scores[:, :n].reshape(-1)
        bidx = self.beam_indices[:, :n, :].reshape(self.B * n, -1)
        gen_len = int(((bidx + 1).bool()).sum(dim=1).max())
        return seqs[:, : self.prompt_len + gen_len], scores, bidx[:, :gen_len]


def ref_constrained_beam_search(logits_fn, processor, input_ids: torch.Tensor, num_beams: int, max_new_tokens: int,
                                eos_token_id: Optional[int], pad_token_id: Optional[int] = None,
                                length_penalty: float = 1.0, early_stopping=False,
                                num_return_sequences: Optional[int] = None, max_steps: Optional[int] = None):
    """Whole loop: ``logits_fn(step, flat_running_sequences) -> [B*K, V]`` raw logits;
    HF normalises them (utils.py:3388) before the processor sees them."""
    prompt = input_ids.repeat_interleave(num_beams, dim=0)
    state = RefBeamSearch(prompt, num_beams, input_ids.shape[1] + max_new_tokens, eos_token_id, pad_token_id,
                          length_penalty, early_stopping)
    traces = []
    step = 0
    while True:
        flat = state.flat_running_sequences()
        logits = logits_fn(step, flat).to(torch.float32)
        log_probs = torch.nn.functional.log_softmax(logits, dim=-1)
        log_probs = processor(flat, log_probs)
        traces.append(state.step(log_probs))
        step += 1
        if not state.has_unfinished() or (max_steps is not None and step >= max_steps):
            break
    return state, traces
