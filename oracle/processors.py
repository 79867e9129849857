"""Oracle (test infrastructure): the reference's trie and logits processors on the CPU.

Restates, in plain Python + PyTorch-CPU:
  * ``Trie``                      -- src/llamafactory/extras/trie.py:18-97
  * ``build_trie_from_index``     -- evaluate.py:48-104
  * ``ConstrainedLogitsProcessor``-- logit_processor.py:8-49
  * seeded variant                -- analyze_rl_beam_hint.py:553-560
  * ``FixedHintConstrainedLogitsProcessor`` and helpers
                                  -- fixed_hint_logit_processor.py:8-96
The arithmetic is kept in the reference's order (log_softmax, a -inf mask
tensor, one index assignment per row, one add) because this module is also the
timed CPU baseline.
"""

from __future__ import annotations

import json
from typing import Callable, Optional

import torch


# ------------------------------------------------------------------ trie ---
class DictTrie:
    """Nested-dict token trie (trie.py:18-97, without the unused ``append`` chaining)."""

    def __init__(self, sequences=None):
        self.trie_dict: dict = {}
        self.len = 0
        for seq in sequences or []:
            self.add(seq)

    def add(self, sequence) -> None:  # trie.py:36-38,50-55
        level = self.trie_dict
        for tok in sequence:
            if tok not in level:
                level[tok] = {}
            level = level[tok]
        self.len += 1

    def get(self, prefix_sequence) -> list[int]:  # trie.py:40-41,57-81
        level = self.trie_dict
        for tok in prefix_sequence:
            if tok not in level:
                return []
            level = level[tok]
        return list(level.keys())

    def __len__(self) -> int:  # trie.py:93-94
        return self.len

    def __iter__(self):  # trie.py:83-91
        stack = [([], self.trie_dict)]
        while stack:
            path, level = stack.pop()
            if not level:
                yield path
                continue
            for tok in reversed(list(level)):
                stack.append((path + [tok], level[tok]))


def ref_build_trie_from_index(index_path: str, tokenizer, prefix: Optional[str] = None, sid_levels: int = -1):
    """evaluate.py:63-95 -> (trie, prompt_suffix_ids, prefix_index)."""
    with open(index_path, encoding="utf-8") as f:
        index_data = json.load(f)
    unique_items = set()
    for raw in index_data.values():
        if isinstance(raw, list):  # evaluate.py:48-54
            toks = [str(t).strip() for t in raw if str(t).strip()]
        elif isinstance(raw, str):
            toks = [raw.strip()] if raw.strip() else []
        else:
            toks = []
        if sid_levels > 0:  # evaluate.py:57-60
            toks = toks[:sid_levels]
        if toks:
            unique_items.add("".join(toks))
    if prefix is None:  # evaluate.py:38-45
        msgs = [{"role": "user", "content": "x"}]
        a = tokenizer.apply_chat_template(msgs, add_generation_prompt=True, tokenize=False)
        b = tokenizer.apply_chat_template(msgs, add_generation_prompt=False, tokenize=False)
        prefix = a[len(b) :]
    prefix_ids = tokenizer(prefix, add_special_tokens=False).input_ids
    seqs = []
    for item in unique_items:
        ids = tokenizer(prefix + item, add_special_tokens=False).input_ids
        seqs.append(prefix_ids + ids[len(prefix_ids) :] + [tokenizer.eos_token_id])
    return DictTrie(seqs), prefix_ids, len(prefix_ids)


# ----------------------------------------------------- allowed-token rules ---
def _as_list(x) -> list[int]:
    return x.tolist() if hasattr(x, "tolist") else list(x)


def count_window_allowed(rows, fn: Callable, prefix_index: int, prefix_ids, count: int):
    """Per-row allowed sets of the count-based processor and the counter after
    the row loop, before the per-call increment (logit_processor.py:37-46).

    ``rows`` is a 2-D tensor or a list of token lists; as in the reference each
    row is sliced first and converted to a Python list afterwards (two
    conversions per row).  Raises AssertionError at the first empty set.
    """
    allowed = []
    for sent in rows:
        if _as_list(sent[-prefix_index:]) == prefix_ids:  # shared counter drops to 0 mid-loop (39-40)
            count = 0
        window = _as_list(sent[-prefix_index - count :])
        toks = fn(window)
        assert len(toks) > 0, "No valid tokens for prefix_ids"
        allowed.append(list(toks))
    return allowed, count


def fixed_hint_allowed_row(sent: list[int], prefix_ids, fn: Callable, eos_token_id) -> list[int]:
    """fixed_hint_logit_processor.py:18-61 for one row."""
    if eos_token_id is not None and sent and sent[-1] == eos_token_id:
        return [eos_token_id]
    start = 0
    if prefix_ids:
        n = len(prefix_ids)
        starts = [s for s in range(len(sent) - n + 1) if sent[s : s + n] == prefix_ids]
        for s in starts:
            if fn(sent[s:]):
                start = s
                break
        else:
            raise ValueError(
                "Failed to find a valid trie prefix match inside current sequence. "
                f"prefix_ids={prefix_ids}, candidate_starts={starts}, sent={sent}"
            )
    toks = fn(sent[start:])
    if len(toks) == 0:
        raise AssertionError("No valid tokens for prefix_ids")
    return list(toks)


def masked_log_softmax(scores: torch.Tensor, allowed: list[list[int]]) -> torch.Tensor:
    """log_softmax + (-inf mask with zeros at the allowed ids), as the reference
    builds it (logit_processor.py:30-31,46,49)."""
    logp = torch.nn.functional.log_softmax(scores, dim=-1)
    mask = torch.full_like(logp, float("-inf"))
    for r, toks in enumerate(allowed):
        mask[r, toks] = 0
    return logp + mask


def ref_repetition_penalty(input_ids: torch.Tensor, scores: torch.Tensor, penalty: float,
                           prompt_ignore_length: int = 0) -> torch.Tensor:
    """transformers' RepetitionPenaltyLogitsProcessor, 2-D path (transformers 5.15.0,
    generation/logits_process.py, ``__call__``): gather the scores of the row's tokens, multiply the
    negative ones by the penalty and divide the others, scatter back.  generate() applies it to
    log_softmax(logits) ahead of the custom processors (generation/utils.py:3388-3389)."""
    if prompt_ignore_length:
        input_ids = input_ids[:, prompt_ignore_length:]
    score = torch.gather(scores, 1, input_ids)
    score = torch.where(score < 0, score * penalty, score / penalty)
    return scores.scatter(1, input_ids, score)


# ------------------------------------------------------------- processors ---
class RefConstrainedLogitsProcessor:
    """logit_processor.py:8-49 (+ ``initial_count``: analyze_rl_beam_hint.py:553-560)."""

    def __init__(self, prefix_allowed_tokens_fn, num_beams, prefix_index=3, prefix_ids=None, eos_token_id=None,
                 initial_count: int = 0):
        self._prefix_allowed_tokens_fn = prefix_allowed_tokens_fn
        self._num_beams = num_beams
        self.prefix_index = prefix_index
        self.prefix_ids = prefix_ids
        self.eos_token_id = eos_token_id
        self.initial_count = initial_count
        self.count = initial_count

    def reset(self) -> None:
        self.count = self.initial_count

    def allowed_sets(self, input_ids: torch.Tensor) -> list[list[int]]:
        assert input_ids.dim() == 2, "input_ids must be a 2D tensor"
        input_ids.view(-1, self._num_beams, input_ids.shape[1])  # same shape check as the reference (36)
        allowed, self.count = count_window_allowed(
            input_ids, self._prefix_allowed_tokens_fn, self.prefix_index, self.prefix_ids, self.count
        )
        return allowed

    def __call__(self, input_ids: torch.Tensor, scores: torch.Tensor) -> torch.Tensor:
        allowed = self.allowed_sets(input_ids)
        out = masked_log_softmax(scores, allowed)
        self.count += 1
        return out


class RefFixedHintConstrainedLogitsProcessor:
    """fixed_hint_logit_processor.py:64-96."""

    def __init__(self, prefix_allowed_tokens_fn, num_beams, prefix_index=3, prefix_ids=None, eos_token_id=None):
        self._prefix_allowed_tokens_fn = prefix_allowed_tokens_fn
        self._num_beams = num_beams
        self.prefix_index = prefix_index
        self.prefix_ids = prefix_ids
        self.eos_token_id = eos_token_id

    def allowed_sets(self, input_ids: torch.Tensor) -> list[list[int]]:
        assert input_ids.dim() == 2, "input_ids must be a 2D tensor"
        input_ids.view(-1, self._num_beams, input_ids.shape[1])
        return [
            fixed_hint_allowed_row(sent, self.prefix_ids, self._prefix_allowed_tokens_fn, self.eos_token_id)
            for sent in (row.tolist() for row in input_ids)
        ]

    def __call__(self, input_ids: torch.Tensor, scores: torch.Tensor) -> torch.Tensor:
        return masked_log_softmax(scores, self.allowed_sets(input_ids))


def compact_from_dense(dense: torch.Tensor, cmax: int):
    """(count, token, logp) form of a processor output: the finite columns of
    each row, ascending by token id -- what the fused GPU entry point emits."""
    R = dense.shape[0]
    count = torch.zeros(R, dtype=torch.int32)
    tok = torch.zeros((R, cmax), dtype=torch.int32)
    logp = torch.zeros((R, cmax), dtype=torch.float32)
    for r in range(R):
        cols = torch.nonzero(dense[r] != float("-inf")).flatten()
        n = cols.numel()
        count[r] = n
        tok[r, :n] = cols.to(torch.int32)
        logp[r, :n] = dense[r, cols].float()
    return count, tok, logp
