"""Stateless trie-constrained processor for prompts that carry an oracle SID hint.

Drop-in for ``FixedHintConstrainedLogitsProcessor``
(reference: fixed_hint_logit_processor.py:64-96).  Per row, the allowed set is
``{eos}`` if the row already ends in EOS; otherwise the continuation of the
first occurrence of ``prefix_ids`` in the whole row that is still a valid trie
path (fixed_hint_logit_processor.py:18-61).  The scan runs on the device, one
wave per row.
"""

from __future__ import annotations

from typing import Callable, Optional

from . import _lib
from .logit_processor import _TrieConstrainedBase


def find_last_prefix_match_start(sent: list[int], prefix_ids: list[int]) -> int:
    """Start of the last occurrence of ``prefix_ids`` in ``sent``
    (host helper; reference: fixed_hint_logit_processor.py:8-15)."""
    n = len(prefix_ids) if prefix_ids else 0
    if n == 0:
        return 0
    for start in reversed(range(len(sent) - n + 1)):
        if sent[start : start + n] == prefix_ids:
            return start
    raise ValueError("Failed to find prompt suffix prefix_ids inside current sequence.")


def find_valid_prefix_match_start(
    sent: list[int], prefix_ids: list[int], prefix_allowed_tokens_fn: Callable[[list[int]], list[int]]
) -> int:
    """First occurrence of ``prefix_ids`` whose continuation the trie accepts
    (host helper; reference: fixed_hint_logit_processor.py:18-40)."""
    n = len(prefix_ids) if prefix_ids else 0
    if n == 0:
        return 0
    starts = [s for s in range(len(sent) - n + 1) if sent[s : s + n] == prefix_ids]
    for s in starts:
        if prefix_allowed_tokens_fn(sent[s:]):
            return s
    raise ValueError(
        "Failed to find a valid trie prefix match inside current sequence. "
        f"prefix_ids={prefix_ids}, candidate_starts={starts}, sent={sent}"
    )


def resolve_allowed_tokens_for_sequence(
    sent: list[int],
    prefix_ids: list[int],
    prefix_allowed_tokens_fn: Callable[[list[int]], list[int]],
    eos_token_id: Optional[int],
) -> list[int]:
    """Allowed next tokens of one row (host helper; reference: fixed_hint_logit_processor.py:43-61)."""
    if eos_token_id is not None and sent and sent[-1] == eos_token_id:
        return [eos_token_id]
    start = find_valid_prefix_match_start(sent, prefix_ids, prefix_allowed_tokens_fn)
    allowed = prefix_allowed_tokens_fn(sent[start:])
    if len(allowed) == 0:
        raise AssertionError("No valid tokens for prefix_ids")
    return allowed


class FixedHintConstrainedLogitsProcessor(_TrieConstrainedBase):
    """Prefix-scan processor; same constructor as the count-based one, no ``count``."""

    _mode = _lib.GRX_MODE_FIXED_HINT
