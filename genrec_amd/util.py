"""Builders that turn an index.json into a ready ``LogitsProcessorList``.

Mirror of the reference's ``util.py`` (build_constrained_logits_processor 70-99,
build_fixed_hint_constrained_logits_processor 102-130, _extract_number 133-134,
main-process printing 24-38): same names and signatures, so ``trl_trainer.py``
and ``evaluate.py`` can import them from here unchanged.
"""

from __future__ import annotations

import os
import re
from typing import Any, Optional

from transformers import LogitsProcessorList

from .fixed_hint_logit_processor import FixedHintConstrainedLogitsProcessor
from .logit_processor import ConstrainedLogitsProcessor
from .trie import build_trie_from_index, create_prefix_allowed_tokens_fn


_RANK_VARIABLES = ("RANK", "LOCAL_RANK", "ACCELERATE_PROCESS_INDEX", "PROCESS_INDEX")


def is_main_process() -> bool:
    """Rank 0 or no launcher at all: the first rank variable that parses as an integer decides
    (same variables and precedence as the reference's util.py:24-33)."""
    for raw in map(os.environ.get, _RANK_VARIABLES):
        try:
            return int(raw) == 0
        except (TypeError, ValueError):  # unset or not a number: look at the next variable
            pass
    return True


def print_main_process(*args: Any, **kwargs: Any) -> None:
    if is_main_process():
        print(*args, **kwargs)


def _build(cls, label: str, index_path: str, tokenizer, prefix, num_beams: int, sid_levels: int,
           **processor_options) -> LogitsProcessorList:
    print_main_process(f"Building {label}Trie from {index_path}...")
    trie, prompt_suffix_ids, prefix_index = build_trie_from_index(
        index_path, tokenizer, prefix=prefix, sid_levels=sid_levels
    )
    print_main_process(f"{label}Trie built: prefix_index={prefix_index}, num_items={len(trie)}")
    fn = create_prefix_allowed_tokens_fn(trie, prompt_suffix_ids)
    return LogitsProcessorList(
        [
            cls(
                prefix_allowed_tokens_fn=fn,
                num_beams=num_beams,
                prefix_index=prefix_index,
                prefix_ids=prompt_suffix_ids,
                eos_token_id=tokenizer.eos_token_id,
                **processor_options,
            )
        ]
    )


def build_constrained_logits_processor(index_path: str, tokenizer, prefix: Optional[str] = None, num_beams: int = 50,
                                       sid_levels: int = -1, **processor_options) -> LogitsProcessorList:
    """Trie from the index + count-based processor in a ``LogitsProcessorList`` (reference signature;
    ``processor_options`` are this package's keyword-only extras, e.g. ``assume_normalized=True``)."""
    return _build(ConstrainedLogitsProcessor, "", index_path, tokenizer, prefix, num_beams, sid_levels,
                  **processor_options)


def build_fixed_hint_constrained_logits_processor(index_path: str, tokenizer, prefix: Optional[str] = None,
                                                  num_beams: int = 50, sid_levels: int = -1,
                                                  **processor_options) -> LogitsProcessorList:
    """Trie from the index + fixed-hint processor in a ``LogitsProcessorList``."""
    return _build(FixedHintConstrainedLogitsProcessor, "FixedHint ", index_path, tokenizer, prefix, num_beams,
                  sid_levels, **processor_options)


def _extract_number(text: str) -> list[int]:
    return [int(m) for m in re.findall(r"_(\d+)", text)]
