"""genrec_amd: trie-constrained semantic-ID decoding for MI355X.

Drop-in, GPU-resident replacements for the decoding hot path of WncFht/GenRec:
the trie, the two constrained logits processors, the beam step, the rule /
prefix / NDCG rewards, HR@k / NDCG@k and the group advantages.  Everything
computes in libgenrec_hip.so (hand-written gfx950 kernels behind the C ABI in
include/genrec_hip.h); there is no CPU fallback.
"""

from .trie import (  # noqa: F401
    FlatTrie,
    build_trie_from_index,
    create_prefix_allowed_tokens_fn,
)

__all__ = [
    "FlatTrie",
    "build_trie_from_index",
    "create_prefix_allowed_tokens_fn",
    "ConstrainedLogitsProcessor",
    "FixedHintConstrainedLogitsProcessor",
]


def __getattr__(name):
    # the processors import torch + transformers; load them on first use
    if name == "ConstrainedLogitsProcessor":
        from .logit_processor import ConstrainedLogitsProcessor

        return ConstrainedLogitsProcessor
    if name == "FixedHintConstrainedLogitsProcessor":
        from .fixed_hint_logit_processor import FixedHintConstrainedLogitsProcessor

        return FixedHintConstrainedLogitsProcessor
    raise AttributeError(name)
