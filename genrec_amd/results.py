"""Evaluation outputs straight from beam token ids.

The reference decodes every beam with ``tokenizer.batch_decode``, attaches the
strings to the test samples as ``"predict"``, dumps ``result.json`` and then
parses those strings again for the metrics (evaluate.py:372-405, 125-204).  Here
the ids leave the GPU once: the strings for ``result.json`` come from one table
lookup per token, and the metrics come from the ids themselves
(``genrec_amd.metrics.compute_metrics_from_ids``) without a detour through text.
"""

from __future__ import annotations

import json
import os
from typing import Optional, Sequence

import numpy as np
import torch

from .metrics import compute_metrics_from_ids, metrics_from_histogram, save_metrics  # noqa: F401


def token_text_table(tokenizer, n_rows: int, skip_special_tokens: bool = True) -> list[str]:
    """``table[token_id]`` = the text ``batch_decode`` contributes for that id: the token's own
    string for SID tokens, ``""`` for special tokens (EOS, padding) when they are skipped.
    Only ids that can occur in a completion need a faithful entry."""
    table = [""] * n_rows
    special = set(getattr(tokenizer, "all_special_ids", None) or [])
    for tid in (getattr(tokenizer, "eos_token_id", None), getattr(tokenizer, "pad_token_id", None)):
        if tid is not None:
            special.add(int(tid))
    added = getattr(tokenizer, "_id", None)  # SidTokenizer: text -> id
    if added is not None:
        for text, tid in added.items():
            table[tid] = text
        return table
    for tid in range(n_rows):
        if skip_special_tokens and tid in special:
            continue
        table[tid] = tokenizer.decode([tid], skip_special_tokens=skip_special_tokens,
                                      clean_up_tokenization_spaces=False)
    return table


def decode_beams(beam_ids, table: Sequence[str]) -> list[list[str]]:
    """[n, n_beam, L] ids -> per sample the list of beam strings (evaluate.py:372-380)."""
    ids = beam_ids.detach().cpu().numpy() if isinstance(beam_ids, torch.Tensor) else np.asarray(beam_ids)
    lut = np.asarray(table, dtype=object)
    pieces = lut[ids]  # [n, n_beam, L] strings
    return [["".join(row) for row in sample] for sample in pieces]


def write_result_json(test_data: list, beam_ids, table: Sequence[str], result_json_path: str,
                      compute_metrics_flag: bool = True, target_ids: Optional[torch.Tensor] = None,
                      topk_list=None) -> Optional[dict]:
    """Attach ``"predict"`` to every sample and write ``result.json`` exactly as the reference does
    (evaluate.py:382-389: ``json.dump(test_data, indent=4, ensure_ascii=False)``).  With
    ``compute_metrics_flag`` the metrics are computed from ``beam_ids`` against ``target_ids``
    [n, L] on the device and saved next to the result file like evaluate.py:391-393 does."""
    outputs = decode_beams(beam_ids, table)
    for i, sample in enumerate(test_data):
        sample["predict"] = outputs[i] if i < len(outputs) else []
    os.makedirs(os.path.dirname(result_json_path) or ".", exist_ok=True)
    with open(result_json_path, "w", encoding="utf-8") as f:
        json.dump(test_data, f, indent=4, ensure_ascii=False)
    if not compute_metrics_flag or target_ids is None:
        return None
    # tokens without text (EOS, padding) must not tell a beam from its target, as they do not in the
    # decoded strings: map them to one filler and bring both sides to the same width
    beams = beam_ids if isinstance(beam_ids, torch.Tensor) else torch.as_tensor(np.asarray(beam_ids))
    dev = beams.device
    silent = torch.tensor([t == "" for t in table], dtype=torch.bool, device=dev)
    width = max(beams.shape[-1], target_ids.shape[-1])

    def norm(ids):
        ids = ids.to(device=dev, dtype=torch.int64)
        ids = torch.where(silent[ids.clamp(0, len(table) - 1)], torch.full_like(ids, -1), ids)
        pad = width - ids.shape[-1]
        return torch.nn.functional.pad(ids, (0, pad), value=-1) if pad else ids

    metrics = compute_metrics_from_ids(norm(beams).to(torch.int32), norm(target_ids).to(torch.int32), topk_list)
    if metrics:
        metrics_path = os.path.join(os.path.dirname(result_json_path), "metrics.json")
        valid = [k for k in (topk_list or [1, 3, 5, 10, 20, 50]) if f"HR@{k}" in metrics]
        save_metrics(metrics, metrics_path, valid)
    return metrics
