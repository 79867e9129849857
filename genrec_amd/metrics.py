"""HR@k / NDCG@k of beam-search rankings, ranked on the device.

Mirror of ``compute_metrics`` (reference: evaluate.py:125-204): the rank of the
first beam equal to the target decides HR@k and the ``1/ln(rank+2)`` gain.  The
kernel (``grx_hit_ranks``) finds that rank per sample and builds the rank
histogram; HR and NDCG follow from the histogram in float64 on the host, and
the 4-decimal JSON/TSV dump is the reference's.
"""

from __future__ import annotations

import ctypes as C
import json
import math
import os
from typing import Optional

import numpy as np
import torch

from . import _lib

DEFAULT_TOPK = [1, 3, 5, 10, 20, 50]


def hit_ranks(beams: torch.Tensor, target: torch.Tensor) -> tuple[torch.Tensor, torch.Tensor]:
    """beams [n, n_beam, L] / target [n, L] int32 on the device ->
    (first_hit [n] with n_beam meaning "no beam matches", rank histogram [n_beam + 1])."""
    _lib.require_cuda(beams, "beams")
    _lib.require_cuda(target, "target")
    beams = beams.to(torch.int32).contiguous()
    target = target.to(torch.int32).contiguous()
    n, n_beam, L = beams.shape
    assert target.shape == (n, L)
    dev = beams.device
    first = torch.empty(n, dtype=torch.int32, device=dev)
    hist = torch.zeros(n_beam + 1, dtype=torch.int32, device=dev)
    index = dev.index if dev.index is not None else torch.cuda.current_device()
    with _lib.on_device(index):
        _lib.check(
            _lib.lib().grx_hit_ranks(beams.data_ptr(), target.data_ptr(), n, n_beam, L, first.data_ptr(),
                                     hist.data_ptr(), _lib.current_stream_ptr(dev))
        )
    return first, hist


def metrics_from_histogram(hist, n: int, n_beam: int, topk_list=None) -> dict:
    """HR@k and NDCG@k from the per-rank sample counts (reference: evaluate.py:143,168-180)."""
    topk_list = DEFAULT_TOPK if topk_list is None else topk_list
    hist = [int(x) for x in (hist.tolist() if hasattr(hist, "tolist") else hist)]
    valid = [k for k in topk_list if k <= n_beam] if n_beam > 0 else list(topk_list)
    out = {}
    for k in valid:
        if k > n_beam:
            hr, dcg = 0, 0.0
        else:
            hr = sum(hist[:k])
            dcg = 0.0
            for rank in range(k):
                dcg += hist[rank] * (1.0 / math.log(rank + 2))
        out[f"HR@{k}"] = hr / n
        out[f"NDCG@{k}"] = dcg / n / (1.0 / math.log(2))
    return out


def compute_metrics_from_ids(beams: torch.Tensor, target: torch.Tensor, topk_list=None) -> dict:
    """Metrics straight from code / token rows on the device."""
    n, n_beam, _ = beams.shape
    if n == 0:
        return {}
    _, hist = hit_ranks(beams, target)
    return metrics_from_histogram(hist.cpu(), n, n_beam, topk_list)


def _clean_pred(p) -> str:
    return str(p).strip('"\n').strip()


def compute_metrics(test_data: list, topk_list: list = None, save_path: Optional[str] = None) -> dict:
    """Drop-in for the reference's ``compute_metrics`` on ``result.json`` records
    (``{"predict": [...], "output": ...}``).  Strings are dictionary-encoded on the
    host; the ranking runs on the device."""
    if topk_list is None:
        topk_list = DEFAULT_TOPK
    n = len(test_data)
    if n == 0:
        return {}
    first_predict = test_data[0].get("predict", [])
    if isinstance(first_predict, str):
        n_beam = 1 if first_predict else 0
    else:
        n_beam = len(first_predict) if first_predict else 0

    vocab: dict[str, int] = {}

    def code(s: str) -> int:
        return vocab.setdefault(s, len(vocab))

    width = 0
    rows, targets = [], []
    for sample in test_data:
        predict = sample.get("predict", "")
        target = sample.get("output", "")
        if isinstance(predict, str):
            preds = [predict.strip('"\n').strip()] if predict else [""]
        else:
            preds = [_clean_pred(p) for p in predict]
        if isinstance(target, list):
            tgt = target[0].strip('"\n').strip(" ") if target else ""
        else:
            tgt = str(target).strip(' \n"')
        rows.append([code(p) for p in preds])
        targets.append(code(tgt))
        width = max(width, len(preds))
    lens = [len(r) for r in rows]
    if width > 0 and not torch.cuda.is_available():
        raise _lib.GenrecHipError("compute_metrics ranks on the GPU; no HIP device is visible and there is no CPU path")
    uniform = min(lens) == max(lens) and width == n_beam
    if width == 0:
        first = [0] * n  # no predictions anywhere: no cut-off can hit
    else:
        dev = torch.device("cuda", torch.cuda.current_device())
        # shorter prediction lists are padded with a code no target has
        padded = [r + [-1] * (width - len(r)) for r in rows]
        beams = torch.tensor(padded, dtype=torch.int32, device=dev).view(n, width, 1)
        tgt = torch.tensor(targets, dtype=torch.int32, device=dev).view(n, 1)
        first_hit, hist = hit_ranks(beams, tgt)
        if uniform:
            metrics = metrics_from_histogram(hist.cpu(), n, n_beam, topk_list)
            if metrics and save_path:
                save_metrics(metrics, save_path, [k for k in topk_list if f"HR@{k}" in metrics])
            return metrics
        first = first_hit.cpu().tolist()
    # Prediction lists of different lengths (or a first sample without predictions, which makes every
    # cut-off valid): the reference takes the cut-offs from the first sample and skips, per sample, every
    # k larger than that sample's own list (evaluate.py:137-143, 168-170).  The ranks are known; the
    # per-sample rule is applied on the host.
    valid = [k for k in topk_list if k <= n_beam] if n_beam > 0 else list(topk_list)
    metrics = {}
    for k in valid:
        hr, dcg = 0, 0.0
        for i in range(n):
            if k <= lens[i] and first[i] < k:
                hr += 1
                dcg += 1.0 / math.log(first[i] + 2)
        metrics[f"HR@{k}"] = hr / n
        metrics[f"NDCG@{k}"] = dcg / n / (1.0 / math.log(2))
    if metrics and save_path:
        save_metrics(metrics, save_path, [k for k in topk_list if f"HR@{k}" in metrics])
    return metrics


def save_metrics(metrics: dict, save_path: str, valid_topk: list) -> None:
    """Write ``metrics.json`` (values rounded to 4 decimals, indent 2) and, next to it, a one-row
    ``metrics.tsv`` whose row label is the result directory without its first path component --
    the two files evaluate.py:182-202 leaves behind (there through pandas; the bytes are the same)."""
    shown = {}
    for name, value in metrics.items():
        try:
            shown[name] = round(float(value), 4)
        except (TypeError, ValueError):
            shown[name] = value
    folder = os.path.dirname(save_path) or "."
    os.makedirs(folder, exist_ok=True)
    with open(save_path, "w", encoding="utf-8") as f:
        json.dump(shown, f, indent=2, ensure_ascii=False)

    columns = [f"{kind}@{k}" for kind in ("HR", "NDCG") for k in valid_topk]
    label = folder.split("/", 1)[1] if "/" in folder else folder
    cells = ["" if shown.get(c) is None else repr(shown[c]) for c in columns]
    with open(os.path.splitext(save_path)[0] + ".tsv", "w", encoding="utf-8", newline="") as f:
        f.write("\t".join(["path"] + columns) + "\n")
        f.write("\t".join([label] + cells) + "\n")
