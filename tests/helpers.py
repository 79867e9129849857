"""Shared test helpers: golden-fixture loading and index.json reconstruction."""

from __future__ import annotations

import gzip
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
INDEX_NAMES = ("Industrial_and_Scientific", "Instruments", "Games", "Arts")


def load_npz(name: str):
    data = np.load(os.path.join(GOLDEN, name))
    meta = json.loads(bytes(data["meta_json"]).decode()) if "meta_json" in data.files else {}
    return data, meta


def load_json_gz(name: str):
    with gzip.open(os.path.join(GOLDEN, name), "rt", encoding="utf-8") as f:
        return json.load(f)


def index_from_codes(name: str) -> dict:
    """Rebuild the ``{item_id: ["<a_x>", ...]}`` index of a reference data file
    from the committed per-item code table (tests/golden/sids_<name>.npz)."""
    codes = np.load(os.path.join(GOLDEN, f"sids_{name}.npz"))["codes"]
    letters = "abcdefghij"
    return {
        str(i): [f"<{letters[l]}_{int(c)}>" for l, c in enumerate(row)] for i, row in enumerate(codes)
    }


def write_index_file(name: str, directory) -> str:
    path = os.path.join(str(directory), f"{name}.index.json")
    with open(path, "w", encoding="utf-8") as f:
        json.dump(index_from_codes(name), f)
    return path


def live(scores, threshold: float = -1.0e8):
    """Mask of beam slots that are not -1e9 placeholders / -inf fillers."""
    return scores > threshold
