"""Prefix trie as a flat CSR child table, plus the index.json -> trie builder.

Host-side mirror of ``llamafactory.extras.trie.Trie``
(reference: src/llamafactory/extras/trie.py:18-97) and of
``build_trie_from_index`` / ``create_prefix_allowed_tokens_fn``
(reference: evaluate.py:48-104).  The table is what the kernels walk:

    node 0 is the root;
    the children of node n are edges child_off[n] .. child_off[n+1]-1;
    child_tok[e] is the token on edge e, ascending inside a node;
    edge e leads to node e + 1.
"""

from __future__ import annotations

import ctypes as C
import hashlib
import json
import os
import struct
from collections import deque
from typing import Callable, Iterable, Iterator, Optional, Sequence

import numpy as np

from . import _lib


class FlatTrie:
    """Token trie stored as (child_off, child_tok) int32 arrays.

    ``get(prefix)`` answers like the reference ``Trie.get`` (trie.py:40-41,57-81):
    the tokens that may follow ``prefix``, ``[]`` when the prefix leaves the
    trie or ends on a leaf.  The reference returns children in first-insertion
    order; here they come back ascending (callers only use them as a set).
    ``len(trie)`` is the number of sequences added (trie.py:23-27,93-94).
    """

    def __init__(self, sequences: Optional[Iterable[Sequence[int]]] = None):
        self.len = 0
        nested: dict = {}
        if sequences:
            for seq in sequences:
                node = nested
                for tok in seq:
                    node = node.setdefault(int(tok), {})
                self.len += 1
        self._set_from_nested(nested)
        self._device_handles: dict = {}

    # -- construction ---------------------------------------------------------
    def _set_from_nested(self, nested: dict) -> None:
        off = [0]
        toks: list[int] = []
        queue = deque([nested])
        while queue:
            node = queue.popleft()
            for tok in sorted(node):
                toks.append(tok)
                queue.append(node[tok])
            off.append(len(toks))
        # nodes are numbered in BFS order and children are appended in that same
        # order, so the child behind edge e is node e + 1
        self.child_off = np.asarray(off, dtype=np.int32)
        self.child_tok = np.asarray(toks, dtype=np.int32)
        if len(toks) and (min(toks) < 0 or max(toks) > 0x7FFFFFFF):
            raise ValueError("token ids must fit in int32")
        assert self.child_off.shape[0] == self.child_tok.shape[0] + 2

    @classmethod
    def from_nested_dict(cls, trie_dict: dict, length: Optional[int] = None) -> "FlatTrie":
        """From the nested ``{token: {token: ...}}`` dict a reference Trie keeps in ``trie_dict``."""
        out = cls()
        out._set_from_nested(trie_dict)
        out.len = length if length is not None else sum(1 for _ in out)
        return out

    @classmethod
    def from_reference_trie(cls, trie) -> "FlatTrie":
        """From an object with the reference Trie's attributes (``trie_dict``, ``len``)."""
        if getattr(trie, "append_trie", None) is not None:
            raise NotImplementedError("chained tries (Trie.append) are not used by GenRec and not supported")
        return cls.from_nested_dict(trie.trie_dict, length=len(trie))

    @classmethod
    def from_allowed_tokens_fn(
        cls, fn: Callable[[list], list], max_depth: int = 64, max_nodes: int = 5_000_000
    ) -> "FlatTrie":
        """Enumerate an opaque ``prefix_allowed_tokens_fn`` breadth-first from the empty prefix.

        Works for any function of the form ``lambda ids: trie.get(ids)``
        (reference: evaluate.py:98-104).  One call per trie node.
        """
        nested: dict = {}
        queue = deque([([], nested)])
        nodes = 0
        while queue:
            prefix, node = queue.popleft()
            if len(prefix) > max_depth:
                raise ValueError(f"prefix_allowed_tokens_fn does not terminate within depth {max_depth}")
            for tok in fn(list(prefix)):
                child: dict = {}
                node[int(tok)] = child
                queue.append((prefix + [int(tok)], child))
                nodes += 1
                if nodes > max_nodes:
                    raise ValueError("prefix_allowed_tokens_fn enumerates more than max_nodes nodes")
        return cls.from_nested_dict(nested)

    # -- binary form ------------------------------------------------------------
    # Layout (little endian): 8-byte magic "GRXTRIE1", int32 n_nodes, int32 n_edges, int64 len,
    # 32-byte key (SHA-256 of whatever the trie was built from, zeros when unknown),
    # child_off int32[n_nodes + 1], child_tok int32[n_edges].
    MAGIC = b"GRXTRIE1"

    def save(self, path: str, key: bytes = b"") -> None:
        key = (key + b"\0" * 32)[:32]
        tmp = f"{path}.tmp{os.getpid()}"
        with open(tmp, "wb") as f:
            f.write(self.MAGIC)
            f.write(struct.pack("<iiq", self.n_nodes, self.n_edges, self.len))
            f.write(key)
            f.write(np.ascontiguousarray(self.child_off, dtype="<i4").tobytes())
            f.write(np.ascontiguousarray(self.child_tok, dtype="<i4").tobytes())
        os.replace(tmp, path)  # readers never see a half-written cache

    @classmethod
    def load(cls, path: str, key: Optional[bytes] = None) -> "FlatTrie":
        """Read a file written by ``save``; with ``key`` given, a different stored key raises KeyError."""
        with open(path, "rb") as f:
            head = f.read(8 + 16 + 32)
            if len(head) != 56 or head[:8] != cls.MAGIC:
                raise ValueError(f"{path} is not a trie cache")
            n_nodes, n_edges, length = struct.unpack("<iiq", head[8:24])
            if key is not None and head[24:56] != (key + b"\0" * 32)[:32]:
                raise KeyError(f"{path} was built from different inputs")
            if n_nodes < 1 or n_edges != n_nodes - 1:
                raise ValueError(f"{path}: inconsistent header")
            off = np.frombuffer(f.read(4 * (n_nodes + 1)), dtype="<i4")
            tok = np.frombuffer(f.read(4 * n_edges), dtype="<i4")
        if off.shape[0] != n_nodes + 1 or tok.shape[0] != n_edges or off[0] != 0 or off[-1] != n_edges:
            raise ValueError(f"{path}: truncated or corrupt")
        out = cls()
        out.child_off = off.astype(np.int32)
        out.child_tok = tok.astype(np.int32)
        out.len = int(length)
        return out

    # -- queries (host) ---------------------------------------------------------
    @property
    def n_nodes(self) -> int:
        return int(self.child_off.shape[0] - 1)

    @property
    def n_edges(self) -> int:
        return int(self.child_tok.shape[0])

    @property
    def max_fanout(self) -> int:
        return int(np.diff(self.child_off).max()) if self.n_nodes else 0

    @property
    def max_token(self) -> int:
        return int(self.child_tok.max()) if self.n_edges else -1

    def node_of(self, prefix: Sequence[int]) -> int:
        """Node reached from the root along ``prefix``, or -1."""
        node = 0
        off, tok = self.child_off, self.child_tok
        for t in prefix:
            lo, hi = int(off[node]), int(off[node + 1])
            pos = lo + int(np.searchsorted(tok[lo:hi], t))
            if pos >= hi or int(tok[pos]) != int(t):
                return -1
            node = pos + 1
        return node

    def children(self, node: int) -> list[int]:
        return self.child_tok[int(self.child_off[node]) : int(self.child_off[node + 1])].tolist()

    def get(self, prefix_sequence: Sequence[int]) -> list[int]:
        node = self.node_of(prefix_sequence)
        return [] if node < 0 else self.children(node)

    def __getitem__(self, prefix_sequence: Sequence[int]) -> list[int]:
        return self.get(prefix_sequence)

    def __len__(self) -> int:
        return self.len

    def __iter__(self) -> Iterator[list[int]]:
        """Root-to-leaf token sequences (reference: trie.py:83-91)."""

        def walk(node: int, path: list[int]):
            lo, hi = int(self.child_off[node]), int(self.child_off[node + 1])
            if lo == hi:
                yield path
                return
            for e in range(lo, hi):
                yield from walk(e + 1, path + [int(self.child_tok[e])])

        return walk(0, [])

    def level_fanout(self) -> list[tuple[int, float]]:
        """(max, mean) number of children per depth, root first."""
        out = []
        level = [0]
        while level:
            deg = [int(self.child_off[n + 1] - self.child_off[n]) for n in level]
            if not any(deg):
                break
            out.append((max(deg), sum(deg) / len(deg)))
            level = [e + 1 for n in level for e in range(int(self.child_off[n]), int(self.child_off[n + 1]))]
        return out

    # -- device -----------------------------------------------------------------
    def device_handle(self, device) -> int:
        """grx_trie* for ``device`` (a torch.device); uploaded once and kept."""
        import torch

        device = torch.device(device)
        if device.type != "cuda":
            raise _lib.GenrecHipError(f"the trie lives in HBM; got device {device}")
        key = device.index if device.index is not None else torch.cuda.current_device()
        handle = self._device_handles.get(key)
        if handle is None:
            out = C.c_void_p()
            off = np.ascontiguousarray(self.child_off, dtype=np.int32)
            tok = np.ascontiguousarray(self.child_tok, dtype=np.int32)
            with torch.cuda.device(key):
                _lib.check(
                    _lib.lib().grx_trie_create(
                        off.ctypes.data_as(C.c_void_p),
                        tok.ctypes.data_as(C.c_void_p) if tok.size else None,
                        self.n_nodes,
                        self.n_edges,
                        C.byref(out),
                    )
                )
            handle = _TrieHandle(out.value)
            self._device_handles[key] = handle
        return handle.ptr


class _TrieHandle:
    def __init__(self, ptr: int):
        self.ptr = ptr

    def __del__(self):
        try:
            if self.ptr:
                _lib.lib().grx_trie_destroy(self.ptr)
        except Exception:
            pass
        self.ptr = None


# ---------------------------------------------------------------- index.json ---
def normalize_sid_tokens(raw_tokens) -> list[str]:
    """Token strings of one index entry (reference: evaluate.py:48-54)."""
    if isinstance(raw_tokens, str):
        raw_tokens = [raw_tokens]
    elif not isinstance(raw_tokens, list):
        return []
    cleaned = (str(tok).strip() for tok in raw_tokens)
    return [tok for tok in cleaned if tok]


def select_sid_tokens(tokens: list[str], sid_levels: int) -> list[str]:
    """Keep the first ``sid_levels`` levels; all when <= 0 (reference: evaluate.py:57-60)."""
    return tokens if sid_levels <= 0 else tokens[:sid_levels]


def get_assistant_prefix_from_tokenizer(tokenizer) -> str:
    """Text the chat template appends for generation (reference: evaluate.py:38-45)."""
    try:
        msgs = [{"role": "user", "content": "x"}]
        with_gen = tokenizer.apply_chat_template(msgs, add_generation_prompt=True, tokenize=False)
        without_gen = tokenizer.apply_chat_template(msgs, add_generation_prompt=False, tokenize=False)
        return with_gen[len(without_gen) :]
    except Exception:
        raise ValueError("Failed to get assistant prefix from tokenizer.")


def index_item_strings(index_path: str, sid_levels: int = -1) -> set[str]:
    """Unique concatenated SID strings of an index file (reference: evaluate.py:69-78)."""
    with open(index_path, encoding="utf-8") as f:
        index_data = json.load(f)
    items = set()
    for raw in index_data.values():
        tokens = select_sid_tokens(normalize_sid_tokens(raw), sid_levels)
        if tokens:
            items.add("".join(tokens))
    return items


def _trie_cache_key(index_path: str, tokenizer, prefix: str, prefix_ids: list[int], sid_levels: int) -> bytes:
    """Everything the token sequences depend on: the index bytes, the level cut, the prefix and
    its ids, EOS, and the ids the tokenizer gives to every SID token of the index."""
    h = hashlib.sha256()
    with open(index_path, "rb") as f:
        h.update(f.read())
    h.update(repr((sid_levels, prefix, prefix_ids, tokenizer.eos_token_id)).encode())
    with open(index_path, encoding="utf-8") as f:
        toks = sorted({t for raw in json.load(f).values() for t in normalize_sid_tokens(raw)})
    ids = tokenizer("".join(toks), add_special_tokens=False).input_ids if toks else []
    h.update(np.asarray(ids, dtype="<i8").tobytes())
    return h.digest()


def build_trie_from_index(index_path: str, tokenizer, prefix: Optional[str] = None, sid_levels: int = -1,
                          cache: bool = False):
    """index.json -> (trie, prompt_suffix_ids, prefix_index) (reference: evaluate.py:63-95).

    Every trie path is ``prefix_ids + sid_ids + [eos]``; ``len(trie)`` is the
    number of distinct items.

    ``cache=True`` (extension) keeps the flat trie in ``<index_path>.<key>.grxtrie`` next to the index
    and reads it back when index, tokenizer ids, prefix and level cut are unchanged: every rank
    and every run after the first skips the JSON parse and the per-item tokenizer calls.
    """
    if prefix is None:
        prefix = get_assistant_prefix_from_tokenizer(tokenizer)
    prefix_ids = list(tokenizer(prefix, add_special_tokens=False).input_ids)
    cache_path = key = None
    if cache:
        key = _trie_cache_key(index_path, tokenizer, prefix, prefix_ids, sid_levels)
        cache_path = f"{index_path}.{key.hex()[:16]}.grxtrie"
        if os.path.exists(cache_path):
            try:
                return FlatTrie.load(cache_path, key), prefix_ids, len(prefix_ids)
            except (KeyError, ValueError, OSError):
                pass  # stale or damaged: rebuild below
    items = index_item_strings(index_path, sid_levels)
    eos_id = tokenizer.eos_token_id
    sequences = []
    for item in items:
        full = tokenizer(prefix + item, add_special_tokens=False).input_ids
        sequences.append(prefix_ids + list(full[len(prefix_ids) :]) + [eos_id])
    trie = FlatTrie(sequences)
    if cache_path is not None:
        try:
            trie.save(cache_path, key)
        except OSError:
            pass  # read-only data directory: the cache is only an optimisation
    return trie, prefix_ids, len(prefix_ids)


def create_prefix_allowed_tokens_fn(trie, prompt_suffix_ids: list[int]) -> Callable[[list[int]], list[int]]:
    """Closure ``ids -> trie.get(ids)`` (reference: evaluate.py:98-104).

    The closure carries the trie in ``__genrec_trie__`` so a processor can
    take the CSR table directly instead of enumerating the function.
    """

    def prefix_allowed_tokens_fn(input_ids: list[int]) -> list[int]:
        return trie.get(input_ids)

    prefix_allowed_tokens_fn.__genrec_trie__ = trie
    return prefix_allowed_tokens_fn


def as_flat_trie(source) -> FlatTrie:
    """FlatTrie from any accepted trie source: a FlatTrie, a reference-style Trie
    (``trie_dict``), or a ``prefix_allowed_tokens_fn`` (tagged or opaque)."""
    if isinstance(source, FlatTrie):
        return source
    tagged = getattr(source, "__genrec_trie__", None)
    if tagged is not None:
        return as_flat_trie(tagged)
    if hasattr(source, "trie_dict"):
        return FlatTrie.from_reference_trie(source)
    if callable(source):
        # a closure over a reference-style Trie (evaluate.py:98-104) hands it over directly
        for cell in getattr(source, "__closure__", None) or ():
            try:
                captured = cell.cell_contents
            except ValueError:
                continue
            if isinstance(captured, FlatTrie) or hasattr(captured, "trie_dict"):
                return as_flat_trie(captured)
        return FlatTrie.from_allowed_tokens_fn(source)
    raise TypeError(f"cannot build a trie from {type(source).__name__}")
