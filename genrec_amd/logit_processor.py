"""Trie-constrained logits processors running on MI355X.

Drop-in for the reference's ``ConstrainedLogitsProcessor``
(reference: logit_processor.py:8-49), its seeded subclass
(reference: analyze_rl_beam_hint.py:553-560) and -- in
``fixed_hint_logit_processor.py`` of this package --
``FixedHintConstrainedLogitsProcessor``.  Same constructor, attributes and
``__call__(input_ids, scores) -> scores`` contract; the per-row Python loop,
the nested-dict trie walk and the three dense tensor passes are replaced by two
HIP kernels working on a CSR trie resident in HBM, and the shared ``count``
lives on the device so that decode steps need no host round trip.
"""

from __future__ import annotations

import ctypes as C
import os
from typing import Callable, Optional

import torch
from transformers.generation import LogitsProcessor

from . import _lib
from .trie import FlatTrie, as_flat_trie

_EMPTY_MSG = "No valid tokens for prefix_ids"
_NOMATCH_MSG = "Failed to find a valid trie prefix match inside current sequence."


def _eager_errors_env() -> Optional[bool]:
    """GENREC_AMD_EAGER_ERRORS=1 / 0 forces eager / deferred error reporting everywhere; unset: the default."""
    v = os.environ.get("GENREC_AMD_EAGER_ERRORS")
    if v is None or v == "":
        return None
    return v not in ("0", "false", "False")


class _DeviceState:
    """grx_state of one processor on one device."""

    def __init__(self, device_index: int):
        self.device_index = device_index
        out = C.c_void_p()
        with torch.cuda.device(device_index):
            _lib.check(_lib.lib().grx_state_create(C.byref(out)))
        self.ptr = out.value

    def __del__(self):
        try:
            if self.ptr:
                _lib.lib().grx_state_destroy(self.ptr)
        except Exception:
            pass
        self.ptr = None


class Candidates:
    """Compact candidate rows on the device (the ``grx_cand`` layout of the C ABI).

    ``buf`` is int32 [rows, stride, 2]: entry 0 of a row is (count, bits of the
    row's log-sum-exp), entry 1 + i is (token_i, bits of logp_i).  ``stride`` is a
    multiple of 16 entries so every row starts on its own 128-byte line.
    """

    def __init__(self, rows: int, cmax: int, device):
        self.rows, self.cmax = int(rows), int(cmax)
        self.stride = int(_lib.lib().grx_candidate_stride(self.cmax))
        self.buf = torch.empty((self.rows, self.stride, 2), dtype=torch.int32, device=device)
        assert self.buf.data_ptr() % 128 == 0

    @property
    def count(self) -> torch.Tensor:
        return self.buf[:, 0, 0]

    @property
    def logsumexp(self) -> torch.Tensor:
        return self.buf[:, 0, 1].view(torch.float32)

    @property
    def tokens(self) -> torch.Tensor:
        return self.buf[:, 1 : 1 + self.cmax, 0]

    @property
    def logp(self) -> torch.Tensor:
        return self.buf[:, 1 : 1 + self.cmax, 1].view(torch.float32)

    @classmethod
    def from_parts(cls, count: torch.Tensor, tokens: torch.Tensor, logp: torch.Tensor, device) -> "Candidates":
        """Pack separate (count [R], tokens [R, cmax], logp [R, cmax]) tensors."""
        out = cls(count.shape[0], tokens.shape[1], device)
        out.buf.zero_()
        out.buf[:, 0, 0] = count.to(device=device, dtype=torch.int32)
        out.buf[:, 1 : 1 + out.cmax, 0] = tokens.to(device=device, dtype=torch.int32)
        out.buf[:, 1 : 1 + out.cmax, 1] = logp.to(device=device, dtype=torch.float32).view(torch.int32)
        return out


class ChunkedStep:
    """One processor call over column slices of the logits (``grx_constrained_partial`` per slice,
    ``grx_constrained_finish`` at the end).  Built by ``processor.chunked_step``."""

    def __init__(self, proc, input_ids, cand: Candidates, v_total: int, bounds, dtype, one_row_per_group: bool,
                 parts: Optional[int] = None):
        if proc.assume_normalized:
            raise _lib.GenrecHipError("the chunked step does not support assume_normalized (it needs whole rows)")
        bounds = [(int(a), int(b)) for a, b in bounds]
        covered = sorted(bounds)
        if not covered or covered[0][0] != 0 or covered[-1][1] != int(v_total) or any(
                covered[i][1] != covered[i + 1][0] for i in range(len(covered) - 1)) or any(a >= b for a, b in covered):
            raise ValueError(f"column ranges {bounds} do not cover [0, {v_total}) exactly once")
        _lib.require_cuda(input_ids, "input_ids")
        assert input_ids.dim() == 2, "input_ids must be a 2D tensor"
        if not one_row_per_group and input_ids.shape[0] % proc._num_beams != 0:
            raise RuntimeError(
                f"shape '[-1, {proc._num_beams}, {input_ids.shape[1]}]' is invalid for input of size {input_ids.numel()}")
        if input_ids.dtype != torch.int64:
            input_ids = input_ids.to(torch.int64)
        if input_ids.shape[1] > 0 and input_ids.stride(1) != 1:
            input_ids = input_ids.contiguous()
        self.proc, self.ids, self.cand = proc, input_ids, cand
        self.R = input_ids.shape[0]
        assert cand.rows == self.R and cand.buf.device == input_ids.device
        self.v_total = int(v_total)
        self.dtype_code = _lib.torch_dtype_code(dtype)
        self.dtype = dtype
        lib = _lib.lib()
        self.slots = {}
        n = 0
        for a, b in bounds:
            p = int(parts) if parts else int(lib.grx_chunk_parts(self.R, b - a, self.dtype_code))
            self.slots[a] = (b, n, p)
            n += p
        self.n_slots = n
        self.index = proc._device_index(input_ids)
        self.partials = proc._partials(self.index, self.R * n * 2)
        # fused repetition penalty: the slices leave the raw logits of every row's own tokens here
        self.pen_width = max(input_ids.shape[1] - proc.prompt_ignore_length, 0) if proc.repetition_penalty != 1.0 else 0
        self.pen_logits = proc._pen_logits(self.index, self.R * self.pen_width) if self.pen_width else None
        self.fed = 0
        with torch.cuda.device(self.index):
            self.state = proc._state(self.index)
            self.ws = proc._workspace(self.index, self.R)
        self.con = proc._constraint()

    def _rows(self) -> "_lib.GrxRows":
        rows = _lib.GrxRows()
        rows.input_ids = self.ids.data_ptr()
        rows.ids_stride = self.ids.stride(0) if self.R > 1 else max(self.ids.shape[1], 1)
        rows.R, rows.T = self.R, self.ids.shape[1]
        return rows

    def feed(self, v_begin: int, scores: torch.Tensor) -> None:
        """``scores`` [rows, v_end - v_begin]: the logits of tokens ``v_begin .. v_end - 1``."""
        proc = self.proc
        v_end, slot, parts = self.slots[int(v_begin)]
        _lib.require_cuda(scores, "scores")
        if scores.shape != (self.R, v_end - v_begin) or scores.dtype != self.dtype or scores.device != self.ids.device:
            raise ValueError(f"slice at {v_begin} must be [{self.R}, {v_end - v_begin}] {self.dtype} on {self.ids.device}")
        if scores.stride(1) != 1:
            scores = scores.contiguous()
        rows = self._rows()
        rows.scores = scores.data_ptr()
        rows.scores_stride = scores.stride(0) if self.R > 1 else scores.shape[1]
        rows.V = scores.shape[1]
        rows.dtype = self.dtype_code
        chunk = _lib.GrxChunk()
        chunk.v_begin, chunk.v_total = int(v_begin), self.v_total
        chunk.slot, chunk.n_slots, chunk.parts = slot, self.n_slots, parts
        chunk.first = 1 if self.fed == 0 else 0
        if self.pen_logits is not None:
            chunk.pen_logits, chunk.pen_stride = self.pen_logits.data_ptr(), self.pen_width
        if chunk.first:
            proc._take_planned(rows, self.ids)
        with torch.cuda.device(self.index):
            _lib.check(
                _lib.lib().grx_constrained_partial(
                    proc.trie.device_handle(self.ids.device), self.state.ptr, C.byref(rows), C.byref(self.con),
                    C.byref(chunk), self.partials.data_ptr(), self.cand.buf.data_ptr(), self.cand.stride,
                    self.cand.cmax, self.ws.data_ptr(), self.ws.numel(), _lib.current_stream_ptr(self.ids.device),
                )
            )
        self.fed += 1
        proc._active = self.index

    def finish(self) -> Candidates:
        if self.fed != len(self.slots):
            raise RuntimeError(f"{self.fed} of {len(self.slots)} column slices were fed")
        with torch.cuda.device(self.index):
            if self.pen_logits is not None:
                rows = self._rows()
                _lib.check(
                    _lib.lib().grx_constrained_finish_penalised(
                        C.byref(rows), C.byref(self.con), self.v_total, self.n_slots, self.partials.data_ptr(),
                        self.pen_logits.data_ptr(), self.pen_width, self.cand.buf.data_ptr(), self.cand.stride,
                        self.cand.cmax, self.ws.data_ptr(), self.ws.numel(), _lib.current_stream_ptr(self.ids.device),
                    )
                )
            else:
                _lib.check(
                    _lib.lib().grx_constrained_finish(
                        self.R, self.n_slots, self.partials.data_ptr(), self.cand.buf.data_ptr(), self.cand.stride,
                        self.cand.cmax, self.ws.data_ptr(), self.ws.numel(), _lib.current_stream_ptr(self.ids.device),
                    )
                )
        if self.proc.sync_errors:
            self.proc.check_errors()
        return self.cand


class _TrieConstrainedBase(LogitsProcessor):
    """State and launch plumbing shared by the count-based and fixed-hint processors."""

    _mode = _lib.GRX_MODE_COUNT

    def __init__(
        self,
        prefix_allowed_tokens_fn: Callable[[list[int]], list[int]],
        num_beams: int,
        prefix_index: int = 3,
        prefix_ids: Optional[list[int]] = None,
        eos_token_id: Optional[int] = None,
        *,
        sync_errors: Optional[bool] = None,
        repetition_penalty: float = 1.0,
        prompt_ignore_length: int = 0,
        assume_normalized: bool = False,
        reuse_output: bool = False,
    ):
        self._prefix_allowed_tokens_fn = prefix_allowed_tokens_fn
        self._num_beams = num_beams
        self.prefix_index = prefix_index
        self.eos_token_id = eos_token_id
        self.prefix_ids = prefix_ids
        # When do the reference's exceptions (logit_processor.py:43-44, fixed_hint_logit_processor.py:37-40,
        # 59-60) surface?  The kernels record the first failing row on the device; raising needs a read-back.
        # Default: the processor surface (`__call__`, what generate() drives -- a loop that synchronises every
        # step anyway) raises inside the call like the reference; the fused forms (`candidates_into`, the
        # sliced step) defer to reset() / `count` / check_errors() so that decode steps stay asynchronous.
        # sync_errors=True / False (or GENREC_AMD_EAGER_ERRORS=1 / 0) forces one behaviour for both.
        forced = _eager_errors_env() if sync_errors is None else bool(sync_errors)
        self.sync_errors = bool(forced)            # fused forms
        self.sync_errors_dense = True if forced is None else bool(forced)
        # Extension: transformers' RepetitionPenaltyLogitsProcessor folded into the same pass.
        # generate() runs that processor on log_softmax(logits) right before this one when
        # repetition_penalty != 1 (evaluate.py:218 sets 1.1); callers of the fused loop, which
        # bypasses generate(), pass the value here instead.
        if not repetition_penalty > 0:
            raise ValueError(f"`penalty` has to be a strictly positive float, but is {repetition_penalty}")
        self.repetition_penalty = float(repetition_penalty)
        self.prompt_ignore_length = int(prompt_ignore_length or 0)
        # Extension: the caller vouches that `scores` rows are log-softmax values already, as they are
        # when generate() calls this processor with nothing ahead of it in the list (the reference's own
        # log_softmax then moves values by < 1e-6).  The rows are not read: half the traffic per call.
        self.assume_normalized = bool(assume_normalized)
        if self.assume_normalized and self.repetition_penalty != 1.0:
            raise ValueError("assume_normalized cannot be combined with the fused repetition penalty")
        # Extension: `__call__` returns the SAME tensor every time (per device and shape) instead of a
        # new one -- the next call overwrites what the previous one returned.  transformers' beam loop
        # reads the result once (it adds the running scores into a new tensor, and clones what it keeps
        # for output_scores), so this is safe there; it is not for a caller that holds on to results.
        # The kernel then only moves the few finite positions instead of writing rows x vocab of -inf.
        self.reuse_output = bool(reuse_output)
        self._kept: dict[int, dict] = {}
        self._trie: Optional[FlatTrie] = None
        self._states: dict[int, _DeviceState] = {}
        self._workspaces: dict[int, torch.Tensor] = {}
        self._partial_bufs: dict[int, torch.Tensor] = {}
        self._pen_bufs: dict[int, torch.Tensor] = {}
        self._active: Optional[int] = None  # device index of the last call
        self._call_seq = 0                  # processor calls made so far (host side)
        self._planned: Optional[tuple] = None  # rows whose plans a beam kernel has already written
        self._con_key, self._con_base = None, None  # the settings the cached grx_constraint was built from

    # -- trie ---------------------------------------------------------------------
    @property
    def trie(self) -> FlatTrie:
        """CSR trie behind ``prefix_allowed_tokens_fn`` (built on first use)."""
        if self._trie is None:
            self._trie = as_flat_trie(self._prefix_allowed_tokens_fn)
        return self._trie

    # -- device plumbing --------------------------------------------------------------
    def _state(self, index: int) -> _DeviceState:
        st = self._states.get(index)
        if st is None:
            st = _DeviceState(index)
            self._states[index] = st
            self._on_new_state(st)
        return st

    def _on_new_state(self, st: _DeviceState) -> None:
        pass

    def _workspace(self, index: int, rows: int) -> torch.Tensor:
        key = index
        ws = self._workspaces.get(key)
        nbytes = int(_lib.lib().grx_workspace_bytes(rows))
        if ws is None or ws.numel() < nbytes:
            # zero-filled once; the kernels hand it back zero-filled after every call
            ws = torch.zeros(nbytes, dtype=torch.uint8, device=torch.device("cuda", index))
            self._workspaces[key] = ws
        return ws

    def _partials(self, index: int, n_floats: int) -> torch.Tensor:
        buf = self._partial_bufs.get(index)
        if buf is None or buf.numel() < n_floats:
            buf = torch.empty(n_floats, dtype=torch.float32, device=torch.device("cuda", index))
            self._partial_bufs[index] = buf
        return buf

    def _pen_logits(self, index: int, n_floats: int) -> torch.Tensor:
        buf = self._pen_bufs.get(index)
        if buf is None or buf.numel() < n_floats:
            buf = torch.empty(n_floats, dtype=torch.float32, device=torch.device("cuda", index))
            self._pen_bufs[index] = buf
        return buf

    # -- planning ahead (fused loop) -----------------------------------------------------
    def plan_request(self, device, n_rows: int):
        """A ``grx_plan_request`` for the kernel that produces the rows of the next call (beam
        init / beam step), plus the objects that must outlive the launch."""
        index = device.index if device.index is not None else torch.cuda.current_device()
        with _lib.on_device(index):
            st = self._state(index)
            ws = self._workspace(index, n_rows)
        con = self._constraint()
        req = _lib.GrxPlanRequest()
        req.trie = self.trie.device_handle(device)
        req.state = st.ptr
        req.con = C.addressof(con)
        req.workspace = ws.data_ptr()
        req.workspace_bytes = ws.numel()
        return req, (con, ws)

    def note_planned(self, rows_view: torch.Tensor) -> None:
        """The kernel given ``plan_request`` has planned exactly ``rows_view`` ([R, T] view of the
        running sequences) for the next call; anything that touches the counter cancels it."""
        self._planned = (rows_view.data_ptr(), rows_view.shape[0], rows_view.shape[1],
                         rows_view.stride(0) if rows_view.shape[0] > 1 else None, self._call_seq)

    def note_planned_rows(self, ptr: int, n_rows: int, n_tokens: int, row_stride: Optional[int]) -> None:
        """``note_planned`` from the numbers that describe the view (address, [R, T], row stride or None for one row)."""
        self._planned = (ptr, n_rows, n_tokens, row_stride, self._call_seq)

    def _take_planned(self, rows: _lib.GrxRows, input_ids: torch.Tensor) -> None:
        token = self._planned
        self._planned = None
        self._call_seq += 1
        if token is None:
            return
        ptr, R, T, stride, seq = token
        if (ptr == input_ids.data_ptr() and R == rows.R and T == rows.T and seq == self._call_seq - 1
                and (stride is None or stride == input_ids.stride(0))):
            rows.planned = 1

    def _constraint(self) -> _lib.GrxConstraint:
        """A fresh ``grx_constraint`` of the processor's current settings (callers fill in keep_top / row_bias).
        Copied from a block that is rebuilt only when one of the settings has changed."""
        pre = self.prefix_ids
        key = (self._mode, self.prefix_index, None if pre is None else tuple(pre), self.eos_token_id,
               self.repetition_penalty, self.prompt_ignore_length, self.assume_normalized)
        if key != self._con_key:
            con = _lib.GrxConstraint()
            con.mode = self._mode
            con.prefix_index = int(self.prefix_index)
            if pre is None:
                con.n_prefix = -1
            else:
                ids = [int(t) for t in pre]
                if len(ids) > _lib.GRX_MAX_PREFIX:
                    raise _lib.GenrecHipError(f"prefix_ids longer than {_lib.GRX_MAX_PREFIX} tokens is not supported")
                con.n_prefix = len(ids)
                for i, t in enumerate(ids):
                    con.prefix_ids[i] = t
            con.eos_token_id = -1 if self.eos_token_id is None else int(self.eos_token_id)
            con.repetition_penalty = self.repetition_penalty
            con.penalty_skip = self.prompt_ignore_length
            con.assume_normalized = 1 if self.assume_normalized else 0
            self._con_base, self._con_key = con, key
        return _lib.GrxConstraint.from_buffer_copy(self._con_base)

    def _rows(self, input_ids: torch.Tensor, scores: torch.Tensor, whole_groups: bool = True
              ) -> tuple[_lib.GrxRows, torch.Tensor, torch.Tensor]:
        assert input_ids.dim() == 2, "input_ids must be a 2D tensor"
        _lib.require_cuda(scores, "scores")
        _lib.require_cuda(input_ids, "input_ids")
        if scores.dim() != 2 or scores.shape[0] != input_ids.shape[0]:
            raise ValueError(f"scores must be [rows, vocab] with rows == {input_ids.shape[0]}; got {tuple(scores.shape)}")
        if whole_groups and input_ids.shape[0] % self._num_beams != 0:
            # the reference fails on view(-1, num_beams, seq_len) (logit_processor.py:36)
            raise RuntimeError(
                f"shape '[-1, {self._num_beams}, {input_ids.shape[1]}]' is invalid for input of size {input_ids.numel()}"
            )
        if input_ids.shape[1] == 0:
            # the reference's view(-1, num_beams, 0) cannot infer the group dimension (logit_processor.py:36)
            raise RuntimeError(
                f"cannot reshape tensor of 0 elements into shape [-1, {self._num_beams}, 0] because the "
                "unspecified dimension size -1 can be any value and is ambiguous"
            )
        if input_ids.dtype != torch.int64:
            input_ids = input_ids.to(torch.int64)
        if input_ids.shape[1] > 0 and input_ids.stride(1) != 1:
            input_ids = input_ids.contiguous()
        if scores.stride(1) != 1:
            scores = scores.contiguous()
        rows = _lib.GrxRows()
        rows.input_ids = input_ids.data_ptr()
        rows.ids_stride = input_ids.stride(0) if input_ids.shape[0] > 1 else max(input_ids.shape[1], 1)
        rows.R = input_ids.shape[0]
        rows.T = input_ids.shape[1]
        rows.scores = scores.data_ptr()
        rows.scores_stride = scores.stride(0) if scores.shape[0] > 1 else scores.shape[1]
        rows.V = scores.shape[1]
        rows.dtype = _lib.torch_dtype_code(scores.dtype)
        return rows, input_ids, scores

    def _device_index(self, t: torch.Tensor) -> int:
        return t.device.index if t.device.index is not None else torch.cuda.current_device()

    # -- the processor surface -----------------------------------------------------------
    def __call__(self, input_ids: torch.LongTensor, scores: torch.FloatTensor) -> torch.FloatTensor:
        """``log_softmax(scores) + mask`` as a new [rows, vocab] tensor of the dtype of ``scores``
        (logit_processor.py:29-49 / fixed_hint_logit_processor.py:79-96).  The signature stays
        exactly (input_ids, scores): ``LogitsProcessorList`` inspects it and demands keyword
        arguments for anything more."""
        return self.call_into(input_ids, scores, None)

    def call_into(self, input_ids: torch.LongTensor, scores: torch.FloatTensor,
                  out: Optional[torch.Tensor]) -> torch.FloatTensor:
        """``__call__`` writing into ``out`` (extension): a caller-owned result tensor, fp32 or the
        dtype of ``scores``; rows may be padded (``out.stride(0) >= vocab``).  Saves the allocation
        and lets bf16 logits produce fp32 log-probs."""
        rows, ids_keep, scores_keep = self._rows(input_ids, scores)
        index = self._device_index(scores_keep)
        kept = None
        if out is None and self.reuse_output and rows.R > 0:
            key = (tuple(scores_keep.shape), scores_keep.dtype, self.eos_token_id)
            kept = self._kept.get(index)
            if kept is None or kept["key"] != key:
                self._kept.pop(index, None)  # release the old tensor before allocating the new one
                kept = {"key": key, "fresh": 1,
                        "out": torch.empty_like(scores_keep, memory_format=torch.contiguous_format),
                        "last": torch.zeros(int(_lib.lib().grx_reuse_bytes(rows.R)), dtype=torch.uint8,
                                            device=scores_keep.device)}
                self._kept[index] = kept
            out = kept["out"]
        elif out is None:
            out = torch.empty_like(scores_keep, memory_format=torch.contiguous_format)
        else:
            _lib.require_cuda(out, "out")
            if out.shape != scores_keep.shape or out.device != scores_keep.device or out.stride(1) != 1:
                raise ValueError("out must be a [rows, vocab] tensor on the device of scores with unit column stride")
            if out.dtype not in (torch.float32, scores_keep.dtype):
                raise ValueError("out must be float32 or have the dtype of scores")
        self._take_planned(rows, ids_keep)
        with _lib.on_device(index):
            st = self._state(index)
            ws = self._workspace(index, rows.R)
            con = self._constraint()
            if kept is not None:
                _lib.check(
                    _lib.lib().grx_constrained_logsoftmax_reuse(
                        self.trie.device_handle(scores_keep.device), st.ptr, C.byref(rows), C.byref(con),
                        out.data_ptr(), out.shape[1], _lib.torch_dtype_code(out.dtype), kept["last"].data_ptr(),
                        kept["fresh"], ws.data_ptr(), ws.numel(), _lib.current_stream_ptr(scores_keep.device),
                    )
                )
                kept["fresh"] = 0  # enqueued: from here on the tensor and its record belong together
            else:
                _lib.check(
                    _lib.lib().grx_constrained_logsoftmax(
                        self.trie.device_handle(scores_keep.device),
                        st.ptr,
                        C.byref(rows),
                        C.byref(con),
                        out.data_ptr(),
                        out.stride(0) if out.shape[0] > 1 else out.shape[1],
                        _lib.torch_dtype_code(out.dtype),
                        ws.data_ptr(),
                        ws.numel(),
                        _lib.current_stream_ptr(scores_keep.device),
                    )
                )
        self._active = index
        if self.sync_errors_dense:
            self.check_errors()
        return out

    def candidates(self, input_ids: torch.LongTensor, scores: torch.Tensor, cmax: Optional[int] = None
                   ) -> "Candidates":
        """Fused form of ``__call__``: the allowed next tokens of every row and their
        full-vocabulary log-probabilities (fp32) -- the finite entries of what
        ``__call__`` would return -- as a :class:`Candidates` buffer.  Advances
        ``count`` exactly like ``__call__``."""
        if cmax is None:
            cmax = max(self.trie.max_fanout, 1)
        cand = Candidates(input_ids.shape[0], cmax, scores.device)
        self.candidates_into(input_ids, scores, cand)
        return cand

    KEEP_TOP_MAX, KEEP_TOP_MAX_FANOUT = 512, 1024  # limits of the kernel's per-row selection
    KEEP_TOP_SLACK = 64                              # rows up to keep_top + this many entries are emitted whole

    def keep_top_for(self, n_keep: int) -> int:
        """The ``keep_top`` to pass for a beam step that keeps ``n_keep`` candidates per group: ``n_keep``
        when the kernel can rank this trie's rows (0 = no truncation otherwise)."""
        ok = 0 < n_keep <= self.KEEP_TOP_MAX and self.trie.max_fanout <= self.KEEP_TOP_MAX_FANOUT
        return int(n_keep) if ok else 0

    def candidate_width(self, n_keep: int = 0) -> int:
        """Entries a candidate row needs: the widest node of the trie, or ``n_keep`` if rows are cut there."""
        width = max(self.trie.max_fanout, 1)
        keep = self.keep_top_for(n_keep)
        return min(width, keep + self.KEEP_TOP_SLACK) if keep else width

    def candidates_into(self, input_ids, scores, cand: "Candidates", one_row_per_group: bool = False,
                        keep_top: int = 0, row_bias: Optional[torch.Tensor] = None, beam_step=None,
                        shared_prefix: Optional[tuple] = None) -> None:
        """``candidates`` writing into a caller-owned buffer (no allocation per step).

        ``keep_top`` > 0: rows with more than ``keep_top + KEEP_TOP_SLACK`` allowed tokens emit only their
        ``keep_top`` best (in no particular order), ranked by ``logp + row_bias[row]`` (fp32 [rows]; the running beam scores) descending, token
        ascending -- the order the beam step ranks by, which never takes more than its ``n_keep`` from one
        row: with ``keep_top = n_keep`` the search is unchanged and the candidate traffic shrinks.

        ``one_row_per_group``: the rows are prompts, one per beam group, instead of ``num_beams``
        identical copies of each (the first step of a search: the copies would give identical
        results, see ``BeamSearchState.step(..., shared_rows=True)``).  The shared counter moves
        exactly as for the full batch.

        ``shared_prefix`` ``(n_tokens, group)``: the caller's promise that the first ``n_tokens`` tokens of the rows
        are equal within every group of ``group`` consecutive rows (the prompt part of beam rows; see
        ``BeamSearchState.shared_prefix``).  Only the fused repetition penalty uses it: it reads those tokens from
        a group's first row, which the group's other rows then find in the cache.  Same results.

        ``beam_step`` (a filled ``GrxBeamStepArgs`` that reads ``cand``; see ``BeamSearchState.fused_step``):
        the beam step that consumes the candidates is carried out by this same library call
        (``grx_constrained_beam_step``) -- inside the streaming kernel where the library can."""
        rows, ids_keep, scores_keep = self._rows(input_ids, scores, whole_groups=not one_row_per_group)
        index = self._device_index(scores_keep)
        assert cand.rows == rows.R and cand.buf.device == scores_keep.device
        if row_bias is not None:
            _lib.require_cuda(row_bias, "row_bias")
            if row_bias.dtype != torch.float32 or row_bias.numel() != rows.R or not row_bias.is_contiguous():
                raise ValueError("row_bias must be a contiguous float32 tensor with one value per row")
        self._take_planned(rows, ids_keep)
        with _lib.on_device(index):
            st = self._state(index)
            ws = self._workspace(index, rows.R)
            con = self._constraint()
            con.keep_top = int(keep_top)
            con.row_bias = row_bias.data_ptr() if (row_bias is not None and keep_top) else None
            if shared_prefix is not None and self.repetition_penalty != 1.0:
                con.shared_prefix_len, con.shared_group = min(int(shared_prefix[0]), rows.T), int(shared_prefix[1])
                if self.sync_errors and con.shared_prefix_len > 0 and rows.R % max(con.shared_group, 1) == 0:
                    # the checking mode synchronises anyway: hold the caller to its promise
                    head = ids_keep[:, : con.shared_prefix_len].view(-1, con.shared_group, con.shared_prefix_len)
                    if not bool((head == head[:, :1]).all()):
                        raise ValueError("shared_prefix: the rows of a group do not share their first "
                                         f"{con.shared_prefix_len} tokens")
            if beam_step is not None:
                _lib.check(
                    _lib.lib().grx_constrained_beam_step(
                        self.trie.device_handle(scores_keep.device), st.ptr, C.byref(rows), C.byref(con),
                        cand.buf.data_ptr(), cand.stride, cand.cmax, ws.data_ptr(), ws.numel(), C.byref(beam_step),
                        _lib.current_stream_ptr(index),
                    )
                )
            else:
                _lib.check(
                    _lib.lib().grx_constrained_candidates(
                        self.trie.device_handle(scores_keep.device),
                        st.ptr,
                        C.byref(rows),
                        C.byref(con),
                        cand.buf.data_ptr(),
                        cand.stride,
                        cand.cmax,
                        ws.data_ptr(),
                        ws.numel(),
                        _lib.current_stream_ptr(scores_keep.device),
                    )
                )
        self._active = index
        if self.sync_errors:
            self.check_errors()

    def lm_head_candidates_into(self, input_ids, hidden: torch.Tensor, weight: torch.Tensor, cand: "Candidates",
                                one_row_per_group: bool = False) -> None:
        """``candidates_into`` for logits that are never written out: ``hidden`` [rows, H] and ``weight``
        [vocab, H] (the output projection's ``nn.Linear.weight``; both bf16) are multiplied tile by tile
        on the matrix cores inside the call (``grx_lm_head_candidates``), and only per-tile (max, sum-of-exp)
        pairs plus the logits of the vocabulary's top columns -- where the trie's tokens live -- leave the
        multiplication.  Same candidate rows as ``candidates_into(input_ids, (hidden @ weight.T).float(), cand)``
        up to fp32 summation order.  No repetition penalty, ``assume_normalized`` or ``keep_top`` in this form."""
        _lib.require_cuda(hidden, "hidden")
        _lib.require_cuda(weight, "weight")
        if hidden.dim() != 2 or weight.dim() != 2 or hidden.shape[1] != weight.shape[1]:
            raise ValueError("hidden must be [rows, H] and weight [vocab, H]")
        if hidden.dtype != torch.bfloat16 or weight.dtype != torch.bfloat16:
            raise TypeError("the fused LM head takes bfloat16 hidden states and weights")
        if hidden.stride(1) != 1:
            hidden = hidden.contiguous()
        if weight.stride(1) != 1:
            weight = weight.contiguous()
        if input_ids.shape[0] != hidden.shape[0]:
            raise ValueError(f"hidden must have one row per sequence ({input_ids.shape[0]}); got {hidden.shape[0]}")
        assert input_ids.dim() == 2, "input_ids must be a 2D tensor"
        _lib.require_cuda(input_ids, "input_ids")
        if not one_row_per_group and input_ids.shape[0] % self._num_beams != 0:
            raise RuntimeError(
                f"shape '[-1, {self._num_beams}, {input_ids.shape[1]}]' is invalid for input of size {input_ids.numel()}")
        if input_ids.dtype != torch.int64:
            input_ids = input_ids.to(torch.int64)
        if input_ids.shape[1] > 0 and input_ids.stride(1) != 1:
            input_ids = input_ids.contiguous()
        rows = _lib.GrxRows()
        rows.input_ids = input_ids.data_ptr()
        rows.ids_stride = input_ids.stride(0) if input_ids.shape[0] > 1 else max(input_ids.shape[1], 1)
        rows.R, rows.T = input_ids.shape[0], input_ids.shape[1]
        rows.V = weight.shape[0]
        rows.scores_stride = weight.shape[0]
        rows.dtype = _lib.GRX_BF16
        index = self._device_index(hidden)
        assert cand.rows == rows.R and cand.buf.device == hidden.device
        head = _lib.GrxLmHead()
        head.hidden, head.hidden_stride = hidden.data_ptr(), hidden.stride(0) if hidden.shape[0] > 1 else hidden.shape[1]
        head.weight, head.weight_stride = weight.data_ptr(), weight.stride(0)
        head.H = hidden.shape[1]
        self._take_planned(rows, input_ids)
        with _lib.on_device(index):
            st = self._state(index)
            ws = self._workspace(index, rows.R)
            con = self._constraint()
            handle = self.trie.device_handle(hidden.device)
            need = int(_lib.lib().grx_lm_head_scratch_bytes(handle, rows.R, rows.V))
            scratch = self._partials(index, (need + 3) // 4)
            _lib.check(
                _lib.lib().grx_lm_head_candidates(
                    handle, st.ptr, C.byref(rows), C.byref(con), C.byref(head), scratch.data_ptr(), scratch.numel() * 4,
                    cand.buf.data_ptr(), cand.stride, cand.cmax, ws.data_ptr(), ws.numel(),
                    _lib.current_stream_ptr(hidden.device),
                )
            )
        self._active = index
        if self.sync_errors:
            self.check_errors()

    def chunked_step(self, input_ids, cand: "Candidates", v_total: int, bounds, dtype,
                     one_row_per_group: bool = False, parts: Optional[int] = None) -> "ChunkedStep":
        """``candidates_into`` for logits that arrive as column slices of the vocabulary (the LM head
        run slice by slice, so that the [rows, vocab] tensor never exists): returns a
        :class:`ChunkedStep`; ``feed`` it every slice, then ``finish``.  ``bounds`` lists the
        ``(v_begin, v_end)`` column ranges that will be fed; together they must cover
        ``[0, v_total)`` exactly once.  Advances ``count`` exactly like one ``candidates_into``."""
        return ChunkedStep(self, input_ids, cand, v_total, bounds, dtype, one_row_per_group, parts)

    # -- errors ----------------------------------------------------------------------------
    def _read_state(self, index: int) -> tuple[int, int, int, int]:
        st = self._states.get(index)
        if st is None:
            return (0, 0, -1, 0)
        buf = (C.c_int32 * 4)()
        with torch.cuda.device(index):
            _lib.check(
                _lib.lib().grx_state_read(st.ptr, C.byref(buf), _lib.current_stream_ptr(torch.device("cuda", index)))
            )
        return (buf[0], buf[1], buf[2], buf[3])

    def check_errors(self) -> None:
        """Raise what the reference would have raised inside ``__call__``.

        The kernels record the first (lowest-row) failure of a call on the
        device; reading it back synchronises the stream.  Called after every
        ``__call__`` (unless ``sync_errors=False`` / ``GENREC_AMD_EAGER_ERRORS=0``); the fused forms
        leave it to ``reset()``, reading ``count``, or an explicit call.
        """
        for index in list(self._states):
            _, code, row, _ = self._read_state(index)
            if code == _lib.GRX_ROWERR_NONE:
                continue
            with torch.cuda.device(index):
                _lib.check(
                    _lib.lib().grx_state_clear_error(
                        self._states[index].ptr, _lib.current_stream_ptr(torch.device("cuda", index))
                    )
                )
            if code == _lib.GRX_ROWERR_EMPTY:
                raise AssertionError(_EMPTY_MSG)
            if code == _lib.GRX_ROWERR_NOMATCH:
                raise ValueError(f"{_NOMATCH_MSG} prefix_ids={self.prefix_ids}, row={row}")
            raise _lib.GenrecHipError(f"row {row}: allowed-token set larger than the candidate buffer")


class ConstrainedLogitsProcessor(_TrieConstrainedBase):
    """Count-based window processor (reference: logit_processor.py:8-49).

    ``count`` is one counter shared by all rows: row r's window is its last
    ``prefix_index + count`` tokens, ``count`` advances once per call, and when
    ``prefix_ids`` is given a row ending in exactly those tokens resets it
    (logit_processor.py:39-41,48).  ``initial_count`` reproduces the seeded
    variant whose counter starts at, and resets to, the number of hint tokens
    already in the prompt (analyze_rl_beam_hint.py:553-560).
    """

    _mode = _lib.GRX_MODE_COUNT

    def __init__(
        self,
        prefix_allowed_tokens_fn,
        num_beams: int,
        prefix_index: int = 3,
        prefix_ids: Optional[list[int]] = None,
        eos_token_id: Optional[int] = None,
        *,
        initial_count: int = 0,
        sync_errors: Optional[bool] = None,
        repetition_penalty: float = 1.0,
        prompt_ignore_length: int = 0,
        assume_normalized: bool = False,
        reuse_output: bool = False,
    ):
        super().__init__(
            prefix_allowed_tokens_fn, num_beams, prefix_index, prefix_ids, eos_token_id, sync_errors=sync_errors,
            repetition_penalty=repetition_penalty, prompt_ignore_length=prompt_ignore_length,
            assume_normalized=assume_normalized, reuse_output=reuse_output,
        )
        self.initial_count = int(initial_count)
        self._count_host = self.initial_count  # value to push to a device on first use / after a set

    def _on_new_state(self, st: _DeviceState) -> None:
        self._push_count(st, self._count_host)

    def _push_count(self, st: _DeviceState, value: int) -> None:
        with torch.cuda.device(st.device_index):
            _lib.check(
                _lib.lib().grx_state_set_count(
                    st.ptr, int(value), _lib.current_stream_ptr(torch.device("cuda", st.device_index))
                )
            )

    @property
    def count(self) -> int:
        """The shared step counter.  Reading it synchronises with the device."""
        if self._active is None or self._active not in self._states:
            return self._count_host
        self.check_errors()
        return self._read_state(self._active)[0]

    @count.setter
    def count(self, value: int) -> None:
        self._planned = None  # plans made with the old counter are void
        self._count_host = int(value)
        for st in self._states.values():
            self._push_count(st, self._count_host)

    def reset(self) -> None:
        if self._states:
            self.check_errors()
        self.count = self.initial_count
