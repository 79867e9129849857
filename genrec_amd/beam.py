"""Beam search over compact trie candidates on MI355X.

The reference lets ``transformers`` run beam search (``model.generate(num_beams=K)``
from evaluate.py:363-371, MIMIGenRec.py:79, analyze_rl_beam_hint.py:609-623), which
adds the running scores to a dense [R, V] tensor and takes ``topk`` over
[B, K*V].  Here the processor emits only the allowed tokens of each row, and
one kernel (``grx_beam_step``) performs the whole iteration for a prompt group:
accumulate, top-2K, parent/token split, stopping criteria, running top-K,
finished-beam merge, sequence / parent gathers and the early-stop heuristic --
the arithmetic of ``GenerationMixin._beam_search`` (transformers 5.15.0,
generation/utils.py:3418-3502), with ties ordered by (score desc, beam*V+token asc).
"""

from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Callable, Optional

import torch

from . import _lib

NEG = -1.0e9


@dataclass
class BeamStepResult:
    topk_scores: torch.Tensor   # [B, n_keep] fp32
    topk_parent: torch.Tensor   # [B, n_keep] int32, beam inside the group
    topk_token: torch.Tensor    # [B, n_keep] int32
    topk_hit: torch.Tensor      # [B, n_keep] uint8
    next_parent: torch.Tensor   # [B*K] int32 flat KV-cache reorder index


class BeamSearchState:
    """Device-resident beam state, laid out like the loop variables of HF's
    ``_beam_search`` (utils.py:3316-3351) and double-buffered so that a step
    reads one copy and writes the other."""

    def __init__(self, input_ids: torch.Tensor, num_beams: int, max_length: int, eos_token_id: Optional[int],
                 pad_token_id: Optional[int] = None, length_penalty: float = 1.0, early_stopping=False,
                 shared_prompt: bool = True, planner=None, first_step_shared: bool = False):
        _lib.require_cuda(input_ids, "input_ids")
        R, cur_len = input_ids.shape
        if R % num_beams:
            raise ValueError(f"{R} rows do not split into groups of {num_beams} beams")
        if max_length <= cur_len:
            raise ValueError("max_length must exceed the prompt length")
        dev = input_ids.device
        self.device = dev
        self.K = num_beams
        self.B = R // num_beams
        self.max_length = int(max_length)
        self.eos = eos_token_id
        self.length_penalty = float(length_penalty)
        self.early_stopping = early_stopping
        self.prompt_len = cur_len
        self.cur_len = cur_len
        self.n_keep = max(2, 1 + (1 if eos_token_id is not None else 0)) * num_beams
        # when every beam of a group starts from the same prompt, re-ordering beams never changes
        # positions < prompt_len, so the kernel only moves the generated suffix
        self.copy_from = cur_len if shared_prompt else 0
        # transformers' own expression (generation/utils.py:3319 of 5.15.0): `pad_token_id or eos_token_id[0]`,
        # under which a pad token id of 0 counts as absent and finished rows trail EOS.  Kept as is: the
        # contract is generate()'s output, and its recorded traces (pad 0, eos 2) do trail EOS.
        self.fill = (pad_token_id or eos_token_id) if eos_token_id is not None else -1
        B, K, L = self.B, self.K, self.max_length
        lg = L - cur_len

        def pair(shape, dtype):
            return [torch.empty(shape, dtype=dtype, device=dev) for _ in range(2)]

        self.run_seq = pair((B, K, L), torch.int64)
        self.fin_seq = pair((B, K, L), torch.int64)
        self.run_score = pair((B, K), torch.float32)
        self.fin_score = pair((B, K), torch.float32)
        self.run_bidx = pair((B, K, lg), torch.int32)
        self.fin_bidx = pair((B, K, lg), torch.int32)
        self.fin_done = pair((B, K), torch.uint8)
        self.unsat = torch.empty(B, dtype=torch.uint8, device=dev)
        self._step_out = None
        self._arg_blocks = [None, None]  # grx_beam_step_args per buffer half (see _step_args)
        self.reset(input_ids, planner, first_step_shared)

    def prompt_rows(self) -> torch.Tensor:
        """[B, cur_len] view of the first beam of every group: the rows of a first processor call that
        sees one row per group (``candidates_into(..., one_row_per_group=True)``)."""
        return self.flat_running_sequences()[:: self.K]

    def reset(self, input_ids: torch.Tensor, planner=None, first_step_shared: bool = False) -> None:
        """Start a new search from ``input_ids`` [B*K, prompt_len] (same shape as at
        construction) in the existing buffers: one kernel launch, no allocation.

        ``planner`` (a genrec_amd processor): also work out the trie node of every prompt row for
        the processor's next call, which then skips its own planning kernel.  The processor's
        counter must already be what that call should see (reset it first).  ``first_step_shared``:
        that call will be made on ``prompt_rows()`` (one row per group), so only those are planned."""
        _lib.require_cuda(input_ids, "input_ids")
        if tuple(input_ids.shape) != (self.B * self.K, self.prompt_len):
            raise ValueError("reset() needs input_ids of the shape the state was built for")
        if input_ids.dtype != torch.int64:
            input_ids = input_ids.to(torch.int64)
        if input_ids.stride(1) != 1:
            input_ids = input_ids.contiguous()
        a = _lib.GrxBeamInitArgs()
        a.B, a.K, a.prompt_len, a.max_length = self.B, self.K, self.prompt_len, self.max_length
        a.fill_token = int(self.fill)
        a.input_ids = input_ids.data_ptr()
        a.ids_stride = input_ids.stride(0) if input_ids.shape[0] > 1 else max(self.prompt_len, 1)
        for h in range(2):
            a.run_seq[h] = self.run_seq[h].data_ptr()
            a.fin_seq[h] = self.fin_seq[h].data_ptr()
            a.run_score[h] = self.run_score[h].data_ptr()
            a.fin_score[h] = self.fin_score[h].data_ptr()
            a.run_bidx[h] = self.run_bidx[h].data_ptr()
            a.fin_bidx[h] = self.fin_bidx[h].data_ptr()
            a.fin_done[h] = self.fin_done[h].data_ptr()
        a.unsat = self.unsat.data_ptr()
        keep = None
        if planner is not None:
            req, keep = planner.plan_request(self.device, self.B if first_step_shared else self.B * self.K)
            a.plan_next = C.addressof(req)
            a.plan_one_per_group = 1 if first_step_shared else 0
        index = self.device.index if self.device.index is not None else torch.cuda.current_device()
        with _lib.on_device(index):
            _lib.check(_lib.lib().grx_beam_init(C.byref(a), _lib.current_stream_ptr(index)))
        del keep
        self.cur = 0  # index of the buffer that holds the current state
        self.cur_len = self.prompt_len
        if planner is not None:
            # the plans describe the caller's prompt rows; the running-sequence view holds the same tokens
            planner.note_planned(self.prompt_rows() if first_step_shared else self.flat_running_sequences())

    # -- views of the current state -------------------------------------------------
    @property
    def running_sequences(self) -> torch.Tensor:
        return self.run_seq[self.cur]

    @property
    def running_beam_scores(self) -> torch.Tensor:
        return self.run_score[self.cur]

    @property
    def sequences(self) -> torch.Tensor:
        return self.fin_seq[self.cur]

    @property
    def beam_scores(self) -> torch.Tensor:
        return self.fin_score[self.cur]

    @property
    def beam_indices(self) -> torch.Tensor:
        return self.fin_bidx[self.cur]

    @property
    def is_sent_finished(self) -> torch.Tensor:
        return self.fin_done[self.cur].bool()

    def shared_prefix(self) -> Optional[tuple]:
        """``(n_tokens, K)`` when the K rows of every group started from one prompt (``shared_prompt``): beam search
        only ever re-orders whole rows inside a group, so their first ``n_tokens`` tokens stay equal.  For
        ``candidates_into(..., shared_prefix=...)``."""
        return (self.copy_from, self.K) if self.copy_from > 0 else None

    def row_bias(self) -> torch.Tensor:
        """[B*K] running beam scores: what the next step adds to a row's log-probs before ranking
        (``candidates_into(..., keep_top=state.n_keep, row_bias=state.row_bias())``)."""
        return self.run_score[self.cur].view(-1)

    def flat_running_sequences(self) -> torch.Tensor:
        """[B*K, cur_len] view (row stride max_length) to hand to the processor."""
        return self.run_seq[self.cur].view(self.B * self.K, self.max_length)[:, : self.cur_len]

    def _outputs(self) -> BeamStepResult:
        if self._step_out is None:
            B, n, dev = self.B, self.n_keep, self.device
            self._step_out = BeamStepResult(
                torch.empty((B, n), dtype=torch.float32, device=dev),
                torch.empty((B, n), dtype=torch.int32, device=dev),
                torch.empty((B, n), dtype=torch.int32, device=dev),
                torch.empty((B, n), dtype=torch.uint8, device=dev),
                torch.empty(B * self.K, dtype=torch.int32, device=dev),
            )
        return self._step_out

    def _step_args(self, cand, vocab_size: int, planner, shared_rows: bool):
        """The ``grx_beam_step_args`` of the next step (reads the current buffers, writes the others),
        the objects that must outlive the launch, and whether the kernel plans the next processor call.

        The block of each buffer half is built once and kept: the state's buffers live as long as the
        state, the library reads the block during the call only, and a step changes a handful of fields."""
        i, o = self.cur, 1 - self.cur
        a = self._arg_blocks[i]
        if a is None:
            a = _lib.GrxBeamStepArgs()
            out = self._outputs()
            a.B, a.K, a.n_keep = self.B, self.K, self.n_keep
            a.prompt_len, a.max_length = self.prompt_len, self.max_length
            a.copy_from = self.copy_from
            a.run_seq_in, a.run_seq_out = self.run_seq[i].data_ptr(), self.run_seq[o].data_ptr()
            a.run_score_in, a.run_score_out = self.run_score[i].data_ptr(), self.run_score[o].data_ptr()
            a.run_bidx_in, a.run_bidx_out = self.run_bidx[i].data_ptr(), self.run_bidx[o].data_ptr()
            a.fin_seq_in, a.fin_seq_out = self.fin_seq[i].data_ptr(), self.fin_seq[o].data_ptr()
            a.fin_score_in, a.fin_score_out = self.fin_score[i].data_ptr(), self.fin_score[o].data_ptr()
            a.fin_bidx_in, a.fin_bidx_out = self.fin_bidx[i].data_ptr(), self.fin_bidx[o].data_ptr()
            a.fin_done_in, a.fin_done_out = self.fin_done[i].data_ptr(), self.fin_done[o].data_ptr()
            a.unsat = self.unsat.data_ptr()
            a.topk_score, a.topk_parent = out.topk_scores.data_ptr(), out.topk_parent.data_ptr()
            a.topk_token, a.topk_hit = out.topk_token.data_ptr(), out.topk_hit.data_ptr()
            a.next_parent = out.next_parent.data_ptr()
            self._arg_blocks[i] = a
        a.V = int(vocab_size)
        a.cmax = cand.cmax
        a.cur_len = self.cur_len
        # plain attributes of the state: read every step, so that changing one between steps is honoured
        a.eos_token_id = -1 if self.eos is None else int(self.eos)
        a.early_stopping = 1 if self.early_stopping is True else (2 if self.early_stopping == "never" else 0)
        a.length_penalty = self.length_penalty
        assert cand.rows == (self.B if shared_rows else self.B * self.K)
        a.cand, a.cand_stride = cand.buf.data_ptr(), cand.stride
        a.cand_shared = 1 if shared_rows else 0
        keep = None
        plan = planner is not None and self.cur_len + 1 < self.max_length
        if plan:
            req, keep = planner.plan_request(self.device, self.B * self.K)
            a.plan_next = C.addressof(req)
            keep = (req, keep)
        else:
            a.plan_next = None
        return a, keep, plan

    def _advance(self, planner, plan: bool) -> BeamStepResult:
        self.cur = 1 - self.cur
        self.cur_len += 1
        if plan:
            # what flat_running_sequences() will be at the next call, without making the view
            R = self.B * self.K
            planner.note_planned_rows(self.run_seq[self.cur].data_ptr(), R, self.cur_len,
                                      self.max_length if R > 1 else None)
        return self._outputs()

    def step(self, cand, vocab_size: int, planner=None, shared_rows: bool = False) -> BeamStepResult:
        """Advance by one token from the processor's compact candidates (a ``Candidates`` buffer).
        ``planner`` (the processor): the kernel also plans the processor's next call on the new
        running sequences, so that call starts streaming at once.

        ``shared_rows``: ``cand`` holds ONE row per group which all K beams extend from.  Valid
        whenever the K rows of every group are equal, i.e. at the first step of a search whose beams
        are copies of one prompt: the results are those of K equal candidate rows, and the
        processor call that made ``cand`` read 1/K of the logits."""
        a, keep, plan = self._step_args(cand, vocab_size, planner, shared_rows)
        index = self.device.index if self.device.index is not None else torch.cuda.current_device()
        with _lib.on_device(index):
            _lib.check(_lib.lib().grx_beam_step(C.byref(a), _lib.current_stream_ptr(index)))
        del keep
        return self._advance(planner, plan)

    def fused_step(self, processor, scores: torch.Tensor, cand, keep_top: int = 0, planner=None) -> BeamStepResult:
        """``processor.candidates_into(flat_running_sequences(), scores, cand, keep_top, row_bias())``
        followed by ``step(cand, vocab, planner)`` as ONE library call (``grx_constrained_beam_step``):
        where the library can, a prompt group's beam step runs inside the streaming kernel as soon as the
        group's rows are done.  Same results as the two calls."""
        a, keep, plan = self._step_args(cand, scores.shape[-1], planner, False)
        processor.candidates_into(self.flat_running_sequences(), scores, cand, keep_top=keep_top,
                                  row_bias=self.row_bias() if keep_top else None, beam_step=a,
                                  shared_prefix=self.shared_prefix())
        del keep
        return self._advance(planner, plan)

    def has_unfinished(self) -> bool:
        """HF's loop condition (utils.py:3055-3075).  Synchronises."""
        out = self._outputs()
        improvement_possible = bool(self.unsat.any())
        exists_open_beam = not (bool(self.fin_done[self.cur].all()) and self.early_stopping is True)
        valid_continuations = not bool(out.topk_hit.all())
        return improvement_possible and exists_open_beam and valid_continuations

    def finalize(self, num_return_sequences: Optional[int] = None):
        """(sequences, sequences_scores, beam_indices) cropped as HF does (utils.py:3512-3523)."""
        n = self.K if num_return_sequences is None else num_return_sequences
        seqs = self.fin_seq[self.cur][:, :n, :].reshape(self.B * n, -1)
        scores = self.fin_score[self.cur][:, :n].reshape(-1)
        bidx = self.fin_bidx[self.cur][:, :n, :].reshape(self.B * n, -1)
        gen_len = int(((bidx + 1).bool()).sum(dim=1).max())
        return seqs[:, : self.prompt_len + gen_len], scores, bidx[:, :gen_len]


def constrained_beam_search(
    logits_fn: Callable[[int, torch.Tensor, Optional[torch.Tensor]], torch.Tensor],
    processor,
    input_ids: torch.Tensor,
    num_beams: int,
    max_new_tokens: int,
    eos_token_id: Optional[int],
    pad_token_id: Optional[int] = None,
    length_penalty: float = 1.0,
    early_stopping=False,
    num_return_sequences: Optional[int] = None,
    max_steps: Optional[int] = None,
    first_step_shared: bool = False,
):
    """Trie-constrained beam search with the fused step.

    ``input_ids`` is [B, T] (one row per prompt).  ``logits_fn(step, flat_sequences,
    next_parent)`` returns the raw next-token logits [B*K, V] for the running
    beams (``next_parent`` is the flat KV-cache reorder index of the previous
    step, ``None`` at step 0).  ``processor`` is a genrec_amd processor; its
    ``candidates_into`` replaces ``log_softmax`` + mask + dense top-k.

    ``max_steps`` fixes the number of iterations (a trie of L levels always
    takes L + 1) and avoids the per-step host check of the loop condition.
    ``first_step_shared``: at step 0 ``logits_fn`` is handed the B prompt rows and
    returns [B, V] (one row per group; the K copies would be identical).
    Returns ``(state, sequences, sequences_scores, beam_indices)``.
    """
    prompts = input_ids.repeat_interleave(num_beams, dim=0)
    if hasattr(processor, "reset"):
        processor.reset()  # before the state is built: its initialisation plans the first call
    state = BeamSearchState(prompts, num_beams, input_ids.shape[1] + max_new_tokens, eos_token_id, pad_token_id,
                            length_penalty, early_stopping, shared_prompt=True, planner=processor,
                            first_step_shared=first_step_shared)
    from .logit_processor import Candidates

    # a group keeps n_keep candidates, so no row needs to hand over more than its n_keep best
    keep = processor.keep_top_for(state.n_keep)
    cmax = processor.candidate_width(state.n_keep)
    cand = Candidates(state.B * num_beams, cmax, input_ids.device)
    step = 0
    next_parent = None
    while True:
        if step == 0 and first_step_shared:
            flat = state.prompt_rows()
            logits = logits_fn(step, flat, next_parent)
            first = Candidates(state.B, cmax, input_ids.device)
            processor.candidates_into(flat, logits, first, one_row_per_group=True, keep_top=keep)
            res = state.step(first, logits.shape[-1], planner=processor, shared_rows=True)
        else:
            flat = state.flat_running_sequences()
            logits = logits_fn(step, flat, next_parent)
            processor.candidates_into(flat, logits, cand, keep_top=keep, row_bias=state.row_bias(),
                                      shared_prefix=state.shared_prefix())
            res = state.step(cand, logits.shape[-1], planner=processor)
        next_parent = res.next_parent
        step += 1
        if max_steps is not None:
            if step >= max_steps:
                break
        elif not state.has_unfinished():
            break
    seqs, scores, bidx = state.finalize(num_return_sequences)
    return state, seqs, scores, bidx
