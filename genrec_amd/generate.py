"""Trie-constrained beam search for a Hugging Face causal LM with the fused decoding step.

The reference decodes with ``model.generate(num_beams=K, logits_processor=[processor])``
(evaluate.py:363-371, MIMIGenRec.py:79, hint_sft_trainer.py:355-356), i.e. transformers'
``_beam_search``: every step it normalises the [B*K, V] logits, runs the processors, adds the
running scores, takes a dense top-2K and re-orders the KV cache.  ``constrained_generate`` keeps
the model side of that loop (forward pass, KV cache, attention mask, position ids) and replaces
everything behind the logits by two kernels per step (masked log-softmax to compact candidates,
beam step).  Outputs mirror ``generate(..., return_dict_in_generate=True)``: ``sequences``,
``sequences_scores`` and ``beam_indices`` of the ``num_return_sequences`` best beams per prompt.

It also prefills each prompt ONCE and repeats the cache for the K beams, where
``generate()`` prefills K identical copies of every prompt (generation/utils.py
``_expand_inputs_for_generation``).
"""

from __future__ import annotations

from dataclasses import dataclass
from typing import Optional

import torch

from .beam import BeamSearchState
from .logit_processor import Candidates


class SlicedLMHead:
    """The LM head run over column slices of the vocabulary, each slice handed to the processor's
    sliced step as soon as it exists, so that the [rows, vocab] logits are never written out
    (the reference's model forward materialises them and ``logit_processor.py:30`` reads them back:
    MIMIGenRec.py:79, evaluate.py:363-371).

    ``head`` is the model's output embedding (a plain ``torch.nn.Linear``).  ``slice_columns`` bounds
    the slice buffer (rows x slice_columns elements, three of them with ``overlap``).  With
    ``overlap`` the slice kernels run on a second stream beside the matmul of the next slice."""

    def __init__(self, head: torch.nn.Linear, slice_columns: int, overlap: bool = False):
        if type(head) is not torch.nn.Linear:
            raise TypeError(f"the sliced LM head needs a plain torch.nn.Linear output embedding, got {type(head)}")
        self.weight, self.bias = head.weight, head.bias
        self.V = self.weight.shape[0]
        cols = max(64, int(slice_columns))
        self.bounds = [(a, min(a + cols, self.V)) for a in range(0, self.V, cols)]
        self.cols = cols
        self.overlap = bool(overlap)
        self._bufs: list[torch.Tensor] = []
        self._side: Optional[torch.cuda.Stream] = None

    def _buffers(self, rows: int, like: torch.Tensor) -> list[torch.Tensor]:
        n = 3 if self.overlap else 1
        need = rows * self.cols
        if len(self._bufs) != n or self._bufs[0].numel() < need or self._bufs[0].dtype != like.dtype:
            self._bufs = [torch.empty(need, dtype=like.dtype, device=like.device) for _ in range(n)]
        return self._bufs

    def candidates_into(self, processor, input_ids: torch.Tensor, hidden: torch.Tensor, cand,
                        one_row_per_group: bool = False) -> None:
        """``processor.candidates_into(input_ids, hidden @ weight.T + bias, cand)`` without the logits."""
        R = hidden.shape[0]
        hidden = hidden.contiguous()
        bufs = self._buffers(R, hidden)
        call = processor.chunked_step(input_ids, cand, self.V, self.bounds, hidden.dtype,
                                      one_row_per_group=one_row_per_group)
        dev = hidden.device

        def matmul(a, b, out):
            if self.bias is None:
                torch.matmul(hidden, self.weight[a:b].t(), out=out)
            else:
                torch.addmm(self.bias[a:b], hidden, self.weight[a:b].t(), out=out)

        if not self.overlap:
            for a, b in self.bounds:
                buf = bufs[0][: R * (b - a)].view(R, b - a)
                matmul(a, b, buf)
                call.feed(a, buf)
            call.finish()
            return
        if self._side is None:
            self._side = torch.cuda.Stream(device=dev)
        main, side = torch.cuda.current_stream(dev), self._side
        side.wait_stream(main)  # the rows, the candidate buffer and whoever planned the call come from there
        free: list = [None] * len(bufs)  # event after which a slice buffer may be overwritten
        for i, (a, b) in enumerate(self.bounds):
            j = i % len(bufs)
            buf = bufs[j][: R * (b - a)].view(R, b - a)
            if free[j] is not None:
                main.wait_event(free[j])
            matmul(a, b, buf)
            ready = torch.cuda.Event()
            ready.record(main)
            with torch.cuda.stream(side):
                side.wait_event(ready)
                call.feed(a, buf)
                free[j] = torch.cuda.Event()
                free[j].record(side)
        main.wait_stream(side)
        call.finish()


class FusedLMHead:
    """The LM head multiplied inside the processor call (``grx_lm_head_candidates``): logits are formed tile
    by tile on the matrix cores and reduced on the spot; nothing of shape [rows, vocab] exists.  ``head`` is a
    plain ``torch.nn.Linear`` without bias, bf16, with a hidden size that is a multiple of 64.  No repetition
    penalty and no row cut in this form."""

    def __init__(self, head: torch.nn.Linear):
        if type(head) is not torch.nn.Linear or head.bias is not None:
            raise TypeError("the fused LM head needs a plain torch.nn.Linear output embedding without bias")
        if head.weight.dtype != torch.bfloat16 or head.weight.shape[1] % 64:
            raise TypeError("the fused LM head needs bfloat16 weights and a hidden size that is a multiple of 64")
        self.weight = head.weight
        self.V = self.weight.shape[0]

    def candidates_into(self, processor, input_ids: torch.Tensor, hidden: torch.Tensor, cand,
                        one_row_per_group: bool = False) -> None:
        processor.lm_head_candidates_into(input_ids, hidden, self.weight, cand, one_row_per_group=one_row_per_group)


def _model_body(model):
    """The transformer without its LM head: what returns ``last_hidden_state``.  ``base_model_prefix`` names it
    on the model that owns the head (``model.base_model`` of a PEFT wrapper is the adapter wrapper instead)."""
    inner = model
    for _ in range(4):  # through wrappers that merely hold the causal LM (peft: .base_model.model)
        prefix = getattr(inner, "base_model_prefix", None)
        if prefix and hasattr(inner, prefix) and getattr(inner, prefix) is not inner:
            return getattr(inner, prefix)
        nxt = getattr(getattr(inner, "base_model", None), "model", None)
        if nxt is None or nxt is inner:
            break
        inner = nxt
    raise TypeError("cannot find the transformer body of this model (no base_model_prefix attribute): the sliced / "
                    "fused LM head needs the last hidden state")


def reorder_generated_suffix(cache, beam_idx: torch.Tensor, prompt_len: int) -> None:
    """KV-cache re-ordering for beams that share their prompt.

    ``generate()`` re-orders the whole cache every step (``cache.reorder_cache``: an ``index_select`` over
    [rows, heads, T, D] per layer, keys and values).  The K beams of a group are copies of one prompt and a
    beam's parent always lies in its own group, so positions below ``prompt_len`` are identical across the
    rows being exchanged: only the generated suffix has to move -- a few tokens instead of the whole prompt.
    Caches whose layers are not plain [rows, heads, T, D] key / value tensors take the general route."""
    layers = getattr(cache, "layers", None)
    if not layers or not all(
            getattr(l, "keys", None) is not None and torch.is_tensor(l.keys) and l.keys.dim() == 4
            and not getattr(l, "is_sliding", False) for l in layers):
        cache.reorder_cache(beam_idx)
        return
    for layer in layers:
        if layer.keys.shape[2] <= prompt_len:
            continue  # nothing generated yet: every row of a group holds the same cache
        idx = beam_idx.to(layer.keys.device)
        layer.keys[:, :, prompt_len:, :] = layer.keys[:, :, prompt_len:, :].index_select(0, idx)
        layer.values[:, :, prompt_len:, :] = layer.values[:, :, prompt_len:, :].index_select(0, idx)


@dataclass
class ConstrainedBeamOutput:
    sequences: torch.Tensor         # [B * num_return_sequences, prompt_len + generated]
    sequences_scores: torch.Tensor  # [B * num_return_sequences]
    beam_indices: torch.Tensor      # [B * num_return_sequences, generated]
    steps: int


@torch.no_grad()
def constrained_generate(
    model,
    input_ids: torch.Tensor,
    processor,
    num_beams: int,
    max_new_tokens: int,
    attention_mask: Optional[torch.Tensor] = None,
    eos_token_id: Optional[int] = None,
    pad_token_id: Optional[int] = None,
    length_penalty: float = 1.0,
    early_stopping=False,
    num_return_sequences: Optional[int] = None,
    max_steps: Optional[int] = None,
    share_prefill: bool = True,
    lm_head_slice: Optional[int] = None,
    lm_head_overlap: bool = False,
    lm_head: Optional[str] = None,
) -> ConstrainedBeamOutput:
    """``input_ids`` [B, T] (left-padded like the reference's batches) on the model's device.

    ``processor`` is a genrec_amd processor (pass ``repetition_penalty=`` to it for what
    ``generate(repetition_penalty=...)`` would do).  ``max_steps`` fixes the number of decoding
    steps (a trie of L levels always takes L + 1) and avoids the per-step host check of HF's loop
    condition.  ``share_prefill=False`` prefills all K copies like ``generate()`` does.
    ``lm_head_slice`` (columns): run the LM head slice by slice through :class:`SlicedLMHead` instead
    of materialising [rows, vocab] logits (the fused repetition penalty works there too).
    ``lm_head="fused"``: the LM head is multiplied inside the processor call (:class:`FusedLMHead`; bf16 model,
    no repetition penalty): no logits tensor at all.
    """
    if lm_head not in (None, "fused"):
        raise ValueError(f'lm_head must be None or "fused", got {lm_head!r}')
    if lm_head and lm_head_slice:
        raise ValueError("choose one of lm_head_slice and lm_head")
    B, T = input_ids.shape
    K = int(num_beams)
    dev = input_ids.device
    if attention_mask is None:
        attention_mask = torch.ones_like(input_ids)
    eos = eos_token_id if eos_token_id is not None else getattr(model.config, "eos_token_id", None)
    pad = pad_token_id if pad_token_id is not None else getattr(model.config, "pad_token_id", None)
    if isinstance(eos, (list, tuple)):
        eos = eos[0]

    if hasattr(processor, "reset"):
        processor.reset()
    rows = input_ids.repeat_interleave(K, dim=0)
    # With a shared prefill the K beams of a group start from one logit row: the first processor call
    # reads B rows instead of B * K identical ones and the first beam step extends all beams from it.
    state = BeamSearchState(rows, K, T + max_new_tokens, eos, pad, length_penalty, early_stopping,
                            shared_prompt=True, planner=processor, first_step_shared=share_prefill)
    # whole-logits steps hand over at most n_keep candidates per row (all a group can keep); the sliced
    # step emits every allowed token
    keep = 0 if (lm_head_slice or lm_head) else processor.keep_top_for(state.n_keep)
    cmax = processor.candidate_width(state.n_keep if keep else 0)
    cand = Candidates(B * K, cmax, dev)

    # ---- prefill (only the last position's logits are needed: the LM head is the widest matmul)
    import inspect

    last_only = {"logits_to_keep": 1} if "logits_to_keep" in inspect.signature(model.forward).parameters else {}
    sliced = None
    if lm_head_slice or lm_head:
        sliced = (FusedLMHead(model.get_output_embeddings()) if lm_head else
                  SlicedLMHead(model.get_output_embeddings(), lm_head_slice, lm_head_overlap))
        body = _model_body(model)  # the transformer without the LM head

        def forward(**kw):
            out = body(**kw)
            return out.last_hidden_state[:, -1, :], out.past_key_values
    else:
        def forward(**kw):
            out = model(**kw, **({} if kw.get("past_key_values") is not None else last_only))
            return out.logits[:, -1, :], out.past_key_values
    vocab = model.get_output_embeddings().weight.shape[0] if sliced is not None else None
    # as prepare_inputs_for_generation builds them: running count of real tokens, 1 on padding
    position_ids = (attention_mask.long().cumsum(-1) - 1).masked_fill(attention_mask == 0, 1)
    # `logits` is [rows, V] -- or, with a sliced LM head, the last hidden states [rows, H] it starts from
    if share_prefill:
        logits, cache = forward(input_ids=input_ids, attention_mask=attention_mask, position_ids=position_ids,
                                use_cache=True, return_dict=True)
        cache.batch_repeat_interleave(K)  # `logits` stays [B, ...]: one row per group
        mask = attention_mask.repeat_interleave(K, dim=0)
    else:
        mask = attention_mask.repeat_interleave(K, dim=0)
        logits, cache = forward(input_ids=rows, attention_mask=mask,
                                position_ids=position_ids.repeat_interleave(K, dim=0), use_cache=True,
                                return_dict=True)
    next_pos = mask.long().sum(dim=1, keepdim=True)  # position of the first generated token

    def candidates(ids, x, buf, one_row_per_group=False):
        if sliced is not None:
            sliced.candidates_into(processor, ids, x, buf, one_row_per_group=one_row_per_group)
        else:
            processor.candidates_into(ids, x, buf, one_row_per_group=one_row_per_group, keep_top=keep,
                                      row_bias=None if one_row_per_group else state.row_bias(),
                                      shared_prefix=None if one_row_per_group else state.shared_prefix())

    step = 0
    while True:
        logits = logits.contiguous()
        V = vocab if vocab is not None else logits.shape[-1]
        if step == 0 and share_prefill:
            first = Candidates(B, cmax, dev)
            candidates(state.prompt_rows(), logits, first, one_row_per_group=True)
            res = state.step(first, V, planner=processor, shared_rows=True)
        else:
            candidates(state.flat_running_sequences(), logits, cand)
            res = state.step(cand, V, planner=processor)
        step += 1
        if max_steps is not None:
            if step >= max_steps:
                break
        elif not state.has_unfinished():
            break
        if step >= max_new_tokens:
            break
        # ---- model side of the next step: re-order the cache, feed the tokens just chosen.  Rows of a group
        # share their prompt's cache bit for bit only when one prefill was copied K times; K separately
        # prefilled copies are re-ordered whole, as generate() does
        if share_prefill:
            reorder_generated_suffix(cache, res.next_parent.long(), T)
        else:
            cache.reorder_cache(res.next_parent.long())
        new_tokens = state.running_sequences.view(B * K, -1)[:, state.cur_len - 1 : state.cur_len]
        mask = torch.cat([mask, torch.ones((B * K, 1), dtype=mask.dtype, device=dev)], dim=1)
        logits, cache = forward(input_ids=new_tokens, attention_mask=mask, position_ids=next_pos,
                                past_key_values=cache, use_cache=True, return_dict=True)
        next_pos = next_pos + 1
    seqs, scores, bidx = state.finalize(num_return_sequences)
    processor.check_errors()
    return ConstrainedBeamOutput(seqs, scores, bidx, step)


def make_fused_generate(model, processor, num_beams: int = 50, max_completion_length: int = 128,
                        length_penalty: float = 1.0, max_steps: Optional[int] = None,
                        lm_head_slice: Optional[int] = None, lm_head_overlap: bool = False,
                        lm_head: Optional[str] = None):
    """A ``generate(*args, **kwargs)`` to install on a model in place of the wrapper the reference
    builds in ``_make_mimigenrec_generate`` (MIMIGenRec.py:9-81): the trainer hands over every prompt
    ``num_beams`` times ([B * K, T] ``input_ids`` / ``attention_mask``), the wrapper keeps the first copy
    of each group (MIMIGenRec.py:74-78), resets the processor (26-48, 57-59) and beam-searches with
    ``num_return_sequences = num_beams`` (66-67), returning the [B * K, T + generated] token tensor that
    ``model.generate`` returns without ``return_dict_in_generate``.  Here the search is the fused loop.

    Sampling arguments (temperature, top_p, top_k) have no effect on beam search without sampling, which
    is what every launcher of the reference uses; pass ``repetition_penalty`` to the processor."""
    K = int(num_beams)

    def generate(*args, **kwargs):
        input_ids = kwargs.get("input_ids", args[0] if args else None)
        if input_ids is None or input_ids.dim() != 2 or input_ids.shape[0] % K:
            raise ValueError(f"generate() needs input_ids of shape [batch * {K}, T]")
        attention_mask = kwargs.get("attention_mask")
        bs, dim = input_ids.shape[0] // K, input_ids.shape[-1]
        prompts = input_ids.view(bs, K, dim)[:, 0, :]
        mask = None if attention_mask is None else attention_mask.view(bs, K, dim)[:, 0, :]
        out = constrained_generate(model, prompts.contiguous(), processor, K, max_completion_length,
                                   attention_mask=None if mask is None else mask.contiguous(),
                                   length_penalty=length_penalty, num_return_sequences=K, max_steps=max_steps,
                                   lm_head_slice=lm_head_slice, lm_head_overlap=lm_head_overlap, lm_head=lm_head)
        return out.sequences

    return generate
