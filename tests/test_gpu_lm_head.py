"""GPU: the LM head inside the step (``grx_lm_head_candidates`` / ``processor.lm_head_candidates_into``).

The reference's forward pass writes logits = hidden x weight^T as an [R, V] tensor (MIMIGenRec.py:79,
evaluate.py:363-371) which logit_processor.py:30 then reads back.  Here the product is formed tile by tile on
the matrix cores and consumed on the spot.  Checked against (a) the streaming kernel on the materialised
fp32 product (tokens bit-exact, log-probs to fp32 summation order) and (b) a float64 log-softmax of the
exact products of the bf16 operands (2e-3, the bound for bf16 inputs)."""

import tempfile

import pytest
import torch

from genrec_amd import _lib
from genrec_amd.fixed_hint_logit_processor import FixedHintConstrainedLogitsProcessor
from genrec_amd.logit_processor import Candidates, ConstrainedLogitsProcessor
from genrec_amd.synth import ASSISTANT_PREFIX, SidTokenizer, sample_decode_rows
from genrec_amd.trie import FlatTrie, build_trie_from_index, create_prefix_allowed_tokens_fn
from helpers import write_index_file
from test_gpu_processors import _dev

pytestmark = pytest.mark.gpu


def _instruments(K, cls=ConstrainedLogitsProcessor):
    path = write_index_file("Instruments", tempfile.mkdtemp())
    tok = SidTokenizer.from_index(path)
    trie, suffix, pidx = build_trie_from_index(path, tok, prefix=ASSISTANT_PREFIX)
    fn = create_prefix_allowed_tokens_fn(trie, suffix)
    return tok, trie, (lambda: cls(fn, K, pidx, None, tok.eos_token_id, sync_errors=True))


def _compare(got, want, exact_logp=None, tol=2e-5):
    assert torch.equal(got.count, want.count)
    valid = torch.arange(got.cmax, device=got.count.device)[None, :] < got.count[:, None]
    assert torch.equal(got.tokens[valid], want.tokens[valid])
    assert float((got.logp[valid] - want.logp[valid]).abs().max()) <= tol
    assert float((got.logsumexp - want.logsumexp).abs().max()) <= tol
    if exact_logp is not None:
        ex = exact_logp.gather(1, torch.where(valid, got.tokens, 0).long())
        assert float((got.logp.double() - ex)[valid].abs().max()) <= 2e-3


@pytest.mark.parametrize("R,H,step", [(48, 256, 0), (304, 512, 1), (1024, 1024, 2)])
def test_fused_lm_head_equals_the_materialised_product(R, H, step):
    """Row counts that are not multiples of the 256-row tile, a vocabulary (152,640) that is not a multiple of
    the 256-column tile, several decode depths."""
    dev = _dev()
    K = 16
    tok, trie, make = _instruments(K)
    V = tok.vocab_size_padded
    g = torch.Generator(device=dev).manual_seed(R + H)
    hidden = (torch.randn((R, H), generator=g, device=dev) * 0.5).to(torch.bfloat16)
    weight = (torch.randn((V, H), generator=g, device=dev) * 0.2).to(torch.bfloat16)
    ids = torch.from_numpy(sample_decode_rows(trie, R // K, K, step, prompt_len=17, seed=R)).to(dev)
    logits = hidden.float() @ weight.float().T  # exact products of the bf16 operands, fp32 sums
    proc = make()
    proc.count = step
    want = proc.candidates(ids, logits)
    proc.count = step
    got = Candidates(R, trie.max_fanout, dev)
    proc.lm_head_candidates_into(ids, hidden, weight, got)
    assert proc.count == step + 1
    exact = torch.log_softmax(hidden.double() @ weight.double().T, dim=-1)
    _compare(got, want, exact)


def test_fused_lm_head_with_tokens_below_the_kept_columns_and_the_fixed_hint_mode():
    """A trie whose EOS id is 2 and whose item tokens sit in the middle of the vocabulary: nothing lies within
    the kept top columns, every logit comes from the finishing kernel's dot products."""
    dev = _dev()
    V, H, R, K = 20480, 128, 64, 4
    seqs = [[11, 12, 13, 5000 + a, 9000 + b, 2] for a in range(6) for b in range(5)] + \
           [[11, 12, 13, 20470 + a, 9000, 2] for a in range(4)]  # ... and some that do
    fn = create_prefix_allowed_tokens_fn(FlatTrie(seqs), [11, 12, 13])
    g = torch.Generator(device=dev).manual_seed(5)
    hidden = (torch.randn((R, H), generator=g, device=dev)).to(torch.bfloat16)
    weight = (torch.randn((V, H), generator=g, device=dev) * 0.3).to(torch.bfloat16)
    logits = hidden.float() @ weight.float().T
    for cls in (ConstrainedLogitsProcessor, FixedHintConstrainedLogitsProcessor):
        for depth, tail in ((0, []), (1, [5003]), (2, [20471, 9000])):
            ids = torch.tensor([[7, 8, 9, 11, 12, 13] + tail] * R, device=dev)
            proc = cls(fn, K, 3, [11, 12, 13], 2, sync_errors=True)
            proc.count = depth
            want = proc.candidates(ids, logits)
            proc.count = depth
            got = Candidates(R, want.cmax, dev)
            proc.lm_head_candidates_into(ids, hidden, weight, got)
            _compare(got, want)


def test_fused_lm_head_reports_row_errors_and_refuses_what_it_does_not_serve():
    dev = _dev()
    K = 2
    fn = create_prefix_allowed_tokens_fn(FlatTrie([[11, 12, 13, 50, 2]]), [11, 12, 13])
    hidden = torch.zeros((2, 64), dtype=torch.bfloat16, device=dev)
    weight = torch.zeros((512, 64), dtype=torch.bfloat16, device=dev)
    cand = Candidates(2, 4, dev)
    proc = ConstrainedLogitsProcessor(fn, K, 3, None, 2, sync_errors=True)
    with pytest.raises(AssertionError, match="No valid tokens"):
        proc.lm_head_candidates_into(torch.tensor([[1, 2, 3, 4]] * 2, device=dev), hidden, weight, cand)
    pen = ConstrainedLogitsProcessor(fn, K, 3, None, 2, sync_errors=True, repetition_penalty=1.1)
    with pytest.raises(_lib.GenrecHipError, match="repetition penalty"):
        pen.lm_head_candidates_into(torch.tensor([[5, 11, 12, 13]] * 2, device=dev), hidden, weight, cand)
    with pytest.raises(TypeError):
        proc.lm_head_candidates_into(torch.tensor([[5, 11, 12, 13]] * 2, device=dev), hidden.float(), weight.float(), cand)
    with pytest.raises(_lib.GenrecHipError, match="multiple of 64"):
        proc.lm_head_candidates_into(torch.tensor([[5, 11, 12, 13]] * 2, device=dev), hidden[:, :32].contiguous(),
                                     weight[:, :32].contiguous(), cand)


@pytest.mark.parametrize("kind,share", [("count", True), ("fixed", True), ("count", False)])
def test_generation_with_the_fused_lm_head_equals_generation_from_whole_logits(kind, share):
    """constrained_generate(lm_head="fused") -- transformer body, LM head multiplied inside the processor call --
    returns the beams of the same loop fed the model's own bf16 logits.  The fused head keeps fp32 sums where the
    model rounds its logits to bf16, so scores agree to bf16 resolution and beams are compared where the whole-logits
    run leaves a clear margin between neighbours."""
    from transformers import Qwen2Config, Qwen2ForCausalLM

    from genrec_amd.generate import constrained_generate
    from helpers import load_npz

    dev = _dev()
    data, meta = load_npz("beams.npz")
    prefix, eos, pad, V = meta["prefix"], meta["eos"], meta["pad"], meta["V"]
    fn = create_prefix_allowed_tokens_fn(FlatTrie(data["trie_sequences"].tolist()), prefix)
    K = 8
    torch.manual_seed(11)
    cfg = Qwen2Config(vocab_size=V, hidden_size=64, intermediate_size=128, num_hidden_layers=2,
                      num_attention_heads=2, num_key_value_heads=2, max_position_embeddings=128,
                      eos_token_id=eos, pad_token_id=pad, bos_token_id=None, tie_word_embeddings=False)
    model = Qwen2ForCausalLM(cfg).eval()
    with torch.no_grad():
        for p in model.parameters():
            p.mul_(4.0)
    model = model.to(dev).to(torch.bfloat16)
    cls = ConstrainedLogitsProcessor if kind == "count" else FixedHintConstrainedLogitsProcessor
    g = torch.Generator().manual_seed(21)
    rows, lens = [], [9, 6, 11, 8]
    for n in lens:
        body = torch.randint(20, 90, (n,), generator=g).tolist()
        rows.append([pad] * (max(lens) - n) + body + prefix)
    prompts = torch.tensor(rows, device=dev)
    mask = (prompts != pad).long()
    # reference run: the same body, the head applied in fp32 to the bf16 hidden states (what the fused head computes)
    head = model.get_output_embeddings()

    class _Fp32Head(torch.nn.Module):
        def __init__(self, w):
            super().__init__()
            self.weight = w

        def forward(self, h):
            return h.float() @ self.weight.float().T

    model.set_output_embeddings(_Fp32Head(head.weight))
    want = constrained_generate(model, prompts, cls(fn, K, 3, prefix, eos), K, 8, attention_mask=mask,
                                eos_token_id=eos, pad_token_id=pad, length_penalty=0.0, share_prefill=share)
    model.set_output_embeddings(head)
    got = constrained_generate(model, prompts, cls(fn, K, 3, prefix, eos), K, 8, attention_mask=mask,
                               eos_token_id=eos, pad_token_id=pad, length_penalty=0.0, share_prefill=share,
                               lm_head="fused")
    live = want.sequences_scores > -1e8
    assert int(live.sum()) >= 12 and got.steps == want.steps
    assert torch.equal(got.sequences[live], want.sequences[live])
    assert torch.equal(got.beam_indices[live], want.beam_indices[live])
    assert float((got.sequences_scores[live] - want.sequences_scores[live]).abs().max()) <= 1e-4
