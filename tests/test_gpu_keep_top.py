"""GPU: candidate rows cut to their ``keep_top`` best (``grx_constraint.keep_top`` / ``row_bias``).

A beam step that keeps n_keep candidates per group never takes more than n_keep from one row, so the
streaming kernel may hand over just those -- if it ranks exactly like the beam step: by the fp32 sum
``logp + running score`` descending, token ascending.  Checked here against a stable sort of the
whole-row candidates, bit for bit, and through searches that must not change."""

import numpy as np
import pytest
import torch

from test_gpu_processors import _dev

pytestmark = pytest.mark.gpu
SLACK = 64  # GRX_KEEP_TOP_SLACK: rows that would shed fewer entries are emitted whole


def _wide_trie(fanouts, seed=0, eos=2, prefix=(11, 12, 13), first_token=100):
    """Items over levels with the given fan-outs (level l uses tokens from its own range)."""
    rng = np.random.default_rng(seed)
    seqs, base = [], first_token
    level_tokens = []
    for f in fanouts:
        level_tokens.append(list(range(base, base + f)))
        base += f
    for t0 in level_tokens[0]:
        for _ in range(3):
            rest = [int(rng.choice(toks)) for toks in level_tokens[1:]]
            seqs.append(list(prefix) + [t0] + rest + [eos])
    return seqs, base


def _expected(full, bias, keep):
    """Per row: (tokens, logp) of the `keep` best by (fp32(logp + bias) desc, token asc); rows of at most
    `keep` candidates whole, tokens ascending."""
    count = full.count.cpu().numpy()
    tok = full.tokens.cpu().numpy()
    lp = full.logp.cpu().numpy()
    out = []
    for r in range(len(count)):
        n = int(count[r])
        t, v = tok[r, :n], lp[r, :n]
        if n <= keep + SLACK:
            out.append((t, v))
            continue
        s = (v + np.float32(bias[r])).astype(np.float32)
        order = np.lexsort((t, -s.astype(np.float64)))  # primary: sum descending; ties: token ascending
        out.append((t[order[:keep]], v[order[:keep]]))
    return out


def _same_entries(got_t, got_v, want_t, want_v, r):
    """The same (token, logp bits) pairs; rows that were cut come in no particular order."""
    g = np.argsort(got_t, kind="stable")
    w = np.argsort(want_t, kind="stable")
    assert np.array_equal(got_t[g], want_t[w]), r
    assert np.array_equal(got_v[g].view(np.int32), want_v[w].view(np.int32)), r


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("fan,keep,rows", [(162, 32, 64), (700, 32, 48), (700, 100, 16), (1024, 512, 8), (40, 64, 32), (97, 32, 32), (96, 32, 32)])
def test_rows_are_cut_to_their_best_in_beam_order(dtype, fan, keep, rows):
    from genrec_amd.logit_processor import Candidates, ConstrainedLogitsProcessor
    from genrec_amd.trie import FlatTrie, create_prefix_allowed_tokens_fn

    dev = _dev()
    seqs, V = _wide_trie([fan, 5])
    V = (V + 63) // 64 * 64
    prefix, eos = [11, 12, 13], 2
    fn = create_prefix_allowed_tokens_fn(FlatTrie(seqs), prefix)
    K = 8
    torch.manual_seed(fan + keep)
    ids = torch.cat([torch.randint(20, 90, (rows, 5)), torch.tensor([prefix] * rows)], dim=1).to(dev)
    # coarse logits: many exact ties inside a row, so the tie rule decides which tokens survive
    scores = (torch.randint(-8, 8, (rows, V)).float() * 0.5).to(dtype).to(dev)
    bias = (torch.randn(rows) * 3).to(dev)
    bias[::5] = -1.0e9  # placeholder beams: sums collapse onto few values
    proc = ConstrainedLogitsProcessor(fn, K, 3, None, eos, sync_errors=True)
    proc.count = 0
    full = proc.candidates(ids, scores)
    assert int(full.count.max()) == fan
    cut = Candidates(rows, proc.candidate_width(keep), dev)
    assert cut.cmax == min(fan, keep + SLACK)
    cut.buf.fill_(-5)
    proc.count = 0
    proc.candidates_into(ids, scores, cut, keep_top=keep, row_bias=bias)
    assert torch.equal(cut.logsumexp, full.logsumexp)
    want = _expected(full, bias.cpu().numpy(), keep)
    got_n, got_t, got_v = cut.count.cpu().numpy(), cut.tokens.cpu().numpy(), cut.logp.cpu().numpy()
    for r, (t, v) in enumerate(want):
        assert got_n[r] == len(t)
        _same_entries(got_t[r, : len(t)], got_v[r, : len(t)], t, v, r)
    # whole lines: zero padding behind the kept entries
    n_line = ((cut.count + 1 + 15) // 16) * 16
    e = torch.arange(cut.stride, device=dev)[None, :]
    pad = (e > cut.count[:, None]) & (e < n_line[:, None])
    assert not bool(cut.buf[pad].any())
    # no bias: ranked by the log-probs alone
    proc.count = 0
    proc.candidates_into(ids, scores, cut, keep_top=keep)
    for r, (t, v) in enumerate(_expected(full, np.zeros(rows, dtype=np.float32), keep)):
        _same_entries(cut.tokens[r, : len(t)].cpu().numpy(), cut.logp[r, : len(t)].cpu().numpy(), t, v, r)
    proc.check_errors()


def test_cut_rows_under_a_repetition_penalty_are_ranked_by_the_penalised_values():
    from genrec_amd.logit_processor import Candidates, ConstrainedLogitsProcessor
    from genrec_amd.trie import FlatTrie, create_prefix_allowed_tokens_fn

    dev = _dev()
    seqs, V = _wide_trie([200, 4])
    V = (V + 63) // 64 * 64
    prefix, eos = [11, 12, 13], 2
    fn = create_prefix_allowed_tokens_fn(FlatTrie(seqs), prefix)
    rows, keep = 32, 24
    torch.manual_seed(3)
    # history full of first-level tokens: the penalty moves exactly the values that compete
    hist = torch.randint(100, 300, (rows, 12))
    ids = torch.cat([hist, torch.tensor([prefix] * rows)], dim=1).to(dev)
    scores = (torch.randn(rows, V) * 3).to(dev)
    proc = ConstrainedLogitsProcessor(fn, 8, 3, None, eos, sync_errors=True, repetition_penalty=1.3)
    proc.count = 0
    full = proc.candidates(ids, scores)
    bias = torch.randn(rows, device=dev)
    cut = Candidates(rows, proc.candidate_width(keep), dev)
    proc.count = 0
    proc.candidates_into(ids, scores, cut, keep_top=keep, row_bias=bias)
    for r, (t, v) in enumerate(_expected(full, bias.cpu().numpy(), keep)):
        assert int(cut.count[r]) == len(t) == keep
        _same_entries(cut.tokens[r, : len(t)].cpu().numpy(), cut.logp[r, : len(t)].cpu().numpy(), t, v, r)
    proc.check_errors()


@pytest.mark.parametrize("K", [4, 16])
def test_a_search_over_cut_rows_equals_the_search_over_whole_rows(K):
    """Step by step on the same state: candidates cut to n_keep with the running scores as bias give
    the beam step exactly the results of the whole rows (all n_keep slots, not just the live ones)."""
    from genrec_amd.beam import BeamSearchState
    from genrec_amd.logit_processor import Candidates, ConstrainedLogitsProcessor
    from genrec_amd.trie import FlatTrie, create_prefix_allowed_tokens_fn

    dev = _dev()
    seqs, V = _wide_trie([120, 60, 30], seed=K)
    V = (V + 63) // 64 * 64
    prefix, eos = [11, 12, 13], 2
    fn = create_prefix_allowed_tokens_fn(FlatTrie(seqs), prefix)
    B = 24
    torch.manual_seed(K)
    prompts = torch.cat([torch.randint(20, 90, (B, 6)), torch.tensor([prefix] * B)], dim=1).to(dev)
    table = [(torch.randint(-20, 20, (B * K, V)).float() * 0.25).to(dev) for _ in range(4)]
    outs = []
    for cut in (False, True):
        proc = ConstrainedLogitsProcessor(fn, K, 3, prefix, eos, sync_errors=True)
        proc.reset()
        state = BeamSearchState(prompts.repeat_interleave(K, dim=0), K, prompts.shape[1] + 5, eos, 0)
        keep = state.n_keep if cut else 0
        cand = Candidates(B * K, proc.candidate_width(state.n_keep if cut else 0), dev)
        assert cand.cmax == (min(120, state.n_keep + SLACK) if cut else 120)
        trace = []
        for step in range(4):
            proc.candidates_into(state.flat_running_sequences(), table[step], cand, keep_top=keep,
                                 row_bias=state.row_bias() if cut else None)
            res = state.step(cand, V)
            trace.append([t.clone() for t in (res.topk_scores, res.topk_parent, res.topk_token, res.topk_hit)])
        outs.append((trace, state.finalize()))
    for a, b in zip(outs[0][0], outs[1][0]):
        finite = torch.isfinite(a[0])
        assert torch.equal(finite, torch.isfinite(b[0]))
        for x, y in zip(a, b):
            assert torch.equal(x[finite], y[finite])
    for x, y in zip(outs[0][1], outs[1][1]):
        assert torch.equal(x, y)


def test_keep_top_is_refused_where_it_cannot_be_honoured():
    from genrec_amd import _lib
    from genrec_amd.logit_processor import Candidates, ConstrainedLogitsProcessor
    from genrec_amd.trie import FlatTrie, create_prefix_allowed_tokens_fn

    dev = _dev()
    seqs, V = _wide_trie([1100])
    V = (V + 63) // 64 * 64
    fn = create_prefix_allowed_tokens_fn(FlatTrie(seqs), [11, 12, 13])
    proc = ConstrainedLogitsProcessor(fn, 4, 3, None, 2)
    assert proc.keep_top_for(8) == 0 and proc.candidate_width(8) == 1100  # the helper falls back to whole rows
    ids = torch.tensor([[30, 11, 12, 13]] * 4, device=dev)
    scores = torch.zeros(4, V, device=dev)
    cand = Candidates(4, 1100, dev)
    with pytest.raises(_lib.GenrecHipError, match="fan-out"):
        proc.candidates_into(ids, scores, cand, keep_top=8)
    with pytest.raises(_lib.GenrecHipError, match="keep_top"):
        proc.candidates_into(ids, scores, cand, keep_top=513)
    proc.candidates_into(ids, scores, cand)  # still serves whole rows
    assert int(cand.count[0]) == 1100
    # dense calls ignore the field
    small, V2 = _wide_trie([50])
    p2 = ConstrainedLogitsProcessor(create_prefix_allowed_tokens_fn(FlatTrie(small), [11, 12, 13]), 4, 3, None, 2)
    out = p2(ids, torch.zeros(4, 256, device=dev))
    assert int(torch.isfinite(out[0]).sum()) == 50
