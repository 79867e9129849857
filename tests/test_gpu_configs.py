"""The benchmark configurations at their full sizes, through the C ABI.

A CPU run of the whole configuration would take minutes to hours, so each
configuration is checked three ways:

  * a sample of prompt groups goes through the CPU oracle end to end
    (sequences and beam parents bit-exact, scores within tolerance);
  * every row of every step is checked against properties computed
    independently on the device or with vectorised numpy: the allowed tokens
    are the children of the trie node the row's window leads to (bit-exact),
    the log-sum-exp equals a float64 ``torch.logsumexp`` (1e-5), each
    candidate's log-prob equals ``logit - lse`` (1e-5), and the kept top-2K
    equals a stable ``torch.sort`` of the accumulated scores (bit-exact);
  * the final ranking of every group is checked for validity (items of the
    index, distinct, EOS-terminated), order (scores descending) and value
    (score = sum of the path's log-probs / length ** length_penalty,
    re-derived from the logits through the returned beam parents).
"""

import os
import tempfile

import numpy as np
import pytest
import torch

from helpers import write_index_file

pytestmark = pytest.mark.gpu

LIVE = -1.0e8
LSE_TOL = 1e-5        # fp32 arithmetic on identical inputs (bf16 inputs are widened exactly)
ORACLE_SCORE_TOL = 5e-5  # torch's CPU fp32 log_softmax is itself up to 2.2e-5 off at V = 152k


def _dev():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch.device("cuda", 0)


def _build(index_path, K, processor, initial_count=None):  # -> tokenizer, trie, prefix ids, prefix_index, processor
    from genrec_amd.fixed_hint_logit_processor import FixedHintConstrainedLogitsProcessor
    from genrec_amd.logit_processor import ConstrainedLogitsProcessor
    from genrec_amd.synth import ASSISTANT_PREFIX, SidTokenizer
    from genrec_amd.trie import build_trie_from_index, create_prefix_allowed_tokens_fn

    tok = SidTokenizer.from_index(index_path)
    trie, suffix, pidx = build_trie_from_index(index_path, tok, prefix=ASSISTANT_PREFIX)
    fn = create_prefix_allowed_tokens_fn(trie, suffix)
    if processor == "fixed_hint":
        proc = FixedHintConstrainedLogitsProcessor(fn, K, pidx, suffix, tok.eos_token_id, sync_errors=True)
    else:
        extra = {} if initial_count is None else {"initial_count": initial_count}
        proc = ConstrainedLogitsProcessor(fn, K, pidx, suffix, tok.eos_token_id, sync_errors=True, **extra)
    return tok, trie, suffix, pidx, proc


class CsrWalker:
    """Vectorised trie walk over the flat child table (numpy)."""

    def __init__(self, trie):
        self.off = np.asarray(trie.child_off, dtype=np.int64)
        self.tok = np.asarray(trie.child_tok, dtype=np.int64)
        parent = np.repeat(np.arange(len(self.off) - 1, dtype=np.int64), np.diff(self.off))
        self.key = (parent << 24) | self.tok  # edges are ordered by (parent, token): already sorted
        assert np.all(np.diff(self.key) > 0)

    def nodes(self, rows, start):
        """Node reached from the root by rows[r, start[r]:]; -1 when the path leaves the trie."""
        R, T = rows.shape
        node = np.zeros(R, dtype=np.int64)
        for p in range(int(start.min()), T):
            active = (start <= p) & (node >= 0)
            want = (np.maximum(node, 0) << 24) | rows[:, p]
            pos = np.minimum(np.searchsorted(self.key, want), len(self.key) - 1)
            hit = self.key[pos] == want
            node = np.where(active, np.where(hit, pos + 1, -1), node)
        return node

    def children(self, node, cmax):
        """(count [R], tokens [R, cmax] zero-padded) of the given nodes."""
        lo = self.off[np.maximum(node, 0)]
        cnt = np.where(node >= 0, self.off[np.maximum(node, 0) + 1] - lo, 0)
        cols = np.arange(cmax)[None, :]
        at = np.minimum(lo[:, None] + cols, len(self.tok) - 1)
        toks = np.where(cols < cnt[:, None], self.tok[at], 0)
        return cnt, toks


def _fill_logits(R, V, dtype, dev, seed):
    gen = torch.Generator(device=dev).manual_seed(seed)
    x = torch.empty((R, V), dtype=dtype, device=dev)
    for r0 in range(0, R, 2048):
        r1 = min(R, r0 + 2048)
        x[r0:r1] = (torch.randn((r1 - r0, V), generator=gen, device=dev) * 4.0).to(dtype)
    return x


def _exact_lse(x):
    out = torch.empty(x.shape[0], dtype=torch.float64, device=x.device)
    for r0 in range(0, x.shape[0], 1024):
        out[r0 : r0 + 1024] = torch.logsumexp(x[r0 : r0 + 1024].double(), dim=-1)
    return out


def _run_config(index_path, processor, B, K, dtype, hint_depths=None, initial_count=None, sample_groups=8,
                seed=0, check_permutation=False, length_penalty=1.0):
    from oracle.beam import ref_constrained_beam_search
    from oracle.processors import DictTrie, RefConstrainedLogitsProcessor, RefFixedHintConstrainedLogitsProcessor

    from genrec_amd.beam import BeamSearchState
    from genrec_amd.logit_processor import Candidates
    from genrec_amd.synth import sample_decode_rows

    dev = _dev()
    tok, trie, suffix, pidx, proc = _build(index_path, K, processor, initial_count)
    V, eos, pad = tok.vocab_size_padded, tok.eos_token_id, tok.pad_token_id
    R = B * K
    n_sid = len(trie.level_fanout()) - len(suffix) - 1          # SID levels
    steps = n_sid + 1 - (min(hint_depths) if hint_depths is not None else 0)
    prompts_np = sample_decode_rows(trie, B, 1, 0, prompt_len=61, seed=seed + 11, hint_depths=hint_depths)
    depth = np.asarray(hint_depths if hint_depths is not None else [0] * B, dtype=np.int64)
    prefix_at = np.repeat((int(depth.max()) - depth) + 61, K)  # column where the assistant prefix starts
    prompts = torch.from_numpy(prompts_np).to(dev)
    T0 = prompts.shape[1]
    logits = [_fill_logits(R, V, dtype, dev, seed * 100 + t) for t in range(steps)]
    walker = CsrWalker(trie)
    cmax = max(trie.max_fanout, 1)

    def run(prompt_rows, logits_list, record):
        state = BeamSearchState(prompt_rows, K, T0 + steps + 1, eos, pad, length_penalty, False, shared_prompt=True)
        cand = Candidates(R, cmax, dev)
        if hasattr(proc, "reset"):
            proc.reset()
        for t in range(steps):
            flat = state.flat_running_sequences()
            run_in = state.running_beam_scores.clone()
            proc.candidates_into(flat, logits_list[t], cand)
            res = state.step(cand, V)
            if record is not None:
                record.append(dict(flat=flat.clone(), run_in=run_in, buf=cand.buf.clone(),
                                   topk_scores=res.topk_scores.clone(), topk_token=res.topk_token.clone(),
                                   topk_parent=res.topk_parent.clone()))
        proc.check_errors()
        # a trie of L levels is exhausted after L + 1 steps; groups with fewer than K items keep HF's
        # loop condition true (their finished list never fills), so the step count is fixed instead
        return state

    trace = []
    state = run(prompts.repeat_interleave(K, dim=0).contiguous(), logits, trace)

    # ---- every row of every step ------------------------------------------------------
    lses = []
    for t, rec in enumerate(trace):
        snap = Candidates(R, cmax, dev)
        snap.buf = rec["buf"]
        rows = rec["flat"].cpu().numpy()
        node = walker.nodes(rows, prefix_at)
        want_cnt, want_tok = walker.children(node, cmax)
        if processor == "fixed_hint":  # a row that ends with EOS may only continue with EOS
            ended = rows[:, -1] == eos
            want_cnt = np.where(ended, 1, want_cnt)
            want_tok = np.where(ended[:, None], np.where(np.arange(cmax)[None, :] == 0, eos, 0), want_tok)
        assert want_cnt.min() >= 1
        got_cnt = snap.count.cpu().numpy()
        assert np.array_equal(got_cnt, want_cnt), f"step {t}: allowed-set sizes differ"
        valid = np.arange(cmax)[None, :] < want_cnt[:, None]
        got_tok = snap.tokens.cpu().numpy()
        assert np.array_equal(np.where(valid, got_tok, 0), want_tok), f"step {t}: allowed tokens differ"

        exact = _exact_lse(logits[t])
        lse = snap.logsumexp
        assert float((lse.double() - exact).abs().max()) <= LSE_TOL, f"step {t}: log-sum-exp"
        valid_d = torch.from_numpy(valid).to(dev)
        tok_d = torch.from_numpy(want_tok).to(dev)
        want_lp = torch.gather(logits[t], 1, tok_d).double() - exact[:, None]
        zero = torch.zeros((), dtype=torch.float64, device=dev)  # entries past a row's count are unspecified
        err = torch.where(valid_d, (snap.logp.double() - want_lp).abs(), zero).max()
        assert float(err) <= LSE_TOL, f"step {t}: candidate log-probs"
        lses.append(lse.clone())

        # top-2K: candidates are laid out in (beam, token) order, so a stable sort by score
        # descending is the canonical order (score desc, beam * V + token asc)
        acc = torch.where(valid_d, snap.logp + rec["run_in"].view(R, 1), torch.tensor(float("-inf"), device=dev))
        acc = acc.view(B, K * cmax)
        n_keep = rec["topk_scores"].shape[1]
        order = torch.sort(acc, dim=1, descending=True, stable=True).indices[:, :n_keep]
        want_scores = torch.gather(acc, 1, order)
        live = want_scores > LIVE
        assert torch.equal(rec["topk_scores"][live], want_scores[live]), f"step {t}: top-2K scores"
        want_parent = (order // cmax).int()
        want_token = torch.gather(tok_d.view(B, K * cmax), 1, order).int()
        assert torch.equal(rec["topk_parent"][live], want_parent[live]), f"step {t}: beam parents"
        assert torch.equal(rec["topk_token"][live], want_token[live]), f"step {t}: selected tokens"

    # ---- final rankings ---------------------------------------------------------------
    seqs, scores, bidx = state.finalize()
    seqs, scores, bidx = seqs.view(B, K, -1), scores.view(B, K), bidx.view(B, K, -1)
    live = scores > LIVE
    assert bool(live[:, 0].all()), "every group returns at least one item"
    n_items_below = _items_below(walker, prompts_np, prefix_at[::K])
    assert np.array_equal(live.sum(dim=1).cpu().numpy(), np.minimum(n_items_below, K)), "live beams per group"
    masked = torch.where(live, scores, torch.tensor(float("-inf"), device=dev))
    assert bool((masked[:, :-1] >= masked[:, 1:]).all()), "rankings are sorted by score"
    gen_len = (bidx >= 0).sum(dim=2)                                       # [B, K]
    total = torch.zeros((B, K), dtype=torch.float64, device=dev)
    for t in range(bidx.shape[2]):
        row = bidx[:, :, t].long().clamp(min=0).view(-1)
        token = seqs[:, :, T0 + t].reshape(-1)
        lp = logits[t][row, token].double() - lses[t][row].double()
        total += torch.where(bidx[:, :, t] >= 0, lp.view(B, K), torch.zeros((), dtype=torch.float64, device=dev))
    want = total / gen_len.clamp(min=1).double() ** length_penalty
    score_err = torch.where(live, (scores.double() - want).abs(), torch.zeros((), dtype=torch.float64, device=dev))
    assert float(score_err.max()) <= 1e-4, "scores are path log-prob sums"
    seq_np, live_np, len_np = seqs.cpu().numpy(), live.cpu().numpy(), gen_len.cpu().numpy()
    flat_seq = seq_np.reshape(R, -1)
    # a returned row is prefix + item + EOS: the walk ends on a childless node, reached through EOS
    end = T0 + len_np.reshape(-1)
    padded = flat_seq.copy()
    cols = np.arange(padded.shape[1])[None, :]
    assert np.all(np.where((cols >= end[:, None]), padded, pad)[live_np.reshape(-1)] == pad), "padding after EOS"
    last = flat_seq[np.arange(R), np.maximum(end - 1, 0)]
    assert np.all(last[live_np.reshape(-1)] == eos)
    item_rows = np.where(cols < end[:, None], padded, -1)
    leaf = _walk_to_end(walker, item_rows, np.repeat(prefix_at[::K], K))
    assert np.all(leaf[live_np.reshape(-1)] >= 0), "returned rows are items of the index"
    for g in range(0, B, max(1, B // 64)):  # distinct items inside a group (spot groups)
        rows_g = [tuple(r[T0:]) for r, ok in zip(seq_np[g], live_np[g]) if ok]
        assert len(set(rows_g)) == len(rows_g)

    # ---- a sample of groups through the CPU oracle ---------------------------------------
    G = min(sample_groups, B)
    o_trie = DictTrie(sorted(trie))
    if processor == "fixed_hint":
        ref_proc = RefFixedHintConstrainedLogitsProcessor(o_trie.get, K, pidx, suffix, eos)
    elif initial_count is not None:
        ref_proc = RefConstrainedLogitsProcessor(o_trie.get, K, pidx, suffix, eos, initial_count=initial_count)
    else:
        ref_proc = RefConstrainedLogitsProcessor(o_trie.get, K, pidx, suffix, eos)
    cpu_logits = [x[: G * K].float().cpu() for x in logits]
    ref_state, _ = ref_constrained_beam_search(lambda step, flat: cpu_logits[step], ref_proc,
                                               torch.from_numpy(prompts_np[:G]), K, steps + 1, eos, pad,
                                               length_penalty, max_steps=steps)
    want_seqs, want_scores, want_bidx = ref_state.finalize()
    w_live = want_scores > LIVE
    got_seqs = seqs[:G].reshape(G * K, -1).cpu()
    got_bidx = bidx[:G].reshape(G * K, -1).cpu()
    got_scores = scores[:G].reshape(-1).cpu()
    assert torch.equal(w_live, got_scores > LIVE)
    width = min(got_seqs.shape[1], want_seqs.shape[1])
    assert torch.equal(got_seqs[:, :width][w_live], want_seqs[:, :width][w_live]), "item rankings vs the oracle"
    bw = min(got_bidx.shape[1], want_bidx.shape[1])
    assert torch.equal(got_bidx[:, :bw][w_live].long(), want_bidx[:, :bw][w_live].long()), "beam parents vs the oracle"
    # the oracle's per-step fp32 log_softmax error adds up over the summed steps unless the length divides it out
    oracle_tol = ORACLE_SCORE_TOL * max(1.0, float(steps) ** (1.0 - length_penalty))
    assert float((got_scores[w_live] - want_scores[w_live]).abs().max()) <= oracle_tol

    # ---- groups are independent: reversing their order reverses the results, bit for bit ----
    if check_permutation:
        flipped = [x.view(B, K, V).flip(0).reshape(R, V).contiguous() for x in logits]
        del logits
        state2 = run(prompts.flip(0).repeat_interleave(K, dim=0).contiguous(), flipped, None)
        seqs2, scores2, bidx2 = state2.finalize()
        assert torch.equal(scores2.view(B, K).flip(0), scores)
        live2 = scores > LIVE
        assert torch.equal(seqs2.view(B, K, -1).flip(0)[live2], seqs[live2])
        base = (torch.arange(B, device=dev) * K).view(B, 1, 1)
        rel, rel2 = bidx - base.int(), bidx2.view(B, K, -1).flip(0) - base.flip(0).int()
        ok = (bidx >= 0) & live2[:, :, None]
        assert torch.equal(rel[ok], rel2[ok])
    return dict(R=R, V=V, steps=steps, cmax=cmax, sequences=seqs.cpu(), scores=scores.cpu(), beam_indices=bidx.cpu())


def _items_below(walker, prompts_np, prefix_at):
    """Number of index items below each prompt's hint node (how many beams can ever be live)."""
    node = walker.nodes(prompts_np, prefix_at)
    n_nodes = len(walker.off) - 1
    leaves = np.zeros(n_nodes, dtype=np.int64)
    leaves[np.diff(walker.off) == 0] = 1
    # children have larger numbers than parents (breadth-first numbering): accumulate bottom-up
    parent = np.repeat(np.arange(n_nodes, dtype=np.int64), np.diff(walker.off))
    for e in range(len(parent) - 1, -1, -1):
        leaves[parent[e]] += leaves[e + 1]
    return leaves[node]


def _walk_to_end(walker, rows, start):
    """Node reached by rows[r, start[r]:] ignoring the -1 tail; -1 unless it is a childless node."""
    R, T = rows.shape
    node = np.zeros(R, dtype=np.int64)
    for p in range(int(start.min()), T):
        active = (start <= p) & (node >= 0) & (rows[:, p] >= 0)
        want = (np.maximum(node, 0) << 24) | np.maximum(rows[:, p], 0)
        pos = np.minimum(np.searchsorted(walker.key, want), len(walker.key) - 1)
        hit = walker.key[pos] == want
        node = np.where(active, np.where(hit, pos + 1, -1), node)
    childless = walker.off[np.maximum(node, 0) + 1] == walker.off[np.maximum(node, 0)]
    return np.where((node >= 0) & childless, node, -1)


def _synth_index(cb):
    from genrec_amd.synth import synthetic_index, write_index

    return write_index(os.path.join(tempfile.mkdtemp(), f"cb{cb}.index.json"), synthetic_index(9922, 4, cb, seed=0))


def test_c1_industrial_fp32_whole_batch_against_the_oracle():
    """Industrial_and_Scientific trie, beam 16, batch 64, fp32: every group goes through the oracle."""
    path = write_index_file("Industrial_and_Scientific", tempfile.mkdtemp())
    info = _run_config(path, "count", B=64, K=16, dtype=torch.float32, sample_groups=64, seed=1)
    assert info["R"] == 1024 and info["V"] == 152320


def test_c2_instruments_bf16_8192_rows():
    """Instruments 4-256 trie, beam 16, batch 512, bf16 logits (the benchmark workload)."""
    path = write_index_file("Instruments", tempfile.mkdtemp())
    info = _run_config(path, "count", B=512, K=16, dtype=torch.bfloat16, seed=2, check_permutation=True)
    assert info["R"] == 8192 and info["V"] == 152640 and info["steps"] == 5


def test_c3_games_fixed_hint_32768_rows():
    """Games 4-256 trie, fixed-hint processor, beam 16, batch 2048, hint depth 0..3 per group."""
    path = write_index_file("Games", tempfile.mkdtemp())
    depths = np.random.default_rng(5).integers(0, 4, size=2048).tolist()
    info = _run_config(path, "fixed_hint", B=2048, K=16, dtype=torch.bfloat16, hint_depths=depths, seed=3,
                       sample_groups=12)
    assert info["R"] == 32768 and info["V"] == 152704


def test_c3_seeded_count_processor_agrees_with_fixed_hint():
    """With a uniform hint depth the seeded count-based processor (initial_count = depth) and the
    fixed-hint processor see the same windows: same rankings."""
    path = write_index_file("Games", tempfile.mkdtemp())
    a = _run_config(path, "count", B=256, K=16, dtype=torch.bfloat16, hint_depths=[2] * 256, initial_count=2, seed=4)
    b = _run_config(path, "fixed_hint", B=256, K=16, dtype=torch.bfloat16, hint_depths=[2] * 256, seed=4)
    live = a["scores"] > LIVE
    assert torch.equal(a["scores"], b["scores"])
    assert torch.equal(a["sequences"][live], b["sequences"][live])
    assert torch.equal(a["beam_indices"][live], b["beam_indices"][live])


def test_evaluation_style_50_beams_arts_length_penalty_0():
    """evaluate.py's settings: 50 beams, length_penalty 0 (evaluate.sh:36), the widest real trie (Arts,
    fan-out 242): the beam step takes its wide-group route (K > 16, 100 kept candidates)."""
    path = write_index_file("Arts", tempfile.mkdtemp())
    info = _run_config(path, "count", B=96, K=50, dtype=torch.float32, seed=6, sample_groups=3, length_penalty=0.0)
    assert info["R"] == 4800 and info["V"] == 152704 and info["cmax"] == 242


@pytest.mark.parametrize("cb", [32, 64, 128, 512])
def test_c4_codebook_sweep_65536_rows(cb):
    """Synthetic 4-level tries with codebook sizes 32..512 (root fan-out = codebook), beam 16, batch 4096."""
    info = _run_config(_synth_index(cb), "count", B=4096, K=16, dtype=torch.bfloat16, seed=10 + cb, sample_groups=4)
    want_v = {32: 151936, 64: 151936, 128: 152192, 512: 153728}[cb]
    assert info["R"] == 65536 and info["V"] == want_v and info["cmax"] == cb


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_c2_dense_processor_surface_8192_rows(dtype):
    """The dense [R, V] result an unmodified transformers loop receives, at the benchmark's full size
    (8192 x 152,640): finite exactly at the compact candidates' tokens, with their log-probs (rounded to
    the output type), -inf everywhere else; the same call again into a caller-owned fp32 buffer."""
    from genrec_amd.synth import sample_decode_rows

    dev = _dev()
    path = write_index_file("Instruments", tempfile.mkdtemp())
    K, B = 16, 512
    tok, trie, suffix, pidx, proc = _build(path, K, "count")
    V, R = tok.vocab_size_padded, B * K
    ids = torch.from_numpy(sample_decode_rows(trie, B, K, 1, prompt_len=61, seed=3)).to(dev)
    x = _fill_logits(R, V, dtype, dev, 77)
    proc.count = 1
    cand = proc.candidates(ids, x)
    proc.count = 1
    out = proc(ids, x)
    assert out.shape == x.shape and out.dtype == dtype
    cnt = cand.count.long()
    assert int(cnt.min()) >= 1
    valid = torch.arange(cand.cmax, device=dev)[None, :] < cnt[:, None]
    tokens = torch.where(valid, cand.tokens.long(), 0)  # entries past a row's count are never written

    def check(result, tol):
        for r0 in range(0, R, 1024):  # chunks bound the boolean temporaries
            sl = slice(r0, r0 + 1024)
            assert torch.equal(torch.isfinite(result[sl]).sum(dim=1), cnt[sl]), "allowed-token sets differ"
            got = result[sl].gather(1, tokens[sl]).float()
            want = cand.logp[sl]
            err = ((got - want).abs() / tol(want))[valid[sl]]
            assert float(err.max()) <= 1.0
        assert bool((result[~torch.isfinite(result)] == float("-inf")).all())

    if dtype == torch.bfloat16:
        check(out, lambda w: w.abs() * 2.0 ** -8 + 1e-6)  # one bf16 rounding of the fp32 log-prob
    else:
        check(out, lambda w: torch.full_like(w, 4e-6))    # same arithmetic, different summation order
    del out
    proc.count = 1
    wide = torch.empty((R, V), dtype=torch.float32, device=dev)
    assert proc.call_into(ids, x, wide) is wide
    check(wide, lambda w: torch.full_like(w, 4e-6))
    proc.check_errors()
